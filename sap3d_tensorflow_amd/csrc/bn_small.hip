// BatchNorm for SMALL tensors (M = B*D*H*W <= 1024 rows: every layer of stage 3 at batch 8) on
// gfx950: statistics + normalise + ReLU + residual in ONE launch, and the whole backward in ONE
// launch.  tf.layers.batch_normalization on rank-5 input is per-channel, so a block that owns 8
// channels needs no other block: it keeps its [M x 8] slab in REGISTERS (<= 8 float4 per thread per
// tensor), reduces with wave shuffles + a 4-entry LDS exchange, and writes the result.  No atomics,
// no statistics arena, no finalize launch; variance is the two-pass form TF's tf.nn.moments uses.
// Modes are those of bn_apply_kernel (p3d_kernels.h); reference p3d.py:56-81,88,114,127,133-134.
#include "p3d_kernels.h"
#include <cstdlib>

namespace {

constexpr int CB = 8;          // smallest channel slab a block may own (dispatch picks 8 or 16)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float4 gate4(float4 g, float4 pre) {
    return make_float4(pre.x > 0.f ? g.x : 0.f, pre.y > 0.f ? g.y : 0.f, pre.z > 0.f ? g.z : 0.f, pre.w > 0.f ? g.w : 0.f);
}
__device__ __forceinline__ float4 shfl4(float4 v, int o) {
    return make_float4(__shfl_xor(v.x, o), __shfl_xor(v.y, o), __shfl_xor(v.z, o), __shfl_xor(v.w, o));
}

// Sums of `v` and `w` over all threads of the block that share (threadIdx.x % G); both broadcast back.
template <int G>
__device__ __forceinline__ void block_sum2(float4& v, float4& w, float4* xch /*[2][4][G][2]*/, int phase) {
#pragma unroll
    for (int o = G; o < 64; o <<= 1) { v = add4(v, shfl4(v, o)); w = add4(w, shfl4(w, o)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = threadIdx.x % G;
    float4* slot = xch + (phase & 1) * 8 * G;
    if (lane < G) { slot[(wave * G + g) * 2] = v; slot[(wave * G + g) * 2 + 1] = w; }
    __syncthreads();
    v = add4(add4(slot[g * 2], slot[(G + g) * 2]), add4(slot[(2 * G + g) * 2], slot[(3 * G + g) * 2]));
    w = add4(add4(slot[g * 2 + 1], slot[(G + g) * 2 + 1]), add4(slot[(2 * G + g) * 2 + 1], slot[(3 * G + g) * 2 + 1]));
}

// One-pass mean / biased variance of the slab held in registers, shifted by the channel's first row so that
// E[d^2] - E[d]^2 does not cancel (d = x - x[0]).
template <int G, int MJ>
__device__ __forceinline__ void bn_moments(const float4 (&v)[MJ], const float4 shift, int nj, int M, int rowslot, float4* xch,
                                           int& phase, float4& mean, float4& var) {
    constexpr int RS = 256 / G;
    float4 s = f4(0.f), q = f4(0.f);
#pragma unroll
    for (int j = 0; j < MJ; ++j)
        if (j < nj && rowslot + RS * j < M) { const float4 d = sub4(v[j], shift); s = add4(s, d); q = fma4(d, d, q); }
    block_sum2<G>(s, q, xch, phase++);
    const float invM = 1.f / (float)M;
    const float4 md = mul4(s, f4(invM));
    mean = add4(shift, md);
    var = sub4(mul4(q, f4(invM)), mul4(md, md));
    var = make_float4(fmaxf(var.x, 0.f), fmaxf(var.y, 0.f), fmaxf(var.z, 0.f), fmaxf(var.w, 0.f));
}

// The per-channel parameters a block needs, loaded TOGETHER WITH its data: read where they are used, behind the block-wide
// reduction, each of them is one more dependent trip to memory (~1.5 us of a 6-9 us launch).
struct BnChan { float4 gamma, beta, mm, mv; };
__device__ __forceinline__ BnChan bn_chan_load(const BnParams& bn, int c, bool need_moving) {
    BnChan p;
    p.gamma = ld4(bn.gamma + c); p.beta = ld4(bn.beta + c);
    p.mm = f4(0.f); p.mv = f4(1.f);
    if (need_moving) { p.mm = ld4(bn.moving_mean + c); p.mv = ld4(bn.moving_var + c); }
    return p;
}
__device__ __forceinline__ void bn_coeffs(const BnParams& bn, const BnChan& ch, int c, bool use_batch, bool update_moving, float eps, bool writer,
                                          float4& mean, float4& var, float4& scale, float4& shift) {
    if (!use_batch) { mean = ch.mm; var = ch.mv; }
    const float4 inv = make_float4(1.f / sqrtf(var.x + eps), 1.f / sqrtf(var.y + eps), 1.f / sqrtf(var.z + eps), 1.f / sqrtf(var.w + eps));
    scale = mul4(ch.gamma, inv);
    shift = sub4(ch.beta, mul4(mean, scale));
    if (writer) {
        st4(bn.scale + c, scale); st4(bn.shift + c, shift); st4(bn.mean + c, mean); st4(bn.invstd + c, inv);
        if (use_batch && update_moving) {     // moving -= (moving - batch) * (1 - 0.99), biased variance (Appendix A.4)
            st4(bn.moving_mean + c, sub4(ch.mm, mul4(sub4(ch.mm, mean), f4(1.0f - 0.99f))));
            st4(bn.moving_var + c, sub4(ch.mv, mul4(sub4(ch.mv, var), f4(1.0f - 0.99f))));
        }
    }
}

template <int MODE, int CBW>
__global__ __launch_bounds__(256) void bn_small_fwd_kernel(BnSmallArgs a) {
    P3D_CHAIN_PRIO();
    p3d_warm_kernargs<BnSmallArgs>();
    constexpr bool TWO = (MODE == 2 || MODE == 3);
    constexpr int G = CBW / 4, RS = 256 / G, MJ = 1024 / RS;
    __shared__ float4 xch[16 * G];
    const int c = blockIdx.x * CBW + (threadIdx.x % G) * 4;
    const int rowslot = threadIdx.x / G;
    const int nj = (a.M + RS - 1) / RS;
    const BnChan ch1 = bn_chan_load(a.bn1, c, !a.batch1 || (a.update_moving && rowslot == 0));
    BnChan ch2 = ch1;
    if (TWO) ch2 = bn_chan_load(a.bn2, c, !a.batch2 || (a.update_moving && rowslot == 0));
    float4 v1[MJ], v2[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int row = rowslot + RS * j;
        v1[j] = f4(0.f); v2[j] = f4(0.f);
        if (j < nj && row < a.M) {
            v1[j] = ld4(a.y1 + (long long)row * a.ld1 + c);
            if (MODE != 0) v2[j] = ld4(a.y2 + (long long)row * a.ld2 + c);
        }
    }
    int phase = 0;
    float4 mean1 = f4(0.f), var1 = f4(1.f), sc1, sh1, mean2 = f4(0.f), var2 = f4(1.f), sc2 = f4(0.f), sh2 = f4(0.f);
    if (a.batch1) bn_moments<G, MJ>(v1, ld4(a.y1 + c), nj, a.M, rowslot, xch, phase, mean1, var1);
    bn_coeffs(a.bn1, ch1, c, a.batch1, a.update_moving, a.eps, rowslot == 0, mean1, var1, sc1, sh1);
    if (TWO) {
        if (a.batch2) bn_moments<G, MJ>(v2, ld4(a.y2 + c), nj, a.M, rowslot, xch, phase, mean2, var2);
        bn_coeffs(a.bn2, ch2, c, a.batch2, a.update_moving, a.eps, rowslot == 0, mean2, var2, sc2, sh2);
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int row = rowslot + RS * j;
        if (j < nj && row < a.M) {
            const float4 v = fma4(sc1, v1[j], sh1);
            float4 z;
            if (MODE == 0) z = relu4(v);
            else if (MODE == 1) z = relu4(add4(v, v2[j]));
            else if (MODE == 2) z = relu4(add4(v, fma4(sc2, v2[j], sh2)));
            else if (MODE == 3) z = add4(relu4(v), relu4(fma4(sc2, v2[j], sh2)));
            else z = add4(v2[j], relu4(v));
            st4(a.z + (long long)row * a.ldz + c, z);
        }
    }
}

template <int MODE, int CBW>
__global__ __launch_bounds__(256) void bn_small_bwd_kernel(BnSmallArgs a) {
    P3D_CHAIN_PRIO();
    p3d_warm_kernargs<BnSmallArgs>();
    constexpr bool TWO = (MODE == 2 || MODE == 3);
    constexpr int G = CBW / 4, RS = 256 / G, MJ = 1024 / RS;
    __shared__ float4 xch[16 * G];
    const int c = blockIdx.x * CBW + (threadIdx.x % G) * 4;
    const int rowslot = threadIdx.x / G;
    const int nj = (a.M + RS - 1) / RS;
    const float4 sc1 = ld4(a.bn1.scale + c), sh1 = ld4(a.bn1.shift + c), m1 = ld4(a.bn1.mean + c), i1 = ld4(a.bn1.invstd + c);
    float4 sc2 = f4(0.f), sh2 = f4(0.f), m2 = f4(0.f), i2 = f4(0.f);
    if (TWO) { sc2 = ld4(a.bn2.scale + c); sh2 = ld4(a.bn2.shift + c); m2 = ld4(a.bn2.mean + c); i2 = ld4(a.bn2.invstd + c); }
    const float4 gam1 = ld4(a.bn1.gamma + c);            // with the data, not behind the reduction (one dependent trip to memory less)
    float4 gam2 = f4(0.f);
    if (TWO) gam2 = ld4(a.bn2.gamma + c);
    float4 g1[MJ], xh1[MJ], g2[MJ], xh2[TWO ? MJ : 1];
    float4 s1 = f4(0.f), sx1 = f4(0.f), s2 = f4(0.f), sx2 = f4(0.f);
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int row = rowslot + RS * j;
        g1[j] = f4(0.f); xh1[j] = f4(0.f); g2[j] = f4(0.f);
        if (TWO) xh2[TWO ? j : 0] = f4(0.f);
        if (j < nj && row < a.M) {
            const float4 dz = ld4(a.dz + (long long)row * a.lddz + c);
            const float4 y1 = ld4(a.y1 + (long long)row * a.ld1 + c);
            const float4 v1 = fma4(sc1, y1, sh1);
            xh1[j] = mul4(sub4(y1, m1), i1);
            if (MODE == 0) g1[j] = gate4(dz, v1);
            else {
                const float4 y2 = ld4(a.y2 + (long long)row * a.ld2 + c);
                if (MODE == 1) { g1[j] = gate4(dz, add4(v1, y2)); g2[j] = g1[j]; }
                else if (MODE == 4) { g1[j] = gate4(dz, v1); g2[j] = dz; }
                else {
                    const float4 v2 = fma4(sc2, y2, sh2);
                    xh2[TWO ? j : 0] = mul4(sub4(y2, m2), i2);
                    if (MODE == 2) { g1[j] = gate4(dz, add4(v1, v2)); g2[j] = g1[j]; }
                    else { g1[j] = gate4(dz, v1); g2[j] = gate4(dz, v2); }
                }
            }
            s1 = add4(s1, g1[j]); sx1 = fma4(g1[j], xh1[j], sx1);
            if (TWO) { s2 = add4(s2, g2[j]); sx2 = fma4(g2[j], xh2[TWO ? j : 0], sx2); }
        }
    }
    int phase = 0;
    block_sum2<G>(s1, sx1, xch, phase++);
    if (TWO) block_sum2<G>(s2, sx2, xch, phase++);
    if (rowslot == 0) {
        st4(a.dbeta1 + c, s1); st4(a.dgamma1 + c, sx1);
        if (TWO) { st4(a.dbeta2 + c, s2); st4(a.dgamma2 + c, sx2); }
    }
    const float invM = 1.f / (float)a.M;
    const float4 k1 = mul4(gam1, i1);
    const float4 c1 = mul4(s1, f4(invM)), cx1 = mul4(sx1, f4(invM));
    float4 k2 = f4(0.f), c2 = f4(0.f), cx2 = f4(0.f);
    if (TWO) { k2 = mul4(gam2, i2); c2 = mul4(s2, f4(invM)); cx2 = mul4(sx2, f4(invM)); }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int row = rowslot + RS * j;
        if (j < nj && row < a.M) {
            float4 d = a.batch1 ? mul4(k1, sub4(sub4(g1[j], c1), mul4(xh1[j], cx1))) : mul4(k1, g1[j]);
            st4(a.dy1 + (long long)row * a.lddy1 + c, d);
            if (MODE != 0) {
                float4 e;
                if (TWO) e = a.batch2 ? mul4(k2, sub4(sub4(g2[j], c2), mul4(xh2[TWO ? j : 0], cx2))) : mul4(k2, g2[j]);
                else e = g2[j];
                float* dst = a.dy2 + (long long)row * a.lddy2 + c;
                if (a.acc2) e = add4(e, ld4(dst));
                st4(dst, e);
            }
        }
    }
}

template <int CBW>
hipError_t launch_small(const BnSmallArgs& a, bool bwd, hipStream_t s) {
    const dim3 g(a.C / CBW), b(256);
#define P3D_SM(M_)                                                                          \
    case M_:                                                                                \
        if (bwd) hipLaunchKernelGGL((bn_small_bwd_kernel<M_, CBW>), g, b, 0, s, a);         \
        else hipLaunchKernelGGL((bn_small_fwd_kernel<M_, CBW>), g, b, 0, s, a);             \
        break;
    switch (a.mode) {
        P3D_SM(0) P3D_SM(1) P3D_SM(2) P3D_SM(3) P3D_SM(4)
        default: return hipErrorInvalidValue;
    }
#undef P3D_SM
    return hipGetLastError();
}

}  // namespace

bool p3d_bn_small_ok(long M, int C) { return M <= 1024 && (C % CB) == 0; }

// wide tensors get 16-channel slabs (64-byte row segments), narrow ones 8-channel slabs (more blocks)
static hipError_t dispatch(const BnSmallArgs& a, bool bwd, hipStream_t s) {
    if (!p3d_bn_small_ok(a.M, a.C)) return hipErrorInvalidValue;
    static const int forced = p3d_tune_env("P3D_BN_CB") ? atoi(p3d_tune_env("P3D_BN_CB")) : 0;       // tuning: 4, 8 or 16 channels per block
    if (forced == 4) return launch_small<4>(a, bwd, s);
    if (forced == 8) return launch_small<8>(a, bwd, s);
    if (forced == 16 && a.C % 16 == 0) return launch_small<16>(a, bwd, s);
    const bool wide = a.C >= 512 && a.C % 16 == 0 && a.mode != 3 && a.mode != 2;
    return wide ? launch_small<16>(a, bwd, s) : launch_small<8>(a, bwd, s);
}
hipError_t p3d_bn_small_fwd(const BnSmallArgs& a, hipStream_t s) { return dispatch(a, false, s); }
hipError_t p3d_bn_small_bwd(const BnSmallArgs& a, hipStream_t s) { return dispatch(a, true, s); }
