// HBM-bound passes of the P3D path on gfx950: BatchNorm finalize / fused normalise+ReLU+add
// passes and their backward, SAME max-pooling, Smooth-L1 loss, Adam.  All are float4-vectorised
// over the channel axis of NDHWC rows (C % 4 == 0) with explicit row strides so they operate in
// place on channel slices of the decoder's concat buffers (tf.concat at reference
// p3d.py:203,208 never materialises).
#include "p3d_kernels.h"
#include "det_reduce.h"
#define P3D_SEED(a) ((a).seed_dev ? *(a).seed_dev : (a).seed)   // wave-uniform; device-resident under graph replay

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float4 gate4(float4 g, float4 pre) {   // g where pre > 0 (tf.nn.relu gradient)
    return make_float4(pre.x > 0.f ? g.x : 0.f, pre.y > 0.f ? g.y : 0.f, pre.z > 0.f ? g.z : 0.f, pre.w > 0.f ? g.w : 0.f);
}

// counter-based uniform in [0,1): splitmix64 finaliser of (seed, element index)
__device__ __forceinline__ float u01(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float4 dropmask4(unsigned long long seed, long long e0, float rate, float scale) {
    return make_float4(u01(seed, e0) >= rate ? scale : 0.f, u01(seed, e0 + 1) >= rate ? scale : 0.f,
                       u01(seed, e0 + 2) >= rate ? scale : 0.f, u01(seed, e0 + 3) >= rate ? scale : 0.f);
}

// ------------------------------------------------------------------------------------------------
// Folds of 16 replicas / partials with shuffles (one load latency instead of a chain of 16); lane 0 of the group
// finishes the channel.  These tiny kernels sit on the critical path ~100 times per step.
static_assert(P3D_STAT_REPLICAS == 16, "finalize kernels fold 16 replicas with 4 shuffle steps");
__device__ __forceinline__ double fold16(double v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// BatchNorm finalize.  LPC lanes per channel (16 or 64): lane r adds partials r, r + LPC, ... in double, in that order,
// then the lanes are folded by a fixed shuffle tree -- the statistics are bit-reproducible (the partials are plain
// stores of the producer's epilogue, not atomics).
template <int LPC>
__global__ __launch_bounds__(256) void bn_finalize_kernel(BnParams bn, double invM, int use_batch, int update_moving, float eps) {
    P3D_CHAIN_PRIO();
    const int r = threadIdx.x % LPC;
    const int c = blockIdx.x * (256 / LPC) + threadIdx.x / LPC;
    const bool ok = c < bn.C;
    double s1 = 0.0, s2 = 0.0;
    if (use_batch) {
        if (ok) {
            const float2* part = reinterpret_cast<const float2*>(bn.statpart) + c;
#pragma unroll 4
            for (int q = r; q < bn.nparts; q += LPC) {
                const float2 v = part[(size_t)q * bn.C];
                s1 += (double)v.x; s2 += (double)v.y;
            }
        }
#pragma unroll
        for (int o = LPC / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    }
    if (!ok || r) return;
    double mean, var;
    if (use_batch) {
        mean = s1 * invM;
        var = s2 * invM - mean * mean;
        if (var < 0.0) var = 0.0;
        if (update_moving) {      // moving -= (moving - batch) * (1 - 0.99)   (biased variance, Appendix A.4)
            bn.moving_mean[c] -= (bn.moving_mean[c] - (float)mean) * (1.0f - 0.99f);
            bn.moving_var[c] -= (bn.moving_var[c] - (float)var) * (1.0f - 0.99f);
        }
    } else {
        mean = bn.moving_mean[c];
        var = bn.moving_var[c];
    }
    const double inv = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)bn.gamma[c] * inv);
    bn.scale[c] = sc;
    bn.shift[c] = (float)((double)bn.beta[c] - mean * (double)bn.gamma[c] * inv);
    bn.mean[c] = (float)mean;
    bn.invstd[c] = (float)inv;
}

// Per-channel (sum, sumsq) of y.  Thread = one float4 channel group, RPI rows per block pass.  Block b writes partial b.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* y, int ld, long long M, int C, float* statpart) {
    P3D_CHAIN_PRIO();
    __shared__ float red[256][8];
    const int c4n = C >> 2;
    const int rpi = 256 / c4n;
    const int tid = threadIdx.x;
    const int sub = tid / c4n;
    const int cg = tid - sub * c4n;
    const int c = cg << 2;
    float4 s1 = f4(0.f), s2 = f4(0.f);
    if (sub < rpi)
        for (long long row = (long long)blockIdx.x * rpi + sub; row < M; row += (long long)gridDim.x * rpi) {
            const float4 v = ld4(y + row * ld + c);
            s1 = add4(s1, v);
            s2 = fma4(v, v, s2);
        }
    float* r = red[tid];
    r[0] = s1.x; r[1] = s1.y; r[2] = s1.z; r[3] = s1.w; r[4] = s2.x; r[5] = s2.y; r[6] = s2.z; r[7] = s2.w;
    __syncthreads();
    if (tid < c4n) {
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = 0.f;
        for (int s = 0; s < rpi; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] += red[s * c4n + tid][q];
        float* st = statpart + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int q = 0; q < 4; ++q) { st[2 * (c + q) + 0] = t[q]; st[2 * (c + q) + 1] = t[4 + q]; }
    }
}

// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs a) {
    p3d_warm_kernarg_lines<(int)(sizeof(BnApplyArgs) / 64)>();
    P3D_CHAIN_PRIO();
    const int c4n = a.C >> 2;
    const long long total = a.M * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int c = (int)(i - row * c4n) << 2;
        float4 v = fma4(ld4(a.scale1 + c), ld4(a.y1 + row * a.ld1 + c), ld4(a.shift1 + c));
        float4 z;
        if (MODE == 0) z = relu4(v);
        else if (MODE == 1) z = relu4(add4(v, ld4(a.y2 + row * a.ld2 + c)));
        else if (MODE == 2) z = relu4(add4(v, fma4(ld4(a.scale2 + c), ld4(a.y2 + row * a.ld2 + c), ld4(a.shift2 + c))));
        else if (MODE == 3) z = add4(relu4(v), relu4(fma4(ld4(a.scale2 + c), ld4(a.y2 + row * a.ld2 + c), ld4(a.shift2 + c))));
        else z = add4(ld4(a.y2 + row * a.ld2 + c), relu4(v));
        if (a.drop_scale > 0.f) z = mul4(z, dropmask4(P3D_SEED(a), row * a.C + c, a.drop_rate, a.drop_scale));
        st4(a.z + row * a.ldz + c, z);
    }
}

// BatchNorm finalize + apply in ONE launch for mid-size tensors (stage 2: 49-98 statistics partials per channel).  A block
// owns 64 channels x a row chunk: it first folds the partials of ITS channels (4 lanes per channel, double accumulation,
// fixed order -- every row chunk of a channel group computes the same bits), then normalises its rows.  No separate
// bn_finalize launch on the forward chain; the blocks of row chunk 0 publish scale / shift / mean / invstd for the backward
// pass and update the moving statistics.
__device__ __forceinline__ void bn_fold64(const BnParams& bn, int c0, double invM, int use_batch, int update_moving, float eps, bool publish,
                                          float* sc_lds, float* sh_lds) {
    const int ch = c0 + (threadIdx.x >> 2), r = threadIdx.x & 3;
    double s1 = 0.0, s2 = 0.0;
    if (use_batch) {
        const float2* part = reinterpret_cast<const float2*>(bn.statpart) + ch;
#pragma unroll 8
        for (int q = r; q < bn.nparts; q += 4) {
            const float2 v = part[(size_t)q * bn.C];
            s1 += (double)v.x; s2 += (double)v.y;
        }
        s1 += __shfl_xor(s1, 1); s2 += __shfl_xor(s2, 1);
        s1 += __shfl_xor(s1, 2); s2 += __shfl_xor(s2, 2);
    }
    if (r) return;
    double mean, var;
    if (use_batch) {
        mean = s1 * invM;
        var = s2 * invM - mean * mean;
        if (var < 0.0) var = 0.0;
        if (publish && update_moving) {
            bn.moving_mean[ch] -= (bn.moving_mean[ch] - (float)mean) * (1.0f - 0.99f);
            bn.moving_var[ch] -= (bn.moving_var[ch] - (float)var) * (1.0f - 0.99f);
        }
    } else {
        mean = bn.moving_mean[ch];
        var = bn.moving_var[ch];
    }
    const double inv = 1.0 / sqrt(var + (double)eps);
    const float sc = (float)((double)bn.gamma[ch] * inv);
    const float sh = (float)((double)bn.beta[ch] - mean * (double)bn.gamma[ch] * inv);
    sc_lds[ch - c0] = sc; sh_lds[ch - c0] = sh;
    if (publish) { bn.scale[ch] = sc; bn.shift[ch] = sh; bn.mean[ch] = (float)mean; bn.invstd[ch] = (float)inv; }
}

template <int MODE>
__global__ __launch_bounds__(256) void bn_fold_apply_kernel(BnApplyArgs a, BnParams bn1, BnParams bn2, double invM, int batch1, int batch2,
                                                           int update_moving, float eps, int rows_per_block) {
    p3d_warm_kernarg_lines<(int)((sizeof(BnApplyArgs) + 2 * sizeof(BnParams)) / 64)>();
    P3D_CHAIN_PRIO();
    constexpr bool TWO = (MODE == 2 || MODE == 3);
    __shared__ __attribute__((aligned(16))) float sc1[64], sh1[64], sc2[64], sh2[64];
    const int c0 = blockIdx.y * 64;
    const bool publish = blockIdx.x == 0;
    bn_fold64(bn1, c0, invM, batch1, update_moving, eps, publish, sc1, sh1);
    if (TWO) bn_fold64(bn2, c0, invM, batch2, update_moving, eps, publish, sc2, sh2);
    __syncthreads();
    const int cl = (threadIdx.x & 15) << 2, c = c0 + cl;
    const float4 s1v = ld4(sc1 + cl), h1v = ld4(sh1 + cl);
    float4 s2v = f4(0.f), h2v = f4(0.f);
    if (TWO) { s2v = ld4(sc2 + cl); h2v = ld4(sh2 + cl); }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < a.M ? r0 + rows_per_block : a.M;
    for (long long row = r0 + (threadIdx.x >> 4); row < r1; row += 16) {
        const float4 v = fma4(s1v, ld4(a.y1 + row * a.ld1 + c), h1v);
        float4 z;
        if (MODE == 0) z = relu4(v);
        else if (MODE == 1) z = relu4(add4(v, ld4(a.y2 + row * a.ld2 + c)));
        else if (MODE == 2) z = relu4(add4(v, fma4(s2v, ld4(a.y2 + row * a.ld2 + c), h2v)));
        else if (MODE == 3) z = add4(relu4(v), relu4(fma4(s2v, ld4(a.y2 + row * a.ld2 + c), h2v)));
        else z = add4(ld4(a.y2 + row * a.ld2 + c), relu4(v));
        st4(a.z + row * a.ldz + c, z);
    }
}

// gradient entering BN1 / BN2 (after the ReLU gates) for one float4 of one row
template <int MODE>
__device__ __forceinline__ void bn_bwd_gates(const BnBwdArgs& a, long long row, int c, float4& g1, float4& g2,
                                             float4& y1, float4& y2) {
    float4 dz = ld4(a.dz + row * a.lddz + c);
    if (a.drop_scale > 0.f) dz = mul4(dz, dropmask4(P3D_SEED(a), row * a.C + c, a.drop_rate, a.drop_scale));
    y1 = ld4(a.y1 + row * a.ld1 + c);
    const float4 v1 = fma4(ld4(a.scale1 + c), y1, ld4(a.shift1 + c));
    y2 = f4(0.f);
    if (MODE == 0) { g1 = gate4(dz, v1); g2 = f4(0.f); }
    else if (MODE == 1) { y2 = ld4(a.y2 + row * a.ld2 + c); g1 = gate4(dz, add4(v1, y2)); g2 = g1; }
    else if (MODE == 2) {
        y2 = ld4(a.y2 + row * a.ld2 + c);
        const float4 v2 = fma4(ld4(a.scale2 + c), y2, ld4(a.shift2 + c));
        g1 = gate4(dz, add4(v1, v2)); g2 = g1;
    } else if (MODE == 3) {
        y2 = ld4(a.y2 + row * a.ld2 + c);
        const float4 v2 = fma4(ld4(a.scale2 + c), y2, ld4(a.shift2 + c));
        g1 = gate4(dz, v1); g2 = gate4(dz, v2);
    } else { g1 = gate4(dz, v1); g2 = dz; }
}

// Per-channel sums of g and g*xhat.  Thread = one float4 channel group, RPI rows per block pass.
template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBwdArgs a) {
    p3d_warm_kernarg_lines<(int)(sizeof(BnBwdArgs) / 64)>();
    P3D_CHAIN_PRIO();
    constexpr bool TWO = (MODE == 2 || MODE == 3);
    __shared__ float red[256][TWO ? 16 : 8];
    const int c4n = a.C >> 2;
    const int rpi = 256 / c4n;                 // rows handled per block iteration (c4n <= 256)
    const int tid = threadIdx.x;
    const int sub = tid / c4n;
    const int cg = tid - sub * c4n;
    const int c = cg << 2;
    const bool active = sub < rpi;
    float4 s1 = f4(0.f), sx1 = f4(0.f), s2 = f4(0.f), sx2 = f4(0.f);
    if (active) {
        const float4 m1 = ld4(a.mean1 + c), i1 = ld4(a.invstd1 + c);
        float4 m2 = f4(0.f), i2 = f4(0.f);
        if (TWO) { m2 = ld4(a.mean2 + c); i2 = ld4(a.invstd2 + c); }
        // two rows per trip, a grid stride apart, on two accumulator sets: twice the loads in flight per thread (the large tensors --
        // stem / deconv3: 401 408 rows on 512 blocks -- ran at 2.5 TB/s on one row per trip), a fixed order all the same
        const long long stride = (long long)gridDim.x * rpi;
        float4 t1 = f4(0.f), tx1 = f4(0.f), t2 = f4(0.f), tx2 = f4(0.f);
        long long row = (long long)blockIdx.x * rpi + sub;
        for (; row + stride < a.M; row += 2 * stride) {
            float4 g1, g2, y1, y2, h1, h2, z1, z2;
            bn_bwd_gates<MODE>(a, row, c, g1, g2, y1, y2);
            bn_bwd_gates<MODE>(a, row + stride, c, h1, h2, z1, z2);
            s1 = add4(s1, g1);
            sx1 = fma4(g1, mul4(sub4(y1, m1), i1), sx1);
            t1 = add4(t1, h1);
            tx1 = fma4(h1, mul4(sub4(z1, m1), i1), tx1);
            if (TWO) {
                s2 = add4(s2, g2);
                sx2 = fma4(g2, mul4(sub4(y2, m2), i2), sx2);
                t2 = add4(t2, h2);
                tx2 = fma4(h2, mul4(sub4(z2, m2), i2), tx2);
            }
        }
        if (row < a.M) {
            float4 g1, g2, y1, y2;
            bn_bwd_gates<MODE>(a, row, c, g1, g2, y1, y2);
            s1 = add4(s1, g1);
            sx1 = fma4(g1, mul4(sub4(y1, m1), i1), sx1);
            if (TWO) {
                s2 = add4(s2, g2);
                sx2 = fma4(g2, mul4(sub4(y2, m2), i2), sx2);
            }
        }
        s1 = add4(s1, t1); sx1 = add4(sx1, tx1);
        if (TWO) { s2 = add4(s2, t2); sx2 = add4(sx2, tx2); }
    }
    float* r = red[tid];
    r[0] = s1.x; r[1] = s1.y; r[2] = s1.z; r[3] = s1.w; r[4] = sx1.x; r[5] = sx1.y; r[6] = sx1.z; r[7] = sx1.w;
    if (TWO) { r[8] = s2.x; r[9] = s2.y; r[10] = s2.z; r[11] = s2.w; r[12] = sx2.x; r[13] = sx2.y; r[14] = sx2.z; r[15] = sx2.w; }
    __syncthreads();
    if (tid < c4n) {
        constexpr int NV = TWO ? 16 : 8;
        float t[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) t[q] = 0.f;
        for (int s = 0; s < rpi; ++s)
#pragma unroll
            for (int q = 0; q < NV; ++q) t[q] += red[s * c4n + tid][q];
        // block b's partial sums, plain stores (bit-reproducible; folded in block order by bn_bwd_finalize)
        float* p1 = a.part1 + ((size_t)blockIdx.x * a.C + c) * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) { p1[2 * q] = t[q]; p1[2 * q + 1] = t[4 + q]; }
        if (TWO) {
            float* p2 = a.part2 + ((size_t)blockIdx.x * a.C + c) * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) { p2[2 * q] = t[8 + q]; p2[2 * q + 1] = t[12 + q]; }
        }
    }
}

// Folds the per-block partial sums in block order (double accumulation, fixed shuffle tree): coef[c] = (sum g / M,
// sum g*xhat / M) and the BN parameter gradients (each BN parameter is produced exactly once per step, so they are
// written, not accumulated).
// LPC lanes per channel (16, or 64 when there are many partials: 100-500 per channel in stages 1-2, where 16 lanes walk them in
// 6-8 dependent batches of loads -- this kernel sits between the reduce and the apply pass of every large BatchNorm backward).
template <int LPC>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(BnBwdArgs a, int two) {
    p3d_warm_kernarg_lines<(int)(sizeof(BnBwdArgs) / 64)>();
    P3D_CHAIN_PRIO();
    const int r = threadIdx.x % LPC;
    const int c = blockIdx.x * (256 / LPC) + threadIdx.x / LPC;
    const bool ok = c < a.C;
    const double invM = 1.0 / (double)a.M;
    double s1 = 0.0, s2 = 0.0, t1 = 0.0, t2 = 0.0;
    if (ok) {
        const float2* p1 = reinterpret_cast<const float2*>(a.part1) + c;
        const float2* p2 = reinterpret_cast<const float2*>(a.part2) + c;
#pragma unroll 4
        for (int q = r; q < a.nparts; q += LPC) {
            const float2 v = p1[(size_t)q * a.C];
            s1 += (double)v.x; s2 += (double)v.y;
            if (two) { const float2 w = p2[(size_t)q * a.C]; t1 += (double)w.x; t2 += (double)w.y; }
        }
    }
#pragma unroll
    for (int o = LPC / 2; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o);
        if (two) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
    }
    if (!ok || r) return;
    a.dbeta1[c] = (float)s1; a.dgamma1[c] = (float)s2;
    a.coef1[2 * c] = (float)(s1 * invM); a.coef1[2 * c + 1] = (float)(s2 * invM);
    if (two) {
        a.dbeta2[c] = (float)t1; a.dgamma2[c] = (float)t2;
        a.coef2[2 * c] = (float)(t1 * invM); a.coef2[2 * c + 1] = (float)(t2 * invM);
    }
}

// Fused BatchNorm backward (conv_igemm2.hip, bn_grad_fold_channel) for MANY partials: 16 lanes per channel fold the gating
// launch's per-tile (sum g, sum g*xhat) in a fixed order and publish k1 / k2 / k3 and the parameter gradients.
__global__ __launch_bounds__(256) void bn_grad_finalize_kernel(BnGradFold f) {
    P3D_CHAIN_PRIO();
    const int r = threadIdx.x & 15;
    const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = c < f.C;
    double sg = 0.0, sgx = 0.0;
    if (ok) {
        const float2* part = reinterpret_cast<const float2*>(f.part) + c;
#pragma unroll 4
        for (int q = r; q < f.nparts; q += 16) {
            const float2 v = part[(size_t)q * f.C];
            sg += (double)v.x; sgx += (double)v.y;
        }
    }
    sg = fold16(sg); sgx = fold16(sgx);
    if (!ok || r) return;
    const float inv = f.invstd[c];
    const float c1 = (float)(sg * f.inv_m), c2 = (float)(sgx * f.inv_m);
    const float k1 = f.gamma[c] * inv;
    const float k2 = -k1 * inv * c2;
    const float k3 = -k1 * c1 - k2 * f.mean[c];
    f.coef[c] = k1; f.coef[f.C + c] = k2; f.coef[2 * f.C + c] = k3;
    f.dgamma[c] = (float)sgx; f.dbeta[c] = (float)sg;
}

__device__ __forceinline__ void ldcoef(const float* coef, int c, float4& c1, float4& c2) {
    const float4 lo = ld4(coef + 2 * c), hi = ld4(coef + 2 * c + 4);     // (s,x,s,x) (s,x,s,x)
    c1 = make_float4(lo.x, lo.z, hi.x, hi.z);
    c2 = make_float4(lo.y, lo.w, hi.y, hi.w);
}

template <int MODE>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a) {
    p3d_warm_kernarg_lines<(int)(sizeof(BnBwdArgs) / 64)>();
    P3D_CHAIN_PRIO();
    constexpr bool TWO = (MODE == 2 || MODE == 3);
    const int c4n = a.C >> 2;
    const long long total = a.M * c4n;
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long long i = gtid; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int c = (int)(i - row * c4n) << 2;
        float4 g1, g2, y1, y2;
        bn_bwd_gates<MODE>(a, row, c, g1, g2, y1, y2);
        {
            const float4 k = mul4(ld4(a.gamma1 + c), ld4(a.invstd1 + c));
            float4 d;
            if (a.batch1) {
                const float4 xh = mul4(sub4(y1, ld4(a.mean1 + c)), ld4(a.invstd1 + c));
                float4 c1, c2;
                ldcoef(a.coef1, c, c1, c2);
                d = mul4(k, sub4(sub4(g1, c1), mul4(xh, c2)));
            } else d = mul4(k, g1);
            float* dst = a.dy1 + row * a.lddy1 + c;
            if (a.acc1) d = add4(d, ld4(dst));
            st4(dst, d);
        }
        if (MODE != 0) {
            float4 d;
            if (TWO) {
                const float4 k = mul4(ld4(a.gamma2 + c), ld4(a.invstd2 + c));
                if (a.batch2) {
                    const float4 xh = mul4(sub4(y2, ld4(a.mean2 + c)), ld4(a.invstd2 + c));
                    float4 c1, c2;
                    ldcoef(a.coef2, c, c1, c2);
                    d = mul4(k, sub4(sub4(g2, c1), mul4(xh, c2)));
                } else d = mul4(k, g2);
            } else d = g2;            // residual branch (modes 1, 4)
            float* dst = a.dy2 + row * a.lddy2 + c;
            if (a.acc2) d = add4(d, ld4(dst));
            st4(dst, d);
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(PoolArgs a) {
    p3d_warm_kernarg_lines<(int)(sizeof(PoolArgs) / 64)>();
    P3D_CHAIN_PRIO();
    const int c4n = a.C >> 2;
    const long long total = (long long)a.N * a.Do * a.Ho * a.Wo * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long pos = i / c4n;
        const int c = (int)(i - pos * c4n) << 2;
        const long long opos = pos;
        const int ow = (int)(pos % a.Wo); pos /= a.Wo;
        const int oh = (int)(pos % a.Ho); pos /= a.Ho;
        const int od = (int)(pos % a.Do); const int n = (int)(pos / a.Do);
        float4 best = f4(-INFINITY);
        unsigned bx = 0, by = 0, bz = 0, bw = 0;          // tap of the first maximum, per channel
        for (int kd = 0; kd < a.kd; ++kd) {
            const int id = od * a.sd - a.pd + kd;
            if ((unsigned)id >= (unsigned)a.Di) continue;
            for (int kh = 0; kh < a.kh; ++kh) {
                const int ih = oh * a.sh - a.ph + kh;
                if ((unsigned)ih >= (unsigned)a.Hi) continue;
                for (int kw = 0; kw < a.kw; ++kw) {
                    const int iw = ow * a.sw - a.pw + kw;
                    if ((unsigned)iw >= (unsigned)a.Wi) continue;
                    const float4 v = ld4(a.x + ((((long long)n * a.Di + id) * a.Hi + ih) * a.Wi + iw) * a.ldx + c);
                    const unsigned t = (unsigned)((kd * a.kh + kh) * a.kw + kw);
                    if (v.x > best.x) { best.x = v.x; bx = t; }
                    if (v.y > best.y) { best.y = v.y; by = t; }
                    if (v.z > best.z) { best.z = v.z; bz = t; }
                    if (v.w > best.w) { best.w = v.w; bw = t; }
                }
            }
        }
        st4(a.y + opos * a.ldy + c, best);
        if (a.idx) a.idx[opos * c4n + (c >> 2)] = bx | (by << 8) | (bz << 16) | (bw << 24);
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* pred, const float* target, long long n,
                                                        double* loss_out, float* dl, int through_sigmoid, double* part,
                                                        unsigned* counter) {
    P3D_CHAIN_PRIO();
    __shared__ double wsum[4];
    __shared__ int last_flag;
    double acc = 0.0;
    auto one = [&](float p, float t, float& g) {
        const float d = p - t;
        const float ad = fabsf(d);
        acc += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
        g = ad < 1.f ? d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        if (through_sigmoid) g *= p * (1.f - p);
    };
    const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long long)gridDim.x * blockDim.x;
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(dl)) & 15) == 0) {
        // 16 bytes per lane and tensor (the scalar form ran at 0.86 TB/s: 22 us for 19 MB on the main stream)
        const long long n4 = n >> 2;
        for (long long i = gtid; i < n4; i += gsz) {
            const float4 p = ld4(pred + 4 * i), t = ld4(target + 4 * i);
            float4 g;
            one(p.x, t.x, g.x); one(p.y, t.y, g.y); one(p.z, t.z, g.z); one(p.w, t.w, g.w);
            st4(dl + 4 * i, g);
        }
    } else {
        for (long long i = gtid; i < n; i += gsz) one(pred[i], target[i], dl[i]);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    // per-block partial, then the last arriving block adds the partials in block order (no atomics: the loss is
    // bit-reproducible)
    if (threadIdx.x == 0) p3d_store_wt(part, blockIdx.x, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    if (!p3d_last_block_wt(counter, gridDim.x, &last_flag)) return;
    double t = 0.0;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += part[b];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) *loss_out += wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long long n4,
                                                   long long n, float lr_arg, const float* lr_dev, float b1, float b2, float eps) {
    const float lr_t = lr_dev ? *lr_dev : lr_arg;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i << 2;
        if (e + 3 < n) {
            const float4 gg = ld4(g + e);
            float4 mm = ld4(m + e), vv = ld4(v + e), pp = ld4(p + e);
            const float gs[4] = {gg.x, gg.y, gg.z, gg.w};
            float ms[4] = {mm.x, mm.y, mm.z, mm.w}, vs[4] = {vv.x, vv.y, vv.z, vv.w}, ps[4] = {pp.x, pp.y, pp.z, pp.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ms[q] = b1 * ms[q] + (1.f - b1) * gs[q];
                vs[q] = b2 * vs[q] + (1.f - b2) * gs[q] * gs[q];
                ps[q] -= lr_t * ms[q] / (sqrtf(vs[q]) + eps);
            }
            st4(m + e, make_float4(ms[0], ms[1], ms[2], ms[3]));
            st4(v + e, make_float4(vs[0], vs[1], vs[2], vs[3]));
            st4(p + e, make_float4(ps[0], ps[1], ps[2], ps[3]));
        } else {
            for (long long q = e; q < n; ++q) {
                const float gq = g[q];
                const float mq = b1 * m[q] + (1.f - b1) * gq;
                const float vq = b2 * v[q] + (1.f - b2) * gq * gq;
                m[q] = mq; v[q] = vq;
                p[q] -= lr_t * mq / (sqrtf(vq) + eps);
            }
        }
    }
}

__global__ __launch_bounds__(256) void add_inplace_kernel(float* dst, int lddst, const float* src, int ldsrc, long long M, int C, int copy) {
    const int c4n = C >> 2;
    const long long total = M * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int c = (int)(i - row * c4n) << 2;
        float4 v = ld4(src + row * ldsrc + c);
        if (!copy) v = add4(v, ld4(dst + row * lddst + c));
        st4(dst + row * lddst + c, v);
    }
}

__global__ __launch_bounds__(256) void fill_uniform_kernel(float* p, long long n, float lo, float hi, unsigned long long seed) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        p[i] = lo + (hi - lo) * u01(seed, (unsigned long long)i);
}

// tf.truncated_normal(stddev): N(0, stddev) re-drawn until it falls within two standard deviations (what
// tf.contrib.layers.variance_scaling_initializer(uniform=False) draws from; utils/network.py:212-213,264).  Counter-based:
// draw k of element i uses the uniform pair (2 (i + k n), 2 (i + k n) + 1).
__global__ __launch_bounds__(256) void fill_trunc_normal_kernel(float* p, long long n, float stddev, unsigned long long seed) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float z = 0.f;
        for (int k = 0; k < 32; ++k) {
            const unsigned long long e = 2ull * (unsigned long long)(i + (long long)k * n);
            const float u1 = fmaxf(u01(seed, e), 1e-12f), u2 = u01(seed, e + 1);
            z = sqrtf(-2.f * logf(u1)) * cosf(6.2831853071795865f * u2);
            if (fabsf(z) <= 2.f) break;
            z = 0.f;                                 // (32 rejections in a row: probability 1e-43)
        }
        p[i] = stddev * z;
    }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* dy, int ld, long long M, int C, float* out, float* part,
                                                     unsigned* counters) {
    // thread = one channel; blocks stride over rows; per-block partials, folded in block order by the last arriver of
    // each channel group (no atomics)
    __shared__ int last_flag;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    float acc = 0.f;
    if (c < C) {
        // row order as before, eight rows' loads in flight
        long long row = blockIdx.x;
        const long long st = gridDim.x;
        for (; row + 7 * st < M; row += 8 * st) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dy[(row + u * st) * ld + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; row < M; row += st) acc += dy[row * ld + c];
        p3d_store_wt(part, (size_t)blockIdx.x * C + c, acc);
    }
    if (!p3d_last_block_wt(counters + blockIdx.y, gridDim.x, &last_flag)) return;
    if (c >= C) return;
    float t = 0.f;
    unsigned b = 0;
    for (; b + 7 < gridDim.x; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u) * C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < gridDim.x; ++b) t += part[(size_t)b * C + c];
    out[c] += t;
}

// Column sums of a [M x C] tensor, C % 4 == 0, C <= 1024: a thread owns one float4 column group of one row slot
// (256 / (C/4) rows per pass, eight independent loads in flight), the block folds its row slots through LDS in slot order,
// and the last arriving block folds the per-block partials in block order -- bit-identical run to run, no atomics.
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ dy, int ld, long long M, int C, float* out,
                                                      float* part, unsigned* counters) {
    __shared__ float4 red[256];
    __shared__ int last_flag;
    const int L = C >> 2, R = 256 / L;
    const int lane = threadIdx.x % L, rs = threadIdx.x / L;
    const bool live = rs < R;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const long long stride = (long long)gridDim.x * R;
        long long row = (long long)blockIdx.x * R + rs;
        const float* base = dy + lane * 4;
        for (; row + 7 * stride < M; row += 8 * stride) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (row + u * stride) * ld);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; row < M; row += stride) {
            const float4 v = *reinterpret_cast<const float4*>(base + row * ld);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < L) {
        float4 t = red[threadIdx.x];
        for (int r = 1; r < R; ++r) { const float4 v = red[r * L + threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        p3d_store_wt4(part, (size_t)blockIdx.x * C + threadIdx.x * 4, t);      // write-through: no release fence below
    }
    if (!p3d_last_block_wt(counters, gridDim.x, &last_flag)) return;
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        // the partials come from other CUs' write-through stores: every load is a long-latency miss, so eight in flight (one at a
        // time this fold was 40 of deconv3's 62 us: 64 dependent trips), added in the same order
        const float* src = part + lane * 4;
        unsigned b = rs;
        for (; b + 7u * R < gridDim.x; b += 8u * R) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(b + u * R) * C);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; b < gridDim.x; b += R) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)b * C);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    __syncthreads();
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < L) {
        float4 t = red[threadIdx.x];
        for (int r = 1; r < R; ++r) { const float4 v = red[r * L + threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        float4* o = reinterpret_cast<float4*>(out + threadIdx.x * 4);
        float4 cur = *o;
        cur.x += t.x; cur.y += t.y; cur.z += t.z; cur.w += t.w;
        *o = cur;
    }
}

inline unsigned grid_for(long long total, int per_block = 256, int cap = 4096) {
    long long b = (total + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace

hipError_t p3d_bn_finalize(const BnParams& bn, long M, int use_batch, int update_moving, float eps, hipStream_t s) {
    if (use_batch && (!bn.statpart || bn.nparts < 1)) return hipErrorInvalidValue;
    if (use_batch && bn.nparts > 64)
        hipLaunchKernelGGL(bn_finalize_kernel<64>, dim3((bn.C + 3) / 4), dim3(256), 0, s, bn, 1.0 / (double)M, use_batch, update_moving, eps);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<16>, dim3((bn.C + 15) / 16), dim3(256), 0, s, bn, 1.0 / (double)M, use_batch, update_moving, eps);
    return hipGetLastError();
}

int p3d_bn_stats_parts(long M, int C) {
    if ((C & 3) || C > 1024 || C < 4) return 0;
    const int rpi = 256 / (C >> 2);
    long long blocks = (M + (long long)rpi * 8 - 1) / ((long long)rpi * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;
    return (int)blocks;
}

hipError_t p3d_bn_stats(const float* y, int ld, long M, int C, float* statpart, hipStream_t s) {
    if ((C & 3) || C > 1024 || (ld & 3)) return hipErrorInvalidValue;
    const long long blocks = p3d_bn_stats_parts(M, C);
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, s, y, ld, (long long)M, C, statpart);
    return hipGetLastError();
}

hipError_t p3d_bn_apply(const BnApplyArgs& a, hipStream_t s) {
    if ((a.C & 3) || (a.ld1 & 3) || (a.ldz & 3)) return hipErrorInvalidValue;
    const unsigned g = grid_for(a.M * (a.C >> 2));
    switch (a.mode) {
        case 0: hipLaunchKernelGGL(bn_apply_kernel<0>, dim3(g), dim3(256), 0, s, a); break;
        case 1: hipLaunchKernelGGL(bn_apply_kernel<1>, dim3(g), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(bn_apply_kernel<2>, dim3(g), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(bn_apply_kernel<3>, dim3(g), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(bn_apply_kernel<4>, dim3(g), dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

bool p3d_bn_fold_apply_ok(long M, int C, int nparts1, int nparts2, float drop_scale) {
    static const bool off = p3d_tune_env("P3D_BN_FOLD_APPLY") && atoi(p3d_tune_env("P3D_BN_FOLD_APPLY")) == 0;      // A/B runs
    return !off && (C % 64) == 0 && nparts1 <= 128 && nparts2 <= 128 && M >= 1024 && !(drop_scale > 0.f);
}
hipError_t p3d_bn_fold_apply(const BnApplyArgs& a, const BnParams& bn1, const BnParams& bn2, int batch1, int batch2, int update_moving,
                             float eps, hipStream_t s) {
    const bool two = a.mode == 2 || a.mode == 3;
    if ((a.ld1 & 3) || (a.ldz & 3) || !p3d_bn_fold_apply_ok(a.M, a.C, batch1 ? bn1.nparts : 0, two && batch2 ? bn2.nparts : 0, a.drop_scale))
        return hipErrorInvalidValue;
    if ((batch1 && (!bn1.statpart || bn1.nparts < 1)) || (two && batch2 && (!bn2.statpart || bn2.nparts < 1))) return hipErrorInvalidValue;
    const int groups = a.C / 64;
    int chunks = 256 / groups;                         // ~256 blocks in all
    if (chunks < 1) chunks = 1;
    long rows = (a.M + chunks - 1) / chunks;
    rows = (rows + 15) / 16 * 16;
    chunks = (int)((a.M + rows - 1) / rows);
    const dim3 g((unsigned)chunks, (unsigned)groups);
    const double invM = 1.0 / (double)a.M;
#define P3D_FA(M_) case M_: hipLaunchKernelGGL(bn_fold_apply_kernel<M_>, g, dim3(256), 0, s, a, bn1, bn2, invM, batch1, batch2, update_moving, eps, (int)rows); break;
    switch (a.mode) { P3D_FA(0) P3D_FA(1) P3D_FA(2) P3D_FA(3) P3D_FA(4) default: return hipErrorInvalidValue; }
#undef P3D_FA
    return hipGetLastError();
}

int p3d_bn_bwd_parts(long M, int C) {
    if ((C & 3) || C > 1024 || C < 4) return 0;
    const int rpi = 256 / (C >> 2);
    long long blocks = (M + (long long)rpi * 8 - 1) / ((long long)rpi * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 512) blocks = 512;       // (1024: no faster on the 401 408-row tensors, 2-6 us slower on stage 1's, finalize +0.6 us)
    return (int)blocks;
}

hipError_t p3d_bn_bwd_reduce(const BnBwdArgs& a, hipStream_t s) {
    if ((a.C & 3) || a.C > 1024 || !a.part1 || a.nparts != p3d_bn_bwd_parts(a.M, a.C)) return hipErrorInvalidValue;
    const unsigned g = (unsigned)a.nparts;
    switch (a.mode) {
        case 0: hipLaunchKernelGGL(bn_bwd_reduce_kernel<0>, dim3(g), dim3(256), 0, s, a); break;
        case 1: hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(g), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(bn_bwd_reduce_kernel<2>, dim3(g), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(bn_bwd_reduce_kernel<3>, dim3(g), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(g), dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t p3d_bn_bwd_finalize(const BnBwdArgs& a, hipStream_t s) {
    const int two = (a.mode == 2 || a.mode == 3) ? 1 : 0;
    if (a.nparts >= 64) hipLaunchKernelGGL(bn_bwd_finalize_kernel<64>, dim3((a.C + 3) / 4), dim3(256), 0, s, a, two);
    else hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3((a.C + 15) / 16), dim3(256), 0, s, a, two);
    return hipGetLastError();
}

hipError_t p3d_bn_grad_finalize(const BnGradFold& f, hipStream_t s) {
    if (!f.part || f.nparts <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_grad_finalize_kernel, dim3((f.C + 15) / 16), dim3(256), 0, s, f);
    return hipGetLastError();
}

hipError_t p3d_bn_bwd_apply(const BnBwdArgs& a, hipStream_t s) {
    if (a.C & 3) return hipErrorInvalidValue;
    unsigned g = grid_for(a.M * (a.C >> 2));
    switch (a.mode) {
        case 0: hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, dim3(g), dim3(256), 0, s, a); break;
        case 1: hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(g), dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(bn_bwd_apply_kernel<2>, dim3(g), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(bn_bwd_apply_kernel<3>, dim3(g), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(g), dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t p3d_maxpool_fwd(const PoolArgs& a, hipStream_t s) {
    if (a.C & 3) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.Do * a.Ho * a.Wo * (a.C >> 2);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// Non-overlapping windows (k == s, no padding: the temporal pools p3d.py:183,189,195): every input cell belongs to
// exactly one window, so dx is written (or accumulated) directly -- no atomics, no zero fill.
__global__ __launch_bounds__(256) void maxpool_bwd_disjoint_kernel(PoolArgs a, int accumulate) {
    p3d_warm_kernarg_lines<(int)(sizeof(PoolArgs) / 64)>();
    P3D_CHAIN_PRIO();
    const int c4n = a.C >> 2;
    const long long total = (long long)a.N * a.Do * a.Ho * a.Wo * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long pos = i / c4n;
        const int c = (int)(i - pos * c4n) << 2;
        const long long opos = pos;
        const int ow = (int)(pos % a.Wo); pos /= a.Wo;
        const int oh = (int)(pos % a.Ho); pos /= a.Ho;
        const int od = (int)(pos % a.Do); const int n = (int)(pos / a.Do);
        const float4 g = ld4(a.dy + opos * a.lddy + c);
        const float4 y = ld4(a.y + opos * a.ldy + c);
        float4 left = g;                       // gradient not yet handed to an earlier (first) maximum
        bool done[4] = {false, false, false, false};
        for (int kd = 0; kd < a.kd; ++kd)
            for (int kh = 0; kh < a.kh; ++kh)
                for (int kw = 0; kw < a.kw; ++kw) {
                    const long long ipos = (((long long)n * a.Di + od * a.sd + kd) * a.Hi + oh * a.sh + kh) * a.Wi + ow * a.sw + kw;
                    const float4 v = ld4(a.x + ipos * a.ldx + c);
                    float4 d = f4(0.f);
                    if (!done[0] && v.x == y.x) { d.x = left.x; done[0] = true; }
                    if (!done[1] && v.y == y.y) { d.y = left.y; done[1] = true; }
                    if (!done[2] && v.z == y.z) { d.z = left.z; done[2] = true; }
                    if (!done[3] && v.w == y.w) { d.w = left.w; done[3] = true; }
                    float* dst = a.dx + ipos * a.lddx + c;
                    if (accumulate) d = add4(d, ld4(dst));
                    st4(dst, d);
                }
    }
}

// Overlapping windows (pool1, p3d.py:177: k = [2,3,3], s = 2): thread = one input cell x 4 channels; it visits the
// windows that contain the cell (at most ceil(k/s) per axis) and takes dy where the stored arg-max tap is its own.
__global__ __launch_bounds__(256) void maxpool_bwd_gather_kernel(PoolArgs a, int accumulate) {
    p3d_warm_kernarg_lines<(int)(sizeof(PoolArgs) / 64)>();
    P3D_CHAIN_PRIO();
    const int c4n = a.C >> 2;
    const long long total = (long long)a.N * a.Di * a.Hi * a.Wi * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long pos = i / c4n;
        const int c = (int)(i - pos * c4n) << 2;
        const long long ipos = pos;
        const int iw = (int)(pos % a.Wi); pos /= a.Wi;
        const int ih = (int)(pos % a.Hi); pos /= a.Hi;
        const int id = (int)(pos % a.Di); const int n = (int)(pos / a.Di);
        const int pdv = id + a.pd, phv = ih + a.ph, pwv = iw + a.pw;
        float4 acc = f4(0.f);
        if (a.kd <= 2 * a.sd && a.kh <= 2 * a.sh && a.kw <= 2 * a.sw) {
            // at most two windows per axis (pool1: [2,3,3] by 2): the up to eight (index word, dy) pairs are loaded together and
            // then taken in the order of the general loops below -- those wait for one pair at a time (149 us for pool1 at 8 clips)
            int od2[2], oh2[2], ow2[2];
            bool vd[2], vh[2], vw[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                od2[q] = pdv / a.sd - q; vd[q] = od2[q] >= 0 && od2[q] < a.Do && pdv - od2[q] * a.sd < a.kd;
                oh2[q] = phv / a.sh - q; vh[q] = oh2[q] >= 0 && oh2[q] < a.Ho && phv - oh2[q] * a.sh < a.kh;
                ow2[q] = pwv / a.sw - q; vw[q] = ow2[q] >= 0 && ow2[q] < a.Wo && pwv - ow2[q] * a.sw < a.kw;
            }
            unsigned wd[8]; float4 g8[8]; unsigned t8[8]; bool ok[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int qd = q >> 2, qh = (q >> 1) & 1, qw = q & 1;
                ok[q] = vd[qd] && vh[qh] && vw[qw];
                t8[q] = (unsigned)(((pdv - od2[qd] * a.sd) * a.kh + (phv - oh2[qh] * a.sh)) * a.kw + (pwv - ow2[qw] * a.sw));
                const long long opos = ok[q] ? (((long long)n * a.Do + od2[qd]) * a.Ho + oh2[qh]) * a.Wo + ow2[qw] : 0;
                wd[q] = ok[q] ? a.idx[opos * c4n + (c >> 2)] : 0xffffffffu;
                g8[q] = ok[q] ? ld4(a.dy + opos * a.lddy + c) : f4(0.f);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (!ok[q]) continue;
                const unsigned w = wd[q], t = t8[q];
                if ((w & 255u) == t) acc.x += g8[q].x;
                if (((w >> 8) & 255u) == t) acc.y += g8[q].y;
                if (((w >> 16) & 255u) == t) acc.z += g8[q].z;
                if ((w >> 24) == t) acc.w += g8[q].w;
            }
        } else
        for (int od = pdv / a.sd; od >= 0 && pdv - od * a.sd < a.kd; --od) {
            if (od >= a.Do) continue;
            for (int oh = phv / a.sh; oh >= 0 && phv - oh * a.sh < a.kh; --oh) {
                if (oh >= a.Ho) continue;
                for (int ow = pwv / a.sw; ow >= 0 && pwv - ow * a.sw < a.kw; --ow) {
                    if (ow >= a.Wo) continue;
                    const unsigned t = (unsigned)(((pdv - od * a.sd) * a.kh + (phv - oh * a.sh)) * a.kw + (pwv - ow * a.sw));
                    const long long opos = (((long long)n * a.Do + od) * a.Ho + oh) * a.Wo + ow;
                    const unsigned w = a.idx[opos * c4n + (c >> 2)];
                    const float4 g = ld4(a.dy + opos * a.lddy + c);
                    if ((w & 255u) == t) acc.x += g.x;
                    if (((w >> 8) & 255u) == t) acc.y += g.y;
                    if (((w >> 16) & 255u) == t) acc.z += g.z;
                    if ((w >> 24) == t) acc.w += g.w;
                }
            }
        }
        float* dst = a.dx + ipos * a.lddx + c;
        if (accumulate) acc = add4(acc, ld4(dst));
        st4(dst, acc);
    }
}

bool p3d_maxpool_disjoint(const PoolArgs& a) {
    return a.kd == a.sd && a.kh == a.sh && a.kw == a.sw && a.pd == 0 && a.ph == 0 && a.pw == 0 &&
           a.Do * a.sd == a.Di && a.Ho * a.sh == a.Hi && a.Wo * a.sw == a.Wi;
}

hipError_t p3d_maxpool_bwd_disjoint(const PoolArgs& a, int accumulate, hipStream_t s) {
    if ((a.C & 3) || !p3d_maxpool_disjoint(a)) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.Do * a.Ho * a.Wo * (a.C >> 2);
    hipLaunchKernelGGL(maxpool_bwd_disjoint_kernel, dim3(grid_for(total)), dim3(256), 0, s, a, accumulate);
    return hipGetLastError();
}


hipError_t p3d_smooth_l1(const float* pred, const float* target, long n, double* loss_out, float* dlogits,
                         int through_sigmoid, hipStream_t s) {
    const unsigned g = grid_for(n, 256, 1024);
    float* slab = nullptr; unsigned* cnt = nullptr;
    const hipError_t e = p3d_stream_scratch(s, 2 * (size_t)g, 1, &slab, &cnt);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(g), dim3(256), 0, s, pred, target, (long long)n, loss_out, dlogits, through_sigmoid,
                       reinterpret_cast<double*>(slab), cnt);
    return hipGetLastError();
}

hipError_t p3d_adam(float* p, const float* g, float* m, float* v, long n, float lr_t, const float* lr_dev, float b1, float b2,
                    float eps, hipStream_t s) {
    const long long n4 = ((long long)n + 3) / 4;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4)), dim3(256), 0, s, p, g, m, v, n4, (long long)n, lr_t, lr_dev, b1, b2, eps);
    return hipGetLastError();
}

__global__ void set_step_scalars_kernel(unsigned long long* seed_dst, float* lr_dst, unsigned long long seed, float lr_t) {
    if (seed_dst) *seed_dst = seed;
    if (lr_dst) *lr_dst = lr_t;
}
hipError_t p3d_set_step_scalars(unsigned long long* seed_dst, float* lr_dst, unsigned long long seed, float lr_t, hipStream_t s) {
    hipLaunchKernelGGL(set_step_scalars_kernel, dim3(1), dim3(1), 0, s, seed_dst, lr_dst, seed, lr_t);
    return hipGetLastError();
}

hipError_t p3d_add_inplace(float* dst, int lddst, const float* src, int ldsrc, long M, int C, hipStream_t s) {
    if (C & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for((long long)M * (C >> 2))), dim3(256), 0, s, dst, lddst, src, ldsrc,
                       (long long)M, C, 0);
    return hipGetLastError();
}

hipError_t p3d_copy_strided(float* dst, int lddst, const float* src, int ldsrc, long M, int C, hipStream_t s) {
    if (C & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for((long long)M * (C >> 2))), dim3(256), 0, s, dst, lddst, src, ldsrc,
                       (long long)M, C, 1);
    return hipGetLastError();
}

hipError_t p3d_fill_uniform(float* p, long n, float lo, float hi, unsigned long long seed, hipStream_t s) {
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, (long long)n, lo, hi, seed);
    return hipGetLastError();
}

hipError_t p3d_fill_trunc_normal(float* p, long n, float stddev, unsigned long long seed, hipStream_t s) {
    hipLaunchKernelGGL(fill_trunc_normal_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, (long long)n, stddev, seed);
    return hipGetLastError();
}

hipError_t p3d_colsum(const float* dy, int ld, long M, int C, float* out, hipStream_t s) {
    if ((C & 3) == 0 && C >= 4 && C <= 1024 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const int R = 256 / (C >> 2);
        long long b = (M + (long long)R * 8 - 1) / ((long long)R * 8);      // >= 8 rows per thread before another block pays
        if (b > 1024) b = 1024;
        if (b < 1) b = 1;
        float* slab = nullptr; unsigned* cnt = nullptr;
        const hipError_t e = p3d_stream_scratch(s, (size_t)b * C, 1, &slab, &cnt);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)b), dim3(256), 0, s, dy, ld, (long long)M, C, out, slab, cnt);
        return hipGetLastError();
    }
    long long bx = (M + 63) / 64;
    if (bx > 128) bx = 128;
    if (bx < 1) bx = 1;
    const unsigned gy = (unsigned)((C + 255) / 256);
    float* slab = nullptr; unsigned* cnt = nullptr;
    const hipError_t e = p3d_stream_scratch(s, (size_t)bx * C, gy, &slab, &cnt);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)bx, gy), dim3(256), 0, s, dy, ld, (long long)M, C, out, slab, cnt);
    return hipGetLastError();
}

// ---- stem (firstconv1, p3d.py:172: [1,7,7,3,64] stride [1,2,2]) on the pipelined kernels -----------------------
// With 3 input channels a pixel is 12 bytes and nothing is 16-byte aligned.  The input is therefore re-laid with a
// fourth, zero, channel and the SAME padding of the W axis written out: x4[row][pad_before + w][0..3], row = (n,d,h),
// Wp = W + total W padding.  A kernel row kh then reads ONE contiguous, aligned run of kw*4 floats per output
// position (start 2*wo pixels), i.e. the stem is a 7-tap convolution with K = 28 on the ordinary implicit-GEMM and
// weight-gradient kernels; the weights are packed to [kh][kw*4 + ci][Cout] with zeros at ci = 3.
namespace {
__global__ __launch_bounds__(256) void stem_pad_kernel(const float* x, float* x4, long long rows, int W, int Wp, int pad) {
    const long long total = rows * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / W;
        const int w = (int)(i - r * W);
        const float* src = x + i * 3;
        *reinterpret_cast<float4*>(x4 + (r * Wp + pad + w) * 4) = make_float4(src[0], src[1], src[2], 0.f);
    }
}
// w [kh][kw][3][Co] -> w4 [kh][kw*4 + ci][Co]
__global__ __launch_bounds__(256) void stem_pack_w_kernel(const float* w, float* w4, int taps_hw, int Co) {
    const int total = taps_hw * 4 * Co;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Co, ci = (i / Co) & 3, t = i / (4 * Co);
        w4[i] = ci < 3 ? w[(t * 3 + ci) * Co + co] : 0.f;
    }
}
// dw [kh][kw][3][Co] += dw4 [kh][kw*4 + ci][Co]
__global__ __launch_bounds__(256) void stem_unpack_dw_kernel(const float* dw4, float* dw, int taps_hw, int Co) {
    const int total = taps_hw * 3 * Co;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % Co, ci = (i / Co) % 3, t = i / (3 * Co);
        dw[i] += dw4[((t * 4) + ci) * Co + co];
    }
}
}  // namespace

hipError_t p3d_stem_pad(const float* x, float* x4, long long rows, int W, int Wp, int pad, hipStream_t s) {
    hipLaunchKernelGGL(stem_pad_kernel, dim3(grid_for(rows * W)), dim3(256), 0, s, x, x4, rows, W, Wp, pad);
    return hipGetLastError();
}
hipError_t p3d_stem_pack_w(const float* w, float* w4, int taps_hw, int Co, hipStream_t s) {
    hipLaunchKernelGGL(stem_pack_w_kernel, dim3((taps_hw * 4 * Co + 255) / 256), dim3(256), 0, s, w, w4, taps_hw, Co);
    return hipGetLastError();
}
hipError_t p3d_stem_unpack_dw(const float* dw4, float* dw, int taps_hw, int Co, hipStream_t s) {
    hipLaunchKernelGGL(stem_unpack_dw_kernel, dim3((taps_hw * 3 * Co + 255) / 256), dim3(256), 0, s, dw4, dw, taps_hw, Co);
    return hipGetLastError();
}

hipError_t p3d_maxpool_bwd_gather(const PoolArgs& a, int accumulate, hipStream_t s) {
    if ((a.C & 3) || !a.idx || a.kd * a.kh * a.kw > 250) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.Di * a.Hi * a.Wi * (a.C >> 2);
    hipLaunchKernelGGL(maxpool_bwd_gather_kernel, dim3(grid_for(total)), dim3(256), 0, s, a, accumulate);
    return hipGetLastError();
}
