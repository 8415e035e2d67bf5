// GroupNorm (reference gn/p3d_gn.py:24-46 == utils/network.py:65-87; eps 1e-5, G = min(32, C), statistics per
// sample and group over (C/G, D, H, W)) and the fused normalise / ReLU / add passes of the GN bottleneck
// (gn/p3d_gn.py:100-179), forward and backward, for gfx950.  NDHWC: a sample is R = D*H*W consecutive rows.
//
// The reference transposes to NCDHW and back around every GroupNorm; here nothing moves: per-(sample,
// channel) sums are reduced where the data lies, a finalize pass folds the C/G channels of a group and emits
// per-(sample, channel) scale/shift tables, and the apply pass is the BatchNorm apply with a table row per
// sample.  Modes: 0 relu(gn(y1)); 1 relu(gn(y1) + r); 2 relu(gn(y1) + gn(y2)) (forward only); 3 relu(gn(y1)) + relu(gn(y2)) (ST_B);
// 4 r + relu(gn(y1)) (ST_C); 5 gn(y1); 6 relu(gn(y1) + r * cs[n,c] * ss[pos]) (CBAM-scaled residual,
// gn/p3d_gn.py:175-177 with utils/network.py:249,274 folded in).
#include "p3d_kernels.h"
#include "det_reduce.h"
#define P3D_SEED(a) ((a).seed_dev ? *(a).seed_dev : (a).seed)   // wave-uniform; device-resident under graph replay

namespace {

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

// ---- statistics: sums[n][c] += (sum, sumsq) over a slice of the sample's rows.  grid = (row slices, N).
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* y, int ld, int R, int C, double* sums, float* part,
                                                       unsigned* counters) {
    P3D_CHAIN_PRIO();
    __shared__ float red[256][8];
    __shared__ int last_flag;
    const int c4n = C >> 2, rpi = 256 / c4n;
    const int tid = threadIdx.x, sub = tid / c4n, c = (tid - sub * c4n) << 2;
    const int n = blockIdx.y;
    float4 s1 = f4(0.f), s2 = f4(0.f);
    if (sub < rpi)
        for (int r = blockIdx.x * rpi + sub; r < R; r += gridDim.x * rpi) {
            const float4 v = ld4(y + ((long long)n * R + r) * ld + c);
            s1 = add4(s1, v);
            s2 = fma4(v, v, s2);
        }
    float* q = red[tid];
    q[0] = s1.x; q[1] = s1.y; q[2] = s1.z; q[3] = s1.w; q[4] = s2.x; q[5] = s2.y; q[6] = s2.z; q[7] = s2.w;
    __syncthreads();
    if (tid < c4n) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = 0.f;
        for (int s = 0; s < rpi; ++s)
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] += red[s * c4n + tid][k];
        // this row slice's partial (plain stores); the last arriving slice of the sample folds them in slice order
        const size_t dst = (((size_t)blockIdx.x * gridDim.y + n) * C + c) * 2;      // write-through: no release fence below
        p3d_store_wt4(part, dst, make_float4(t[0], t[4], t[1], t[5]));
        p3d_store_wt4(part, dst + 4, make_float4(t[2], t[6], t[3], t[7]));
    }
    if (!p3d_last_block_wt(counters + n, gridDim.x, &last_flag)) return;
    if (tid < c4n) {
        double acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.0;
        // slice order, eight slices' loads in flight (the serial form spent ~0.3 us of L2 latency per slice: 68 us for 256)
        for (unsigned b0 = 0; b0 < gridDim.x; b0 += 8) {
            float4 v[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const unsigned b = min(b0 + u, gridDim.x - 1);
                const float* src = part + (((size_t)b * gridDim.y + n) * C + c) * 2;
                v[u][0] = ld4(src); v[u][1] = ld4(src + 4);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (b0 + u >= gridDim.x) break;
                acc[0] += (double)v[u][0].x; acc[1] += (double)v[u][0].y; acc[2] += (double)v[u][0].z; acc[3] += (double)v[u][0].w;
                acc[4] += (double)v[u][1].x; acc[5] += (double)v[u][1].y; acc[6] += (double)v[u][1].z; acc[7] += (double)v[u][1].w;
            }
        }
        double* out = sums + ((long long)n * C + c) * 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) out[k] = acc[k];
    }
}

// ---- finalize: thread per (n, c); the C/G (a power of two <= 32) lanes of a group fold with shuffles
__global__ void gn_finalize_kernel(GnParams p, int N, int R, float eps) {
    P3D_CHAIN_PRIO();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = p.C / p.G;
    const bool ok = i < N * p.C;
    double s1 = ok ? p.sums[(long long)i * 2] : 0.0, s2 = ok ? p.sums[(long long)i * 2 + 1] : 0.0;
    for (int o = 1; o < cg; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (!ok) return;
    const int c = i % p.C;
    const double cnt = (double)R * cg;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)p.gamma[c] * inv;
    p.scale[i] = (float)sc;
    p.shift[i] = (float)((double)p.beta[c] - mean * sc);
    p.mean[i] = (float)mean;
    p.invstd[i] = (float)inv;
}

// ---- apply
template <int MODE>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnApplyArgs a) {
    p3d_warm_kernargs<GnApplyArgs>();
    P3D_CHAIN_PRIO();
    const int c4n = a.C >> 2;
    const long long total = a.M * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int c = (int)(i - row * c4n) << 2;
        const long long t = (row / a.R) * a.C + c;          // table row of this sample
        const float4 v = fma4(ld4(a.g1.scale + t), ld4(a.y1 + row * a.ld1 + c), ld4(a.g1.shift + t));
        float4 z;
        if (MODE == 0) z = relu4(v);
        else if (MODE == 5) z = v;
        else if (MODE == 1) z = relu4(add4(v, ld4(a.y2 + row * a.ld2 + c)));
        else if (MODE == 2) z = relu4(add4(v, fma4(ld4(a.g2.scale + t), ld4(a.y2 + row * a.ld2 + c), ld4(a.g2.shift + t))));   // forward only
        else if (MODE == 3) z = add4(relu4(v), relu4(fma4(ld4(a.g2.scale + t), ld4(a.y2 + row * a.ld2 + c), ld4(a.g2.shift + t))));
        else if (MODE == 4) z = add4(ld4(a.y2 + row * a.ld2 + c), relu4(v));
        else z = relu4(add4(v, mul4(mul4(ld4(a.y2 + row * a.ld2 + c), ld4(a.cs + t)), f4(a.ss[row]))));
        if (a.drop_scale > 0.f) z = mul4(z, dropmask4(P3D_SEED(a), row * a.C + c, a.drop_rate, a.drop_scale));
        st4(a.z + row * a.ldz + c, z);
    }
}

// gradient entering GN1 / GN2 (after gates) for one float4
template <int MODE>
__device__ __forceinline__ void gn_gates(const GnApplyArgs& a, long long row, int c, long long t, float4& g1, float4& g2,
                                         float4& y1, float4& y2) {
    float4 dz = ld4(a.dz + row * a.ldz + c);
    if (a.drop_scale > 0.f) dz = mul4(dz, dropmask4(P3D_SEED(a), row * a.C + c, a.drop_rate, a.drop_scale));
    y1 = ld4(a.y1 + row * a.ld1 + c);
    const float4 v1 = fma4(ld4(a.g1.scale + t), y1, ld4(a.g1.shift + t));
    y2 = f4(0.f); g2 = f4(0.f);
    if (MODE == 0) g1 = gate4(dz, v1);
    else if (MODE == 5) g1 = dz;
    else if (MODE == 1) { y2 = ld4(a.y2 + row * a.ld2 + c); g1 = gate4(dz, add4(v1, y2)); g2 = g1; }
    else if (MODE == 3) {
        y2 = ld4(a.y2 + row * a.ld2 + c);
        const float4 v2 = fma4(ld4(a.g2.scale + t), y2, ld4(a.g2.shift + t));
        g1 = gate4(dz, v1); g2 = gate4(dz, v2);
    } else if (MODE == 4) { g1 = gate4(dz, v1); g2 = dz; }
    else {
        y2 = ld4(a.y2 + row * a.ld2 + c);
        const float4 res = mul4(mul4(y2, ld4(a.cs + t)), f4(a.ss[row]));
        g1 = gate4(dz, add4(v1, res)); g2 = g1;            // g2 = gradient w.r.t. the CBAM output
    }
}

// per-(n,c) sums of g and g*xhat over a slice of the sample's rows.  grid = (row slices, N)
template <int MODE>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(GnApplyArgs a) {
    p3d_warm_kernargs<GnApplyArgs>();
    P3D_CHAIN_PRIO();
    constexpr bool TWO = (MODE == 3);
    __shared__ float red[256][TWO ? 16 : 8];
    __shared__ int last_flag;
    const int c4n = a.C >> 2, rpi = 256 / c4n;
    const int tid = threadIdx.x, sub = tid / c4n, c = (tid - sub * c4n) << 2;
    const int n = blockIdx.y;
    const long long t = (long long)n * a.C + c;
    float4 s1 = f4(0.f), sx1 = f4(0.f), s2 = f4(0.f), sx2 = f4(0.f);
    if (sub < rpi) {
        const float4 m1 = ld4(a.g1.mean + t), i1 = ld4(a.g1.invstd + t);
        float4 m2 = f4(0.f), i2 = f4(0.f);
        if (TWO) { m2 = ld4(a.g2.mean + t); i2 = ld4(a.g2.invstd + t); }
        for (int r = blockIdx.x * rpi + sub; r < a.R; r += gridDim.x * rpi) {
            const long long row = (long long)n * a.R + r;
            float4 g1, g2, y1, y2;
            gn_gates<MODE>(a, row, c, t, g1, g2, y1, y2);
            s1 = add4(s1, g1);
            sx1 = fma4(g1, mul4(sub4(y1, m1), i1), sx1);
            if (TWO) { s2 = add4(s2, g2); sx2 = fma4(g2, mul4(sub4(y2, m2), i2), sx2); }
        }
    }
    float* q = red[tid];
    q[0] = s1.x; q[1] = s1.y; q[2] = s1.z; q[3] = s1.w; q[4] = sx1.x; q[5] = sx1.y; q[6] = sx1.z; q[7] = sx1.w;
    if (TWO) { q[8] = s2.x; q[9] = s2.y; q[10] = s2.z; q[11] = s2.w; q[12] = sx2.x; q[13] = sx2.y; q[14] = sx2.z; q[15] = sx2.w; }
    __syncthreads();
    if (tid < c4n) {
        constexpr int NV = TWO ? 16 : 8;
        float tt[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) tt[k] = 0.f;
        for (int s = 0; s < rpi; ++s)
#pragma unroll
            for (int k = 0; k < NV; ++k) tt[k] += red[s * c4n + tid][k];
        // this row slice's partials (plain stores, [slice][n][C][NV/4 pairs]); folded in slice order by the last arriver
        const size_t dst = (((size_t)blockIdx.x * gridDim.y + n) * a.C + c) * (NV / 4);      // write-through: no release fence below
        if (TWO) {
#pragma unroll
            for (int k = 0; k < 4; ++k) p3d_store_wt4(a.part, dst + 4 * k, make_float4(tt[k], tt[4 + k], tt[8 + k], tt[12 + k]));
        } else {
            p3d_store_wt4(a.part, dst, make_float4(tt[0], tt[4], tt[1], tt[5]));
            p3d_store_wt4(a.part, dst + 4, make_float4(tt[2], tt[6], tt[3], tt[7]));
        }
    }
    if (!p3d_last_block_wt(a.counters + n, gridDim.x, &last_flag)) return;
    if (tid < c4n) {
        constexpr int NV = TWO ? 16 : 8;
        double acc[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] = 0.0;
        // slice order, four slices' loads in flight
        for (unsigned b0 = 0; b0 < gridDim.x; b0 += 4) {
            float4 v[4][NV / 4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned b = min(b0 + u, gridDim.x - 1);
                const float* src = a.part + (((size_t)b * gridDim.y + n) * a.C + c) * (NV / 4);
#pragma unroll
                for (int q = 0; q < NV / 4; ++q) v[u][q] = ld4(src + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (b0 + u >= gridDim.x) break;
#pragma unroll
                for (int q = 0; q < NV / 4; ++q) {
                    acc[4 * q] += (double)v[u][q].x; acc[4 * q + 1] += (double)v[u][q].y;
                    acc[4 * q + 2] += (double)v[u][q].z; acc[4 * q + 3] += (double)v[u][q].w;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a.g1.sums[(t + k) * 2] = acc[k * (NV / 4) + 0];
            a.g1.sums[(t + k) * 2 + 1] = acc[k * (NV / 4) + 1];
            if (TWO) { a.g2.sums[(t + k) * 2] = acc[k * (NV / 4) + 2]; a.g2.sums[(t + k) * 2 + 1] = acc[k * (NV / 4) + 3]; }
        }
    }
}

// thread per (n, c): fold gamma-weighted sums over the group's lanes -> coefficients  dy = k*g - c1 - xhat*c2
__global__ void gn_bwd_finalize_kernel(GnParams p, int N, int R) {
    P3D_CHAIN_PRIO();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = p.C / p.G;
    const bool ok = i < N * p.C;
    const int c = ok ? i % p.C : 0;
    const double gm = ok ? (double)p.gamma[c] : 0.0;
    double A = ok ? gm * p.sums[(long long)i * 2] : 0.0, B = ok ? gm * p.sums[(long long)i * 2 + 1] : 0.0;
    for (int o = 1; o < cg; o <<= 1) { A += __shfl_xor(A, o); B += __shfl_xor(B, o); }
    if (!ok) return;
    const double cnt = (double)R * cg;
    const double inv = p.invstd[i];
    p.coef[(long long)i * 3 + 0] = (float)(gm * inv);
    p.coef[(long long)i * 3 + 1] = (float)(inv * A / cnt);
    p.coef[(long long)i * 3 + 2] = (float)(inv * B / cnt);
}
// thread per channel: dgamma = sum_n sum(g*xhat), dbeta = sum_n sum(g)
__global__ void gn_bwd_params_kernel(GnParams p, int N, float* dgamma, float* dbeta) {
    P3D_CHAIN_PRIO();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= p.C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < N; ++n) {
        s1 += p.sums[((long long)n * p.C + c) * 2];
        s2 += p.sums[((long long)n * p.C + c) * 2 + 1];
    }
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
}

__device__ __forceinline__ float4 gn_dx(const GnParams& g, long long t, float4 gr, float4 y) {
    const float4 xh = mul4(sub4(y, ld4(g.mean + t)), ld4(g.invstd + t));
    const float* cf = g.coef + t * 3;       // 4 channels x (k, c1, c2)
    const float4 a = ld4(cf), b = ld4(cf + 4), c = ld4(cf + 8);
    const float4 k = make_float4(a.x, a.w, b.z, c.y), c1 = make_float4(a.y, b.x, b.w, c.z), c2 = make_float4(a.z, b.y, c.x, c.w);
    return sub4(sub4(mul4(k, gr), c1), mul4(xh, c2));
}

template <int MODE>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnApplyArgs a) {
    p3d_warm_kernargs<GnApplyArgs>();
    P3D_CHAIN_PRIO();
    const int c4n = a.C >> 2;
    const long long total = a.M * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int c = (int)(i - row * c4n) << 2;
        const long long t = (row / a.R) * a.C + c;
        float4 g1, g2, y1, y2;
        gn_gates<MODE>(a, row, c, t, g1, g2, y1, y2);
        st4(a.dy1 + row * a.lddy1 + c, gn_dx(a.g1, t, g1, y1));
        if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 6) {
            float4 d = (MODE == 3) ? gn_dx(a.g2, t, g2, y2) : g2;
            float* dst = a.dy2 + row * a.lddy2 + c;
            if (a.acc2) d = add4(d, ld4(dst));
            st4(dst, d);
        }
    }
}

// ---- small tensors: one launch forward, one launch backward --------------------------------------------------
// A (sample, group) slab of R rows x C/G channels that fits the registers of one block (R * C/G / 4 <= 2048 float4:
// every GroupNorm of stage 3 at 2x7x7, the narrow ones of stage 2) needs no other block for its statistics, so the
// stats / finalize / apply launches (and reduce / finalize / params / apply backward) collapse into one each, like
// bn_small.hip does for BatchNorm.  grid = (G, N); thread = (channel quad of the group, row lane).  Tables
// (scale, shift, mean, invstd per (n, c)) are still written: CBAM's block-end pass and mixed paths read them.
constexpr int GS_MJ = 8;

__device__ __forceinline__ float block_sum1(float v, float* xch4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) xch4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (xch4[0] + xch4[1]) + (xch4[2] + xch4[3]);
}
__device__ __forceinline__ float hsum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

template <int MODE>
__global__ __launch_bounds__(256) void gn_small_fwd_kernel(GnApplyArgs a) {
    p3d_warm_kernargs<GnApplyArgs>();
    P3D_CHAIN_PRIO();
    constexpr bool TWO = (MODE == 3);
    __shared__ float xch[4];
    const int cpg = a.C / a.g1.G, c4n = cpg >> 2, RL = 256 / c4n;
    const int n = blockIdx.y, c = blockIdx.x * cpg + (threadIdx.x % c4n) * 4, rl = threadIdx.x / c4n;
    const long long t = (long long)n * a.C + c;
    const float cnt = (float)a.R * (float)cpg;
    float4 v1[GS_MJ], v2[GS_MJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < GS_MJ; ++j) {
        const int r = rl + RL * j;
        v1[j] = f4(0.f); v2[j] = f4(0.f);
        if (r < a.R) {
            const long long row = (long long)n * a.R + r;
            v1[j] = ld4(a.y1 + row * a.ld1 + c);
            s1 += hsum4(v1[j]);
            if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 6) v2[j] = ld4(a.y2 + row * a.ld2 + c);
            if (TWO) s2 += hsum4(v2[j]);
        }
    }
    // two-pass moments from the registers
    const float mean1 = block_sum1(s1, xch) / cnt;
    float q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int j = 0; j < GS_MJ; ++j)
        if (rl + RL * j < a.R) { const float4 d = sub4(v1[j], f4(mean1)); q1 += hsum4(mul4(d, d)); }
    const float inv1 = 1.f / sqrtf(block_sum1(q1, xch) / cnt + a.eps);
    const float4 sc1 = mul4(ld4(a.g1.gamma + c), f4(inv1));
    const float4 sh1 = sub4(ld4(a.g1.beta + c), mul4(f4(mean1), sc1));
    if (rl == 0) { st4(a.g1.scale + t, sc1); st4(a.g1.shift + t, sh1); st4(a.g1.mean + t, f4(mean1)); st4(a.g1.invstd + t, f4(inv1)); }
    float4 sc2 = f4(0.f), sh2 = f4(0.f);
    if (TWO) {
        const float mean2 = block_sum1(s2, xch) / cnt;
#pragma unroll
        for (int j = 0; j < GS_MJ; ++j)
            if (rl + RL * j < a.R) { const float4 d = sub4(v2[j], f4(mean2)); q2 += hsum4(mul4(d, d)); }
        const float inv2 = 1.f / sqrtf(block_sum1(q2, xch) / cnt + a.eps);
        sc2 = mul4(ld4(a.g2.gamma + c), f4(inv2));
        sh2 = sub4(ld4(a.g2.beta + c), mul4(f4(mean2), sc2));
        if (rl == 0) { st4(a.g2.scale + t, sc2); st4(a.g2.shift + t, sh2); st4(a.g2.mean + t, f4(mean2)); st4(a.g2.invstd + t, f4(inv2)); }
    }
    float4 cs = f4(0.f);
    if (MODE == 6) cs = ld4(a.cs + t);
#pragma unroll
    for (int j = 0; j < GS_MJ; ++j) {
        const int r = rl + RL * j;
        if (r < a.R) {
            const long long row = (long long)n * a.R + r;
            const float4 v = fma4(sc1, v1[j], sh1);
            float4 z;
            if (MODE == 0) z = relu4(v);
            else if (MODE == 5) z = v;
            else if (MODE == 1) z = relu4(add4(v, v2[j]));
            else if (MODE == 3) z = add4(relu4(v), relu4(fma4(sc2, v2[j], sh2)));
            else if (MODE == 4) z = add4(v2[j], relu4(v));
            else z = relu4(add4(v, mul4(mul4(v2[j], cs), f4(a.ss[row]))));
            if (a.drop_scale > 0.f) z = mul4(z, dropmask4(P3D_SEED(a), row * a.C + c, a.drop_rate, a.drop_scale));
            st4(a.z + row * a.ldz + c, z);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void gn_small_bwd_kernel(GnApplyArgs a) {
    p3d_warm_kernargs<GnApplyArgs>();
    P3D_CHAIN_PRIO();
    constexpr bool TWO = (MODE == 3);
    __shared__ float4 red[256][TWO ? 4 : 2];
    __shared__ float coef[4];                       // c1, c2 of GN1 (and GN2)
    const int cpg = a.C / a.g1.G, c4n = cpg >> 2, RL = 256 / c4n;
    const int n = blockIdx.y, cq = threadIdx.x % c4n, c = blockIdx.x * cpg + cq * 4, rl = threadIdx.x / c4n;
    const long long t = (long long)n * a.C + c;
    const float cnt = (float)a.R * (float)cpg;
    const float4 m1 = ld4(a.g1.mean + t), i1 = ld4(a.g1.invstd + t);
    float4 m2 = f4(0.f), i2 = f4(0.f);
    if (TWO) { m2 = ld4(a.g2.mean + t); i2 = ld4(a.g2.invstd + t); }
    float4 g1[GS_MJ], xh1[GS_MJ], g2[GS_MJ], xh2[TWO ? GS_MJ : 1];
    float4 s1 = f4(0.f), sx1 = f4(0.f), s2 = f4(0.f), sx2 = f4(0.f);
#pragma unroll
    for (int j = 0; j < GS_MJ; ++j) {
        const int r = rl + RL * j;
        g1[j] = f4(0.f); xh1[j] = f4(0.f); g2[j] = f4(0.f);
        if (TWO) xh2[TWO ? j : 0] = f4(0.f);
        if (r < a.R) {
            const long long row = (long long)n * a.R + r;
            float4 y1, y2;
            gn_gates<MODE>(a, row, c, t, g1[j], g2[j], y1, y2);
            xh1[j] = mul4(sub4(y1, m1), i1);
            s1 = add4(s1, g1[j]); sx1 = fma4(g1[j], xh1[j], sx1);
            if (TWO) { xh2[TWO ? j : 0] = mul4(sub4(y2, m2), i2); s2 = add4(s2, g2[j]); sx2 = fma4(g2[j], xh2[TWO ? j : 0], sx2); }
        }
    }
    red[threadIdx.x][0] = s1; red[threadIdx.x][1] = sx1;
    if (TWO) { red[threadIdx.x][2] = s2; red[threadIdx.x][3] = sx2; }
    __syncthreads();
    if (threadIdx.x < c4n) {                        // per-channel totals of this sample; parameter gradients add over samples
        float4 tt[TWO ? 4 : 2];
#pragma unroll
        for (int k = 0; k < (TWO ? 4 : 2); ++k) tt[k] = f4(0.f);
        for (int q = 0; q < RL; ++q)
#pragma unroll
            for (int k = 0; k < (TWO ? 4 : 2); ++k) tt[k] = add4(tt[k], red[q * c4n + threadIdx.x][k]);
#pragma unroll
        for (int k = 0; k < (TWO ? 4 : 2); ++k) red[threadIdx.x][k] = tt[k];      // (row lane 0 slot: safe, only this thread reads it)
        // per-sample partial parameter gradients [n][C][4 float4 slots]; summed over samples, in sample order, by the
        // block of this group that finishes last (end of the kernel)
        // (write-through stores: the hand-over at the end of the kernel then needs no release fence)
#pragma unroll
        for (int k = 0; k < (TWO ? 4 : 2); ++k) p3d_store_wt4(a.part, ((size_t)n * a.C + c) * 4 + 4 * k, tt[k]);      // slot k: row stride is 4 channels
    }
    __syncthreads();
    if (threadIdx.x == 0) {                         // gamma-weighted group sums -> the two mean terms
        float A1 = 0.f, B1 = 0.f, A2 = 0.f, B2 = 0.f;
        for (int q = 0; q < c4n; ++q) {
            const float4 gm = ld4(a.g1.gamma + blockIdx.x * cpg + q * 4);
            A1 += hsum4(mul4(gm, red[q][0])); B1 += hsum4(mul4(gm, red[q][1]));
            if (TWO) {
                const float4 gm2 = ld4(a.g2.gamma + blockIdx.x * cpg + q * 4);
                A2 += hsum4(mul4(gm2, red[q][2])); B2 += hsum4(mul4(gm2, red[q][3]));
            }
        }
        coef[0] = A1 / cnt; coef[1] = B1 / cnt; coef[2] = A2 / cnt; coef[3] = B2 / cnt;
    }
    __syncthreads();
    const float4 k1 = mul4(ld4(a.g1.gamma + c), i1), c11 = mul4(i1, f4(coef[0])), c12 = mul4(i1, f4(coef[1]));
    float4 k2 = f4(0.f), c21 = f4(0.f), c22 = f4(0.f);
    if (TWO) { k2 = mul4(ld4(a.g2.gamma + c), i2); c21 = mul4(i2, f4(coef[2])); c22 = mul4(i2, f4(coef[3])); }
#pragma unroll
    for (int j = 0; j < GS_MJ; ++j) {
        const int r = rl + RL * j;
        if (r < a.R) {
            const long long row = (long long)n * a.R + r;
            st4(a.dy1 + row * a.lddy1 + c, sub4(sub4(mul4(k1, g1[j]), c11), mul4(xh1[j], c12)));
            if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 6) {
                float4 d = TWO ? sub4(sub4(mul4(k2, g2[j]), c21), mul4(xh2[TWO ? j : 0], c22)) : g2[j];
                float* dst = a.dy2 + row * a.lddy2 + c;
                if (a.acc2) d = add4(d, ld4(dst));
                st4(dst, d);
            }
        }
    }
    // parameter gradients: the block of this group that finishes last adds the per-sample partials in sample order
    __shared__ int last_flag;
    if (!p3d_last_block_wt(a.counters + blockIdx.x, gridDim.y, &last_flag)) return;
    if (threadIdx.x < c4n) {
        float4 acc[TWO ? 4 : 2];
#pragma unroll
        for (int k = 0; k < (TWO ? 4 : 2); ++k) acc[k] = f4(0.f);
        for (unsigned n2 = 0; n2 < gridDim.y; ++n2) {
            const float4* pp = reinterpret_cast<const float4*>(a.part) + ((size_t)n2 * a.C + c);
#pragma unroll
            for (int k = 0; k < (TWO ? 4 : 2); ++k) acc[k] = add4(acc[k], pp[k]);
        }
        st4(a.dbeta1 + c, add4(ld4(a.dbeta1 + c), acc[0]));
        st4(a.dgamma1 + c, add4(ld4(a.dgamma1 + c), acc[1]));
        if (TWO) {
            st4(a.dbeta2 + c, add4(ld4(a.dbeta2 + c), acc[2]));
            st4(a.dgamma2 + c, add4(ld4(a.dgamma2 + c), acc[3]));
        }
    }
}

inline unsigned grid_for(long long total, int cap = 4096) {
    long long b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}
inline dim3 slice_grid(int R, int C, int N) {
    const int rpi = 256 / (C >> 2);
    long long bx = (R + (long long)rpi * 32 - 1) / ((long long)rpi * 32);
    if (bx < 1) bx = 1;
    if (bx > 256) bx = 256;
    return dim3((unsigned)bx, (unsigned)N);
}

}  // namespace

hipError_t p3d_gn_stats(const float* y, int ld, int N, int R, int C, double* sums, hipStream_t s) {
    if ((C & 3) || C > 1024 || (ld & 3)) return hipErrorInvalidValue;
    const dim3 grid = slice_grid(R, C, N);
    float* part = nullptr; unsigned* cnt = nullptr;
    const hipError_t e = p3d_stream_scratch(s, (size_t)grid.x * N * C * 2, (size_t)N, &part, &cnt);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(256), 0, s, y, ld, R, C, sums, part, cnt);
    return hipGetLastError();
}
hipError_t p3d_gn_finalize(const GnParams& p, int N, int R, float eps, hipStream_t s) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((N * p.C + 255) / 256), dim3(256), 0, s, p, N, R, eps);
    return hipGetLastError();
}

#define P3D_GN_SWITCH(KERNEL, GRID)                                                         \
    switch (a.mode) {                                                                       \
        case 0: hipLaunchKernelGGL(KERNEL<0>, GRID, dim3(256), 0, s, a); break;             \
        case 1: hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(256), 0, s, a); break;             \
        case 3: hipLaunchKernelGGL(KERNEL<3>, GRID, dim3(256), 0, s, a); break;             \
        case 4: hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), 0, s, a); break;             \
        case 5: hipLaunchKernelGGL(KERNEL<5>, GRID, dim3(256), 0, s, a); break;             \
        case 6: hipLaunchKernelGGL(KERNEL<6>, GRID, dim3(256), 0, s, a); break;             \
        default: return hipErrorInvalidValue;                                               \
    }

hipError_t p3d_gn_apply(const GnApplyArgs& a, hipStream_t s) {
    if ((a.C & 3) || a.C > 1024) return hipErrorInvalidValue;
    const dim3 g(grid_for(a.M * (a.C >> 2)));
    // mode 2 = relu(norm1(y1) + norm2(y2)): the projected residual of the BatchNorm bottleneck normalised per
    // sample (p3d_predict_windows); it has no backward
    if (a.mode == 2) { hipLaunchKernelGGL(gn_apply_kernel<2>, g, dim3(256), 0, s, a); return hipGetLastError(); }
    P3D_GN_SWITCH(gn_apply_kernel, g)
    return hipGetLastError();
}
hipError_t p3d_gn_bwd_reduce(const GnApplyArgs& a0, hipStream_t s) {
    if ((a0.C & 3) || a0.C > 1024) return hipErrorInvalidValue;
    const int N = (int)(a0.M / a0.R);
    const dim3 g = slice_grid(a0.R, a0.C, N);
    GnApplyArgs a = a0;
    const hipError_t e = p3d_stream_scratch(s, (size_t)g.x * N * a.C * 4, (size_t)N, &a.part, &a.counters);
    if (e != hipSuccess) return e;
    P3D_GN_SWITCH(gn_bwd_reduce_kernel, g)
    return hipGetLastError();
}
hipError_t p3d_gn_bwd_finalize(const GnParams& p, int N, int R, float* dgamma, float* dbeta, hipStream_t s) {
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((N * p.C + 255) / 256), dim3(256), 0, s, p, N, R);
    hipLaunchKernelGGL(gn_bwd_params_kernel, dim3((p.C + 255) / 256), dim3(256), 0, s, p, N, dgamma, dbeta);
    return hipGetLastError();
}
hipError_t p3d_gn_bwd_apply(const GnApplyArgs& a, hipStream_t s) {
    if ((a.C & 3) || a.C > 1024) return hipErrorInvalidValue;
    const dim3 g(grid_for(a.M * (a.C >> 2)));
    P3D_GN_SWITCH(gn_bwd_apply_kernel, g)
    return hipGetLastError();
}

bool p3d_gn_small_ok(int R, int C, int G) {
    if (G < 1 || C % G) return false;
    const int cpg = C / G;
    if ((cpg & 3) || (256 % (cpg >> 2))) return false;
    return (long long)R * (cpg >> 2) <= 256 * GS_MJ;
}
hipError_t p3d_gn_small_fwd(const GnApplyArgs& a, hipStream_t s) {
    if (!p3d_gn_small_ok(a.R, a.C, a.g1.G) || (a.M % a.R)) return hipErrorInvalidValue;
    const dim3 g((unsigned)a.g1.G, (unsigned)(a.M / a.R));
    P3D_GN_SWITCH(gn_small_fwd_kernel, g)
    return hipGetLastError();
}
hipError_t p3d_gn_small_bwd(const GnApplyArgs& a0, hipStream_t s) {
    if (!p3d_gn_small_ok(a0.R, a0.C, a0.g1.G) || (a0.M % a0.R)) return hipErrorInvalidValue;
    const dim3 g((unsigned)a0.g1.G, (unsigned)(a0.M / a0.R));
    GnApplyArgs a = a0;
    const hipError_t e = p3d_stream_scratch(s, (size_t)g.y * a.C * 4, (size_t)g.x, &a.part, &a.counters);
    if (e != hipSuccess) return e;
    P3D_GN_SWITCH(gn_small_bwd_kernel, g)
    return hipGetLastError();
}
