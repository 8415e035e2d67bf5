// Output head of p3d_unet on gfx950: tf.layers.conv3d_transpose(x, 1, 3, [2,2,2], 'same') followed by
// tf.sigmoid (reference p3d.py:217-219), forward and both gradients.  Cout = 1 makes this an
// HBM-bound stencil, not a GEMM, so it stays on the VALU: out[2i+k] += <x[i,:], K[k,0,:]> per axis
// (k=3 s=2: pad_before 0, the element at 2*I is dropped).
//
// forward : one thread per INPUT lattice point g produces the 2x2x2 output cube at 2g+p; parity
//           p=0 on an axis takes taps {0 from i=g, 2 from i=g-1}, p=1 takes tap 1 from i=g,
//           so each thread reads the 8 neighbours g-{0,1}^3 once and does the 27 tap dots.
// dgrad   : dx[g,c] = sum_k dlogits[2g+k] * K[k,0,c]   (27 taps, bounds-checked)
// wgrad   : dK[k,0,c] = sum_g dlogits[2g+k] * x[g,c],  dbias = sum dlogits
#include "p3d_kernels.h"
#include "det_reduce.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__global__ __launch_bounds__(256) void head_fwd_kernel(HeadArgs a) {
    P3D_CHAIN_PRIO();
    extern __shared__ float kw[];     // [27][C]
    const int C = a.C;
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) kw[i] = a.k[i];
    __syncthreads();
    const long long total = (long long)a.N * a.D * a.H * a.W;
    const float bias = a.bias[0];
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        long long t = g;
        const int w = (int)(t % a.W); t /= a.W;
        const int h = (int)(t % a.H); t /= a.H;
        const int d = (int)(t % a.D); const int n = (int)(t / a.D);
        float out[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) out[q] = bias;
        for (int dd = 0; dd < 2; ++dd) {
            if (d - dd < 0) continue;
            for (int dh = 0; dh < 2; ++dh) {
                if (h - dh < 0) continue;
                for (int dw = 0; dw < 2; ++dw) {
                    if (w - dw < 0) continue;
                    const float* xr = a.x + ((((long long)n * a.D + (d - dd)) * a.H + (h - dh)) * a.W + (w - dw)) * C;
                    // taps reachable from this neighbour: per axis delta=0 -> (k=0,p=0),(k=1,p=1); delta=1 -> (k=2,p=0)
                    const int nkd = dd ? 1 : 2, nkh = dh ? 1 : 2, nkw = dw ? 1 : 2;
                    for (int c = 0; c < C; c += 4) {
                        const float4 xv = ld4(xr + c);
                        for (int a0 = 0; a0 < nkd; ++a0) {
                            const int kd = dd ? 2 : a0, pd = dd ? 0 : a0;
                            for (int a1 = 0; a1 < nkh; ++a1) {
                                const int kh = dh ? 2 : a1, ph = dh ? 0 : a1;
                                for (int a2 = 0; a2 < nkw; ++a2) {
                                    const int kk = dw ? 2 : a2, pw = dw ? 0 : a2;
                                    const float* kp = kw + ((kd * 3 + kh) * 3 + kk) * C + c;
                                    out[(pd * 2 + ph) * 2 + pw] += xv.x * kp[0] + xv.y * kp[1] + xv.z * kp[2] + xv.w * kp[3];
                                }
                            }
                        }
                    }
                }
            }
        }
        const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
#pragma unroll
        for (int pd = 0; pd < 2; ++pd)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const long long o = (((long long)n * Do + 2 * d + pd) * Ho + 2 * h + ph) * Wo + 2 * w;
                const float v0 = out[(pd * 2 + ph) * 2], v1 = out[(pd * 2 + ph) * 2 + 1];
                *reinterpret_cast<float2*>(a.logits + o) = make_float2(v0, v1);
                *reinterpret_cast<float2*>(a.pred + o) =
                    a.sigmoid ? make_float2(1.f / (1.f + expf(-v0)), 1.f / (1.f + expf(-v1))) : make_float2(v0, v1);
            }
    }
}

// The same forward with L = C/4 lanes per input position (L a power of two): every lane takes four channels of each of
// the eight neighbours -- a position's row is one coalesced 16*L-byte read instead of 64 lanes striding C floats apart
// (which thrashed the L1: 208 GB/s at 32x224x224) -- and the eight partial outputs are summed across the L lanes by
// xor-shuffles in a fixed order.
template <int L>
__global__ __launch_bounds__(256) void head_fwd_lanes_kernel(HeadArgs a) {
    P3D_CHAIN_PRIO();
    extern __shared__ float kw[];     // [27][C]
    const int C = a.C;
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) kw[i] = a.k[i];
    __syncthreads();
    const long long total = (long long)a.N * a.D * a.H * a.W;
    const float bias = a.bias[0];
    const int part = threadIdx.x % L, c = part * 4;
    const long long stride = (long long)gridDim.x * (256 / L);
    for (long long g = (long long)blockIdx.x * (256 / L) + threadIdx.x / L; g < total; g += stride) {
        long long t = g;
        const int w = (int)(t % a.W); t /= a.W;
        const int h = (int)(t % a.H); t /= a.H;
        const int d = (int)(t % a.D); const int n = (int)(t / a.D);
        float out[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) out[q] = 0.f;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int dw = 0; dw < 2; ++dw) {
                    if (d - dd < 0 || h - dh < 0 || w - dw < 0) continue;
                    const float4 xv = ld4(a.x + ((((long long)n * a.D + (d - dd)) * a.H + (h - dh)) * a.W + (w - dw)) * C + c);
                    // taps reachable from this neighbour: per axis delta=0 -> (k=0,p=0),(k=1,p=1); delta=1 -> (k=2,p=0)
#pragma unroll
                    for (int a0 = 0; a0 < (dd ? 1 : 2); ++a0)
#pragma unroll
                        for (int a1 = 0; a1 < (dh ? 1 : 2); ++a1)
#pragma unroll
                            for (int a2 = 0; a2 < (dw ? 1 : 2); ++a2) {
                                const int kd = dd ? 2 : a0, pd = dd ? 0 : a0, kh = dh ? 2 : a1, ph = dh ? 0 : a1, kk = dw ? 2 : a2, pw = dw ? 0 : a2;
                                const float4 kv = *reinterpret_cast<const float4*>(kw + ((kd * 3 + kh) * 3 + kk) * C + c);
                                out[(pd * 2 + ph) * 2 + pw] += xv.x * kv.x + xv.y * kv.y + xv.z * kv.z + xv.w * kv.w;
                            }
                }
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1)
#pragma unroll
            for (int q = 0; q < 8; ++q) out[q] += __shfl_xor(out[q], o);
        if (part == 0) {
            const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
#pragma unroll
            for (int pd = 0; pd < 2; ++pd)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const long long o = (((long long)n * Do + 2 * d + pd) * Ho + 2 * h + ph) * Wo + 2 * w;
                    const float v0 = out[(pd * 2 + ph) * 2] + bias, v1 = out[(pd * 2 + ph) * 2 + 1] + bias;
                    *reinterpret_cast<float2*>(a.logits + o) = make_float2(v0, v1);
                    *reinterpret_cast<float2*>(a.pred + o) =
                        a.sigmoid ? make_float2(1.f / (1.f + expf(-v0)), 1.f / (1.f + expf(-v1))) : make_float2(v0, v1);
                }
        }
    }
}

__global__ __launch_bounds__(256) void head_bwd_input_kernel(HeadArgs a) {
    P3D_CHAIN_PRIO();
    extern __shared__ float kw[];
    const int C = a.C, c4n = C >> 2;
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) kw[i] = a.k[i];
    __syncthreads();
    const long long total = (long long)a.N * a.D * a.H * a.W * c4n;
    const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i / c4n;
        const int c = (int)(i - t * c4n) << 2;
        const long long g = t;
        const int w = (int)(t % a.W); t /= a.W;
        const int h = (int)(t % a.H); t /= a.H;
        const int d = (int)(t % a.D); const int n = (int)(t / a.D);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int kd = 0; kd < 3; ++kd) {
            const int od = 2 * d + kd;
            if (od >= Do) continue;
            for (int kh = 0; kh < 3; ++kh) {
                const int oh = 2 * h + kh;
                if (oh >= Ho) continue;
                for (int kk = 0; kk < 3; ++kk) {
                    const int ow = 2 * w + kk;
                    if (ow >= Wo) continue;
                    const float gdl = a.dlogits[(((long long)n * Do + od) * Ho + oh) * Wo + ow];
                    const float* kp = kw + ((kd * 3 + kh) * 3 + kk) * C + c;
                    acc.x += gdl * kp[0]; acc.y += gdl * kp[1]; acc.z += gdl * kp[2]; acc.w += gdl * kp[3];
                }
            }
        }
        *reinterpret_cast<float4*>(a.dx + g * C + c) = acc;
    }
}

// block: 256 threads = (256/C) position lanes x C channels; each thread keeps 27 tap accumulators.
__global__ __launch_bounds__(256) void head_bwd_filter_kernel(HeadArgs a) {
    __shared__ float red[256];
    const int C = a.C;
    const int lanes = 256 / C;
    const int c = threadIdx.x % C, sub = threadIdx.x / C;
    const long long total = (long long)a.N * a.D * a.H * a.W;
    const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
    float acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = 0.f;
    float bsum = 0.f;
    if (sub < lanes) {
        for (long long g = (long long)blockIdx.x * lanes + sub; g < total; g += (long long)gridDim.x * lanes) {
            long long t = g;
            const int w = (int)(t % a.W); t /= a.W;
            const int h = (int)(t % a.H); t /= a.H;
            const int d = (int)(t % a.D); const int n = (int)(t / a.D);
            const float xv = a.x[g * C + c];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const int od = 2 * d + kd, oh = 2 * h + kh, ow = 2 * w + kk;
                        float gdl = 0.f;
                        if (od < Do && oh < Ho && ow < Wo) gdl = a.dlogits[(((long long)n * Do + od) * Ho + oh) * Wo + ow];
                        acc[(kd * 3 + kh) * 3 + kk] += gdl * xv;
                        if (c == 0 && kd < 2 && kh < 2 && kk < 2) bsum += gdl;   // each output counted once
                    }
        }
    }
    for (int q = 0; q < 27; ++q) {
        red[threadIdx.x] = (sub < lanes) ? acc[q] : 0.f;
        __syncthreads();
        if (threadIdx.x < C) {
            float s = 0.f;
            for (int l = 0; l < lanes; ++l) s += red[l * C + threadIdx.x];
            p3d_store_wt(a.part, ((size_t)blockIdx.x * 28 + q) * C + threadIdx.x, s);
        }
        __syncthreads();
    }
    red[threadIdx.x] = (sub < lanes && c == 0) ? bsum : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[l * C];
        p3d_store_wt(a.part, ((size_t)blockIdx.x * 28 + 27) * C, s);
    }
    // the last arriving block adds every block's partial filter gradient in block order (no atomics)
    __shared__ int last_flag;
    if (!p3d_last_block_wt(a.counter, gridDim.x, &last_flag)) return;
    for (int i = threadIdx.x; i < 27 * C + 1; i += blockDim.x) {
        const size_t slot = i < 27 * C ? (size_t)i : (size_t)27 * C;
        float t = 0.f;
#pragma unroll 8
        for (unsigned b = 0; b < gridDim.x; ++b) t += a.part[(size_t)b * 28 * C + slot];
        if (i < 27 * C) a.dk[i] += t; else a.dbias[0] += t;
    }
}

// The same gradient for C % 4 == 0: a thread owns four channels of one position slot (one 16-byte load of x per 27
// gathered logit gradients and 108 FMAs), and the per-block partials are folded in two levels -- the last arriver of
// every FOLD consecutive blocks folds that group in block order, the last of those folds the groups in group order --
// so the serial tail reads FOLD + blocks/FOLD partials instead of `blocks`.
constexpr int HEAD_FOLD = 32;
__global__ __launch_bounds__(256) void head_bwd_filter4_kernel(HeadArgs a) {
    __shared__ float4 red[256];
    __shared__ int last_flag;
    const int C = a.C, L = C >> 2, R = 256 / L;
    const int lane = threadIdx.x % L, rs = threadIdx.x / L;
    const bool live = rs < R;
    const long long total = (long long)a.N * a.D * a.H * a.W;
    const int Do = 2 * a.D, Ho = 2 * a.H, Wo = 2 * a.W;
    float4 acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    float bsum = 0.f;
    if (live) {
        for (long long g = (long long)blockIdx.x * R + rs; g < total; g += (long long)gridDim.x * R) {
            long long t = g;
            const int w = (int)(t % a.W); t /= a.W;
            const int h = (int)(t % a.H); t /= a.H;
            const int d = (int)(t % a.D); const int n = (int)(t / a.D);
            const float4 xv = *reinterpret_cast<const float4*>(a.x + g * C + lane * 4);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const int od = 2 * d + kd, oh = 2 * h + kh, ow = 2 * w + kk;
                        float gdl = 0.f;
                        if (od < Do && oh < Ho && ow < Wo) gdl = a.dlogits[(((long long)n * Do + od) * Ho + oh) * Wo + ow];
                        float4& s = acc[(kd * 3 + kh) * 3 + kk];
                        s.x = fmaf(gdl, xv.x, s.x); s.y = fmaf(gdl, xv.y, s.y); s.z = fmaf(gdl, xv.z, s.z); s.w = fmaf(gdl, xv.w, s.w);
                        if (lane == 0 && kd < 2 && kh < 2 && kk < 2) bsum += gdl;   // each output counted once
                    }
        }
    }
#pragma unroll
    for (int q = 0; q < 27; ++q) {
        red[threadIdx.x] = live ? acc[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        if (threadIdx.x < L) {
            float4 s = red[threadIdx.x];
            for (int r = 1; r < R; ++r) { const float4 v = red[r * L + threadIdx.x]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
            p3d_store_wt4(a.part, (size_t)blockIdx.x * 28 * C + q * C + threadIdx.x * 4, s);      // write-through: see p3d_last_block_wt
        }
        __syncthreads();
    }
    red[threadIdx.x].x = (live && lane == 0) ? bsum : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int r = 0; r < R; ++r) s += red[r * L].x;
        p3d_store_wt(a.part, (size_t)blockIdx.x * 28 * C + 27 * C, s);
    }
    const int nvals = 27 * C + 1;
    const unsigned ngroups = (gridDim.x + HEAD_FOLD - 1) / HEAD_FOLD, group = blockIdx.x / HEAD_FOLD;
    const unsigned first = group * HEAD_FOLD, in_group = min((unsigned)HEAD_FOLD, gridDim.x - first);
    float* gpart = a.part + (size_t)gridDim.x * 28 * C;
    if (!p3d_last_block_wt(a.counter + 1 + group, in_group, &last_flag)) return;
    for (int i = threadIdx.x; i < nvals; i += blockDim.x) {
        float t = 0.f;
#pragma unroll 8
        for (unsigned b = 0; b < in_group; ++b) t += a.part[(size_t)(first + b) * 28 * C + i];
        p3d_store_wt(gpart, (size_t)group * 28 * C + i, t);
    }
    if (!p3d_last_block_wt(a.counter, ngroups, &last_flag)) return;
    for (int i = threadIdx.x; i < nvals; i += blockDim.x) {
        float t = 0.f;
#pragma unroll 8
        for (unsigned gidx = 0; gidx < ngroups; ++gidx) t += gpart[(size_t)gidx * 28 * C + i];
        if (i < 27 * C) a.dk[i] += t; else a.dbias[0] += t;
    }
}

// ---- tf.layers.conv3d(x, 1, 3, 1, 'same') head of the GN decoder-block network (gn/p3d_gn.py:537): the same
// Cout = 1 stencil at stride 1, SAME padding 1 on every side.  logits[o] = bias + sum_k <x[o+k-1,:], K[k,:,0]>.
__global__ __launch_bounds__(256) void headc_fwd_kernel(HeadArgs a) {
    P3D_CHAIN_PRIO();
    extern __shared__ float kw[];     // [27][C]
    const int C = a.C;
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) kw[i] = a.k[i];
    __syncthreads();
    const long long total = (long long)a.N * a.D * a.H * a.W;
    const float bias = a.bias[0];
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        long long t = g;
        const int w = (int)(t % a.W); t /= a.W;
        const int h = (int)(t % a.H); t /= a.H;
        const int d = (int)(t % a.D); const int n = (int)(t / a.D);
        float out = bias;
        for (int kd = 0; kd < 3; ++kd) {
            const int id = d + kd - 1;
            if (id < 0 || id >= a.D) continue;
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = h + kh - 1;
                if (ih < 0 || ih >= a.H) continue;
                for (int kk = 0; kk < 3; ++kk) {
                    const int iw = w + kk - 1;
                    if (iw < 0 || iw >= a.W) continue;
                    const float* xr = a.x + ((((long long)n * a.D + id) * a.H + ih) * a.W + iw) * C;
                    const float* kp = kw + ((kd * 3 + kh) * 3 + kk) * C;
                    for (int c = 0; c < C; c += 4) {
                        const float4 xv = ld4(xr + c);
                        out += xv.x * kp[c] + xv.y * kp[c + 1] + xv.z * kp[c + 2] + xv.w * kp[c + 3];
                    }
                }
            }
        }
        a.logits[g] = out;
        a.pred[g] = a.sigmoid ? 1.f / (1.f + expf(-out)) : out;
    }
}

// dx[i,c] = sum_k dlogits[i-k+1] * K[k,c]
__global__ __launch_bounds__(256) void headc_bwd_input_kernel(HeadArgs a) {
    P3D_CHAIN_PRIO();
    extern __shared__ float kw[];
    const int C = a.C, c4n = C >> 2;
    for (int i = threadIdx.x; i < 27 * C; i += blockDim.x) kw[i] = a.k[i];
    __syncthreads();
    const long long total = (long long)a.N * a.D * a.H * a.W * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i / c4n;
        const int c = (int)(i - t * c4n) << 2;
        const long long g = t;
        const int w = (int)(t % a.W); t /= a.W;
        const int h = (int)(t % a.H); t /= a.H;
        const int d = (int)(t % a.D); const int n = (int)(t / a.D);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int kd = 0; kd < 3; ++kd) {
            const int od = d - kd + 1;
            if (od < 0 || od >= a.D) continue;
            for (int kh = 0; kh < 3; ++kh) {
                const int oh = h - kh + 1;
                if (oh < 0 || oh >= a.H) continue;
                for (int kk = 0; kk < 3; ++kk) {
                    const int ow = w - kk + 1;
                    if (ow < 0 || ow >= a.W) continue;
                    const float gdl = a.dlogits[(((long long)n * a.D + od) * a.H + oh) * a.W + ow];
                    const float* kp = kw + ((kd * 3 + kh) * 3 + kk) * C + c;
                    acc.x += gdl * kp[0]; acc.y += gdl * kp[1]; acc.z += gdl * kp[2]; acc.w += gdl * kp[3];
                }
            }
        }
        *reinterpret_cast<float4*>(a.dx + g * C + c) = acc;
    }
}

// dK[k,c] = sum_i dlogits[i-k+1] * x[i,c]; block = (256/C) position lanes x C channels, 27 accumulators per thread
__global__ __launch_bounds__(256) void headc_bwd_filter_kernel(HeadArgs a) {
    __shared__ float red[256];
    const int C = a.C;
    const int lanes = 256 / C;
    const int c = threadIdx.x % C, sub = threadIdx.x / C;
    const long long total = (long long)a.N * a.D * a.H * a.W;
    float acc[27];
#pragma unroll
    for (int q = 0; q < 27; ++q) acc[q] = 0.f;
    float bsum = 0.f;
    if (sub < lanes) {
        for (long long g = (long long)blockIdx.x * lanes + sub; g < total; g += (long long)gridDim.x * lanes) {
            long long t = g;
            const int w = (int)(t % a.W); t /= a.W;
            const int h = (int)(t % a.H); t /= a.H;
            const int d = (int)(t % a.D); const int n = (int)(t / a.D);
            const float xv = a.x[g * C + c];
            if (c == 0) bsum += a.dlogits[g];
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const int od = d - kd + 1, oh = h - kh + 1, ow = w - kk + 1;
                        float gdl = 0.f;
                        if (od >= 0 && od < a.D && oh >= 0 && oh < a.H && ow >= 0 && ow < a.W)
                            gdl = a.dlogits[(((long long)n * a.D + od) * a.H + oh) * a.W + ow];
                        acc[(kd * 3 + kh) * 3 + kk] += gdl * xv;
                    }
        }
    }
    for (int q = 0; q < 27; ++q) {
        red[threadIdx.x] = (sub < lanes) ? acc[q] : 0.f;
        __syncthreads();
        if (threadIdx.x < C) {
            float s = 0.f;
            for (int l = 0; l < lanes; ++l) s += red[l * C + threadIdx.x];
            p3d_store_wt(a.part, ((size_t)blockIdx.x * 28 + q) * C + threadIdx.x, s);
        }
        __syncthreads();
    }
    red[threadIdx.x] = (sub < lanes && c == 0) ? bsum : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int l = 0; l < lanes; ++l) s += red[l * C];
        p3d_store_wt(a.part, ((size_t)blockIdx.x * 28 + 27) * C, s);
    }
    // the last arriving block adds every block's partial filter gradient in block order (no atomics)
    __shared__ int last_flag;
    if (!p3d_last_block_wt(a.counter, gridDim.x, &last_flag)) return;
    for (int i = threadIdx.x; i < 27 * C + 1; i += blockDim.x) {
        const size_t slot = i < 27 * C ? (size_t)i : (size_t)27 * C;
        float t = 0.f;
#pragma unroll 8
        for (unsigned b = 0; b < gridDim.x; ++b) t += a.part[(size_t)b * 28 * C + slot];
        if (i < 27 * C) a.dk[i] += t; else a.dbias[0] += t;
    }
}

}  // namespace

hipError_t p3d_headc_fwd(const HeadArgs& a, hipStream_t s) {
    if ((a.C & 3) || a.C > 256) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.D * a.H * a.W;
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(headc_fwd_kernel, dim3((unsigned)b), dim3(256), 27 * a.C * sizeof(float), s, a);
    return hipGetLastError();
}

hipError_t p3d_headc_bwd_input(const HeadArgs& a, hipStream_t s) {
    if ((a.C & 3) || a.C > 256) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.D * a.H * a.W * (a.C >> 2);
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(headc_bwd_input_kernel, dim3((unsigned)b), dim3(256), 27 * a.C * sizeof(float), s, a);
    return hipGetLastError();
}

hipError_t p3d_headc_bwd_filter(const HeadArgs& a, hipStream_t s) {
    if (a.C > 256 || a.C < 1 || (256 % a.C)) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.D * a.H * a.W;
    const int lanes = 256 / a.C;
    long long b = (total + (long long)lanes * 32 - 1) / ((long long)lanes * 32);
    if (b > 256) b = 256;       // the last arriving block folds b partial gradients: keep that tail short
    if (b < 1) b = 1;
    HeadArgs aa = a;
    const hipError_t e = p3d_stream_scratch(s, (size_t)b * 28 * a.C, 1, &aa.part, &aa.counter);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(headc_bwd_filter_kernel, dim3((unsigned)b), dim3(256), 0, s, aa);
    return hipGetLastError();
}

hipError_t p3d_head_fwd(const HeadArgs& a, hipStream_t s) {
    if ((a.C & 3) || a.C > 256) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.D * a.H * a.W;
    const int L = a.C >> 2;
    if (L > 1 && L <= 64 && (L & (L - 1)) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0) {
        long long b = (total * L + 255) / 256;
        if (b > 16384) b = 16384;
        const size_t sm = 27 * a.C * sizeof(float);
#define P3D_HL(L_) case L_: hipLaunchKernelGGL(head_fwd_lanes_kernel<L_>, dim3((unsigned)b), dim3(256), sm, s, a); break;
        switch (L) { P3D_HL(2) P3D_HL(4) P3D_HL(8) P3D_HL(16) P3D_HL(32) P3D_HL(64) }
#undef P3D_HL
        return hipGetLastError();
    }
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)b), dim3(256), 27 * a.C * sizeof(float), s, a);
    return hipGetLastError();
}

hipError_t p3d_head_bwd_input(const HeadArgs& a, hipStream_t s) {
    if ((a.C & 3) || a.C > 256) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.D * a.H * a.W * (a.C >> 2);
    long long b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    hipLaunchKernelGGL(head_bwd_input_kernel, dim3((unsigned)b), dim3(256), 27 * a.C * sizeof(float), s, a);
    return hipGetLastError();
}

hipError_t p3d_head_bwd_filter(const HeadArgs& a, hipStream_t s) {
    if (a.C > 256 || a.C < 1) return hipErrorInvalidValue;
    const long long total = (long long)a.N * a.D * a.H * a.W;
    if ((a.C & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0) {
        const int R = 256 / (a.C >> 2);
        long long b = (total + (long long)R * 4 - 1) / ((long long)R * 4);
        if (b > 1024) b = 1024;
        if (b < 1) b = 1;
        const long long groups = (b + HEAD_FOLD - 1) / HEAD_FOLD;
        HeadArgs aa = a;
        const hipError_t e = p3d_stream_scratch(s, (size_t)(b + groups) * 28 * a.C, (int)(1 + groups), &aa.part, &aa.counter);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(head_bwd_filter4_kernel, dim3((unsigned)b), dim3(256), 0, s, aa);
        return hipGetLastError();
    }
    const int lanes = 256 / a.C;
    long long b = (total + (long long)lanes * 32 - 1) / ((long long)lanes * 32);
    if (b > 256) b = 256;       // the last arriving block folds b partial gradients: keep that tail short
    if (b < 1) b = 1;
    HeadArgs aa = a;
    const hipError_t e = p3d_stream_scratch(s, (size_t)b * 28 * a.C, 1, &aa.part, &aa.counter);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(head_bwd_filter_kernel, dim3((unsigned)b), dim3(256), 0, s, aa);
    return hipGetLastError();
}
