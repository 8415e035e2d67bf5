// CBAM on the bottleneck residual (reference utils/network.py:198-274, called at gn/p3d_gn.py:175) for gfx950.
//   channel attention : cs[n,c] = sigmoid(MLP(mean_{dhw} x) + MLP(max_{dhw} x)),  MLP = C -> C/8 (ReLU) -> C, shared
//   spatial attention : f = x*cs;  ss[pos] = sigmoid(conv7x7x7([mean_c f, max_c f]))   (2 -> 1 channels, no bias)
//   output            : f * ss   -- never materialised: the block-end pass (gn_apply mode 6) multiplies in place.
// The reference does ~3 extra full passes over the widest tensor of every block; here the forward touches x twice
// (pool, channel-pool of x*cs) and the backward three times.  Everything else lives on [N,C] or [positions]
// vectors.  reduce_max gradients are split equally between tied maxima (TF's _MinOrMaxGrad), which matters here
// because x is post-ReLU and whole channels / positions can be zero.
#include "p3d_kernels.h"
#include "det_reduce.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

// (sum, max, ties) merge
__device__ __forceinline__ void merge(float& s, float& m, float& t, float s2, float m2, float t2) {
    s += s2;
    if (m2 > m) { m = m2; t = t2; } else if (m2 == m) t += t2;
}

// grid (chunks, N); thread = channel (strided if C > 256); rows of the chunk are walked serially (coalesced over c)
__global__ __launch_bounds__(256) void chan_pool_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    const int R = a.D * a.H * a.W;
    const int n = blockIdx.y, ch = blockIdx.x;
    const int per = (R + a.chunks - 1) / a.chunks;
    const int r0 = ch * per, r1 = min(R, r0 + per);
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        float s = 0.f, m = -INFINITY, t = 0.f;
        int r = r0;
        for (; r + 7 < r1; r += 8) {               // eight rows' loads in flight; rows are still merged in order
            float v8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v8[q] = a.x[((long long)n * R + r + q) * a.ld + c];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float v = v8[q];
                s += v;
                if (v > m) { m = v; t = 1.f; } else if (v == m) t += 1.f;
            }
        }
        for (; r < r1; ++r) {
            const float v = a.x[((long long)n * R + r) * a.ld + c];
            s += v;
            if (v > m) { m = v; t = 1.f; } else if (v == m) t += 1.f;
        }
        float* o = a.part + (((long long)n * a.chunks + ch) * a.C + c) * 3;
        o[0] = s; o[1] = m; o[2] = t;
    }
}

// Shared MLP, hidden layer.  grid (ceil(Ch/8), N): every block folds the sample's chunk partials into avg / max
// (block x == 0 publishes them), then computes 8 hidden units: thread = (unit u, one of 32 interleaved slices of
// the C inputs), slices folded through LDS.  The old one-block-per-sample version walked 512 dependent loads per
// thread and cost 180 us per site.
__global__ __launch_bounds__(256) void chan_hidden_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    extern __shared__ float sm[];                  // avg[C] max[C] part[32][8][2]
    float* avg = sm; float* mx = sm + a.C; float* part = mx + a.C;
    const int R = a.D * a.H * a.W, n = blockIdx.y, j0 = blockIdx.x * 8;
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        float s = 0.f, m = -INFINITY, t = 0.f;
        for (int ch = 0; ch < a.chunks; ++ch) {
            const float* o = a.part + (((long long)n * a.chunks + ch) * a.C + c) * 3;
            merge(s, m, t, o[0], o[1], o[2]);
        }
        avg[c] = s / (float)R; mx[c] = m;
        if (blockIdx.x == 0) { a.avg[(long long)n * a.C + c] = avg[c]; a.mx[(long long)n * a.C + c] = m; a.ties[(long long)n * a.C + c] = t; }
    }
    __syncthreads();
    {
        const int u = threadIdx.x & 7, sl = threadIdx.x >> 3, j = j0 + u;
        float sa = 0.f, sb = 0.f;
        if (j < a.Ch) {
#pragma unroll 8
            for (int c = sl; c < a.C; c += 32) { const float w = a.k0[(long long)c * a.Ch + j]; sa += avg[c] * w; sb += mx[c] * w; }
        }
        part[(sl * 8 + u) * 2] = sa; part[(sl * 8 + u) * 2 + 1] = sb;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int u = threadIdx.x & 7, br = threadIdx.x >> 3, j = j0 + u;
        if (j < a.Ch) {
            float s = a.b0[j];
            for (int q = 0; q < 32; ++q) s += part[(q * 8 + u) * 2 + br];
            (br ? a.hmx : a.havg)[(long long)n * a.Ch + j] = fmaxf(s, 0.f);
        }
    }
}

// Output layer + sigmoid.  grid (ceil(C/256), N), thread = channel; k1 rows are read coalesced over c.
__global__ __launch_bounds__(256) void chan_out_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    extern __shared__ float h[];                   // [Ch] = havg + hmax (the MLP is shared, so its outputs add)
    const int n = blockIdx.y;
    for (int j = threadIdx.x; j < a.Ch; j += blockDim.x) h[j] = a.havg[(long long)n * a.Ch + j] + a.hmx[(long long)n * a.Ch + j];
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < a.C) {
        float o = 2.f * a.b1[c];
#pragma unroll 16
        for (int j = 0; j < a.Ch; ++j) o += h[j] * a.k1[(long long)j * a.C + c];
        a.cs[(long long)n * a.C + c] = sigm(o);
    }
}

// wave per position: channel mean / max of f = x * cs
__global__ __launch_bounds__(256) void spat_pool_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    const int R = a.D * a.H * a.W;
    const long long M = (long long)a.N * R;
    const int lane = threadIdx.x & 63;
    for (long long pos = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); pos < M; pos += (long long)gridDim.x * 4) {
        const int n = (int)(pos / R);
        float s = 0.f, m = -INFINITY;
        for (int c = lane * 4; c < a.C; c += 256) {
            const float4 v = ld4(a.x + pos * a.ld + c), k = ld4(a.cs + (long long)n * a.C + c);
            const float f0 = v.x * k.x, f1 = v.y * k.y, f2 = v.z * k.z, f3 = v.w * k.w;
            s += f0 + f1 + f2 + f3;
            m = fmaxf(fmaxf(m, fmaxf(f0, f1)), fmaxf(f2, f3));
        }
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); m = fmaxf(m, __shfl_xor(m, o)); }
        if (lane == 0) { a.sp[pos * 2] = s / (float)a.C; a.sp[pos * 2 + 1] = m; }
    }
}

// eight lanes per position (lane k < 7 takes the kernel plane kd = k; the stage-3 maps have 784 positions, a thread per
// position left the chip to four blocks): 7x7x7 SAME conv over the 2-channel map, planes folded by a fixed shuffle tree, sigmoid
__global__ __launch_bounds__(256) void spat_conv_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    __shared__ float kw[686];
    for (int i = threadIdx.x; i < 686; i += blockDim.x) kw[i] = a.k7[i];
    __syncthreads();
    const int R = a.D * a.H * a.W;
    const long long M = (long long)a.N * R;
    const int kd = threadIdx.x & 7;
    for (long long p0 = (long long)blockIdx.x * 32; p0 < M; p0 += (long long)gridDim.x * 32) {
        const long long pos = p0 + (threadIdx.x >> 3);
        float acc = 0.f;
        if (pos < M && kd < 7) {
            long long t = pos;
            const int w = (int)(t % a.W); t /= a.W;
            const int h = (int)(t % a.H); t /= a.H;
            const int d = (int)(t % a.D); const int n = (int)(t / a.D);
            const int id = d + kd - 3;
            if ((unsigned)id < (unsigned)a.D) {
                for (int kh = 0; kh < 7; ++kh) {
                    const int ih = h + kh - 3;
                    if ((unsigned)ih >= (unsigned)a.H) continue;
                    // the seven taps of a kernel row: loads first (taps outside the map read as 0), sums in tap order
                    const float* row = a.sp + (((long long)n * a.D + id) * a.H + ih) * a.W * 2;
                    float2 v7[7];
#pragma unroll
                    for (int kk = 0; kk < 7; ++kk) {
                        const int iw = w + kk - 3;
                        v7[kk] = (unsigned)iw < (unsigned)a.W ? *reinterpret_cast<const float2*>(row + iw * 2) : make_float2(0.f, 0.f);
                    }
#pragma unroll
                    for (int kk = 0; kk < 7; ++kk) {
                        const float* k = kw + ((kd * 7 + kh) * 7 + kk) * 2;
                        if ((unsigned)(w + kk - 3) < (unsigned)a.W) acc += v7[kk].x * k[0] + v7[kk].y * k[1];
                    }
                }
            }
        }
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
        if (pos < M && kd == 0) a.ss[pos] = sigm(acc);
    }
}

// ---- backward ---------------------------------------------------------------------------------------------
// wave per position: dss = sum_c dout*f ; dpre = dss * ss * (1 - ss)
__global__ __launch_bounds__(256) void bwd_dpre_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    const int R = a.D * a.H * a.W;
    const long long M = (long long)a.N * R;
    const int lane = threadIdx.x & 63;
    for (long long pos = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); pos < M; pos += (long long)gridDim.x * 4) {
        const int n = (int)(pos / R);
        float s = 0.f;
        for (int c = lane * 4; c < a.C; c += 256) {
            const float4 v = ld4(a.x + pos * a.ld + c), k = ld4(a.cs + (long long)n * a.C + c), g = ld4(a.dout + pos * a.C + c);
            s += g.x * v.x * k.x + g.y * v.y * k.y + g.z * v.z * k.z + g.w * v.w * k.w;
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) { const float ss = a.ss[pos]; a.dpre[pos] = s * ss * (1.f - ss); }
    }
}

// eight lanes per position (lane k < 7: kernel plane kd = k): dsp[pos][ch] = sum_taps dpre[pos - tap + 3] * K[tap][ch]
// (transpose of the SAME conv), planes folded by a fixed shuffle tree
__global__ __launch_bounds__(256) void bwd_spat_conv_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    __shared__ float kw[686];
    for (int i = threadIdx.x; i < 686; i += blockDim.x) kw[i] = a.k7[i];
    __syncthreads();
    const int R = a.D * a.H * a.W;
    const long long M = (long long)a.N * R;
    const int kd = threadIdx.x & 7;
    for (long long p0 = (long long)blockIdx.x * 32; p0 < M; p0 += (long long)gridDim.x * 32) {
        const long long pos = p0 + (threadIdx.x >> 3);
        float a0 = 0.f, a1 = 0.f;
        if (pos < M && kd < 7) {
            long long t = pos;
            const int w = (int)(t % a.W); t /= a.W;
            const int h = (int)(t % a.H); t /= a.H;
            const int d = (int)(t % a.D); const int n = (int)(t / a.D);
            const int od = d - kd + 3;
            if ((unsigned)od < (unsigned)a.D) {
                for (int kh = 0; kh < 7; ++kh) {
                    const int oh = h - kh + 3;
                    if ((unsigned)oh >= (unsigned)a.H) continue;
                    const float* row = a.dpre + (((long long)n * a.D + od) * a.H + oh) * a.W;
                    float g7[7];
#pragma unroll
                    for (int kk = 0; kk < 7; ++kk) {
                        const int ow = w - kk + 3;
                        g7[kk] = (unsigned)ow < (unsigned)a.W ? row[ow] : 0.f;
                    }
#pragma unroll
                    for (int kk = 0; kk < 7; ++kk) {
                        const float* k = kw + ((kd * 7 + kh) * 7 + kk) * 2;
                        if ((unsigned)(w - kk + 3) < (unsigned)a.W) { a0 += g7[kk] * k[0]; a1 += g7[kk] * k[1]; }
                    }
                }
            }
        }
        a0 += __shfl_xor(a0, 1); a0 += __shfl_xor(a0, 2); a0 += __shfl_xor(a0, 4);
        a1 += __shfl_xor(a1, 1); a1 += __shfl_xor(a1, 2); a1 += __shfl_xor(a1, 4);
        if (pos < M && kd == 0) { a.dsp[pos * 2] = a0; a.dsp[pos * 2 + 1] = a1; }
    }
}

// dK7[tap][ch] = sum_pos sp[pos + tap - 3][ch] * dpre[pos].  Block = a chunk of 64 positions (coordinates and dpre
// staged in LDS once), thread = one tap (343 of 384 threads); one atomic pair per (block, tap).
__global__ __launch_bounds__(384) void bwd_k7_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    __shared__ int pd[64], ph[64], pw[64], pn[64];
    __shared__ float pg[64];
    const int R = a.D * a.H * a.W;
    const long long M = (long long)a.N * R;
    const int tap = threadIdx.x;
    const int kd = tap / 49 - 3, kh = (tap / 7) % 7 - 3, kk = tap % 7 - 3;
    float a0 = 0.f, a1 = 0.f;
    for (long long base = (long long)blockIdx.x * 64; base < M; base += (long long)gridDim.x * 64) {
        __syncthreads();
        if (threadIdx.x < 64) {
            const long long pos = base + threadIdx.x;
            if (pos < M) {
                long long t = pos;
                pw[threadIdx.x] = (int)(t % a.W); t /= a.W;
                ph[threadIdx.x] = (int)(t % a.H); t /= a.H;
                pd[threadIdx.x] = (int)(t % a.D); pn[threadIdx.x] = (int)(t / a.D);
                pg[threadIdx.x] = a.dpre[pos];
            } else { pn[threadIdx.x] = -1; pg[threadIdx.x] = 0.f; pd[threadIdx.x] = ph[threadIdx.x] = pw[threadIdx.x] = 0; }
        }
        __syncthreads();
        if (tap < 343) {
            for (int i0 = 0; i0 < 64; i0 += 8) {       // eight positions' loads in flight; summed in position order
                float2 v8[8];
                bool ok[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = i0 + q, n = pn[i];
                    const int id = pd[i] + kd, ih = ph[i] + kh, iw = pw[i] + kk;
                    ok[q] = n >= 0 && (unsigned)id < (unsigned)a.D && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
                    v8[q] = ok[q] ? *reinterpret_cast<const float2*>(a.sp + ((((long long)n * a.D + id) * a.H + ih) * a.W + iw) * 2)
                                  : make_float2(0.f, 0.f);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (ok[q]) { a0 += pg[i0 + q] * v8[q].x; a1 += pg[i0 + q] * v8[q].y; }
            }
        }
    }
    // per-block partials; the last arriving block adds them in block order (no atomics)
    __shared__ int last_flag;
    if (tap < 343) { p3d_store_wt(a.k7part, ((size_t)blockIdx.x * 343 + tap) * 2, a0); p3d_store_wt(a.k7part, ((size_t)blockIdx.x * 343 + tap) * 2 + 1, a1); }
    if (!p3d_last_block_wt(a.k7counter, gridDim.x, &last_flag)) return;
    if (tap < 343) {
        float t0 = 0.f, t1 = 0.f;
        unsigned b = 0;
        for (; b + 7 < gridDim.x; b += 8) {            // block order, eight partials in flight
            float2 v8[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v8[q] = *reinterpret_cast<const float2*>(a.k7part + ((size_t)(b + q) * 343 + tap) * 2);
#pragma unroll
            for (int q = 0; q < 8; ++q) { t0 += v8[q].x; t1 += v8[q].y; }
        }
        for (; b < gridDim.x; ++b) { t0 += a.k7part[((size_t)b * 343 + tap) * 2]; t1 += a.k7part[((size_t)b * 343 + tap) * 2 + 1]; }
        a.dk7[tap * 2] += t0; a.dk7[tap * 2 + 1] += t1;
    }
}

// grid (chunks, N), wave per position inside the chunk: df = dout*ss + dmean/C + dmax*[f == max]/ties;
// dx (+)= df*cs ; per-(n,c) partial of dcs = sum_pos df*x kept in registers, folded through LDS.
__global__ __launch_bounds__(256) void bwd_df_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    extern __shared__ float red[];                 // [4][C]
    const int R = a.D * a.H * a.W;
    const int n = blockIdx.y, ch = blockIdx.x;
    const int per = (R + a.chunks - 1) / a.chunks;
    const int r0 = ch * per, r1 = min(R, r0 + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 dcs[4];                                  // C <= 1024: up to 4 float4 per lane
#pragma unroll
    for (int q = 0; q < 4; ++q) dcs[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0 + wave; r < r1; r += 4) {
        const long long pos = (long long)n * R + r;
        const float ss = a.ss[pos], dmean = a.dsp[pos * 2] / (float)a.C, dmax = a.dsp[pos * 2 + 1], fmx = a.sp[pos * 2 + 1];
        // ties of the channel max at this position
        float ties = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane * 4 + q * 256;
            if (c < a.C) {
                const float4 v = ld4(a.x + pos * a.ld + c), k = ld4(a.cs + (long long)n * a.C + c);
                ties += (v.x * k.x == fmx) + (v.y * k.y == fmx) + (v.z * k.z == fmx) + (v.w * k.w == fmx);
            }
        }
        for (int o = 32; o > 0; o >>= 1) ties += __shfl_xor(ties, o);
        const float dmx = dmax / ties;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane * 4 + q * 256;
            if (c < a.C) {
                const float4 v = ld4(a.x + pos * a.ld + c), k = ld4(a.cs + (long long)n * a.C + c), g = ld4(a.dout + pos * a.C + c);
                float4 df;
                df.x = g.x * ss + dmean + (v.x * k.x == fmx ? dmx : 0.f);
                df.y = g.y * ss + dmean + (v.y * k.y == fmx ? dmx : 0.f);
                df.z = g.z * ss + dmean + (v.z * k.z == fmx ? dmx : 0.f);
                df.w = g.w * ss + dmean + (v.w * k.w == fmx ? dmx : 0.f);
                dcs[q].x += df.x * v.x; dcs[q].y += df.y * v.y; dcs[q].z += df.z * v.z; dcs[q].w += df.w * v.w;
                float4 d = make_float4(df.x * k.x, df.y * k.y, df.z * k.z, df.w * k.w);
                float* dst = a.dx + pos * a.lddx + c;
                if (a.accx) { const float4 o = ld4(dst); d.x += o.x; d.y += o.y; d.z += o.z; d.w += o.w; }
                st4(dst, d);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane * 4 + q * 256;
        if (c < a.C) st4(red + wave * a.C + c, dcs[q]);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += blockDim.x)
        a.dcs_part[((long long)n * a.chunks + ch) * a.C + c] = red[c] + red[a.C + c] + red[2 * a.C + c] + red[3 * a.C + c];
}

// MLP backward, step 1.  grid (ceil(Ch/8), N): every block folds dO[n,c] = dcs*cs*(1-cs) into LDS (block x == 0
// publishes it in a.dO), then 8 hidden units per block: 32 lanes per unit walk the k1 row coalesced and fold by
// shuffles.  dh [n][2][Ch] = gradients of the hidden activations of the avg / max branch.
__global__ __launch_bounds__(256) void bwd_mlp1_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    extern __shared__ float dO[];                  // [C]
    const int n = blockIdx.y, j0 = blockIdx.x * 8;
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        float s = 0.f;
#pragma unroll 8
        for (int ch = 0; ch < a.chunks; ++ch) s += a.dcs_part[((long long)n * a.chunks + ch) * a.C + c];
        const float cs = a.cs[(long long)n * a.C + c];
        dO[c] = s * cs * (1.f - cs);
        if (blockIdx.x == 0) a.dO[(long long)n * a.C + c] = dO[c];
    }
    __syncthreads();
    const int sl = threadIdx.x & 31, j = j0 + (threadIdx.x >> 5);
    float s = 0.f;
    if (j < a.Ch) {
#pragma unroll 8
        for (int c = sl; c < a.C; c += 32) s += a.k1[(long long)j * a.C + c] * dO[c];
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (sl == 0 && j < a.Ch) {
        const float ha = a.havg[(long long)n * a.Ch + j], hm = a.hmx[(long long)n * a.Ch + j];
        a.dh[(long long)n * 2 * a.Ch + j] = ha > 0.f ? s : 0.f;
        a.dh[(long long)n * 2 * a.Ch + a.Ch + j] = hm > 0.f ? s : 0.f;
    }
}

// MLP backward, step 2a -- thread per (n, c): gradients of the pooled vectors; block 0 also folds db0, and the
// threads of sample 0 fold db1.
__global__ __launch_bounds__(256) void bwd_mlp2a_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.N * a.C) {
        const int n = i / a.C, c = i - n * a.C;
        const float* dh = a.dh + (long long)n * 2 * a.Ch;
        const float* w = a.k0 + (long long)c * a.Ch;
        float da = 0.f, dm = 0.f;
        // (the k0 row of a thread is contiguous: 16-byte loads, eight in flight; same order of the sums as the scalar loop)
        int j = 0;
        if ((a.Ch & 3) == 0)
            for (; j + 31 < a.Ch; j += 32) {
                float4 wv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) wv[q] = *reinterpret_cast<const float4*>(w + j + 4 * q);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int b = j + 4 * q;
                    da += wv[q].x * dh[b]; dm += wv[q].x * dh[a.Ch + b];
                    da += wv[q].y * dh[b + 1]; dm += wv[q].y * dh[a.Ch + b + 1];
                    da += wv[q].z * dh[b + 2]; dm += wv[q].z * dh[a.Ch + b + 2];
                    da += wv[q].w * dh[b + 3]; dm += wv[q].w * dh[a.Ch + b + 3];
                }
            }
        for (; j < a.Ch; ++j) { da += w[j] * dh[j]; dm += w[j] * dh[a.Ch + j]; }
        a.davg[i] = da; a.dmx[i] = dm;
        if (n == 0) {
            float s = 0.f;
#pragma unroll 8
            for (int m = 0; m < a.N; ++m) s += a.dO[(long long)m * a.C + c];
            a.db1[c] += 2.f * s;
        }
    }
    if (blockIdx.x == 0)
        for (int j = threadIdx.x; j < a.Ch; j += blockDim.x) {
            float s = 0.f;
            for (int n = 0; n < a.N; ++n) s += a.dh[(long long)n * 2 * a.Ch + j] + a.dh[(long long)n * 2 * a.Ch + a.Ch + j];
            a.db0[j] += s;
        }
}
// step 2b -- thread per weight element (c, j): dk0[c][j], dk1[j][c] (each owned by one thread: plain RMW)
__global__ __launch_bounds__(256) void bwd_mlp2b_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.C * a.Ch) return;
    const int c = i / a.Ch, j = i - c * a.Ch;
    float g0 = 0.f, g1 = 0.f;
    for (int n = 0; n < a.N; ++n) {
        const float* dh = a.dh + (long long)n * 2 * a.Ch;
        g0 += a.avg[(long long)n * a.C + c] * dh[j] + a.mx[(long long)n * a.C + c] * dh[a.Ch + j];
        g1 += (a.havg[(long long)n * a.Ch + j] + a.hmx[(long long)n * a.Ch + j]) * a.dO[(long long)n * a.C + c];
    }
    a.dk0[i] += g0;
    a.dk1[(long long)j * a.C + c] += g1;
}

// dx += davg/R + dmax * [x == max over the sample's rows] / ties
__global__ __launch_bounds__(256) void bwd_chan_kernel(CbamArgs a) {
    p3d_warm_kernargs<CbamArgs>();
    P3D_CHAIN_PRIO();
    const int R = a.D * a.H * a.W, c4n = a.C >> 2;
    const long long total = (long long)a.N * R * c4n;
    const float invR = 1.f / (float)R;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long pos = i / c4n;
        const int c = (int)(i - pos * c4n) << 2;
        const long long t = (pos / R) * a.C + c;
        const float4 v = ld4(a.x + pos * a.ld + c), m = ld4(a.mx + t), ti = ld4(a.ties + t), da = ld4(a.davg + t), dm = ld4(a.dmx + t);
        float* dst = a.dx + pos * a.lddx + c;
        float4 d = ld4(dst);
        d.x += da.x * invR + (v.x == m.x ? dm.x / ti.x : 0.f);
        d.y += da.y * invR + (v.y == m.y ? dm.y / ti.y : 0.f);
        d.z += da.z * invR + (v.z == m.z ? dm.z / ti.z : 0.f);
        d.w += da.w * invR + (v.w == m.w ? dm.w / ti.w : 0.f);
        st4(dst, d);
    }
}

inline unsigned capped(long long b, int cap) { return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b)); }

}  // namespace

hipError_t p3d_cbam_forward(const CbamArgs& a, hipStream_t s) {
    if ((a.C & 3) || a.C > 1024 || a.Ch < 1 || a.chunks < 1) return hipErrorInvalidValue;
    const long long M = (long long)a.N * a.D * a.H * a.W;
    hipLaunchKernelGGL(chan_pool_kernel, dim3(a.chunks, a.N), dim3(256), 0, s, a);
    hipLaunchKernelGGL(chan_hidden_kernel, dim3((a.Ch + 7) / 8, a.N), dim3(256), (2 * a.C + 512) * sizeof(float), s, a);
    hipLaunchKernelGGL(chan_out_kernel, dim3((a.C + 255) / 256, a.N), dim3(256), a.Ch * sizeof(float), s, a);
    hipLaunchKernelGGL(spat_pool_kernel, dim3(capped((M + 3) / 4, 8192)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(spat_conv_kernel, dim3(capped((M + 31) / 32, 8192)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t p3d_cbam_backward(const CbamArgs& a, hipStream_t s) {
    if ((a.C & 3) || a.C > 1024) return hipErrorInvalidValue;
    const long long M = (long long)a.N * a.D * a.H * a.W;
    hipLaunchKernelGGL(bwd_dpre_kernel, dim3(capped((M + 3) / 4, 8192)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bwd_spat_conv_kernel, dim3(capped((M + 31) / 32, 8192)), dim3(256), 0, s, a);
    {
        const unsigned g7 = capped((M + 63) / 64, 256);      // the last arriver folds g7 partials: keep that tail short
        CbamArgs a7 = a;
        const hipError_t e7 = p3d_stream_scratch(s, (size_t)g7 * 343 * 2, 1, &a7.k7part, &a7.k7counter);
        if (e7 != hipSuccess) return e7;
        hipLaunchKernelGGL(bwd_k7_kernel, dim3(g7), dim3(384), 0, s, a7);
    }
    hipLaunchKernelGGL(bwd_df_kernel, dim3(a.chunks, a.N), dim3(256), 4 * a.C * sizeof(float), s, a);
    hipLaunchKernelGGL(bwd_mlp1_kernel, dim3((a.Ch + 7) / 8, a.N), dim3(256), a.C * sizeof(float), s, a);
    hipLaunchKernelGGL(bwd_mlp2a_kernel, dim3((a.N * a.C + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bwd_mlp2b_kernel, dim3((a.C * a.Ch + 255) / 256), dim3(256), 0, s, a);
    hipLaunchKernelGGL(bwd_chan_kernel, dim3(capped((M * (a.C >> 2) + 255) / 256, 4096)), dim3(256), 0, s, a);
    return hipGetLastError();
}
