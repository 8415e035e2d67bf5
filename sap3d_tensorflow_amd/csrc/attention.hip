// Element / row kernels of the self-attention block `attention()` (reference utils/network.py:157-192, used by
// p3d.p3d_unetplusplus_ds, p3d.py:340-397).  The three products of the block
//     s = g f^T            [N_g x N_f]   (utils/network.py:183)
//     o = softmax(s) h     [N_g x C]     (utils/network.py:184-185)
// and their four gradients are plain per-clip GEMMs and run on the implicit-GEMM / weight-gradient kernels of
// conv_igemm2.hip / conv_wgrad2.hip (a GEMM is a 1x1x1 convolution over the clip's lattice).  What is left here:
//   softmax over the last axis, in place, forward and backward (row-wise, HBM-bound: one read + one write);
//   the mixing  z = relu(bn(conv(o))) * gamma + x  (utils/network.py:191) with its scalar-gamma gradient, and
//   the dropout that follows the last attention block (p3d.py:388) folded into the same pass;
//   row padding of the key / value matrices to a multiple of 4 rows (N_f = 49 at 1x7x7), which keeps every
//   GEMM operand 16-byte aligned; padded score columns are masked out of the softmax and written as 0.
#include "p3d_kernels.h"
#include "det_reduce.h"
#define P3D_SEED(a) ((a).seed_dev ? *(a).seed_dev : (a).seed)   // wave-uniform; device-resident under graph replay

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ float u01(unsigned long long seed, unsigned long long idx) {   // same stream as gn.hip / elementwise.hip
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

template <int TPR>
__device__ __forceinline__ float group_max(float v, float* red) {
    if (TPR == 64) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
        return v;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
template <int TPR>
__device__ __forceinline__ float group_sum(float v, float* red) {
    if (TPR == 64) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        return v;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// TPR threads per row (64: one wave per row, 4 rows per block; 256: one block per row); a row is kept in registers
// when it fits EPT elements per thread, otherwise it is re-read (three passes).
constexpr int EPT = 16;

template <int TPR>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* s, long long rows, int cols, int ld) {
    P3D_CHAIN_PRIO();
    __shared__ float red[4];
    const int rpb = 256 / TPR;
    const int t = threadIdx.x % TPR;
    for (long long r0 = (long long)blockIdx.x * rpb; r0 < rows; r0 += (long long)gridDim.x * rpb) {
        const long long r = r0 + threadIdx.x / TPR;
        const bool live = r < rows;
        float* row = s + (live ? r : 0) * ld;
        if (cols <= TPR * EPT) {
            float v[EPT];
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int c = t + i * TPR;
                v[i] = (live && c < cols) ? row[c] : -INFINITY;
                m = fmaxf(m, v[i]);
            }
            m = group_max<TPR>(m, red);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < EPT; ++i) { v[i] = (t + i * TPR < cols && live) ? expf(v[i] - m) : 0.f; sum += v[i]; }
            sum = group_sum<TPR>(sum, red);
            const float inv = 1.f / sum;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int c = t + i * TPR;
                if (live && c < ld) row[c] = c < cols ? v[i] * inv : 0.f;
            }
        } else {
            float m = -INFINITY;
            for (int c = t; c < cols; c += TPR) m = fmaxf(m, live ? row[c] : -INFINITY);
            m = group_max<TPR>(m, red);
            float sum = 0.f;
            for (int c = t; c < cols; c += TPR) sum += live ? expf(row[c] - m) : 0.f;
            sum = group_sum<TPR>(sum, red);
            const float inv = 1.f / sum;
            if (live) for (int c = t; c < ld; c += TPR) row[c] = c < cols ? expf(row[c] - m) * inv : 0.f;
        }
    }
}

// ds = beta * (dbeta - <beta, dbeta>), written over dbeta; padded columns -> 0
template <int TPR>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* beta, float* d, long long rows, int cols, int ld) {
    P3D_CHAIN_PRIO();
    __shared__ float red[4];
    const int rpb = 256 / TPR;
    const int t = threadIdx.x % TPR;
    for (long long r0 = (long long)blockIdx.x * rpb; r0 < rows; r0 += (long long)gridDim.x * rpb) {
        const long long r = r0 + threadIdx.x / TPR;
        const bool live = r < rows;
        const float* b = beta + (live ? r : 0) * ld;
        float* g = d + (live ? r : 0) * ld;
        if (cols <= TPR * EPT) {
            float bv[EPT], gv[EPT];
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int c = t + i * TPR;
                const bool ok = live && c < cols;
                bv[i] = ok ? b[c] : 0.f; gv[i] = ok ? g[c] : 0.f;
                dot += bv[i] * gv[i];
            }
            dot = group_sum<TPR>(dot, red);
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int c = t + i * TPR;
                if (live && c < ld) g[c] = c < cols ? bv[i] * (gv[i] - dot) : 0.f;
            }
        } else {
            float dot = 0.f;
            for (int c = t; c < cols; c += TPR) dot += live ? b[c] * g[c] : 0.f;
            dot = group_sum<TPR>(dot, red);
            if (live) for (int c = t; c < ld; c += TPR) g[c] = c < cols ? b[c] * (g[c] - dot) : 0.f;
        }
    }
}

// z = r * gamma + x, optional inverted dropout on z (keyed like every other dropout of the path)
__global__ __launch_bounds__(256) void mix_fwd_kernel(AttnMixArgs a) {
    P3D_CHAIN_PRIO();
    const int c4n = a.C >> 2;
    const long long total = a.M * c4n;
    const float gm = a.gamma[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int c = (int)(i - row * c4n) << 2;
        const float4 r = ld4(a.r + row * a.ldr + c), x = ld4(a.x + row * a.ldx + c);
        float4 z = make_float4(fmaf(r.x, gm, x.x), fmaf(r.y, gm, x.y), fmaf(r.z, gm, x.z), fmaf(r.w, gm, x.w));
        if (a.drop_scale > 0.f) {
            const long long e = row * a.C + c;
            z.x *= u01(P3D_SEED(a), e) >= a.drop_rate ? a.drop_scale : 0.f;
            z.y *= u01(P3D_SEED(a), e + 1) >= a.drop_rate ? a.drop_scale : 0.f;
            z.z *= u01(P3D_SEED(a), e + 2) >= a.drop_rate ? a.drop_scale : 0.f;
            z.w *= u01(P3D_SEED(a), e + 3) >= a.drop_rate ? a.drop_scale : 0.f;
        }
        st4(a.z + row * a.ldz + c, z);
    }
}

// dr = dz' * gamma; dx (+)= dz'; dgamma += sum dz' * r        (dz' = dz with the dropout mask)
__global__ __launch_bounds__(256) void mix_bwd_kernel(AttnMixArgs a) {
    P3D_CHAIN_PRIO();
    __shared__ float red[4];
    const int c4n = a.C >> 2;
    const long long total = a.M * c4n;
    const float gm = a.gamma[0];
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int c = (int)(i - row * c4n) << 2;
        float4 dz = ld4(a.dz + row * a.ldz + c);
        if (a.drop_scale > 0.f) {
            const long long e = row * a.C + c;
            dz.x *= u01(P3D_SEED(a), e) >= a.drop_rate ? a.drop_scale : 0.f;
            dz.y *= u01(P3D_SEED(a), e + 1) >= a.drop_rate ? a.drop_scale : 0.f;
            dz.z *= u01(P3D_SEED(a), e + 2) >= a.drop_rate ? a.drop_scale : 0.f;
            dz.w *= u01(P3D_SEED(a), e + 3) >= a.drop_rate ? a.drop_scale : 0.f;
        }
        const float4 r = ld4(a.r + row * a.ldr + c);
        acc += dz.x * r.x + dz.y * r.y + dz.z * r.z + dz.w * r.w;
        st4(a.dr + row * a.ldr + c, make_float4(dz.x * gm, dz.y * gm, dz.z * gm, dz.w * gm));
        float* dx = a.dx + row * a.ldx + c;
        if (a.accx) { const float4 o = ld4(dx); dz.x += o.x; dz.y += o.y; dz.z += o.z; dz.w += o.w; }
        st4(dx, dz);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    // per-block partial; the last arriving block adds them in block order (no atomics)
    __shared__ int last_flag;
    if (threadIdx.x == 0) p3d_store_wt(a.part, blockIdx.x, (red[0] + red[1]) + (red[2] + red[3]));
    if (!p3d_last_block_wt(a.counter, gridDim.x, &last_flag)) return;
    float t = 0.f;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += a.part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) a.dgamma[0] += (red[0] + red[1]) + (red[2] + red[3]);
}

// dst[b][0..Npad) x C  <-  src[b][0..N) x C, rows N..Npad zero  (add = 1: src[b][r] += dst... the reverse, for gradients)
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* src, float* dst, int B, int N, int Npad, int C) {
    P3D_CHAIN_PRIO();
    const int c4n = C >> 2;
    const long long total = (long long)B * Npad * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int c = (int)(i - row * c4n) << 2;
        const int b = (int)(row / Npad), r = (int)(row - (long long)b * Npad);
        st4(dst + row * C + c, r < N ? ld4(src + ((long long)b * N + r) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f));
    }
}
__global__ __launch_bounds__(256) void unpad_rows_kernel(const float* src, float* dst, int B, int N, int Npad, int C) {
    P3D_CHAIN_PRIO();
    const int c4n = C >> 2;
    const long long total = (long long)B * N * c4n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / c4n;
        const int c = (int)(i - row * c4n) << 2;
        const int b = (int)(row / N), r = (int)(row - (long long)b * N);
        st4(dst + row * C + c, ld4(src + ((long long)b * Npad + r) * C + c));
    }
}

inline unsigned capped(long long b, int cap) { return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b)); }

}  // namespace

hipError_t p3d_softmax_rows(float* s, long long rows, int cols, int ld, hipStream_t st) {
    if (cols < 1 || ld < cols) return hipErrorInvalidValue;
    if (cols <= 256) hipLaunchKernelGGL(softmax_fwd_kernel<64>, dim3(capped((rows + 3) / 4, 65535)), dim3(256), 0, st, s, rows, cols, ld);
    else hipLaunchKernelGGL(softmax_fwd_kernel<256>, dim3(capped(rows, 65535)), dim3(256), 0, st, s, rows, cols, ld);
    return hipGetLastError();
}

hipError_t p3d_softmax_rows_bwd(const float* beta, float* d, long long rows, int cols, int ld, hipStream_t st) {
    if (cols < 1 || ld < cols) return hipErrorInvalidValue;
    if (cols <= 256) hipLaunchKernelGGL(softmax_bwd_kernel<64>, dim3(capped((rows + 3) / 4, 65535)), dim3(256), 0, st, beta, d, rows, cols, ld);
    else hipLaunchKernelGGL(softmax_bwd_kernel<256>, dim3(capped(rows, 65535)), dim3(256), 0, st, beta, d, rows, cols, ld);
    return hipGetLastError();
}

hipError_t p3d_attn_mix_fwd(const AttnMixArgs& a, hipStream_t s) {
    if ((a.C & 3) || (a.ldr & 3) || (a.ldx & 3) || (a.ldz & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mix_fwd_kernel, dim3(capped((a.M * (a.C >> 2) + 255) / 256, 8192)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t p3d_attn_mix_bwd(const AttnMixArgs& a, hipStream_t s) {
    if ((a.C & 3) || (a.ldr & 3) || (a.ldx & 3) || (a.ldz & 3)) return hipErrorInvalidValue;
    const unsigned g = capped((a.M * (a.C >> 2) + 255) / 256, 2048);
    AttnMixArgs aa = a;
    const hipError_t e = p3d_stream_scratch(s, g, 1, &aa.part, &aa.counter);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mix_bwd_kernel, dim3(g), dim3(256), 0, s, aa);
    return hipGetLastError();
}

hipError_t p3d_pad_rows(const float* src, float* dst, int B, int N, int Npad, int C, hipStream_t s) {
    if ((C & 3) || Npad < N) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(capped(((long long)B * Npad * (C >> 2) + 255) / 256, 8192)), dim3(256), 0, s, src, dst, B, N, Npad, C);
    return hipGetLastError();
}

hipError_t p3d_unpad_rows(const float* src, float* dst, int B, int N, int Npad, int C, hipStream_t s) {
    if ((C & 3) || Npad < N) return hipErrorInvalidValue;
    hipLaunchKernelGGL(unpad_rows_kernel, dim3(capped(((long long)B * N * (C >> 2) + 255) / 256, 8192)), dim3(256), 0, s, src, dst, B, N, Npad, C);
    return hipGetLastError();
}
