// Micro-benchmark: what this MI355X sustains -- dense fp32 MFMA (v_mfma_f32_32x32x2_f32) and an HBM copy -- next to the
// nominal peaks bench.py prices against (157.3 TFLOP/s, 8 TB/s).  hipcc --offload-arch=gfx950 -O3 peaks.hip -o peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int e = 0; e < 16; ++e) s += acc[t][e];
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void copy_kernel(const float4* src, float4* dst, long long n4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
}
// Eight independent 16-byte loads in flight per lane before the first store (the round-1 kernel above keeps one): the
// form the microarchitecture guide's 6.3 TB/s float4 copy and this library's own streaming kernels (Adam: 5.4 TB/s) use.
template <int U>
__global__ __launch_bounds__(256) void copy_unrolled_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ src, float* out, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    float4 a = make_float4(0, 0, 0, 0);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n4; i += 8 * stride) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

__global__ void empty_kernel(float* p) { if (p == (float*)1) p[0] = 0.f; }
struct Big { char b[336]; };
__global__ void empty_big_kernel(Big a, float* p) { if (p == (float*)1) p[0] = a.b[0]; }

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    {   // host cost of one eager launch on this box (the train step enqueues ~970 of them)
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        float* d; CK(hipMalloc(&d, 16));
        Big big; for (char& c : big.b) c = 1;
        for (int variant = 0; variant < 2; ++variant)
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipStreamSynchronize(st));
                timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
                for (int i = 0; i < 2000; ++i) {
                    if (variant) hipLaunchKernelGGL(empty_big_kernel, dim3(208), dim3(256), 0, st, big, d);
                    else hipLaunchKernelGGL(empty_kernel, dim3(208), dim3(256), 0, st, d);
                }
                clock_gettime(CLOCK_MONOTONIC, &t1);
                CK(hipStreamSynchronize(st));
                timespec t2; clock_gettime(CLOCK_MONOTONIC, &t2);
                auto us = [](timespec a, timespec b) { return (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3; };
                if (rep == 2) printf("eager launch of an empty kernel (%s kernel arguments): host %.2f us per launch, %.2f us per launch until the stream drains\n",
                                     variant ? "336-byte" : "8-byte", us(t0, t1) / 2000, us(t0, t2) / 2000);
            }
    }
    float* out; CK(hipMalloc(&out, 16));
    for (int wps = 1; wps <= 2; ++wps) {           // waves per SIMD: 256 CUs x 4 SIMDs x wps waves
        const int blocks = 256 * wps, iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flop = (double)blocks * 4 /*waves*/ * iters * 32.0 * 4096.0;   // 32 MFMAs x 2*32*32*2 FLOP per iteration
            printf("fp32 MFMA 32x32x2, %d wave(s)/SIMD: %.1f TFLOP/s (%.2f ms)\n", wps, flop / ms * 1e-9, ms);
        }
    }
    const long long bytes = 2ll << 30;
    float4 *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
    CK(hipMemset(src, 1, bytes));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(copy_kernel, dim3(4096), dim3(256), 0, 0, src, dst, bytes / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("HBM copy of %lld MiB, 1 load in flight per lane: %.0f GB/s read+write (%.3f ms)\n", bytes >> 20, 2.0 * bytes / ms * 1e-6, ms);
    }
    for (int blocks : {1024, 2048, 4096, 8192})
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(copy_unrolled_kernel<8>, dim3(blocks), dim3(256), 0, 0, src, dst, bytes / 16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("HBM copy of %lld MiB, 8 loads in flight per lane, %d blocks: %.0f GB/s read+write (%.3f ms)\n", bytes >> 20, blocks, 2.0 * bytes / ms * 1e-6, ms);
        }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(read_kernel, dim3(4096), dim3(256), 0, 0, src, out, bytes / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("HBM read of %lld MiB: %.0f GB/s (%.3f ms)\n", bytes >> 20, 1.0 * bytes / ms * 1e-6, ms);
    }
    return 0;
}
