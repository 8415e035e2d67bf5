// Micro-benchmark: cost of a software grid barrier among co-resident blocks on gfx950 (is a persistent
// multi-op kernel cheaper than dependent launches?).  hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        while (__atomic_load_n(counter, __ATOMIC_RELAXED) < target) __builtin_amdgcn_s_sleep(1);
        __threadfence();
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void rounds_kernel(unsigned* counter, int rounds, float* sink) {
    float v = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        v = v * 1.0001f + 1.f;
        grid_barrier(counter, (unsigned)(r + 1) * gridDim.x);
    }
    if (v == -1.f) sink[0] = v;
}
__global__ void empty_kernel(float* sink) { if (threadIdx.x == 9999) sink[0] = 1.f; }

int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 512, rounds = argc > 2 ? atoi(argv[2]) : 1000;
    unsigned* counter; float* sink;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&sink, 4));
    int maxb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, rounds_kernel, 256, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d, resident blocks/CU %d -> capacity %d, launching %d\n", prop.multiProcessorCount, maxb, maxb * prop.multiProcessorCount, blocks);
    if (blocks > maxb * prop.multiProcessorCount) { printf("too many blocks\n"); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(rounds_kernel, dim3(blocks), dim3(256), 0, 0, counter, rounds, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%d blocks x %d barriers: %.3f ms -> %.2f us per barrier\n", blocks, rounds, ms, 1e3 * ms / rounds);
    }
    CK(hipEventRecord(e0));
    for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, 0, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("1000 dependent empty launches of %d blocks: %.2f us each\n", blocks, ms);
    return 0;
}
