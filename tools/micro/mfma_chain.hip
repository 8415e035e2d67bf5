// How fast does one wave per SIMD issue v_mfma_f32_32x32x2_f32?  Dependent chain on one accumulator vs 2 / 4 independent
// accumulators, and 1 / 2 / 4 waves per SIMD.  Prints cycles per MFMA per wave and the implied fraction of the matrix peak.
//   build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_chain.hip -o tools/micro/bin/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(int threads, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<256, threads>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mfmas = (double)iters * 16;                       // per wave
    const double waves = 256.0 * threads / 64;
    const double tf = mfmas * waves * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
    printf("acc chains %d, waves/SIMD %d: %.1f shader-clock ticks per MFMA per wave (timer), %.1f TFLOP/s over 256 blocks (%.3f ms)\n", NACC, threads / 256,
           (double)c / mfmas, tf, ms);
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    for (int threads : {256, 512, 1024}) { run<1>(threads, out, cyc); run<2>(threads, out, cyc); run<4>(threads, out, cyc); }
    return 0;
}
