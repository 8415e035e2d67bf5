// Micro-benchmark behind stem_wgrad.hip: what a wave that owns a 160 x 64 accumulator (10 tiles of v_mfma_f32_32x32x2_f32, 160
// registers, two such waves per SIMD) sustains -- alone, with VALU work between its MFMAs (the BatchNorm arithmetic on the B operand),
// and with LDS reads for the A operand.   hipcc --offload-arch=gfx950 -O3 mfma_mix.hip -o mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VALU, int LDSR, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mix_kernel(float* out, int slots, float k0, float k1) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += WAVES * 64) lds[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc[5][2];
    for (int t = 0; t < 5; ++t) for (int h = 0; h < 2; ++h) for (int e = 0; e < 16; ++e) acc[t][h][e] = 0.f;
    float a[5], b[2];
    for (int t = 0; t < 5; ++t) a[t] = threadIdx.x * 1e-3f + t;
    b[0] = blockIdx.x * 1e-3f; b[1] = b[0] + 1.f;
    const int lane = threadIdx.x & 63;
    for (int s = 0; s < slots; ++s) {
        if (LDSR) {
#pragma unroll
            for (int t = 0; t < 5; ++t) a[t] = lds[(lane + 64 * t + 16 * s) & 8191];
        }
        if (VALU) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v = b[h];
#pragma unroll
                for (int q = 0; q < VALU / 2; ++q) v = fmaf(v, k0, k1);
                b[h] = v;
            }
        }
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[t][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[h], acc[t][h], 0, 0, 0);
    }
    float sum = 0.f;
    for (int t = 0; t < 5; ++t) for (int h = 0; h < 2; ++h) for (int e = 0; e < 16; ++e) sum += acc[t][h][e];
    if (sum == 123.456f) out[0] = sum;
}

template <int VALU, int LDSR, int WAVES>
void run(const char* what, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int slots = 20000, blocks = 256 * 8 / WAVES;          // two waves per SIMD on every CU
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix_kernel<VALU, LDSR, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, out, slots, 1.0001f, 1e-3f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("%-64s %6.1f TFLOP/s of MFMA (%.2f ms)\n", what, (double)blocks * WAVES * slots * 10 * 4096.0 / ms * 1e-9, ms);
    }
}

int main() {
    float* out; CK(hipMalloc(&out, 16));
    run<0, 0, 8>("10 MFMAs per slot, nothing else, 512-thread blocks", out);
    run<0, 0, 4>("10 MFMAs per slot, nothing else, 256-thread blocks", out);
    run<8, 0, 8>("+ 8 dependent VALU per slot", out);
    run<20, 0, 8>("+ 20 dependent VALU per slot", out);
    run<40, 0, 8>("+ 40 dependent VALU per slot", out);
    run<0, 1, 8>("+ 5 ds_read_b32 per slot", out);
    run<20, 1, 8>("+ 20 VALU + 5 ds_read_b32 per slot", out);
    return 0;
}
