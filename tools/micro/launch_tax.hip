// What a dependent launch on one stream costs while ANOTHER stream has a kernel running (the side stream's filter gradients beside
// the main stream's chain of ~850 small launches).   hipcc --offload-arch=gfx950 -O3 launch_tax.hip -o launch_tax
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void tiny_kernel(float* p) { if (p[0] == 123.f) p[1] = 1.f; }
// `blocks` blocks that stay resident for about `us` microseconds each (wall clock), reading a little memory
__global__ __launch_bounds__(256) void resident_kernel(const float* src, float* out, long long ticks) {
    const long long t0 = wall_clock64();
    float acc = 0.f;
    long long i = threadIdx.x + 256ll * blockIdx.x;
    while (wall_clock64() - t0 < ticks) { acc += src[i & 0xfffff]; i += 4096; }
    if (acc == 1.2345f) out[0] = acc;
}

int main() {
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    float *p, *src, *out; CK(hipMalloc(&p, 64)); CK(hipMemset(p, 0, 64)); CK(hipMalloc(&src, 4 << 20)); CK(hipMemset(src, 0, 4 << 20)); CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 2000;
    const long long ticks = 100000000ll * 30 / 1000;       // wall_clock64 runs at 100 MHz: 30 ms
    for (int other = 0; other <= 3; ++other) {
        const int blocks = other == 0 ? 0 : other == 1 ? 1 : other == 2 ? 64 : 256;
        for (int rep = 0; rep < 2; ++rep) {
            if (blocks) hipLaunchKernelGGL(resident_kernel, dim3(blocks), dim3(256), 0, b, src, out, ticks);
            CK(hipEventRecord(e0, a));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, a, p);
            CK(hipEventRecord(e1, a));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipStreamSynchronize(b));
            if (rep == 1) printf("%4d resident block(s) on the other stream: %.2f us per dependent launch of a tiny kernel\n", blocks, ms * 1e3 / N);
        }
    }
    return 0;
}
