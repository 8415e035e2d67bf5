// Micro-benchmark of the one-launch small-tensor BatchNorm (bn_small.hip) in the setting of a stage-3 bottleneck: NL
// layers with their own tensors (cold: 300+ MB in rotation), each launch behind a small producer kernel that has just
// written its input (as the conv / input-gradient kernel does in the net).  Prints us per BN launch = chain with BN
// minus chain without.   P3D_BN_CB=4|8|16 ./bn_chain    (not part of the product or the tests)
#include "../../sap3d_tensorflow_amd/csrc/p3d_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void producer_kernel(float4* p, long long n4, float v) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p[i] = make_float4(v, v + 1.f, v - 1.f, v * 0.5f);
}

struct Layer { float *y1, *y2, *z, *dz, *dy1, *dy2, *par; };

int main() {
    const int M = 784, NL = 96;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int C : {256, 1024}) {
        std::vector<Layer> L(NL);
        const size_t n = (size_t)M * C;
        for (auto& l : L) {
            for (float** p : {&l.y1, &l.y2, &l.z, &l.dz, &l.dy1, &l.dy2}) { CK(hipMalloc((void**)p, n * 4)); CK(hipMemset(*p, 0, n * 4)); }
            CK(hipMalloc((void**)&l.par, (size_t)16 * C * 4)); CK(hipMemset(l.par, 0, (size_t)16 * C * 4));
            hipLaunchKernelGGL(producer_kernel, dim3(208), dim3(256), 0, s, (float4*)l.y1, (long long)n / 4, 0.3f);
            hipLaunchKernelGGL(producer_kernel, dim3(208), dim3(256), 0, s, (float4*)l.y2, (long long)n / 4, -0.2f);
        }
        CK(hipStreamSynchronize(s));
        auto args = [&](const Layer& l, int mode) {
            BnSmallArgs a; memset(&a, 0, sizeof(a));
            a.mode = mode; a.M = M; a.C = C; a.y1 = l.y1; a.ld1 = C; a.y2 = l.y2; a.ld2 = C;
            float* q = l.par;
            BnParams* b[2] = {&a.bn1, &a.bn2};
            for (int i = 0; i < 2; ++i) {
                b[i]->gamma = q; b[i]->beta = q + C; b[i]->moving_mean = q + 2 * C; b[i]->moving_var = q + 3 * C;
                b[i]->scale = q + 4 * C; b[i]->shift = q + 5 * C; b[i]->mean = q + 6 * C; b[i]->invstd = q + 7 * C; b[i]->C = C;
                q += 8 * C;
            }
            a.batch1 = a.batch2 = 1; a.update_moving = 1; a.eps = 1e-3f;
            a.z = l.z; a.ldz = C; a.dz = l.dz; a.lddz = C; a.dy1 = l.dy1; a.lddy1 = C; a.dy2 = l.dy2; a.lddy2 = C; a.acc2 = mode == 1;
            a.dgamma1 = l.par + 8 * C + 4 * C; a.dbeta1 = a.dgamma1 + C; a.dgamma2 = a.dbeta1 + C; a.dbeta2 = a.dgamma2 + C;
            return a;
        };
        for (int mode : {0, 1}) {
            for (int bwd = 0; bwd < 2; ++bwd) {
                float ms[2] = {0, 0};
                for (int with_bn = 0; with_bn < 2; ++with_bn) {
                    for (int rep = 0; rep < 3; ++rep) {
                        CK(hipEventRecord(e0, s));
                        for (auto& l : L) {
                            float* in = bwd ? l.dz : l.y1;
                            hipLaunchKernelGGL(producer_kernel, dim3(208), dim3(256), 0, s, (float4*)in, (long long)n / 4, 0.1f * (rep + 1));
                            if (with_bn) { const BnSmallArgs a = args(l, mode); CK(bwd ? p3d_bn_small_bwd(a, s) : p3d_bn_small_fwd(a, s)); }
                        }
                        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&ms[with_bn], e0, e1));
                    }
                }
                printf("C=%4d mode %d %s: %.2f us per BN launch (producer-only chain %.2f us per launch)\n", C, mode, bwd ? "bwd" : "fwd",
                       (ms[1] - ms[0]) * 1e3 / NL, ms[0] * 1e3 / NL);
            }
        }
        for (auto& l : L) for (float* p : {l.y1, l.y2, l.z, l.dz, l.dy1, l.dy2, l.par}) CK(hipFree(p));
    }
    return 0;
}
