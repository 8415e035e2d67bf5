// Tuning tool (not part of the product or the tests): where the time of ONE small-M conv launch (conv_small.hip built with
// -DP3D_TUNE_STAMPS) goes -- per-block wall-clock stamps in a dependent chain of launches with cold weights.
//   usage: sm_stamps   (links conv_igemm2.o + a -DP3D_TUNE_STAMPS build of conv_small.hip)
#include "../../sap3d_tensorflow_amd/csrc/p3d_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
hipError_t p3d_convsm_read_stamps(unsigned long long* host);
struct Shape { const char* name; int K, Nc, kd, kh, kw, wT; };
__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned z = (unsigned)i * 2654435761u + seed; z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
        p[i] = ((int)(z & 0xffff) - 32768) * (scale / 32768.f);
    }
}
int main() {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* zeros; CK(hipMalloc((void**)&zeros, 1024)); CK(hipMemset(zeros, 0, 1024));
    const int B = 8, D = 2, H = 7, W = 7, NW = 8;
    const long long M = (long long)B * D * H * W;
    const Shape shapes[] = {{"conv1 f", 1024, 256, 1, 1, 1, 0}, {"convS f", 256, 256, 1, 3, 3, 0}, {"convT f", 256, 256, 3, 1, 1, 0}, {"conv3 f", 256, 1024, 1, 1, 1, 0},
                            {"convS d", 256, 256, 1, 3, 3, 1}, {"conv3 d", 1024, 256, 1, 1, 1, 1}};
    float* flush; const size_t flush_bytes = 512u << 20; CK(hipMalloc((void**)&flush, flush_bytes));
    for (const Shape& s : shapes) {
        const long long wsz = (long long)s.kd * s.kh * s.kw * s.K * s.Nc;
        float *x, *y, *w;
        CK(hipMalloc((void**)&x, M * s.K * 4)); CK(hipMalloc((void**)&y, M * s.Nc * 4)); CK(hipMalloc((void**)&w, wsz * NW * 4));
        fill_kernel<<<1024, 256, 0, st>>>(x, M * s.K, 1u, 1.f);
        fill_kernel<<<1024, 256, 0, st>>>(w, wsz * NW, 2u, 0.05f);
        IgemmArgs a; memset(&a, 0, sizeof(a));
        a.x = x; a.N = B; a.Di = D; a.Hi = H; a.Wi = W; a.ldx = s.K; a.K = s.K; a.Gd = D; a.Gh = H; a.Gw = W; a.isd = a.ish = a.isw = 1;
        a.y = y; a.Do = D; a.Ho = H; a.Wo = W; a.ldy = s.Nc; a.Nc = s.Nc; a.osd = a.osh = a.osw = 1; a.wT = s.wT; a.zeros = zeros;
        int t = 0;
        for (int kd = 0; kd < s.kd; ++kd) for (int kh = 0; kh < s.kh; ++kh) for (int kw = 0; kw < s.kw; ++kw) {
            a.taps[t].dd = (int16_t)(kd - (s.kd - 1) / 2); a.taps[t].dh = (int16_t)(kh - (s.kh - 1) / 2); a.taps[t].dw = (int16_t)(kw - (s.kw - 1) / 2); a.taps[t].widx = (int16_t)t; ++t;
        }
        a.ntaps = t; a.w = w;
        const P3dIgemmPlan pl = p3d_igemm2_plan(a, 1);
        if (!pl.small) { printf("%s: not a small-M launch\n", s.name); continue; }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(flush, rep, flush_bytes, st));
            for (int i = 0; i < NW; ++i) { a.w = w + (long long)i * wsz; CK(p3d_launch_igemm2(a, pl, st)); }      // stamps: the LAST launch of the chain
            CK(hipStreamSynchronize(st));
        }
        static unsigned long long h[1024][8];
        CK(p3d_convsm_read_stamps(&h[0][0]));
        const int nb = (int)(((M + pl.bm - 1) / pl.bm) * ((s.Nc + pl.bn - 1) / pl.bn));
        unsigned long long t0 = ~0ull, t5 = 0;
        for (int b = 0; b < nb; ++b) { t0 = std::min(t0, h[b][0]); t5 = std::max(t5, h[b][5]); }
        double avg[8] = {0}, mx[8] = {0};
        for (int b = 0; b < nb; ++b) for (int i = 0; i < 8; ++i) { const double v = (double)(h[b][i] - t0) * 0.01; avg[i] += v / nb; mx[i] = std::max(mx[i], v); }
        printf("%-8s %-22s %3d blocks: first entry -> last exit %.2f us | mean (max) us after the first entry: entry %.2f (%.2f)  arguments in %.2f  first tap set %.2f  ring primed %.2f (%.2f)  first data %.2f (%.2f)  loop done %.2f (%.2f)  summed %.2f (%.2f)  exit %.2f (%.2f)\n",
               s.name, pl.name, nb, (double)(t5 - t0) * 0.01, avg[0], mx[0], avg[6], avg[7], avg[1], mx[1], avg[2], mx[2], avg[3], mx[3], avg[4], mx[4], avg[5], mx[5]);
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(w));
    }
    return 0;
}
