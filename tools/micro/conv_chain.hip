// Micro-benchmark of the product conv kernels on the launch-latency-bound shapes of stages 2-3 (not part of the product
// or the tests).  Links conv_igemm2.hip / conv_wgrad2.hip as they are and sweeps (tile, K-slices, block mapping) per
// shape: NW dependent launches back to back on one stream, each with its own weights (cold, as in the real net where
// every bottleneck owns its filters), the same activations (warm: just written by the previous kernel).
//   usage: conv_chain [reps]
#include "../../sap3d_tensorflow_amd/csrc/p3d_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int N, D, H, W, K, Nc, kd, kh, kw, wT; };

static IgemmArgs make_args(const Shape& s, const float* x, float* y, const float* w, const float* zeros) {
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.N = s.N; a.Di = s.D; a.Hi = s.H; a.Wi = s.W; a.ldx = s.K; a.K = s.K;
    a.Gd = s.D; a.Gh = s.H; a.Gw = s.W; a.isd = a.ish = a.isw = 1;
    a.y = y; a.Do = s.D; a.Ho = s.H; a.Wo = s.W; a.ldy = s.Nc; a.Nc = s.Nc; a.osd = a.osh = a.osw = 1;
    a.w = w; a.wT = s.wT; a.zeros = zeros;
    int t = 0;
    for (int kd = 0; kd < s.kd; ++kd) for (int kh = 0; kh < s.kh; ++kh) for (int kw = 0; kw < s.kw; ++kw) {
        a.taps[t].dd = (int16_t)(kd - (s.kd - 1) / 2); a.taps[t].dh = (int16_t)(kh - (s.kh - 1) / 2); a.taps[t].dw = (int16_t)(kw - (s.kw - 1) / 2);
        a.taps[t].widx = (int16_t)t; ++t;
    }
    a.ntaps = t;
    return a;
}

static WgradArgs make_wargs(const Shape& s, const float* x, const float* dy, float* dw, const float* zeros) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.N = s.N; a.Di = s.D; a.Hi = s.H; a.Wi = s.W; a.ldx = s.K; a.K = s.K;
    a.Gd = s.D; a.Gh = s.H; a.Gw = s.W; a.isd = a.ish = a.isw = 1;
    a.dy = dy; a.ldy = s.Nc; a.Nc = s.Nc; a.dw = dw; a.ksplit = 1; a.zeros = zeros;
    int t = 0;
    for (int kd = 0; kd < s.kd; ++kd) for (int kh = 0; kh < s.kh; ++kh) for (int kw = 0; kw < s.kw; ++kw) {
        a.taps[t].dd = (int16_t)(kd - (s.kd - 1) / 2); a.taps[t].dh = (int16_t)(kh - (s.kh - 1) / 2); a.taps[t].dw = (int16_t)(kw - (s.kw - 1) / 2);
        a.taps[t].widx = (int16_t)t; ++t;
    }
    a.ntaps = t;
    return a;
}

__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned z = (unsigned)i * 2654435761u + seed; z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
        p[i] = ((int)(z & 0xffff) - 32768) * (scale / 32768.f);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const bool quick = argc > 2;          // any second argument: 64x64 tiles only
    const int NW = 24;                 // distinct weight sets cycled through (cold weights)
    const int NWU = getenv("CHAIN_WARM") ? 1 : NW;      // CHAIN_WARM=1: every launch reads the same weights (hot in L2)
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* zeros; CK(hipMalloc((void**)&zeros, 1024)); CK(hipMemset(zeros, 0, 1024));
    const int B = 8;
    const Shape shapes[] = {
        {"L3conv1 f", B, 2, 7, 7, 1024, 256, 1, 1, 1, 0}, {"L3convS f", B, 2, 7, 7, 256, 256, 1, 3, 3, 0},
        {"L3convT f", B, 2, 7, 7, 256, 256, 3, 1, 1, 0},  {"L3conv3 f", B, 2, 7, 7, 256, 1024, 1, 1, 1, 0},
        {"L3conv1 d", B, 2, 7, 7, 256, 1024, 1, 1, 1, 1}, {"L3convS d", B, 2, 7, 7, 256, 256, 1, 3, 3, 1},
        {"L3conv3 d", B, 2, 7, 7, 1024, 256, 1, 1, 1, 1},
        {"L2conv1 f", B, 4, 14, 14, 512, 128, 1, 1, 1, 0}, {"L2convS f", B, 4, 14, 14, 128, 128, 1, 3, 3, 0},
        {"L2convT f", B, 4, 14, 14, 128, 128, 3, 1, 1, 0}, {"L2conv3 f", B, 4, 14, 14, 128, 512, 1, 1, 1, 0},
        {"L1convS f", B, 8, 28, 28, 64, 64, 1, 3, 3, 0},   {"L1conv3 f", B, 8, 28, 28, 64, 256, 1, 1, 1, 0},
        {"L1conv1 f", B, 8, 28, 28, 256, 64, 1, 1, 1, 0},  {"L1conv1a f", B, 8, 28, 28, 64, 64, 1, 1, 1, 0},
        {"L1conv1 d", B, 8, 28, 28, 64, 256, 1, 1, 1, 1},  {"L1conv3 d", B, 8, 28, 28, 256, 64, 1, 1, 1, 1},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float* flush; const size_t flush_bytes = 512u << 20; CK(hipMalloc((void**)&flush, flush_bytes));
    for (const Shape& s : shapes) {
        const long long M = (long long)s.N * s.D * s.H * s.W;
        const long long wsz = (long long)s.kd * s.kh * s.kw * s.K * s.Nc;
        float *x, *y, *yref, *w;
        CK(hipMalloc((void**)&x, M * s.K * 4)); CK(hipMalloc((void**)&y, M * s.Nc * 4)); CK(hipMalloc((void**)&yref, M * s.Nc * 4));
        CK(hipMalloc((void**)&w, wsz * NW * 4));
        fill_kernel<<<1024, 256, 0, st>>>(x, M * s.K, 1u, 1.f);
        fill_kernel<<<1024, 256, 0, st>>>(w, wsz * NW, 2u, 0.05f);
        CK(hipStreamSynchronize(st));
        std::vector<float> href(M * s.Nc), hy(M * s.Nc);
        bool have_ref = false;
        printf("%-10s M=%lld K=%d Nc=%d taps=%d  GF=%.3f\n", s.name, M, s.K, s.Nc, s.kd * s.kh * s.kw, 2.0 * M * s.K * s.Nc * s.kd * s.kh * s.kw / 1e9);
        const int tiles_cfg[3][2] = {{64, 64}, {128, 64}, {128, 128}};
        for (int ti = 0; ti < 3; ++ti) {
            if (ti == 2 && s.Nc <= 64) continue;
            if (quick && ti > 0) continue;
            const int kst = s.kd * s.kh * s.kw * ((s.K + 31) / 32);
            for (int sp : {1, 2, 4, 8, 16}) {
                if (sp > kst / 2) continue;
                const long long tl = ((M + tiles_cfg[ti][0] - 1) / tiles_cfg[ti][0]) * ((s.Nc + tiles_cfg[ti][1] - 1) / tiles_cfg[ti][1]);
                if (tl * sp > 2048) continue;
                {
                    p3d_igemm2_override(ti, sp);
                    IgemmArgs a = make_args(s, x, y, w, zeros);
                    const P3dIgemmPlan pl = p3d_igemm2_plan(a, 1);
                    float best = 1e30f;
                    for (int r = 0; r < reps + 1; ++r) {
                        if (NWU > 1) CK(hipMemsetAsync(flush, r, flush_bytes, st));        // push the weights out of L2 / Infinity Cache
                        CK(hipEventRecord(e0, st));
                        for (int i = 0; i < NW; ++i) { a.w = w + (long long)(i % NWU) * wsz; CK(p3d_launch_igemm2(a, pl, st)); }
                        CK(hipEventRecord(e1, st));
                        CK(hipEventSynchronize(e1));
                        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                        if (r > 0) best = std::min(best, ms);
                    }
                    // correctness against the first plan, and run-to-run bit equality
                    a.w = w; CK(p3d_launch_igemm2(a, pl, st)); CK(hipStreamSynchronize(st));
                    CK(hipMemcpy(hy.data(), y, M * s.Nc * 4, hipMemcpyDeviceToHost));
                    double err = 0, mag = 0;
                    if (!have_ref) { href = hy; have_ref = true; }
                    for (long long i = 0; i < M * s.Nc; ++i) { err = std::max(err, (double)fabsf(hy[i] - href[i])); mag = std::max(mag, (double)fabsf(href[i])); }
                    std::vector<float> hy2(M * s.Nc);
                    CK(p3d_launch_igemm2(a, pl, st)); CK(hipStreamSynchronize(st));
                    CK(hipMemcpy(hy2.data(), y, M * s.Nc * 4, hipMemcpyDeviceToHost));
                    const bool same = memcmp(hy.data(), hy2.data(), M * s.Nc * 4) == 0;
                    const double us = best * 1e3 / NW;
                    printf("   tile %3dx%-3d splits %2d blocks %4lld : %7.2f us  %6.1f TF/s  relerr %.1e %s\n", pl.bm, pl.bn, pl.splits,
                           tl * sp, us, 2.0 * M * s.K * s.Nc * s.kd * s.kh * s.kw / us / 1e6, err / (mag + 1e-30), same ? "" : "NONDETERMINISTIC");
                    fflush(stdout);
                }
            }
        }
        {   // the planner's own choice (no override): the streaming kernel for the dense 1x1x1 shapes it takes
            p3d_igemm2_override(-1, 0);
            IgemmArgs a = make_args(s, x, y, w, zeros);
            const P3dIgemmPlan pl = p3d_igemm2_plan(a, 1);
            float best = 1e30f;
            for (int r = 0; r < reps + 1; ++r) {
                if (NWU > 1) CK(hipMemsetAsync(flush, r, flush_bytes, st));
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < NW; ++i) { a.w = w + (long long)(i % NWU) * wsz; CK(p3d_launch_igemm2(a, pl, st)); }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) best = std::min(best, ms);
            }
            a.w = w; CK(p3d_launch_igemm2(a, pl, st)); CK(hipStreamSynchronize(st));
            CK(hipMemcpy(hy.data(), y, M * s.Nc * 4, hipMemcpyDeviceToHost));
            double err = 0, mag = 0;
            for (long long i = 0; i < M * s.Nc; ++i) { err = std::max(err, (double)fabsf(hy[i] - href[i])); mag = std::max(mag, (double)fabsf(href[i])); }
            const double us = best * 1e3 / NW;
            printf("   planner: %-24s blocks %4d : %7.2f us  %6.1f TF/s  relerr %.1e\n", pl.name, pl.stream_blocks, us,
                   2.0 * M * s.K * s.Nc * s.kd * s.kh * s.kw / us / 1e6, err / (mag + 1e-30));
            fflush(stdout);
        }
        CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(yref)); CK(hipFree(w));
    }
    p3d_igemm2_override(-1, 0);

    // ---- weight gradients of one stage-3 / stage-2 bottleneck: four launches vs one grouped launch -------------------
    for (int stage = 3; stage >= 2; --stage) {
        const int D = stage == 3 ? 2 : 4, HW = stage == 3 ? 7 : 14, P = stage == 3 ? 256 : 128;
        const Shape bs[4] = {{"conv1", B, D, HW, HW, 4 * P, P, 1, 1, 1, 0}, {"convS", B, D, HW, HW, P, P, 1, 3, 3, 0},
                             {"convT", B, D, HW, HW, P, P, 3, 1, 1, 0}, {"conv3", B, D, HW, HW, P, 4 * P, 1, 1, 1, 0}};
        const long long M = (long long)B * D * HW * HW;
        float *xw, *xn, *dyw, *dyn, *dw;
        CK(hipMalloc((void**)&xw, M * 4 * P * 4)); CK(hipMalloc((void**)&xn, M * P * 4));
        CK(hipMalloc((void**)&dyw, M * 4 * P * 4)); CK(hipMalloc((void**)&dyn, M * P * 4));
        const long long wtot = (long long)(4 * P * P) * 2 + 12ll * P * P;
        CK(hipMalloc((void**)&dw, wtot * NW * 4));
        fill_kernel<<<1024, 256, 0, st>>>(xw, M * 4 * P, 3u, 1.f); fill_kernel<<<1024, 256, 0, st>>>(xn, M * P, 4u, 1.f);
        fill_kernel<<<1024, 256, 0, st>>>(dyw, M * 4 * P, 5u, 1.f); fill_kernel<<<1024, 256, 0, st>>>(dyn, M * P, 6u, 1.f);
        CK(hipStreamSynchronize(st));
        double gf = 0;
        for (auto& s : bs) gf += 2.0 * M * s.K * s.Nc * s.kd * s.kh * s.kw / 1e9;
        for (int grouped = 0; grouped < 2; ++grouped) {
            float best = 1e30f;
            for (int r = 0; r < reps + 1; ++r) {
                CK(hipMemsetAsync(dw, 0, wtot * NW * 4, st));
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < NW; ++i) {
                    float* d = dw + (long long)i * wtot;
                    WgradArgs wa[4] = {make_wargs(bs[0], xw, dyn, d, zeros), make_wargs(bs[1], xn, dyn, d + 4ll * P * P, zeros),
                                       make_wargs(bs[2], xn, dyn, d + 13ll * P * P, zeros), make_wargs(bs[3], xn, dyw, d + 16ll * P * P, zeros)};
                    if (grouped) CK(p3d_launch_wgrad2_group(wa, 4, st));
                    else for (auto& a : wa) CK(p3d_launch_wgrad2(a, st));
                }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) best = std::min(best, ms);
            }
            const double us = best * 1e3 / NW;
            printf("stage %d bottleneck wgrad x4 (%s): %7.2f us per bottleneck  %6.1f TF/s\n", stage, grouped ? "one grouped launch" : "four launches", us, gf * 1e3 / us);
        }
        // determinism + grouped == separate
        std::vector<float> g1(wtot), g2(wtot), g3(wtot);
        for (int pass = 0; pass < 3; ++pass) {
            CK(hipMemsetAsync(dw, 0, wtot * 4, st));
            WgradArgs wa[4] = {make_wargs(bs[0], xw, dyn, dw, zeros), make_wargs(bs[1], xn, dyn, dw + 4ll * P * P, zeros),
                               make_wargs(bs[2], xn, dyn, dw + 13ll * P * P, zeros), make_wargs(bs[3], xn, dyw, dw + 16ll * P * P, zeros)};
            if (pass < 2) CK(p3d_launch_wgrad2_group(wa, 4, st));
            else for (auto& a : wa) CK(p3d_launch_wgrad2(a, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy((pass == 0 ? g1 : pass == 1 ? g2 : g3).data(), dw, wtot * 4, hipMemcpyDeviceToHost));
        }
        double err = 0, mag = 0;
        for (long long i = 0; i < wtot; ++i) { err = std::max(err, (double)fabsf(g1[i] - g3[i])); mag = std::max(mag, (double)fabsf(g3[i])); }
        printf("   grouped twice bit-identical: %s; grouped vs separate relerr %.1e\n", memcmp(g1.data(), g2.data(), wtot * 4) == 0 ? "yes" : "NO", err / (mag + 1e-30));
        CK(hipFree(xw)); CK(hipFree(xn)); CK(hipFree(dyw)); CK(hipFree(dyn)); CK(hipFree(dw));
    }
    return 0;
}
