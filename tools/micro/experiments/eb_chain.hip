// Micro-benchmark (not part of the product or the tests): the forward pass of stage-3 ST_A bottlenecks (p3d.py:56-63,83-136;
// M = 8 x 2x7x7 = 784 positions, 1024 -> 256 -> 256 -> 256 -> 1024 channels) as a dependent chain with cold weights,
//   (a) every conv followed by its one-launch small-tensor BatchNorm (bn_small_fwd)         -- 8 launches per bottleneck
//   (b) BatchNorm in the conv's own epilogue (BnEpi: granule exchange among the row tiles)  -- 4 launches per bottleneck
// and the two results against each other.  Links the product objects conv_igemm2.o / bn_small.o as they are.
//   usage: eb_chain [reps] [bottlenecks]
#include "../../sap3d_tensorflow_amd/csrc/p3d_kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Conv { int K, Nc, kd, kh, kw; };

static IgemmArgs make_args(const Conv& s, int B, const float* x, float* y, const float* w, const float* bias, const float* zeros) {
    const int D = 2, H = 7, W = 7;
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.N = B; a.Di = D; a.Hi = H; a.Wi = W; a.ldx = s.K; a.K = s.K;
    a.Gd = D; a.Gh = H; a.Gw = W; a.isd = a.ish = a.isw = 1;
    a.y = y; a.Do = D; a.Ho = H; a.Wo = W; a.ldy = s.Nc; a.Nc = s.Nc; a.osd = a.osh = a.osw = 1;
    a.w = w; a.wT = 0; a.zeros = zeros; a.bias = bias;
    int t = 0;
    for (int kd = 0; kd < s.kd; ++kd) for (int kh = 0; kh < s.kh; ++kh) for (int kw = 0; kw < s.kw; ++kw) {
        a.taps[t].dd = (int16_t)(kd - (s.kd - 1) / 2); a.taps[t].dh = (int16_t)(kh - (s.kh - 1) / 2); a.taps[t].dw = (int16_t)(kw - (s.kw - 1) / 2);
        a.taps[t].widx = (int16_t)t; ++t;
    }
    a.ntaps = t;
    return a;
}

__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale, float offset) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned z = (unsigned)i * 2654435761u + seed; z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
        p[i] = offset + ((int)(z & 0xffff) - 32768) * (scale / 32768.f);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const int NB = argc > 2 ? atoi(argv[2]) : 12;          // bottlenecks in the chain (distinct weights: cold)
    const int B = 8, M = B * 98, P = 256;
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* zeros; CK(hipMalloc((void**)&zeros, 1024)); CK(hipMemset(zeros, 0, 1024));
    const Conv cv[4] = {{4 * P, P, 1, 1, 1}, {P, P, 1, 3, 3}, {P, P, 3, 1, 1}, {P, 4 * P, 1, 1, 1}};
    long long woff[5] = {0};
    for (int i = 0; i < 4; ++i) woff[i + 1] = woff[i] + (long long)cv[i].kd * cv[i].kh * cv[i].kw * cv[i].K * cv[i].Nc;
    const long long wtot = woff[4];
    float *w, *bnp, *act[2], *y[4], *z[3], *bias;
    CK(hipMalloc((void**)&w, wtot * NB * 4));
    // per bottleneck and BN: gamma, beta, moving mean, moving var, scale, shift, mean, invstd  (8 x 1024 floats each, 4 BNs)
    const long long bnsz = 8 * 1024;
    CK(hipMalloc((void**)&bnp, bnsz * 4 * NB * 4));
    CK(hipMalloc((void**)&bias, 1024 * 4));
    for (int i = 0; i < 2; ++i) CK(hipMalloc((void**)&act[i], (long long)M * 4 * P * 4));
    for (int i = 0; i < 4; ++i) CK(hipMalloc((void**)&y[i], (long long)M * 4 * P * 4));
    for (int i = 0; i < 3; ++i) CK(hipMalloc((void**)&z[i], (long long)M * P * 4));
    float* x0; CK(hipMalloc((void**)&x0, (long long)M * 4 * P * 4));
    fill_kernel<<<1024, 256, 0, st>>>(x0, (long long)M * 4 * P, 1u, 1.f, 0.3f);
    fill_kernel<<<1024, 256, 0, st>>>(w, wtot * NB, 2u, 0.05f, 0.f);
    fill_kernel<<<64, 256, 0, st>>>(bias, 1024, 5u, 0.1f, 0.f);
    CK(hipStreamSynchronize(st));
    float* flush; const size_t flush_bytes = 512u << 20; CK(hipMalloc((void**)&flush, flush_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    auto reset_bn = [&]() {
        fill_kernel<<<256, 256, 0, st>>>(bnp, bnsz * 4 * NB, 7u, 0.f, 0.f);
        for (long long b = 0; b < 4ll * NB; ++b) {
            fill_kernel<<<8, 256, 0, st>>>(bnp + b * bnsz, 1024, 11u + (unsigned)b, 0.5f, 1.0f);            // gamma in [0.5, 1.5]
            fill_kernel<<<8, 256, 0, st>>>(bnp + b * bnsz + 1024, 1024, 99u + (unsigned)b, 0.3f, 0.f);      // beta
            fill_kernel<<<8, 256, 0, st>>>(bnp + b * bnsz + 3 * 1024, 1024, 0u, 0.f, 1.f);                  // moving var = 1
        }
    };
    auto bn_of = [&](int nb, int i, int C) {
        float* b = bnp + ((long long)nb * 4 + i) * bnsz;
        BnParams p; memset(&p, 0, sizeof(p));
        p.gamma = b; p.beta = b + 1024; p.moving_mean = b + 2048; p.moving_var = b + 3072; p.scale = b + 4096; p.shift = b + 5120;
        p.mean = b + 6144; p.invstd = b + 7168; p.C = C;
        return p;
    };
    // one bottleneck: in -> out (both [M][1024]); eb: BatchNorm in the conv epilogues
    auto bottleneck = [&](int nb, const float* in, float* out, bool eb, bool no_bn = false) {
        const float* src[4] = {in, z[0], z[1], z[2]};
        float* dst[4] = {z[0], z[1], z[2], out};
        for (int i = 0; i < 4; ++i) {
            const float* wi = w + (long long)nb * wtot + woff[i];
            IgemmArgs a = make_args(cv[i], B, src[i], y[i], wi, (i == 1 || i == 2) ? bias : nullptr, zeros);
            const BnParams bp = bn_of(nb, i, cv[i].Nc);
            if (eb) {
                a.eb.mode = i == 3 ? 2 : 1;
                a.eb.z = dst[i]; a.eb.ldz = cv[i].Nc; a.eb.r = i == 3 ? in : nullptr; a.eb.ldr = 4 * P;
                a.eb.gamma = bp.gamma; a.eb.beta = bp.beta; a.eb.scale = bp.scale; a.eb.shift = bp.shift; a.eb.mean = bp.mean; a.eb.invstd = bp.invstd;
                a.eb.moving_mean = bp.moving_mean; a.eb.moving_var = bp.moving_var; a.eb.update_moving = 1;
                a.eb.inv_m = 1.0 / M; a.eb.eps = 1e-3f;
            }
            const P3dIgemmPlan pl = p3d_igemm2_plan(a, 1);
            if (eb && !p3d_igemm2_eb_ok(a, pl)) { fprintf(stderr, "conv %d: plan %dx%d x%d cannot carry the BatchNorm\n", i, pl.bm, pl.bn, pl.splits); exit(2); }
            CK(p3d_launch_igemm2(a, pl, st));
            if (!eb && !no_bn) {
                BnSmallArgs s; memset(&s, 0, sizeof(s));
                s.mode = i == 3 ? 1 : 0; s.M = M; s.C = cv[i].Nc; s.y1 = y[i]; s.ld1 = cv[i].Nc;
                if (i == 3) { s.y2 = in; s.ld2 = 4 * P; }
                s.bn1 = bp; s.batch1 = 1; s.update_moving = 1; s.eps = 1e-3f; s.z = dst[i]; s.ldz = cv[i].Nc;
                CK(p3d_bn_small_fwd(s, st));
            }
        }
    };
    p3d_igemm2_override(0, 0);              // the pipelined kernel throughout (the epilogue BatchNorm lives there)
    {   // timing only: the convs without any BatchNorm (what the launches of their own cost in the chain)
        float best = 1e30f;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipMemsetAsync(flush, r, flush_bytes, st));
            CK(hipEventRecord(e0, st));
            for (int nb = 0; nb < NB; ++nb) bottleneck(nb, act[nb & 1], act[(nb + 1) & 1], false, true);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) best = std::min(best, ms);
        }
        printf("%-46s %8.2f us per bottleneck forward (timing only)\n", "convs alone, no BatchNorm at all", best * 1e3 / NB);
    }
    std::vector<float> res[2], mm[2];
    for (int eb = 0; eb < 2; ++eb) {
        float best = 1e30f;
        for (int r = 0; r < reps + 1; ++r) {
            reset_bn();
            CK(hipMemcpyAsync(act[0], x0, (long long)M * 4 * P * 4, hipMemcpyDeviceToDevice, st));
            CK(hipMemsetAsync(flush, r, flush_bytes, st));            // push the weights out of L2 / Infinity Cache
            CK(hipEventRecord(e0, st));
            for (int nb = 0; nb < NB; ++nb) bottleneck(nb, act[nb & 1], act[(nb + 1) & 1], eb != 0);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) best = std::min(best, ms);
        }
        res[eb].resize((size_t)M * 4 * P); mm[eb].resize(1024);
        CK(hipMemcpy(res[eb].data(), act[NB & 1], res[eb].size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(mm[eb].data(), bn_of(NB - 1, 3, 1024).moving_var, 1024 * 4, hipMemcpyDeviceToHost));
        printf("%-46s %8.2f us per bottleneck forward (%d launches each, chain of %d, cold weights)\n",
               eb ? "BatchNorm in the conv epilogue (BnEpi)" : "conv + bn_small_fwd launches", best * 1e3 / NB, eb ? 4 : 8, NB);
        printf("   granule sweeps that gave up: %lld\n", p3d_eb_timeouts());
    }
    double err = 0, mag = 0, e2 = 0, m2 = 0;
    for (size_t i = 0; i < res[0].size(); ++i) { err = std::max(err, (double)fabsf(res[0][i] - res[1][i])); mag = std::max(mag, (double)fabsf(res[0][i])); }
    for (int i = 0; i < 1024; ++i) { e2 = std::max(e2, (double)fabsf(mm[0][i] - mm[1][i])); m2 = std::max(m2, (double)fabsf(mm[0][i])); }
    printf("output of bottleneck %d: max |a - b| / max |a| = %.2e   (moving variance of its last BatchNorm: %.2e)\n", NB, err / (mag + 1e-30), e2 / (m2 + 1e-30));
    // run-to-run bit equality of the epilogue form
    std::vector<float> again(res[1].size());
    reset_bn();
    CK(hipMemcpyAsync(act[0], x0, (long long)M * 4 * P * 4, hipMemcpyDeviceToDevice, st));
    for (int nb = 0; nb < NB; ++nb) bottleneck(nb, act[nb & 1], act[(nb + 1) & 1], true);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(again.data(), act[NB & 1], again.size() * 4, hipMemcpyDeviceToHost));
    printf("epilogue form twice: %s\n", memcmp(again.data(), res[1].data(), again.size() * 4) == 0 ? "bit-identical" : "DIFFERENT BITS");
    printf("dirty arrival counters: %lld, sweeps that gave up: %lld\n", p3d_scratch_dirty_counters(), p3d_eb_timeouts());
    return 0;
}
