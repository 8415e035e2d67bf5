// Micro-benchmark of the fused-BatchNorm variants of igemm2 (conv_igemm2.hip) on the stage-2/3 shapes: what each
// ingredient (statistics epilogue, coefficient table from published values / from partials, operand transform, gate
// epilogue) adds to a launch in a stream-ordered chain.  Not part of the product or the tests.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/fused_chain.hip sap3d_tensorflow_amd/csrc/conv_igemm2.hip -o tools/micro/bin/fused_chain
#include "../../sap3d_tensorflow_amd/csrc/p3d_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int N, D, H, W, K, Nc, kd, kh, kw, wT; };
__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale, float offset) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned z = (unsigned)i * 2654435761u + seed; z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
        p[i] = ((int)(z & 0xffff) - 32768) * (scale / 32768.f) + offset;
    }
}
static float* dev(long long n, unsigned seed, float scale, float offset, hipStream_t st) {
    float* p; CK(hipMalloc((void**)&p, n * 4));
    fill_kernel<<<512, 256, 0, st>>>(p, n, seed, scale, offset);
    return p;
}
int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const int NW = 24;
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float* pages; CK(hipMalloc((void**)&pages, 2048)); CK(hipMemset(pages, 0, 1024));
    std::vector<unsigned> nan(256, 0x7fc00000u); CK(hipMemcpy(pages + 256, nan.data(), 1024, hipMemcpyHostToDevice));
    const int B = 8;
    const Shape shapes[] = {
        {"L3convS f", B, 2, 7, 7, 256, 256, 1, 3, 3, 0}, {"L3convT f", B, 2, 7, 7, 256, 256, 3, 1, 1, 0}, {"L3conv3 f", B, 2, 7, 7, 256, 1024, 1, 1, 1, 0},
        {"L2convS f", B, 4, 14, 14, 128, 128, 1, 3, 3, 0}, {"L2conv3 f", B, 4, 14, 14, 128, 512, 1, 1, 1, 0},
        {"L3convS d", B, 2, 7, 7, 256, 256, 1, 3, 3, 1}, {"L3conv3 d", B, 2, 7, 7, 1024, 256, 1, 1, 1, 1}, {"L3conv1 d", B, 2, 7, 7, 256, 1024, 1, 1, 1, 1},
        {"L2convS d", B, 4, 14, 14, 128, 128, 1, 3, 3, 1},
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        const long long M = (long long)s.N * s.D * s.H * s.W;
        const long long wsz = (long long)s.kd * s.kh * s.kw * s.K * s.Nc;
        float* x = dev(M * s.K, 1, 1.f, 0.f, st); float* x2 = dev(M * s.K, 7, 1.f, 0.f, st);
        float* y = dev(M * s.Nc, 2, 1.f, 0.f, st); float* yb = dev(M * s.Nc, 3, 1.f, 0.f, st); float* g1 = dev(M * s.Nc, 4, 1.f, 0.f, st); float* g2 = dev(M * s.Nc, 5, 1.f, 0.f, st);
        float* w = dev(wsz * NW, 6, 0.05f, 0.f, st);
        const int C = std::max(s.K, s.Nc);
        float* ones = dev(C, 8, 0.1f, 1.f, st); float* small = dev(C, 9, 0.1f, 0.f, st);
        float* scr = dev(8 * C, 10, 0.f, 0.f, st);       // published outputs
        float* part = dev(64ll * C * 2, 11, 0.f, 30.f, st);   // partials (30, 30): mean small, variance positive
        float* statpart = dev(4096ll * C * 2, 12, 0.f, 0.f, st);
        float* coef = dev(3 * C, 13, 0.1f, 0.5f, st);
        CK(hipStreamSynchronize(st));
        printf("%-10s M=%lld K=%d Nc=%d taps=%d\n", s.name, M, s.K, s.Nc, s.kd * s.kh * s.kw);
        auto base = [&]() {
            IgemmArgs a; memset(&a, 0, sizeof(a));
            a.x = x; a.N = s.N; a.Di = s.D; a.Hi = s.H; a.Wi = s.W; a.ldx = s.K; a.K = s.K;
            a.Gd = s.D; a.Gh = s.H; a.Gw = s.W; a.isd = a.ish = a.isw = 1;
            a.y = y; a.Do = s.D; a.Ho = s.H; a.Wo = s.W; a.ldy = s.Nc; a.Nc = s.Nc; a.osd = a.osh = a.osw = 1;
            a.w = w; a.wT = s.wT; a.zeros = pages;
            int t = 0;
            for (int kd = 0; kd < s.kd; ++kd) for (int kh = 0; kh < s.kh; ++kh) for (int kw = 0; kw < s.kw; ++kw) {
                a.taps[t].dd = (int16_t)(kd - (s.kd - 1) / 2); a.taps[t].dh = (int16_t)(kh - (s.kh - 1) / 2); a.taps[t].dw = (int16_t)(kw - (s.kw - 1) / 2);
                a.taps[t].widx = (int16_t)t; ++t;
            }
            a.ntaps = t;
            return a;
        };
        auto fold = [&](bool parts, int np) {
            BnFold f; memset(&f, 0, sizeof(f));
            f.gamma = ones; f.beta = small; f.C = s.K; f.scale = parts ? scr : ones; f.shift = parts ? scr + C : small; f.mean = scr + 2 * C; f.invstd = scr + 3 * C;
            f.moving_mean = scr + 4 * C; f.moving_var = scr + 5 * C; f.inv_m = 1.0 / (double)M; f.eps = 1e-3f;
            if (parts) { f.part = part; f.nparts = np; f.publish = 1; }
            return f;
        };
        auto gfold = [&](bool parts, int np) {
            BnGradFold f; memset(&f, 0, sizeof(f));
            f.gamma = ones; f.mean = small; f.invstd = ones; f.C = s.K; f.coef = parts ? scr : coef; f.dgamma = scr + 4 * C; f.dbeta = scr + 5 * C; f.inv_m = 1.0 / (double)M;
            if (parts) { f.part = part; f.nparts = np; f.publish = 1; }
            return f;
        };
        auto gate = [&](float* yy, float* out, float* pp) {
            BnGate g; memset(&g, 0, sizeof(g));
            g.y = yy; g.ldy = s.Nc; g.scale = ones; g.shift = small; g.mean = small; g.invstd = ones; g.out = out; g.ldo = s.Nc; g.part = pp;
            return g;
        };
        const int np = (int)((M + 63) / 64) <= P3D_FOLD_MAX ? (int)((M + 63) / 64) : 0;
        struct V { const char* name; IgemmArgs a; };
        std::vector<V> vs;
        { IgemmArgs a = base(); vs.push_back({"plain", a}); }
        if (!s.wT) {
            { IgemmArgs a = base(); a.statpart = statpart; vs.push_back({"plain + stats epilogue", a}); }
            { IgemmArgs a = base(); a.statpart = statpart; a.at_mode = P3D_AT_RELU1; a.f1 = fold(false, 0); vs.push_back({"relu1 (published table) + stats", a}); }
            if (np) { IgemmArgs a = base(); a.statpart = statpart; a.at_mode = P3D_AT_RELU1; a.f1 = fold(true, np); vs.push_back({"relu1 (folds partials) + stats", a}); }
            { IgemmArgs a = base(); a.statpart = statpart; a.at_mode = P3D_AT_RELU2; a.x2 = x2; a.ldx2 = s.K; a.f1 = fold(false, 0); a.f2 = fold(false, 0); vs.push_back({"relu2 (published) + stats", a}); }
        } else {
            { IgemmArgs a = base(); a.ngate = 1; a.gate[0] = gate(yb, g1, statpart); vs.push_back({"gate", a}); }
            { IgemmArgs a = base(); a.ngate = 2; a.gate[0] = gate(yb, g1, statpart); a.gate[1] = gate(y, g2, statpart + 2048ll * C); vs.push_back({"two gates", a}); }
            { IgemmArgs a = base(); a.at_mode = P3D_AT_GRAD; a.x2 = x2; a.ldx2 = s.K; a.gf = gfold(false, 0); vs.push_back({"bngrad (published coef)", a}); }
            if (np) { IgemmArgs a = base(); a.at_mode = P3D_AT_GRAD; a.x2 = x2; a.ldx2 = s.K; a.gf = gfold(true, np); vs.push_back({"bngrad (folds partials)", a}); }
            { IgemmArgs a = base(); a.at_mode = P3D_AT_GRAD; a.x2 = x2; a.ldx2 = s.K; a.gf = gfold(false, 0); a.ngate = 1; a.gate[0] = gate(yb, g1, statpart); vs.push_back({"bngrad (published) + gate", a}); }
        }
        for (auto& v : vs) {
            IgemmArgs a = v.a;
            const P3dIgemmPlan pl = p3d_igemm2_plan(a, 1);
            float best = 1e30f;
            for (int r = 0; r < reps + 1; ++r) {
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < NW; ++i) { a.w = w + (long long)i * wsz; CK(p3d_launch_igemm2(a, pl, st)); }
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) best = std::min(best, ms);
            }
            printf("   %dx%d splits %d  %-70s %7.2f us\n", pl.bm, pl.bn, pl.splits, v.name, best * 1e3 / NW);
            fflush(stdout);
        }
        {   // co-residency: the same plain chain on TWO streams at once (twice the blocks on the chip).  If two blocks that share
            // a CU (two waves per SIMD) interleave for free, both chains finish in the time of one.
            static hipStream_t st2 = nullptr;
            if (!st2) CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
            IgemmArgs a = vs[0].a, b = vs[0].a;
            b.y = yb;
            const P3dIgemmPlan pl = p3d_igemm2_plan(a, 1);
            hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
            float best = 1e30f;
            for (int r = 0; r < reps + 1; ++r) {
                CK(hipEventRecord(e0, st));
                CK(hipStreamWaitEvent(st2, e0, 0));
                for (int i = 0; i < NW; ++i) {
                    a.w = w + (long long)i * wsz; b.w = a.w;
                    CK(p3d_launch_igemm2(a, pl, st)); CK(p3d_launch_igemm2(b, pl, st2));
                }
                CK(hipEventRecord(f1, st2));
                CK(hipStreamWaitEvent(st, f1, 0));
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) best = std::min(best, ms);
            }
            printf("   %dx%d splits %d  %-70s %7.2f us per PAIR of launches\n", pl.bm, pl.bn, pl.splits, "plain, two chains on two streams", best * 1e3 / NW);
        }
        for (float* p : {x, x2, y, yb, g1, g2, w, ones, small, scr, part, statpart, coef}) CK(hipFree(p));
    }
    return 0;
}
