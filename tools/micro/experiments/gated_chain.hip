// Run-ahead chain micro-benchmark (round 5): does a counter-gated EARLY launch of the next chain kernel beat
// the dependent launch?  The product kernels as they are (conv_igemm2.o, bn_small.o, conv_wgrad2.o), on a chain of NB stage-3
// bottleneck forwards at 8 clips (784 rows): conv1 1024->256, bn+relu, convS 1x3x3, bn+relu, convT 3x1x1, bn+relu, conv3 256->1024,
// bn + residual + relu -- 8 dependent launches per bottleneck, every bottleneck with its own (cold) filters and activations.
//   mode plain : all launches on ONE stream (what the product does today)
//   mode events: launches alternate between two streams with an event between consecutive ones (cost of the cross-stream edge alone)
//   mode gated : launches alternate between two streams with NO event; launch k+1 is resident while k runs, prefetches its weights,
//                and waits on the chain counter (p3d_kernels.h, P3dChain); k stores write-through and signals
// Optionally a filter-gradient launch loop runs on a third stream (the side stream's co-resident blocks).
// Checks: the last bottleneck's output is bit-identical in all modes; the fail word stays 0.
//   usage: gated_chain [reps] [replicas] [side: 0/1] [NB]
#include "../../sap3d_tensorflow_amd/csrc/p3d_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long long n, unsigned seed, float scale, float bias) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned z = (unsigned)i * 2654435761u + seed; z ^= z >> 15; z *= 2246822519u; z ^= z >> 13;
        p[i] = ((int)(z & 0xffff) - 32768) * (scale / 32768.f) + bias;
    }
}
static float* dev(long long n) { float* p; CK(hipMalloc((void**)&p, (size_t)n * 4)); return p; }
static void fill(float* p, long long n, unsigned seed, float scale, float bias, hipStream_t s) { fill_kernel<<<512, 256, 0, s>>>(p, n, seed, scale, bias); }

static const int B = 8, D = 2, H = 7, W = 7, P = 256;
static const long long M = (long long)B * D * H * W;

static IgemmArgs conv_args(const float* x, int K, float* y, int Nc, const float* w, int kd, int kh, int kw, const float* zeros) {
    IgemmArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.N = B; a.Di = D; a.Hi = H; a.Wi = W; a.ldx = K; a.K = K;
    a.Gd = D; a.Gh = H; a.Gw = W; a.isd = a.ish = a.isw = 1;
    a.y = y; a.Do = D; a.Ho = H; a.Wo = W; a.ldy = Nc; a.Nc = Nc; a.osd = a.osh = a.osw = 1;
    a.w = w; a.wT = 0; a.zeros = zeros;
    int t = 0;
    for (int a0 = 0; a0 < kd; ++a0) for (int b0 = 0; b0 < kh; ++b0) for (int c0 = 0; c0 < kw; ++c0) {
        a.taps[t].dd = (int16_t)(a0 - (kd - 1) / 2); a.taps[t].dh = (int16_t)(b0 - (kh - 1) / 2); a.taps[t].dw = (int16_t)(c0 - (kw - 1) / 2);
        a.taps[t].widx = (int16_t)t; ++t;
    }
    a.ntaps = t;
    return a;
}
struct BnBuf { float *gamma, *beta, *mm, *mv, *tab; };
static BnSmallArgs bn_args(int mode, const float* y1, const float* y2, int C, const BnBuf& b, float* z) {
    BnSmallArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = mode; a.M = (int)M; a.C = C; a.y1 = y1; a.ld1 = C; a.y2 = y2; a.ld2 = C;
    a.bn1.gamma = b.gamma; a.bn1.beta = b.beta; a.bn1.moving_mean = b.mm; a.bn1.moving_var = b.mv;
    a.bn1.scale = b.tab; a.bn1.shift = b.tab + C; a.bn1.mean = b.tab + 2 * C; a.bn1.invstd = b.tab + 3 * C; a.bn1.C = C;
    a.batch1 = 1; a.update_moving = 0; a.eps = 1e-3f; a.z = z; a.ldz = C;
    return a;
}
static WgradArgs wgrad_args(const float* x, int K, const float* dy, int Nc, float* dw, int kd, int kh, int kw, const float* zeros) {
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.N = B; a.Di = D; a.Hi = H; a.Wi = W; a.ldx = K; a.K = K; a.Gd = D; a.Gh = H; a.Gw = W; a.isd = a.ish = a.isw = 1;
    a.dy = dy; a.ldy = Nc; a.Nc = Nc; a.dw = dw; a.ksplit = 1; a.zeros = zeros; a.polite = 1;
    int t = 0;
    for (int a0 = 0; a0 < kd; ++a0) for (int b0 = 0; b0 < kh; ++b0) for (int c0 = 0; c0 < kw; ++c0) {
        a.taps[t].dd = (int16_t)(a0 - (kd - 1) / 2); a.taps[t].dh = (int16_t)(b0 - (kh - 1) / 2); a.taps[t].dw = (int16_t)(c0 - (kw - 1) / 2);
        a.taps[t].widx = (int16_t)t; ++t;
    }
    a.ntaps = t;
    return a;
}

struct Launch { bool conv; IgemmArgs ia; P3dIgemmPlan pl; BnSmallArgs ba; long long signals; };

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 5;
    const int replicas = argc > 2 ? atoi(argv[2]) : 32;
    const bool with_side = argc > 3 ? atoi(argv[3]) != 0 : false;
    const int NB = argc > 4 ? atoi(argv[4]) : 12;
    CK(hipSetDevice(0));
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t sa, sb, sc;
    CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, least));
    float* zeros = dev(256); CK(hipMemsetAsync(zeros, 0, 1024, sa));
    unsigned* cnt; CK(hipMalloc((void**)&cnt, 32 * P3D_CHAIN_STRIDE * 4 + 128)); CK(hipMemsetAsync(cnt, 0, 32 * P3D_CHAIN_STRIDE * 4 + 128, sa));
    unsigned* fail = cnt + 32 * P3D_CHAIN_STRIDE;

    // ---- the chain ---------------------------------------------------------------------------------------------------------
    std::vector<Launch> chain;
    float* x0 = dev(M * 4 * P); fill(x0, M * 4 * P, 11u, 1.f, 0.f, sa);
    const float* x = x0;
    float* last_out = nullptr;
    for (int b = 0; b < NB; ++b) {
        float* w1 = dev(4ll * P * P); float* wS = dev(9ll * P * P); float* wT = dev(3ll * P * P); float* w3 = dev(4ll * P * P);
        fill(w1, 4ll * P * P, 100u + b, 0.05f, 0.f, sa); fill(wS, 9ll * P * P, 200u + b, 0.05f, 0.f, sa);
        fill(wT, 3ll * P * P, 300u + b, 0.05f, 0.f, sa); fill(w3, 4ll * P * P, 400u + b, 0.05f, 0.f, sa);
        float *y1 = dev(M * P), *z1 = dev(M * P), *yS = dev(M * P), *zS = dev(M * P), *yT = dev(M * P), *zT = dev(M * P);
        float *y3 = dev(M * 4 * P), *out = dev(M * 4 * P);
        BnBuf bn[4];
        for (int q = 0; q < 4; ++q) {
            const int C = q == 3 ? 4 * P : P;
            bn[q].gamma = dev(C); bn[q].beta = dev(C); bn[q].mm = dev(C); bn[q].mv = dev(C); bn[q].tab = dev(4 * C);
            fill(bn[q].gamma, C, 500u + 4 * b + q, 0.5f, 1.f, sa); fill(bn[q].beta, C, 600u + 4 * b + q, 0.3f, 0.f, sa);
            CK(hipMemsetAsync(bn[q].mm, 0, C * 4, sa)); fill(bn[q].mv, C, 1u, 0.f, 1.f, sa);
        }
        auto conv = [&](const float* in, int K, float* o, int Nc, const float* w, int kd, int kh, int kw) {
            Launch l; memset(&l, 0, sizeof(l)); l.conv = true; l.ia = conv_args(in, K, o, Nc, w, kd, kh, kw, zeros);
            l.pl = p3d_igemm2_plan(l.ia, 1); l.signals = p3d_igemm2_tiles(l.ia, l.pl);
            if (!p3d_igemm2_chainable(l.ia)) { fprintf(stderr, "conv not chainable\n"); exit(1); }
            chain.push_back(l);
        };
        auto norm = [&](int mode, const float* in, const float* r, int C, const BnBuf& bb, float* o) {
            Launch l; memset(&l, 0, sizeof(l)); l.conv = false; l.ba = bn_args(mode, in, r, C, bb, o); l.signals = p3d_bn_small_blocks(l.ba);
            chain.push_back(l);
        };
        conv(x, 4 * P, y1, P, w1, 1, 1, 1);   norm(0, y1, nullptr, P, bn[0], z1);
        conv(z1, P, yS, P, wS, 1, 3, 3);      norm(0, yS, nullptr, P, bn[1], zS);
        conv(zS, P, yT, P, wT, 3, 1, 1);      norm(0, yT, nullptr, P, bn[2], zT);
        conv(zT, P, y3, 4 * P, w3, 1, 1, 1);  norm(1, y3, x, 4 * P, bn[3], out);
        x = out; last_out = out;
    }
    // side stream: the four filter gradients of a stage-3 bottleneck as one grouped launch (320 tiles, one 82 KB block per CU)
    float *gxw = dev(M * 4 * P), *gxn = dev(M * P), *gdw = dev(20ll * P * P);
    fill(gxw, M * 4 * P, 21u, 1.f, 0.f, sa); fill(gxn, M * P, 22u, 1.f, 0.f, sa);
    CK(hipMemsetAsync(gdw, 0, 20ll * P * P * 4, sa));
    float* flush; const size_t flush_bytes = 512u << 20; CK(hipMalloc((void**)&flush, flush_bytes));
    CK(hipStreamSynchronize(sa));
    printf("chain: %d bottlenecks, %zu launches; plans:", NB, chain.size());
    for (int i = 0; i < 8; ++i) if (chain[i].conv) printf(" %s x%d (%lld tiles)", chain[i].pl.name, chain[i].pl.splits, chain[i].signals); else printf(" bn(%lld)", chain[i].signals);
    printf("\n");

    hipEvent_t e0, ea, eb, ex; CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb)); CK(hipEventCreateWithFlags(&ex, hipEventDisableTiming));
    std::vector<hipEvent_t> evs(chain.size());
    for (auto& e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    unsigned total = 0;          // counter value the chain has reached (host's running sum; wraps with the device counter)
    std::vector<float> ref(M * 4 * P), got(M * 4 * P);

    auto issue = [&](Launch l, hipStream_t s, bool gated, bool first) {
        P3dChain ch; memset(&ch, 0, sizeof(ch));
        if (gated) { ch.cnt = cnt; ch.fail = fail; ch.wait_for = total; ch.wait = first ? 0 : 1; ch.signal = 1; ch.replicas = replicas; total += (unsigned)l.signals; }
        if (l.conv) { l.ia.chain = ch; CK(p3d_launch_igemm2(l.ia, l.pl, s)); }
        else { l.ba.chain = ch; CK(p3d_bn_small_fwd(l.ba, s)); }
    };
    auto side_burst = [&](int n) {
        for (int i = 0; i < n; ++i) {
            WgradArgs wa[4] = {wgrad_args(gxw, 4 * P, gxn, P, gdw, 1, 1, 1, zeros), wgrad_args(gxn, P, gxn, P, gdw + 4ll * P * P, 1, 3, 3, zeros),
                               wgrad_args(gxn, P, gxn, P, gdw + 13ll * P * P, 3, 1, 1, zeros), wgrad_args(gxn, P, gxw, 4 * P, gdw + 16ll * P * P, 1, 1, 1, zeros)};
            CK(p3d_launch_wgrad2_group(wa, 4, sc));
        }
    };

    const char* names[3] = {"plain (one stream)", "events (two streams, an event per edge)", "gated (two streams, counter-gated run-ahead)"};
    double us_per_launch[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipMemsetAsync(flush, r, flush_bytes, sa));              // filters out of L2 / Infinity Cache
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb)); CK(hipStreamSynchronize(sc));
            if (with_side) side_burst(NB * 2 + 4);                      // keeps the side stream busy for the whole chain
            CK(hipEventRecord(e0, sa));
            CK(hipEventRecord(ex, sa)); CK(hipStreamWaitEvent(sb, ex, 0));       // both streams start behind e0
            for (size_t i = 0; i < chain.size(); ++i) {
                hipStream_t s = (mode == 0 || (i & 1) == 0) ? sa : sb;
                if (mode == 1 && i > 0) CK(hipStreamWaitEvent(s, evs[i - 1], 0));
                issue(chain[i], s, mode == 2, i == 0);
                if (mode == 1) CK(hipEventRecord(evs[i], s));
            }
            CK(hipEventRecord(ea, sa)); CK(hipEventRecord(eb, sb));
            CK(hipEventSynchronize(ea)); CK(hipEventSynchronize(eb));
            float ma, mb; CK(hipEventElapsedTime(&ma, e0, ea)); CK(hipEventElapsedTime(&mb, e0, eb));
            if (r > 0) best = std::min(best, std::max(ma, mb));
            CK(hipStreamSynchronize(sc));
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), last_out, M * 4 * P * 4, hipMemcpyDeviceToHost));
        unsigned hf = 0; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
        if (mode == 0) ref = got;
        double mx = 0; long long diff = 0;
        for (long long i = 0; i < M * 4 * P; ++i) { diff += got[i] != ref[i]; mx = std::max(mx, (double)fabsf(got[i])); }
        us_per_launch[mode] = best * 1e3 / chain.size();
        printf("%-48s %8.1f us per bottleneck, %6.2f us per launch; output %s (max |out| %.3f); fail word %u\n", names[mode],
               best * 1e3 / NB, us_per_launch[mode], diff ? "DIFFERS" : "bit-identical", mx, hf);
        if (diff) printf("   %lld of %lld elements differ\n", diff, M * 4 * P);
        fflush(stdout);
    }
    printf("saved per launch by the gated run-ahead: %.2f us (replicas %d, side stream %s)\n", us_per_launch[0] - us_per_launch[2], replicas,
           with_side ? "busy" : "idle");
    return 0;
}
