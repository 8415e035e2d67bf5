// Weight gradient of the P3D convolutions on gfx950 (TF Conv3DBackpropFilterV2 behind tf.nn.conv3d / tf.layers.conv3d /
// conv3d_transpose at reference p3d.py:19,24,86,112,125,172,200-217): for every kernel tap,
//   dW[tap][k][n] += sum over the dense lattice m of  Xgathered[m + tap][k] * dY[m][n],
// with the machinery of conv_igemm2.hip: both operands are [position][channel] rows exactly as
// they lie in NDHWC memory, streamed global -> LDS by LDS-DMA into a 3-stage ring (one raw
// s_barrier + one counted vmcnt per 32-position step), consumed by v_mfma_f32_32x32x2_f32 with
// k = position.  Padded / out-of-range rows come from a zero page.
//
// GROUPED: one launch carries up to P3D_WGRAD_GROUP independent problems (the filter gradients of one
// bottleneck: 1x1x1 reduce, 1x3x3, 3x1x1, 1x1x1 expand -- tf.gradients of p3d.py:86-125).  Stage 3 has
// only 784 positions to reduce over, so one problem offers 48-144 output tiles; four together fill the
// 256 CUs without cutting the position range, i.e. without any cross-block sum.
//
// DETERMINISTIC: where the position range IS cut (big tensors, few filter tiles), every cut stores its
// partial tile to a scratch slab and takes an arrival ticket; the block with the last ticket adds the
// slabs in cut order and updates dW (and the bias gradient) with plain read-modify-writes.  No float
// atomics anywhere: two runs give bit-identical gradients.
//
// FUSED BatchNorm (round 3, template parameter FUSED; see p3d_kernels.h "BatchNorm fused into the convolutions' operand
// paths"): a problem may read its gathered operand as relu(s1*x + t1) [+ relu(s2*x2 + t2)] and its dense operand as
// k1*dy + k2*dy2 + k3 -- the normalised activations and BatchNorm's input gradients are never stored.  The per-channel
// coefficients sit in registers (the channels of a lane's DMA chunks are fixed for the whole kernel); the lane that fetched
// a chunk transforms it on its way into the tile the fragments are read from, padded / out-of-range chunks stay zero.
#include "p3d_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BKM = 32;
#ifndef P3D_WGRAD_POLITE_ROWS
#define P3D_WGRAD_POLITE_ROWS 8192
#endif
#ifndef P3D_WGRAD64_LDS_KB
#define P3D_WGRAD64_LDS_KB 82      // one 64x64 block per CU; see launch_group_t
#endif
template <int BM>
struct WRing { static constexpr int stages = (BM >= 128) ? 2 : 3; };   // 128x128: 64 KB -> two blocks per CU

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One problem of a group, as the kernel sees it (kernel-argument space: keep it compact).
struct WProb {
    const float* x; const float* dy; float* dw; float* dbias;
    int N, Di, Hi, Wi, ldx, K;
    int Gd, Gh, Gw, isd, ish, isw;
    P3dFastDiv fGd, fGh, fGw;
    int ldy, Nc;
    int ksplit;          // cuts of the position range
    int blk0;            // first block of this problem in the launch's 1-D grid
    int tile0;           // first (tile) slot of this problem in the scratch slabs / counters
    int ntaps;
    int pair;            // K <= 32: a 64-row tile holds TWO taps (rows 0-31 tap 2i, rows 32-63 tap 2i+1) -- the stem's K = 28
    // fused BatchNorm (WgradArgs): operand transforms and their per-channel coefficients
    int xt, dyt;
    int ldx2, ldy2;
    const float* x2; const float* xs1; const float* xt1; const float* xs2; const float* xt2;
    const float* dy2; const float* dcoef;
    signed char tap[P3D_MAX_TAPS][4];      // dd, dh, dw, weight slab
};
struct WGroup {
    int nprob;
    const float* zeros;
    float* slab; unsigned* cnt;
    int kstride;         // slab index of (slot, cut) = slot * kstride + cut
    WProb p[P3D_WGRAD_GROUP];
};
static_assert(sizeof(WGroup) <= 4000, "kernel arguments must stay under the 4 KB kernarg segment");

// Per-lane loader state in registers: the lattice coordinates of the rows this lane fetches, advanced
// by 32 positions per step with small-integer reciprocal carries (no per-step division).
template <int LA, int LB>
struct WState {
    unsigned m[LA];                 // position index of A row i (this step)
    int gw[LA], gh[LA], gd[LA], n[LA];
    int kc[LA];                     // channel offset of this lane's 16-byte chunk
    const float* bptr[LB];          // dY row pointer (+ chunk), advanced by 32 rows per step
    unsigned bm[LB];
    int bnc[LB];                    // dY column of this lane's chunk
    bool bok[LB];
    unsigned rGw, rGh, rGd;         // ceil(2^16 / extent)
};

template <int BM, int BN>
__device__ __forceinline__ void wloader_init(const WProb& p, WState<BM / 32, BN / 32>& st, unsigned ms, int k0, int n0,
                                             int wave, int lane) {
    constexpr int LA = BM / 32, LB = BN / 32;
    constexpr int A_LPR = BM / 4, A_RPP = 64 / A_LPR, B_LPR = BN / 4, B_RPP = 64 / B_LPR;
    st.rGw = 65536u / (unsigned)p.Gw + 1; st.rGh = 65536u / (unsigned)p.Gh + 1; st.rGd = 65536u / (unsigned)p.Gd + 1;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const unsigned m = ms + (i * 4 + wave) * A_RPP + lane / A_LPR;
        st.m[i] = m;
        const unsigned t1 = p3d_div(m, p.fGw), t2 = p3d_div(t1, p.fGh), nn = p3d_div(t2, p.fGd);
        st.gw[i] = (int)(m - t1 * (unsigned)p.Gw); st.gh[i] = (int)(t1 - t2 * (unsigned)p.Gh);
        st.gd[i] = (int)(t2 - nn * (unsigned)p.Gd); st.n[i] = (int)nn;
        st.kc[i] = p.pair ? ((lane % A_LPR) % (A_LPR / 2)) * 4 : k0 + (lane % A_LPR) * 4;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const unsigned m = ms + (i * 4 + wave) * B_RPP + lane / B_LPR;
        const int nc = n0 + (lane % B_LPR) * 4;
        st.bm[i] = m; st.bnc[i] = nc; st.bok[i] = nc < p.Nc;
        st.bptr[i] = p.dy + (long long)m * p.ldy + nc;
    }
}

// Always LA + LB loads (rows past the slice end, padded rows and channel tails read the zero page).
template <int BM, int BN>
__device__ __forceinline__ void advance_rows(const WProb& p, WState<BM / 32, BN / 32>& st, int i) {
    st.m[i] += BKM;
    int gw = st.gw[i] + BKM;
    const int q1 = (int)(((unsigned)gw * st.rGw) >> 16);
    gw -= q1 * p.Gw;
    int gh = st.gh[i] + q1;
    const int q2 = (int)(((unsigned)gh * st.rGh) >> 16);
    gh -= q2 * p.Gh;
    int gd = st.gd[i] + q2;
    const int q3 = (int)(((unsigned)gd * st.rGd) >> 16);
    gd -= q3 * p.Gd;
    st.gw[i] = gw; st.gh[i] = gh; st.gd[i] = gd; st.n[i] += q3;
}
template <int BM, int BN>
__device__ __forceinline__ void issue_stage(const WProb& p, const float* zeros, int tdd, int tdh, int tdw, float* __restrict__ a_dst,
                                            float* __restrict__ b_dst, WState<BM / 32, BN / 32>& st, unsigned me, int wave,
                                            int lane) {
    constexpr int LA = BM / 32, LB = BN / 32;
    const float* zp = zeros + 4 * (lane & 7);
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int id = st.gd[i] * p.isd + tdd, ih = st.gh[i] * p.ish + tdh, iw = st.gw[i] * p.isw + tdw;
        const bool ok = st.m[i] < me && st.kc[i] < p.K && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                        (unsigned)iw < (unsigned)p.Wi;
        const float* src = ok ? p.x + ((((long long)st.n[i] * p.Di + id) * p.Hi + ih) * p.Wi + iw) * p.ldx + st.kc[i] : zp;
        glds16(src, a_dst + (i * 4 + wave) * 256);
        advance_rows<BM, BN>(p, st, i);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const bool ok = st.bm[i] < me && st.bok[i];
        glds16(ok ? st.bptr[i] : zp, b_dst + (i * 4 + wave) * 256);
        st.bm[i] += BKM;
        st.bptr[i] += (long long)BKM * p.ldy;
    }
}

// ---- fused BatchNorm: both operands take a detour through raw LDS slots (conv_igemm2.hip, "fused BatchNorm") ----------
// The lane DMAs its chunks (of one or two sources per operand) into raw slots STAGES steps ahead; one step before use it
// reads ITS OWN chunks back, applies the per-channel transform -- its chunk's channels never change, so the coefficients
// sit in registers -- keeps padded / out-of-range chunks at zero, and writes the chunk to its place in a two-slot ring of
// finished tiles.  The fragment reads and MFMAs are the plain kernel's.
struct WMeta { unsigned aok, bok; };     // bit i: chunk i of the step is a real element run
template <int BM, int BN>
struct WRaw {       // raw DMA targets of one in-flight step and the ring slot its transformed tiles go to
    float* a; float* a2; float* b; float* b2;
};
template <int BM, int BN>
__device__ __forceinline__ void issue_raw(const WProb& p, const float* zeros, int tdd, int tdh, int tdw, const WRaw<BM, BN> d, WMeta& mt,
                                          WState<BM / 32, BN / 32>& st, unsigned me, bool two_x, bool two_dy, int wave, int lane) {
    constexpr int LA = BM / 32, LB = BN / 32;
    const float* zp = zeros + 4 * (lane & 7);
    mt.aok = 0; mt.bok = 0;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int id = st.gd[i] * p.isd + tdd, ih = st.gh[i] * p.ish + tdh, iw = st.gw[i] * p.isw + tdw;
        const bool ok = st.m[i] < me && st.kc[i] < p.K && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                        (unsigned)iw < (unsigned)p.Wi;
        mt.aok |= (ok ? 1u : 0u) << i;
        const long long row = (((long long)st.n[i] * p.Di + id) * p.Hi + ih) * p.Wi + iw;
        glds16(ok ? p.x + row * p.ldx + st.kc[i] : zp, d.a + (i * 4 + wave) * 256);
        if (two_x) glds16(ok ? p.x2 + row * p.ldx2 + st.kc[i] : zp, d.a2 + (i * 4 + wave) * 256);
        advance_rows<BM, BN>(p, st, i);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const bool ok = st.bm[i] < me && st.bok[i];
        mt.bok |= (ok ? 1u : 0u) << i;
        glds16(ok ? st.bptr[i] : zp, d.b + (i * 4 + wave) * 256);
        if (two_dy) glds16(ok ? p.dy2 + (long long)st.bm[i] * p.ldy2 + st.bnc[i] : zp, d.b2 + (i * 4 + wave) * 256);
        st.bm[i] += BKM;
        st.bptr[i] += (long long)BKM * p.ldy;
    }
}
struct WCoef {      // this lane's four channels (x) / columns (dy)
    float4 s1, t1, s2, t2;      // x:  relu(s1*x + t1) + relu(s2*x2 + t2)
    float4 k1, k2, k3;          // dy: k1*dy + k2*dy2 + k3
};
template <int BM, int BN>
struct WXIn { float4 au[BM / 32], av[BM / 32], bu[BN / 32], bv[BN / 32]; };
template <int BM, int BN>
__device__ __forceinline__ void raw_load(const WRaw<BM, BN> r, WXIn<BM, BN>& x, bool two_x, bool two_dy, int wave, int lane) {
    constexpr int LA = BM / 32, LB = BN / 32;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int at = (i * 4 + wave) * 256 + lane * 4;
        x.au[i] = *reinterpret_cast<const float4*>(r.a + at);
        x.av[i] = two_x ? *reinterpret_cast<const float4*>(r.a2 + at) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int at = (i * 4 + wave) * 256 + lane * 4;
        x.bu[i] = *reinterpret_cast<const float4*>(r.b + at);
        x.bv[i] = two_dy ? *reinterpret_cast<const float4*>(r.b2 + at) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__device__ __forceinline__ float4 wrelu4(float4 s, float4 u, float4 t) {
    return make_float4(fmaxf(fmaf(s.x, u.x, t.x), 0.f), fmaxf(fmaf(s.y, u.y, t.y), 0.f), fmaxf(fmaf(s.z, u.z, t.z), 0.f),
                       fmaxf(fmaf(s.w, u.w, t.w), 0.f));
}
template <int BM, int BN>
__device__ __forceinline__ void transform_store(const WXIn<BM, BN>& x, const WMeta& mt, const WCoef& cf, int xt, int dyt,
                                                float* __restrict__ a_dst, float* __restrict__ b_dst, int wave, int lane) {
    constexpr int LA = BM / 32, LB = BN / 32;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        float4 a = x.au[i];
        if (xt) {
            a = wrelu4(cf.s1, a, cf.t1);
            if (xt == 2) { const float4 q = wrelu4(cf.s2, x.av[i], cf.t2); a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w; }
        }
        if (!((mt.aok >> i) & 1u)) a = zero;
        *reinterpret_cast<float4*>(a_dst + (i * 4 + wave) * 256 + lane * 4) = a;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        float4 b = x.bu[i];
        if (dyt) {
            const float4 v = x.bv[i];
            b = make_float4(fmaf(cf.k1.x, b.x, fmaf(cf.k2.x, v.x, cf.k3.x)), fmaf(cf.k1.y, b.y, fmaf(cf.k2.y, v.y, cf.k3.y)),
                            fmaf(cf.k1.z, b.z, fmaf(cf.k2.z, v.z, cf.k3.z)), fmaf(cf.k1.w, b.w, fmaf(cf.k2.w, v.w, cf.k3.w)));
        }
        if (!((mt.bok >> i) & 1u)) b = zero;
        *reinterpret_cast<float4*>(b_dst + (i * 4 + wave) * 256 + lane * 4) = b;
    }
}

// All fragment reads of a stage are issued before its first MFMA (one exposed LDS latency per step instead of one per
// pair of MFMAs); the stage is then consumed in two halves so that the next refill's address arithmetic and DMA issue
// run while the first half's MFMAs execute (same arrangement as conv_igemm2.hip's pipe_step).
template <int BM, int BN>
struct WFrags { float a[BKM / 2][BM / 64]; float b[BKM / 2][BN / 64]; };

template <int BM, int BN>
__device__ __forceinline__ void load_wfrags(const float* __restrict__ a_st, const float* __restrict__ b_st, WFrags<BM, BN>& f,
                                            int wm, int wn, int h, int l31) {
    constexpr int TM = BM / 64, TN = BN / 64;
#pragma unroll
    for (int k2 = 0; k2 < BKM / 2; ++k2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[k2][i] = a_st[(2 * k2 + h) * BM + wm * (BM / 2) + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[k2][j] = b_st[(2 * k2 + h) * BN + wn * (BN / 2) + j * 32 + l31];
    }
}
// the bias gradient (column sums of dY) rides along in registers: rows past the end of the range are zero in LDS
template <int BM, int BN, int K0, int K1>
__device__ __forceinline__ void mfma_wfrags(const WFrags<BM, BN>& f, f32x16 (&acc)[BM / 64][BN / 64], float (&bsum)[BN / 64], bool do_bias) {
    constexpr int TM = BM / 64, TN = BN / 64;
#pragma unroll
    for (int k2 = K0 / 2; k2 < K1 / 2; ++k2) {
        if (do_bias) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bsum[j] += f.b[k2][j];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[k2][i], f.b[k2][j], acc[i][j], 0, 0, 0);
    }
}

template <int BM, int BN>
__device__ __forceinline__ void pipe_step(const WProb& p, const float* zeros, int tdd, int tdh, int tdw, float* __restrict__ a_dst,
                                          float* __restrict__ b_dst, const float* __restrict__ a_src,
                                          const float* __restrict__ b_src, f32x16 (&acc)[BM / 64][BN / 64], float (&bsum)[BN / 64],
                                          bool do_bias, WState<BM / 32, BN / 32>& st, unsigned me, int wave, int lane, int wm,
                                          int wn) {
    constexpr int LPS = BM / 32 + BN / 32;
    wait_vmcnt<(WRing<BM>::stages - 2) * LPS>();
    __builtin_amdgcn_s_barrier();
    WFrags<BM, BN> f;
    load_wfrags<BM, BN>(a_src, b_src, f, wm, wn, lane >> 5, lane & 31);
    __builtin_amdgcn_sched_barrier(0);      // keep every read above the MFMAs (hipcc otherwise sinks them back, pair by pair)
    mfma_wfrags<BM, BN, 0, BKM / 2>(f, acc, bsum, do_bias);
    issue_stage<BM, BN>(p, zeros, tdd, tdh, tdw, a_dst, b_dst, st, me, wave, lane);
    mfma_wfrags<BM, BN, BKM / 2, BKM>(f, acc, bsum, do_bias);
}
// fused: consume ring slot `cur`; transform the next step's raw chunks into ring slot `nxt`, then re-target that raw slot
template <int BM, int BN>
__device__ __forceinline__ void pipe_step_fused(const WProb& p, const float* zeros, int tdd, int tdh, int tdw, const WRaw<BM, BN> raw,
                                                WMeta& mt, const WCoef& cf, float* __restrict__ a_nxt, float* __restrict__ b_nxt,
                                                const float* __restrict__ a_cur, const float* __restrict__ b_cur,
                                                f32x16 (&acc)[BM / 64][BN / 64], float (&bsum)[BN / 64], bool do_bias,
                                                WState<BM / 32, BN / 32>& st, unsigned me, bool two_x, bool two_dy, int wave, int lane,
                                                int wm, int wn) {
    constexpr int LA = BM / 32, LB = BN / 32;
    if constexpr (WRing<BM>::stages == 2) {
        wait_vmcnt<0>();
    } else {      // one step's loads may still fly: LA + LB, + LA / + LB for second sources (block-uniform)
        if (two_x) { if (two_dy) wait_vmcnt<2 * LA + 2 * LB>(); else wait_vmcnt<2 * LA + LB>(); }
        else { if (two_dy) wait_vmcnt<LA + 2 * LB>(); else wait_vmcnt<LA + LB>(); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my transformed chunks of this step are in LDS
    __builtin_amdgcn_s_barrier();
    WFrags<BM, BN> f;
    load_wfrags<BM, BN>(a_cur, b_cur, f, wm, wn, lane >> 5, lane & 31);
    WXIn<BM, BN> xin;
    raw_load<BM, BN>(raw, xin, two_x, two_dy, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    mfma_wfrags<BM, BN, 0, BKM / 2>(f, acc, bsum, do_bias);
    transform_store<BM, BN>(xin, mt, cf, p.xt, p.dyt, a_nxt, b_nxt, wave, lane);
    issue_raw<BM, BN>(p, zeros, tdd, tdh, tdw, raw, mt, st, me, two_x, two_dy, wave, lane);
    mfma_wfrags<BM, BN, BKM / 2, BKM>(f, acc, bsum, do_bias);
}

template <int BM, int BN, bool FUSED>
constexpr size_t wsmem_bytes() {
    constexpr int S = WRing<BM>::stages;
    // plain: S stages of (A, B).  fused: two finished (A, B) tiles + S - 1 raw slots of (A, A2, B, B2)
    const size_t ring = FUSED ? (size_t)(2 + 2 * (S - 1)) * BKM * (BM + BN) * 4 : (size_t)S * BKM * (BM + BN) * 4;
    const size_t tile = (size_t)BM * (BN + 4) * 4 + BN * 4 + 16;      // staged tile + bias sums + reducer flag
    return ring > tile ? ring : tile;
}

template <int BM, int BN, bool FUSED>
__global__ __launch_bounds__(256) void wgrad2_kernel(const WGroup g) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_STAGE = BKM * BM, B_STAGE = BKM * BN;
    constexpr int STAGES = WRing<BM>::stages;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + (FUSED ? 2 : STAGES) * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    // which problem of the group does this block belong to (block ranges are ascending)
    int pi = 0;
#pragma unroll
    for (int q = 1; q < P3D_WGRAD_GROUP; ++q)
        if (q < g.nprob && (int)blockIdx.x >= g.p[q].blk0) pi = q;
    // copy the problem out of the kernel-argument segment once: a reference indexed by `pi` makes hipcc re-load its
    // fields (s_load + wait) inside the pipeline loop
    WProb p;
    {
        const WProb& src = g.p[pi];
        p.x = src.x; p.dy = src.dy; p.dw = src.dw; p.dbias = src.dbias;
        p.N = src.N; p.Di = src.Di; p.Hi = src.Hi; p.Wi = src.Wi; p.ldx = src.ldx; p.K = src.K;
        p.Gd = src.Gd; p.Gh = src.Gh; p.Gw = src.Gw; p.isd = src.isd; p.ish = src.ish; p.isw = src.isw;
        p.fGd = src.fGd; p.fGh = src.fGh; p.fGw = src.fGw;
        p.ldy = src.ldy; p.Nc = src.Nc; p.ksplit = src.ksplit; p.blk0 = src.blk0; p.tile0 = src.tile0; p.ntaps = src.ntaps;
        p.pair = src.pair;
        p.xt = FUSED ? src.xt : 0; p.dyt = FUSED ? src.dyt : 0; p.ldx2 = src.ldx2; p.ldy2 = src.ldy2;
        p.x2 = src.x2; p.xs1 = src.xs1; p.xt1 = src.xt1; p.xs2 = src.xs2; p.xt2 = src.xt2; p.dy2 = src.dy2; p.dcoef = src.dcoef;
    }
    const bool two_x = FUSED && p.xt == 2, two_dy = FUSED && p.dyt != 0;

    const long long M = (long long)p.N * p.Gd * p.Gh * p.Gw;
    const int KT = p.pair ? 1 : (p.K + BM - 1) / BM, NT = (p.Nc + BN - 1) / BN;
    // Block order (speed only): the blocks of one XCD (block index mod 8) take CONSECUTIVE (cut, tile) pairs, cut-major, so
    // the taps and channel tiles that walk the same position range -- the same x and dy rows -- share that XCD's L2.
    const int local = (int)blockIdx.x - p.blk0;
    const int ntiles = (p.pair ? (p.ntaps + 1) / 2 : p.ntaps) * KT * NT;
    int v = local;
    {
        const int T = ntiles * p.ksplit;
        if (T >= 64) { const int x = local & 7, q = T >> 3, r = T & 7; v = x * q + min(x, r) + (local >> 3); }
    }
    const int cut = v / ntiles;
    const int tile_local = v - cut * ntiles;
    int b = tile_local;
    const int nt = b % NT; b /= NT;
    const int kt = b % KT;
    const int ti = b / KT;
    // pair mode: the lanes that fetch the upper half of a row (channels 32-63 of the tile) belong to the second tap of the
    // pair; a missing second tap (odd tap count) is sent out of range so that those lanes read the zero page
    const int tA = p.pair ? 2 * ti : ti;
    const bool upper = p.pair && (lane % (BM / 4)) >= BM / 8;
    const int tL = upper ? tA + 1 : tA;                      // this LANE's tap
    const bool tap_ok = tL < p.ntaps;
    const int tLc = tap_ok ? tL : tA;
    const int tdd = tap_ok ? (int)g.p[pi].tap[tLc][0] : 30000, tdh = g.p[pi].tap[tLc][1], tdw = g.p[pi].tap[tLc][2];
    const int widx = g.p[pi].tap[tA][3];
    const int widx2 = (p.pair && tA + 1 < p.ntaps) ? (int)g.p[pi].tap[tA + 1][3] : -1;
    const int k0 = kt * BM, n0 = nt * BN;

    long long chunk = (M + p.ksplit - 1) / p.ksplit;
    chunk = (chunk + BKM - 1) / BKM * BKM;
    const long long ms = (long long)cut * chunk;
    const long long me = (ms + chunk < M) ? ms + chunk : M;
    const int nsteps = me > ms ? (int)((me - ms + BKM - 1) / BKM) : 0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const bool do_bias = p.dbias != nullptr && ti == 0 && kt == 0;
    float bsum[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bsum[j] = 0.f;

    WState<BM / 32, BN / 32> st;
    wloader_init<BM, BN>(p, st, (unsigned)ms, k0, n0, wave, lane);
    const unsigned meu = (unsigned)me;
    if constexpr (!FUSED) {
        if (STAGES == 3) {
            float* A0 = As; float* A1 = As + A_STAGE; float* A2 = As + 2 * A_STAGE;
            float* B0 = Bs; float* B1 = Bs + B_STAGE; float* B2 = Bs + 2 * B_STAGE;
            issue_stage<BM, BN>(p, g.zeros, tdd, tdh, tdw, A0, B0, st, meu, wave, lane);
            issue_stage<BM, BN>(p, g.zeros, tdd, tdh, tdw, A1, B1, st, meu, wave, lane);
            for (int base = 0; base < nsteps; base += 3) {
                pipe_step<BM, BN>(p, g.zeros, tdd, tdh, tdw, A2, B2, A0, B0, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
                if (base + 1 < nsteps) pipe_step<BM, BN>(p, g.zeros, tdd, tdh, tdw, A0, B0, A1, B1, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
                if (base + 2 < nsteps) pipe_step<BM, BN>(p, g.zeros, tdd, tdh, tdw, A1, B1, A2, B2, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
            }
        } else {
            float* A0 = As; float* A1 = As + A_STAGE;
            float* B0 = Bs; float* B1 = Bs + B_STAGE;
            issue_stage<BM, BN>(p, g.zeros, tdd, tdh, tdw, A0, B0, st, meu, wave, lane);
            for (int base = 0; base < nsteps; base += 2) {
                pipe_step<BM, BN>(p, g.zeros, tdd, tdh, tdw, A1, B1, A0, B0, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
                if (base + 1 < nsteps) pipe_step<BM, BN>(p, g.zeros, tdd, tdh, tdw, A0, B0, A1, B1, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
            }
        }
    } else {
        // raw slots behind the two finished tiles: [S-1] x (A, A2, B, B2)
        float* rawbase = Bs + 2 * B_STAGE;
        constexpr int RS = STAGES - 1, RAW_STEP = 2 * (A_STAGE + B_STAGE);
        WRaw<BM, BN> raw[RS];
#pragma unroll
        for (int q = 0; q < RS; ++q) {
            float* r = rawbase + q * RAW_STEP;
            raw[q].a = r; raw[q].a2 = r + A_STAGE; raw[q].b = r + 2 * A_STAGE; raw[q].b2 = r + 2 * A_STAGE + B_STAGE;
        }
        // this lane's coefficients: its chunk covers channels kc .. kc+3 of x and columns nc .. nc+3 of dy, for the whole kernel
        WCoef cf;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        cf.s1 = zero4; cf.t1 = zero4; cf.s2 = zero4; cf.t2 = zero4; cf.k1 = zero4; cf.k2 = zero4; cf.k3 = zero4;
        {
            const int kc = k0 + (lane % (BM / 4)) * 4, nc = n0 + (lane % (BN / 4)) * 4;
            if (p.xt && kc < p.K) {
                cf.s1 = *reinterpret_cast<const float4*>(p.xs1 + kc); cf.t1 = *reinterpret_cast<const float4*>(p.xt1 + kc);
                if (p.xt == 2) { cf.s2 = *reinterpret_cast<const float4*>(p.xs2 + kc); cf.t2 = *reinterpret_cast<const float4*>(p.xt2 + kc); }
            }
            if (p.dyt && nc < p.Nc) {
                cf.k1 = *reinterpret_cast<const float4*>(p.dcoef + nc); cf.k2 = *reinterpret_cast<const float4*>(p.dcoef + p.Nc + nc);
                cf.k3 = *reinterpret_cast<const float4*>(p.dcoef + 2 * p.Nc + nc);
            }
        }
        WMeta meta[RS];
        issue_raw<BM, BN>(p, g.zeros, tdd, tdh, tdw, raw[0], meta[0], st, meu, two_x, two_dy, wave, lane);
        if constexpr (STAGES == 3) issue_raw<BM, BN>(p, g.zeros, tdd, tdh, tdw, raw[1], meta[1], st, meu, two_x, two_dy, wave, lane);
        wait_vmcnt<0>();                    // own chunks only: no barrier needed before reading them back
        {
            WXIn<BM, BN> xin;
            raw_load<BM, BN>(raw[0], xin, two_x, two_dy, wave, lane);
            transform_store<BM, BN>(xin, meta[0], cf, p.xt, p.dyt, As, Bs, wave, lane);
        }
        issue_raw<BM, BN>(p, g.zeros, tdd, tdh, tdw, raw[0], meta[0], st, meu, two_x, two_dy, wave, lane);
        // step s consumes ring slot s % 2, transforms raw slot (s + 1) % (S - 1) into ring slot (s + 1) % 2 and re-targets it
        for (int base = 0; base < nsteps; base += 2) {
            pipe_step_fused<BM, BN>(p, g.zeros, tdd, tdh, tdw, raw[1 % RS], meta[1 % RS], cf, As + A_STAGE, Bs + B_STAGE, As, Bs, acc, bsum,
                                    do_bias, st, meu, two_x, two_dy, wave, lane, wm, wn);
            if (base + 1 < nsteps)
                pipe_step_fused<BM, BN>(p, g.zeros, tdd, tdh, tdw, raw[0], meta[0], cf, As, Bs, As + A_STAGE, Bs + B_STAGE, acc, bsum,
                                        do_bias, st, meu, two_x, two_dy, wave, lane, wm, wn);
        }
    }

    // ---- epilogue: stage the tile through LDS (row-wise float4 global traffic) -----------------------------------
    constexpr int LDT = BN + 4, F4R = BN / 4;
    float* tile = As;
    float* bias_lds = tile + BM * LDT;                           // [BN]
    int* flag = reinterpret_cast<int*>(bias_lds + BN);
    wait_vmcnt<0>();
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                tile[r * LDT + wn * (BN / 2) + j * 32 + l31] = acc[i][j][e];
            }
    if (do_bias && wm == 0) {      // column sums of dY: the two position parities (lane halves) of the waves that hold these columns
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float t = bsum[j] + __shfl_xor(bsum[j], 32);
            if (h == 0) bias_lds[wn * (BN / 2) + j * 32 + l31] = t;
        }
    }
    __syncthreads();

    if (p.ksplit > 1) {
        constexpr int SLAB = BM * BN + BN;                       // tile + bias partial
        const int slot = p.tile0 + tile_local;
        // write-through (sc1) slab stores: no release fence needed (conv_igemm2.hip; Guideline 16 recipe R1)
        float* myslab = g.slab + ((size_t)slot * g.kstride + cut) * SLAB;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(myslab, 0, SLAB * 4, 0x00020000);
#pragma unroll 4
        for (int i = tid; i < BM * F4R; i += 256) {
            const int r = i / F4R, c4 = (i - r * F4R) * 4;
            const float4 v = *reinterpret_cast<const float4*>(tile + r * LDT + c4);
            const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(u, rs, (r * BN + c4) * 4, 0, 16);
        }
        if (tid < BN) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(bias_lds[tid]), rs, (BM * BN + tid) * 4, 0, 16);
        wait_vmcnt<0>();
        __syncthreads();
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(g.cnt + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == (unsigned)(p.ksplit - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                g.cnt[slot] = 0;
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        const float* slabs = g.slab + (size_t)slot * g.kstride * SLAB;
        // The slabs come from other CUs' write-through stores: every load is a long-latency miss.  Sixteen in flight per lane
        // (four tile positions x four cuts), added in cut order -- with up to 256 cuts (the stem's filter gradient) the fold
        // is a large part of the launch.
        constexpr int PER_LANE = BM * F4R / 256;
        static_assert(PER_LANE % 4 == 0, "the fold takes four tile positions at a time");
#pragma unroll 1
        for (int i0 = 0; i0 < PER_LANE; i0 += 4) {
            float4 v[4];
            const float* src[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + (i0 + q) * 256;
                const int r = i / F4R, c4 = (i - r * F4R) * 4;
                src[q] = slabs + r * BN + c4;
                v[q] = *reinterpret_cast<const float4*>(src[q]);
            }
            int s = 1;
            for (; s + 3 < p.ksplit; s += 4) {
                float4 t4[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) t4[t][q] = *reinterpret_cast<const float4*>(src[q] + (size_t)(s + t) * SLAB);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q].x += t4[t][q].x; v[q].y += t4[t][q].y; v[q].z += t4[t][q].z; v[q].w += t4[t][q].w; }
            }
            for (; s < p.ksplit; ++s) {
                float4 t4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) t4[q] = *reinterpret_cast<const float4*>(src[q] + (size_t)s * SLAB);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q].x += t4[q].x; v[q].y += t4[q].y; v[q].z += t4[q].z; v[q].w += t4[q].w; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + (i0 + q) * 256;
                const int r = i / F4R, c4 = (i - r * F4R) * 4;
                *reinterpret_cast<float4*>(tile + r * LDT + c4) = v[q];
            }
        }
        if (tid < BN) {
            float t = 0.f;
            int s = 0;
            for (; s + 7 < p.ksplit; s += 8) {                  // cut order, eight loads in flight
                float u[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) u[q] = slabs[(size_t)(s + q) * SLAB + BM * BN + tid];
#pragma unroll
                for (int q = 0; q < 8; ++q) t += u[q];
            }
            for (; s < p.ksplit; ++s) t += slabs[(size_t)s * SLAB + BM * BN + tid];
            bias_lds[tid] = t;
        }
        __syncthreads();
    }

    float* dwt = p.dw + (long long)widx * p.K * p.Nc;
#pragma unroll 4
    for (int i = tid; i < BM * F4R; i += 256) {
        const int r = i / F4R, c4 = (i - r * F4R) * 4;
        int row = k0 + r;
        const int col = n0 + c4;
        float* base = dwt;
        if (p.pair) {
            row = r & 31;
            if (r >= 32) { if (widx2 < 0) continue; base = p.dw + (long long)widx2 * p.K * p.Nc; }
        }
        if (row >= p.K || col >= p.Nc) continue;
        float* dst = base + (long long)row * p.Nc + col;
        float4 v = *reinterpret_cast<const float4*>(tile + r * LDT + c4);
        const float4 o = *reinterpret_cast<const float4*>(dst);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        *reinterpret_cast<float4*>(dst) = v;
    }
    if (do_bias && tid < BN && (n0 + tid) < p.Nc) p.dbias[n0 + tid] += bias_lds[tid];
}

// LDS request of a launch = residency limiter (see launch_group_t): bytes per block of a (tm x tn) tile.
size_t lds_need(int tm, int tn, bool fused) {
    if (fused) {
        if (tm == 128 && tn == 128) return wsmem_bytes<128, 128, true>();
        if (tm == 128) return wsmem_bytes<128, 64, true>();
        if (tn == 128) return wsmem_bytes<64, 128, true>();
        return wsmem_bytes<64, 64, true>();
    }
    if (tm == 128 && tn == 128) return wsmem_bytes<128, 128, false>();
    if (tm == 128) return wsmem_bytes<128, 64, false>();
    if (tn == 128) return wsmem_bytes<64, 128, false>();
    return wsmem_bytes<64, 64, false>();
}
// tuning knobs: compile-time defaults; a -DP3D_TUNING build (tools/*.sh via P3D_EXTRA_HIPCC_FLAGS) reads them from the environment once
struct WTune { long lds_kb = P3D_WGRAD64_LDS_KB; int slots = 0; bool no_rect = false; long long polite_rows = P3D_WGRAD_POLITE_ROWS; };
const WTune& wtune() {
    static const WTune t = [] {
        WTune w;
        if (const char* e = p3d_tune_env("P3D_WGRAD_LDS_KB")) { const long v = atol(e); if (v > 0 && v <= 160) w.lds_kb = v; }
        if (const char* e = p3d_tune_env("P3D_WGRAD_SLOTS")) w.slots = atoi(e);
        if (p3d_tune_env("P3D_WGRAD_NO_RECT")) w.no_rect = true;
        if (const char* e = p3d_tune_env("P3D_WGRAD_POLITE_ROWS")) w.polite_rows = atoll(e);
        return w;
    }();
    return t;
}
size_t lds_request(int tm, int tn, bool fused) {
    const size_t need = lds_need(tm, tn, fused);
    if (tm != 64 || tn != 64) return need;
    const size_t want = (size_t)wtune().lds_kb * 1024;
    return want < need ? need : want;
}

// test hook (p3d_wgrad2_force_tile): force the tile of single-problem launches where the problem allows it
int g_force_tm = 0, g_force_tn = 0;

struct WPlan { int tm, tn; long long tiles; int ks; double cost; };
long long tiles_of(const WgradArgs& a, int tm, int tn) {
    if (a.pair) return (long long)((a.ntaps + 1) / 2) * ((a.Nc + tn - 1) / tn);
    return (long long)a.ntaps * ((a.K + tm - 1) / tm) * ((a.Nc + tn - 1) / tn);
}
// Pick the tile and the number of position-range cuts with a small cost model: blocks run in rounds of
// `slots` (256 CUs x resident blocks per CU), a round lasts (steps per block + fixed overhead) step-times; a 128x128 step
// is ~3.2x a 64x64 step (4x the MFMAs, better LDS-DMA efficiency), a 64x128 / 128x64 step ~1.75x.  This avoids e.g. 540
// blocks on 512 slots (a second round with 28 blocks).  `other_tiles`: 64x64 tiles of the other problems in the same launch
// (groups use 64x64 only).
WPlan plan(const WgradArgs& a, long long other_tiles = 0) {
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const long long steps = (M + BKM - 1) / BKM;
    auto best_for = [&](int TM_, int TN_) {
        WPlan w;
        w.tm = TM_; w.tn = TN_; w.ks = 1; w.cost = 1e300;
        w.tiles = tiles_of(a, TM_, TN_);
        // the model keeps 3 (2) slots per CU whatever the residency limit: with one resident block per CU the launch then
        // runs in ~3 short rounds, and short blocks are what lets main-stream blocks in (17.47 vs 17.78 ms / step)
        const int per_cu = wtune().slots;
        const int area = TM_ * TN_;
        const long long slots = 256 * (per_cu > 0 ? per_cu : (area > 4096 ? 2 : 3));
        const double step_time = area == 16384 ? 3.2 : (area == 8192 ? 1.75 : 1.0), overhead = 8.0;
        // one tile or two over a long position range (a 1x1x1 conv to 32 channels at 1.6 M positions): many short cuts
        const long long kcap = w.tiles + other_tiles <= 8 ? 256 : 64;
        const long long kmax = std::max<long long>(1, std::min<long long>(steps / 4, kcap));
        for (long long ks = 1; ks <= kmax; ks = ks < 16 ? ks + 1 : ks + ks / 8) {
            const long long blocks = (w.tiles + other_tiles) * ks;
            const long long rounds = (blocks + slots - 1) / slots;
            // a cut costs its slab store, and the last arriver reads ks slabs
            const double per_block = (double)((steps + ks - 1) / ks) + overhead + (ks > 1 ? 2.0 + 0.5 * ks : 0.0);
            const double cost = rounds * per_block * step_time;
            if (cost < w.cost * 0.98) { w.cost = cost; w.ks = (int)ks; }
        }
        return w;
    };
    // A launch over many positions (the decoder's deconvs, stage 1) keeps every CU busy for hundreds of steps; there a
    // 64x128 tile does more per LDS-DMA byte: deconv3's filter gradient at 32x224x224 alone 7.76 -> 6.85 ms (128x128: 6.88),
    // the step 89.7 -> 87.6 ms (128x128: 87.0, but 17.34 vs 17.22 ms on the 16x112x112 step; 128x64: worse on both).  The
    // cut count still comes from the model above: a cost model of its own that counted work per CU picked fewer, longer
    // blocks and lost 0.2-3.6 ms per step on every workload although each filter gradient alone was faster.
    if (a.pair) return best_for(64, 64);
    const bool forced = g_force_tm != 0;
    // (from 2048 positions: deconv2's filter gradient at 8 clips of 16x112x112, 6272 positions x 18 taps, 493 -> 359 us; deconv1's
    // at 32x224x224, 3136 positions, 473 -> 340 us)
    // Round 3, per-launch sweeps (tools/op_times.py --wgrad-tile): launches with 256 or more 64x64 tiles take part whatever their
    // position count (deconv1 at 8 clips of 16x112x112, 392 positions: 89 -> 69 us).  128x128 tiles win ALONE wherever there are
    // 128 or more of them (the GN head's 1792 -> 1024 conv 23.2 -> 21.9 ms, the unet++ decoder convs 6-11 %) and change
    // nothing in the step (same-box A/B over five workloads, tools/archive/ab_wgrad_big.sh: within 0.2 % either way): 64x128 stays.
    const bool busy = other_tiles == 0 && (M >= 2048 || tiles_of(a, 64, 64) >= 256) && a.Nc >= 128 && !wtune().no_rect && !forced;
    static const bool big_tiles = [] { const char* e = p3d_tune_env("P3D_TUNE_WGRAD_BIG"); return e && atoi(e); }();      // A/B (tools/archive/ab_wgrad_big.sh)
    if (busy && big_tiles && a.K >= 128 && tiles_of(a, 128, 128) >= 128) return best_for(128, 128);
    if (busy) return best_for(64, 128);
    WPlan best = best_for(64, 64);
    if (other_tiles == 0) {
        if (forced) {
            if ((g_force_tm == 64 || a.K >= 128) && (g_force_tn == 64 || a.Nc >= 128)) return best_for(g_force_tm, g_force_tn);
            return best;
        }
        if (a.K >= 128 && a.Nc >= 128) { const WPlan w = best_for(128, 128); if (w.cost <= best.cost) best = w; }
    }
    return best;
}

bool wgrad_ok(const WgradArgs& a) {
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (M >= (1ll << 31) - 4096 || a.Gw > 400 || a.Gh > 400 || a.Gd > 400) return false;   // reciprocal carries
    if (a.ntaps > P3D_MAX_TAPS || !a.zeros) return false;
    if ((a.K & 3) || (a.ldx & 3) || (a.Nc & 3) || (a.ldy & 3)) return false;
    if (a.pair && a.K > 32) return false;
    if (a.xt < 0 || a.xt > 2 || (a.xt && (a.pair || !a.xs1 || !a.xt1))) return false;
    if (a.xt == 2 && (!a.x2 || !a.xs2 || !a.xt2 || (a.ldx2 & 3))) return false;
    if (a.dyt && (!a.dy2 || !a.dcoef || (a.ldy2 & 3) || a.pair)) return false;
    for (int t = 0; t < a.ntaps; ++t)
        if (a.taps[t].dd < -128 || a.taps[t].dd > 127 || a.taps[t].dh < -128 || a.taps[t].dh > 127 || a.taps[t].dw < -128 ||
            a.taps[t].dw > 127 || a.taps[t].widx < 0 || a.taps[t].widx > 127)
            return false;
    return true;
}

void fill_prob(WProb& p, const WgradArgs& a) {
    p.x = a.x; p.dy = a.dy; p.dw = a.dw; p.dbias = a.dbias;
    p.N = a.N; p.Di = a.Di; p.Hi = a.Hi; p.Wi = a.Wi; p.ldx = a.ldx; p.K = a.K;
    p.Gd = a.Gd; p.Gh = a.Gh; p.Gw = a.Gw; p.isd = a.isd; p.ish = a.ish; p.isw = a.isw;
    p.fGd = p3d_fastdiv((unsigned)a.Gd); p.fGh = p3d_fastdiv((unsigned)a.Gh); p.fGw = p3d_fastdiv((unsigned)a.Gw);
    p.ldy = a.ldy; p.Nc = a.Nc; p.ntaps = a.ntaps; p.pair = a.pair;
    p.xt = a.xt; p.dyt = a.dyt ? 1 : 0; p.ldx2 = a.ldx2; p.ldy2 = a.ldy2;
    p.x2 = a.x2; p.xs1 = a.xs1; p.xt1 = a.xt1; p.xs2 = a.xs2; p.xt2 = a.xt2; p.dy2 = a.dy2; p.dcoef = a.dcoef;
    for (int t = 0; t < a.ntaps; ++t) {
        p.tap[t][0] = (signed char)a.taps[t].dd; p.tap[t][1] = (signed char)a.taps[t].dh;
        p.tap[t][2] = (signed char)a.taps[t].dw; p.tap[t][3] = (signed char)a.taps[t].widx;
    }
}

__global__ void wgrad_nop_kernel(int) {}

template <int BM, int BN, bool FUSED>
hipError_t launch_group_t(WGroup& g, long long blocks, long long slabs, int slots, bool greedy, bool polite, hipStream_t s) {
    // timing diagnostic (WRONG RESULTS, tuning build): P3D_TUNE_WGRAD_EMPTY=1 launches an empty kernel in place of every filter gradient --
    // what the side stream's LAUNCHES (fork events, dispatches, end-of-kernel fences) cost the main stream without their work;
    // =2: a grid of the real size whose blocks return at once
    static const int empty = [] { const char* e = p3d_tune_env("P3D_TUNE_WGRAD_EMPTY"); return e ? atoi(e) : 0; }();
    if (empty) {
        hipLaunchKernelGGL(wgrad_nop_kernel, dim3(empty == 2 ? (unsigned)blocks : 1u), dim3(256), 0, s, 0);
        return hipGetLastError();
    }
    // LDS request = residency limiter.  The filter gradients share the chip with the main stream's chain of small launches,
    // whose blocks (igemm2 64x64: 48.5 KB of LDS) must find room on every CU while a filter-gradient launch is resident:
    // three 48 KB blocks per CU leave 16 KB, and every main-stream launch then waits for a filter-gradient block to
    // retire.  82 KB per 64x64 block = one block per CU: 17.97 -> 17.78 ms / step on one box, 17.93 -> 17.47 on another
    // (55 KB, two per CU: 17.68).  The plan below still cuts for three slots per CU -- short blocks retire often.
    static const size_t sm = [] {
        const size_t want = lds_request(BM, BN, FUSED);
        hipFuncSetAttribute((const void*)wgrad2_kernel<BM, BN, FUSED>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)std::max(want, lds_request(64, 64, FUSED)));
        return want;
    }();
    g.slab = nullptr; g.cnt = nullptr;
    if (slabs > 0) {
        const hipError_t e = p3d_stream_scratch(s, (size_t)slabs * (BM * BN + BN), (size_t)slots, &g.slab, &g.cnt);
        if (e != hipSuccess) return e;
    }
    // greedy: nothing else runs beside this launch (the stem's, last of the backward pass) -- full residency
    // not greedy: a 64x64 tile takes its polite request (one block per CU); a larger tile at least as much when the launch
    // is marked polite (it was parked to run beside the encoder's chain)
    size_t lds = greedy ? wsmem_bytes<BM, BN, FUSED>() : sm;
    if (polite && lds < lds_request(64, 64, FUSED)) lds = lds_request(64, 64, FUSED);
    hipLaunchKernelGGL((wgrad2_kernel<BM, BN, FUSED>), dim3((unsigned)blocks), dim3(256), lds, s, g);
    return hipGetLastError();
}

// A group of 3x3x3 / 1x3x3 convs over many positions with 128-wide outputs (the unet++ head's nodes) takes the 64x128 tile like such
// a launch alone: p3d_unetplusplus_nonsa 56.3 -> 54.3 ms / step (tools/ab/wgrad_group_rect_ab.sh, same box; the other six
// workloads within 0.2 %).  Groups that hold a bottleneck's 1x1x1 convs stay on 64x64: with those included the 32x224x224 step
// lost 0.8 %.  P3D_TUNE_WGRAD_GROUP_RECT (tuning build): the position count from which, 0 = never.
bool group_takes_rect(const std::vector<const WgradArgs*>& live) {
    static const long long rect_rows = [] { const char* e = p3d_tune_env("P3D_TUNE_WGRAD_GROUP_RECT"); return e ? atoll(e) : 2048ll; }();
    bool all = live.size() > 1 && rect_rows > 0 && !wtune().no_rect && g_force_tm == 0;
    for (auto* a : live) all = all && (long long)a->N * a->Gd * a->Gh * a->Gw >= rect_rows && a->Nc % 128 == 0 && !a->pair && a->ntaps >= 9;
    return all;
}

}  // namespace

const char* p3d_wgrad2_variant(const WgradArgs& a) {
    const WPlan w = plan(a);
    return w.tm == 128 ? (w.tn == 128 ? "wgrad2_kernel<128,128>" : "wgrad2_kernel<128,64>")
                       : (w.tn == 128 ? "wgrad2_kernel<64,128>" : "wgrad2_kernel<64,64>");
}
const char* p3d_wgrad2_group_variant(const WgradArgs* probs, int n, bool fused) {
    if (n == 1) return p3d_wgrad2_variant(probs[0]);
    std::vector<const WgradArgs*> live;
    for (int i = 0; i < n; ++i)
        if ((long long)probs[i].N * probs[i].Gd * probs[i].Gh * probs[i].Gw > 0 && probs[i].ntaps > 0) live.push_back(&probs[i]);
    if (group_takes_rect(live)) return fused ? "wgrad2_kernel<64,128,fused>(grouped)" : "wgrad2_kernel<64,128>(grouped)";
    return fused ? "wgrad2_kernel<64,64,fused>(grouped)" : "wgrad2_kernel<64,64>(grouped)";
}
void p3d_wgrad2_force_tile(int tm, int tn) {
    const bool ok = (tm == 64 || tm == 128) && (tn == 64 || tn == 128);
    g_force_tm = ok ? tm : 0; g_force_tn = ok ? tn : 0;
}

// One launch for up to P3D_WGRAD_GROUP problems.  A single problem may take the 128x128 tile; groups use 64x64 (64x128: below).
hipError_t p3d_launch_wgrad2_group(const WgradArgs* probs, int n, hipStream_t s) {
    std::vector<const WgradArgs*> live;
    for (int i = 0; i < n; ++i) {
        const WgradArgs& a = probs[i];
        const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
        if (M <= 0 || a.ntaps <= 0) continue;
        if (!wgrad_ok(a)) return hipErrorInvalidValue;
        live.push_back(&a);
    }
    if (live.empty()) return hipSuccess;
    if ((int)live.size() > P3D_WGRAD_GROUP) return hipErrorInvalidValue;
    WGroup g;
    memset(&g, 0, sizeof(g));
    g.nprob = (int)live.size();
    g.zeros = live[0]->zeros;
    bool fused = false;
    for (auto* a : live) if (a->xt || a->dyt) fused = true;
    WPlan solo = live.size() == 1 ? plan(*live[0]) : WPlan{64, 64, 0, 1, 0.0};
    if (live.size() > 1 && group_takes_rect(live)) solo.tn = 128;
    // tuning build: P3D_TUNE_WGRAD_GROUP_TILE = 1 / 2 / 3 puts the groups over few positions (the bottlenecks of stages 2-3) on 64x128 /
    // 128x128 / 128x64 tiles (fewer, longer blocks that pull fewer operand bytes beside the main stream's chain)
    static const int grp_tile = [] { const char* e = p3d_tune_env("P3D_TUNE_WGRAD_GROUP_TILE"); return e ? atoi(e) : 0; }();
    if (grp_tile && live.size() > 1 && g_force_tm == 0) {
        bool ok = true;
        const int gtm = grp_tile >= 2 ? 128 : 64, gtn = grp_tile <= 2 ? 128 : 64;
        for (auto* a : live) ok = ok && !a->pair && a->K % gtm == 0 && a->Nc % gtn == 0 && (long long)a->N * a->Gd * a->Gh * a->Gw <= wtune().polite_rows;
        if (ok) { solo.tm = gtm; solo.tn = gtn; }
    }
    const int tm = solo.tm, tn = solo.tn;
    long long tiles64_all = 0;
    for (auto* a : live) tiles64_all += tiles_of(*a, 64, 64);
    // cuts per problem (the model of plan(): the launch's tiles on 512-768 slots, blocks of steps / cuts + overhead)
    int cuts[P3D_WGRAD_GROUP];
    long long steps_of[P3D_WGRAD_GROUP], shortest = 1ll << 60;
    for (size_t q = 0; q < live.size(); ++q) {
        const WgradArgs& a = *live[q];
        const WPlan w = live.size() == 1 ? plan(a) : plan(a, tiles64_all - tiles_of(a, 64, 64));
        cuts[q] = w.ks;
        steps_of[q] = ((long long)a.N * a.Gd * a.Gh * a.Gw + BKM - 1) / BKM;
        shortest = std::min(shortest, (steps_of[q] + cuts[q] - 1) / cuts[q]);
    }
    // A group may hold problems over different position counts (the queue packs across stage boundaries: five stage-2 problems and
    // stage 1's projection gave 728 blocks of 28 steps beside 28 blocks of 224 -- 253 us at 28 TFLOP/s where its neighbours run at
    // 75-80, round 5): a problem whose blocks would run more than twice the group's shortest is cut further, down to that length.
    static const bool no_balance = p3d_tune_env("P3D_TUNE_WGRAD_NO_BALANCE") != nullptr;      // A/B runs (tuning build)
    if (live.size() > 1 && !no_balance)
        for (size_t q = 0; q < live.size(); ++q) {
            const long long len = (steps_of[q] + cuts[q] - 1) / cuts[q];
            if (len <= 2 * shortest) continue;
            const long long cap = std::max<long long>(1, std::min<long long>(steps_of[q] / 4, 64));
            cuts[q] = (int)std::max<long long>(cuts[q], std::min(cap, (steps_of[q] + shortest - 1) / shortest));
        }
    long long blocks = 0;
    int tile0 = 0, kstride = 1;
    for (size_t q = 0; q < live.size(); ++q) {
        const WgradArgs& a = *live[q];
        WProb& p = g.p[q];
        fill_prob(p, a);
        const long long tiles = tiles_of(a, tm, tn);
        p.ksplit = cuts[q];
        p.blk0 = (int)blocks;
        p.tile0 = tile0;
        blocks += tiles * p.ksplit;
        tile0 += (int)tiles;
        kstride = std::max(kstride, p.ksplit);
    }
    g.kstride = kstride;       // slab of (slot, cut) = slot * kstride + cut: disjoint whatever each problem's cut count
    static const bool trace = p3d_tune_env("P3D_TUNE_WGRAD_TRACE") != nullptr;      // tuning build: what a grouped launch is made of
    if (trace) {
        fprintf(stderr, "[p3d] wgrad group %dx%d, %lld blocks:", tm, tn, blocks);
        for (size_t q = 0; q < live.size(); ++q)
            fprintf(stderr, "  [M %lld K %d N %d taps %d cuts %d]", (long long)live[q]->N * live[q]->Gd * live[q]->Gh * live[q]->Gw, live[q]->K, live[q]->Nc,
                    live[q]->ntaps, g.p[q].ksplit);
        fprintf(stderr, "\n");
    }
    if (blocks >= (1ll << 31)) return hipErrorInvalidValue;
    const long long slabs = kstride > 1 ? (long long)tile0 * kstride : 0;
    // Residency (launch_group_t): problems over few positions belong to the encoder's later stages, whose main-stream launches
    // are small and must find LDS on every CU -> one block per CU; problems over many positions run beside chip-filling
    // launches, where limiting residency only slows the filter gradient (unet++ head: 66.5 vs 60.1 ms / step) -> full.
    bool polite = false, force = false;
    long long max_m = 0;
    for (auto* a : live) {
        polite |= a->polite != 0; force |= a->greedy != 0;
        max_m = std::max(max_m, (long long)a->N * a->Gd * a->Gh * a->Gw);
    }
    const bool greedy = !polite && (force || max_m > wtune().polite_rows);
    if (fused) {
        if (tm == 128) return tn == 128 ? launch_group_t<128, 128, true>(g, blocks, slabs, tile0, greedy, polite, s)
                                        : launch_group_t<128, 64, true>(g, blocks, slabs, tile0, greedy, polite, s);
        return tn == 128 ? launch_group_t<64, 128, true>(g, blocks, slabs, tile0, greedy, polite, s)
                         : launch_group_t<64, 64, true>(g, blocks, slabs, tile0, greedy, polite, s);
    }
    if (tm == 128) return tn == 128 ? launch_group_t<128, 128, false>(g, blocks, slabs, tile0, greedy, polite, s)
                                    : launch_group_t<128, 64, false>(g, blocks, slabs, tile0, greedy, polite, s);
    return tn == 128 ? launch_group_t<64, 128, false>(g, blocks, slabs, tile0, greedy, polite, s)
                     : launch_group_t<64, 64, false>(g, blocks, slabs, tile0, greedy, polite, s);
}

hipError_t p3d_launch_wgrad2(const WgradArgs& a, hipStream_t s) { return p3d_launch_wgrad2_group(&a, 1, s); }
