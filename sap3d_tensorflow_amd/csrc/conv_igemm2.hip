// Pipelined implicit-GEMM convolution for gfx950 -- the workhorse of the P3D path (every conv /
// conv-input-gradient / conv3d_transpose, the Cin=3 stem on its packed 4-channel form included; reference
// p3d.py:19,24,86,112,125,172,200-216).  The geometry contract is IgemmArgs (p3d_kernels.h): a dense output lattice and
// an input lattice gathered through kernel taps.  Machinery:
//
//  * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction)
//    into a 3-stage ring; loads for step s+2 are in flight while step s is on the matrix cores;
//    ONE raw s_barrier per step with a counted s_waitcnt vmcnt (never 0 in the steady state);
//  * the LDS images are lane-linear (a glds constraint), so the bank-conflict swizzle is applied
//    on the per-lane SOURCE address and again on the ds_read_b128 (16-byte chunk q of row r is
//    stored at chunk q ^ ((r >> 1) & 7): conflict-free for the 32x32x2 A/B fragment reads);
//  * rows that fall into SAME padding, and channel tails, read from a zero page instead of
//    branching, so every wave issues the same number of loads per step (the vmcnt count);
//  * K-slicing: layers with few output tiles (M = B*98 positions in stage 3) cut the (tap, k-chunk) step
//    range into slices so that they still cover 256 CUs.  Every slice stores its partial tile to a scratch
//    slab and takes an arrival ticket; the block that draws the last ticket sums the slabs IN SLICE ORDER and
//    writes the output (bias, accumulate and BatchNorm statistics included), so results are bit-reproducible --
//    fp32 atomics were not (cdna_hip_programming.md, "Projection GEMM at M = 256", item 2);
//  * FUSED BatchNorm (round 3; p3d.py:56-81,88-97).  The bn -> relu between two convs of a bottleneck is applied to the
//    consumer conv's A operand on its way into LDS (template parameter AT, P3D_AT_*): the normalised tensor is never
//    written or read.  The block builds the per-channel coefficient table in LDS in its prologue -- from the producer's per-tile
//    statistics partials (fixed order, double accumulation) or from the published scale / shift; block 0 publishes
//    scale / shift / mean / invstd for the backward pass and updates the moving statistics.  The A operand of these
//    variants is staged through registers (plain global loads, transformed by the lane that fetched them, written to a
//    two-slot LDS ring one step ahead of use; "fused BatchNorm" below), so padded taps stay exactly zero and the MFMA
//    loop is the plain kernel's.  Input-gradient launches gate their result with the ReLU mask of the BatchNorm they
//    differentiate through and leave (sum g, sum g*xhat) per tile (BnGate), so BatchNorm's backward needs no pass of
//    its own either.
//
// fp32 in / fp32 accumulate: v_mfma_f32_32x32x2_f32, exact fp32 at the fp32 peak (157 TFLOP/s).
#if !defined(__gfx950__) && !defined(__gfx942__) && defined(__HIP_DEVICE_COMPILE__)
#error "the K-slice exchange (write-through slab stores + relaxed ticket + one-lane acquire) is written for gfx942 / gfx950 cache semantics"
#endif
#include "p3d_kernels.h"
#include "igemm_epilogue.h"
#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
#ifndef P3D_RING64
#define P3D_RING64 3
#endif
template <int BM, int BN>
struct Ring { static constexpr int stages = (BM >= 128) ? 2 : P3D_RING64; };   // 64x64: deep ring, few steps per K-slice

// operand transform traits
template <int AT>
struct ATr {
    static constexpr bool two = (AT == P3D_AT_RELU2 || AT == P3D_AT_GRAD);          // second A source
    static constexpr int ncoef = AT == P3D_AT_RELU1 ? 2 : AT == P3D_AT_RELU2 ? 4 : AT == P3D_AT_GRAD ? 3 : 0;
};

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
    // LDS destination = wave-uniform base + lane * 16 B
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Per-lane loader state, all in registers (an LDS read in the load path would make hipcc drain the
// in-flight LDS-DMA first).  Row pointers are rebuilt only when the kernel tap changes; within a
// tap a step just advances 32 floats along the channel run.  The A and the B operand have cursors of
// their own: the fused-BatchNorm path fetches A one step further ahead than B.
template <int LA, bool TWO>
struct ALoad {
    int base[LA];            // n * Di*Hi*Wi, or -1 for rows past M
    int dhw[LA];             // (g_d*is_d) << 20 | (g_h*is_h) << 10 | (g_w*is_w)
    const float* aptr[LA];   // row start for the current tap (+ this lane's 16-byte chunk), null when padded
    const float* aptr2[TWO ? LA : 1];   // same row of the second source
    int achunk[LA];          // 4 * logical chunk held by this lane's slot
    int t, kc;               // next step to issue: tap index, k-chunk index
    int issued;              // steps issued so far
};
template <int LB>
struct BLoad {
    int boff[LB];            // weight element offset of this lane's piece within a tap slab (k0 = 0)
    int bk[LB];              // WT: 4 * logical chunk; !WT: k row within the step
    bool bok[LB];            // n inside Nc
    int t, kc, issued;
    const float* w;          // weight tensor of this launch / class
};

template <int BM, bool TWO>
__device__ __forceinline__ void a_init(const IgemmArgs& p, const Geo& geo, ALoad<BM / 32, TWO>& st, unsigned m0, unsigned M, int wave, int lane,
                                       int s_begin, int kchunks) {
    constexpr int LA = BM / 32;
    const int a_slot = lane & 7, a_sub = lane >> 3;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int r = (i * 4 + wave) * 8 + a_sub;
        const unsigned m = m0 + r;
        st.base[i] = -1; st.dhw[i] = 0; st.aptr[i] = nullptr;
        if (TWO) st.aptr2[TWO ? i : 0] = nullptr;
        st.achunk[i] = 4 * (a_slot ^ ((r >> 1) & 7));
        if (m < M) {
            const unsigned t1 = p3d_div(m, geo.fGw), gw = m - t1 * (unsigned)geo.Gw;
            const unsigned t2 = p3d_div(t1, geo.fGh), gh = t1 - t2 * (unsigned)geo.Gh;
            const unsigned n = p3d_div(t2, geo.fGd), gd = t2 - n * (unsigned)geo.Gd;
            st.base[i] = (int)n * p.Di * p.Hi * p.Wi;
            st.dhw[i] = (int)(((gd * p.isd) << 20) | ((gh * p.ish) << 10) | (gw * p.isw));
        }
    }
    st.t = s_begin / kchunks; st.kc = s_begin - st.t * kchunks; st.issued = 0;
}
template <int BN, bool WT>
__device__ __forceinline__ void b_init(const IgemmArgs& p, const float* w, BLoad<BN / 32>& st, int n0, int wave, int lane, int s_begin, int kchunks) {
    constexpr int LB = BN / 32;
    st.w = w;
    const int a_slot = lane & 7, a_sub = lane >> 3;
    if (!WT) {
        constexpr int LANES_PER_ROW = BN / 4, ROWS_PER_PIECE = 64 / LANES_PER_ROW;
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int kr = (i * 4 + wave) * ROWS_PER_PIECE + lane / LANES_PER_ROW;
            const int nc = n0 + (lane % LANES_PER_ROW) * 4;
            st.bk[i] = kr; st.bok[i] = nc < p.Nc; st.boff[i] = kr * p.Nc + nc;
        }
    } else {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int r = (i * 4 + wave) * 8 + a_sub;
            st.bk[i] = 4 * (a_slot ^ ((r >> 1) & 7));
            st.bok[i] = (n0 + r) < p.Nc;
            st.boff[i] = (n0 + r) * p.K + st.bk[i];
        }
    }
    st.t = s_begin / kchunks; st.kc = s_begin - st.t * kchunks; st.issued = 0;
}

template <int LA, bool TWO>
__device__ __forceinline__ void a_set_tap(const IgemmArgs& p, P3dKTap* taps, ALoad<LA, TWO>& st) {
    const P3dTap tap = p3d_ktap(taps, st.t);
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int id = (st.dhw[i] >> 20) + tap.dd, ih = ((st.dhw[i] >> 10) & 1023) + tap.dh, iw = (st.dhw[i] & 1023) + tap.dw;
        const bool ok = st.base[i] >= 0 && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                        (unsigned)iw < (unsigned)p.Wi;
        const long long row = (long long)st.base[i] + ((long long)id * p.Hi + ih) * p.Wi + iw;
        st.aptr[i] = ok ? p.x + row * p.ldx + st.achunk[i] : nullptr;
        if (TWO) st.aptr2[TWO ? i : 0] = ok ? p.x2 + row * p.ldx2 + st.achunk[i] : nullptr;
    }
}

// Issue the LDS-DMA of the next (tap, k-chunk) step of one operand -- ALWAYS the same number of loads, so the counted
// vmcnt never changes; steps past the end of this block's slice fetch the zero page.  The destinations are
// __restrict__ so that, inlined next to the fragment reads, hipcc knows the reads cannot alias the DMA
// targets and does not put s_waitcnt vmcnt(0) in front of them.
template <int BM>
__device__ __forceinline__ void issue_a_dma(const IgemmArgs& p, P3dKTap* taps, float* __restrict__ a_dst, ALoad<BM / 32, false>& st, int nsteps,
                                            int kchunks, bool first, int wave, int lane) {
    constexpr int LA = BM / 32;
    const bool live = st.issued < nsteps;
    if (live && (first || st.kc == 0)) a_set_tap(p, taps, st);      // wave-uniform, once per tap
    const int k0 = st.kc * BK;
    const float* zp = p.zeros + 4 * (lane & 7);
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const bool ok = live && st.aptr[i] != nullptr && (k0 + st.achunk[i]) < p.K;
        glds16(ok ? st.aptr[i] + k0 : zp, a_dst + (i * 4 + wave) * 8 * BK);
    }
    ++st.issued;
    if (++st.kc == kchunks) { st.kc = 0; ++st.t; }
}
template <int BN, bool WT>
__device__ __forceinline__ void issue_b_dma(const IgemmArgs& p, P3dKTap* taps, float* __restrict__ b_dst, BLoad<BN / 32>& st, int nsteps, int kchunks,
                                            int wave, int lane) {
    // (the weight base travels in the loader state: per class in a grouped launch)
    constexpr int LB = BN / 32;
    const bool live = st.issued < nsteps;
    const int k0 = st.kc * BK;
    const float* zp = p.zeros + 4 * (lane & 7);
    const float* wt = st.w + (long long)taps[live ? st.t : 0].widx * p.K * p.Nc;
    if (!WT) {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const bool ok = live && st.bok[i] && (k0 + st.bk[i]) < p.K;
            glds16(ok ? wt + (long long)k0 * p.Nc + st.boff[i] : zp, b_dst + (i * 4 + wave) * 256);
        }
    } else {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const bool ok = live && st.bok[i] && (k0 + st.bk[i]) < p.K;
            glds16(ok ? wt + k0 + st.boff[i] : zp, b_dst + (i * 4 + wave) * 8 * BK);
        }
    }
    ++st.issued;
    if (++st.kc == kchunks) { st.kc = 0; ++st.t; }
}

// ---- fused BatchNorm: the A operand takes a detour through raw LDS slots ---------------------------------------------
// A lane DMAs the same 16-byte chunks as in the plain kernel (of one or two source tensors) into RAW slots, STAGES
// steps ahead; one step before a chunk is consumed the lane that fetched it reads it back (its own bytes: no barrier
// needed, only its own vmcnt), applies the per-channel transform -- that lane knows whether the chunk was a padded tap
// or a row past the end, those stay exactly zero -- and writes the result to the chunk's place in a two-slot ring of
// transformed tiles.  Fragment reads and MFMAs are those of the plain kernel, every A element is transformed once per
// block (not once per wave that reads it), and the transform's few LDS / VALU instructions sit between the two MFMA
// halves of a step.
struct AMeta {
    unsigned ok;             // bit i: chunk i is a real element run (else it stays zero)
    int k0;                  // channel base of the step (coefficient table offset)
};
template <int BM, int AT>
__device__ __forceinline__ void issue_a_raw(const IgemmArgs& p, P3dKTap* taps, float* __restrict__ raw_dst, float* __restrict__ raw2_dst, AMeta& r,
                                            ALoad<BM / 32, ATr<AT>::two>& st, int nsteps, int kchunks, bool first, int wave, int lane) {
    constexpr int LA = BM / 32;
    constexpr bool TWO = ATr<AT>::two;
    const bool live = st.issued < nsteps;
    if (live && (first || st.kc == 0)) a_set_tap(p, taps, st);
    const int k0 = st.kc * BK;
    const float* zp = p.zeros + 4 * (lane & 7);
    r.ok = 0; r.k0 = k0;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const bool ok = live && st.aptr[i] != nullptr && (k0 + st.achunk[i]) < p.K;
        r.ok |= (ok ? 1u : 0u) << i;
        // always one DMA per chunk and source (the counted vmcnt): padded chunks read the zero page
        glds16(ok ? st.aptr[i] + k0 : zp, raw_dst + (i * 4 + wave) * 8 * BK);
        if (TWO) glds16(ok ? st.aptr2[TWO ? i : 0] + k0 : zp, raw2_dst + (i * 4 + wave) * 8 * BK);
    }
    ++st.issued;
    if (++st.kc == kchunks) { st.kc = 0; ++st.t; }
}
__device__ __forceinline__ float4 relu_affine4(float4 s, float4 u, float4 t) {
    return make_float4(fmaxf(fmaf(s.x, u.x, t.x), 0.f), fmaxf(fmaf(s.y, u.y, t.y), 0.f), fmaxf(fmaf(s.z, u.z, t.z), 0.f),
                       fmaxf(fmaf(s.w, u.w, t.w), 0.f));
}
// The transform in two parts: the LDS reads (the lane's raw chunk(s) and its channels' coefficients) are issued with the
// step's fragment reads, the arithmetic and the store of the transformed chunk run between the MFMA halves, when the data
// has long arrived -- a wait there would idle the matrix pipe (one wave per SIMD: nothing else fills it).
template <int LA, int AT>
struct XIn {
    float4 u[LA];
    float4 v[ATr<AT>::two ? LA : 1];
    float4 c[LA][ATr<AT>::ncoef];
};
template <int BM, int AT>
__device__ __forceinline__ void transform_load(const AMeta& r, const float* __restrict__ raw_src, const float* __restrict__ raw2_src,
                                               const int (&achunk)[BM / 32], const float* __restrict__ tab, int kp,
                                               XIn<BM / 32, AT>& x, int wave, int lane) {
    constexpr int LA = BM / 32;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int at = (i * 4 + wave) * 8 * BK + lane * 4;      // this lane's chunk: where its DMA put it, where the tile wants it
        x.u[i] = *reinterpret_cast<const float4*>(raw_src + at);
        if (ATr<AT>::two) x.v[ATr<AT>::two ? i : 0] = *reinterpret_cast<const float4*>(raw2_src + at);
        const float* c = tab + r.k0 + achunk[i];
#pragma unroll
        for (int q = 0; q < ATr<AT>::ncoef; ++q) x.c[i][q] = *reinterpret_cast<const float4*>(c + q * kp);
    }
}
template <int BM, int AT>
__device__ __forceinline__ void transform_store(const AMeta& r, const XIn<BM / 32, AT>& x, float* __restrict__ a_dst, int wave, int lane) {
    constexpr int LA = BM / 32;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int at = (i * 4 + wave) * 8 * BK + lane * 4;
        const float4 u = x.u[i];
        float4 a;
        if constexpr (AT == P3D_AT_RELU1) {
            a = relu_affine4(x.c[i][0], u, x.c[i][1]);
        } else if constexpr (AT == P3D_AT_RELU2) {
            const float4 p1 = relu_affine4(x.c[i][0], u, x.c[i][1]);
            const float4 p2 = relu_affine4(x.c[i][2], x.v[i], x.c[i][3]);
            a = make_float4(p1.x + p2.x, p1.y + p2.y, p1.z + p2.z, p1.w + p2.w);
        } else {
            const float4 v = x.v[i], k1 = x.c[i][0], k2 = x.c[i][1], k3 = x.c[i][2];
            a = make_float4(fmaf(k1.x, u.x, fmaf(k2.x, v.x, k3.x)), fmaf(k1.y, u.y, fmaf(k2.y, v.y, k3.y)),
                            fmaf(k1.z, u.z, fmaf(k2.z, v.z, k3.z)), fmaf(k1.w, u.w, fmaf(k2.w, v.w, k3.w)));
        }
        if (!((r.ok >> i) & 1u)) a = make_float4(0.f, 0.f, 0.f, 0.f);      // SAME padding, rows past M, channel tails
        *reinterpret_cast<float4*>(a_dst + at) = a;
    }
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x4 to_half4(float4 v) { f16x4 r = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; return r; }

// Fragments of one whole stage (BK / 8 = 4 c-iterations): every ds_read of the step is issued BEFORE the first MFMA, so
// the LDS latency is paid once per step instead of once per c-iteration (each 32x32x2 MFMA chain on one accumulator is
// 4 x 64 cycles: with the reads interleaved the matrix pipe idled ~120 cycles in every 256).
template <int BM, int BN, bool WT>
struct Frags {
    float4 a[BK / 8][BM / 64];
    float4 b[BK / 8][BN / 64];
};
template <int BM, int BN, bool WT>
__device__ __forceinline__ void load_frags(const float* __restrict__ a_st, const float* __restrict__ b_st, Frags<BM, BN, WT>& f,
                                           int wm, int wn, int h, int l31) {
    constexpr int TM = BM / 64, TN = BN / 64;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * (BM / 2) + i * 32 + l31;
            const int slot = (2 * c + h) ^ ((r >> 1) & 7);
            f.a[c][i] = *reinterpret_cast<const float4*>(a_st + r * BK + slot * 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * (BN / 2) + j * 32 + l31;
            if (WT) {
                const int slot = (2 * c + h) ^ ((col >> 1) & 7);
                f.b[c][j] = *reinterpret_cast<const float4*>(b_st + col * BK + slot * 4);
            } else {
                const float* bp = b_st + (c * 8 + 4 * h) * BN + col;
                f.b[c][j] = make_float4(bp[0], bp[BN], bp[2 * BN], bp[3 * BN]);
            }
        }
    }
}

// F16: the pointwise-conv option of BASELINE configs[4] -- operands stay fp32 in HBM and LDS, the fragments are
// rounded to fp16 in registers and one v_mfma_f32_32x32x8_f16 (fp32 accumulate) replaces four fp32 MFMAs: a lane's
// float4 fragment holds k = 8c + 4h + {0..3}, which is exactly the A / B operand layout of that instruction.
// c-iterations [C0, C1) of one stage: the stage is consumed in two halves so that the address arithmetic and DMA issue
// of the next refill can run while the first half's MFMAs execute (pipe_step).
template <int BM, int BN, bool WT, bool F16, int C0, int C1>
__device__ __forceinline__ void mfma_frags(const Frags<BM, BN, WT>& f, f32x16 (&acc)[BM / 64][BN / 64]) {
    constexpr int TM = BM / 64, TN = BN / 64;
#pragma unroll
    for (int c = C0; c < C1; ++c) {
        if constexpr (F16) {
            f16x4 ah[TM], bh[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ah[i] = to_half4(f.a[c][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bh[j] = to_half4(f.b[c][j]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s == 0 ? f.a[c][i].x : s == 1 ? f.a[c][i].y : s == 2 ? f.a[c][i].z : f.a[c][i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float bv = s == 0 ? f.b[c][j].x : s == 1 ? f.b[c][j].y : s == 2 ? f.b[c][j].z : f.b[c][j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
        }
    }
}

template <int BM, int BN, int AT>
struct Lds {      // float offsets inside the ring region
    static constexpr int STAGES = Ring<BM, BN>::stages;
    static constexpr int A_STAGE = BM * BK, B_STAGE = BK * BN;
    static constexpr int A_SLOTS = AT ? 2 : STAGES;                            // fused: transformed tiles, written one step ahead
    static constexpr int B_OFF = A_SLOTS * A_STAGE;
    static constexpr int RAW_SLOTS = AT ? STAGES - 1 : 0;                      // fused: raw DMA targets, per source
    static constexpr int RAW_OFF = B_OFF + STAGES * B_STAGE;
    static constexpr int RAW2_OFF = RAW_OFF + RAW_SLOTS * A_STAGE;
    static constexpr int RING = RAW2_OFF + (ATr<AT>::two ? RAW_SLOTS * A_STAGE : 0);
    static constexpr int TILE = BM * (BN + 4) + 2 * 4 * BN * 2 + 4;            // staged tile + statistics exchange (two gates) + reducer flag
    static constexpr int FIXED = RING > TILE ? RING : TILE;                    // the coefficient table follows
};

// One pipeline step of the plain kernel with COMPILE-TIME stage addresses and restrict-qualified views of the ring.
// Branch-free: wait for this step's loads, barrier, read the stage's fragments, first half of the MFMAs, issue step+2,
// second half.
template <int BM, int BN, bool WT, bool F16>
__device__ __forceinline__ void pipe_step(const IgemmArgs& p, P3dKTap* taps, float* __restrict__ a_dst, float* __restrict__ b_dst,
                                          const float* __restrict__ a_src, const float* __restrict__ b_src,
                                          f32x16 (&acc)[BM / 64][BN / 64], ALoad<BM / 32, false>& sa, BLoad<BN / 32>& sb, int nsteps,
                                          int kchunks, int wave, int lane, int wm, int wn) {
    constexpr int LPS = BM / 32 + BN / 32;
    // loads of this step have landed for this wave; with a 3-stage ring the next step's may still fly
    wait_vmcnt<(Ring<BM, BN>::stages - 2) * LPS>();
    __builtin_amdgcn_s_barrier();      // ... and for every wave; everyone is also done reading the stage refilled next
    Frags<BM, BN, WT> f;
    load_frags<BM, BN, WT>(a_src, b_src, f, wm, wn, lane >> 5, lane & 31);
    __builtin_amdgcn_sched_barrier(0);      // keep every read above the MFMAs (hipcc otherwise sinks half of them back)
    mfma_frags<BM, BN, WT, F16, 0, BK / 16>(f, acc);
    issue_a_dma<BM>(p, taps, a_dst, sa, nsteps, kchunks, false, wave, lane);
    issue_b_dma<BN, WT>(p, taps, b_dst, sb, nsteps, kchunks, wave, lane);
    mfma_frags<BM, BN, WT, F16, BK / 16, BK / 8>(f, acc);
}
// ... and of the fused-BatchNorm kernel: the A slot consumed now was written (transformed) during the previous step;
// between the MFMA halves the lane transforms the NEXT step's A chunks out of their raw slot, then re-targets that raw slot
// with the DMA of STAGES steps ahead and issues the B DMA of STAGES - 1 steps ahead.
template <int BM, int BN, bool WT, bool F16, int AT>
__device__ __forceinline__ void pipe_step_fused(const IgemmArgs& p, P3dKTap* taps, float* __restrict__ a_next, float* __restrict__ b_dst,
                                                float* __restrict__ raw, float* __restrict__ raw2,
                                                const float* __restrict__ a_src, const float* __restrict__ b_src,
                                                const float* __restrict__ tab, int kp, f32x16 (&acc)[BM / 64][BN / 64],
                                                AMeta& r, ALoad<BM / 32, ATr<AT>::two>& sa, BLoad<BN / 32>& sb,
                                                int nsteps, int kchunks, int wave, int lane, int wm, int wn) {
    constexpr int LPS = (BM / 32) * (ATr<AT>::two ? 2 : 1) + BN / 32;
    wait_vmcnt<(Ring<BM, BN>::stages - 2) * LPS>();      // this step's B tile and the next step's raw A chunks have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my transformed chunks of this step are in LDS
    __builtin_amdgcn_s_barrier();
    Frags<BM, BN, WT> f;
    load_frags<BM, BN, WT>(a_src, b_src, f, wm, wn, lane >> 5, lane & 31);
    XIn<BM / 32, AT> xin;
    transform_load<BM, AT>(r, raw, raw2, sa.achunk, tab, kp, xin, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    mfma_frags<BM, BN, WT, F16, 0, BK / 16>(f, acc);
    transform_store<BM, AT>(r, xin, a_next, wave, lane);
    issue_a_raw<BM, AT>(p, taps, raw, raw2, r, sa, nsteps, kchunks, false, wave, lane);
    issue_b_dma<BN, WT>(p, taps, b_dst, sb, nsteps, kchunks, wave, lane);
    mfma_frags<BM, BN, WT, F16, BK / 16, BK / 8>(f, acc);
}

template <int BM, int BN, bool WT, bool F16, int K>
struct StepLoop {
    static __device__ __forceinline__ void run(const IgemmArgs& p, P3dKTap* taps, float* ring, f32x16 (&acc)[BM / 64][BN / 64],
                                               ALoad<BM / 32, false>& sa, BLoad<BN / 32>& sb, int base, int nsteps, int kchunks,
                                               int wave, int lane, int wm, int wn) {
        using L = Lds<BM, BN, 0>;
        if constexpr (K < L::STAGES) {
            if (base + K < nsteps) {
                constexpr int D = (K + L::STAGES - 1) % L::STAGES;      // stage refilled while stage K is consumed
                pipe_step<BM, BN, WT, F16>(p, taps, ring + D * L::A_STAGE, ring + L::B_OFF + D * L::B_STAGE, ring + K * L::A_STAGE,
                                           ring + L::B_OFF + K * L::B_STAGE, acc, sa, sb, nsteps, kchunks, wave, lane, wm, wn);
            }
            StepLoop<BM, BN, WT, F16, K + 1>::run(p, taps, ring, acc, sa, sb, base, nsteps, kchunks, wave, lane, wm, wn);
        }
    }
};
// fused: the period of (B slot, A slot, raw slot) is STAGES * (STAGES - 1) steps, all indices compile-time
template <int BM, int BN, bool WT, bool F16, int AT, int K>
struct FusedLoop {
    static __device__ __forceinline__ void run(const IgemmArgs& p, P3dKTap* taps, float* ring, const float* tab, int kp, f32x16 (&acc)[BM / 64][BN / 64],
                                               AMeta (&meta)[Ring<BM, BN>::stages - 1], ALoad<BM / 32, ATr<AT>::two>& sa,
                                               BLoad<BN / 32>& sb, int base, int nsteps, int kchunks, int wave, int lane, int wm, int wn) {
        using L = Lds<BM, BN, AT>;
        constexpr int S = L::STAGES, PERIOD = S * (S - 1);
        if constexpr (K < PERIOD) {
            if (base + K < nsteps) {
                constexpr int RS = (K + 1) % (S - 1);       // raw slot of step K + 1 (and then of step K + S)
                pipe_step_fused<BM, BN, WT, F16, AT>(p, taps, ring + ((K + 1) % 2) * L::A_STAGE, ring + L::B_OFF + ((K + S - 1) % S) * L::B_STAGE,
                                                     ring + L::RAW_OFF + RS * L::A_STAGE, ring + L::RAW2_OFF + RS * L::A_STAGE,
                                                     ring + (K % 2) * L::A_STAGE, ring + L::B_OFF + (K % S) * L::B_STAGE, tab, kp, acc,
                                                     meta[RS], sa, sb, nsteps, kchunks, wave, lane, wm, wn);
            }
            FusedLoop<BM, BN, WT, F16, AT, K + 1>::run(p, taps, ring, tab, kp, acc, meta, sa, sb, base, nsteps, kchunks, wave, lane, wm, wn);
        }
    }
};

// ---- coefficient tables of the fused BatchNorms (one thread per channel, every block computes the same bits) ------------
// channel k's sums over up to P3D_FOLD_MAX per-tile partials, in tile order, double accumulation.  Every load is issued
// before the first add (indices past the end re-read the last partial and are not added): ONE memory latency, where a
// counted loop's remainder iterations would each pay their own (measured: +7 us per launch at 13 partials).
__device__ __forceinline__ void fold_partials(const float* part_base, int nparts, int C, int k, double& s1, double& s2) {
    const float2* part = reinterpret_cast<const float2*>(part_base) + k;
    float2 v[P3D_FOLD_MAX];
#pragma unroll
    for (int q = 0; q < P3D_FOLD_MAX; ++q) v[q] = part[(size_t)min(q, nparts - 1) * C];
#pragma unroll
    for (int q = 0; q < P3D_FOLD_MAX; ++q)
        if (q < nparts) { s1 += (double)v[q].x; s2 += (double)v[q].y; }
}
// forward: scale / shift from the producer's per-tile (sum, sumsq): mean and variance in double (E[x^2] - mean^2 cancels),
// the rest in float; `owner` (block 0 of the launch, when the launch is the BN's publisher) stores what the backward
// pass needs and updates the moving statistics (tf.layers.batch_normalization, momentum 0.99)
__device__ __forceinline__ void bn_fold_channel(const BnFold& f, int k, bool owner, float& sc, float& sh) {
    if (!f.part) { sc = f.scale[k]; sh = f.shift[k]; return; }
    double s1 = 0.0, s2 = 0.0;
    fold_partials(f.part, f.nparts, f.C, k, s1, s2);
    const double mean = s1 * f.inv_m;
    double var = s2 * f.inv_m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, varf = (float)var;
    const float inv = 1.0f / sqrtf(varf + f.eps);
    sc = f.gamma[k] * inv;
    sh = f.beta[k] - meanf * sc;
    if (owner && f.publish) {
        f.scale[k] = sc; f.shift[k] = sh; f.mean[k] = meanf; f.invstd[k] = inv;
        if (f.update_moving) {      // moving -= (moving - batch) * (1 - 0.99)   (biased variance)
            f.moving_mean[k] -= (f.moving_mean[k] - meanf) * (1.0f - 0.99f);
            f.moving_var[k] -= (f.moving_var[k] - varf) * (1.0f - 0.99f);
        }
    }
}
// backward: dy = gamma*invstd * (g - mean(g) - xhat * mean(g*xhat))  as  k1*g + k2*y + k3
__device__ __forceinline__ void bn_grad_fold_channel(const BnGradFold& f, int k, bool owner, float& k1, float& k2, float& k3) {
    if (!f.part) { k1 = f.coef[k]; k2 = f.coef[f.C + k]; k3 = f.coef[2 * f.C + k]; return; }
    double sg = 0.0, sgx = 0.0;
    fold_partials(f.part, f.nparts, f.C, k, sg, sgx);
    const float inv = f.invstd[k];
    const float c1 = (float)(sg * f.inv_m), c2 = (float)(sgx * f.inv_m);
    k1 = f.gamma[k] * inv;
    k2 = -k1 * inv * c2;
    k3 = -k1 * c1 - k2 * f.mean[k];
    if (owner && f.publish) {
        f.coef[k] = k1; f.coef[f.C + k] = k2; f.coef[2 * f.C + k] = k3;
        f.dgamma[k] = (float)sgx; f.dbeta[k] = (float)sg;
    }
}

template <int AT>
__device__ __forceinline__ void build_table(const IgemmArgs& p, float* tab, int kp, bool owner) {
    for (int k = threadIdx.x; k < kp; k += 256) {
        if constexpr (AT == P3D_AT_RELU1 || AT == P3D_AT_RELU2) {
            float s = 0.f, t = 0.f;
            if (k < p.K) bn_fold_channel(p.f1, k, owner, s, t);
            tab[k] = s; tab[kp + k] = t;
            if constexpr (AT == P3D_AT_RELU2) {
                float s2 = 0.f, t2 = 0.f;
                if (k < p.K) bn_fold_channel(p.f2, k, owner, s2, t2);
                tab[2 * kp + k] = s2; tab[3 * kp + k] = t2;
            }
        } else if constexpr (AT == P3D_AT_GRAD) {
            float k1 = 0.f, k2 = 0.f, k3 = 0.f;
            if (k < p.K) bn_grad_fold_channel(p.gf, k, owner, k1, k2, k3);
            tab[k] = k1; tab[kp + k] = k2; tab[2 * kp + k] = k3;
        }
    }
}

template <int BM, int BN, bool WT, bool F16, int AT>
__device__ __forceinline__ void igemm2_body(const IgemmArgs& p, const Geo& geo, const int tile_id, const int slice) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LA = BM / 32;                 // A chunks per lane per step (per source)
    using L = Lds<BM, BN, AT>;
    constexpr int STAGES = L::STAGES;

    P3D_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* rowIdx = reinterpret_cast<int*>(smem);                           // [BM] output row of tile row r, -1 past M
    float* ring = reinterpret_cast<float*>(rowIdx + BM);                  // ring; the epilogue tile overlays it
    float* tab = ring + L::FIXED;                                         // [ncoef][kp] operand-transform coefficients

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const long long M = (long long)p.N * geo.Gd * geo.Gh * geo.Gw;
    const int NT = (p.Nc + BN - 1) / BN;
    const int nsplit = geo.nsplit;
    P3dKTap* taps = geo.taps;
    const int nt = tile_id % NT;
    const int mt = tile_id / NT;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;
    const int kchunks = (p.K + BK - 1) / BK;
    const int kp = kchunks * BK;
    const float4 bias4 = igemm_bias_prefetch<BN>(geo.bias, n0, p.Nc);      // for the epilogue; in flight behind the whole main loop

    const unsigned Mu = (unsigned)M, m0u = (unsigned)m0;      // launcher guarantees M < 2^31
    // the epilogue's output-row table: written once the first operand loads are in flight (it used to stand between the kernel
    // entry and them); nobody reads it before the barrier that ends the main loop
    auto fill_row_table = [&]() {
        for (int r = tid; r < BM; r += 256) {
            const unsigned m = m0u + r;
            int ro = -1;
            if (m < Mu) {
                const unsigned t1 = p3d_div(m, geo.fGw), gw = m - t1 * (unsigned)geo.Gw;
                const unsigned t2 = p3d_div(t1, geo.fGh), gh = t1 - t2 * (unsigned)geo.Gh;
                const unsigned n = p3d_div(t2, geo.fGd), gd = t2 - n * (unsigned)geo.Gd;
                const int od = gd * p.osd + geo.ood, oh = gh * p.osh + geo.ooh, ow = gw * p.osw + geo.oow;
                ro = (((int)n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
            }
            rowIdx[r] = ro;
        }
    };
    // ---- this block's slice of the (tap, k-chunk) steps --------------------------------------------
    const int total_steps = geo.ntaps * kchunks;
    const int per = (total_steps + nsplit - 1) / nsplit;
    const int s_begin = slice * per;
    const int s_end = min(total_steps, s_begin + per);
    const int nsteps = max(s_end - s_begin, 0);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    ALoad<LA, ATr<AT>::two> sa;
    BLoad<BN / 32> sb;
    a_init<BM, ATr<AT>::two>(p, geo, sa, m0u, Mu, wave, lane, s_begin, kchunks);
    b_init<BN, WT>(p, geo.w, sb, n0, wave, lane, s_begin, kchunks);
    if constexpr (AT == P3D_AT_NONE) {
        // prologue: STAGES-1 steps in flight; then step k computes from stage k % STAGES while refilling the stage
        // that was consumed one step earlier.  All stage addresses are compile-time constants (StepLoop).
#pragma unroll
        for (int k = 0; k < STAGES - 1; ++k) {
            issue_a_dma<BM>(p, taps, ring + k * L::A_STAGE, sa, nsteps, kchunks, k == 0, wave, lane);
            issue_b_dma<BN, WT>(p, taps, ring + L::B_OFF + k * L::B_STAGE, sb, nsteps, kchunks, wave, lane);
        }
        fill_row_table();
        for (int base = 0; base < nsteps; base += STAGES)
            StepLoop<BM, BN, WT, F16, 0>::run(p, taps, ring, acc, sa, sb, base, nsteps, kchunks, wave, lane, wm, wn);
    } else {
        // fused prologue, in the issue order the steady state has (pipe_step_fused's counted wait relies on it):
        //   A(0);  [A(1), B(0)]  (3-stage ring only);  the coefficient table (its loads queue behind those);  transform A(0);
        //   [A(STAGES-1), B(STAGES-2)]
        AMeta meta[STAGES - 1];
        float* raw = ring + L::RAW_OFF;
        float* raw2 = ring + L::RAW2_OFF;
        fill_row_table();
        issue_a_raw<BM, AT>(p, taps, raw, raw2, meta[0], sa, nsteps, kchunks, true, wave, lane);
        if constexpr (STAGES == 3) {
            issue_a_raw<BM, AT>(p, taps, raw + L::A_STAGE, raw2 + L::A_STAGE, meta[1], sa, nsteps, kchunks, false, wave, lane);
            issue_b_dma<BN, WT>(p, taps, ring + L::B_OFF, sb, nsteps, kchunks, wave, lane);
        }
        build_table<AT>(p, tab, kp, tile_id == 0 && slice == 0);
        wait_vmcnt<0>();
        __syncthreads();                                    // table complete; the DMA above has landed
        {
            XIn<LA, AT> xin;
            transform_load<BM, AT>(meta[0], raw, raw2, sa.achunk, tab, kp, xin, wave, lane);
            transform_store<BM, AT>(meta[0], xin, ring, wave, lane);
        }
        issue_a_raw<BM, AT>(p, taps, raw, raw2, meta[0], sa, nsteps, kchunks, false, wave, lane);
        issue_b_dma<BN, WT>(p, taps, ring + L::B_OFF + (STAGES - 2) * L::B_STAGE, sb, nsteps, kchunks, wave, lane);
        for (int base = 0; base < nsteps; base += STAGES * (STAGES - 1))
            FusedLoop<BM, BN, WT, F16, AT, 0>::run(p, taps, ring, tab, kp, acc, meta, sa, sb, base, nsteps, kchunks, wave, lane, wm, wn);
    }

    // ---- epilogue ----------------------------------------------------------------------------------
    // Stage the tile through LDS (the ring is free once the tail DMA has landed) so that global traffic is row-wise
    // float4 -- bias, optional accumulate (batched loads instead of 64 dependent dword read-modify-writes per lane)
    // and the per-channel statistics all come from the staged tile.
    constexpr int LDT = BN + 4;
    constexpr int F4R = BN / 4;
    float* tile = ring;
    float* sred = tile + BM * LDT;                              // [2][4][BN][2] statistics exchange
    int* flag = reinterpret_cast<int*>(sred + 2 * 4 * BN * 2);  // "this block reduces the slices" (same LDS array: no second object)
    wait_vmcnt<0>();
    __syncthreads();                                            // also orders rowIdx (written above) before its readers
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                tile[r * LDT + wn * (BN / 2) + j * 32 + l31] = acc[i][j][e];
            }
    __syncthreads();

    if (nsplit > 1) {
        // -- partial tile -> slab, arrival ticket; the last arriver folds the slabs back into the LDS tile ----------
        // Slab stores are WRITE-THROUGH (sc1): they need no release fence (whose L2 write-back costs 2-6 us per block);
        // every storing wave drains its stores, the block meets at a barrier, one lane takes the ticket
        // (cdna_hip_programming.md Guideline 16, recipe R1).
        float* myslab = geo.slab + ((size_t)tile_id * nsplit + slice) * (BM * BN);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(myslab, 0, BM * BN * 4, 0x00020000);
#pragma unroll 4
        for (int i = tid; i < BM * F4R; i += 256) {
            const int r = i / F4R, c4 = (i - r * F4R) * 4;
            const float4 v = *reinterpret_cast<const float4*>(tile + r * LDT + c4);
            const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(u, rs, (r * BN + c4) * 4, 0, 16);      // aux 16 = sc1
        }
        wait_vmcnt<0>();                                        // every storing wave drains its stores ...
        __syncthreads();
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(geo.cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == (unsigned)(nsplit - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // drop this CU's stale lines before the plain slab loads
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                geo.cnt[tile_id] = 0;                           // ready for the next launch that uses this scratch
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        const float* slabs = geo.slab + (size_t)tile_id * nsplit * (BM * BN);
        // The slabs come from other CUs' write-through stores: every load is a long-latency miss, so keep 16 of them in
        // flight per lane (4 tile positions x 4 slices) and add in slice order.
        constexpr int PER_LANE = BM * F4R / 256;                // float4 positions per lane: 4 / 8 / 16
        static_assert(PER_LANE % 4 == 0, "reducer unrolls four tile positions");
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
            int sidx = 1;
            for (; sidx + 3 < nsplit; sidx += 4) {
                float4 a[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[t][q] = *reinterpret_cast<const float4*>(src[q] + (size_t)(sidx + t) * (BM * BN));
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { v[q].x += a[t][q].x; v[q].y += a[t][q].y; v[q].z += a[t][q].z; v[q].w += a[t][q].w; }
            }
            for (; sidx < nsplit; ++sidx) {
                float4 a[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = *reinterpret_cast<const float4*>(src[q] + (size_t)sidx * (BM * BN));
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q].x += a[q].x; v[q].y += a[q].y; v[q].z += a[q].z; v[q].w += a[q].w; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = tid + (i0 + q) * 256;
                const int r = i / F4R, c4 = (i - r * F4R) * 4;
                *reinterpret_cast<float4*>(tile + r * LDT + c4) = v[q];
            }
        }
        __syncthreads();
    }

    igemm_tile_epilogue<BM, BN>(p, geo, tile, sred, rowIdx, mt, n0, M, bias4);
}

// Block -> tile so that the blocks of one XCD (block index mod 8) work on CONSECUTIVE tiles: the N tiles of one row range and
// the row ranges next to it (the neighbours a k x k x k tap reaches into) then share that XCD's L2 instead of being
// fetched by eight of them.  A bijection on [0, T); launches with few tiles keep the identity.
__device__ __forceinline__ int xcd_tile(int b, int T, int min_tiles) {
    if (T < min_tiles) return b;
    const int x = b & 7, q = T >> 3, r = T & 7;
    return x * q + min(x, r) + (b >> 3);
}

template <int BM, int BN, bool WT, bool F16 = false, int AT = 0>
__global__ __launch_bounds__(256) void igemm2_kernel(const IgemmArgs p) {
    p3d_warm_kernargs<IgemmArgs>();
    Geo geo;
    geo.Gd = p.Gd; geo.Gh = p.Gh; geo.Gw = p.Gw; geo.fGd = p.fGd; geo.fGh = p.fGh; geo.fGw = p.fGw;
    geo.ood = p.ood; geo.ooh = p.ooh; geo.oow = p.oow; geo.stat_base = p.stat_base; geo.ntaps = p.ntaps; geo.taps = p3d_kernarg_taps(offsetof(IgemmArgs, taps));
    geo.w = p.w; geo.bias = p.bias; geo.y = p.y; geo.statpart = p.statpart;
    geo.nsplit = p.nsplit; geo.slab = p.slab; geo.cnt = p.cnt;
    const int tile = p.nsplit == 1 ? xcd_tile((int)blockIdx.x, (int)gridDim.x, p.xcd_min_tiles) : (int)blockIdx.x;
    igemm2_body<BM, BN, WT, F16, AT>(p, geo, tile, (int)blockIdx.y);      // block -> (output tile, K-slice)
}
// One launch for the residue classes of a transposed conv / strided input gradient: block ranges per class, heaviest class
// first (a class with eight taps runs eight times as long per tile as one with a single tap: the late blocks are the short ones)
template <int BM, int BN, bool WT, bool F16 = false>
__global__ __launch_bounds__(256) void igemm2_group_kernel(const IgemmGroupArgs g) {
    p3d_warm_kernargs<IgemmArgs>();            // the part every class shares (the class tables follow it)
    int c = 0;
#pragma unroll
    for (int q = 1; q < P3D_IGEMM_CLASSES; ++q)
        if (q < g.nclass && (int)blockIdx.x >= g.cls[q].blk0) c = q;
    Geo geo;
    geo.Gd = g.cls[c].Gd; geo.Gh = g.cls[c].Gh; geo.Gw = g.cls[c].Gw; geo.fGd = g.cls[c].fGd; geo.fGh = g.cls[c].fGh; geo.fGw = g.cls[c].fGw;
    geo.ood = g.cls[c].ood; geo.ooh = g.cls[c].ooh; geo.oow = g.cls[c].oow; geo.stat_base = g.cls[c].stat_base; geo.ntaps = g.cls[c].ntaps;
    geo.taps = p3d_kernarg_taps(offsetof(IgemmGroupArgs, cls) + (size_t)c * sizeof(IgemmClass) + offsetof(IgemmClass, taps));
    geo.w = g.cls[c].w; geo.bias = g.cls[c].bias; geo.y = g.cls[c].y; geo.statpart = g.cls[c].statpart;
    // K-sliced classes: consecutive blocks of a class are the slices of one tile
    // (a tail class starts at tile0 of its grid: the body indexes scratch by tile, so the class's share is rebased by tile0)
    const int tile0 = g.cls[c].tile0;
    geo.nsplit = g.cls[c].nsplit;
    geo.slab = g.common.slab + (g.cls[c].slab0 - (long long)tile0 * geo.nsplit * (BM * BN));
    geo.cnt = g.common.cnt + (g.cls[c].cnt0 - tile0);
    const int local = (int)blockIdx.x - g.cls[c].blk0;
    const int tile = geo.nsplit == 1 ? xcd_tile(local, g.cls[c].ntiles, g.common.xcd_min_tiles) : local / geo.nsplit;
    igemm2_body<BM, BN, WT, F16, 0>(g.common, geo, tile0 + tile, local % geo.nsplit);
}

constexpr int MAX_TABLE_FLOATS = 4 * 2048;        // coefficient table: up to 4 coefficients x 2048 reduction channels

template <int BM, int BN, int AT>
constexpr size_t smem_fixed_bytes() { return (size_t)BM * 4 + (size_t)Lds<BM, BN, AT>::FIXED * 4; }

template <int BM, int BN, bool WT, bool F16, int AT>
hipError_t launch_variant(const IgemmArgs& a, dim3 grid, hipStream_t s) {
    constexpr size_t fixed = smem_fixed_bytes<BM, BN, AT>();
    static std::once_flag once;
    std::call_once(once, [] {
        hipFuncSetAttribute((const void*)igemm2_kernel<BM, BN, WT, F16, AT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(fixed + (AT ? MAX_TABLE_FLOATS * 4 : 0)));
    });
    const size_t kp = (size_t)((a.K + BK - 1) / BK) * BK;
    const size_t sm = fixed + (size_t)ATr<AT>::ncoef * kp * 4;
    hipLaunchKernelGGL((igemm2_kernel<BM, BN, WT, F16, AT>), grid, dim3(256), sm, s, a);
    return hipGetLastError();
}

int xcd_min_tiles() {                // P3D_TUNE_XCD_MIN_TILES (tuning builds): A/B runs of the XCD-contiguous tile order
    static const int v = [] { const char* e = p3d_tune_env("P3D_TUNE_XCD_MIN_TILES"); return e ? atoi(e) : 64; }();
    return v;
}
bool tail_split_enabled() {          // P3D_TUNE_NO_TAIL=1 (tuning builds): A/B runs without the K-sliced tail class
    static const bool on = [] { const char* e = p3d_tune_env("P3D_TUNE_NO_TAIL"); return !(e && atoi(e)); }();
    return on;
}
// K-slices for the `rem` tiles of a launch's last round (1: leave them whole).  Whole, they cost one tile time whatever
// their number; cut into s slices they run in ceil(rem * s / 256) rounds of 1/s tile time each, plus the slab exchange
// (3 + 0.4 s us, the fit of heuristic_plan).  136 tiles of 144 steps (deconv2's input gradient): s = 7 -> 4 rounds of 1/7.
int tail_slices(long long rem, int steps, int bm, int bn) {
    if (rem <= 0 || !tail_split_enabled()) return 1;
    const double step_us = bm * bn == 16384 ? 2.3 : (bm * bn == 8192 ? 1.35 : 0.78);
    const double whole = steps * step_us;
    double best = whole * 0.9;                    // a cut has to win 10 % of a tile time
    int best_s = 1;
    for (int sl = 2; sl <= 16 && steps / sl >= 4; ++sl) {
        const long long rounds = (rem * sl + 255) / 256;
        const double cost = (double)rounds * ((steps + sl - 1) / sl) * step_us + 3.0 + 0.4 * sl;
        if (cost < best) { best = cost; best_s = sl; }
    }
    return best_s;
}

template <int BM, int BN>
hipError_t launch_t(const IgemmArgs& a0, const P3dIgemmPlan& pl, hipStream_t s) {
    IgemmArgs a = a0;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const long long tiles = ((M + BM - 1) / BM) * ((a.Nc + BN - 1) / BN);
    const int splits = pl.splits < 1 ? 1 : pl.splits;
    a.nsplit = splits;
    a.xcd_min_tiles = xcd_min_tiles();
    a.slab = nullptr; a.cnt = nullptr;
    if (splits > 1) {
        const hipError_t e = p3d_stream_scratch(s, (size_t)tiles * splits * BM * BN, (size_t)tiles, &a.slab, &a.cnt);
        if (e != hipSuccess) return e;
    }
    const dim3 grid((unsigned)tiles, (unsigned)splits);
    // forward convs ([K][Nc] weights) take the relu transforms, input gradients ([Nc][K] weights) the gradient transform
    if (a.wT) {
        if (a.at_mode == P3D_AT_GRAD) return a.f16 ? launch_variant<BM, BN, true, true, P3D_AT_GRAD>(a, grid, s) : launch_variant<BM, BN, true, false, P3D_AT_GRAD>(a, grid, s);
        if (a.at_mode != P3D_AT_NONE) return hipErrorInvalidValue;
        return a.f16 ? launch_variant<BM, BN, true, true, 0>(a, grid, s) : launch_variant<BM, BN, true, false, 0>(a, grid, s);
    }
    switch (a.at_mode) {
        case P3D_AT_NONE: return a.f16 ? launch_variant<BM, BN, false, true, 0>(a, grid, s) : launch_variant<BM, BN, false, false, 0>(a, grid, s);
        case P3D_AT_RELU1: return a.f16 ? launch_variant<BM, BN, false, true, P3D_AT_RELU1>(a, grid, s) : launch_variant<BM, BN, false, false, P3D_AT_RELU1>(a, grid, s);
        case P3D_AT_RELU2: return a.f16 ? launch_variant<BM, BN, false, true, P3D_AT_RELU2>(a, grid, s) : launch_variant<BM, BN, false, false, P3D_AT_RELU2>(a, grid, s);
        default: return hipErrorInvalidValue;
    }
}

// ---- per-stream scratch (partial tiles + arrival counters) -----------------------------------------------------
struct Scratch { float* slab = nullptr; size_t slab_floats = 0; unsigned* cnt = nullptr; size_t counters = 0; };
std::vector<void*> g_scratch_allocs;       // every buffer ever handed out (outgrown ones stay valid until shutdown)
std::map<hipStream_t, Scratch> g_scratch;
std::mutex g_scratch_mutex;

}  // namespace

hipError_t p3d_stream_scratch(hipStream_t s, size_t slab_floats, size_t counters, float** slab, unsigned** cnt) {
    std::lock_guard<std::mutex> g(g_scratch_mutex);
    Scratch& sc = g_scratch[s];
    if (slab_floats > sc.slab_floats) {
        // grow with headroom; the outgrown buffer is deliberately not freed (a captured graph may still name it)
        const size_t want = slab_floats + slab_floats / 2 + (1u << 20);
        float* p = nullptr;
        const hipError_t e = hipMalloc((void**)&p, want * sizeof(float));
        if (e != hipSuccess) return e;
        sc.slab = p; sc.slab_floats = want; g_scratch_allocs.push_back(p);
    }
    if (counters > sc.counters) {
        const size_t want = counters * 2 + 4096;
        unsigned* p = nullptr;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(unsigned));
        if (e != hipSuccess) return e;
        // tickets start at zero; every reducer re-zeroes its own.  Zeroed IN STREAM ORDER: a null-stream memset is not ordered
        // against the launch that follows on a non-blocking stream
        e = hipMemsetAsync(p, 0, want * sizeof(unsigned), s);
        if (e != hipSuccess) return e;
        sc.cnt = p; sc.counters = want; g_scratch_allocs.push_back(p);
    }
    *slab = sc.slab; *cnt = sc.cnt;
    return hipSuccess;
}

// Test hook: arrival counters that are not zero although nothing is in flight (every K-sliced launch must leave its counters
// as it found them); -1 on a HIP error.
long long p3d_scratch_dirty_counters() {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    std::lock_guard<std::mutex> g(g_scratch_mutex);
    long long dirty = 0;
    for (auto& kv : g_scratch) {
        const Scratch& sc = kv.second;
        if (!sc.cnt || !sc.counters) continue;
        std::vector<unsigned> h(sc.counters);
        if (hipMemcpyAsync(h.data(), sc.cnt, sc.counters * sizeof(unsigned), hipMemcpyDeviceToHost, nullptr) != hipSuccess ||
            hipStreamSynchronize(nullptr) != hipSuccess)
            return -1;      // (the device is idle: synchronised above)
        for (unsigned v : h) dirty += v != 0;
    }
    return dirty;
}

void p3d_release_scratch() {
    std::lock_guard<std::mutex> g(g_scratch_mutex);
    for (void* p : g_scratch_allocs) hipFree(p);
    g_scratch_allocs.clear();
    g_scratch.clear();
}

namespace {

// ---- plan: tile and K-slices per launch ------------------------------------------------------------------------
// Big layers take the biggest tile that still yields enough blocks (least LDS traffic per FLOP).  Layers with few
// output tiles cut K into slices: wall time is about one block's latency until blocks exceed the 256 CUs, so the
// slice count aims at 0.75-1x the CU count, never 2x (cdna_hip_programming.md, "Projection GEMM at M = 256", item 1).
// Every plan is numerically valid; plans differ only in speed and in the (fixed, per-plan) summation order of the K-slices.
const char* plan_name(int bm, int bn) {
    return bm == 128 ? (bn == 128 ? "igemm2_kernel<128,128>" : "igemm2_kernel<128,64>") : "igemm2_kernel<64,64>";
}

struct PlanOverride { int tile = -1, splits = 0; };
PlanOverride g_override;
std::mutex g_plan_mutex;

P3dIgemmPlan heuristic_plan(const IgemmArgs& a) {
    P3dIgemmPlan pl;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const int kchunks = (a.K + BK - 1) / BK;
    const int steps = a.ntaps * kchunks;
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((a.Nc + bn - 1) / bn); };
    const long long want = 384;
    // (with one to four K steps a tile is mostly prologue and epilogue, and twice as many 128x64 tiles quantise better on the
    // chip: stage 1's 64 -> 256 convs 44.5 -> 40.5 us, their 256 <- 64 input gradients 51 -> 45.6 us; tools/op_times.py --igemm-tile)
    // 128x128 from 512 tiles (two per CU): 392 tiles -- stage 2's 1x3x3 convs at 32x224x224 -- run 20 % faster as 784 of 128x64
    // ... and not when the last 128-column tile would be half empty or worse (192 output channels, the unet++ head's x_1_1 input
    // gradient: 2707 us as two column tiles of 128, 2165 us as three of 64)
    const bool ragged128 = (a.Nc % 128) != 0 && (a.Nc % 128) <= 64;
    if (a.Nc > 64 && tiles(128, 128) >= 512 && steps > 4 && !ragged128) { pl.bm = 128; pl.bn = 128; }
    else if (tiles(128, 64) >= want || (a.Nc <= 64 && tiles(128, 64) >= 128)) { pl.bm = 128; pl.bn = 64; }
    else { pl.bm = 64; pl.bn = 64; }
    pl.splits = 1;
    const long long t = tiles(pl.bm, pl.bn);
    if (pl.bm == 64 && pl.bn == 64 && t < 256 && steps >= 8) {       // (196 tiles x 36 steps, stage 2's 1x3x3 convs: two slices, 38.4 -> 35.5 us)
        // Measured on MI355X (tools/micro/conv_chain.hip, M = 784 rows, cold weights): a 64x64 step costs ~0.78 us while
        // at most one block sits on a CU and proportionally more beyond that; cutting K into s slices costs ~3 + 0.4 s us
        // (write-through slab stores, ticket, the last arriver's s slab reads, 16 loads in flight per lane).  Pick the
        // power of two that minimises the sum; more than 8 slices never paid.
        double best_cost = 1e30;
        for (int sp = 1; sp <= 8; sp *= 2) {
            if (sp > 1 && steps / sp < 3) break;
            const double per_block = (double)((steps + sp - 1) / sp) * 0.78 * std::max(1.0, (double)(t * sp) / 256.0);
            const double cost = per_block + (sp > 1 ? 3.0 + 0.4 * sp : 0.0);
            if (cost < best_cost) { best_cost = cost; pl.splits = sp; }
        }
    }
    pl.name = plan_name(pl.bm, pl.bn);
    return pl;
}

}  // namespace

void p3d_tune_begin(hipStream_t) {}
void p3d_tune_end() {}
void p3d_igemm2_override(int tile, int splits) {      // tests / tools only
    std::lock_guard<std::mutex> g(g_plan_mutex);
    g_override.tile = tile; g_override.splits = splits;
}

// Tile / K-slice choice for one launch.
P3dIgemmPlan p3d_igemm2_plan(const IgemmArgs& a, int) {
    P3dIgemmPlan pl = heuristic_plan(a);
    PlanOverride ov;
    { std::lock_guard<std::mutex> g(g_plan_mutex); ov = g_override; }
    const int steps = a.ntaps * ((a.K + BK - 1) / BK);
    if (ov.tile == 0) { pl.bm = 64; pl.bn = 64; }
    else if (ov.tile == 1) { pl.bm = 128; pl.bn = 64; }
    else if (ov.tile == 2 && a.Nc > 64) { pl.bm = 128; pl.bn = 128; }
    if (ov.splits >= 1 && ov.splits <= steps) pl.splits = ov.splits;
    pl.name = plan_name(pl.bm, pl.bn);
    // dense 1x1x1 convs over many positions: the weights-resident streaming kernel (a forced tile / slice count keeps the tiled one)
    if (ov.tile < 0 && ov.splits < 1) {
        pl.stream_blocks = p3d_pw_stream_blocks(a);
        if (pl.stream_blocks > 0) { pl.splits = 1; pl.name = "pw_stream_kernel"; }
    }
    return pl;
}

hipError_t p3d_launch_igemm2(const IgemmArgs& a0, const P3dIgemmPlan& pl, hipStream_t s) {
    IgemmArgs a = a0;
    a.fGd = p3d_fastdiv((unsigned)a.Gd); a.fGh = p3d_fastdiv((unsigned)a.Gh); a.fGw = p3d_fastdiv((unsigned)a.Gw);
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (M <= 0 || a.Nc <= 0) return hipSuccess;
    if (M >= (1ll << 31) || (long long)a.N * a.Di * a.Hi * a.Wi >= (1ll << 31)) return hipErrorInvalidValue;
    if ((long long)a.N * a.Do * a.Ho * a.Wo >= (1ll << 31)) return hipErrorInvalidValue;                    // int row index
    if (a.Gd * a.isd >= 1024 || a.Gh * a.ish >= 1024 || a.Gw * a.isw >= 1024) return hipErrorInvalidValue;   // packed coords
    if (a.ntaps > P3D_MAX_TAPS) return hipErrorInvalidValue;
    if ((a.K & 3) || (a.ldx & 3) || !a.zeros) return hipErrorInvalidValue;
    if ((a.Nc & 3) || (a.ldy & 3)) return hipErrorInvalidValue;
    if (pl.splits > 1 && pl.splits > a.ntaps * ((a.K + BK - 1) / BK)) return hipErrorInvalidValue;
    if (a.at_mode != P3D_AT_NONE) {
        const long long kp = (long long)((a.K + BK - 1) / BK) * BK;
        if (kp * 4 > MAX_TABLE_FLOATS) return hipErrorInvalidValue;
        if ((a.at_mode == P3D_AT_RELU2 || a.at_mode == P3D_AT_GRAD) && (!a.x2 || (a.ldx2 & 3))) return hipErrorInvalidValue;
    }
    if (a.at_mode == P3D_AT_RELU1 || a.at_mode == P3D_AT_RELU2) {
        if (a.f1.part && (a.f1.nparts < 1 || a.f1.nparts > P3D_FOLD_MAX)) return hipErrorInvalidValue;
        if (a.at_mode == P3D_AT_RELU2 && a.f2.part && (a.f2.nparts < 1 || a.f2.nparts > P3D_FOLD_MAX)) return hipErrorInvalidValue;
    }
    if (a.at_mode == P3D_AT_GRAD && a.gf.part && (a.gf.nparts < 1 || a.gf.nparts > P3D_FOLD_MAX)) return hipErrorInvalidValue;
    if (a.ngate < 0 || a.ngate > 2) return hipErrorInvalidValue;
    for (int q = 0; q < a.ngate; ++q)
        if (!a.gate[q].y || !a.gate[q].out || !a.gate[q].part || (a.gate[q].ldy & 3) || (a.gate[q].ldo & 3)) return hipErrorInvalidValue;
    if (a.ngate && a.statpart) return hipErrorInvalidValue;
    if (pl.stream_blocks > 0) return p3d_launch_pw_stream(a, s);
    if (p3d_igemm2_tail_split(a, pl)) return p3d_launch_igemm2_group(&a, 1, pl, s);      // full rounds + a K-sliced tail class
    if (pl.bm == 128 && pl.bn == 128) return launch_t<128, 128>(a, pl, s);
    if (pl.bm == 128 && pl.bn == 64) return launch_t<128, 64>(a, pl, s);
    return launch_t<64, 64>(a, pl, s);
}

// ---- grouped launch of residue classes ---------------------------------------------------------------------------------
bool p3d_igemm2_groupable(const IgemmArgs* v, int n, const P3dIgemmPlan& pl) {
    if (n < 2 || n > P3D_IGEMM_CLASSES) return false;
    for (int i = 0; i < n; ++i) {
        const IgemmArgs& a = v[i];
        // shared: the gathered operand and every extent / stride; per class: grid, offsets, taps, weights, bias, output, statistics
        if (a.at_mode != P3D_AT_NONE || a.ngate || a.x != v[0].x || a.wT != v[0].wT || a.K != v[0].K || a.Nc != v[0].Nc || a.N != v[0].N ||
            a.f16 != v[0].f16 || a.accum != v[0].accum || a.ldx != v[0].ldx || a.ldy != v[0].ldy || a.Di != v[0].Di || a.Hi != v[0].Hi ||
            a.Wi != v[0].Wi || a.Do != v[0].Do || a.Ho != v[0].Ho || a.Wo != v[0].Wo || a.isd != v[0].isd || a.ish != v[0].ish ||
            a.isw != v[0].isw || a.osd != v[0].osd || a.osh != v[0].osh || a.osw != v[0].osw)
            return false;
        const P3dIgemmPlan q = p3d_igemm2_plan(a, 1);
        if (q.bm != pl.bm || q.bn != pl.bn) return false;          // (each class keeps its own K-slice count)
    }
    return true;
}

namespace {
template <int BM, int BN>
hipError_t launch_group_t(IgemmGroupArgs& g, const long long* tiles, hipStream_t s) {
    long long blocks = 0, slabs = 0, counters = 0;
    for (int q = 0; q < g.nclass; ++q) {
        g.cls[q].blk0 = (int)blocks; blocks += tiles[q] * g.cls[q].nsplit;
        g.cls[q].ntiles = (int)tiles[q];
        g.cls[q].slab0 = 0; g.cls[q].cnt0 = 0;
        if (g.cls[q].nsplit > 1) {
            g.cls[q].slab0 = slabs * (long long)(BM * BN); g.cls[q].cnt0 = counters;
            slabs += tiles[q] * g.cls[q].nsplit; counters += tiles[q];
        }
    }
    if (blocks <= 0 || blocks >= (1ll << 31)) return blocks <= 0 ? hipSuccess : hipErrorInvalidValue;
    g.common.slab = nullptr; g.common.cnt = nullptr;
    if (slabs > 0) {
        const hipError_t e = p3d_stream_scratch(s, (size_t)slabs * BM * BN, (size_t)counters, &g.common.slab, &g.common.cnt);
        if (e != hipSuccess) return e;
    }
    constexpr size_t sm = smem_fixed_bytes<BM, BN, 0>();
    static std::once_flag once;
    std::call_once(once, [] {
        hipFuncSetAttribute((const void*)igemm2_group_kernel<BM, BN, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipFuncSetAttribute((const void*)igemm2_group_kernel<BM, BN, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipFuncSetAttribute((const void*)igemm2_group_kernel<BM, BN, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipFuncSetAttribute((const void*)igemm2_group_kernel<BM, BN, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
    });
    const dim3 grid((unsigned)blocks), blk(256);
    const IgemmArgs& a = g.common;
    if (a.f16) {
        if (a.wT) hipLaunchKernelGGL((igemm2_group_kernel<BM, BN, true, true>), grid, blk, sm, s, g);
        else hipLaunchKernelGGL((igemm2_group_kernel<BM, BN, false, true>), grid, blk, sm, s, g);
    } else {
        if (a.wT) hipLaunchKernelGGL((igemm2_group_kernel<BM, BN, true, false>), grid, blk, sm, s, g);
        else hipLaunchKernelGGL((igemm2_group_kernel<BM, BN, false, false>), grid, blk, sm, s, g);
    }
    return hipGetLastError();
}
}  // namespace


// Would the tail of this single launch be cut into K-slices (then it goes out through the grouped kernel)?
bool p3d_igemm2_tail_split(const IgemmArgs& a, const P3dIgemmPlan& pl) {
    if (a.at_mode || a.ngate || pl.splits > 1 || pl.stream_blocks > 0 || !tail_split_enabled()) return false;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const long long tiles = ((M + pl.bm - 1) / pl.bm) * ((a.Nc + pl.bn - 1) / pl.bn);
    const int steps = a.ntaps * ((a.K + BK - 1) / BK);
    return tiles >= 256 && tail_slices(tiles % 256, steps, pl.bm, pl.bn) >= 2;
}

hipError_t p3d_launch_igemm2_group(const IgemmArgs* v, int n, const P3dIgemmPlan& pl, hipStream_t s) {
    if (n == 1 ? !p3d_igemm2_tail_split(v[0], pl) : !p3d_igemm2_groupable(v, n, pl)) return hipErrorInvalidValue;
    // the checks of the single launch, per class
    for (int i = 0; i < n; ++i) {
        const IgemmArgs& a = v[i];
        const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
        if (M >= (1ll << 31) || (long long)a.N * a.Di * a.Hi * a.Wi >= (1ll << 31) || (long long)a.N * a.Do * a.Ho * a.Wo >= (1ll << 31))
            return hipErrorInvalidValue;
        if (a.Gd * a.isd >= 1024 || a.Gh * a.ish >= 1024 || a.Gw * a.isw >= 1024 || a.ntaps > P3D_MAX_TAPS) return hipErrorInvalidValue;
        if ((a.K & 3) || (a.ldx & 3) || !a.zeros || (a.Nc & 3) || (a.ldy & 3)) return hipErrorInvalidValue;
    }
    // heaviest class first
    int order[P3D_IGEMM_CLASSES];
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order, order + n, [&](int x, int y) { return v[x].ntaps > v[y].ntaps; });
    IgemmGroupArgs g;
    memset(&g, 0, sizeof(g));
    g.common = v[0];
    g.common.xcd_min_tiles = xcd_min_tiles();
    g.common.nsplit = 1; g.common.slab = nullptr; g.common.cnt = nullptr;
    long long tiles[P3D_IGEMM_CLASSES];
    int nc = 0;
    for (int k = 0; k < n; ++k) {
        const IgemmArgs& a = v[order[k]];
        const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
        if (M <= 0) continue;
        IgemmClass& c = g.cls[nc];
        c.Gd = a.Gd; c.Gh = a.Gh; c.Gw = a.Gw;
        c.fGd = p3d_fastdiv((unsigned)a.Gd); c.fGh = p3d_fastdiv((unsigned)a.Gh); c.fGw = p3d_fastdiv((unsigned)a.Gw);
        c.ood = a.ood; c.ooh = a.ooh; c.oow = a.oow; c.stat_base = a.stat_base; c.ntaps = a.ntaps;
        c.w = a.w; c.bias = a.bias; c.y = a.y; c.statpart = a.statpart;
        c.nsplit = std::max(1, p3d_igemm2_plan(a, 1).splits);
        for (int t = 0; t < a.ntaps; ++t) c.taps[t] = a.taps[t];
        tiles[nc] = ((M + pl.bm - 1) / pl.bm) * ((a.Nc + pl.bn - 1) / pl.bn);
        ++nc;
    }
    // Tail of the last wave: with one block per CU a launch of T tiles takes ceil(T / 256) rounds, and a last round that fills
    // a small part of the chip costs a whole tile time (deconv3's input gradient at 8 clips: 784 tiles of 128x128 = 3.06
    // rounds).  The tiles of that round become a class of their own, K-sliced so that its blocks fill the round.
    if (nc >= 1 && nc < P3D_IGEMM_CLASSES && g.cls[nc - 1].nsplit == 1) {
        long long blocks = 0;
        for (int q = 0; q < nc; ++q) blocks += tiles[q] * g.cls[q].nsplit;
        const long long rem = blocks % 256;
        const IgemmClass& last = g.cls[nc - 1];
        const int steps = last.ntaps * ((g.common.K + BK - 1) / BK);
        const int sl = tail_slices(rem, steps, pl.bm, pl.bn);
        if (blocks >= 256 && rem > 0 && rem < tiles[nc - 1] && sl >= 2) {
            g.cls[nc] = last;
            g.cls[nc].tile0 = (int)(tiles[nc - 1] - rem);
            g.cls[nc].nsplit = sl;
            tiles[nc] = rem;
            tiles[nc - 1] -= rem;
            ++nc;
        }
    }
    g.nclass = nc;
    if (nc == 0) return hipSuccess;
    if (pl.bm == 128 && pl.bn == 128) return launch_group_t<128, 128>(g, tiles, s);
    if (pl.bm == 128 && pl.bn == 64) return launch_group_t<128, 64>(g, tiles, s);
    return launch_group_t<64, 64>(g, tiles, s);
}
