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
//
// fp32 in / fp32 accumulate: v_mfma_f32_32x32x2_f32, exact fp32 at the fp32 peak (157 TFLOP/s).
#include "p3d_kernels.h"
#include <algorithm>
#include <cstdio>
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
// Loader waves (-DP3D_LW64=1; OFF: measured, no gain).  A 64x64 step costs 0.78 us; with the DMA issue compiled out it
// costs 0.545 us, with the MFMAs compiled out 0.42 us (tools/micro/conv_chain.hip, -DP3D_TUNE_NO_DMA / _NO_MFMA), which
// suggested that DMA issue (~150 cycles per global_load_lds in the issuing wave) and MFMA issue serialise in one wave's
// instruction stream.  The 8-wave form below -- waves 0-3 only read fragments and issue MFMAs, waves 4-7 only wait for
// and issue the LDS-DMA, one of each per SIMD, loaders leaving after the last step -- removes that serialisation, and
// the step still costs 0.77 us (stage-3 shapes: 57.0 vs 58.4 us per launch at 72 steps; grouped filter gradients 43.9
// vs 38.3 us, slower).  Deeper rings (4, 5 stages) change nothing either.  What is left is the matrix pipe itself on
// real (non-zero) operands at the clock the chip holds under load: the DMA-free figure above ran on stale LDS contents.
#ifndef P3D_LW64
#define P3D_LW64 0
#endif
template <int BM, int BN>
struct Loaders { static constexpr bool on = (BM == 64 && BN == 64 && P3D_LW64); static constexpr int threads = on ? 512 : 256; };

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_wave_base) {
    // LDS destination = wave-uniform base + lane * 16 B
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int V>
struct IC { static constexpr int value = V; };

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Per-lane loader state, all in registers (an LDS read in the load path would make hipcc drain the
// in-flight LDS-DMA first).  Row pointers are rebuilt only when the kernel tap changes; within a
// tap a step just advances 32 floats along the channel run.
template <int LA, int LB>
struct LoadState {
    int base[LA];            // n * Di*Hi*Wi, or -1 for rows past M
    int dhw[LA];             // (g_d*is_d) << 20 | (g_h*is_h) << 10 | (g_w*is_w)
    const float* aptr[LA];   // row start for the current tap (+ this lane's 16-byte chunk), null when padded
    int achunk[LA];          // 4 * logical chunk held by this lane's slot
    int boff[LB];            // weight element offset of this lane's piece within a tap slab (k0 = 0)
    int bk[LB];              // WT: 4 * logical chunk; !WT: k row within the step
    bool bok[LB];            // n inside Nc
    const float* wt;         // slab of the current tap
    int t, kc;               // next step to issue: tap index, k-chunk index
    int issued;              // steps issued so far
};

template <int BM, int BN, bool WT>
__device__ __forceinline__ void loader_init(const IgemmArgs& p, LoadState<BM / 32, BN / 32>& st, unsigned m0, unsigned M,
                                            int n0, int wave, int lane, int s_begin, int kchunks) {
    constexpr int LA = BM / 32, LB = BN / 32;
    const int a_slot = lane & 7, a_sub = lane >> 3;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int r = (i * 4 + wave) * 8 + a_sub;
        const unsigned m = m0 + r;
        st.base[i] = -1; st.dhw[i] = 0; st.aptr[i] = nullptr;
        st.achunk[i] = 4 * (a_slot ^ ((r >> 1) & 7));
        if (m < M) {
            const unsigned gw = m % (unsigned)p.Gw; unsigned t = m / (unsigned)p.Gw;
            const unsigned gh = t % (unsigned)p.Gh; t /= (unsigned)p.Gh;
            const unsigned gd = t % (unsigned)p.Gd; const unsigned n = t / (unsigned)p.Gd;
            st.base[i] = (int)n * p.Di * p.Hi * p.Wi;
            st.dhw[i] = (int)(((gd * p.isd) << 20) | ((gh * p.ish) << 10) | (gw * p.isw));
        }
    }
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
    st.t = s_begin / kchunks; st.kc = s_begin - st.t * kchunks; st.issued = 0; st.wt = p.w;
}

template <int LA, int LB>
__device__ __forceinline__ void loader_set_tap(const IgemmArgs& p, LoadState<LA, LB>& st) {
    const P3dTap tap = p.taps[st.t];
    st.wt = p.w + (long long)tap.widx * p.K * p.Nc;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int id = (st.dhw[i] >> 20) + tap.dd, ih = ((st.dhw[i] >> 10) & 1023) + tap.dh, iw = (st.dhw[i] & 1023) + tap.dw;
        const bool ok = st.base[i] >= 0 && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                        (unsigned)iw < (unsigned)p.Wi;
        st.aptr[i] = ok ? p.x + ((long long)st.base[i] + ((long long)id * p.Hi + ih) * p.Wi + iw) * p.ldx + st.achunk[i] : nullptr;
    }
}

// Issue the LDS-DMA of the next (tap, k-chunk) step -- ALWAYS the same number of loads, so the counted
// vmcnt never changes; steps past the end of this block's slice fetch the zero page.  a_dst / b_dst are
// __restrict__ so that, inlined next to compute_stage, hipcc knows the fragment reads cannot alias the DMA
// targets and does not put s_waitcnt vmcnt(0) in front of them.
template <int BM, int BN, bool WT>
__device__ __forceinline__ void issue_stage(const IgemmArgs& p, float* __restrict__ a_dst, float* __restrict__ b_dst,
                                            LoadState<BM / 32, BN / 32>& st, int nsteps, int kchunks, bool first, int wave,
                                            int lane) {
    constexpr int LA = BM / 32, LB = BN / 32;
    const bool live = st.issued < nsteps;
    if (live && (first || st.kc == 0)) loader_set_tap(p, st);      // wave-uniform, once per tap
    const int k0 = st.kc * BK;
    const float* zp = p.zeros + 4 * (lane & 7);
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const bool ok = live && st.aptr[i] != nullptr && (k0 + st.achunk[i]) < p.K;
        glds16(ok ? st.aptr[i] + k0 : zp, a_dst + (i * 4 + wave) * 8 * BK);
    }
    if (!WT) {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const bool ok = live && st.bok[i] && (k0 + st.bk[i]) < p.K;
            glds16(ok ? st.wt + (long long)k0 * p.Nc + st.boff[i] : zp, b_dst + (i * 4 + wave) * 256);
        }
    } else {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const bool ok = live && st.bok[i] && (k0 + st.bk[i]) < p.K;
            glds16(ok ? st.wt + k0 + st.boff[i] : zp, b_dst + (i * 4 + wave) * 8 * BK);
        }
    }
    ++st.issued;
    if (++st.kc == kchunks) { st.kc = 0; ++st.t; }
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x4 to_half4(float4 v) { f16x4 r = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; return r; }

// F16: the pointwise-conv option of BASELINE configs[4] -- operands stay fp32 in HBM and LDS, the fragments are
// rounded to fp16 in registers and one v_mfma_f32_32x32x8_f16 (fp32 accumulate) replaces four fp32 MFMAs: a lane's
// float4 fragment holds k = 8c + 4h + {0..3}, which is exactly the A / B operand layout of that instruction.
// c-iterations [C0, C1) of one stage (BK / 8 = 4 in all): the stage is consumed in two halves so that the address
// arithmetic and DMA issue of the next refill can run while the first half's MFMAs execute (pipe_step).
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

// One pipeline step with COMPILE-TIME stage addresses and restrict-qualified views of the ring.
// Branch-free: wait for this step's loads, barrier, read the stage's fragments, first half of the MFMAs, issue step+2,
// second half.
template <int BM, int BN, bool WT, bool F16>
__device__ __forceinline__ void pipe_step(const IgemmArgs& p, float* __restrict__ a_dst, float* __restrict__ b_dst,
                                          const float* __restrict__ a_src, const float* __restrict__ b_src,
                                          f32x16 (&acc)[BM / 64][BN / 64], LoadState<BM / 32, BN / 32>& st, int nsteps,
                                          int kchunks, int wave, int lane, int wm, int wn, bool loads, bool computes) {
    constexpr int LPS = BM / 32 + BN / 32;
    constexpr bool LW = Loaders<BM, BN>::on;
#if defined(P3D_TUNE_STAMPS)           // tools/micro only: shader-cycle stamps of block 0 / wave 0 (conv_chain.hip prints them)
    const unsigned long long stamp0 = __builtin_readcyclecounter();
#endif
    // loads of this step have landed for this wave; with a 3-stage ring the next step's may still fly
    if (!LW || loads) wait_vmcnt<(Ring<BM, BN>::stages - 2) * LPS>();
    __builtin_amdgcn_s_barrier();      // ... and for every wave; everyone is also done reading the stage refilled next
#if defined(P3D_TUNE_STAMPS)
    const unsigned long long stamp1 = __builtin_readcyclecounter();
#endif
    if constexpr (LW) {
        if (loads) {
            issue_stage<BM, BN, WT>(p, a_dst, b_dst, st, nsteps, kchunks, false, wave, lane);
        } else {
            Frags<BM, BN, WT> f;
            load_frags<BM, BN, WT>(a_src, b_src, f, wm, wn, lane >> 5, lane & 31);
            __builtin_amdgcn_sched_barrier(0);
            mfma_frags<BM, BN, WT, F16, 0, BK / 8>(f, acc);
        }
        (void)computes;
        return;
    }
    Frags<BM, BN, WT> f;
    load_frags<BM, BN, WT>(a_src, b_src, f, wm, wn, lane >> 5, lane & 31);
    __builtin_amdgcn_sched_barrier(0);      // keep every read above the MFMAs (hipcc otherwise sinks half of them back)
#if defined(P3D_TUNE_NO_MFMA)               // tools/micro only: what does a step cost without the matrix work / without the DMA
    acc[0][0][0] += f.a[0][0].x * f.b[0][0].x + f.a[BK / 8 - 1][0].w * f.b[BK / 8 - 1][0].w;
    issue_stage<BM, BN, WT>(p, a_dst, b_dst, st, nsteps, kchunks, false, wave, lane);
#elif defined(P3D_TUNE_NO_DMA)
    mfma_frags<BM, BN, WT, F16, 0, BK / 16>(f, acc);
    ++st.issued; if (++st.kc == kchunks) { st.kc = 0; ++st.t; }
    mfma_frags<BM, BN, WT, F16, BK / 16, BK / 8>(f, acc);
#else
    mfma_frags<BM, BN, WT, F16, 0, BK / 16>(f, acc);
    issue_stage<BM, BN, WT>(p, a_dst, b_dst, st, nsteps, kchunks, false, wave, lane);
    mfma_frags<BM, BN, WT, F16, BK / 16, BK / 8>(f, acc);
#endif
#if defined(P3D_TUNE_STAMPS)
    if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        const unsigned long long stamp2 = __builtin_readcyclecounter();
        atomicAdd(p.stamps + 0, stamp1 - stamp0); atomicAdd(p.stamps + 1, stamp2 - stamp1); atomicAdd(p.stamps + 2, 1ull);
    }
#endif
}

template <int BM, int BN, bool WT, int K>
struct PrologueLoop {
    static __device__ __forceinline__ void run(const IgemmArgs& p, float* As, float* Bs, LoadState<BM / 32, BN / 32>& st, int nsteps,
                                               int kchunks, int wave, int lane) {
        constexpr int STAGES = Ring<BM, BN>::stages;
        if constexpr (K < STAGES - 1) {
            issue_stage<BM, BN, WT>(p, As + K * (BM * BK), Bs + K * (BK * BN), st, nsteps, kchunks, K == 0, wave, lane);
            PrologueLoop<BM, BN, WT, K + 1>::run(p, As, Bs, st, nsteps, kchunks, wave, lane);
        }
    }
};

template <int BM, int BN, bool WT, bool F16, int K>
struct StepLoop {
    static __device__ __forceinline__ void run(const IgemmArgs& p, float* As, float* Bs, f32x16 (&acc)[BM / 64][BN / 64],
                                               LoadState<BM / 32, BN / 32>& st, int base, int nsteps, int kchunks, int wave,
                                               int lane, int wm, int wn, bool loads, bool computes) {
        constexpr int STAGES = Ring<BM, BN>::stages;
        if constexpr (K < STAGES) {
            if (base + K < nsteps) {
                constexpr int D = (K + STAGES - 1) % STAGES;      // stage refilled while stage K is consumed
                pipe_step<BM, BN, WT, F16>(p, As + D * (BM * BK), Bs + D * (BK * BN), As + K * (BM * BK), Bs + K * (BK * BN), acc, st,
                                      nsteps, kchunks, wave, lane, wm, wn, loads, computes);
            }
            StepLoop<BM, BN, WT, F16, K + 1>::run(p, As, Bs, acc, st, base, nsteps, kchunks, wave, lane, wm, wn, loads, computes);
        }
    }
};

template <int BM, int BN, bool WT, bool F16 = false>
__global__ __launch_bounds__((Loaders<BM, BN>::threads)) void igemm2_kernel(const IgemmArgs p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LA = BM / 32;                 // A glds per wave per step
    constexpr int A_STAGE = BM * BK;            // floats
    constexpr int B_STAGE = BK * BN;
    constexpr int STAGES = Ring<BM, BN>::stages;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long* rowOut = reinterpret_cast<long long*>(smem);               // [BM]
    float* As = reinterpret_cast<float*>(rowOut + BM);                    // ring; the epilogue tile overlays it
    float* Bs = As + STAGES * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr bool LW = Loaders<BM, BN>::on;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;                              // piece / sub-tile index of this wave in its role
    const bool loads = !LW || wave8 >= 4, computes = !LW || wave8 < 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

#if defined(P3D_TUNE_STAMPS)
    const unsigned long long kstamp0 = __builtin_readcyclecounter();
#endif
    const long long M = (long long)p.N * p.Gd * p.Gh * p.Gw;
    const int NT = (p.Nc + BN - 1) / BN;
    const int nsplit = p.nsplit;
    // block -> (output tile, K-slice).  xmap: consecutive blocks are the slices of one tile, so with round-robin
    // dispatch over the 8 XCDs slice s of every tile lands on XCD s (mod 8) and the A rows / weight slabs of a slice
    // are fetched into one L2 once instead of eight times.  Placement is a speed matter only.
    const int tile_id = p.xmap ? (int)(blockIdx.x / nsplit) : (int)blockIdx.x;
    const int slice = p.xmap ? (int)(blockIdx.x - (unsigned)tile_id * nsplit) : (int)blockIdx.y;
    const int nt = tile_id % NT;
    const int mt = tile_id / NT;
    const long long m0 = (long long)mt * BM;
    const int n0 = nt * BN;

    const unsigned Mu = (unsigned)M, m0u = (unsigned)m0;      // launcher guarantees M < 2^31
    for (int r = tid; r < BM; r += Loaders<BM, BN>::threads) {
        const unsigned m = m0u + r;
        long long ro = -1;
        if (m < Mu) {
            const unsigned gw = m % (unsigned)p.Gw; unsigned t = m / (unsigned)p.Gw;
            const unsigned gh = t % (unsigned)p.Gh; t /= (unsigned)p.Gh;
            const unsigned gd = t % (unsigned)p.Gd; const unsigned n = t / (unsigned)p.Gd;
            const int od = gd * p.osd + p.ood, oh = gh * p.osh + p.ooh, ow = gw * p.osw + p.oow;
            ro = ((((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * p.ldy;
        }
        rowOut[r] = ro;
    }

    // ---- this block's slice of the (tap, k-chunk) steps --------------------------------------------
    const int kchunks = (p.K + BK - 1) / BK;
    const int total_steps = p.ntaps * kchunks;
    const int per = (total_steps + nsplit - 1) / nsplit;
    const int s_begin = slice * per;
    const int s_end = min(total_steps, s_begin + per);
#if defined(P3D_TUNE_NO_LOOP)               // tools/micro only: the fixed cost of a launch (prologue + epilogue, no K loop)
    const int nsteps = 0;
#else
    const int nsteps = max(s_end - s_begin, 0);
#endif

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#if defined(P3D_TUNE_STAMPS)
    const unsigned long long kstampA = __builtin_readcyclecounter();     // after the output-row table
#endif
    LoadState<LA, BN / 32> st;
    if (loads) loader_init<BM, BN, WT>(p, st, m0u, Mu, n0, wave, lane, s_begin, kchunks);
#if defined(P3D_TUNE_STAMPS)
    const unsigned long long kstampB = __builtin_readcyclecounter();     // after the loader's index arithmetic
    if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { atomicAdd(p.stamps + 6, kstampA - kstamp0); atomicAdd(p.stamps + 7, kstampB - kstampA); }
#endif
    // prologue: STAGES-1 steps in flight; then step k computes from stage k % STAGES while refilling the stage
    // that was consumed one step earlier.  All stage addresses are compile-time constants (StepLoop).
    if (loads) PrologueLoop<BM, BN, WT, 0>::run(p, As, Bs, st, nsteps, kchunks, wave, lane);
#if defined(P3D_TUNE_STAMPS)
    const unsigned long long kstamp1 = __builtin_readcyclecounter();
#endif
    for (int base = 0; base < nsteps; base += STAGES)
        StepLoop<BM, BN, WT, F16, 0>::run(p, As, Bs, acc, st, base, nsteps, kchunks, wave, lane, wm, wn, loads, computes);
#if defined(P3D_TUNE_STAMPS)
    const unsigned long long kstamp2 = __builtin_readcyclecounter();
#endif

    // ---- epilogue ----------------------------------------------------------------------------------
    // Stage the tile through LDS (the ring is free once the tail DMA has landed) so that global traffic is row-wise
    // float4 -- bias, optional accumulate (batched loads instead of 64 dependent dword read-modify-writes per lane)
    // and the per-channel statistics all come from the staged tile.
    constexpr int LDT = BN + 4;
    constexpr int F4R = BN / 4;
    float* tile = As;
    float* sred = tile + BM * LDT;                              // [4][BN][2] statistics exchange
    int* flag = reinterpret_cast<int*>(sred + 4 * BN * 2);      // "this block reduces the slices" (same LDS array: no second object)
    wait_vmcnt<0>();
    __syncthreads();                                            // also orders rowOut (written above) before its readers
    if (LW && !computes) return;                                // loader waves are done: the epilogue is the compute waves'
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
        float* myslab = p.slab + ((size_t)tile_id * nsplit + slice) * (BM * BN);
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
            const unsigned ticket = __hip_atomic_fetch_add(p.cnt + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == (unsigned)(nsplit - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // drop this CU's stale lines before the plain slab loads
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                p.cnt[tile_id] = 0;                             // ready for the next launch that uses this scratch
            }
            *flag = last;
        }
        __syncthreads();
        if (!*flag) return;
        const float* slabs = p.slab + (size_t)tile_id * nsplit * (BM * BN);
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

    // -- output rows: bias, optional accumulate, row-wise float4 stores; the stored values go back to the tile for the
    //    statistics pass --------------------------------------------------------------------------------------------
    const bool want_stats = p.statpart != nullptr;
#pragma unroll 4
    for (int i = tid; i < BM * F4R; i += 256) {
        const int r = i / F4R, c4 = (i - r * F4R) * 4;
        const long long ro = rowOut[r];
        const int col = n0 + c4;
        if (ro < 0 || col >= p.Nc) continue;
        float4 v = *reinterpret_cast<const float4*>(tile + r * LDT + c4);
        if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
        if (want_stats) *reinterpret_cast<float4*>(tile + r * LDT + c4) = v;
        float* dst = p.y + ro + col;
        if (p.accum) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *reinterpret_cast<float4*>(dst) = v;
    }
    if (want_stats) {
        // per-channel (sum, sumsq) over this tile's valid rows: 4 row groups x BN columns, folded in a fixed order
        __syncthreads();
        constexpr int RG = 256 / BN, RPG = BM / RG;
        const int col = tid % BN, rg = tid / BN;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
        for (int r = rg * RPG; r < (rg + 1) * RPG; ++r)
            if (rowOut[r] >= 0) { const float v = tile[r * LDT + col]; s1 += v; s2 = fmaf(v, v, s2); }
        sred[(rg * BN + col) * 2] = s1; sred[(rg * BN + col) * 2 + 1] = s2;
        __syncthreads();
        if (tid < BN && (n0 + tid) < p.Nc) {
            float t1 = sred[tid * 2], t2 = sred[tid * 2 + 1];
#pragma unroll
            for (int g = 1; g < RG; ++g) { t1 += sred[(g * BN + tid) * 2]; t2 += sred[(g * BN + tid) * 2 + 1]; }
            float* dst = p.statpart + ((size_t)(p.stat_base + mt) * p.Nc + n0 + tid) * 2;
            dst[0] = t1; dst[1] = t2;
        }
    }
#if defined(P3D_TUNE_STAMPS)
    if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long kstamp3 = __builtin_readcyclecounter();
        atomicAdd(p.stamps + 3, kstamp1 - kstamp0); atomicAdd(p.stamps + 4, kstamp3 - kstamp2); atomicAdd(p.stamps + 5, 1ull);
    }
#endif
}

template <int BM, int BN>
constexpr size_t smem_bytes() {
    const size_t ring = (size_t)Ring<BM, BN>::stages * (BM * BK + BK * BN) * 4;
    const size_t tile = (size_t)BM * (BN + 4) * 4 + 4 * BN * 2 * 4 + 16;   // staged tile + statistics exchange + reducer flag
    return BM * 8 + (ring > tile ? ring : tile);
}

template <int BM, int BN>
hipError_t launch_t(const IgemmArgs& a0, const P3dIgemmPlan& pl, hipStream_t s) {
    IgemmArgs a = a0;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const long long tiles = ((M + BM - 1) / BM) * ((a.Nc + BN - 1) / BN);
    const int splits = pl.splits < 1 ? 1 : pl.splits;
    a.nsplit = splits; a.xmap = (splits > 1 && pl.xmap) ? 1 : 0;
    a.slab = nullptr; a.cnt = nullptr;
    if (splits > 1) {
        const hipError_t e = p3d_stream_scratch(s, (size_t)tiles * splits * BM * BN, (size_t)tiles, &a.slab, &a.cnt);
        if (e != hipSuccess) return e;
    }
    const dim3 grid = a.xmap ? dim3((unsigned)(tiles * splits)) : dim3((unsigned)tiles, (unsigned)splits);
    constexpr size_t sm = smem_bytes<BM, BN>();
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)igemm2_kernel<BM, BN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipFuncSetAttribute((const void*)igemm2_kernel<BM, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipFuncSetAttribute((const void*)igemm2_kernel<BM, BN, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipFuncSetAttribute((const void*)igemm2_kernel<BM, BN, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        attr_done = true;
    }
    if (a.f16) {
        if (a.wT) hipLaunchKernelGGL((igemm2_kernel<BM, BN, true, true>), grid, dim3(Loaders<BM, BN>::threads), sm, s, a);
        else      hipLaunchKernelGGL((igemm2_kernel<BM, BN, false, true>), grid, dim3(Loaders<BM, BN>::threads), sm, s, a);
        return hipGetLastError();
    }
    if (a.wT) hipLaunchKernelGGL((igemm2_kernel<BM, BN, true>), grid, dim3(Loaders<BM, BN>::threads), sm, s, a);
    else      hipLaunchKernelGGL((igemm2_kernel<BM, BN, false>), grid, dim3(Loaders<BM, BN>::threads), sm, s, a);
    return hipGetLastError();
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
        e = hipMemset(p, 0, want * sizeof(unsigned));        // tickets start at zero; every reducer re-zeroes its own
        if (e != hipSuccess) return e;
        sc.cnt = p; sc.counters = want; g_scratch_allocs.push_back(p);
    }
    *slab = sc.slab; *cnt = sc.cnt;
    return hipSuccess;
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
// slice count aims at 0.75-1x the CU count, never 2x (cdna_hip_programming.md, "Projection GEMM at M = 256", item 1),
// and is a divisor pattern of 8 so that xmap can give each XCD one slice.  Every plan is numerically valid; plans
// differ only in speed and in the (fixed, per-plan) summation order of the K-slices.
const char* plan_name(int bm, int bn) {
    return bm == 128 ? (bn == 128 ? "igemm2_kernel<128,128>" : "igemm2_kernel<128,64>") : "igemm2_kernel<64,64>";
}

struct PlanOverride { int tile = -1, splits = 0, xmap = -1; };
PlanOverride g_override;
std::once_flag g_override_once;
std::mutex g_plan_mutex;

void read_override_env() {
    // Tuning sweeps only (tools/): force the tile, the K-slice count or the block mapping.  Read once; a forced value
    // that a shape cannot take is ignored for that shape.
    if (const char* e = getenv("P3D_TILE")) g_override.tile = atoi(e);
    if (const char* e = getenv("P3D_SPLITS")) g_override.splits = atoi(e);
    if (const char* e = getenv("P3D_XMAP")) g_override.xmap = atoi(e);
    if (g_override.tile >= 0 || g_override.splits > 0 || g_override.xmap >= 0)
        fprintf(stderr, "[p3d] WARNING: igemm2 plan override in effect (P3D_TILE=%d P3D_SPLITS=%d P3D_XMAP=%d): tuning only\n",
                g_override.tile, g_override.splits, g_override.xmap);
}

P3dIgemmPlan heuristic_plan(const IgemmArgs& a) {
    P3dIgemmPlan pl;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const int kchunks = (a.K + BK - 1) / BK;
    const int steps = a.ntaps * kchunks;
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((a.Nc + bn - 1) / bn); };
    const long long want = 384;
    if (a.Nc > 64 && tiles(128, 128) >= want) { pl.bm = 128; pl.bn = 128; }
    else if (tiles(128, 64) >= want || (a.Nc <= 64 && tiles(128, 64) >= 128)) { pl.bm = 128; pl.bn = 64; }
    else { pl.bm = 64; pl.bn = 64; }
    pl.splits = 1; pl.xmap = 0;
    const long long t = tiles(pl.bm, pl.bn);
    if (pl.bm == 64 && pl.bn == 64 && t < 150 && steps >= 8) {
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
void p3d_igemm2_override(int tile, int splits, int xmap) {      // tools/micro only
    std::call_once(g_override_once, read_override_env);
    std::lock_guard<std::mutex> g(g_plan_mutex);
    g_override.tile = tile; g_override.splits = splits; g_override.xmap = xmap;
}

// Tile / K-slice choice for one launch.
P3dIgemmPlan p3d_igemm2_plan(const IgemmArgs& a, int) {
    std::call_once(g_override_once, read_override_env);
    P3dIgemmPlan pl = heuristic_plan(a);
    PlanOverride ov;
    { std::lock_guard<std::mutex> g(g_plan_mutex); ov = g_override; }
    const int steps = a.ntaps * ((a.K + BK - 1) / BK);
    if (ov.tile == 0) { pl.bm = 64; pl.bn = 64; }
    else if (ov.tile == 1) { pl.bm = 128; pl.bn = 64; }
    else if (ov.tile == 2 && a.Nc > 64) { pl.bm = 128; pl.bn = 128; }
    if (ov.splits >= 1 && ov.splits <= steps) { pl.splits = ov.splits; if (pl.splits == 1) pl.xmap = 0; }
    if (ov.xmap >= 0) pl.xmap = (ov.xmap && pl.splits > 1) ? 1 : 0;
    pl.name = plan_name(pl.bm, pl.bn);
    return pl;
}

namespace { hipError_t launch_plan(const IgemmArgs& a, const P3dIgemmPlan& pl, hipStream_t s); }
hipError_t p3d_launch_igemm2(const IgemmArgs& a, const P3dIgemmPlan& pl, hipStream_t s) { return launch_plan(a, pl, s); }

namespace {
hipError_t launch_plan(const IgemmArgs& a0, const P3dIgemmPlan& pl, hipStream_t s) {
    IgemmArgs a = a0;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (M <= 0 || a.Nc <= 0) return hipSuccess;
    if (M >= (1ll << 31) || (long long)a.N * a.Di * a.Hi * a.Wi >= (1ll << 31)) return hipErrorInvalidValue;
    if (a.Gd * a.isd >= 1024 || a.Gh * a.ish >= 1024 || a.Gw * a.isw >= 1024) return hipErrorInvalidValue;   // packed coords
    if (a.ntaps > P3D_MAX_TAPS) return hipErrorInvalidValue;
    if ((a.K & 3) || (a.ldx & 3) || !a.zeros) return hipErrorInvalidValue;
    if ((a.Nc & 3) || (a.ldy & 3)) return hipErrorInvalidValue;
    if (pl.splits > 1 && pl.splits > a.ntaps * ((a.K + BK - 1) / BK)) return hipErrorInvalidValue;
    if (pl.bm == 128 && pl.bn == 128) return launch_t<128, 128>(a, pl, s);
    if (pl.bm == 128 && pl.bn == 64) return launch_t<128, 64>(a, pl, s);
    return launch_t<64, 64>(a, pl, s);
}
}  // namespace
