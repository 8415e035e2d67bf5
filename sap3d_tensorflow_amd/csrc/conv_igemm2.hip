// Pipelined implicit-GEMM convolution for gfx950 -- the workhorse of the P3D path (every conv /
// conv-input-gradient / conv3d_transpose except the Cin=3 stem; reference p3d.py:19,24,86,112,125,
// 200-216).  Same geometry contract as conv_igemm.hip (p3d_kernels.h), different machinery:
//
//  * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction)
//    into a 3-stage ring; loads for step s+2 are in flight while step s is on the matrix cores;
//    ONE raw s_barrier per step with a counted s_waitcnt vmcnt (never 0 in the steady state);
//  * the LDS images are lane-linear (a glds constraint), so the bank-conflict swizzle is applied
//    on the per-lane SOURCE address and again on the ds_read_b128 (16-byte chunk q of row r is
//    stored at chunk q ^ ((r >> 1) & 7): conflict-free for the 32x32x2 A/B fragment reads);
//  * rows that fall into SAME padding, and channel tails, read from a zero page instead of
//    branching, so every wave issues the same number of loads per step (the vmcnt count);
//  * split-K: gridDim.y slices the (tap, k-chunk) step range so that layers with few output
//    tiles (M = B*98 positions in stage 3) still cover 256 CUs; partial tiles are combined with
//    fp32 global atomics into a pre-zeroed output (bias rides on slice 0).
//
// fp32 in / fp32 accumulate: v_mfma_f32_32x32x2_f32, exact fp32 at the fp32 peak (157 TFLOP/s).
#include "p3d_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int STAGES = 3;

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

// Per-lane loader state: the LA rows (8-row pieces) this lane fetches every step, kept in registers
// (an LDS read here would make hipcc drain the in-flight LDS-DMA first).
template <int LA>
struct RowRegs {
    int base[LA];      // n * Di*Hi*Wi, or -1 for rows past M
    int dhw[LA];       // (g_d*is_d) << 20 | (g_h*is_h) << 10 | (g_w*is_w)
};

// Issue the LDS-DMA of one (tap, k-chunk) step.  a_dst / b_dst are __restrict__ so that, once
// inlined next to compute_stage, hipcc knows the fragment reads cannot alias the DMA targets and
// does not put s_waitcnt vmcnt(0) in front of them.
template <int BM, int BN, bool WT>
__device__ __forceinline__ void issue_stage(const IgemmArgs& p, float* __restrict__ a_dst, float* __restrict__ b_dst,
                                            const RowRegs<BM / 32>& rr, int gs, int kchunks, int n0, int wave, int lane) {
    constexpr int LA = BM / 32, LB = BN / 32;
    const int a_slot = lane & 7, a_sub = lane >> 3;
    const int t = gs / kchunks;
    const int k0 = (gs - t * kchunks) * BK;
    const P3dTap tap = p.taps[t];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int piece = i * 4 + wave;                 // 8 rows x 128 B
        const int r = piece * 8 + a_sub;
        const int q = a_slot ^ ((r >> 1) & 7);          // logical chunk stored in this lane's slot
        const int id = (rr.dhw[i] >> 20) + tap.dd, ih = ((rr.dhw[i] >> 10) & 1023) + tap.dh, iw = (rr.dhw[i] & 1023) + tap.dw;
        const bool ok = rr.base[i] >= 0 && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                        (unsigned)iw < (unsigned)p.Wi && (k0 + 4 * q) < p.K;
        const float* src = ok ? p.x + (((long long)rr.base[i] + ((long long)id * p.Hi + ih) * p.Wi + iw) * p.ldx + k0 + 4 * q)
                              : p.zeros + 4 * a_slot;
        glds16(src, a_dst + piece * 8 * BK);
    }
    const float* wt = p.w + (long long)tap.widx * p.K * p.Nc;
    if (!WT) {
        // [k][n] image, rows of BN floats, linear
        constexpr int LANES_PER_ROW = BN / 4;
        constexpr int ROWS_PER_PIECE = 64 / LANES_PER_ROW;
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int piece = i * 4 + wave;
            const int kr = piece * ROWS_PER_PIECE + lane / LANES_PER_ROW;
            const int nc = (lane % LANES_PER_ROW) * 4;
            const bool ok = (k0 + kr) < p.K && (n0 + nc) < p.Nc;
            const float* src = ok ? wt + (long long)(k0 + kr) * p.Nc + n0 + nc : p.zeros + 4 * a_slot;
            glds16(src, b_dst + piece * 256);
        }
    } else {
        // [n][k] image like A (rows of 32 floats, swizzled)
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int piece = i * 4 + wave;
            const int r = piece * 8 + a_sub;
            const int q = a_slot ^ ((r >> 1) & 7);
            const bool ok = (n0 + r) < p.Nc && (k0 + 4 * q) < p.K;
            const float* src = ok ? wt + (long long)(n0 + r) * p.K + k0 + 4 * q : p.zeros + 4 * a_slot;
            glds16(src, b_dst + piece * 8 * BK);
        }
    }
}

template <int BM, int BN, bool WT>
__device__ __forceinline__ void compute_stage(const float* __restrict__ a_st, const float* __restrict__ b_st,
                                              f32x16 (&acc)[BM / 64][BN / 64], int wm, int wn, int h, int l31) {
    constexpr int TM = BM / 64, TN = BN / 64;
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
        float4 a[TM];
        float4 b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * (BM / 2) + i * 32 + l31;
            const int slot = (2 * c + h) ^ ((r >> 1) & 7);
            a[i] = *reinterpret_cast<const float4*>(a_st + r * BK + slot * 4);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * (BN / 2) + j * 32 + l31;
            if (WT) {
                const int slot = (2 * c + h) ^ ((col >> 1) & 7);
                b[j] = *reinterpret_cast<const float4*>(b_st + col * BK + slot * 4);
            } else {
                const float* bp = b_st + (c * 8 + 4 * h) * BN + col;
                b[j] = make_float4(bp[0], bp[BN], bp[2 * BN], bp[3 * BN]);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float av = s == 0 ? a[i].x : s == 1 ? a[i].y : s == 2 ? a[i].z : a[i].w;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float bv = s == 0 ? b[j].x : s == 1 ? b[j].y : s == 2 ? b[j].z : b[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                }
            }
    }
}

// One pipeline step with COMPILE-TIME stage addresses and restrict-qualified views of the ring.
template <int BM, int BN, bool WT>
__device__ __forceinline__ void pipe_step(const IgemmArgs& p, float* __restrict__ a_dst, float* __restrict__ b_dst,
                                          const float* __restrict__ a_src, const float* __restrict__ b_src,
                                          f32x16 (&acc)[BM / 64][BN / 64], const RowRegs<BM / 32>& rr, int step, int nsteps,
                                          int s_begin, int kchunks, int n0, int wave, int lane, int wm, int wn) {
    constexpr int LPS = BM / 32 + BN / 32;
    // loads of `step` have landed for this wave (those of step+1 may still fly) ...
    if (step + 1 < nsteps) wait_vmcnt<LPS>(); else wait_vmcnt<0>();
    // ... and, past the barrier, for every wave; every wave is also done reading stage (step-1)%3
    __builtin_amdgcn_s_barrier();
    if (step + 2 < nsteps) issue_stage<BM, BN, WT>(p, a_dst, b_dst, rr, s_begin + step + 2, kchunks, n0, wave, lane);
    compute_stage<BM, BN, WT>(a_src, b_src, acc, wm, wn, lane >> 5, lane & 31);
}

template <int BM, int BN, bool WT>
__global__ __launch_bounds__(256) void igemm2_kernel(const IgemmArgs p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int LA = BM / 32;                 // A glds per wave per step
    constexpr int A_STAGE = BM * BK;            // floats
    constexpr int B_STAGE = BK * BN;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + STAGES * A_STAGE;
    long long* rowOut = reinterpret_cast<long long*>(Bs + STAGES * B_STAGE);
    float* sred = reinterpret_cast<float*>(rowOut + BM);      // [2][BN][2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const long long M = (long long)p.N * p.Gd * p.Gh * p.Gw;
    const int NT = (p.Nc + BN - 1) / BN;
    const int nt = blockIdx.x % NT;
    const long long m0 = (long long)(blockIdx.x / NT) * BM;
    const int n0 = nt * BN;

    auto decode = [&](long long m, int& n, int& gd, int& gh, int& gw) {
        gw = (int)(m % p.Gw); long long t = m / p.Gw;
        gh = (int)(t % p.Gh); t /= p.Gh;
        gd = (int)(t % p.Gd); n = (int)(t / p.Gd);
    };
    for (int r = tid; r < BM; r += 256) {
        const long long m = m0 + r;
        long long ro = -1;
        if (m < M) {
            int n, gd, gh, gw;
            decode(m, n, gd, gh, gw);
            const int od = gd * p.osd + p.ood, oh = gh * p.osh + p.ooh, ow = gw * p.osw + p.oow;
            ro = ((((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * p.ldy;
        }
        rowOut[r] = ro;
    }
    RowRegs<LA> rr;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int r = (i * 4 + wave) * 8 + (lane >> 3);
        const long long m = m0 + r;
        rr.base[i] = -1; rr.dhw[i] = 0;
        if (m < M) {
            int n, gd, gh, gw;
            decode(m, n, gd, gh, gw);
            rr.base[i] = n * p.Di * p.Hi * p.Wi;
            rr.dhw[i] = ((gd * p.isd) << 20) | ((gh * p.ish) << 10) | (gw * p.isw);
        }
    }

    // ---- this block's slice of the (tap, k-chunk) steps --------------------------------------------
    const int kchunks = (p.K + BK - 1) / BK;
    const int total_steps = p.ntaps * kchunks;
    const int nsplit = gridDim.y;
    const int per = (total_steps + nsplit - 1) / nsplit;
    const int s_begin = blockIdx.y * per;
    const int s_end = min(total_steps, s_begin + per);
    const int nsteps = s_end - s_begin;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float* A0 = As; float* A1 = As + A_STAGE; float* A2 = As + 2 * A_STAGE;
    float* B0 = Bs; float* B1 = Bs + B_STAGE; float* B2 = Bs + 2 * B_STAGE;
    if (nsteps > 0) issue_stage<BM, BN, WT>(p, A0, B0, rr, s_begin, kchunks, n0, wave, lane);
    if (nsteps > 1) issue_stage<BM, BN, WT>(p, A1, B1, rr, s_begin + 1, kchunks, n0, wave, lane);
    for (int base = 0; base < nsteps; base += STAGES) {
        pipe_step<BM, BN, WT>(p, A2, B2, A0, B0, acc, rr, base, nsteps, s_begin, kchunks, n0, wave, lane, wm, wn);
        if (base + 1 < nsteps)
            pipe_step<BM, BN, WT>(p, A0, B0, A1, B1, acc, rr, base + 1, nsteps, s_begin, kchunks, n0, wave, lane, wm, wn);
        if (base + 2 < nsteps)
            pipe_step<BM, BN, WT>(p, A1, B1, A2, B2, acc, rr, base + 2, nsteps, s_begin, kchunks, n0, wave, lane, wm, wn);
    }
    __syncthreads();      // rowOut written above is read below (also when nsteps == 0)

    // ---- epilogue ----------------------------------------------------------------------------------
    const bool split = nsplit > 1;
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + l31;
        const bool cok = col < p.Nc;
        const float bv = (p.bias && cok && blockIdx.y == 0) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const long long ro = rowOut[r];
                if (ro >= 0 && cok) {
                    float v = acc[i][j][e] + bv;
                    float* dst = p.y + ro + col;
                    if (split) {
                        unsafeAtomicAdd(dst, v);
                    } else {
                        if (p.accum) v += *dst;
                        *dst = v;
                        s1[j] += v;
                        s2[j] += v * v;
                    }
                }
            }
        }
    }
    if (p.stats && !split) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
            if (h == 0) {
                sred[(wm * BN + wn * (BN / 2) + j * 32 + l31) * 2 + 0] = s1[j];
                sred[(wm * BN + wn * (BN / 2) + j * 32 + l31) * 2 + 1] = s2[j];
            }
        }
        __syncthreads();
        if (tid < BN && (n0 + tid) < p.Nc) {
            double* st = p.stats + (size_t)(blockIdx.x % P3D_STAT_REPLICAS) * 2 * p.Nc;
            unsafeAtomicAdd(&st[2 * (n0 + tid) + 0], (double)(sred[tid * 2] + sred[(BN + tid) * 2]));
            unsafeAtomicAdd(&st[2 * (n0 + tid) + 1], (double)(sred[tid * 2 + 1] + sred[(BN + tid) * 2 + 1]));
        }
    }
}

template <int BM, int BN>
constexpr size_t smem_bytes() {
    return (size_t)STAGES * (BM * BK + BK * BN) * 4 + BM * 8 + 2 * BN * 2 * 4;
}

template <int BM, int BN>
hipError_t launch_t(const IgemmArgs& a, int splits, hipStream_t s) {
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const long long tiles = ((M + BM - 1) / BM) * ((a.Nc + BN - 1) / BN);
    dim3 grid((unsigned)tiles, (unsigned)splits);
    constexpr size_t sm = smem_bytes<BM, BN>();
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)igemm2_kernel<BM, BN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipFuncSetAttribute((const void*)igemm2_kernel<BM, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        attr_done = true;
    }
    if (a.wT) hipLaunchKernelGGL((igemm2_kernel<BM, BN, true>), grid, dim3(256), sm, s, a);
    else      hipLaunchKernelGGL((igemm2_kernel<BM, BN, false>), grid, dim3(256), sm, s, a);
    return hipGetLastError();
}

}  // namespace

// Tile / split choice.  Prefer the biggest tile (least LDS traffic per FLOP) that still yields
// enough blocks; then slice K until ~2 blocks per CU exist.  Splitting needs a zeroed output and
// cannot carry the statistics epilogue or accumulate mode, so the caller must allow it.
P3dIgemmPlan p3d_igemm2_plan(const IgemmArgs& a, int allow_split) {
    P3dIgemmPlan pl;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const int kchunks = (a.K + BK - 1) / BK;
    const int steps = a.ntaps * kchunks;
    auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((a.Nc + bn - 1) / bn); };
    const long long want = 384;
    if (a.Nc > 64 && tiles(128, 128) >= want) { pl.bm = 128; pl.bn = 128; }
    else if (tiles(128, 64) >= want || (a.Nc <= 64 && tiles(128, 64) >= 128)) { pl.bm = 128; pl.bn = 64; }
    else { pl.bm = 64; pl.bn = 64; }
    pl.splits = 1;
    if (allow_split && steps >= 4) {
        const long long t = tiles(pl.bm, pl.bn);
        if (t < want) {
            long long s = (want + t - 1) / t;
            const long long smax = steps / 2;          // at least 2 steps per slice
            if (s > smax) s = smax;
            if (s < 1) s = 1;
            pl.splits = (int)s;
        }
    }
    pl.name = pl.bm == 128 ? (pl.bn == 128 ? "igemm2_kernel<128,128>" : "igemm2_kernel<128,64>") : "igemm2_kernel<64,64>";
    return pl;
}

hipError_t p3d_launch_igemm2(const IgemmArgs& a, const P3dIgemmPlan& pl, hipStream_t s) {
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (M <= 0 || a.Nc <= 0) return hipSuccess;
    if (a.ntaps > P3D_MAX_TAPS || a.stem_wfloats) return hipErrorInvalidValue;
    if ((a.K & 3) || (a.ldx & 3) || !a.zeros) return hipErrorInvalidValue;
    if (!a.wT && (a.Nc & 3)) return hipErrorInvalidValue;
    if (pl.splits > 1 && (a.accum || a.stats)) return hipErrorInvalidValue;
    if (pl.bm == 128 && pl.bn == 128) return launch_t<128, 128>(a, pl.splits, s);
    if (pl.bm == 128 && pl.bn == 64) return launch_t<128, 64>(a, pl.splits, s);
    return launch_t<64, 64>(a, pl.splits, s);
}
