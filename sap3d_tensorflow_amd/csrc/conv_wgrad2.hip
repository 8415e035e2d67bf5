// Pipelined weight-gradient kernel for gfx950 (see conv_wgrad.hip for the math and the reference
// call sites; this is the same contraction,  dW[tap][k][n] += sum_m Xg[m+tap][k] * dY[m][n],
// with the machinery of conv_igemm2.hip): both operands are [position][channel] rows exactly as
// they lie in NDHWC memory, streamed global -> LDS by LDS-DMA into a 3-stage ring (one raw
// s_barrier + one counted vmcnt per 32-position step), consumed by v_mfma_f32_32x32x2_f32 with
// k = position.  Padded / out-of-range rows come from a zero page.  The position range is split
// over gridDim.y; partial tiles are added with fp32 atomics (dW is zeroed once per step).
#include "p3d_kernels.h"
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BKM = 32;
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

// Per-lane loader state in registers: the lattice coordinates of the rows this lane fetches, advanced
// by 32 positions per step with small-integer reciprocal carries (no per-step division).
template <int LA, int LB>
struct WState {
    unsigned m[LA];                 // position index of A row i (this step)
    int gw[LA], gh[LA], gd[LA], n[LA];
    int kc[LA];                     // channel offset of this lane's 16-byte chunk
    const float* bptr[LB];          // dY row pointer (+ chunk), advanced by 32 rows per step
    unsigned bm[LB];
    bool bok[LB];
    unsigned rGw, rGh, rGd;         // ceil(2^16 / extent)
};

template <int BM, int BN>
__device__ __forceinline__ void wloader_init(const WgradArgs& p, WState<BM / 32, BN / 32>& st, unsigned ms, int k0, int n0,
                                             int wave, int lane) {
    constexpr int LA = BM / 32, LB = BN / 32;
    constexpr int A_LPR = BM / 4, A_RPP = 64 / A_LPR, B_LPR = BN / 4, B_RPP = 64 / B_LPR;
    st.rGw = 65536u / (unsigned)p.Gw + 1; st.rGh = 65536u / (unsigned)p.Gh + 1; st.rGd = 65536u / (unsigned)p.Gd + 1;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const unsigned m = ms + (i * 4 + wave) * A_RPP + lane / A_LPR;
        st.m[i] = m;
        st.gw[i] = (int)(m % (unsigned)p.Gw); unsigned t = m / (unsigned)p.Gw;
        st.gh[i] = (int)(t % (unsigned)p.Gh); t /= (unsigned)p.Gh;
        st.gd[i] = (int)(t % (unsigned)p.Gd); st.n[i] = (int)(t / (unsigned)p.Gd);
        st.kc[i] = k0 + (lane % A_LPR) * 4;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const unsigned m = ms + (i * 4 + wave) * B_RPP + lane / B_LPR;
        const int nc = n0 + (lane % B_LPR) * 4;
        st.bm[i] = m; st.bok[i] = nc < p.Nc;
        st.bptr[i] = p.dy + (long long)m * p.ldy + nc;
    }
}

// Always LA + LB loads (rows past the slice end, padded rows and channel tails read the zero page).
template <int BM, int BN>
__device__ __forceinline__ void issue_stage(const WgradArgs& p, const P3dTap tap, float* __restrict__ a_dst,
                                            float* __restrict__ b_dst, WState<BM / 32, BN / 32>& st, unsigned me, int wave,
                                            int lane) {
    constexpr int LA = BM / 32, LB = BN / 32;
    const float* zp = p.zeros + 4 * (lane & 7);
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int id = st.gd[i] * p.isd + tap.dd, ih = st.gh[i] * p.ish + tap.dh, iw = st.gw[i] * p.isw + tap.dw;
        const bool ok = st.m[i] < me && st.kc[i] < p.K && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                        (unsigned)iw < (unsigned)p.Wi;
        const float* src = ok ? p.x + ((((long long)st.n[i] * p.Di + id) * p.Hi + ih) * p.Wi + iw) * p.ldx + st.kc[i] : zp;
        glds16(src, a_dst + (i * 4 + wave) * 256);
        // advance 32 positions
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
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const bool ok = st.bm[i] < me && st.bok[i];
        glds16(ok ? st.bptr[i] : zp, b_dst + (i * 4 + wave) * 256);
        st.bm[i] += BKM;
        st.bptr[i] += (long long)BKM * p.ldy;
    }
}

// rows [K0, K1) of one stage: consumed in two halves so that the next refill's address arithmetic and DMA issue run
// while the first half's MFMAs execute (same arrangement as conv_igemm2.hip's pipe_step)
template <int BM, int BN, int K0 = 0, int K1 = BKM>
__device__ __forceinline__ void compute_stage(const float* __restrict__ a_st, const float* __restrict__ b_st,
                                              f32x16 (&acc)[BM / 64][BN / 64], float& bsum, bool do_bias, int wm, int wn,
                                              int h, int l31) {
    constexpr int TM = BM / 64, TN = BN / 64;
#pragma unroll
    for (int k = K0; k < K1; k += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = a_st[(k + h) * BM + wm * (BM / 2) + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = b_st[(k + h) * BN + wn * (BN / 2) + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (do_bias && threadIdx.x < BN) {
#pragma unroll 8
        for (int k = K0; k < K1; ++k) bsum += b_st[k * BN + threadIdx.x];
    }
}

template <int BM, int BN>
__device__ __forceinline__ void pipe_step(const WgradArgs& p, const P3dTap tap, float* __restrict__ a_dst,
                                          float* __restrict__ b_dst, const float* __restrict__ a_src,
                                          const float* __restrict__ b_src, f32x16 (&acc)[BM / 64][BN / 64], float& bsum,
                                          bool do_bias, WState<BM / 32, BN / 32>& st, unsigned me, int wave, int lane, int wm,
                                          int wn) {
    constexpr int LPS = BM / 32 + BN / 32;
    wait_vmcnt<(WRing<BM>::stages - 2) * LPS>();
    __builtin_amdgcn_s_barrier();
    compute_stage<BM, BN, 0, BKM / 2>(a_src, b_src, acc, bsum, do_bias, wm, wn, lane >> 5, lane & 31);
    issue_stage<BM, BN>(p, tap, a_dst, b_dst, st, me, wave, lane);
    compute_stage<BM, BN, BKM / 2, BKM>(a_src, b_src, acc, bsum, do_bias, wm, wn, lane >> 5, lane & 31);
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void wgrad2_kernel(const WgradArgs p) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_STAGE = BKM * BM, B_STAGE = BKM * BN;
    constexpr int STAGES = WRing<BM>::stages;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);
    float* Bs = As + STAGES * A_STAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const long long M = (long long)p.N * p.Gd * p.Gh * p.Gw;
    const int KT = (p.K + BM - 1) / BM, NT = (p.Nc + BN - 1) / BN;
    int b = blockIdx.x;
    const int nt = b % NT; b /= NT;
    const int kt = b % KT;
    const int ti = b / KT;
    const P3dTap tap = p.taps[ti];
    const int k0 = kt * BM, n0 = nt * BN;

    long long chunk = (M + p.ksplit - 1) / p.ksplit;
    chunk = (chunk + BKM - 1) / BKM * BKM;
    const long long ms = (long long)blockIdx.y * chunk;
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
    float bsum = 0.f;

    WState<BM / 32, BN / 32> st;
    wloader_init<BM, BN>(p, st, (unsigned)ms, k0, n0, wave, lane);
    const unsigned meu = (unsigned)me;
    if (STAGES == 3) {
        float* A0 = As; float* A1 = As + A_STAGE; float* A2 = As + 2 * A_STAGE;
        float* B0 = Bs; float* B1 = Bs + B_STAGE; float* B2 = Bs + 2 * B_STAGE;
        issue_stage<BM, BN>(p, tap, A0, B0, st, meu, wave, lane);
        issue_stage<BM, BN>(p, tap, A1, B1, st, meu, wave, lane);
        for (int base = 0; base < nsteps; base += 3) {
            pipe_step<BM, BN>(p, tap, A2, B2, A0, B0, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
            if (base + 1 < nsteps) pipe_step<BM, BN>(p, tap, A0, B0, A1, B1, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
            if (base + 2 < nsteps) pipe_step<BM, BN>(p, tap, A1, B1, A2, B2, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
        }
    } else {
        float* A0 = As; float* A1 = As + A_STAGE;
        float* B0 = Bs; float* B1 = Bs + B_STAGE;
        issue_stage<BM, BN>(p, tap, A0, B0, st, meu, wave, lane);
        for (int base = 0; base < nsteps; base += 2) {
            pipe_step<BM, BN>(p, tap, A1, B1, A0, B0, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
            if (base + 1 < nsteps) pipe_step<BM, BN>(p, tap, A0, B0, A1, B1, acc, bsum, do_bias, st, meu, wave, lane, wm, wn);
        }
    }
    __syncthreads();

    float* dwt = p.dw + (long long)tap.widx * p.K * p.Nc;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = k0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < p.K && col < p.Nc) unsafeAtomicAdd(&dwt[(long long)row * p.Nc + col], acc[i][j][e]);
            }
        }
    if (do_bias && tid < BN && (n0 + tid) < p.Nc) unsafeAtomicAdd(&p.dbias[n0 + tid], bsum);
}

template <int BM, int BN>
hipError_t launch_t(const WgradArgs& a, long long tiles, hipStream_t s) {
    constexpr size_t sm = (size_t)WRing<BM>::stages * BKM * (BM + BN) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute((const void*)wgrad2_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        attr_done = true;
    }
    hipLaunchKernelGGL((wgrad2_kernel<BM, BN>), dim3((unsigned)tiles, (unsigned)a.ksplit), dim3(256), sm, s, a);
    return hipGetLastError();
}

struct WPlan { int tile; long long tiles; int ks; double cost; };
// Pick the tile and the number of position-range splits with a small cost model: blocks run in rounds of
// `slots` (256 CUs x resident blocks per CU), a round lasts (steps per block + fixed overhead) step-times, and a
// 128x128 step is ~3.2x a 64x64 step (4x the MFMAs, better LDS-DMA efficiency).  This avoids e.g. 540 blocks on
// 512 slots (a second round with 28 blocks).
WPlan plan(const WgradArgs& a) {
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const long long steps = (M + BKM - 1) / BKM;
    auto best_for = [&](int T) {
        WPlan w;
        w.tile = T; w.ks = 1; w.cost = 1e300;
        w.tiles = (long long)a.ntaps * ((a.K + T - 1) / T) * ((a.Nc + T - 1) / T);
        const long long slots = 256 * (T == 128 ? 2 : 3);
        const double step_time = T == 128 ? 3.2 : 1.0, overhead = T == 128 ? 8.0 : 8.0;
        const long long kmax = std::max<long long>(1, std::min<long long>(steps / 4, 4096));
        for (long long ks = 1; ks <= kmax; ks = ks < 16 ? ks + 1 : ks + ks / 8) {
            const long long blocks = w.tiles * ks;
            const long long rounds = (blocks + slots - 1) / slots;
            const double per_block = (double)((steps + ks - 1) / ks) + overhead;
            const double cost = rounds * per_block * step_time;
            if (cost < w.cost * 0.98) { w.cost = cost; w.ks = (int)std::min<long long>(ks, 65535); }
        }
        return w;
    };
    const WPlan small = best_for(64);
    if (a.K >= 128 && a.Nc >= 128) {
        const WPlan big = best_for(128);
        if (big.cost <= small.cost) return big;
    }
    return small;
}

}  // namespace

const char* p3d_wgrad2_variant(const WgradArgs& a) { return plan(a).tile == 128 ? "wgrad2_kernel<128,128>" : "wgrad2_kernel<64,64>"; }

hipError_t p3d_launch_wgrad2(const WgradArgs& a0, hipStream_t s) {
    WgradArgs a = a0;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (M <= 0 || a.ntaps <= 0) return hipSuccess;
    if (M >= (1ll << 31) - 4096 || a.Gw > 400 || a.Gh > 400 || a.Gd > 400) return hipErrorInvalidValue;   // reciprocal carries
    if (a.ntaps > P3D_MAX_TAPS || a.stem_wfloats || !a.zeros) return hipErrorInvalidValue;
    if ((a.K & 3) || (a.ldx & 3) || (a.Nc & 3) || (a.ldy & 3)) return hipErrorInvalidValue;
    const WPlan w = plan(a);
    a.ksplit = w.ks;
    return w.tile == 128 ? launch_t<128, 128>(a, w.tiles, s) : launch_t<64, 64>(a, w.tiles, s);
}
