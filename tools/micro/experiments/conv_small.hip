// Convolutions over FEW output positions (stage 3 of the P3D backbone at 8 clips: M = 784 rows, 256..2304-deep reductions;
// reference p3d.py:56-136 conv1 / convS / convT / conv3 and their input gradients) for gfx950.
//
// The pipelined kernel (conv_igemm2.hip) needs K-slices across blocks to fill 256 CUs when a launch has 52 tiles of 64x64,
// and pays for them with a slab exchange (write-through partial tiles, arrival ticket, the last arriver's slab reads:
// 3-5 us of a 10-20 us launch).  Here the unit of work is ONE WAVE's 32x32 output tile:
//
//  * a block is four waves arranged WM x WN x WK: (32 WM) x (32 WN) output tile, the reduction split WK ways across the
//    block's own waves -- 200 blocks for a 784 x 256 output with NO cross-block exchange; the WK partial tiles meet in
//    LDS once, at the end (fixed order: bit-reproducible);
//  * no wave shares an operand with another one, so the LDS round trip is pure overhead: A and B fragments are loaded
//    straight into registers in the v_mfma_f32_32x32x2_f32 layout (lane (i, h) holds k = 8c + 4h + {0..3} of row / column i:
//    one 16-byte load per operand and four MFMAs), through a ring of RD register sets that keeps RD - 1 loads in flight per
//    lane (the compiler's counted s_waitcnt vmcnt does the rest);
//  * rows that fall into SAME padding, rows past M and the tail of the ring read a zero page: every wave issues the same
//    loads, nothing branches inside the ring;
//  * the epilogue is the pipelined kernel's (igemm_epilogue.h): bias, accumulate, BatchNorm statistics partials, gates.
//
// fp32 in / fp32 accumulate (exact fp32 MFMA, as everywhere on this path).  Geometry contract: IgemmArgs (p3d_kernels.h).
#if !defined(__gfx950__) && !defined(__gfx942__) && defined(__HIP_DEVICE_COMPILE__)
#error "written for gfx942 / gfx950"
#endif
#include "p3d_kernels.h"
#include "igemm_epilogue.h"
#include <mutex>
#include <cstdlib>
#include <cstddef>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#if defined(P3D_TUNE_STAMPS)
// tuning builds only (tools/micro): wall-clock stamps (100 MHz) of every block's wave 0 -- entry, ring primed, first data in, loop
// done, partial tiles summed, exit
__device__ unsigned long long p3d_convsm_stamps[1024][8];
#define STAMP(i) do { if (wave == 0 && lane == 0 && blockIdx.x < 1024) p3d_convsm_stamps[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

#if !defined(P3D_SM_RD)
#define P3D_SM_RD 8
#endif
constexpr int RD = P3D_SM_RD;    // register sets in the ring (groups of 8 reduction channels = 4 MFMAs each)

// The launch-invariant scalars of the main loop, read from the kernel arguments ONCE and pinned in SGPRs: left to itself
// hipcc re-reads kernel arguments where they are used (inside the tap-change branches, too) -- eighteen dependent
// s_load / s_waitcnt round trips stood between kernel entry and the first operand load (2.5 us of a 12 us launch).
template <class T>
__device__ __forceinline__ T pin_s(T v) { asm volatile("" : "+s"(v)); return v; }
struct SmArgs {
    const float* x; const float* w; const float* zeros; P3dKTap* taps;
    int Di, Hi, Wi, ldx, K, Nc;
};

template <bool WT>
struct Cursor {
    const float* ap; int astep;          // this lane's A run for the current tap (+ 4h), floats to advance per group (0: zero page)
    const float* bp; int bstep, bs;      // B: WT: row n of the [Nc][K] slab; !WT: column n of the [K][Nc] slab, bs = row stride
    int tap, kc, g;                      // next group to issue: tap, group within the tap, linear index
    int left;                            // groups until the next event (end of the tap / end of this wave's share)
};

template <bool WT>
__device__ __forceinline__ void cursor_set_tap(const SmArgs& q, Cursor<WT>& c, int gpt, int base, int dhw, int n, bool nok, int h, int g_end) {
    const bool live = c.g < g_end;
    const P3dTap tap = p3d_ktap(q.taps, live ? c.tap : 0);
    const int id = (dhw >> 20) + tap.dd, ih = ((dhw >> 10) & 1023) + tap.dh, iw = (dhw & 1023) + tap.dw;
    const bool ok = live && base >= 0 && (unsigned)id < (unsigned)q.Di && (unsigned)ih < (unsigned)q.Hi && (unsigned)iw < (unsigned)q.Wi;
    const long long row = (long long)base + ((long long)id * q.Hi + ih) * q.Wi + iw;
    const int k = c.kc * 8 + 4 * h;
    c.ap = ok ? q.x + row * q.ldx + k : q.zeros + 4 * h;
    c.astep = ok ? 8 : 0;
#if defined(P3D_TUNE_FAKE_A)       // TIMING ONLY, WRONG RESULTS: a lane-contiguous A load (what the scattered row gather costs the loop)
    c.ap = q.x + (threadIdx.x & 63) * 4 + c.kc * 256; c.astep = 256;
#endif
    const float* wt = q.w + (long long)tap.widx * q.K * q.Nc;
    const bool bok = live && nok;
    if (WT) {
        c.bp = bok ? wt + (long long)n * q.K + k : q.zeros + 4 * h;
        c.bstep = bok ? 8 : 0; c.bs = 0;
#if defined(P3D_TUNE_FAKE_A)
        c.bp = wt + (threadIdx.x & 63) * 4 + c.kc * 256; c.bstep = 256;
#endif
    } else {
        c.bp = bok ? wt + (long long)k * q.Nc + n : q.zeros;
        c.bstep = bok ? 8 * q.Nc : 0; c.bs = bok ? q.Nc : 0;
    }
    c.left = live ? min(gpt - c.kc, g_end - c.g) : 0x7fffffff;
}

template <bool WT>
__device__ __forceinline__ void cursor_issue(const SmArgs& q, Cursor<WT>& c, float4& ra, float4& rb, int gpt, int base, int dhw, int n, bool nok,
                                             int h, int g_end) {
    ra = *reinterpret_cast<const float4*>(c.ap);
    if (WT) rb = *reinterpret_cast<const float4*>(c.bp);
    else { rb.x = c.bp[0]; rb.y = c.bp[c.bs]; rb.z = c.bp[2 * c.bs]; rb.w = c.bp[3 * c.bs]; }
    c.ap += c.astep; c.bp += c.bstep;
    ++c.g; ++c.kc;
    if (--c.left == 0) {                 // wave-uniform, once per tap: next tap, or the zero page past the end of the share
        if (c.kc == gpt) { c.kc = 0; ++c.tap; }
        cursor_set_tap<WT>(q, c, gpt, base, dhw, n, nok, h, g_end);
    }
}

template <int WM, int WN, int WK, bool WT>
__global__ __launch_bounds__(256) void convsm_kernel(const IgemmArgs p) {
    static_assert(WM * WN * WK == 4, "four waves per block");
    constexpr int BM = 32 * WM, BN = 32 * WN, LDT = BN + 4;
    P3D_CHAIN_PRIO();
    p3d_warm_kernargs<IgemmArgs>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* rowIdx = reinterpret_cast<int*>(smem);                    // [BM]
    float* tiles = reinterpret_cast<float*>(rowIdx + BM);          // [WK][BM][LDT] partial tiles; tile 0 becomes the result
    float* sred = tiles + WK * BM * LDT;                           // [2][4][BN][2] statistics exchange of the epilogue

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
    const int h = lane >> 5, l31 = lane & 31;
    STAMP(0);

    SmArgs q;
    q.x = pin_s(p.x); q.w = pin_s(p.w); q.zeros = pin_s(p.zeros); q.taps = p3d_kernarg_taps(offsetof(IgemmArgs, taps));
    q.Di = pin_s(p.Di); q.Hi = pin_s(p.Hi); q.Wi = pin_s(p.Wi); q.ldx = pin_s(p.ldx); q.K = pin_s(p.K); q.Nc = pin_s(p.Nc);
    const int pN = pin_s(p.N), Gd = pin_s(p.Gd), Gh = pin_s(p.Gh), Gw = pin_s(p.Gw);
    P3dFastDiv fGd, fGh, fGw;
    fGd.mul = pin_s(p.fGd.mul); fGd.shift = pin_s(p.fGd.shift); fGh.mul = pin_s(p.fGh.mul); fGh.shift = pin_s(p.fGh.shift);
    fGw.mul = pin_s(p.fGw.mul); fGw.shift = pin_s(p.fGw.shift);
    const int isd = pin_s(p.isd), ish = pin_s(p.ish), isw = pin_s(p.isw), ntaps = pin_s(p.ntaps);
    STAMP(6);

    const long long M = (long long)pN * Gd * Gh * Gw;
    const int NT = (q.Nc + BN - 1) / BN;
    const int nt = (int)blockIdx.x % NT, mt = (int)blockIdx.x / NT;
    const unsigned Mu = (unsigned)M, m0u = (unsigned)mt * BM;
    const int n0 = nt * BN;

    // ---- this lane's A row and B column ---------------------------------------------------------------------------------
    int base = -1, dhw = 0;
    {
        const unsigned m = m0u + wm * 32 + l31;
        if (m < Mu) {
            const unsigned t1 = p3d_div(m, fGw), gw = m - t1 * (unsigned)Gw;
            const unsigned t2 = p3d_div(t1, fGh), gh = t1 - t2 * (unsigned)Gh;
            const unsigned n = p3d_div(t2, fGd), gd = t2 - n * (unsigned)Gd;
            base = (int)n * q.Di * q.Hi * q.Wi;
            dhw = (int)(((gd * isd) << 20) | ((gh * ish) << 10) | (gw * isw));
        }
    }
    const int ncol = n0 + wn * 32 + l31;
    const bool nok = ncol < q.Nc;
    const float4 bias4 = igemm_bias_prefetch<BN>(p.bias, n0, q.Nc);      // for the epilogue; in flight behind the whole main loop

    // ---- this wave's share of the (tap, 8-channel group) reduction ---------------------------------------------------------
    const int gpt = q.K >> 3;                                   // groups per tap (launcher: K % 8 == 0)
    const int total = ntaps * gpt;
    const int per = (total + WK - 1) / WK;
    const int g_begin = min(total, wk * per), g_end = min(total, g_begin + per);

    Cursor<WT> cur;
    cur.g = g_begin; cur.tap = g_begin / gpt; cur.kc = g_begin - cur.tap * gpt;
    cursor_set_tap<WT>(q, cur, gpt, base, dhw, ncol, nok, h, g_end);
    STAMP(7);

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;

    float4 ra[RD], rb[RD];
#pragma unroll
    for (int i = 0; i < RD; ++i) cursor_issue<WT>(q, cur, ra[i], rb[i], gpt, base, dhw, ncol, nok, h, g_end);
    // the output-row table of the epilogue, behind the first loads (nobody reads it before the barrier after the loop)
    for (int r = tid; r < BM; r += 256) {
        const unsigned m = m0u + r;
        int ro = -1;
        if (m < Mu) {
            const unsigned t1 = p3d_div(m, fGw), gw = m - t1 * (unsigned)Gw;
            const unsigned t2 = p3d_div(t1, fGh), gh = t1 - t2 * (unsigned)Gh;
            const unsigned n = p3d_div(t2, fGd), gd = t2 - n * (unsigned)Gd;
            const int od = gd * p.osd + p.ood, oh = gh * p.osh + p.ooh, ow = gw * p.osw + p.oow;
            ro = (((int)n * p.Do + od) * p.Ho + oh) * p.Wo + ow;
        }
        rowIdx[r] = ro;
    }
    STAMP(1);
#if defined(P3D_TUNE_STAMPS)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(2);
#endif
    for (int g = g_begin; g < g_end; g += RD) {
#pragma unroll
        for (int i = 0; i < RD; ++i) {
            const float4 a = ra[i], b = rb[i];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            cursor_issue<WT>(q, cur, ra[i], rb[i], gpt, base, dhw, ncol, nok, h, g_end);      // refill this set RD groups ahead
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    STAMP(3);
    // ---- partial tiles -> LDS, summed over the WK reduction shares in a fixed order ----------------------------------------
    {
        float* t = tiles + wk * BM * LDT;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            t[r * LDT + wn * 32 + l31] = acc[e];
        }
    }
    __syncthreads();
    if (WK > 1) {
        constexpr int F4R = BN / 4;
        for (int i = tid; i < BM * F4R; i += 256) {
            const int r = i / F4R, c4 = (i - r * F4R) * 4;
            float4 v = *reinterpret_cast<const float4*>(tiles + r * LDT + c4);
#pragma unroll
            for (int q = 1; q < WK; ++q) {
                const float4 u = *reinterpret_cast<const float4*>(tiles + q * BM * LDT + r * LDT + c4);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            *reinterpret_cast<float4*>(tiles + r * LDT + c4) = v;
        }
        __syncthreads();
    }
    STAMP(4);
    Geo geo;
    geo.Gd = p.Gd; geo.Gh = p.Gh; geo.Gw = p.Gw; geo.fGd = p.fGd; geo.fGh = p.fGh; geo.fGw = p.fGw;
    geo.ood = p.ood; geo.ooh = p.ooh; geo.oow = p.oow; geo.stat_base = p.stat_base; geo.ntaps = p.ntaps; geo.taps = q.taps;
    geo.w = p.w; geo.bias = p.bias; geo.y = p.y; geo.statpart = p.statpart;
    geo.nsplit = 1; geo.slab = nullptr; geo.cnt = nullptr;
    igemm_tile_epilogue<BM, BN>(p, geo, tiles, sred, rowIdx, mt, n0, M, bias4);
    STAMP(5);
}

template <int WM, int WN, int WK>
hipError_t launch_sm(const IgemmArgs& a, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN;
    constexpr size_t sm = (size_t)BM * 4 + (size_t)WK * BM * (BN + 4) * 4 + (size_t)2 * 4 * BN * 2 * 4 + 16;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const long long tiles = ((M + BM - 1) / BM) * ((a.Nc + BN - 1) / BN);
    const dim3 grid((unsigned)tiles), blk(256);
    // One block per CU: four waves own the four matrix pipes.  The request is sized so that a second block of this kernel does
    // not fit beside the first (LDS is the only resource a launch can be given more of), and beside an 82 KB filter-gradient
    // block of the side stream there is still room for one.
    static const size_t lds = [] { const char* e = p3d_tune_env("P3D_TUNE_SM_LDS_KB"); return (size_t)(e ? atoi(e) : 72) * 1024; }();
    static std::once_flag once;
    std::call_once(once, [] {
        hipFuncSetAttribute((const void*)convsm_kernel<WM, WN, WK, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)convsm_kernel<WM, WN, WK, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    const size_t req = lds > sm ? lds : sm;
    if (a.wT) hipLaunchKernelGGL((convsm_kernel<WM, WN, WK, true>), grid, blk, req, s, a);
    else hipLaunchKernelGGL((convsm_kernel<WM, WN, WK, false>), grid, blk, req, s, a);
    return hipGetLastError();
}

}  // namespace

#if defined(P3D_TUNE_STAMPS)
hipError_t p3d_convsm_read_stamps(unsigned long long* host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(p3d_convsm_stamps), sizeof(p3d_convsm_stamps)); }
#endif

// Which arrangement of the four waves: 0 none (the pipelined kernel keeps the launch), 1: 32x32 tile, reduction split four ways;
// 2: 64x64 tile, every wave the whole reduction; 3: 32x64 tile, reduction split two ways.
int p3d_convsm_shape(const IgemmArgs& a) {
    if (a.at_mode != P3D_AT_NONE || a.f16 || a.eb.mode || (a.K & 7) || (a.Nc & 31)) return 0;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (M > 1024 || M < 256) return 0;
    const long long groups = (long long)a.ntaps * (a.K >> 3);
    const long long t32 = ((M + 31) / 32) * (a.Nc / 32), t64 = ((M + 63) / 64) * ((a.Nc + 63) / 64);
    if (t64 >= 160 && t64 <= 320) return 2;                     // conv3 / conv1's input gradient: 208 tiles of 64x64, K = 256
    if (t32 <= 256 && groups >= 4 * RD) return 1;               // 200 tiles of 32x32
    if (t32 <= 512 && groups >= 2 * RD) return 3;
    return 0;
}

hipError_t p3d_launch_convsm(const IgemmArgs& a0, int shape, hipStream_t s) {
    IgemmArgs a = a0;
    a.fGd = p3d_fastdiv((unsigned)a.Gd); a.fGh = p3d_fastdiv((unsigned)a.Gh); a.fGw = p3d_fastdiv((unsigned)a.Gw);
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (M <= 0 || a.Nc <= 0) return hipSuccess;
    if (M >= (1ll << 31) || (long long)a.N * a.Di * a.Hi * a.Wi >= (1ll << 31) || (long long)a.N * a.Do * a.Ho * a.Wo >= (1ll << 31)) return hipErrorInvalidValue;
    if (a.Gd * a.isd >= 1024 || a.Gh * a.ish >= 1024 || a.Gw * a.isw >= 1024 || a.ntaps > P3D_MAX_TAPS) return hipErrorInvalidValue;
    if ((a.K & 7) || (a.ldx & 3) || !a.zeros || (a.Nc & 3) || (a.ldy & 3) || a.at_mode != P3D_AT_NONE || a.f16 || a.eb.mode) return hipErrorInvalidValue;
    if (a.ngate < 0 || a.ngate > 2 || (a.ngate && a.statpart)) return hipErrorInvalidValue;
    a.nsplit = 1; a.slab = nullptr; a.cnt = nullptr;
    switch (shape) {
        case 1: return launch_sm<1, 1, 4>(a, s);
        case 2: return launch_sm<2, 2, 1>(a, s);
        case 3: return launch_sm<1, 2, 2>(a, s);
        default: return hipErrorInvalidValue;
    }
}
