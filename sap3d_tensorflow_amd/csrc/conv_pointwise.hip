// Dense 1x1x1 convolutions over MANY positions for gfx950 (stage 1 of the P3D backbone: 50 176 positions at 8 clips of 16x112x112,
// 64 / 256 channels; 401 408 at 32x224x224; reference p3d.py:86,125,127 -- conv1 / conv3 / the projection -- and their input gradients).
//     Y[M, N] (+)= X[M, K] W (+ bias),   K * N <= 16 384,  N in {64, 128, 256},  K a multiple of 64.
//
// The tiled kernel (conv_igemm2.hip) spends such a launch in prologues and epilogues (a 128x64 tile over K = 64 is two K steps) and
// quantises badly (1568 tiles on 512 slots: four rounds for 3.06 rounds of work): 31-33 us where the matrix pipe needs 10.5 and a
// BatchNorm apply pass over the same bytes ~18.  Here a block lives for the whole launch (two per CU, slabs dealt round-robin): each wave
// keeps the WEIGHT fragments of its own sub-tiles in registers (K / 2 VGPRs per sub-tile), row slabs of X stream through a three-slot
// LDS-DMA ring, and a finished slab leaves straight from the accumulators -- accumulator rows are positions, columns channels, so one
// dword store per register writes two whole 128-byte row pieces per wave-instruction, addressed as (wave-uniform row pointer in SGPRs)
// + (one 32-bit lane offset).  What was tried on the way (EXPERIMENTS.md, round 5): weights in LDS with one block per CU -- an LDS
// staging tile + whole-row 16-byte stores, or 16-byte stores from transposed accumulators (32 B in each of 32 rows per instruction) --
// sat at 34-37 us whatever else changed: four waves per CU cannot issue the output; three blocks per CU equal two.
//
// vmcnt bookkeeping: LDS-DMA pieces and the output stores count together, in issue order.  DMA(t) is issued two steps ahead; between it
// and the top of step t only DMA(t + 1) and the stores of the slabs that ended in steps t - 2 and t - 1 are issued, so the wait that
// makes DMA(t) visible leaves P + NS * (those slab ends) operations in flight (capped at the counter's 63: stricter is still right).
//
// fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32).  BatchNorm statistics partials come out per block (nparts = grid size), summed in a
// fixed order: bit-reproducible.  Geometry contract: IgemmArgs with one tap and identity lattices (p3d_pw_stream_blocks says whether).
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "conv_pointwise.hip is written for gfx950 (LDS-DMA, v_mfma_f32_32x32x2_f32, 512 VGPRs per SIMD lane)"
#endif
#include "p3d_kernels.h"
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int PW_KC = 64;                 // channels of K per ring slot

__device__ __forceinline__ void pw_glds16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void pw_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory"); }   // (six bits; stricter is still right)

template <int N> struct PwShape {
    static constexpr int SR = N >= 128 ? 32 : 64;          // rows per slab
    static constexpr int NT = N / 32;                      // 32-column sub-tiles
    static constexpr int SPW = N >= 128 ? NT / 4 : 1;      // sub-tiles per wave
    static constexpr int P = SR * PW_KC / 256 / 4;         // DMA pieces (1 KiB) per wave and step: 2 (32 rows) or 4 (64 rows)
    static constexpr int NS = 16 * SPW;                    // dword stores per lane (= store instructions per wave) and slab
};

// The products of one step (k chunk KC of the slab) for one wave.  The slab ring is read through a __restrict__ view: inlined next to the
// LDS-DMA of the step after next, hipcc must know that these reads cannot alias the DMA's target, or it drains the DMA
// (s_waitcnt vmcnt(0)) in front of the first read and nothing overlaps (seen in this kernel's first build).
// D = X-fragment x W-fragment: accumulator rows are positions, columns channels.
template <int SR, int SPW, int KG, int KC>
__device__ __forceinline__ void pw_products(const float* __restrict__ a_st, const float4 (&wreg)[SPW][KG], f32x16 (&acc)[SPW], int row, int h) {
    constexpr int NC = PW_KC / 8;
    float4 a[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int kb = c >> 2, cc = c & 3;
        const int slot = (2 * cc + h) ^ ((row >> 1) & 7);
        a[c] = *reinterpret_cast<const float4*>(a_st + kb * SR * 32 + row * 32 + slot * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const float4 x = a[c], w = wreg[s][KC * NC + c];
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, w.x, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, w.y, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, w.z, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, w.w, acc[s], 0, 0, 0);
        }
}
// One step's refill: P pieces per wave, always (steps past the block's share fetch the zero page).
template <int P>
__device__ __forceinline__ void pw_issue(const float* const (&src)[P], float* __restrict__ dst, const int (&off)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) pw_glds16(src[i], dst + off[i]);
}

// A finished slab leaves straight from the accumulators: lane (l31, h) holds channel 32 (nt0 + s) + l31 of rows 8 g + 4 h + j (register
// 4 g + j).  One dword store per register: a wave-instruction writes two whole 128-byte row pieces (+ bias, statistics, optional accumulate).
template <int SPW, bool ACC>
__device__ __forceinline__ void pw_store_slab(const IgemmArgs& p, f32x16 (&acc)[SPW], long long m0, int ch0, int lane_off, bool want_stats, const float (&bias)[SPW],
                                                 float (&st1)[SPW], float (&st2)[SPW]) {
    // address = (wave-uniform row pointer, in SGPRs) + (one 32-bit lane offset): the saddr form of the store, no per-row address registers
    float* __restrict__ urow = p.y + m0 * p.ldy + ch0;
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
        float old[ACC ? 16 : 1];
        if constexpr (ACC) {
            // an input gradient joining another one: the old values first, a sub-tile's sixteen in flight together (this drains the ring's
            // DMA too; the counted waits of the main loop stay valid: they bound what may be in flight, and less is)
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = (urow + (long long)(8 * (r >> 2) + (r & 3)) * p.ldy + 32 * s)[lane_off];
            pw_wait<0>();
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[s][r] + bias[s];
            acc[s][r] = 0.f;
            if (want_stats) { st1[s] += v; st2[s] = fmaf(v, v, st2[s]); }
            if constexpr (ACC) v += old[r];
            (urow + (long long)(8 * (r >> 2) + (r & 3)) * p.ldy + 32 * s)[lane_off] = v;
        }
    }
}

// The K / 64 steps of one slab, k chunk a compile-time constant (the weight fragments sit in a register array).
template <int N, int K, int KC>
struct pw_slab_steps {
    template <class Issue>
    static __device__ __forceinline__ void run(const IgemmArgs& p, float* ring, const float4 (&wreg)[PwShape<N>::SPW][K / 8], f32x16 (&acc)[PwShape<N>::SPW],
                                               Issue& issue_step, int& ends, int& step, int row, int h) {
        using S = PwShape<N>;
        constexpr int KCH = K / PW_KC;
        if constexpr (KC < KCH) {
            // DMA(step) has landed for this wave: leave DMA(step + 1) and the stores issued after DMA(step) in flight (file header)
            if (ends == 0) pw_wait<S::P>();
            else if (ends == 3) pw_wait<S::P + 2 * S::NS>();
            else pw_wait<S::P + S::NS>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // ... for every wave; everyone is done with the slot refilled next
            issue_step(step + 2);
            pw_products<S::SR, S::SPW, K / 8, KC>(ring + (size_t)(step % 3) * S::SR * PW_KC, wreg, acc, row, h);
            ends = ((ends << 1) & 2) | (KC == KCH - 1 ? 1 : 0);
            ++step;
            pw_slab_steps<N, K, KC + 1>::run(p, ring, wreg, acc, issue_step, ends, step, row, h);
        }
    }
};

// N: output channels, K: input channels.  WT: weights [N][K] (input gradients) instead of [K][N].  ACC: Y += (an instantiation of its own:
// with the accumulate loads behind a run-time branch hipcc drained the vector counter at the join on BOTH paths).
template <int N, int K, bool WT, bool ACC>
__global__ __launch_bounds__(256, 2) void pw_stream_kernel(const IgemmArgs p, const int nslabs) {
    P3D_CHAIN_PRIO();
    p3d_warm_kernargs<IgemmArgs>();
    using S = PwShape<N>;
    constexpr int SR = S::SR, SPW = S::SPW, P = S::P, NS = S::NS;
    constexpr int KG = K / 8, KCH = K / PW_KC;                  // 8-wide k groups; 64-wide k chunks (= steps per slab)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ring = reinterpret_cast<float*>(smem);             // [3][2 kb][SR rows][32]: chunk-swizzled like conv_igemm2's A image
    float* sred = ring + 3 * SR * PW_KC;                      // [2 row halves][N][2]: statistics exchange at the end

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const long long M = (long long)p.N * p.Gd * p.Gh * p.Gw;
    // MFMA roles: N >= 128: every wave multiplies all SR = 32 rows by its SPW sub-tiles; N == 64: wave -> (row half, sub-tile)
    const int rt = N >= 128 ? 0 : (wave >> 1);
    const int nt0 = N >= 128 ? wave * SPW : (wave & 1);

    // ---- slab loader: piece (kb, rg) = rows 8 rg .. 8 rg + 7 of 32-channel block kb; wave w issues pieces w, w + 4, ... ----------------
    const int a_slot = lane & 7, a_sub = lane >> 3;
    auto issue_step = [&](int step) {
        // step -> (slab of this block, k chunk); steps past the block's share fetch the zero page (same number of loads)
        const int sl = step / KCH, kc = step - sl * KCH;
        const long long slab = (long long)blockIdx.x + (long long)sl * gridDim.x;
        const float* src[P];
        int off[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int piece = wave + 4 * i;                   // 0 .. SR / 4 - 1
            const int kb = piece / (SR / 8), rg = piece - kb * (SR / 8);
            const int r = rg * 8 + a_sub;
            const long long m = slab * SR + r;
            const bool ok = slab < nslabs && m < M;
            src[i] = ok ? p.x + m * p.ldx + kc * PW_KC + kb * 32 + 4 * (a_slot ^ ((r >> 1) & 7)) : p.zeros + 4 * a_slot;
            off[i] = kb * SR * 32 + rg * 256;
        }
        pw_issue<P>(src, ring + (size_t)(step % 3) * SR * PW_KC, off);
    };
    const int my_slabs = nslabs > (int)blockIdx.x ? (nslabs - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    issue_step(0);
    issue_step(1);

    // ---- this wave's weight fragments -> registers: fragment (s, c) of lane (n = l31, h) is W[8c + 4h + j][32 (nt0 + s) + n], j = 0..3 ----
    float4 wreg[SPW][KG];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int c = 0; c < KG; ++c) {
            const int n = 32 * (nt0 + s) + l31, k = 8 * c + 4 * h;
            if (WT) wreg[s][c] = *reinterpret_cast<const float4*>(p.w + (long long)n * K + k);
            else { const float* q = p.w + (long long)k * N + n; wreg[s][c] = make_float4(q[0], q[N], q[2 * N], q[3 * N]); }
        }
    // output roles (pw_store_slab): this lane's row within a slab and its first channel; its statistics sums
    const bool want_stats = p.statpart != nullptr;
    float st1[SPW], st2[SPW], bias[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) { st1[s] = 0.f; st2[s] = 0.f; bias[s] = p.bias ? p.bias[32 * (nt0 + s) + l31] : 0.f; }

    f32x16 acc[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
    int ends = 0;        // bit 0: the previous step ended a slab (its stores are in flight), bit 1: the step before it did
    int step = 0;
    for (int sl = 0; sl < my_slabs; ++sl) {
        pw_slab_steps<N, K, 0>::run(p, ring, wreg, acc, issue_step, ends, step, rt * 32 + l31, h);
        // ---- the slab leaves, straight from the accumulators (no staging tile, no second barrier) --------------------------------------
        const long long slab = (long long)blockIdx.x + (long long)sl * gridDim.x;
        pw_store_slab<SPW, ACC>(p, acc, slab * SR + rt * 32, 32 * nt0, 4 * h * p.ldy + l31, want_stats, bias, st1, st2);
    }
    pw_wait<0>();              // the ring's last (zero-page) refills and the last slab's stores
    if (want_stats) {
        // per block and channel: (sum, sumsq) over the block's rows, in a fixed order.  A lane's channel is fixed, its rows are 8 g + 4 h + j of
        // every slab: the two halves of the wave meet by one shuffle (h = 0 first); N = 64: the two row halves (waves w, w + 2) in LDS
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const float o1 = __shfl_xor(st1[s], 32), o2 = __shfl_xor(st2[s], 32);
            if (h == 0) {
                float* d = sred + ((size_t)rt * N + 32 * (nt0 + s) + l31) * 2;
                d[0] = st1[s] + o1; d[1] = st2[s] + o2;
            }
        }
        __syncthreads();
        if (tid < N) {
            float t1 = sred[(size_t)tid * 2], t2 = sred[(size_t)tid * 2 + 1];
            if (N == 64) { t1 += sred[((size_t)N + tid) * 2]; t2 += sred[((size_t)N + tid) * 2 + 1]; }
            float* dst = p.statpart + ((size_t)(p.stat_base + (int)blockIdx.x) * p.Nc + tid) * 2;
            dst[0] = t1; dst[1] = t2;
        }
    }
}

struct PwLaunch { size_t lds; int grid; int nslabs; };

size_t pw_lds_bytes(int N) {
    const int SR = N >= 128 ? 32 : 64;
    return ((size_t)3 * SR * PW_KC + 2 * N * 2) * sizeof(float);
}

bool pw_shape(const IgemmArgs& a, PwLaunch& L) {
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (a.ntaps != 1 || a.taps[0].dd || a.taps[0].dh || a.taps[0].dw || a.at_mode != P3D_AT_NONE || a.ngate || a.f16) return false;
    if (a.isd != 1 || a.ish != 1 || a.isw != 1 || a.osd != 1 || a.osh != 1 || a.osw != 1 || a.ood || a.ooh || a.oow) return false;
    if (a.Gd != a.Di || a.Gh != a.Hi || a.Gw != a.Wi || a.Gd != a.Do || a.Gh != a.Ho || a.Gw != a.Wo) return false;
    // K = 64 only: with K = 256 -> 64 channels (128 weight VGPRs, 64-row slabs: 784 slabs on 512 blocks) this kernel measured 27.8 / 30.7 us
    // against the tiled kernel's 28.0 / 27.0 (forward / input gradient)
    if (a.K != 64 || (a.Nc != 64 && a.Nc != 128 && a.Nc != 256)) return false;
    if ((a.ldx & 3) || (a.ldy & 3) || M < 16384 || M >= (1ll << 31) / 256) return false;
    const int SR = a.Nc >= 128 ? 32 : 64;
    if (M % SR) return false;                    // whole slabs only: every wave issues the same number of stores per slab (the counted waits)
    L.lds = pw_lds_bytes(a.Nc);
    L.nslabs = (int)((M + SR - 1) / SR);
    L.grid = std::min(L.nslabs, 512);            // two blocks per CU (three measured the same); the slabs are dealt round-robin
    return true;
}

template <int N, int K, bool WT, bool ACC>
hipError_t pw_launch_k(const IgemmArgs& a, const PwLaunch& L, hipStream_t s) {
    hipLaunchKernelGGL((pw_stream_kernel<N, K, WT, ACC>), dim3((unsigned)L.grid), dim3(256), L.lds, s, a, L.nslabs);      // (< 64 KB of LDS: no attribute)
    return hipGetLastError();
}
template <int N, int K>
hipError_t pw_launch_t(const IgemmArgs& a, const PwLaunch& L, hipStream_t s) {
    if (a.wT) return a.accum ? pw_launch_k<N, K, true, true>(a, L, s) : pw_launch_k<N, K, true, false>(a, L, s);
    return a.accum ? pw_launch_k<N, K, false, true>(a, L, s) : pw_launch_k<N, K, false, false>(a, L, s);
}

}  // namespace

// Is this launch the streaming kernel's case, and with how many blocks (= statistics partials) would it run?  0: no.
int p3d_pw_stream_blocks(const IgemmArgs& a) {
    static const bool off = p3d_tune_env("P3D_PW_STREAM") && atoi(p3d_tune_env("P3D_PW_STREAM")) == 0;      // A/B runs (tuning build)
    PwLaunch L;
    return !off && pw_shape(a, L) ? L.grid : 0;
}

hipError_t p3d_launch_pw_stream(const IgemmArgs& a0, hipStream_t s) {
    IgemmArgs a = a0;
    PwLaunch L;
    if (!pw_shape(a, L) || !a.zeros) return hipErrorInvalidValue;
    switch (a.Nc) {
        case 64: return pw_launch_t<64, 64>(a, L, s);
        case 128: return pw_launch_t<128, 64>(a, L, s);
        case 256: return pw_launch_t<256, 64>(a, L, s);
        default: return hipErrorInvalidValue;
    }
}
