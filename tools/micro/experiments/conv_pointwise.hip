// Dense 1x1x1 convolutions over MANY positions (stage 1 of the P3D backbone: 50 176 positions at 8 clips of 16x112x112, 64 / 256
// channels; 401 408 at 32x224x224; reference p3d.py:86,125,127 conv1 / conv3 / the projection, and their input gradients) for
// gfx950.  Y[M, N] (+)= X[M, K] W (+ bias): 13-45 FLOP per byte, HBM-bound on this chip.
//
// The tiled kernel (conv_igemm2.hip) spends such a launch in prologues and epilogues: a 128x64 tile over K = 64 is two K steps.
// Here the WEIGHTS stay in LDS for the life of a block (K*N <= 16 384 floats, laid out in MFMA fragment order: conflict-free
// 16-byte reads) and the block streams 32-row slabs of X through a two-slot LDS ring by LDS-DMA: the slab after the one being
// multiplied is already in flight, the slab before it is being stored from the accumulators (row segments of 128 bytes), and
// two such blocks share a CU when LDS allows.  Four waves: N / 32 output sub-tiles spread over them (N = 64: two sub-tiles, the
// reduction split in halves and met in LDS).  Statistics partials for BatchNorm come out per block (nparts = grid size), sums in a
// fixed order: bit-reproducible.
//
// fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32).  Geometry contract: IgemmArgs with one tap and identity lattices.
#if !defined(__gfx950__) && !defined(__gfx942__) && defined(__HIP_DEVICE_COMPILE__)
#error "written for gfx942 / gfx950"
#endif
#include "p3d_kernels.h"
#include <mutex>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int PW_BM = 32;                 // rows per slab

__device__ __forceinline__ void pw_glds16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The products of one slab for one wave.  Its own function with __restrict__ views of the slab slot and the weight image: inlined next
// to the LDS-DMA of the NEXT slab, hipcc must know that these reads cannot alias the DMA's target, or it drains the DMA
// (s_waitcnt vmcnt(0)) in front of the first read and nothing overlaps.
template <int SPW>
__device__ __forceinline__ void pw_products(const float* __restrict__ a_st, const float* __restrict__ wl, f32x16 (&acc)[SPW], int c_begin, int c_end,
                                            int KG, int nt0, int lane) {
    const int h = lane >> 5, l31 = lane & 31;
    for (int c = c_begin; c < c_end; ++c) {
        const int kb = c >> 2, cc = c & 3;
        const int slot = (2 * cc + h) ^ ((l31 >> 1) & 7);
        const float4 a = *reinterpret_cast<const float4*>(a_st + kb * 1024 + l31 * 32 + slot * 4);
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(wl + ((size_t)((nt0 + s) * KG + c) * 64 + lane) * 4);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[s], 0, 0, 0);
            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[s], 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void pw_issue_slab(const float* src, bool ok, float* __restrict__ dst, int KB) {
    for (int kb = 0; kb < KB; ++kb) pw_glds16(ok ? src + kb * 32 : src, dst + kb * 1024);
}

// NT = N / 32 output sub-tiles (2, 4 or 8).  Waves: NT >= 4: wave w owns sub-tiles w*NT/4 .. ; NT == 2: wave w owns sub-tile w & 1 and
// the K half w >> 1.
template <int NT, bool WT>
__global__ __launch_bounds__(256) void pw_stream_kernel(const IgemmArgs p, const int nslabs) {
    P3D_CHAIN_PRIO();
    p3d_warm_kernargs<IgemmArgs>();
    constexpr int N = 32 * NT;
    constexpr int SPW = NT >= 4 ? NT / 4 : 1;          // sub-tiles per wave
    constexpr bool KSPLIT = NT == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int K = p.K, KB = K >> 5, KG = K >> 3;        // 32-wide k blocks, 8-wide k groups
    float* wl = reinterpret_cast<float*>(smem);         // [NT][KG][64 lanes][4]: B fragments as the MFMA wants them
    float* al = wl + (size_t)K * N;                     // [2][KB][32 rows][32]: slab ring, chunk-swizzled like conv_igemm2's
    float* red = al + 2 * (size_t)KB * 1024;            // KSPLIT: [2 sub-tiles][32][36] partial tiles of the upper K half

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, l31 = lane & 31;
    const long long M = (long long)p.N * p.Gd * p.Gh * p.Gw;

    // ---- slab loader: wave w fetches rows 8w .. 8w+7 of every k block (one 1 KB instruction each) ------------------------------
    const int a_slot = lane & 7, a_sub = lane >> 3;
    const int arow = wave * 8 + a_sub;
    const int achunk = 4 * (a_slot ^ ((arow >> 1) & 7));
    auto issue_slab = [&](int slab, int buf) {
        const long long m = (long long)slab * PW_BM + arow;
        const bool ok = slab < nslabs && m < M;
        const float* src = ok ? p.x + m * p.ldx + achunk : p.zeros + 4 * (lane & 7);
        pw_issue_slab(src, ok, al + (size_t)buf * KB * 1024 + wave * 8 * 32, KB);
    };
    int slab = (int)blockIdx.x;
    issue_slab(slab, 0);

    // ---- weights -> LDS in fragment order (once per block) -------------------------------------------------------------------------
    // fragment (nt, c, lane = (n, hh)) holds W[8c + 4hh + j][32 nt + n], j = 0..3
    for (int i = tid; i < NT * KG * 64; i += 256) {
        const int ln = i & 63, c = (i >> 6) % KG, nt = (i >> 6) / KG;
        const int n = 32 * nt + (ln & 31), k = 8 * c + 4 * (ln >> 5);
        float4 v;
        if (WT) v = *reinterpret_cast<const float4*>(p.w + (long long)n * K + k);
        else { const float* q = p.w + (long long)k * N + n; v = make_float4(q[0], q[N], q[2 * N], q[3 * N]); }
        *reinterpret_cast<float4*>(wl + (size_t)i * 4) = v;
    }
    // this wave's share
    const int nt0 = KSPLIT ? (wave & 1) : wave * SPW;
    const int c_begin = KSPLIT ? (wave >> 1) * (KG / 2) : 0, c_end = KSPLIT ? c_begin + KG / 2 : KG;
    float bias_v[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) bias_v[s] = (p.bias && (!KSPLIT || wave < 2)) ? p.bias[32 * (nt0 + s) + l31] : 0.f;
    float st1[SPW], st2[SPW];
#pragma unroll
    for (int s = 0; s < SPW; ++s) { st1[s] = 0.f; st2[s] = 0.f; }
    const bool want_stats = p.statpart != nullptr;

    int buf = 0;
    bool counted = false;          // the previous slab's stores were all issued: the wait below may leave exactly them in flight
    for (; slab < nslabs; slab += (int)gridDim.x, buf ^= 1) {
        // this wave's rows of the slab are in (the DMA was issued BEFORE the previous slab's stores: leave those in flight -- a store
        // is acknowledged a microsecond after it was issued, and waiting for it here would expose that once per slab)
        if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SPW * 16) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // raw barrier: __syncthreads() would drain the vector counter again (the stores just left in flight, and -- further down -- the
        // LDS-DMA of the next slab)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // ... everyone's are; everyone is done reading the other slot
        issue_slab(slab + (int)gridDim.x, buf ^ 1);
        const float* a_st = al + (size_t)buf * KB * 1024;
        f32x16 acc[SPW];
#pragma unroll
        for (int s = 0; s < SPW; ++s)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
        pw_products<SPW>(a_st, wl, acc, c_begin, c_end, KG, nt0, lane);
        if (KSPLIT) {
            // the upper K half hands its partial tile to the lower one through LDS (fixed order: lower + upper)
            float* r = red + (size_t)(wave & 1) * 32 * 36;
            if (wave >= 2) {
#pragma unroll
                for (int e = 0; e < 16; ++e) r[((e & 3) + 8 * (e >> 2) + 4 * h) * 36 + l31] = acc[0][e];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (wave < 2) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[0][e] += r[((e & 3) + 8 * (e >> 2) + 4 * h) * 36 + l31];
            }
        }
        counted = false;
        if (!KSPLIT || wave < 2) {
            const long long m0 = (long long)slab * PW_BM;
            counted = !p.accum && m0 + PW_BM <= M;      // exactly SPW * 16 stores follow, nothing else
            if (counted) {
                // a full slab, plain stores: nothing to wait for, nothing to branch on
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    float* dst = p.y + m0 * p.ldy + 32 * (nt0 + s) + l31;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = acc[s][e] + bias_v[s];
                        if (want_stats) { st1[s] += v; st2[s] = fmaf(v, v, st2[s]); }
                        dst[(long long)((e & 3) + 8 * (e >> 2) + 4 * h) * p.ldy] = v;
                    }
                }
            } else if (m0 + PW_BM <= M) {
                // a full slab that ADDS to what is there (an input gradient joining another one): all sixteen loads of a sub-tile
                // first, then the stores
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    float* dst = p.y + m0 * p.ldy + 32 * (nt0 + s) + l31;
                    float old[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) old[e] = dst[(long long)((e & 3) + 8 * (e >> 2) + 4 * h) * p.ldy];
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = acc[s][e] + bias_v[s];
                        if (want_stats) { st1[s] += v; st2[s] = fmaf(v, v, st2[s]); }
                        dst[(long long)((e & 3) + 8 * (e >> 2) + 4 * h) * p.ldy] = v + old[e];
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    const int col = 32 * (nt0 + s) + l31;
                    for (int e = 0; e < 16; ++e) {
                        const long long m = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        if (m >= M) continue;
                        float v = acc[s][e] + bias_v[s];
                        if (want_stats) { st1[s] += v; st2[s] = fmaf(v, v, st2[s]); }      // statistics of the conv's own output (before accumulate)
                        float* dst = p.y + m * p.ldy + col;
                        if (p.accum) v += *dst;
                        *dst = v;
                    }
                }
            }
        }
        // (`red` needs no barrier of its own: the upper waves write it again only behind the next slab's barrier, which the lower
        //  waves reach after they have read it)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the ring's last (zero-page) refill
    if (p.statpart && (!KSPLIT || wave < 2)) {
        // per block and channel: (sum, sumsq) over the block's slabs; the two half-waves hold different rows of a column
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const float t1 = st1[s] + __shfl_xor(st1[s], 32), t2 = st2[s] + __shfl_xor(st2[s], 32);
            if (h == 0) {
                float* dst = p.statpart + ((size_t)(p.stat_base + (int)blockIdx.x) * p.Nc + 32 * (nt0 + s) + l31) * 2;
                dst[0] = t1; dst[1] = t2;
            }
        }
    }
}

struct PwLaunch { int nt; size_t lds; int grid; int nslabs; };

bool pw_shape(const IgemmArgs& a, PwLaunch& L) {
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (a.ntaps != 1 || a.taps[0].dd || a.taps[0].dh || a.taps[0].dw || a.at_mode != P3D_AT_NONE || a.ngate || a.f16) return false;
    if (a.isd != 1 || a.ish != 1 || a.isw != 1 || a.osd != 1 || a.osh != 1 || a.osw != 1 || a.ood || a.ooh || a.oow) return false;
    if (a.Gd != a.Di || a.Gh != a.Hi || a.Gw != a.Wi || a.Gd != a.Do || a.Gh != a.Ho || a.Gw != a.Wo) return false;
    if ((a.K & 31) || a.K > 256 || (a.Nc != 64 && a.Nc != 128 && a.Nc != 256) || (long long)a.K * a.Nc > 16384) return false;
    if ((a.ldx & 3) || M < 16384 || M >= (1ll << 31) / 256) return false;
    L.nt = a.Nc / 32;
    const size_t w = (size_t)a.K * a.Nc * 4, ring = (size_t)2 * (a.K / 32) * 1024 * 4, red = a.Nc == 64 ? (size_t)2 * 32 * 36 * 4 : 0;
    L.lds = w + ring + red;
    L.nslabs = (int)((M + PW_BM - 1) / PW_BM);
    const int per_cu = L.lds <= 80 * 1024 ? 2 : 1;
    L.grid = (int)std::min<long long>(std::min(L.nslabs, 256 * per_cu), M / 64);      // (statistics partials: one per block, at most one per 64 rows)
    return true;
}

template <int NT>
hipError_t pw_launch_t(const IgemmArgs& a, const PwLaunch& L, hipStream_t s) {
    static std::once_flag once;
    std::call_once(once, [] {
        hipFuncSetAttribute((const void*)pw_stream_kernel<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)pw_stream_kernel<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (a.wT) hipLaunchKernelGGL((pw_stream_kernel<NT, true>), dim3((unsigned)L.grid), dim3(256), L.lds, s, a, L.nslabs);
    else hipLaunchKernelGGL((pw_stream_kernel<NT, false>), dim3((unsigned)L.grid), dim3(256), L.lds, s, a, L.nslabs);
    return hipGetLastError();
}

}  // namespace

// Is this launch the streaming kernel's case, and with how many blocks (= statistics partials) would it run?  0: no.
int p3d_pw_stream_blocks(const IgemmArgs& a) {
    PwLaunch L;
    return pw_shape(a, L) ? L.grid : 0;
}

hipError_t p3d_launch_pw_stream(const IgemmArgs& a0, hipStream_t s) {
    IgemmArgs a = a0;
    PwLaunch L;
    if (!pw_shape(a, L) || !a.zeros || (a.ldy & 3)) return hipErrorInvalidValue;
    switch (L.nt) {
        case 2: return pw_launch_t<2>(a, L, s);
        case 4: return pw_launch_t<4>(a, L, s);
        case 8: return pw_launch_t<8>(a, L, s);
        default: return hipErrorInvalidValue;
    }
}
