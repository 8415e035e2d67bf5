// Implicit-GEMM 3-D convolution for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix
// cores (v_mfma_f32_32x32x2_f32: exact fp32, same peak as the fp32 VALU, leaves the VALU free).
//
// Replaces, for the P3D path: tf.nn.conv3d forward (reference p3d.py:19,24,86,112,125,172,216),
// its input gradient, and tf.layers.conv3d_transpose forward/input-gradient (p3d.py:200,205,210)
// -- see p3d_kernels.h for the shared geometry.  No im2col buffer exists anywhere: the A tile of
// a (tap, k-chunk) step is gathered row by row (one NDHWC position = one contiguous channel run)
// from global memory into LDS, zero-filled where the SAME padding would be.
//
// Block = 256 threads = 4 waves (2 x 2), tile BM x BN x 32, each wave (BM/2) x (BN/2) as
// TM x TN 32x32 MFMA accumulators.  Global loads of step i+1 are issued into registers before
// the MFMAs of step i and written to LDS after them (single LDS buffer, two barriers per step).
#include "p3d_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;          // k-chunk (floats)
constexpr int LDA = BK + 4;     // 36: ds_read_b128 of 16 distinct rows is bank-conflict free

template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmArgs p) {
    constexpr int LDB = BN + 4;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32;     // A float4 per thread per step
    constexpr int RB = BN / 32;     // B float4 per thread per step

    __shared__ __attribute__((aligned(16))) float As[BM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];
    __shared__ int rowN[BM], rowD[BM], rowH[BM], rowW[BM];
    __shared__ long long rowOut[BM];
    __shared__ float sred[2][BN][2];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;

    const long long M = (long long)p.N * p.Gd * p.Gh * p.Gw;
    const int NT = (p.Nc + BN - 1) / BN;
    const int nt = blockIdx.x % NT;
    const long long mt = blockIdx.x / NT;
    const long long m0 = mt * BM;
    const int n0 = nt * BN;
    const bool stem = p.stem_wfloats != 0;

    // ---- per-row coordinates (once per block) ---------------------------------------------------
    for (int r = tid; r < BM; r += 256) {
        long long m = m0 + r;
        if (m < M) {
            int gw = (int)(m % p.Gw); long long t = m / p.Gw;
            int gh = (int)(t % p.Gh); t /= p.Gh;
            int gd = (int)(t % p.Gd); int n = (int)(t / p.Gd);
            rowN[r] = n;
            rowD[r] = gd * p.isd;
            rowH[r] = gh * p.ish;
            rowW[r] = stem ? gw * p.stem_wstep - p.stem_wpad : gw * p.isw;
            int od = gd * p.osd + p.ood, oh = gh * p.osh + p.ooh, ow = gw * p.osw + p.oow;
            rowOut[r] = ((((long long)n * p.Do + od) * p.Ho + oh) * p.Wo + ow) * p.ldy;
        } else {
            rowN[r] = -1; rowD[r] = 0; rowH[r] = 0; rowW[r] = 0; rowOut[r] = -1;
        }
    }
    __syncthreads();

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int kchunks = (p.K + BK - 1) / BK;
    const int nsteps = p.ntaps * kchunks;

    float4 ra[RA], rb[RB];
    const int a_c4 = (tid & 7) * 4;
    const int a_r0 = tid >> 3;

    auto load_tiles = [&](int step) {
        const int t = step / kchunks;
        const int k0 = (step - t * kchunks) * BK;
        const P3dTap tap = p.taps[t];
        // A: gathered rows
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const int r = a_r0 + 32 * j;
            const int n = rowN[r];
            const int id = rowD[r] + tap.dd, ih = rowH[r] + tap.dh;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!stem) {
                const int iw = rowW[r] + tap.dw;
                const bool ok = n >= 0 && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                                (unsigned)iw < (unsigned)p.Wi && (k0 + a_c4) < p.K;
                if (ok) {
                    const long long off = ((((long long)n * p.Di + id) * p.Hi + ih) * p.Wi + iw) * p.ldx + k0 + a_c4;
                    v = *reinterpret_cast<const float4*>(p.x + off);
                }
            } else {
                const bool ok = n >= 0 && (unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi;
                if (ok) {
                    const long long base = (((long long)n * p.Di + id) * p.Hi + ih) * p.stem_wfloats;
                    const int f0 = rowW[r] + k0 + a_c4;
                    float e[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f = f0 + q;
                        e[q] = ((k0 + a_c4 + q) < p.K && (unsigned)f < (unsigned)p.stem_wfloats) ? p.x[base + f] : 0.f;
                    }
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
            }
            ra[j] = v;
        }
        // B: weights of this tap
        const float* wt = p.w + (long long)tap.widx * p.K * p.Nc;
        if (!p.wT) {
            constexpr int F4_PER_ROW = BN / 4;
            constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;
            const int nc = (tid % F4_PER_ROW) * 4;
            const int kr0 = tid / F4_PER_ROW;
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int kr = kr0 + ROWS_PER_PASS * j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((k0 + kr) < p.K && (n0 + nc) < p.Nc)
                    v = *reinterpret_cast<const float4*>(wt + (long long)(k0 + kr) * p.Nc + n0 + nc);
                rb[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int n = a_r0 + 32 * j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((n0 + n) < p.Nc && (k0 + a_c4) < p.K)
                    v = *reinterpret_cast<const float4*>(wt + (long long)(n0 + n) * p.K + k0 + a_c4);
                rb[j] = v;
            }
        }
    };

    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < RA; ++j)
            *reinterpret_cast<float4*>(&As[(a_r0 + 32 * j) * LDA + a_c4]) = ra[j];
        if (!p.wT) {
            constexpr int F4_PER_ROW = BN / 4;
            constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;
            const int nc = (tid % F4_PER_ROW) * 4;
            const int kr0 = tid / F4_PER_ROW;
#pragma unroll
            for (int j = 0; j < RB; ++j)
                *reinterpret_cast<float4*>(&Bs[(kr0 + ROWS_PER_PASS * j) * LDB + nc]) = rb[j];
        } else {
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int n = a_r0 + 32 * j;
                Bs[(a_c4 + 0) * LDB + n] = rb[j].x;
                Bs[(a_c4 + 1) * LDB + n] = rb[j].y;
                Bs[(a_c4 + 2) * LDB + n] = rb[j].z;
                Bs[(a_c4 + 3) * LDB + n] = rb[j].w;
            }
        }
    };

    if (nsteps > 0) {
        load_tiles(0);
        store_tiles();
    }
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        if (step + 1 < nsteps) load_tiles(step + 1);
        // ---- MFMA over the 32-wide chunk.  Within each 8-wide group lane-half h owns k = 4h..4h+3
        // (for A and B alike), so A comes in one ds_read_b128.
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            float4 a[TM];
            float b[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const float4*>(&As[(wm * (BM / 2) + i * 32 + l31) * LDA + c * 8 + 4 * h]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    b[j][s] = Bs[(c * 8 + 4 * h + s) * LDB + wn * (BN / 2) + j * 32 + l31];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s == 0 ? a[i].x : s == 1 ? a[i].y : s == 2 ? a[i].z : a[i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[j][s], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (step + 1 < nsteps) {
            store_tiles();
            __syncthreads();
        }
    }

    // ---- epilogue: bias, optional accumulate, store, per-channel statistics --------------------
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + l31;
        const bool cok = col < p.Nc;
        const float bv = (p.bias && cok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const long long ro = rowOut[r];
                if (ro >= 0 && cok) {
                    float v = acc[i][j][e] + bv;
                    float* dst = p.y + ro + col;
                    if (p.accum) v += *dst;
                    if (p.sigmoid) v = 1.f / (1.f + expf(-v));
                    *dst = v;
                    s1[j] += v;
                    s2[j] += v * v;
                }
            }
        }
    }
    if (p.statpart) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
            if (h == 0) {
                sred[wm][wn * (BN / 2) + j * 32 + l31][0] = s1[j];
                sred[wm][wn * (BN / 2) + j * 32 + l31][1] = s2[j];
            }
        }
        __syncthreads();
        if (tid < BN && (n0 + tid) < p.Nc) {
            float* st = p.statpart + ((size_t)(p.stat_base + mt) * p.Nc + n0 + tid) * 2;      // per-tile partial, plain stores
            st[0] = sred[0][tid][0] + sred[1][tid][0];
            st[1] = sred[0][tid][1] + sred[1][tid][1];
        }
    }
}

}  // namespace

static bool igemm_use_big(const IgemmArgs& a) {
    // Tile choice: big tiles when there is enough work to fill 256 CUs with them.
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    const long long big = ((M + 127) / 128) * ((a.Nc + 127) / 128);
    return big >= 512 && a.Nc >= 128;
}

const char* p3d_igemm_variant(const IgemmArgs& a) { return igemm_use_big(a) ? "igemm_kernel<128,128>" : "igemm_kernel<64,64>"; }

hipError_t p3d_launch_igemm(const IgemmArgs& a, hipStream_t s) {
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (M <= 0 || a.Nc <= 0) return hipSuccess;
    if (a.ntaps > P3D_MAX_TAPS) return hipErrorInvalidValue;
    if (!a.stem_wfloats && ((a.K & 3) || (a.ldx & 3))) return hipErrorInvalidValue;
    if (!a.wT && (a.Nc & 3)) return hipErrorInvalidValue;
    if (a.wT && (a.K & 3)) return hipErrorInvalidValue;
    if (igemm_use_big(a)) {
        const long long blocks = ((M + 127) / 128) * ((a.Nc + 127) / 128);
        hipLaunchKernelGGL((igemm_kernel<128, 128>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    } else {
        const long long blocks = ((M + 63) / 64) * ((a.Nc + 63) / 64);
        hipLaunchKernelGGL((igemm_kernel<64, 64>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
    return hipGetLastError();
}
