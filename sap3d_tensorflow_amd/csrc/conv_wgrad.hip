// Weight gradient of the P3D convolutions on gfx950: for every kernel tap,
//   dW[tap][k][n] += sum over the dense lattice m of  Xgathered[m + tap][k] * dY[m][n]
// (TF Conv3DBackpropFilterV2 behind tf.nn.conv3d / tf.layers.conv3d / conv3d_transpose at
// reference p3d.py:19,24,86,112,125,172,200-217).  The reduction dimension is the POSITION
// index m, so both operands are staged [position][channel] exactly as they sit in NDHWC memory
// and fed to v_mfma_f32_32x32x2_f32 with k = position.  The M range is split across
// gridDim.y and partial tiles are combined with fp32 global atomics (dW is zeroed once per
// step by the caller).  dbias (column sums of dY) rides along in the tap-0 / k-tile-0 blocks.
#include "p3d_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BKM = 32;     // positions per step

template <int BM, int BN>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int RA = BM / 32, RB = BN / 32;
    constexpr int A_F4 = BM / 4, B_F4 = BN / 4;          // float4 per staged row
    constexpr int A_RPP = 256 / A_F4, B_RPP = 256 / B_F4;  // rows per pass

    __shared__ __attribute__((aligned(16))) float As[BKM * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BKM * LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
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
    const bool stem = p.stem_wfloats != 0;

    long long chunk = (M + p.ksplit - 1) / p.ksplit;
    chunk = (chunk + BKM - 1) / BKM * BKM;
    const long long ms = (long long)blockIdx.y * chunk;
    const long long me = (ms + chunk < M) ? ms + chunk : M;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int a_c4 = (tid % A_F4) * 4, a_r0 = tid / A_F4;
    const int b_c4 = (tid % B_F4) * 4, b_r0 = tid / B_F4;
    const bool do_bias = p.dbias != nullptr && ti == 0 && kt == 0;
    float bsum = 0.f;

    float4 ra[RA], rb[RB];

    auto load_tiles = [&](long long mbase) {
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const long long m = mbase + a_r0 + A_RPP * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < me) {
                int gw = (int)(m % p.Gw); long long t = m / p.Gw;
                int gh = (int)(t % p.Gh); t /= p.Gh;
                int gd = (int)(t % p.Gd); int n = (int)(t / p.Gd);
                const int id = gd * p.isd + tap.dd, ih = gh * p.ish + tap.dh;
                if (!stem) {
                    const int iw = gw * p.isw + tap.dw;
                    if ((unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi &&
                        (unsigned)iw < (unsigned)p.Wi && (k0 + a_c4) < p.K) {
                        const long long off = ((((long long)n * p.Di + id) * p.Hi + ih) * p.Wi + iw) * p.ldx + k0 + a_c4;
                        v = *reinterpret_cast<const float4*>(p.x + off);
                    }
                } else if ((unsigned)id < (unsigned)p.Di && (unsigned)ih < (unsigned)p.Hi) {
                    const long long base = (((long long)n * p.Di + id) * p.Hi + ih) * p.stem_wfloats;
                    const int f0 = gw * p.stem_wstep - p.stem_wpad + k0 + a_c4;
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
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            const long long m = mbase + b_r0 + B_RPP * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < me && (n0 + b_c4) < p.Nc)
                v = *reinterpret_cast<const float4*>(p.dy + m * p.ldy + n0 + b_c4);
            rb[j] = v;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < RA; ++j)
            *reinterpret_cast<float4*>(&As[(a_r0 + A_RPP * j) * LDA + a_c4]) = ra[j];
#pragma unroll
        for (int j = 0; j < RB; ++j)
            *reinterpret_cast<float4*>(&Bs[(b_r0 + B_RPP * j) * LDB + b_c4]) = rb[j];
    };

    if (ms < me) {
        load_tiles(ms);
        store_tiles();
    }
    __syncthreads();

    for (long long mb = ms; mb < me; mb += BKM) {
        const bool more = mb + BKM < me;
        if (more) load_tiles(mb + BKM);
#pragma unroll
        for (int k = 0; k < BKM; k += 2) {
            float a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(k + h) * LDA + wm * (BM / 2) + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[j] = Bs[(k + h) * LDB + wn * (BN / 2) + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        if (do_bias && tid < BN) {
#pragma unroll 8
            for (int k = 0; k < BKM; ++k) bsum += Bs[k * LDB + tid];
        }
        __syncthreads();
        if (more) {
            store_tiles();
            __syncthreads();
        }
    }

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

}  // namespace

const char* p3d_wgrad_variant(const WgradArgs& a) { return (a.K >= 128 && a.Nc >= 128) ? "wgrad_kernel<128,128>" : "wgrad_kernel<64,64>"; }

hipError_t p3d_launch_wgrad(const WgradArgs& a0, hipStream_t s) {
    WgradArgs a = a0;
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    if (M <= 0 || a.ntaps <= 0) return hipSuccess;
    if (a.ntaps > P3D_MAX_TAPS) return hipErrorInvalidValue;
    if (!a.stem_wfloats && ((a.K & 3) || (a.ldx & 3))) return hipErrorInvalidValue;
    if ((a.Nc & 3) || (a.ldy & 3)) return hipErrorInvalidValue;
    const bool big = a.K >= 128 && a.Nc >= 128;
    const int T = big ? 128 : 64;
    const long long tiles = (long long)a.ntaps * ((a.K + T - 1) / T) * ((a.Nc + T - 1) / T);
    long long ks = (1024 + tiles - 1) / tiles;
    const long long maxks = (M + 127) / 128;      // at least 128 positions per split
    if (ks > maxks) ks = maxks;
    if (ks < 1) ks = 1;
    if (ks > 65535) ks = 65535;
    a.ksplit = (int)ks;
    dim3 grid((unsigned)tiles, (unsigned)ks);
    if (big) hipLaunchKernelGGL((wgrad_kernel<128, 128>), grid, dim3(256), 0, s, a);
    else     hipLaunchKernelGGL((wgrad_kernel<64, 64>), grid, dim3(256), 0, s, a);
    return hipGetLastError();
}
