// The core of the self-attention block `attention()` (reference utils/network.py:183-185),
//     s = g f^T,   beta = softmax(s),   o = beta h,
// and its gradients, with the [N_g x N_f] score matrix never leaving the chip.  The GEMM path of net.hip stores the scores and
// the attention map (2 x 2.5 GB for the last block of p3d_unetplusplus_ds, p3d.py:388, at 8 clips of 16x112x112 -- and 64x that at
// 32x224x224, which cannot be allocated); here every block recomputes its score tiles from g and f on the matrix cores:
//
//   forward   one wave = 32 queries; loop over key tiles of 32: S^T = f g^T (keys in rows, queries in columns), running
//             maximum / sum per query, o^T += h^T p^T.  Stores o and the log-sum-exp of every row.
//   backward  row dots D = <do, o>; then two kernels that rebuild p = exp(s - lse):
//             per query tile  (loop over keys):    dp^T = h do^T,  ds = p (dp - D),  dg^T += f^T ds^T
//             per key tile    (loop over queries): dh^T += do^T p,  dp = do h^T,  ds = p (dp - D),  df^T += g^T ds
//
// Everything is laid out for v_mfma_f32_32x32x2_f32: lane l supplies A[l % 32][l / 32] and B[l / 32][l % 32], and holds
// C[(e & 3) + 8 (e >> 2) + 4 (l / 32)][l % 32] in accumulator register e.  The score tile is computed TRANSPOSED with
// respect to the operand that stays in registers, so that (i) the softmax reductions of a query run down the 16 registers of
// one lane plus one cross-half shuffle, and (ii) the accumulator registers are, as they stand, the B operand of the next
// product (the K index of that product is simply taken in accumulator order on both sides): no shuffles, no LDS round trip.
// fp32 throughout, as the reference (p3d.py:12); the only deviation from the GEMM path is the order of the sums.
#include "p3d_kernels.h"

#include <hip/hip_runtime.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int frag_row(int e, int hf) { return (e & 3) + 8 * (e >> 2) + 4 * hf; }
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    return z;
}
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ROWS x COLS floats of a row-major tensor -> LDS rows of STRIDE floats, through registers (fetch early, store late);
// rows at or beyond `nrows` become zeros
template <int ROWS, int COLS, int STRIDE>
struct Tile {
    static constexpr int V = ROWS * COLS / 4;
    static constexpr int PER = (V + 255) / 256;
    float4 r[PER];
    __device__ __forceinline__ void fetch(const float* base, int ld, int row0, int nrows, int tid) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * 256;
            r[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < V) {
                const int row = i / (COLS / 4), c4 = (i - row * (COLS / 4)) * 4;
                if (row0 + row < nrows) r[k] = ldg4(base + (long long)(row0 + row) * ld + c4);
            }
        }
    }
    __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + k * 256;
            if (i < V) {
                const int row = i / (COLS / 4), c4 = (i - row * (COLS / 4)) * 4;
                *reinterpret_cast<float4*>(lds + row * STRIDE + c4) = r[k];
            }
        }
    }
};

// acc += A B with A[m = lane % 32][k] read from LDS row (lane % 32) and B[k][n = lane % 32] = breg[.]: the K index runs
// over columns hf * K/2 + t of the LDS row (t < K/2), the same order `load_row_frag` gives the register operand
template <int K, int STRIDE>
__device__ __forceinline__ f32x16 mma_rows(const float* lds, int l31, int hf, const float* breg, f32x16 acc) {
    const float* row = lds + l31 * STRIDE + hf * (K / 2);
    if constexpr ((K / 2) % 4 == 0) {
#pragma unroll
        for (int u = 0; u < K / 2; u += 4) {
            const float4 a = *reinterpret_cast<const float4*>(row + u);
            acc = mfma(a.x, breg[u], acc); acc = mfma(a.y, breg[u + 1], acc);
            acc = mfma(a.z, breg[u + 2], acc); acc = mfma(a.w, breg[u + 3], acc);
        }
    } else {
#pragma unroll
        for (int u = 0; u < K / 2; ++u) acc = mfma(row[u], breg[u], acc);
    }
    return acc;
}

// the register operand of mma_rows: row `r` of a row-major tensor, columns hf * K/2 + t  (zeros when !valid)
template <int K>
__device__ __forceinline__ void load_row_frag(float* dst, const float* base, int ld, long long r, int hf, bool valid) {
    const float* p = base + r * ld + hf * (K / 2);
    if constexpr ((K / 2) % 4 == 0) {
#pragma unroll
        for (int u = 0; u < K / 2; u += 4) {
            const float4 v = valid ? ldg4(p + u) : make_float4(0.f, 0.f, 0.f, 0.f);
            dst[u] = v.x; dst[u + 1] = v.y; dst[u + 2] = v.z; dst[u + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int u = 0; u < K / 2; ++u) dst[u] = valid ? p[u] : 0.f;
    }
}

// acc[m][n] += sum_t A[m = lane % 32 + col0][k_t] * b[t]: the K index is the accumulator-row order frag_row(t, hf) of a
// previous product (whose accumulator registers are `b`), A^T is read from LDS row frag_row(t, hf).  Lanes with
// lane % 32 >= mvalid supply zeros (an M extent below 32).
template <int STRIDE>
__device__ __forceinline__ f32x16 mma_cols(const float* lds, int col, int hf, bool on, const f32x16& b, f32x16 acc) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float a = on ? lds[frag_row(t, hf) * STRIDE + col] : 0.f;
        acc = mfma(a, b[t], acc);
    }
    return acc;
}

// a 32 x 32 accumulator block (rows = channels c0.., columns = the wave's 32 tensor rows) -> row-major global memory, through a
// per-wave LDS patch: dst[(row0 + n) * ld + c0 + m], rows at or beyond nrows and channels at or beyond cmax skipped
__device__ __forceinline__ void store_block_t(const f32x16& acc, float scale_lane, float* patch, float* dst, int ld, long long row0,
                                              long long nrows, int c0, int cmax, int lane) {
    const int l31 = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) patch[l31 * 36 + frag_row(e, hf)] = acc[e] * scale_lane;
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the wave's own LDS writes have landed (one wave, in order)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + k * 64, n = i >> 3, c4 = (i & 7) * 4;
        const float4 v = *reinterpret_cast<const float4*>(patch + n * 36 + c4);
        if (row0 + n < nrows && c0 + c4 < cmax) *reinterpret_cast<float4*>(dst + (row0 + n) * ld + c0 + c4) = v;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(256) void flash_fwd_kernel(FlashAttnArgs a) {
    constexpr int CI = CH / 8, FST = CI + 4, HST = CH + 8;
    __shared__ __attribute__((aligned(16))) float Fs[2][32 * FST];
    __shared__ __attribute__((aligned(16))) float Hs[2][32 * HST];
    __shared__ __attribute__((aligned(16))) float patch[4][32 * 36];
    P3D_CHAIN_PRIO();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int b = blockIdx.y;
    const long long q = (long long)blockIdx.x * 128 + w * 32 + l31;
    const bool qok = q < a.Ng;
    const float* g = a.g + (long long)b * a.Ng * a.ldg;
    const float* f = a.f + (long long)b * a.Nf * a.ldf;
    const float* h = a.h + (long long)b * a.Nf * a.ldh;
    float gq[CI / 2];
    load_row_frag<CI>(gq, g, a.ldg, q, hf, qok);
    f32x16 acc[CH / 32];
#pragma unroll
    for (int cb = 0; cb < CH / 32; ++cb) acc[cb] = zero16();
    float m_run = -INFINITY, l_run = 0.f;
    Tile<32, CI, FST> tf;
    Tile<32, CH, HST> th;
    const int T = (a.Nf + 31) / 32;
    tf.fetch(f, a.ldf, 0, a.Nf, tid); th.fetch(h, a.ldh, 0, a.Nf, tid);
    tf.store(Fs[0], tid); th.store(Hs[0], tid);
    __syncthreads();
    for (int j = 0; j < T; ++j) {
        const int buf = j & 1;
        if (j + 1 < T) { tf.fetch(f, a.ldf, (j + 1) * 32, a.Nf, tid); th.fetch(h, a.ldh, (j + 1) * 32, a.Nf, tid); }
        f32x16 s = mma_rows<CI, FST>(Fs[buf], l31, hf, gq, zero16());        // s[e] = <f[key frag_row(e)], g[q]>
        const int k0 = j * 32;
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (k0 + frag_row(e, hf) >= a.Nf) s[e] = -INFINITY;
            tmax = fmaxf(tmax, s[e]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = __expf(s[e] - m_new); psum += s[e]; }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__any(alpha != 1.f)) {
#pragma unroll
            for (int cb = 0; cb < CH / 32; ++cb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[cb][e] *= alpha;
        }
#pragma unroll
        for (int cb = 0; cb < CH / 32; ++cb) acc[cb] = mma_cols<HST>(Hs[buf], cb * 32 + l31, hf, true, s, acc[cb]);
        if (j + 1 < T) { tf.store(Fs[buf ^ 1], tid); th.store(Hs[buf ^ 1], tid); }
        __syncthreads();
    }
    const float inv = 1.f / l_run;
    float* o = a.o + (long long)b * a.Ng * a.ldo;
    const long long row0 = (long long)blockIdx.x * 128 + w * 32;
#pragma unroll
    for (int cb = 0; cb < CH / 32; ++cb) store_block_t(acc[cb], inv, patch[w], o, a.ldo, row0, a.Ng, cb * 32, CH, lane);
    if (qok && hf == 0) a.lse[(long long)b * a.Ng + q] = m_run + __logf(l_run);
}

// D[row] = <do[row], o[row]>
template <int CH>
__global__ __launch_bounds__(256) void flash_rowdot_kernel(FlashAttnArgs a) {
    constexpr int LPR = CH / 4, RPB = 256 / LPR;            // lanes per row, rows per block
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / LPR;
    const int c4 = (threadIdx.x % LPR) * 4;
    float s = 0.f;
    if (row < (long long)a.B * a.Ng) {
        const float4 x = ldg4(a.d_o + row * a.lddo + c4), y = ldg4(a.o + row * a.ldo + c4);
        s = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o);
    if (row < (long long)a.B * a.Ng && c4 == 0) a.dsum[row] = s;
}

// per query tile: dg
template <int CH>
__global__ __launch_bounds__(256) void flash_bwd_q_kernel(FlashAttnArgs a) {
    constexpr int CI = CH / 8, FST = CI + 4, HST = CH + 4;
    __shared__ __attribute__((aligned(16))) float Fs[2][32 * FST];
    __shared__ __attribute__((aligned(16))) float Hs[2][32 * HST];
    __shared__ __attribute__((aligned(16))) float patch[4][32 * 36];
    P3D_CHAIN_PRIO();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int b = blockIdx.y;
    const long long q = (long long)blockIdx.x * 128 + w * 32 + l31;
    const bool qok = q < a.Ng;
    const float* g = a.g + (long long)b * a.Ng * a.ldg;
    const float* f = a.f + (long long)b * a.Nf * a.ldf;
    const float* h = a.h + (long long)b * a.Nf * a.ldh;
    const float* d_o = a.d_o + (long long)b * a.Ng * a.lddo;
    float gq[CI / 2], doq[CH / 2];
    load_row_frag<CI>(gq, g, a.ldg, q, hf, qok);
    load_row_frag<CH>(doq, d_o, a.lddo, q, hf, qok);
    const float lq = qok ? a.lse[(long long)b * a.Ng + q] : 0.f;
    const float dq = qok ? a.dsum[(long long)b * a.Ng + q] : 0.f;
    f32x16 dg = zero16();
    Tile<32, CI, FST> tf;
    Tile<32, CH, HST> th;
    const int T = (a.Nf + 31) / 32;
    tf.fetch(f, a.ldf, 0, a.Nf, tid); th.fetch(h, a.ldh, 0, a.Nf, tid);
    tf.store(Fs[0], tid); th.store(Hs[0], tid);
    __syncthreads();
    for (int j = 0; j < T; ++j) {
        const int buf = j & 1;
        if (j + 1 < T) { tf.fetch(f, a.ldf, (j + 1) * 32, a.Nf, tid); th.fetch(h, a.ldh, (j + 1) * 32, a.Nf, tid); }
        f32x16 s = mma_rows<CI, FST>(Fs[buf], l31, hf, gq, zero16());        // s[e]  = <f[key], g[q]>
        f32x16 dp = mma_rows<CH, HST>(Hs[buf], l31, hf, doq, zero16());      // dp[e] = <h[key], do[q]>
        const int k0 = j * 32;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = k0 + frag_row(e, hf) < a.Nf ? __expf(s[e] - lq) : 0.f;
            s[e] = p * (dp[e] - dq);                                         // ds^T[key][q]
        }
        dg = mma_cols<FST>(Fs[buf], l31, hf, l31 < CI, s, dg);                // dg^T[c][q] += f[key][c] ds^T[key][q]
        if (j + 1 < T) { tf.store(Fs[buf ^ 1], tid); th.store(Hs[buf ^ 1], tid); }
        __syncthreads();
    }
    float* out = a.dg + (long long)b * a.Ng * a.lddg;
    store_block_t(dg, 1.f, patch[w], out, a.lddg, (long long)blockIdx.x * 128 + w * 32, a.Ng, 0, CI, lane);
}

// per key tile: df, dh
template <int CH>
__global__ __launch_bounds__(256) void flash_bwd_k_kernel(FlashAttnArgs a) {
    constexpr int CI = CH / 8, GST = CI + 4, DST = CH + 8;
    __shared__ __attribute__((aligned(16))) float Gs[2][32 * GST];
    __shared__ __attribute__((aligned(16))) float Ds[2][32 * DST];
    __shared__ float Ls[2][64];                                    // lse (0..31) and row dots (32..63) of the query tile
    __shared__ __attribute__((aligned(16))) float patch[4][32 * 36];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hf = lane >> 5;
    const int b = blockIdx.y;
    const long long k = (long long)blockIdx.x * 128 + w * 32 + l31;
    const bool kok = k < a.Nf;
    const float* g = a.g + (long long)b * a.Ng * a.ldg;
    const float* f = a.f + (long long)b * a.Nf * a.ldf;
    const float* h = a.h + (long long)b * a.Nf * a.ldh;
    const float* d_o = a.d_o + (long long)b * a.Ng * a.lddo;
    const float* lse = a.lse + (long long)b * a.Ng;
    const float* dsum = a.dsum + (long long)b * a.Ng;
    float fk[CI / 2], hk[CH / 2];
    load_row_frag<CI>(fk, f, a.ldf, k, hf, kok);
    load_row_frag<CH>(hk, h, a.ldh, k, hf, kok);
    f32x16 dh[CH / 32], df = zero16();
#pragma unroll
    for (int cb = 0; cb < CH / 32; ++cb) dh[cb] = zero16();
    Tile<32, CI, GST> tg;
    Tile<32, CH, DST> td;
    float tl = 0.f;
    auto fetch_l = [&](int q0) {              // threads 0..31: lse (rows beyond Ng: +inf -> p = 0), 32..63: row dots
        if (tid < 32) tl = q0 + tid < a.Ng ? lse[q0 + tid] : INFINITY;
        else if (tid < 64) tl = q0 + tid - 32 < a.Ng ? dsum[q0 + tid - 32] : 0.f;
    };
    const int T = (a.Ng + 31) / 32;
    tg.fetch(g, a.ldg, 0, a.Ng, tid); td.fetch(d_o, a.lddo, 0, a.Ng, tid); fetch_l(0);
    tg.store(Gs[0], tid); td.store(Ds[0], tid);
    if (tid < 64) Ls[0][tid] = tl;
    __syncthreads();
    for (int j = 0; j < T; ++j) {
        const int buf = j & 1;
        if (j + 1 < T) { tg.fetch(g, a.ldg, (j + 1) * 32, a.Ng, tid); td.fetch(d_o, a.lddo, (j + 1) * 32, a.Ng, tid); fetch_l((j + 1) * 32); }
        f32x16 s = mma_rows<CI, GST>(Gs[buf], l31, hf, fk, zero16());        // s[e] = <g[q frag_row(e)], f[k]>
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = __expf(s[e] - Ls[buf][frag_row(e, hf)]);      // p[q][k]
#pragma unroll
        for (int cb = 0; cb < CH / 32; ++cb) dh[cb] = mma_cols<DST>(Ds[buf], cb * 32 + l31, hf, true, s, dh[cb]);   // dh^T[c][k] += do[q][c] p[q][k]
        f32x16 dp = mma_rows<CH, DST>(Ds[buf], l31, hf, hk, zero16());       // dp[e] = <do[q], h[k]>
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = s[e] * (dp[e] - Ls[buf][32 + frag_row(e, hf)]);
        df = mma_cols<GST>(Gs[buf], l31, hf, l31 < CI, s, df);                // df^T[c][k] += g[q][c] ds[q][k]
        if (j + 1 < T) {
            tg.store(Gs[buf ^ 1], tid); td.store(Ds[buf ^ 1], tid);
            if (tid < 64) Ls[buf ^ 1][tid] = tl;
        }
        __syncthreads();
    }
    const long long row0 = (long long)blockIdx.x * 128 + w * 32;
    float* odh = a.dh + (long long)b * a.Nf * a.lddh;
    float* odf = a.df + (long long)b * a.Nf * a.lddf;
#pragma unroll
    for (int cb = 0; cb < CH / 32; ++cb) store_block_t(dh[cb], 1.f, patch[w], odh, a.lddh, row0, a.Nf, cb * 32, CH, lane);
    store_block_t(df, 1.f, patch[w], odf, a.lddf, row0, a.Nf, 0, CI, lane);
}

hipError_t check(const FlashAttnArgs& a, bool bwd) {
    if (!p3d_flash_attn_ok(a.ch) || a.B <= 0 || a.Ng <= 0 || a.Nf <= 0 || a.B > 65535) return hipErrorInvalidValue;
    const int ci = a.ch / 8;
    if (!a.g || !a.f || !a.h || !a.o || !a.lse || a.ldg < ci || a.ldf < ci || a.ldh < a.ch || a.ldo < a.ch) return hipErrorInvalidValue;
    if ((a.ldg | a.ldf | a.ldh | a.ldo) & 3) return hipErrorInvalidValue;           // 16-byte rows
    if (bwd) {
        if (!a.d_o || !a.dsum || !a.dg || !a.df || !a.dh || a.lddo < a.ch || a.lddg < ci || a.lddf < ci || a.lddh < a.ch) return hipErrorInvalidValue;
        if ((a.lddo | a.lddg | a.lddf | a.lddh) & 3) return hipErrorInvalidValue;
    }
    return hipSuccess;
}

template <int CH>
hipError_t fwd_t(const FlashAttnArgs& a, hipStream_t s) {
    flash_fwd_kernel<CH><<<dim3((unsigned)((a.Ng + 127) / 128), (unsigned)a.B), 256, 0, s>>>(a);
    return hipGetLastError();
}
template <int CH>
hipError_t bwd_t(const FlashAttnArgs& a, hipStream_t s) {
    constexpr int RPB = 256 / (CH / 4);
    const long long rows = (long long)a.B * a.Ng;
    flash_rowdot_kernel<CH><<<dim3((unsigned)((rows + RPB - 1) / RPB)), 256, 0, s>>>(a);
    flash_bwd_q_kernel<CH><<<dim3((unsigned)((a.Ng + 127) / 128), (unsigned)a.B), 256, 0, s>>>(a);
    flash_bwd_k_kernel<CH><<<dim3((unsigned)((a.Nf + 127) / 128), (unsigned)a.B), 256, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace

bool p3d_flash_attn_ok(int ch) { return ch == 32 || ch == 64 || ch == 128 || ch == 256; }

hipError_t p3d_flash_attn_fwd(const FlashAttnArgs& a, hipStream_t s) {
    const hipError_t e = check(a, false);
    if (e != hipSuccess) return e;
    switch (a.ch) {
        case 32: return fwd_t<32>(a, s);
        case 64: return fwd_t<64>(a, s);
        case 128: return fwd_t<128>(a, s);
        default: return fwd_t<256>(a, s);
    }
}

hipError_t p3d_flash_attn_bwd(const FlashAttnArgs& a, hipStream_t s) {
    const hipError_t e = check(a, true);
    if (e != hipSuccess) return e;
    switch (a.ch) {
        case 32: return bwd_t<32>(a, s);
        case 64: return bwd_t<64>(a, s);
        case 128: return bwd_t<128>(a, s);
        default: return bwd_t<256>(a, s);
    }
}
