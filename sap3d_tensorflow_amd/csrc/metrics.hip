// Saliency metrics and the frame pre-processing either side of the P3D path.
//
//  * CC / SIM / NSS / AUC-Judd / AUC-Borji as the reference computes them on the last frame of every validation clip
//    (utils/metrics.py:25-287 with utils/metric_utils.py:10-53; called at train.py:258-260, test.py:160-183).  Maps of
//    one shape only: the reference's resize branch (skimage) is not on the path the trainers take.  Arithmetic is
//    float64 on float32 inputs.  One 256-thread block per map; every reduction folds in a fixed order, counts use
//    integer atomics: results are bit-reproducible.
//  * mapf (dataflow.py:187-216): decoded BGR uint8 frame -> RGB, minus the channel means, bilinear resize (float32 path;
//    the grey density maps go through OpenCV's uint8 fixed-point path, mapf_density_kernel)
//    (cv2.INTER_LINEAR, what tensorpack's imgaug.Resize uses) to the clip size, / 255 -- one pass, written straight into
//    the NDHWC clip buffer; and the grey-level density map -> resize -> / 255.
#include "p3d_kernels.h"
#include <math.h>

namespace {

constexpr int TPB = 256;

// fixed-order block reduction of a double (sum) through LDS; every thread gets the result
__device__ __forceinline__ double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_min(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmin(red[tid], red[tid + o]);
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ double block_max(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmax(red[tid], red[tid + o]);
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}
// min / max that propagate NaN like numpy's np.min / np.max
__device__ __forceinline__ double nan_aware(double folded, double any_nan) { return any_nan > 0.0 ? NAN : folded; }

// CC (utils/metrics.py:227-250): both maps standardised (mean 0, population std 1), then np.corrcoef.  corrcoef of two
// standardised maps equals the covariance over the product of the standard deviations of the ORIGINAL maps; computed
// here with centred second moments (two passes), which is what np.corrcoef does after the (idempotent) standardisation.
__global__ __launch_bounds__(TPB) void cc_kernel(const float* a, const float* b, int n, double* out) {
    __shared__ double red[TPB];
    const float* pa = a + (size_t)blockIdx.x * n;
    const float* pb = b + (size_t)blockIdx.x * n;
    double sa = 0, sb = 0;
    for (int i = threadIdx.x; i < n; i += TPB) { sa += (double)pa[i]; sb += (double)pb[i]; }
    const double ma = block_sum(sa, red) / n, mb = block_sum(sb, red) / n;
    double saa = 0, sbb = 0, sab = 0;
    for (int i = threadIdx.x; i < n; i += TPB) {
        const double da = (double)pa[i] - ma, db = (double)pb[i] - mb;
        saa += da * da; sbb += db * db; sab += da * db;
    }
    saa = block_sum(saa, red); sbb = block_sum(sbb, red); sab = block_sum(sab, red);
    if (threadIdx.x == 0) out[blockIdx.x] = sab / sqrt(saa * sbb);       // 0/0 -> NaN for a flat map, like the reference
}

// SIM (utils/metrics.py:258-287): each map -> range [0,1] -> sum 1; sum of element-wise minima.
__global__ __launch_bounds__(TPB) void sim_kernel(const float* a, const float* b, int n, double* out) {
    __shared__ double red[TPB];
    const float* pa = a + (size_t)blockIdx.x * n;
    const float* pb = b + (size_t)blockIdx.x * n;
    double mna = INFINITY, mxa = -INFINITY, mnb = INFINITY, mxb = -INFINITY, nan = 0;
    for (int i = threadIdx.x; i < n; i += TPB) {
        const double x = pa[i], y = pb[i];
        if (x != x || y != y) nan = 1;
        mna = fmin(mna, x); mxa = fmax(mxa, x); mnb = fmin(mnb, y); mxb = fmax(mxb, y);
    }
    nan = block_max(nan, red);
    mna = nan_aware(block_min(mna, red), nan); mxa = nan_aware(block_max(mxa, red), nan);
    mnb = nan_aware(block_min(mnb, red), nan); mxb = nan_aware(block_max(mxb, red), nan);
    const double ra = mxa - mna, rb = mxb - mnb;
    double sa = 0, sb = 0;
    for (int i = threadIdx.x; i < n; i += TPB) { sa += ((double)pa[i] - mna) / ra; sb += ((double)pb[i] - mnb) / rb; }
    sa = block_sum(sa, red); sb = block_sum(sb, red);
    double acc = 0;
    for (int i = threadIdx.x; i < n; i += TPB) {
        const double x = ((double)pa[i] - mna) / ra / sa, y = ((double)pb[i] - mnb) / rb / sb;
        acc += (x != x) ? x : ((y != y) ? y : fmin(x, y));      // np.minimum propagates NaN from either side
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// NSS (utils/metrics.py:200-224): mean of the standardised saliency map at fixated pixels (fixation map > 0.5).
__global__ __launch_bounds__(TPB) void nss_kernel(const float* s, const float* f, int n, double* out) {
    __shared__ double red[TPB];
    const float* ps = s + (size_t)blockIdx.x * n;
    const float* pf = f + (size_t)blockIdx.x * n;
    double sum = 0;
    for (int i = threadIdx.x; i < n; i += TPB) sum += (double)ps[i];
    const double mean = block_sum(sum, red) / n;
    double var = 0, fs = 0, fc = 0;
    for (int i = threadIdx.x; i < n; i += TPB) {
        const double d = (double)ps[i] - mean;
        var += d * d;
        if (pf[i] > 0.5f) { fs += d; fc += 1.0; }
    }
    var = block_sum(var, red); fs = block_sum(fs, red); fc = block_sum(fc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = (fs / sqrt(var / n)) / fc;   // no fixation: 0/0 = NaN (np.mean of an empty slice)
}

// ---- AUC-Judd (utils/metrics.py:25-85) -----------------------------------------------------------------------------
// thresholds = the saliency values at fixated pixels, descending; above[k] = #{S >= thr_k}; tp[k+1] = (k+1)/n_fix,
// fp[k+1] = (above[k] - k - 1) / (n_pix - n_fix); area by the trapezoid rule between (0,0) and (1,1).
// Instead of n_fix passes over the map: sort the thresholds once, then every pixel finds by bisection the first
// threshold it reaches and bumps that slot's counter (integer atomics: order-independent); above = prefix sum.
// scratch per map: thr[npad] floats + cnt[npad + 1] ints, npad = next power of two >= n_pix.
__global__ __launch_bounds__(TPB) void auc_judd_kernel(const float* s, const float* f, const float* jitter, int n, int npad,
                                                       float* thr_all, int* cnt_all, double* out) {
    __shared__ double red[TPB];
    __shared__ int s_nfix;
    const float* ps = s + (size_t)blockIdx.x * n;
    const float* pf = f + (size_t)blockIdx.x * n;
    const float* pj = jitter ? jitter + (size_t)blockIdx.x * n : nullptr;
    float* thr = thr_all + (size_t)blockIdx.x * npad;
    int* cnt = cnt_all + (size_t)blockIdx.x * (npad + 1);
    const int tid = threadIdx.x;
    // 1. gather S at fixations.  Slot order does not matter (sorted next) but must not depend on timing: each thread owns
    //    a contiguous chunk of pixels, chunk offsets come from a scan of the per-thread counts.
    const int chunk = (n + TPB - 1) / TPB;
    const int i0 = tid * chunk, i1 = min(n, i0 + chunk);
    int mine = 0;
    for (int i = i0; i < i1; ++i) mine += pf[i] > 0.5f;
    __shared__ int offs[TPB + 1];
    offs[tid + 1] = mine;
    if (tid == 0) offs[0] = 0;
    __syncthreads();
    if (tid == 0) { for (int t = 0; t < TPB; ++t) offs[t + 1] += offs[t]; s_nfix = offs[TPB]; }
    __syncthreads();
    const int nfix = s_nfix;
    if (nfix == 0) { if (tid == 0) out[blockIdx.x] = NAN; return; }     // "no fixation to predict"
    {
        int w = offs[tid];
        for (int i = i0; i < i1; ++i)
            if (pf[i] > 0.5f) thr[w++] = ps[i] + (pj ? pj[i] : 0.f);
    }
    // pad to a power of two with -inf (sorts to the end of a descending order)
    int np2 = 1;
    while (np2 < nfix) np2 <<= 1;
    for (int i = nfix + tid; i < np2; i += TPB) thr[i] = -INFINITY;
    for (int i = tid; i <= np2; i += TPB) cnt[i] = 0;
    __syncthreads();
    // 2. bitonic sort, descending
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += TPB) {
                const int l = i ^ j;
                if (l > i) {
                    const float x = thr[i], y = thr[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { thr[i] = y; thr[l] = x; }
                }
            }
            __syncthreads();
        }
    // 3. every pixel: first threshold index j with thr[j] <= v  (all thresholds before it are > v)
    for (int i = tid; i < n; i += TPB) {
        const float v = ps[i] + (pj ? pj[i] : 0.f);
        int lo = 0, hi = nfix;                       // answer in [0, nfix]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (thr[mid] > v) lo = mid + 1; else hi = mid;
        }
        if (lo < nfix) atomicAdd(&cnt[lo], 1);
    }
    __syncthreads();
    // 4. above[k] = prefix sum of cnt; done in place by thread 0 chunks then offsets (n_fix is small next to n_pix)
    {
        const int c2 = (nfix + TPB - 1) / TPB;
        const int k0 = tid * c2, k1 = min(nfix, k0 + c2);
        int run = 0;
        for (int k = k0; k < k1; ++k) { run += cnt[k]; cnt[k] = run; }
        offs[tid + 1] = run;
        if (tid == 0) offs[0] = 0;
        __syncthreads();
        if (tid == 0) for (int t = 0; t < TPB; ++t) offs[t + 1] += offs[t];
        __syncthreads();
        const int add = offs[tid];
        for (int k = k0; k < k1; ++k) cnt[k] += add;
        __syncthreads();
    }
    // 5. trapezoids: points P_0 = (0,0), P_{k+1} = (fp_k, tp_k), P_{nfix+1} = (1,1)
    const double inv_fix = 1.0 / (double)nfix, inv_non = 1.0 / (double)(n - nfix);
    double area = 0;
    for (int k = tid; k <= nfix; k += TPB) {
        // segment from point k to point k + 1
        const double x0 = k == 0 ? 0.0 : (double)(cnt[k - 1] - k) * inv_non, y0 = k == 0 ? 0.0 : (double)k * inv_fix;
        const double x1 = k == nfix ? 1.0 : (double)(cnt[k] - k - 1) * inv_non, y1 = k == nfix ? 1.0 : (double)(k + 1) * inv_fix;
        area += (x1 - x0) * (y1 + y0) * 0.5;
    }
    area = block_sum(area, red);
    if (tid == 0) out[blockIdx.x] = area;
}

// ---- AUC-Borji (utils/metrics.py:88-154), one map per launch, one block per random split ---------------------------
// S = map scaled to [0,1]; S_rand[:, rep] = S at the caller's random pixel indices r[n_fix][n_rep] (numpy draws them:
// random.randint(0, n_pix, [n_fix, n_rep]), utils/metrics.py:139); thresholds arange(0, max(S_fix, S_rand[:,rep]), step)
// reversed; tp = share of S_fix >= thr, fp = share of S_rand >= thr; trapezoid area; the caller averages the splits.
__global__ __launch_bounds__(TPB) void auc_borji_kernel(const float* s, const float* f, const int* r, int n, int nfix, int nrep,
                                                        double step, const int* fix_idx, double* out) {
    __shared__ double red[TPB];
    const int rep = blockIdx.x, tid = threadIdx.x;
    double mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < n; i += TPB) { const double x = s[i]; mn = fmin(mn, x); mx = fmax(mx, x); }
    mn = block_min(mn, red); mx = block_max(mx, red);
    const double rng = mx - mn;
    double top = -INFINITY;
    for (int i = tid; i < nfix; i += TPB) {
        top = fmax(top, ((double)s[fix_idx[i]] - mn) / rng);
        top = fmax(top, ((double)s[r[(size_t)i * nrep + rep]] - mn) / rng);
    }
    top = block_max(top, red);
    // thresholds 0, step, 2 step, ... < top  (np.r_[0:top:step]); taken from the largest down
    int nthr = top > 0.0 ? (int)ceil(top / step) : 0;       // len(np.arange(0, top, step)) = ceil(top / step)
    double area = 0, px = 0, py = 0;                  // previous point, starts at (0,0)
    for (int k = 0; k < nthr; ++k) {
        const double thr = (double)(nthr - 1 - k) * step;
        double ctp = 0, cfp = 0;
        for (int i = tid; i < nfix; i += TPB) {
            ctp += (((double)s[fix_idx[i]] - mn) / rng >= thr) ? 1.0 : 0.0;
            cfp += (((double)s[r[(size_t)i * nrep + rep]] - mn) / rng >= thr) ? 1.0 : 0.0;
        }
        ctp = block_sum(ctp, red); cfp = block_sum(cfp, red);
        const double x = cfp / nfix, y = ctp / nfix;
        area += (x - px) * (y + py) * 0.5;
        px = x; py = y;
    }
    area += (1.0 - px) * (1.0 + py) * 0.5;
    if (tid == 0) out[rep] = area;
}

// fixated pixel indices in ascending order (S[F] of the reference), one block
__global__ __launch_bounds__(TPB) void fix_index_kernel(const float* f, int n, int* idx, int* count) {
    __shared__ int offs[TPB + 1];
    const int tid = threadIdx.x;
    const int chunk = (n + TPB - 1) / TPB;
    const int i0 = tid * chunk, i1 = min(n, i0 + chunk);
    int mine = 0;
    for (int i = i0; i < i1; ++i) mine += f[i] > 0.5f;
    offs[tid + 1] = mine;
    if (tid == 0) offs[0] = 0;
    __syncthreads();
    if (tid == 0) { for (int t = 0; t < TPB; ++t) offs[t + 1] += offs[t]; *count = offs[TPB]; }
    __syncthreads();
    int w = offs[tid];
    for (int i = i0; i < i1; ++i)
        if (f[i] > 0.5f) idx[w++] = i;
}

// ---- mapf (dataflow.py:198-216) ---------------------------------------------------------------------------------------
// cv2.resize(INTER_LINEAR) on float32: source coordinate (d + 0.5) * scale - 0.5, floor, clamped to the border with
// weight 0 past it; horizontal pass then vertical pass, float32 products and sums (no fused multiply-add).
__device__ __forceinline__ void lin_coef(int d, double scale, int extent, int& s0, int& s1, float& w1) {
    float fx = (float)(((double)d + 0.5) * scale - 0.5);       // cv2: fx = (float)((dx + 0.5) * scale_x - 0.5)
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= extent - 1) { sx = extent - 1; fx = 0.f; }
    s0 = sx; s1 = min(sx + 1, extent - 1); w1 = fx;
}
template <int CH>
__global__ __launch_bounds__(TPB) void mapf_kernel(const unsigned char* src, int n_frames, int H0, int W0, float* dst, int H, int W,
                                                   float m0, float m1, float m2, int flip_bgr) {
    const double sx = (double)W0 / W, sy = (double)H0 / H;
    const long long total = (long long)n_frames * H * W;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const long long fr = i / ((long long)W * H);
        int x0, x1, y0, y1; float wx, wy;
        lin_coef(x, sx, W0, x0, x1, wx);
        lin_coef(y, sy, H0, y0, y1, wy);
        const unsigned char* f0 = src + (size_t)fr * H0 * W0 * CH;
        const float mean[3] = {m0, m1, m2};
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int cs = (CH == 3 && flip_bgr) ? 2 - c : c;         // im[:, :, ::-1]
            const float p00 = (float)f0[((size_t)y0 * W0 + x0) * CH + cs] - mean[c];
            const float p01 = (float)f0[((size_t)y0 * W0 + x1) * CH + cs] - mean[c];
            const float p10 = (float)f0[((size_t)y1 * W0 + x0) * CH + cs] - mean[c];
            const float p11 = (float)f0[((size_t)y1 * W0 + x1) * CH + cs] - mean[c];
            const float r0 = __fadd_rn(__fmul_rn(p00, 1.f - wx), __fmul_rn(p01, wx));
            const float r1 = __fadd_rn(__fmul_rn(p10, 1.f - wx), __fmul_rn(p11, wx));
            const float v = __fadd_rn(__fmul_rn(r0, 1.f - wy), __fmul_rn(r1, wy));
            dst[i * CH + c] = v / 255.f;
        }
    }
}

// The grey density maps are resized as uint8 images (dataflow.py:210-214: cv2.imread(GRAYSCALE) -> Resize -> / 255.), i.e.
// through OpenCV's fixed-point INTER_LINEAR (resize.cpp, HResizeLinear / VResizeLinear<uchar, int, short>): 11-bit weights
// cvRound(w * 2048), int32 horizontal pass, vertical pass uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2).
// The horizontal tables zero the weight at a clamped border, the vertical pass clips the row indices and keeps the weights.
__device__ __forceinline__ void lin_coef_u8(int d, double scale, int extent, bool clamp_weight, int& s0, int& s1, int& w0, int& w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(f);
    f -= (float)sx;
    if (clamp_weight) {
        if (sx < 0) { sx = 0; f = 0.f; }
        if (sx >= extent - 1) { sx = extent - 1; f = 0.f; }
    }
    w0 = __float2int_rn(__fmul_rn(1.f - f, 2048.f));      // saturate_cast<short>(cvRound(.)): round half to even
    w1 = __float2int_rn(__fmul_rn(f, 2048.f));
    s0 = min(max(sx, 0), extent - 1); s1 = min(max(sx + 1, 0), extent - 1);
}
__global__ __launch_bounds__(TPB) void mapf_density_kernel(const unsigned char* src, int n_frames, int H0, int W0, float* dst, int H, int W) {
    const double sx = (double)W0 / W, sy = (double)H0 / H;
    const bool same = H0 == H && W0 == W;                 // cv::resize copies when the sizes agree
    const long long total = (long long)n_frames * H * W;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const long long fr = i / ((long long)W * H);
        const unsigned char* f0 = src + (size_t)fr * H0 * W0;
        int v;
        if (same) {
            v = f0[(size_t)y * W0 + x];
        } else {
            int x0, x1, a0, a1, y0, y1, b0, b1;
            lin_coef_u8(x, sx, W0, true, x0, x1, a0, a1);
            lin_coef_u8(y, sy, H0, false, y0, y1, b0, b1);
            const int r0 = (int)f0[(size_t)y0 * W0 + x0] * a0 + (int)f0[(size_t)y0 * W0 + x1] * a1;
            const int r1 = (int)f0[(size_t)y1 * W0 + x0] * a0 + (int)f0[(size_t)y1 * W0 + x1] * a1;
            v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
        }
        dst[i] = (float)((double)v / 255.0);              // numpy: uint8 / 255. is float64, then fed as float32
    }
}

}  // namespace

hipError_t p3d_metric_cc(const float* a, const float* b, int n_maps, int n_pix, double* out, hipStream_t s) {
    hipLaunchKernelGGL(cc_kernel, dim3(n_maps), dim3(TPB), 0, s, a, b, n_pix, out);
    return hipGetLastError();
}
hipError_t p3d_metric_sim(const float* a, const float* b, int n_maps, int n_pix, double* out, hipStream_t s) {
    hipLaunchKernelGGL(sim_kernel, dim3(n_maps), dim3(TPB), 0, s, a, b, n_pix, out);
    return hipGetLastError();
}
hipError_t p3d_metric_nss(const float* sal, const float* fix, int n_maps, int n_pix, double* out, hipStream_t s) {
    hipLaunchKernelGGL(nss_kernel, dim3(n_maps), dim3(TPB), 0, s, sal, fix, n_pix, out);
    return hipGetLastError();
}
int p3d_metric_auc_pad(int n_pix) { int p = 1; while (p < n_pix) p <<= 1; return p; }
hipError_t p3d_metric_auc_judd(const float* sal, const float* fix, const float* jitter, int n_maps, int n_pix, float* thr_scratch,
                               int* cnt_scratch, double* out, hipStream_t s) {
    hipLaunchKernelGGL(auc_judd_kernel, dim3(n_maps), dim3(TPB), 0, s, sal, fix, jitter, n_pix, p3d_metric_auc_pad(n_pix), thr_scratch,
                       cnt_scratch, out);
    return hipGetLastError();
}
hipError_t p3d_metric_fix_index(const float* fix, int n_pix, int* idx, int* count, hipStream_t s) {
    hipLaunchKernelGGL(fix_index_kernel, dim3(1), dim3(TPB), 0, s, fix, n_pix, idx, count);
    return hipGetLastError();
}
hipError_t p3d_metric_auc_borji(const float* sal, const float* fix, const int* rand_idx, int n_pix, int n_fix, int n_rep, double step,
                                const int* fix_idx, double* out_per_rep, hipStream_t s) {
    hipLaunchKernelGGL(auc_borji_kernel, dim3(n_rep), dim3(TPB), 0, s, sal, fix, rand_idx, n_pix, n_fix, n_rep, step, fix_idx, out_per_rep);
    return hipGetLastError();
}
hipError_t p3d_mapf_frames(const unsigned char* bgr, int n_frames, int H0, int W0, float* dst, int H, int W, const float mean_rgb[3],
                           hipStream_t s) {
    const long long total = (long long)n_frames * H * W;
    const unsigned grid = (unsigned)((total + TPB - 1) / TPB > 65535 ? 65535 : (total + TPB - 1) / TPB);
    hipLaunchKernelGGL(mapf_kernel<3>, dim3(grid), dim3(TPB), 0, s, bgr, n_frames, H0, W0, dst, H, W, mean_rgb[0], mean_rgb[1], mean_rgb[2], 1);
    return hipGetLastError();
}
hipError_t p3d_mapf_density(const unsigned char* grey, int n_frames, int H0, int W0, float* dst, int H, int W, hipStream_t s) {
    const long long total = (long long)n_frames * H * W;
    const unsigned grid = (unsigned)((total + TPB - 1) / TPB > 65535 ? 65535 : (total + TPB - 1) / TPB);
    hipLaunchKernelGGL(mapf_density_kernel, dim3(grid), dim3(TPB), 0, s, grey, n_frames, H0, W0, dst, H, W);
    return hipGetLastError();
}
