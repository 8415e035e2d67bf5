// Internal kernel-launch interface of libp3dhip (gfx950 only).
//
// Every convolution-like op of the P3D path (tf.nn.conv3d, its two gradients and
// tf.layers.conv3d_transpose, reference p3d.py:18-27,86,112,125,172,200-217) is
// expressed on ONE geometry: a "dense side" (the SAME conv's output lattice) and
// a "gathered side" (the conv's input lattice, reached through kernel taps).
//   conv forward      : iterate dense side,   gather input  at  g*s + (k - pad)
//   conv dgrad/deconv : iterate one residue class of the input lattice
//                       i = g*s + p, gather dense side at g + (p + pad - k)/s
//   conv wgrad        : reduce over the dense side, X gathered, dY dense
// so a launch is described by an iteration grid, per-tap integer offsets and an
// affine output map.  Layout is NDHWC with an explicit row stride (floats per
// position) so channel slices of concat buffers are addressed in place.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define P3D_MAX_TAPS 27
#ifndef P3D_WGRAD_GROUP
#define P3D_WGRAD_GROUP 6     // weight-gradient problems one grouped launch can carry (kernel-argument space)
#endif
#define P3D_STAT_REPLICAS 16   // lanes that share a channel's partial sums in the finalize kernels (fixed 4-step shuffle fold)
#define P3D_FOLD_MAX 32        // fused BatchNorm: up to this many per-tile partials a consuming launch folds itself (else a finalize launch)

// Tuning / diagnostic switches are environment variables only in a -DP3D_TUNING build (tools/*.sh pass it through
// P3D_EXTRA_HIPCC_FLAGS); in the product build they are absent, so a stray variable cannot change results or drop work.
#include <stdlib.h>
inline const char* p3d_tune_env(const char* name) {
#if defined(P3D_TUNING)
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Division by a launch-invariant extent without the ~40-instruction udiv expansion (the kernels' prologues decode linear
// positions into lattice coordinates): q = floor(n / d) for 0 <= n < 2^31 as mulhi(n, mul) >> shift with
// mul = ceil(2^(31+l) / d), l = ceil(log2 d) (Granlund & Montgomery, "Division by invariant integers", Thm 4.2); d = 1: mul = 0.
// (Measured: no change of any launch time -- the divisions sat in the shadow of the first loads -- kept for the shorter code.)
struct P3dFastDiv { unsigned mul; int shift; };
inline P3dFastDiv p3d_fastdiv(unsigned d) {
    P3dFastDiv f{0u, 0};
    if (d <= 1) return f;
    int l = 0;
    while ((1u << l) < d) ++l;
    f.mul = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
    f.shift = l - 1;
    return f;
}
#if defined(__HIPCC__)
__device__ __forceinline__ unsigned p3d_div(unsigned n, P3dFastDiv f) { return f.mul ? (__umulhi(n, f.mul) >> f.shift) : n; }
#endif

// Wave priority of the kernels on the latency-bound main-stream chain.  The filter-gradient kernels that share the chip with
// them from the side stream stay at priority 0, so a SIMD that hosts both issues the chain's instructions first: the chain
// is what the step waits for, the filter gradients only have to be done by the end (measured: 17.13 -> 16.66 ms per step,
// priority 1 and 3 alike; profiles/r03_prio_ab.json).
#if defined(__HIPCC__)
#define P3D_CHAIN_PRIO() __builtin_amdgcn_s_setprio(2)
// Kernel arguments arrive through the scalar cache, one 64-byte line per first touch, and hipcc reads them where they are used:
// a kernel with a 1 KB argument struct walks into one cold line after another (measured: 13 dependent s_load / s_waitcnt round
// trips, 2.3 us, between the entry of the pipelined conv kernel and its first operand load).  Touch every line of the struct
// at entry, all loads in flight together: one miss latency, and what follows hits the cache.
// ONE asm statement per kernel, the wait inside it: scalar loads return out of order and hipcc does not count an asm statement's
// loads, so a load still in flight behind the statement could land in a register the compiler has meanwhile given to
// something else (it did: wrong convolutions).  All loads of the statement target the same scratch SGPR.
typedef const unsigned __attribute__((address_space(4))) p3d_karg_t;
#define P3D_WL(off) "s_load_dword %0, %1, " #off "\n\t"
#define P3D_WL4(a, b, c, d) P3D_WL(a) P3D_WL(b) P3D_WL(c) P3D_WL(d)
template <int LINES>
__device__ __forceinline__ void p3d_warm_kernarg_lines() {
    p3d_karg_t* ka = (p3d_karg_t*)__builtin_amdgcn_kernarg_segment_ptr();
    unsigned t;
    // the largest supported count that stays inside the struct (never a load past the arguments)
    if constexpr (LINES >= 17)
        asm volatile(P3D_WL4(0x0, 0x40, 0x80, 0xc0) P3D_WL4(0x100, 0x140, 0x180, 0x1c0) P3D_WL4(0x200, 0x240, 0x280, 0x2c0)
                     P3D_WL4(0x300, 0x340, 0x380, 0x3c0) P3D_WL(0x400) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(ka) : "memory");
    else if constexpr (LINES >= 14)
        asm volatile(P3D_WL4(0x0, 0x40, 0x80, 0xc0) P3D_WL4(0x100, 0x140, 0x180, 0x1c0) P3D_WL4(0x200, 0x240, 0x280, 0x2c0)
                     P3D_WL(0x300) P3D_WL(0x340) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(ka) : "memory");
    else if constexpr (LINES >= 11)
        asm volatile(P3D_WL4(0x0, 0x40, 0x80, 0xc0) P3D_WL4(0x100, 0x140, 0x180, 0x1c0) P3D_WL(0x200) P3D_WL(0x240) P3D_WL(0x280)
                     "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(ka) : "memory");
    else if constexpr (LINES >= 8)
        asm volatile(P3D_WL4(0x0, 0x40, 0x80, 0xc0) P3D_WL4(0x100, 0x140, 0x180, 0x1c0) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(ka) : "memory");
    else if constexpr (LINES >= 6)
        asm volatile(P3D_WL4(0x0, 0x40, 0x80, 0xc0) P3D_WL(0x100) P3D_WL(0x140) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(ka) : "memory");
    else if constexpr (LINES >= 4)
        asm volatile(P3D_WL4(0x0, 0x40, 0x80, 0xc0) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(ka) : "memory");
    else if constexpr (LINES >= 2)
        asm volatile(P3D_WL(0x0) P3D_WL(0x40) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(ka) : "memory");
}
template <class Args>
__device__ __forceinline__ void p3d_warm_kernargs() { p3d_warm_kernarg_lines<(int)((sizeof(Args) + 63) / 64)>(); }
#endif

struct P3dTap {
    int16_t dd, dh, dw;   // gathered coordinate = g*is + d{d,h,w}
    int16_t widx;         // which [K][N] slab of the weight tensor
};

// ---- BatchNorm fused into the convolutions' operand paths (reference p3d.py:56-81,88-97: every bn -> relu pair
//      between two convs of a bottleneck) ---------------------------------------------------------------------------
// The normalised tensor is never stored: the conv that consumes it reads the RAW output y of the producing conv and
// applies  relu(scale*y + shift)  to its A fragments between LDS and the matrix cores; the per-channel (scale, shift)
// come from the producer's per-tile statistics partials, folded in the consumer's prologue (few partials) or by a
// finalize launch (many).  Backward mirrors it: an input-gradient launch gates its result with the ReLU mask of the
// BatchNorm OUTPUT it differentiates through and leaves per-tile (sum g, sum g*xhat); the next input-gradient launch
// (and the filter gradients) read  dy = k1*g + k2*y + k3  on their operand path.
struct BnFold {            // forward: (scale, shift) of one BatchNorm over C channels
    const float* gamma; const float* beta;
    const float* part; int nparts;               // producer's (sum, sumsq) partials [nparts][C][2]; null: read scale / shift below
    int C;
    float* scale; float* shift; float* mean; float* invstd;     // written by block 0 when `publish` (else read when part == null)
    float* moving_mean; float* moving_var;       // momentum-0.99 update by the publisher when update_moving
    double inv_m;                                // 1 / rows of the normalised tensor
    float eps; int publish; int update_moving;
};
struct BnGradFold {        // backward: dy = k1*g + k2*y + k3  (g: gated gradient of the BN output, y: BN input)
    const float* gamma; const float* mean; const float* invstd;
    const float* part; int nparts;               // (sum g, sum g*xhat) partials [nparts][C][2]; null: read coef below
    int C;
    float* coef;                                 // [3][C] k1, k2, k3: written by block 0 when `publish`, read by the filter gradients
    float* dgamma; float* dbeta;                 // BN parameter gradients, written by the publisher
    double inv_m; int publish;
};
struct BnGate {            // epilogue of an input-gradient launch: g = (scale*y + shift > 0) ? v : 0
    const float* y; int ldy;                     // the BN's input (a conv output) on this launch's output lattice
    const float* scale; const float* shift; const float* mean; const float* invstd;
    float* out; int ldo;                         // gated gradient
    float* part;                                 // [m tiles][Nc][2]  (sum g, sum g*xhat) per output-tile row, plain stores
};
enum { P3D_AT_NONE = 0, P3D_AT_RELU1 = 1, P3D_AT_RELU2 = 2, P3D_AT_GRAD = 3 };
//  RELU1: a = relu(s1*x + t1)                           RELU2: a = relu(s1*x + t1) + relu(s2*x2 + t2)   (ST_B / ST_C sums)
//  GRAD : a = k1*x + k2*x2 + k3   (x = gated gradient, x2 = BN input); padded taps stay 0

// Implicit-GEMM convolution launch:  Y[m, n] (+)= sum_taps sum_k A[m+tap, k] * B_tap[k, n] (+ bias[n])
struct IgemmArgs {
    const float* x;       // gathered operand (already offset to its channel slice)
    int N, Di, Hi, Wi;    // gathered-side extents
    int ldx;              // floats per gathered-side position
    int K;                // reduction channels (Cin of this GEMM)
    int Gd, Gh, Gw;       // iteration grid per sample; M = N*Gd*Gh*Gw
    P3dFastDiv fGd, fGh, fGw;   // (filled by the launcher)
    int isd, ish, isw;    // gathered coord = g*is + tap offset
    float* y;             // output (already offset to its channel slice)
    int Do, Ho, Wo;       // output extents
    int ldy;              // floats per output position
    int Nc;               // output channels of this GEMM
    int osd, osh, osw, ood, ooh, oow;   // output coord = g*os + oo
    const float* w;       // weights: slab widx is [K][Nc] (wT=0) or [Nc][K] (wT=1)
    int wT;
    const float* bias;    // [Nc] or null
    // BatchNorm statistics epilogue: per-(output tile row, channel) partial (sum, sum of squares) of the stored values,
    // written with plain stores to statpart[(stat_base + m_tile) * Nc + col][2] -- no atomics, so the statistics (and
    // everything downstream) are bit-reproducible; p3d_bn_finalize folds the partials in a fixed order.  Null: none.
    float* statpart;
    int stat_base;
    int accum;            // 1: Y += result (gradient accumulation)
    const float* zeros;   // >= 128 B of zeros in device memory (source for padded / tail lanes)
    // K-slicing (filled by the launcher from the plan): slice s of a tile stores its partial tile to
    // slab[(tile * nsplit + s) * BM*BN], the block whose arrival ticket is the last one sums the slices in slice order
    // (bit-reproducible, unlike atomics), applies bias / accumulate / statistics and writes the output.
    float* slab; unsigned* cnt; int nsplit;
    int f16;              // 1: round the operand fragments to fp16 and use the fp16 MFMA (fp32 accumulate); pointwise convs of configs[4]
    int xcd_min_tiles;    // set by the launcher: launches / classes with at least this many tiles map consecutive tiles to one XCD
    // fused BatchNorm on the A operand (P3D_AT_*): x2 is the second source on the gathered lattice (RELU2, GRAD)
    int at_mode;
    const float* x2; int ldx2;
    BnFold f1, f2;        // RELU1 / RELU2: the BatchNorms of x and x2
    BnGradFold gf;        // GRAD
    // gated epilogue (input gradients through a fused BatchNorm + ReLU): the value v of every output element
    // (after bias / accumulate) goes raw to y when raw_store, and gated to gate[q].out; ngate = 0: plain store to y
    int ngate, raw_store;
    BnGate gate[2];
    int ntaps;
    P3dTap taps[P3D_MAX_TAPS];
};

// What differs between the launches of one group -- the residue classes of a transposed conv / of a strided conv's input
// gradient: iteration grid, output offset, kernel taps.  One grouped launch carries all of them (p3d_launch_igemm2_group):
// a class alone offers too few tiles for 256 CUs (deconv3 at 8 clips: 196 tiles of 128x128 per class, eight classes).
#define P3D_IGEMM_CLASSES 8
struct IgemmClass {
    int Gd, Gh, Gw;
    P3dFastDiv fGd, fGh, fGw;
    int ood, ooh, oow;
    int stat_base;        // first statistics partial of this class (its m tiles follow each other)
    int ntaps;
    int blk0;             // first block of this class in the grouped launch (classes in descending order of work)
    const float* w; const float* bias; float* y; float* statpart;      // per class as well: sibling convs on one input (ST_B)
    int nsplit;           // K-slices of this class (its blocks: tiles x slices, a tile's slices in consecutive blocks)
    int tile0;            // first output tile of this class in its grid (> 0: the K-sliced tail of the class before it)
    int ntiles;           // output tiles of this class
    long long slab0, cnt0;      // this class's share of the launch's slab / counter scratch (floats / counters)
    P3dTap taps[P3D_MAX_TAPS];
};
struct IgemmGroupArgs {
    IgemmArgs common;     // everything the classes share (its own grid / offsets / taps are unused)
    int nclass;
    IgemmClass cls[P3D_IGEMM_CLASSES];
};

// Tile and split-K choice of the pipelined kernel (conv_igemm2.hip)
struct P3dIgemmPlan {
    int bm = 64, bn = 64, splits = 1;
    const char* name = "";
    int stream_blocks = 0;      // > 0: the launch runs on the weights-resident streaming kernel (conv_pointwise.hip) with this many blocks
};
// number of output-tile rows (= statistics partials) a launch with this plan produces
inline int p3d_igemm2_mtiles(const IgemmArgs& a, const P3dIgemmPlan& pl) {
    if (pl.stream_blocks > 0) return pl.stream_blocks;      // one statistics partial per block
    const long long M = (long long)a.N * a.Gd * a.Gh * a.Gw;
    return (int)((M + pl.bm - 1) / pl.bm);
}
// Per-stream scratch for K-sliced launches (partial tiles + arrival counters).  Launches on one stream run in order, so
// they share it; the buffers only grow, and an outgrown buffer stays allocated (captured graphs may still name it).
hipError_t p3d_stream_scratch(hipStream_t s, size_t slab_floats, size_t counters, float** slab, unsigned** cnt);
long long p3d_scratch_dirty_counters();      // test hook: non-zero arrival counters with nothing in flight (must be 0)
void p3d_release_scratch();     // frees every scratch buffer (process shutdown; no launch may be in flight)

// Weight-gradient launch: dW[widx][k][n] += sum_m Xg[m+tap, k] * dY[m, n]
struct WgradArgs {
    const float* x;       // gathered operand (conv input side)
    int N, Di, Hi, Wi, ldx, K;
    int Gd, Gh, Gw;       // dense grid (conv output side); M = N*Gd*Gh*Gw
    int isd, ish, isw;
    const float* dy;      // dense operand
    int ldy, Nc;
    float* dw;            // [slab][K][Nc]; the launch ADDS the gradient to it (single writer per element: plain read-modify-write)
    float* dbias;         // [Nc] or null (column sums of dy, added by tap 0 / k-tile 0 blocks)
    int ksplit;           // (chosen by the launcher)
    int greedy;           // 1: launched when nothing else is running -- take every LDS slot (conv_wgrad2.hip, launch_group_t)
    int polite;           // 1: runs beside a chain of small launches whatever its own size -- one block per CU
    int pair;             // 1 (K <= 32): two taps share a 64-row tile (the stem's 28-float kernel rows)
    const float* zeros;   // zero page
    // fused BatchNorm: operand transforms with per-channel coefficients the forward / the input-gradient launches published
    int xt;               // gathered operand: 0 plain, 1 relu(xs1*x + xt1), 2 relu(xs1*x + xt1) + relu(xs2*x2 + xt2)
    const float* x2; int ldx2;
    const float* xs1; const float* xt1; const float* xs2; const float* xt2;     // [K]
    int dyt;              // dense operand: 0 plain, 1 dcoef[0][n]*dy + dcoef[1][n]*dy2 + dcoef[2][n]
    const float* dy2; int ldy2;
    const float* dcoef;   // [3][Nc]
    int ntaps;
    P3dTap taps[P3D_MAX_TAPS];
};

#ifdef __cplusplus
extern "C++" {
#endif

P3dIgemmPlan p3d_igemm2_plan(const IgemmArgs& a, int allow_split);
// Autotuning window (graph build): inside it, the first plan request for a new shape is decided by timing the
// candidates on stream `s` with the caller's real buffers; outside it, a heuristic answers for unseen shapes.
void p3d_tune_begin(hipStream_t s);
void p3d_tune_end();
hipError_t p3d_launch_igemm2(const IgemmArgs& a, const P3dIgemmPlan& plan, hipStream_t s);
// n launches that differ only in what IgemmClass holds, as ONE launch (same tile shape, no operand transform / gates; n <= 8).
// stat_base of every class must be set by the caller (statistics partials of class q start at its stat_base).
bool p3d_igemm2_tail_split(const IgemmArgs& a, const P3dIgemmPlan& pl);      // a single launch whose last round gets K-sliced (goes out grouped)
bool p3d_igemm2_groupable(const IgemmArgs* v, int n, const P3dIgemmPlan& plan);
hipError_t p3d_launch_igemm2_group(const IgemmArgs* v, int n, const P3dIgemmPlan& plan, hipStream_t s);
void p3d_igemm2_override(int tile, int splits);   // test / tools hook: force the tile (0: 64x64, 1: 128x64, 2: 128x128) and the K-slice count; -1 / 0 = no override
// dense 1x1x1 convs over >= 16 384 positions with K * N <= 16 384 (conv_pointwise.hip): 0 = not that case, else the block count
int p3d_pw_stream_blocks(const IgemmArgs& a);
hipError_t p3d_launch_pw_stream(const IgemmArgs& a, hipStream_t s);
hipError_t p3d_launch_wgrad2(const WgradArgs& a, hipStream_t s);
hipError_t p3d_launch_wgrad2_group(const WgradArgs* probs, int n, hipStream_t s);   // up to P3D_WGRAD_GROUP problems, one launch
const char* p3d_wgrad2_variant(const WgradArgs& a);
const char* p3d_wgrad2_group_variant(const WgradArgs* probs, int n, bool fused);      // the label of a grouped launch (its tile)
void p3d_wgrad2_force_tile(int tm, int tn);      // test hook: tile of single-problem launches (64 / 128 each); 0, 0 = the plan's choice

// ---- BatchNorm (tf.layers.batch_normalization, rank-5, eps 1e-3) ------------------------------
struct BnParams {          // device pointers, all [C]
    const float* gamma; const float* beta;
    float* moving_mean; float* moving_var;
    const float* statpart; // [nparts][C][2] partial (sum, sumsq) written by the producer's epilogue or p3d_bn_stats
    int nparts;
    float* scale; float* shift;      // y_hat = scale*y + shift
    float* mean; float* invstd;      // saved for backward
    int C;
};
// use_batch: statistics from `stats` over M rows, else moving stats.  update_moving: momentum 0.99 update.
hipError_t p3d_bn_finalize(const BnParams& bn, long M, int use_batch, int update_moving, float eps, hipStream_t s);
// statpart[b][c] = (sum, sum of squares) over block b's rows of y, b < p3d_bn_stats_parts(M, C)
// (for producers that cannot do it in their epilogue)
int p3d_bn_stats_parts(long M, int C);
hipError_t p3d_bn_stats(const float* y, int ld, long M, int C, float* statpart, hipStream_t s);

// coefficients of a fused BatchNorm's backward from many partials (few: the consuming launch folds them itself)
hipError_t p3d_bn_grad_finalize(const BnGradFold& f, hipStream_t s);

// Fused normalise/activate/add passes.  Modes (reference p3d.py lines in brackets):
//  0: z = relu(bn1(y1))                         [58-59, 88+97, 173-174, 201-202]
//  1: z = relu(bn1(y1) + r)                     [114,133-134 identity residual]
//  2: z = relu(bn1(y1) + bn2(y2))               [114,127,133-134 projected residual]
//  3: z = relu(bn1(y1)) + relu(bn2(y2))         [ST_B 65-72]
//  4: z = r + relu(bn1(y1))                     [ST_C 74-81]
struct BnApplyArgs {
    int mode;
    long M; int C;
    const float* y1; int ld1; const float* scale1; const float* shift1;
    const float* y2; int ld2; const float* scale2; const float* shift2;   // y2 doubles as r (modes 1,4)
    float* z; int ldz;
    float drop_scale;      // >0: inverted dropout with this keep scale (p3d.py:214), keyed by seed
    float drop_rate; unsigned long long seed;
    const unsigned long long* seed_dev;   // non-null: the seed is read from device memory (captured step graphs)
};
hipError_t p3d_bn_apply(const BnApplyArgs& a, hipStream_t s);
// finalize + apply in one launch when every block can fold its own 64 channels' partials (<= 128 per BN, no dropout)
bool p3d_bn_fold_apply_ok(long M, int C, int nparts1, int nparts2, float drop_scale);
hipError_t p3d_bn_fold_apply(const BnApplyArgs& a, const BnParams& bn1, const BnParams& bn2, int batch1, int batch2, int update_moving,
                             float eps, hipStream_t s);

// Backward of the passes above.  Pass 1 reduces per channel sum(dz') and sum(dz' * xhat) into per-block
// partials part1/part2; a finalize pass folds them in block order into coef1/coef2 ([C][2] floats: the two sums / M) and the parameter gradients; pass 2
// writes the input gradients.
struct BnBwdArgs {
    int mode;
    long M; int C;
    const float* dz; int lddz;
    const float* y1; int ld1; const float* scale1; const float* shift1; const float* mean1; const float* invstd1;
    const float* y2; int ld2; const float* scale2; const float* shift2; const float* mean2; const float* invstd2;
    const float* gamma1; const float* gamma2;
    float* part1; float* part2;      // per-block partial sums [nparts][C][2] (plain stores: no atomics), nparts = p3d_bn_bwd_parts(M, C)
    int nparts;
    float* coef1; float* coef2;
    float* dgamma1; float* dbeta1; float* dgamma2; float* dbeta2;      // parameter grads (written)
    int batch1, batch2;    // 1: batch statistics were used (full BN backward), 0: inference BN
    float* dy1; int lddy1; int acc1;
    float* dy2; int lddy2; int acc2;      // dy2 doubles as dr (modes 1,4)
    float drop_scale; float drop_rate; unsigned long long seed; const unsigned long long* seed_dev;
};
int p3d_bn_bwd_parts(long M, int C);
hipError_t p3d_bn_bwd_reduce(const BnBwdArgs& a, hipStream_t s);
hipError_t p3d_bn_bwd_finalize(const BnBwdArgs& a, hipStream_t s);
hipError_t p3d_bn_bwd_apply(const BnBwdArgs& a, hipStream_t s);

// Small-tensor BatchNorm (bn_small.hip): one launch forward, one launch backward, M <= 1024 rows.
struct BnSmallArgs {
    int mode; int M; int C;
    const float* y1; int ld1;
    const float* y2; int ld2;           // second BN input (modes 2,3) or residual (modes 1,4)
    BnParams bn1, bn2;
    int batch1, batch2;                 // 1: batch statistics, 0: moving statistics
    int update_moving; float eps;
    float* z; int ldz;
    const float* dz; int lddz;          // backward
    float* dy1; int lddy1;
    float* dy2; int lddy2; int acc2;
    float* dgamma1; float* dbeta1; float* dgamma2; float* dbeta2;
};
bool p3d_bn_small_ok(long M, int C);
hipError_t p3d_bn_small_fwd(const BnSmallArgs& a, hipStream_t s);
hipError_t p3d_bn_small_bwd(const BnSmallArgs& a, hipStream_t s);

// ---- GroupNorm (gn.hip; reference gn/p3d_gn.py:24-46).  Tables are indexed [n*C + c]. -------------------
struct GnParams {
    const float* gamma; const float* beta;       // [C]
    double* sums;                                // [N][C][2]: forward (sum, sumsq) or backward (sum g, sum g*xhat)
    float* scale; float* shift; float* mean; float* invstd;     // [N][C]
    float* coef;                                 // [N][C][3] backward coefficients (k, c1, c2)
    int C, G;
};
struct GnApplyArgs {
    int mode;                                    // 0,1,3,4,5,6 (gn.hip header)
    long M; int R; int C;                        // M = N*R rows, R rows per sample
    const float* y1; int ld1; GnParams g1;
    const float* y2; int ld2; GnParams g2;       // second GN input (mode 3) or residual / CBAM input (1,4,6)
    const float* cs; const float* ss;            // mode 6: CBAM channel scale [N][C], spatial scale [M]
    float* z; int ldz;
    const float* dz;                             // backward: gradient of z (same stride as z)
    float* dy1; int lddy1;
    float* dy2; int lddy2; int acc2;             // mode 3: GN2 input grad; 1,4: residual grad; 6: grad of the CBAM output
    float drop_scale; float drop_rate; unsigned long long seed; const unsigned long long* seed_dev;
    // one-launch path for small tensors (p3d_gn_small_*): epsilon, and where the parameter gradients are ADDED
    float eps; float* dgamma1; float* dbeta1; float* dgamma2; float* dbeta2;
    float* part; unsigned* counters;             // backward partial sums + arrival counters (set by the launchers)
};
bool p3d_gn_small_ok(int R, int C, int G);       // a (sample, group) slab fits one block's registers
hipError_t p3d_gn_small_fwd(const GnApplyArgs& a, hipStream_t s);    // stats + tables + normalise/activate in one launch
hipError_t p3d_gn_small_bwd(const GnApplyArgs& a, hipStream_t s);    // whole backward in one launch
hipError_t p3d_gn_stats(const float* y, int ld, int N, int R, int C, double* sums, hipStream_t s);
hipError_t p3d_gn_finalize(const GnParams& p, int N, int R, float eps, hipStream_t s);
hipError_t p3d_gn_apply(const GnApplyArgs& a, hipStream_t s);
hipError_t p3d_gn_bwd_reduce(const GnApplyArgs& a, hipStream_t s);
hipError_t p3d_gn_bwd_finalize(const GnParams& p, int N, int R, float* dgamma, float* dbeta, hipStream_t s);
hipError_t p3d_gn_bwd_apply(const GnApplyArgs& a, hipStream_t s);

// ---- CBAM (cbam.hip; reference utils/network.py:198-274 as used at gn/p3d_gn.py:175) --------------------
struct CbamArgs {
    const float* x; int ld;                      // block residual [N, D,H,W, C]
    int N, D, H, W, C, Ch;                       // Ch = C / 8 hidden units
    const float* k0; const float* b0; const float* k1; const float* b1;     // shared MLP  C->Ch->C
    const float* k7;                             // [7,7,7,2,1]
    int chunks;                                  // row chunks per sample of the pooling / backward passes
    float* part;                                 // [N][chunks][C][3] partial (sum, max, ties-of-max)
    float* avg; float* mx; float* ties;          // [N][C]
    float* havg; float* hmx;                     // [N][Ch] post-ReLU hidden activations
    float* cs;                                   // [N][C]   channel scale = sigmoid(mlp(avg) + mlp(max))
    float* sp;                                   // [M][2]   channel-mean / channel-max of x*cs
    float* ss;                                   // [M]      spatial scale = sigmoid(conv7(sp))
    // backward
    const float* dout;                           // [M][C] gradient of the CBAM output (dense)
    float* dpre;                                 // [M]
    float* dsp;                                  // [M][2]
    float* dcs_part;                             // [N][chunks][C]
    float* dO;                                   // [N][C] gradient of the MLP output (pre-sigmoid)
    float* davg; float* dmx;                     // [N][C] gradients of the pooled vectors
    float* dh;                                   // [N][2][Ch] gradients of the hidden activations (avg, max branch)
    float* dx; int lddx; int accx;               // gradient of x
    float* dk0; float* db0; float* dk1; float* db1; float* dk7;
    float* k7part; unsigned* k7counter;          // dK7 partials [blocks][343][2] + arrival counter (set by the launcher)
};
hipError_t p3d_cbam_forward(const CbamArgs& a, hipStream_t s);
hipError_t p3d_cbam_backward(const CbamArgs& a, hipStream_t s);

// ---- self attention (attention.hip; reference utils/network.py:157-192) -----------------------------------
hipError_t p3d_softmax_rows(float* s, long long rows, int cols, int ld, hipStream_t st);          // in place; columns [cols, ld) := 0
hipError_t p3d_softmax_rows_bwd(const float* beta, float* d, long long rows, int cols, int ld, hipStream_t st);   // d := ds, in place
struct AttnMixArgs {                 // z = r * gamma + x  (utils/network.py:191), optional dropout on z (p3d.py:388)
    long long M; int C;
    const float* r; int ldr;         // relu(bn(conv(o)))
    const float* x; int ldx;         // the block's input
    const float* gamma;              // [1]
    float* z; int ldz;
    float drop_scale; float drop_rate; unsigned long long seed; const unsigned long long* seed_dev;
    // backward
    const float* dz; float* dr; float* dx; int accx; float* dgamma;
    float* part; unsigned* counter;  // dgamma partials per block + arrival counter (set by the launcher)
};
hipError_t p3d_attn_mix_fwd(const AttnMixArgs& a, hipStream_t s);
hipError_t p3d_attn_mix_bwd(const AttnMixArgs& a, hipStream_t s);
// The attention core  o = softmax(g f^T) h  (utils/network.py:183-185) without the score matrix in HBM (attention_flash.hip):
// per clip, g [Ng x ch/8] queries, f [Nf x ch/8] keys, h [Nf x ch] values; clip b of a tensor starts b * rows * ld floats in.
struct FlashAttnArgs {
    int B, Ng, Nf, ch;                   // ch in {32, 64, 128, 256}; ch/8 channels in g and f
    const float* g; int ldg;
    const float* f; int ldf;
    const float* h; int ldh;
    float* o; int ldo;                   // forward output [B][Ng][ch]
    float* lse;                          // [B][Ng]: row maximum + log of the row sum (kept for the backward pass)
    // backward
    const float* d_o; int lddo;          // gradient of o
    float* dsum;                         // [B][Ng] scratch: <d_o, o> per row
    float* dg; int lddg; float* df; int lddf; float* dh; int lddh;       // written, not accumulated
};
bool p3d_flash_attn_ok(int ch);
hipError_t p3d_flash_attn_fwd(const FlashAttnArgs& a, hipStream_t s);
hipError_t p3d_flash_attn_bwd(const FlashAttnArgs& a, hipStream_t s);      // three launches: row dots, dg, (df, dh)
hipError_t p3d_pad_rows(const float* src, float* dst, int B, int N, int Npad, int C, hipStream_t s);     // [B][N][C] -> [B][Npad][C], zero tail
hipError_t p3d_unpad_rows(const float* src, float* dst, int B, int N, int Npad, int C, hipStream_t s);   // the reverse (tail dropped)

// ---- max pool (tf.nn.max_pool3d SAME; p3d.py:177,183,189,195) ---------------------------------
struct PoolArgs {
    const float* x; int N, Di, Hi, Wi, C, ldx;
    float* y; int Do, Ho, Wo, ldy;
    int kd, kh, kw, sd, sh, sw, pd, ph, pw;
    // backward
    const float* dy; int lddy; float* dx; int lddx;
    unsigned* idx;       // optional [N*Do*Ho*Wo][C/4] words, one byte per channel: tap (kd,kh,kw scan order) of the
                         // first maximum; written by the forward, read by p3d_maxpool_bwd_gather
};
hipError_t p3d_maxpool_fwd(const PoolArgs& a, hipStream_t s);
// overlapping windows without atomics or a zero fill: every input cell collects from the (few) windows that contain it,
// using the arg-max taps the forward stored.  Writes dx, or adds to it (accumulate = 1).
hipError_t p3d_maxpool_bwd_gather(const PoolArgs& a, int accumulate, hipStream_t s);
bool p3d_maxpool_disjoint(const PoolArgs& a);                    // k == s, no padding: windows do not overlap
hipError_t p3d_maxpool_bwd_disjoint(const PoolArgs& a, int accumulate, hipStream_t s);   // writes / accumulates dx, no atomics

// ---- output head: tf.layers.conv3d_transpose(x, 1, 3, 2, 'same') + sigmoid (p3d.py:217-219) ---
struct HeadArgs {
    const float* x; int N, D, H, W, C;     // input [N,D,H,W,C], output [N,2D,2H,2W,1]
    const float* k;                        // kernel [3,3,3,1,C]
    const float* bias;                     // [1]
    float* logits; float* pred;            // pre- and post-sigmoid
    int sigmoid;                           // 0: pred = logits (p3d_concat head, p3d.py:275)
    const float* dlogits; float* dx; float* dk; float* dbias;
    float* part; unsigned* counter;        // filter-gradient partials [blocks][28][C] + arrival counter (set by the launcher)
};
hipError_t p3d_head_fwd(const HeadArgs& a, hipStream_t s);
hipError_t p3d_head_bwd_input(const HeadArgs& a, hipStream_t s);    // dx written
hipError_t p3d_head_bwd_filter(const HeadArgs& a, hipStream_t s);   // dk, dbias += (per-block partials folded in block order)
// the stride-1 variant tf.layers.conv3d(x, 1, 3, 1, 'same') (gn/p3d_gn.py:537): D,H,W are both input and output extents
hipError_t p3d_headc_fwd(const HeadArgs& a, hipStream_t s);
hipError_t p3d_headc_bwd_input(const HeadArgs& a, hipStream_t s);
hipError_t p3d_headc_bwd_filter(const HeadArgs& a, hipStream_t s);

// ---- loss: Smooth-L1 sum (utils/network.py:49-62, train.py:159) fused with sigmoid backward ---
// loss_out: double accumulator (zeroed by caller).  dlogits = dL/dpred * pred*(1-pred).
hipError_t p3d_smooth_l1(const float* pred, const float* target, long n, double* loss_out,
                         float* dlogits, int through_sigmoid, hipStream_t s);

// ---- Adam (tf.train.AdamOptimizer, epsilon-hat form; train.py:168) ------------------------------
// lr_dev non-null: the bias-corrected step size is read from device memory (captured step graphs), lr_t is ignored
hipError_t p3d_adam(float* p, const float* g, float* m, float* v, long n, float lr_t, const float* lr_dev, float b1, float b2,
                    float eps, hipStream_t s);
// per-step scalars of a captured train step: scal[0..1] = dropout seed (64 bit), scal[2] = Adam's bias-corrected step size
hipError_t p3d_set_step_scalars(unsigned long long* seed_dst, float* lr_dst, unsigned long long seed, float lr_t, hipStream_t s);

// ---- saliency metrics + frame pre-processing (metrics.hip; utils/metrics.py:25-287, dataflow.py:187-216) ------
hipError_t p3d_metric_cc(const float* a, const float* b, int n_maps, int n_pix, double* out, hipStream_t s);
hipError_t p3d_metric_sim(const float* a, const float* b, int n_maps, int n_pix, double* out, hipStream_t s);
hipError_t p3d_metric_nss(const float* sal, const float* fix, int n_maps, int n_pix, double* out, hipStream_t s);
int p3d_metric_auc_pad(int n_pix);      // scratch per map: pad floats (thresholds) + pad + 1 ints (counters)
hipError_t p3d_metric_auc_judd(const float* sal, const float* fix, const float* jitter, int n_maps, int n_pix, float* thr_scratch,
                               int* cnt_scratch, double* out, hipStream_t s);
hipError_t p3d_metric_fix_index(const float* fix, int n_pix, int* idx, int* count, hipStream_t s);
hipError_t p3d_metric_auc_borji(const float* sal, const float* fix, const int* rand_idx, int n_pix, int n_fix, int n_rep, double step,
                                const int* fix_idx, double* out_per_rep, hipStream_t s);
hipError_t p3d_mapf_frames(const unsigned char* bgr, int n_frames, int H0, int W0, float* dst, int H, int W, const float mean_rgb[3],
                           hipStream_t s);
hipError_t p3d_mapf_density(const unsigned char* grey, int n_frames, int H0, int W0, float* dst, int H, int W, hipStream_t s);

// ---- misc ---------------------------------------------------------------------------------------
hipError_t p3d_add_inplace(float* dst, int lddst, const float* src, int ldsrc, long M, int C, hipStream_t s);
hipError_t p3d_copy_strided(float* dst, int lddst, const float* src, int ldsrc, long M, int C, hipStream_t s);
hipError_t p3d_fill_uniform(float* p, long n, float lo, float hi, unsigned long long seed, hipStream_t s);
hipError_t p3d_fill_trunc_normal(float* p, long n, float stddev, unsigned long long seed, hipStream_t s);   // N(0, stddev) within 2 stddev
hipError_t p3d_colsum(const float* dy, int ld, long M, int C, float* out, hipStream_t s);   // out += column sums
// stem re-layout (elementwise.hip): 3-channel input -> 4-channel rows with the W padding written out; packed weights
hipError_t p3d_stem_pad(const float* x, float* x4, long long rows, int W, int Wp, int pad, hipStream_t s);
hipError_t p3d_stem_pack_w(const float* w, float* w4, int taps_hw, int Co, hipStream_t s);          // [kh*kw][3][Co] -> [kh*kw][4][Co]
hipError_t p3d_stem_unpack_dw(const float* dw4, float* dw, int taps_hw, int Co, hipStream_t s);     // dw += the 3 real channels of dw4
// stem filter gradient in one pass over the output gradient (stem_wgrad.hip): [1,7,7,3,64], stride [1,2,2], even output width
struct StemWgradArgs {
    const float* x4; int Wp; int Hi;           // the padded 4-channel copy of the clip [nimg*Hi][Wp][4] (p3d_stem_pad)
    int nimg, Ho, Wo, pad_h;                   // nimg = N*D frames; SAME padding above the image
    const float* dy; int lddy;                 // output gradient [nimg*Ho*Wo][64]; fused: the gradient of the BatchNorm + ReLU OUTPUT
    int fused;                                 // 1: dy is differentiated through bn -> relu on the fly (bn_bwd_apply, mode 0)
    const float* y; int ldy;                   // fused: the conv's output (the BatchNorm's input)
    const float* scale; const float* shift; const float* mean; const float* invstd; const float* gamma;
    const float* coef; int batch;              // fused: (sum g / M, sum g*xhat / M) per channel (bn_bwd_finalize); batch = 0: moving statistics
    float* part;                               // scratch, p3d_stem_wgrad_part_floats() floats
    float* dw;                                 // [7][7][3][64], added to
    int chunks, chunks_per_img, chunks_per_block, pairs_per_wave, slots_per_wave, lds_row;      // set by the launcher
    int reserved_[4];
};
bool p3d_stem_wgrad_ok(int kd, int kh, int kw, int Cin, int Cout, int sd, int sh, int sw, int Wo);
long p3d_stem_wgrad_part_floats();
hipError_t p3d_stem_wgrad(const StemWgradArgs& a, int* nblocks, hipStream_t s);                   // the blocks' partials -> a.part
hipError_t p3d_stem_wgrad_fold(const float* part, int nblocks, float* dw, hipStream_t s);        // dw += the partials, in block order

#ifdef __cplusplus
}
#endif
