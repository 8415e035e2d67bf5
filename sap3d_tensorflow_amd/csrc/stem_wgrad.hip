// Filter gradient of firstconv1 (p3d.py:172: [1,7,7,3,64], stride [1,2,2]) in ONE pass over the output gradient.
//
// dW[kh][kw][c][co] = sum over output positions p of x[p's patch][kh][kw][c] * dY[p][co]: a 147 x 64 result reduced over
// N*D*Ho*Wo positions (401 408 at 8 clips of 16x112x112: 7.6 GFLOP, 103 MB of dY, 19 MB of clip).  It is the LAST launch of a
// backward pass -- its operand is the last gradient the pass produces -- so nothing hides it.  The generic filter-gradient
// kernel (conv_wgrad2.hip) treats a kernel row as a tap of K = 28 (the 4-channel padded copy of the clip), packs two taps
// into a 64-row tile and walks dY four times: 180 us.  Here:
//
//  * the 147 result rows are enumerated WITHOUT the padding channel (a = kh*21 + kw*3 + c): five 32-row MFMA tiles instead of
//    eight, 10 v_mfma_f32_32x32x2_f32 per pair of positions for all seven kernel rows -- one pass over dY;
//  * a wave owns the whole 160 x 64 accumulator (160 registers).  A block of eight waves walks chunks of four output rows: the
//    13 input rows a chunk touches are staged in LDS (double-buffered, one barrier per chunk; rows outside the image are
//    zeros), each wave takes an eighth of the chunk's position pairs and reads its A operand from LDS in the MFMA layout --
//    ONE ds_read_b32 per lane and tile (lane i of tile t reads element a = 32t + i of the patch; the row stride is 32 mod 64
//    words, so the two or three kernel rows a tile spans fall into different banks);
//  * the B operand comes straight from global memory, one 8-byte load per lane and tensor (two positions x 256 B per wave): lane
//    j holds channels 2j and 2j+1 -- the column order of an accumulator tile is free, it is undone when the block's result is
//    written.  A register ring keeps 3 (any size) or 6 (the reference's clip sizes) pairs of positions in flight across chunk
//    boundaries (raw s_barrier, no vmcnt drain);
//  * FUSED: the BatchNorm + ReLU backward of the stem (bn_bwd_apply, mode 0) is evaluated on the B operand as it arrives
//    (dy = k (g - c1 - xhat c2), g = dz where the normalised value is positive): the stem conv has no input gradient, so its
//    output gradient is read by this kernel only and is never written -- one 309 MB elementwise pass less on the main stream;
//  * the eight waves of a block put their accumulators into LDS side by side and every thread adds up, in wave order, the eight
//    values of its output elements (sw_block_result); the block writes ONE partial (37.6 KB) with plain stores, and
//    stem_wgrad_fold_kernel sums the partials in a fixed order: bit-reproducible, no atomics.
//
// Measured (8 clips of 16x112x112, `bench.py --kernels`): 82 us + 6 us fold = 92 TFLOP/s of the 7.6 GFLOP (0.59 of the fp32 matrix
// roof) against 180 us + a 65 us elementwise pass before.  Versions on the way, and what each step was (tools/ab/stem_parts.sh
// times the kernel with one ingredient compiled out, tools/micro/mfma_mix.hip the same instruction mix without memory):
//   152 us  A operand gathered from global memory (5 loads per pair touching 4-6 cache lines each)
//   134 us  LDS-staged, runtime cursors (60 scalar instructions per pair)
//   122 us  tiled variant: compile-time slots, 7-stage ring; (a one-thread-per-element fold of the 256 partials: 60 us by itself)
//   109 us  B operand through buffer loads (scalar offsets: six 64-bit per-lane pointers less, the staging registers no longer
//           spill -- their reload drained the B pipeline at every chunk), one register set for A / B, LDS reads and arithmetic of
//           the next slot placed between the MFMAs (sched_group_barrier)
//    82 us  the block's result: eight waves side by side in LDS instead of seven rounds of dependent LDS read-add-write pairs
//           (30 us: the kernel without its main loop took 40 us)
// In the train step the kernel shares its 0.4 ms with Adam (346 us of HBM traffic on the main stream), so the step follows the
// bytes, not this kernel: 15.44 -> 15.30 ms came with the first version and stayed there.
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "stem_wgrad.hip sizes its staging ring for gfx950's 160 KB of LDS per CU (up to 128 KB per block)"
#endif
#include <hip/hip_runtime.h>

#include "p3d_kernels.h"

namespace {
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int SW_WAVES = 8, SW_THREADS = SW_WAVES * 64, SW_ROWS = 147, SW_TILES = 5, SW_CO = 64, SW_STAGES = 4, SW_AHEAD = 3;
constexpr int SW_ELEMS = SW_ROWS * SW_CO;
constexpr int SW_RC = 4, SW_SLOTS = 2 * SW_RC + 5;      // output rows per chunk; input rows a chunk touches

struct SwStage { float2 z, y; };
struct SwBn { float scale[2], shift[2], mean[2], invstd[2], k[2], c1[2], c2[2]; };

// wave-uniform position in the block's work: chunk (flat over images), pair i of the wave's share of the chunk -> output row r
// of the chunk, pair pw of that row
struct SwCursor { int chunk, i, img, oh0, lim, f, r, pw; };
__device__ __forceinline__ void sw_enter(SwCursor& c, const StemWgradArgs& a, int chunk, int wave, int chunk_end) {
    const int half = a.Wo >> 1;
    const int ch = chunk < chunk_end ? chunk : chunk_end - 1;    // past the block's run: stay on valid memory, the products are skipped
    c.chunk = chunk; c.i = 0;
    c.img = ch / a.chunks_per_img;
    c.oh0 = (ch - c.img * a.chunks_per_img) * SW_RC;
    int rows = a.Ho - c.oh0;
    if (rows > SW_RC) rows = SW_RC;
    c.lim = chunk < chunk_end ? rows * half : 0;
    c.f = wave * a.pairs_per_wave;
    c.r = c.f / half;
    c.pw = c.f - c.r * half;
}
// a wave's share of a chunk is pairs_per_wave pairs in slots_per_wave (a multiple of the pipeline's 4 stages) slots: the slots
// beyond the share flow through the pipeline like pairs (their loads re-read the chunk's first pair) and their products are skipped
__device__ __forceinline__ bool sw_live(const SwCursor& c, const StemWgradArgs& a) { return c.i < a.pairs_per_wave && c.f < c.lim; }
__device__ __forceinline__ void sw_next(SwCursor& c, const StemWgradArgs& a, int wave, int chunk_end) {
    ++c.f;
    if (++c.pw == (a.Wo >> 1)) { c.pw = 0; ++c.r; }
    if (++c.i == a.slots_per_wave) sw_enter(c, a, c.chunk + 1, wave, chunk_end);
}

template <bool FUSED>
__device__ __forceinline__ void sw_load_b(SwStage& s, const StemWgradArgs& a, const SwCursor& c, int boff_z, int boff_y) {
    const bool live = sw_live(c, a);
    const int r = live ? c.r : 0, pw = live ? c.pw : 0;          // (a pair that is not there: the chunk's first, which is)
    const long long pos = ((long long)c.img * a.Ho + c.oh0 + r) * a.Wo + 2 * pw;
    s.z = *reinterpret_cast<const float2*>(a.dy + pos * a.lddy + boff_z);
    if (FUSED) s.y = *reinterpret_cast<const float2*>(a.y + pos * a.ldy + boff_y);
}

template <bool FUSED>
__device__ __forceinline__ void sw_b_operand(const SwStage& s, const SwBn& bn, float (&b)[2]) {
    const float z[2] = {s.z.x, s.z.y}, y[2] = {s.y.x, s.y.y};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float d = z[h];
        if (FUSED) {       // bn_bwd_apply_kernel<0> (elementwise.hip), one element
            const float v = fmaf(bn.scale[h], y[h], bn.shift[h]);
            const float g = v > 0.f ? z[h] : 0.f;
            const float xh = (y[h] - bn.mean[h]) * bn.invstd[h];       // (moving statistics: c1 = c2 = 0, d = k g)
            d = bn.k[h] * ((g - bn.c1[h]) - xh * bn.c2[h]);
        }
        b[h] = d;
    }
}

// The block's result: the eight waves' accumulators summed in wave order.  In three phases (tiles 0-1, 2-3, 4) every wave puts its
// accumulators of the phase's tiles into its own 16 KB of LDS -- all eight at once, 128 KB -- and every thread then adds up the eight
// values of its output elements and writes them to the block's partial in [a][co] order.  (First version: one wave after the other
// added its 160 registers into ONE 40 KB buffer, 7 rounds of 80 dependent LDS read-add-write pairs: 30 us of a 110 us kernel.)
constexpr int SW_RES_PAIRS = 4;                                        // (tile, half) pairs per phase
constexpr int SW_RES_FLOATS = SW_WAVES * SW_RES_PAIRS * 16 * 64;       // 128 KB
__device__ __forceinline__ void sw_block_result(const f32x16 (&acc)[SW_TILES][2], float* red, float* part_all, int wave, int lane) {
    float* part = part_all + (size_t)blockIdx.x * SW_ELEMS;
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        const int t0 = 2 * ph, nt = ph < 2 ? 2 : 1;
        __syncthreads();                 // the staged rows (or the previous phase) are read
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
            if (tl < nt)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[(((wave * SW_RES_PAIRS) + tl * 2 + h) * 16 + e) * 64 + lane] = acc[t0 + tl][h][e];
        __syncthreads();
        for (int o = threadIdx.x; o < nt * 32 * SW_CO; o += SW_THREADS) {
            const int tl = o >> 11, i = (o >> 6) & 31, co = o & 63;
            const int ai = 32 * (t0 + tl) + i;
            if (ai < SW_ROWS) {
                const int h = co & 1, j = co >> 1;                          // lane j of half h holds channel 2j + h
                const int hh = (i >> 2) & 1, e = (i & 3) + 4 * (i >> 3);    // accumulator row i = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
                const float* src = red + ((tl * 2 + h) * 16 + e) * 64 + hh * 32 + j;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < SW_WAVES; ++w) sum += src[w * SW_RES_PAIRS * 16 * 64];
                part[ai * SW_CO + co] = sum;
            }
        }
    }
}

__device__ __forceinline__ void sw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NST: float4 loads per thread that stage one chunk's 13 input rows (13 * Wp float4 over 512 threads)
// The kernel for ANY even output width (test clips, odd sizes): runtime loops, slots that are skipped.  The reference's clip sizes
// run on stem_wgrad_tiled_kernel below.
template <bool FUSED, int NST>
__global__ __launch_bounds__(SW_THREADS) void stem_wgrad_kernel(StemWgradArgs a) {
    p3d_warm_kernargs<StemWgradArgs>();
    extern __shared__ __attribute__((aligned(16))) float sw_lds[];       // 2 x [13][RS] staged rows; reused for the fold
    const int lane = threadIdx.x & 63, l31 = lane & 31, k = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RS = a.lds_row;                                            // words per staged row (>= Wp*4, 32 mod 64)
    const int buf_words = SW_SLOTS * RS;
    // lane constants: where element a = 32 t + i of a patch sits relative to the patch's first pixel in the staged rows
    int aoff[SW_TILES];
#pragma unroll
    for (int t = 0; t < SW_TILES; ++t) {
        const int ai = 32 * t + l31;
        const int kh = ai / 21, r = ai - kh * 21, kw = r / 3, c = r - kw * 3;
        aoff[t] = ai < SW_ROWS ? kh * RS + (kw + 2 * k) * 4 + c : 3;     // beyond a = 146: a padding channel (0.0)
    }
    const int boff_z = k * a.lddy + 2 * l31, boff_y = k * a.ldy + 2 * l31;
    SwBn bn;
    if (FUSED) {
        const int batch = a.batch;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * l31 + h;
            bn.scale[h] = a.scale[c]; bn.shift[h] = a.shift[c]; bn.mean[h] = a.mean[c]; bn.invstd[h] = a.invstd[c];
            bn.k[h] = a.gamma[c] * bn.invstd[h];
            bn.c1[h] = batch ? a.coef[2 * c] : 0.f; bn.c2[h] = batch ? a.coef[2 * c + 1] : 0.f;
        }
    }
    // staging: float4 q = tid + 512 m of a chunk's [13][Wp] float4 -> row slot, column (fixed per thread)
    int st_slot[NST], st_col[NST];
#pragma unroll
    for (int m = 0; m < NST; ++m) {
        const int q = threadIdx.x + SW_THREADS * m;
        st_slot[m] = q / a.Wp;
        st_col[m] = q - st_slot[m] * a.Wp;
    }
    f32x16 acc[SW_TILES][2];
#pragma unroll
    for (int t = 0; t < SW_TILES; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][h][e] = 0.f;

    const int chunk0 = blockIdx.x * a.chunks_per_block;
    int chunk_end = chunk0 + a.chunks_per_block;
    if (chunk_end > a.chunks) chunk_end = a.chunks;

    // a chunk's input rows: loaded from a clamped (always valid) row, zeroed on the way into LDS when the row is outside the image
    auto stage_load = [&](int chunk, float4 (&v)[NST]) {
        const int img = chunk / a.chunks_per_img;
        const int ih0 = 2 * (chunk - img * a.chunks_per_img) * SW_RC - a.pad_h;
        const float* __restrict__ base = a.x4 + (long long)img * a.Hi * a.Wp * 4;
#pragma unroll
        for (int m = 0; m < NST; ++m) {
            int ih = ih0 + st_slot[m];
            ih = ih < 0 ? 0 : (ih >= a.Hi ? a.Hi - 1 : ih);
            v[m] = *reinterpret_cast<const float4*>(base + ((long long)ih * a.Wp + st_col[m]) * 4);
        }
    };
    auto stage_store = [&](int chunk, int buf, const float4 (&v)[NST]) {
        const int img = chunk / a.chunks_per_img;
        const int ih0 = 2 * (chunk - img * a.chunks_per_img) * SW_RC - a.pad_h;
#pragma unroll
        for (int m = 0; m < NST; ++m) {
            const bool ok = (unsigned)(ih0 + st_slot[m]) < (unsigned)a.Hi;
            if (st_slot[m] < SW_SLOTS)
                *reinterpret_cast<float4*>(sw_lds + buf * buf_words + st_slot[m] * RS + st_col[m] * 4) = ok ? v[m] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    float4 stg[NST];
    stage_load(chunk0, stg);
    stage_store(chunk0, 0, stg);
    // the B pipeline: SW_AHEAD pairs in flight, running across chunk boundaries
    SwStage st[SW_STAGES];
    SwCursor lc, cc;
    sw_enter(lc, a, chunk0, wave, chunk_end);
    cc = lc;
#pragma unroll
    for (int i = 0; i < SW_AHEAD; ++i) {
        sw_load_b<FUSED>(st[i], a, lc, boff_z, boff_y);
        sw_next(lc, a, wave, chunk_end);
    }
    sw_barrier();
    for (int chunk = chunk0; chunk < chunk_end; ++chunk) {
        const int buf = (chunk - chunk0) & 1;
        const bool more = chunk + 1 < chunk_end;
        if (more) stage_load(chunk + 1, stg);
        const float* __restrict__ rows = sw_lds + buf * buf_words;
        for (int i0 = 0; i0 < a.slots_per_wave; i0 += SW_STAGES) {
#pragma unroll
            for (int s = 0; s < SW_STAGES; ++s) {
                sw_load_b<FUSED>(st[(s + SW_AHEAD) % SW_STAGES], a, lc, boff_z, boff_y);
                sw_next(lc, a, wave, chunk_end);
                __builtin_amdgcn_sched_barrier(0);
                if (sw_live(cc, a)) {
                    const float* __restrict__ ap = rows + 2 * cc.r * RS + 16 * cc.pw;
                    float av[SW_TILES];
#pragma unroll
                    for (int t = 0; t < SW_TILES; ++t) av[t] = ap[aoff[t]];
                    float b[2];
                    sw_b_operand<FUSED>(st[s], bn, b);
#pragma unroll
                    for (int t = 0; t < SW_TILES; ++t)
#pragma unroll
                        for (int h = 0; h < 2; ++h) acc[t][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], b[h], acc[t][h], 0, 0, 0);
                }
                sw_next(cc, a, wave, chunk_end);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) stage_store(chunk + 1, buf ^ 1, stg);
        sw_barrier();
    }

    sw_block_result(acc, sw_lds, a.part, wave, lane);
}

// The reference's clip sizes: 4 output rows x (Wo / 2) pairs = 8 waves x PER pairs exactly (PER = 14: 112-pixel clips, 28: 224),
// Ho a multiple of 4 -- every wave owns HALF an output row of every chunk, nothing is masked, and a chunk's PER slots are
// unrolled: the A operand of slot j sits at a compile-time offset from the wave's base (ds_read immediate), the B pointer moves
// by a constant, and a 7-stage register ring (PER is a multiple of 7) keeps 6 pairs of positions in flight across chunk
// boundaries.  One slot = [B loads of slot j+6] [A reads + BatchNorm arithmetic of slot j+1] [10 MFMAs of slot j] in ONE
// basic block per chunk, so the loads and the arithmetic issue in the shadow of the matrix pipe.
constexpr int SW_RING = 7;
// The B operand through buffer loads: the tensor's descriptor and the wave-uniform byte offset of the pair sit in scalar registers,
// the lane's part of the address is ONE 32-bit register for the whole kernel -- no per-lane pointer arithmetic (the flat form kept
// six 64-bit pointers in vector registers, and the staging registers spilled)
typedef float sw_v2f __attribute__((ext_vector_type(2)));
struct SwRsrc { __amdgpu_buffer_rsrc_t z, y; };
template <bool FUSED>
__device__ __forceinline__ void sw_load_at(SwStage& s, const SwRsrc& rs, int off_z, int off_y, int lane_z, int lane_y) {
    const sw_v2f z = __builtin_bit_cast(sw_v2f, __builtin_amdgcn_raw_buffer_load_b64(rs.z, lane_z, off_z, 0));
    s.z = make_float2(z.x, z.y);
    if (FUSED) {
        const sw_v2f y = __builtin_bit_cast(sw_v2f, __builtin_amdgcn_raw_buffer_load_b64(rs.y, lane_y, off_y, 0));
        s.y = make_float2(y.x, y.y);
    }
}

template <bool FUSED, int NST, int PER>
__global__ __launch_bounds__(SW_THREADS) void stem_wgrad_tiled_kernel(StemWgradArgs a) {
    static_assert(PER % SW_RING == 0, "the register ring has to divide a wave's share of a chunk");
    p3d_warm_kernargs<StemWgradArgs>();
    extern __shared__ __attribute__((aligned(16))) float sw_lds[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, k = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int RS = a.lds_row;
    const int buf_words = SW_SLOTS * RS;
    // the wave's half row: output row r of the chunk, first pair pw0
    const int r = wave >> 1, pw0 = (wave & 1) * PER;
    int aoff[SW_TILES];          // words: the lane's element of the patch of the wave's FIRST pair, in buffer 0
#pragma unroll
    for (int t = 0; t < SW_TILES; ++t) {
        const int ai = 32 * t + l31;
        const int kh = ai / 21, rr = ai - kh * 21, kw = rr / 3, c = rr - kw * 3;
        aoff[t] = 2 * r * RS + 16 * pw0 + (ai < SW_ROWS ? kh * RS + (kw + 2 * k) * 4 + c : 3);
    }
    SwBn bn;
    if (FUSED) {
        const int batch = a.batch;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * l31 + h;
            bn.scale[h] = a.scale[c]; bn.shift[h] = a.shift[c]; bn.mean[h] = a.mean[c]; bn.invstd[h] = a.invstd[c];
            bn.k[h] = a.gamma[c] * bn.invstd[h];
            bn.c1[h] = batch ? a.coef[2 * c] : 0.f; bn.c2[h] = batch ? a.coef[2 * c + 1] : 0.f;
        }
    }
    int st_slot[NST], st_col[NST];
#pragma unroll
    for (int m = 0; m < NST; ++m) {
        const int q = threadIdx.x + SW_THREADS * m;
        st_slot[m] = q / a.Wp;
        st_col[m] = q - st_slot[m] * a.Wp;
    }
    f32x16 acc[SW_TILES][2];
#pragma unroll
    for (int t = 0; t < SW_TILES; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][h][e] = 0.f;

    const int chunk0 = blockIdx.x * a.chunks_per_block;
    int chunk_end = chunk0 + a.chunks_per_block;
    if (chunk_end > a.chunks) chunk_end = a.chunks;

    auto stage_load = [&](int chunk, float4 (&v)[NST]) {
        const int img = chunk / a.chunks_per_img;
        const int ih0 = 2 * (chunk - img * a.chunks_per_img) * SW_RC - a.pad_h;
        const float* __restrict__ base = a.x4 + (long long)img * a.Hi * a.Wp * 4;
#pragma unroll
        for (int m = 0; m < NST; ++m) {
            int ih = ih0 + st_slot[m];
            ih = ih < 0 ? 0 : (ih >= a.Hi ? a.Hi - 1 : ih);
            v[m] = *reinterpret_cast<const float4*>(base + ((long long)ih * a.Wp + st_col[m]) * 4);
        }
    };
    auto stage_store = [&](int chunk, int buf, const float4 (&v)[NST]) {
        const int img = chunk / a.chunks_per_img;
        const int ih0 = 2 * (chunk - img * a.chunks_per_img) * SW_RC - a.pad_h;
#pragma unroll
        for (int m = 0; m < NST; ++m) {
            const bool ok = (unsigned)(ih0 + st_slot[m]) < (unsigned)a.Hi;
            if (st_slot[m] < SW_SLOTS)
                *reinterpret_cast<float4*>(sw_lds + buf * buf_words + st_slot[m] * RS + st_col[m] * 4) = ok ? v[m] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // B operand of the wave's first pair of a chunk (positions are contiguous along a row and from row to row: pair j is 2 j further)
    // B operand of the wave's first pair of a chunk (positions are contiguous along a row and from row to row: pair j is 2 j further):
    // byte offsets, wave-uniform; the launcher checked that the tensors are smaller than 2 GB
    const int lane_z = (k * a.lddy + 2 * l31) * 4, lane_y = (k * a.ldy + 2 * l31) * 4;
    const long long npos = (long long)a.nimg * a.Ho * a.Wo;
    SwRsrc rs;
    rs.z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (int)(npos * a.lddy * 4), 0x00020000);
    rs.y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(FUSED ? a.y : a.dy), 0, (int)(npos * (FUSED ? a.ldy : a.lddy) * 4), 0x00020000);
    auto b_base = [&](int chunk, int& oz, int& oy) {
        const int pos = chunk * SW_RC * a.Wo + r * a.Wo + 2 * pw0;      // chunks tile the images: Ho % 4 == 0
        oz = pos * a.lddy * 4;
        oy = pos * a.ldy * 4;
    };
    const int step_z = 8 * a.lddy, step_y = 8 * a.ldy;

    float4 stg[NST];
    stage_load(chunk0, stg);
    stage_store(chunk0, 0, stg);
    SwStage st[SW_RING];
    int pz, py;
    b_base(chunk0, pz, py);
#pragma unroll
    for (int j = 0; j < SW_RING - 1; ++j) sw_load_at<FUSED>(st[j], rs, pz + j * step_z, py + j * step_y, lane_z, lane_y);
    sw_barrier();
    float av[SW_TILES], b[2];
    {
        const float* __restrict__ rows = sw_lds;
#pragma unroll
        for (int t = 0; t < SW_TILES; ++t) av[t] = rows[aoff[t]];
        sw_b_operand<FUSED>(st[0], bn, b);
    }
    for (int chunk = chunk0; chunk < chunk_end; ++chunk) {
        const int buf = (chunk - chunk0) & 1;
        const bool more = chunk + 1 < chunk_end;
        const int next = more ? chunk + 1 : chunk;        // (the block's last chunk: the ring re-reads its own first pairs, unused)
        stage_load(next, stg);
        int pzn, pyn;
        b_base(next, pzn, pyn);
        const float* __restrict__ rows = sw_lds + buf * buf_words;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            constexpr int AH = SW_RING - 1;
            const int jl = j + AH;                        // the slot whose B operand is requested now
            if (jl < PER) sw_load_at<FUSED>(st[jl % SW_RING], rs, pz + jl * step_z, py + jl * step_y, lane_z, lane_y);
            else sw_load_at<FUSED>(st[jl % SW_RING], rs, pzn + (jl - PER) * step_z, pyn + (jl - PER) * step_y, lane_z, lane_y);
            __builtin_amdgcn_sched_barrier(0);            // the loads stay in their slot (the scheduler would bunch them up, and drain)
            // a tile's two MFMAs, then the A operand of the NEXT slot into the register they just read; the next slot's B operand
            // (the BatchNorm arithmetic) spread between them
            float nb[2] = {0.f, 0.f};
            if (j + 1 < PER) sw_b_operand<FUSED>(st[(j + 1) % SW_RING], bn, nb);
#pragma unroll
            for (int t = 0; t < SW_TILES; ++t) {
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[t][h] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], b[h], acc[t][h], 0, 0, 0);
                if (j + 1 < PER) av[t] = rows[aoff[t] + 16 * (j + 1)];
            }
            b[0] = nb[0]; b[1] = nb[1];
#pragma unroll
            for (int q = 0; q < SW_TILES; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) stage_store(next, buf ^ 1, stg);
        sw_barrier();
        if (more) {       // slot 0 of the next chunk
            const float* __restrict__ nrows = sw_lds + (buf ^ 1) * buf_words;
#pragma unroll
            for (int t = 0; t < SW_TILES; ++t) av[t] = nrows[aoff[t]];
            sw_b_operand<FUSED>(st[0], bn, b);
        }
        pz = pzn; py = pyn;
    }
    sw_block_result(acc, sw_lds, a.part, wave, lane);
}

// dw [7][7][3][64] += the blocks' partials.  16 lanes per element, one batch of loads each (a single thread walking 256 partials
// 37.6 KB apart is 32 dependent trips to memory: 60 us); lane g sums partials g, g+16, ... in double, the 16 sums are added in
// lane order: a fixed order.
constexpr int SW_FOLD_G = 16;
__global__ __launch_bounds__(64 * SW_FOLD_G) void stem_wgrad_fold_kernel(const float* __restrict__ part, int nblocks, float* dw) {
    __shared__ double sums[SW_FOLD_G][64];
    const int idx = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double s = 0.0;
    if (idx < SW_ELEMS) {
#pragma unroll 16
        for (int b = g; b < nblocks; b += SW_FOLD_G) s += (double)part[(size_t)b * SW_ELEMS + idx];
    }
    sums[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && idx < SW_ELEMS) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < SW_FOLD_G; ++q) t += sums[q][threadIdx.x];
        dw[idx] += (float)t;
    }
}

int sw_lds_row(int Wp) { return ((Wp * 4 + 31) / 64) * 64 + 32; }       // >= Wp*4 and 32 mod 64
size_t sw_lds_bytes(int Wp) {
    const size_t stage = (size_t)2 * SW_SLOTS * sw_lds_row(Wp) * sizeof(float), red = (size_t)SW_RES_FLOATS * sizeof(float);
    return stage > red ? stage : red;
}
constexpr int SW_MAX_WP = 236;       // 6 staging loads per thread: clips up to 231 pixels wide

// The dynamic-LDS limit of a kernel is a PER-DEVICE attribute: it is set before every launch (a host-side table update: a process
// with handles on two GPUs, or two launching threads, must not find it unset -- ADVICE round 4; the launch itself is ~3 us of
// host time, this call well under one).
template <class K>
hipError_t sw_launch_k(K kernel, const StemWgradArgs& a, unsigned blocks, size_t lds, hipStream_t s) {
    if (lds > 160u * 1024u) return hipErrorInvalidValue;       // one block may take the CU's whole LDS on gfx950, not more
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(SW_THREADS), lds, s, a);
    return hipGetLastError();
}
template <bool FUSED, int NST>
hipError_t sw_launch(const StemWgradArgs& a, unsigned blocks, size_t lds, hipStream_t s) {
    return sw_launch_k(stem_wgrad_kernel<FUSED, NST>, a, blocks, lds, s);
}
template <bool FUSED, int NST, int PER>
hipError_t sw_launch_tiled(const StemWgradArgs& a, unsigned blocks, size_t lds, hipStream_t s) {
    return sw_launch_k(stem_wgrad_tiled_kernel<FUSED, NST, PER>, a, blocks, lds, s);
}
}  // namespace

bool p3d_stem_wgrad_ok(int kd, int kh, int kw, int Cin, int Cout, int sd, int sh, int sw, int Wo) {
    static const bool off = p3d_tune_env("P3D_STEM_ONEPASS") && atoi(p3d_tune_env("P3D_STEM_ONEPASS")) == 0;      // A/B runs
    return !off && kd == 1 && kh == 7 && kw == 7 && Cin == 3 && Cout == SW_CO && sd == 1 && sh == 2 && sw == 2 && Wo >= 2 && (Wo & 1) == 0 &&
           2 * (Wo - 1) + 7 <= SW_MAX_WP;
}
int p3d_stem_wgrad_max_blocks() { return 256; }
long p3d_stem_wgrad_part_floats() { return (long)p3d_stem_wgrad_max_blocks() * SW_ELEMS; }

hipError_t p3d_stem_wgrad(const StemWgradArgs& a0, int* nblocks, hipStream_t s) {
    StemWgradArgs a = a0;
    if (!nblocks) return hipErrorInvalidValue;
    if (!a.x4 || !a.dy || !a.part || !a.dw || (a.Wo & 1) || a.Wo < 2 || a.nimg < 1 || a.Ho < 1 || a.Hi < 1) return hipErrorInvalidValue;
    if (a.pad_h < 0 || a.pad_h > 6 || a.Wp > SW_MAX_WP || 2 * (a.Wo - 1) + 7 > a.Wp) return hipErrorInvalidValue;      // every patch inside the padded row
    if ((a.lddy & 1) || (a.fused && (a.ldy & 1))) return hipErrorInvalidValue;                                          // 8-byte loads
    if (a.fused && (!a.y || !a.scale || !a.shift || !a.mean || !a.invstd || !a.gamma || (a.batch && !a.coef))) return hipErrorInvalidValue;
    a.chunks_per_img = (a.Ho + SW_RC - 1) / SW_RC;
    const long long chunks = (long long)a.nimg * a.chunks_per_img;
    if (chunks > 0x7fffffff) return hipErrorInvalidValue;
    a.chunks = (int)chunks;
    // one block of 8 waves per CU, a contiguous run of chunks each (consecutive chunks share 5 of their 13 input rows)
    long long blocks = chunks < p3d_stem_wgrad_max_blocks() ? chunks : p3d_stem_wgrad_max_blocks();
    a.chunks_per_block = (int)((chunks + blocks - 1) / blocks);
    blocks = (chunks + a.chunks_per_block - 1) / a.chunks_per_block;
    a.pairs_per_wave = (SW_RC * (a.Wo >> 1) + SW_WAVES - 1) / SW_WAVES;
    a.slots_per_wave = (a.pairs_per_wave + SW_STAGES - 1) / SW_STAGES * SW_STAGES;
    a.lds_row = sw_lds_row(a.Wp);
    const size_t lds = sw_lds_bytes(a.Wp);
    const int nst = (SW_SLOTS * a.Wp + SW_THREADS - 1) / SW_THREADS;
    hipError_t e;
    static const bool no_tiled = p3d_tune_env("P3D_STEM_TILED") && atoi(p3d_tune_env("P3D_STEM_TILED")) == 0;      // A/B runs
    const long long npos = (long long)a.nimg * a.Ho * a.Wo;
    const bool small = npos * a.lddy * 4 < (1ll << 31) && (!a.fused || npos * a.ldy * 4 < (1ll << 31));      // buffer descriptors: 2 GB
    const bool tiles = !no_tiled && small && a.Ho % SW_RC == 0 && SW_RC * (a.Wo >> 1) == SW_WAVES * a.pairs_per_wave;
    const unsigned nb = (unsigned)blocks;
    if (tiles && a.pairs_per_wave == 14 && nst <= 3) e = a.fused ? sw_launch_tiled<true, 3, 14>(a, nb, lds, s) : sw_launch_tiled<false, 3, 14>(a, nb, lds, s);
    else if (tiles && a.pairs_per_wave == 28 && nst <= 6) e = a.fused ? sw_launch_tiled<true, 6, 28>(a, nb, lds, s) : sw_launch_tiled<false, 6, 28>(a, nb, lds, s);
    else if (nst <= 3) e = a.fused ? sw_launch<true, 3>(a, nb, lds, s) : sw_launch<false, 3>(a, nb, lds, s);
    else if (nst <= 6) e = a.fused ? sw_launch<true, 6>(a, nb, lds, s) : sw_launch<false, 6>(a, nb, lds, s);
    else return hipErrorInvalidValue;
    if (e != hipSuccess) return e;
    *nblocks = (int)blocks;
    return hipSuccess;
}

hipError_t p3d_stem_wgrad_fold(const float* part, int nblocks, float* dw, hipStream_t s) {
    if (!part || !dw || nblocks < 1 || nblocks > p3d_stem_wgrad_max_blocks()) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stem_wgrad_fold_kernel, dim3((SW_ELEMS + 63) / 64), dim3(64 * SW_FOLD_G), 0, s, part, nblocks, dw);
    return hipGetLastError();
}
