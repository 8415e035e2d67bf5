// Epilogue of the convolution kernels (conv_igemm2.hip): a finished output tile sits in LDS
// (`tile`, row stride BN + 4 floats); this writes it out row-wise as float4 -- bias, optional accumulate, BatchNorm
// statistics partials and the gated form of the fused-BatchNorm input gradients (BnGate).
// Every sum is folded in a fixed order: results are bit-reproducible.
#pragma once
#include "p3d_kernels.h"

// The per-class part of a launch as the kernel body sees it: from the kernel arguments themselves (one class) or from the
// class table of a grouped launch
// The tap table of a launch sits in the kernel arguments.  Read through a plain pointer it is a VECTOR load (hipcc cannot tell
// where the pointer came from), and a vector load's result, used while LDS-DMA is in flight, makes hipcc drain everything
// (s_waitcnt vmcnt(0)): the prologue of every conv launch waited for its first operand tiles before it could issue the
// next ones.  Through a constant-address-space pointer the taps are scalar loads: no vector counter involved.
typedef const P3dTap __attribute__((address_space(4))) P3dKTap;
__device__ __forceinline__ P3dKTap* p3d_kernarg_taps(size_t byte_offset) {
    typedef const char __attribute__((address_space(4))) kchar_t;
    return (P3dKTap*)((kchar_t*)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset);
}
__device__ __forceinline__ P3dTap p3d_ktap(P3dKTap* taps, int i) {
    P3dTap t;
    t.dd = taps[i].dd; t.dh = taps[i].dh; t.dw = taps[i].dw; t.widx = taps[i].widx;
    return t;
}
struct Geo {
    int Gd, Gh, Gw;
    P3dFastDiv fGd, fGh, fGw;
    int ood, ooh, oow, stat_base, ntaps;
    P3dKTap* taps;
    const float* w; const float* bias; float* y; float* statpart;      // a grouped launch may also carry sibling convs (ST_B)
    int nsplit; float* slab; unsigned* cnt;                             // K-slices of this launch / class and their scratch
};

__device__ __forceinline__ float4 shfl_xor4(float4 v, int o) {
    return make_float4(__shfl_xor(v.x, o), __shfl_xor(v.y, o), __shfl_xor(v.z, o), __shfl_xor(v.w, o));
}

// bias4: this thread's four bias values (igemm_bias_prefetch, issued at kernel entry: read here, behind the main loop and a
// barrier, it is a dependent trip to memory in the epilogue of every biased conv)
template <int BN>
__device__ __forceinline__ float4 igemm_bias_prefetch(const float* bias, int n0, int Nc) {
    const int col = n0 + ((int)threadIdx.x % (BN / 4)) * 4;
    return (bias && col < Nc) ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
}
// tile: [BM][BN + 4] floats in LDS, complete and visible to the whole block (a barrier has passed); sred: >= 2*4*BN*2 floats of
// LDS behind it; rowIdx[r]: output row of tile row r, -1 past M; 256 threads
template <int BM, int BN>
__device__ __forceinline__ void igemm_tile_epilogue(const IgemmArgs& p, const Geo& geo, float* tile, float* sred, const int* rowIdx,
                                                    const int mt, const int n0, const long long M, const float4 bias4) {
    constexpr int LDT = BN + 4;
    constexpr int F4R = BN / 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (p.ngate > 0) {
        // -- gated output (input gradient through fused BatchNorm + ReLU pairs): every lane owns ONE float4 column group
        //    (256 % F4R == 0) and walks rows, so the per-channel (sum g, sum g*xhat) accumulate in registers; lanes of a
        //    wave that share the column group are folded by shuffles, the four waves through LDS, all in a fixed order.
        const int c4 = (tid % F4R) * 4, col = n0 + c4;
        const bool cok = col < p.Nc;
        float4 sc[2], sh[2], mu[2], iv[2], s[2], sx[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            s[q] = make_float4(0.f, 0.f, 0.f, 0.f); sx[q] = s[q]; sc[q] = s[q]; sh[q] = s[q]; mu[q] = s[q]; iv[q] = s[q];
            if (q < p.ngate && cok) {
                sc[q] = *reinterpret_cast<const float4*>(p.gate[q].scale + col); sh[q] = *reinterpret_cast<const float4*>(p.gate[q].shift + col);
                mu[q] = *reinterpret_cast<const float4*>(p.gate[q].mean + col); iv[q] = *reinterpret_cast<const float4*>(p.gate[q].invstd + col);
            }
        }
#pragma unroll 2
        for (int r = tid / F4R; r < BM; r += 256 / F4R) {
            const int ro = rowIdx[r];
            if (ro < 0 || !cok) continue;
            float4 v = *reinterpret_cast<const float4*>(tile + r * LDT + c4);
            v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
            if (p.accum) { const float4 o = *reinterpret_cast<const float4*>(geo.y + (long long)ro * p.ldy + col); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            if (p.raw_store) *reinterpret_cast<float4*>(geo.y + (long long)ro * p.ldy + col) = v;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q >= p.ngate) break;
                const float4 y4 = *reinterpret_cast<const float4*>(p.gate[q].y + (long long)ro * p.gate[q].ldy + col);
                float4 g;
                g.x = fmaf(sc[q].x, y4.x, sh[q].x) > 0.f ? v.x : 0.f; g.y = fmaf(sc[q].y, y4.y, sh[q].y) > 0.f ? v.y : 0.f;
                g.z = fmaf(sc[q].z, y4.z, sh[q].z) > 0.f ? v.z : 0.f; g.w = fmaf(sc[q].w, y4.w, sh[q].w) > 0.f ? v.w : 0.f;
                *reinterpret_cast<float4*>(p.gate[q].out + (long long)ro * p.gate[q].ldo + col) = g;
                s[q].x += g.x; s[q].y += g.y; s[q].z += g.z; s[q].w += g.w;
                sx[q].x = fmaf(g.x, (y4.x - mu[q].x) * iv[q].x, sx[q].x); sx[q].y = fmaf(g.y, (y4.y - mu[q].y) * iv[q].y, sx[q].y);
                sx[q].z = fmaf(g.z, (y4.z - mu[q].z) * iv[q].z, sx[q].z); sx[q].w = fmaf(g.w, (y4.w - mu[q].w) * iv[q].w, sx[q].w);
            }
        }
        __syncthreads();                        // everyone is done with the tile: sred may be reused
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q >= p.ngate) break;
#pragma unroll
            for (int o = F4R; o < 64; o <<= 1) {
                const float4 a = shfl_xor4(s[q], o), b = shfl_xor4(sx[q], o);
                s[q].x += a.x; s[q].y += a.y; s[q].z += a.z; s[q].w += a.w;
                sx[q].x += b.x; sx[q].y += b.y; sx[q].z += b.z; sx[q].w += b.w;
            }
            if (lane < F4R) {
                float* d = sred + ((q * 4 + wave) * BN + c4) * 2;
                d[0] = s[q].x; d[1] = sx[q].x; d[2] = s[q].y; d[3] = sx[q].y; d[4] = s[q].z; d[5] = sx[q].z; d[6] = s[q].w; d[7] = sx[q].w;
            }
        }
        __syncthreads();
        if (tid < BN && (n0 + tid) < p.Nc) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q >= p.ngate) break;
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { t1 += sred[((q * 4 + w) * BN + tid) * 2]; t2 += sred[((q * 4 + w) * BN + tid) * 2 + 1]; }
                float* dst = p.gate[q].part + ((size_t)mt * p.Nc + n0 + tid) * 2;
                dst[0] = t1; dst[1] = t2;
            }
        }
        return;
    }

    // -- output rows: bias, optional accumulate, row-wise float4 stores; the stored values go back to the tile for the
    //    statistics pass --------------------------------------------------------------------------------------------
    const bool want_stats = geo.statpart != nullptr;
#pragma unroll 4
    for (int i = tid; i < BM * F4R; i += 256) {
        const int r = i / F4R, c4 = (i - r * F4R) * 4;
        const int ro = rowIdx[r];
        const int col = n0 + c4;
        if (ro < 0 || col >= p.Nc) continue;
        float4 v = *reinterpret_cast<const float4*>(tile + r * LDT + c4);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;          // (every position of this thread sits in the same four columns: 256 % F4R == 0)
        if (want_stats) *reinterpret_cast<float4*>(tile + r * LDT + c4) = v;
        float* dst = geo.y + (long long)ro * p.ldy + col;
        if (p.accum) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *reinterpret_cast<float4*>(dst) = v;
    }
    if (want_stats) {
        // per-channel (sum, sumsq) over this tile's valid rows: 4 row groups x BN columns, folded in a fixed order
        __syncthreads();
        constexpr int RG = 256 / BN, RPG = BM / RG;
        const int col = tid % BN, rg = tid / BN;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
        for (int r = rg * RPG; r < (rg + 1) * RPG; ++r)
            if (rowIdx[r] >= 0) { const float v = tile[r * LDT + col]; s1 += v; s2 = fmaf(v, v, s2); }
        sred[(rg * BN + col) * 2] = s1; sred[(rg * BN + col) * 2 + 1] = s2;
        __syncthreads();
        if (tid < BN && (n0 + tid) < p.Nc) {
            float t1 = sred[tid * 2], t2 = sred[tid * 2 + 1];
#pragma unroll
            for (int g = 1; g < RG; ++g) { t1 += sred[(g * BN + tid) * 2]; t2 += sred[(g * BN + tid) * 2 + 1]; }
            float* dst = geo.statpart + ((size_t)(geo.stat_base + mt) * p.Nc + n0 + tid) * 2;
            dst[0] = t1; dst[1] = t2;
        }
    }
}
