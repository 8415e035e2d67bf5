// Cross-block sums without floating-point atomics (results must be bit-reproducible: ADVICE round 1).
//
// Pattern: every block stores its partial result with WRITE-THROUGH stores (p3d_store_wt*), then calls p3d_last_block_wt();
// exactly one block -- the one whose arrival ticket is the last -- gets `true`, with every other block's partials visible, and
// folds them in BLOCK ORDER.  The protocol is the K-slice exchange of conv_igemm2.hip (cdna_hip_programming.md, Guideline 16):
// storing waves drain their stores, block barrier, one lane takes a relaxed ticket; the last arriver does ONE agent-scope
// acquire before anyone in its block reads.  The counter must be zero at launch; the last arriver re-zeroes it (scratch
// counters of p3d_stream_scratch start zeroed and are only used this way).
// Round 2 used plain stores + an agent-scope RELEASE per block: that release writes back the XCD's whole L2 (2-6 us), paid by
// every block of every reduction -- gn_stats over a 37 MB tensor took 71 us with ~1500 blocks, 30 us with write-through stores
// and 4x fewer blocks; the small GroupNorm backward 22 -> 18 us per launch.
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned p3d_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void p3d_store_wt4(float* base, size_t float_index, float4 v) {       // base: block-uniform
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
    const p3d_u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)(float_index * 4), 0, 16);                 // aux 16 = sc1
}
__device__ __forceinline__ void p3d_store_wt(float* base, size_t float_index, float v) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, (int)(float_index * 4), 0, 16);
}
typedef unsigned p3d_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void p3d_store_wt(double* base, size_t index, double v) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const p3d_u32x2 u = {(unsigned)b, (unsigned)(b >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(u, rs, (int)(index * 8), 0, 16);
}
__device__ __forceinline__ bool p3d_last_block_wt(unsigned* counter, unsigned nblocks, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == nblocks - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *counter = 0;
        }
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0;
}
