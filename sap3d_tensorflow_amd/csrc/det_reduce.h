// Cross-block sums without floating-point atomics (results must be bit-reproducible: ADVICE round 1).
//
// Pattern: every block stores its partial result with plain stores, then calls p3d_last_block(); exactly one block
// -- the one whose arrival ticket is the last -- gets `true`, with every other block's partials visible, and folds
// them in BLOCK ORDER.  The protocol is the split-K one of conv_igemm2.hip (cdna_hip_programming.md, Guideline 16):
// storing waves drain their stores, block barrier, one lane does ONE agent-scope release + ticket; the last arriver
// does ONE agent-scope acquire before anyone in its block reads.  The counter must be zero at launch; the last
// arriver re-zeroes it (scratch counters of p3d_stream_scratch start zeroed and are only used this way).
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ bool p3d_last_block(unsigned* counter, unsigned nblocks, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
