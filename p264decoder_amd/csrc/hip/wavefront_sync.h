// wavefront_sync.h - macroblock-row dependency tracking for the two raster-ordered stages
// (intra reconstruction and the loop filter).
//
// Both stages inherit the reference's raster order (decoder/decoder.c:502-593 and
// core/frame.c:497-642): macroblock (x,y) may start once (x-1,y) and (x+1,y-1) are complete.
// We run one workgroup per picture and hand its macroblock rows to the workgroup's
// wavefronts round-robin; row r publishes "columns < progress[r] are final" in LDS.  All waves
// of a workgroup sit on one CU and share its vector L1, so workgroup-scope release/acquire
// is all the ordering the global-memory pixel traffic needs - no agent-scope fences, no
// dependence on dispatch order or XCD placement.  Every spin is bounded.
#pragma once
#include "device_common.h"

#define ROW_WAVES      16            // wavefronts per picture workgroup (1024 threads)
#define MAX_MB_ROWS    512
#define SPIN_LIMIT     (1 << 22)
#ifndef WAIT_SLEEP
#define WAIT_SLEEP     32            // x64 cycles between polls: a polling wave must not eat the CU's scalar issue slots
#endif

struct RowSync {
    int progress[MAX_MB_ROWS];
};

__device__ __forceinline__ void rows_init(RowSync &s, int mb_h)
{
    for (int i = threadIdx.x; i < mb_h; i += blockDim.x) s.progress[i] = 0;
    __syncthreads();
}

// make this wave's global stores visible to the workgroup, then publish
__device__ __forceinline__ void row_publish(RowSync &s, int row, int cols_done)
{
    __hip_atomic_store(&s.progress[row], cols_done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// wait until row `row` has finished at least `need` columns; returns false on timeout
__device__ __forceinline__ bool row_wait(RowSync &s, int row, int need, int *status)
{
    int spins = 0;
    while (__hip_atomic_load(&s.progress[row], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < need) {
        __builtin_amdgcn_s_sleep(WAIT_SLEEP);
        if (++spins > SPIN_LIMIT) {
            if ((threadIdx.x & 63) == 0) atomicOr(status, 1);
            return false;
        }
    }
    return true;
}
