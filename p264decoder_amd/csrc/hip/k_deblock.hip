// k_deblock.hip - the loop filter's sample kernel (kernel_deblock.h: K4b) as a translation unit of its own, so that it can be
// compiled with the scheduling strategy that suits it (build.py: -mllvm -amdgpu-sched-strategy=max-ilp; see kernel_deblock.h).
// p264hip.hip declares the kernel and launches it.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "p264hip.h"
#include "device_common.h"
#include "kernel_deblock.h"

#ifdef EXPD_STAMPS
// diagnostic build (scratch/r4_stamps.sh): the clock stamps of one wavefront, for p264hip_sync to write out
extern "C" int p264hip_db_stamps_read(unsigned long long *h, size_t bytes)
{
    return hipMemcpyFromSymbol(h, HIP_SYMBOL(g_db_stamps), bytes) == hipSuccess ? 0 : -1;
}
#endif
