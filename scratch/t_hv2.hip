#include <hip/hip_runtime.h>
#include "../p264decoder_amd/csrc/hip/kernel_inter.h"
__global__ void k2(const uint32_t* win, uint32_t* out) {
  __shared__ uint32_t w[56];
  if (threadIdx.x < 56) w[threadIdx.x] = win[threadIdx.x];
  __syncthreads();
  out[threadIdx.x] = hv4(w, 2 + (threadIdx.x>>3), 2 + (threadIdx.x & 7));
}
