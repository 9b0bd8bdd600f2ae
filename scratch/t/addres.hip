#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "device_common.h"
#include "kernel_mc.h"
#include "kernel_intra.h"
__global__ void k(uint32_t *o, const uint32_t *in) { o[threadIdx.x] = add_res4(in[0], in[1], in[2 + threadIdx.x]); }
int main() {
    uint32_t h[2 + 4] = { 0x00000000u, 0x00030007u, 0xfffbffffu, 0x00050001u, 0x7fff8000u, 0x0100ff00u }, *d, *o, r[4];
    hipMalloc(&d, sizeof h); hipMalloc(&o, 16); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 4>>>(o, d); hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++) printf("%08x\n", r[i]);
    return 0;
}
