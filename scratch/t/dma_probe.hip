// round 5 probe: buffer_load_dwordx4 ... lds on gfx950 - where the data lands (M0 base + lane * 16), what the bounds check and a
// cleared EXEC bit do.  hipcc --offload-arch=gfx950 -O2 dma_probe.hip -o dma_probe && ./dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(i32x4 rs, uint32_t voff, uint32_t lds_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_base) : "memory");
}
__global__ void k(const uint8_t *in, uint32_t *out, uint32_t bytes)
{
    __shared__ __attribute__((aligned(16))) uint8_t t[8192];
    for (int i = threadIdx.x; i < 2048; i += 64) ((uint32_t *)t)[i] = 0xdeadbeefu;
    __syncthreads();
    i32x4 rs = { (int)(uintptr_t)in, (int)(((uintptr_t)in >> 32) & 0xffff), (int)bytes, 0x00020000 };
    rs.x = __builtin_amdgcn_readfirstlane(rs.x); rs.y = __builtin_amdgcn_readfirstlane(rs.y); rs.z = __builtin_amdgcn_readfirstlane(rs.z);
    const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)t);
    const int lane = threadIdx.x;
    // A: lane l loads source piece (63 - l) (reversed): destination must be lane-linear at base + 1040
    dma16(rs, (uint32_t)(63 - lane) * 16u, base + 1040u);
    // B: lanes 0..31 in range, 32..47 beyond the buffer (zeros expected), 48..63 masked off (untouched expected), at base + 4096
    uint32_t off = lane < 32 ? (uint32_t)lane * 16u : 0xffffff00u;
    if (lane < 48) dma16(rs, off, base + 4096u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = ((uint32_t *)t)[i];
}
int main()
{
    std::vector<uint8_t> h(1024);
    for (int i = 0; i < 1024; i++) h[i] = (uint8_t)(i / 16);           // piece number in every byte
    uint8_t *d; uint32_t *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 8192);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 1024u);
    std::vector<uint32_t> r(2048);
    hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) { uint32_t want = 0x01010101u * (uint32_t)(63 - l); for (int k2 = 0; k2 < 4; k2++) if (r[(1040 + l * 16) / 4 + k2] != want) bad++; }
    printf("A (reversed source, lane-linear destination at +1040): %s\n", bad ? "MISMATCH" : "ok");
    printf("A neighbours: before %08x after %08x\n", r[1040 / 4 - 1], r[(1040 + 1024) / 4]);
    int badB = 0;
    for (int l = 0; l < 64; l++) {
        uint32_t want = l < 32 ? 0x01010101u * (uint32_t)l : l < 48 ? 0u : 0xdeadbeefu;
        for (int k2 = 0; k2 < 4; k2++) if (r[(4096 + l * 16) / 4 + k2] != want) { if (!badB) printf("B lane %d: got %08x want %08x\n", l, r[(4096 + l * 16) / 4 + k2], want); badB++; }
    }
    printf("B (bounds check -> zeros, masked lanes untouched): %s\n", badB ? "MISMATCH" : "ok");
    return bad || badB;
}
