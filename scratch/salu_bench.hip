// SALU issue-rate probe: N waves per CU, each issuing a long stream of independent s_add.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_salu(int *out, int iters)
{
    int a = blockIdx.x, b = 1, c = 2, d = 3;
    for (int i = 0; i < iters; i++) {
        asm volatile(
            "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
            "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
            "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
            "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
            : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
    }
    if (threadIdx.x == 0 && a + b + c + d == 12345) out[0] = a;
}
__global__ void k_valu(int *out, int iters)
{
    int a = threadIdx.x, b = 1, c = 2, d = 3;
    for (int i = 0; i < iters; i++) {
        asm volatile(
            "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
            "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
            "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
            "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n"
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    if (a + b + c + d == 12345) out[0] = a;
}
int main()
{
    int *d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int kind = 0; kind < 2; kind++)
        for (int threads = 64; threads <= 1024; threads *= 2) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_salu, dim3(256), dim3(threads), 0, 0, d, iters);
                else hipLaunchKernelGGL(k_valu, dim3(256), dim3(threads), 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("%s waves/CU %2d: %.3f ms -> %.2f cycles per instr per wave at 2.4 GHz, %.2f instr/cycle/CU\n", kind ? "VALU" : "SALU",
                                threads / 64, ms, ms * 1e-3 * 2.4e9 / (iters * 16.0), (threads / 64) * iters * 16.0 / (ms * 1e-3 * 2.4e9));
            }
        }
    fflush(stdout);
    return 0;
}
