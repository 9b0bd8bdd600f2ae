// Issue rate of vector instructions on gfx950, measured: every wavefront runs ITER x 8 independent instructions of one kind,
// 8 wavefronts per SIMD.  Prints cycles per wave-instruction and SIMD (a full-rate instruction: 4).
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITER 4096
#define OPS8(stmt) stmt(0) stmt(1) stmt(2) stmt(3) stmt(4) stmt(5) stmt(6) stmt(7)
#define KERNEL(name, asmline)                                                                    \
__global__ __launch_bounds__(256) void name(uint32_t *out, uint32_t seed)                        \
{                                                                                                \
    uint32_t r[8], b = seed + threadIdx.x, c = seed * 3 + 1;                                     \
    for (int i = 0; i < 8; i++) r[i] = seed + i + threadIdx.x;                                   \
    uint64_t w[8]; for (int i = 0; i < 8; i++) w[i] = r[i];                                      \
    for (int it = 0; it < ITER; it++) {                                                          \
        asmline                                                                                  \
    }                                                                                            \
    uint32_t s = 0; for (int i = 0; i < 8; i++) s += r[i] + (uint32_t)w[i];                      \
    if (s == 0x12345678u) out[0] = s;                                                            \
}
#define A1(op) OPS8(A1_##op)
#define S(i, text) asm volatile(text : "+v"(r[i]) : "v"(b), "v"(c));
#define S64(i, text) asm volatile(text : "+v"(w[i]) : "v"(b), "v"(c));
#define X_add(i)    S(i, "v_add_u32 %0, %0, %1")
#define X_mullo(i)  S(i, "v_mul_lo_u32 %0, %0, %1")
#define X_mulhi(i)  S(i, "v_mul_hi_u32 %0, %0, %1")
#define X_mul24(i)  S(i, "v_mul_i32_i24 %0, %0, %1")
#define X_mad24(i)  S(i, "v_mad_i32_i24 %0, %0, %1, %2")
#define X_madu24(i) S(i, "v_mad_u32_u24 %0, %0, %1, %2")
#define X_lshladd(i) S(i, "v_lshl_add_u32 %0, %0, 2, %1")
#define X_add3(i)   S(i, "v_add3_u32 %0, %0, %1, %2")
#define X_perm(i)   S(i, "v_perm_b32 %0, %0, %1, %2")
#define X_align(i)  S(i, "v_alignbyte_b32 %0, %0, %1, 1")
#define X_dot4(i)   S(i, "v_dot4c_i32_i8 %0, %1, %2")
#define X_pkadd(i)  S(i, "v_pk_add_u16 %0, %0, %1")
#define X_pkmul(i)  S(i, "v_pk_mul_lo_u16 %0, %0, %1")
#define X_pkmad(i)  S(i, "v_pk_mad_i16 %0, %0, %1, %2")
#define X_pkmax(i)  S(i, "v_pk_max_i16 %0, %0, %1")
#define X_pkashr(i) S(i, "v_pk_ashrrev_i16 %0, 1, %0")
#define X_sdwa(i)   S(i, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
#define X_bfe(i)    S(i, "v_bfe_u32 %0, %0, 3, 8")
#define X_med3(i)   S(i, "v_med3_i32 %0, %0, %1, %2")
#define X_sad(i)    S(i, "v_sad_u8 %0, %0, %1, %2")
#define X_cndmask(i) S(i, "v_cndmask_b32 %0, %0, %1, vcc")
#define X_bitop3(i) S(i, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
#define X_ashrpk(i) S(i, "v_ashr_pk_u8_i32 %0, %0, %1, %2")
#define X_lshladd64(i) S64(i, "v_lshl_add_u64 %0, %0, 2, %0")
#define X_mad64(i)  S64(i, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
#define X_mov(i)    S(i, "v_mov_b32 %0, %1")
#define X_dpp(i)    S(i, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
#define X_rdlane(i) asm volatile("v_readfirstlane_b32 s20, %0\n v_add_u32 %0, s20, %0" : "+v"(r[i]) : : "s20");
#define DEF(n) KERNEL(k_##n, OPS8(X_##n))
DEF(add) DEF(mullo) DEF(mulhi) DEF(mul24) DEF(mad24) DEF(madu24) DEF(lshladd) DEF(add3) DEF(perm) DEF(align) DEF(dot4) DEF(pkadd) DEF(pkmul)
DEF(pkmad) DEF(pkmax) DEF(pkashr) DEF(sdwa) DEF(bfe) DEF(med3) DEF(sad) DEF(cndmask) DEF(bitop3) DEF(ashrpk) DEF(lshladd64) DEF(mad64) DEF(mov) DEF(dpp)
#define X_sub(i)    S(i, "v_sub_u32 %0, %0, %1")
#define X_and(i)    S(i, "v_and_b32 %0, %0, %1")
#define X_or(i)     S(i, "v_or_b32 %0, %0, %1")
#define X_xor(i)    S(i, "v_xor_b32 %0, %0, %1")
#define X_shl(i)    S(i, "v_lshlrev_b32 %0, 1, %0")
#define X_shr(i)    S(i, "v_lshrrev_b32 %0, 1, %0")
#define X_shrv(i)   S(i, "v_lshrrev_b32 %0, %1, %0")
#define X_ashr(i)   S(i, "v_ashrrev_i32 %0, 1, %0")
#define X_min(i)    S(i, "v_min_i32 %0, %0, %1")
#define X_max(i)    S(i, "v_max_u32 %0, %0, %1")
#define X_andor(i)  S(i, "v_and_or_b32 %0, %0, %1, %2")
#define X_or3(i)    S(i, "v_or3_b32 %0, %0, %1, %2")
#define X_bfi(i)    S(i, "v_bfi_b32 %0, %0, %1, %2")
#define X_addu16(i) S(i, "v_add_u16 %0, %0, %1")
#define X_pksub(i)  S(i, "v_pk_sub_i16 %0, %0, %1")
#define X_pkmin(i)  S(i, "v_pk_min_i16 %0, %0, %1")
#define X_pkshl(i)  S(i, "v_pk_lshlrev_b16 %0, 1, %0")
#define X_cmp(i)    asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[i]), "v"(b) : "vcc");
#define X_cmpcnd(i) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(b), "v"(c) : "vcc");
#define X_cnd64(i)  asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(b), "v"(c) : "s20", "s21");
#define X_cmps(i)   asm volatile("v_cmp_lt_u32_e64 s[22:23], %0, %1" : : "v"(r[i]), "v"(b) : "s22", "s23");
#define X_mbcnt(i)  S(i, "v_mbcnt_lo_u32_b32 %0, -1, %0")
#define X_sat(i)    S(i, "v_add_u16 %0, %0, %1 clamp")
#define X_addco(i)  asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(r[i]) : "v"(b) : "vcc");
#define X_mov64(i)  S64(i, "v_mov_b64 %0, %0")
#define X_pkmov(i)  S64(i, "v_pk_mov_b32 %0, %0, %0")
#define X_ubfe(i)   S(i, "v_bfe_i32 %0, %0, 3, 8")
#define X_sub16sd(i) S(i, "v_sub_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1")
DEF(sub) DEF(and) DEF(or) DEF(xor) DEF(shl) DEF(shr) DEF(shrv) DEF(ashr) DEF(min) DEF(max) DEF(andor) DEF(or3) DEF(bfi) DEF(addu16) DEF(pksub) DEF(pkmin) DEF(pkshl)
DEF(cmp) DEF(cmpcnd) DEF(cnd64) DEF(cmps) DEF(mbcnt) DEF(sat) DEF(addco) DEF(mov64) DEF(pkmov) DEF(ubfe) DEF(sub16sd)
template <typename K> static void run(const char *name, K k, uint32_t *d, int cus, double mhz, int per = 1)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = cus * 8;                                   // 8 workgroups of 4 wavefronts per CU = 8 wavefronts per SIMD
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = 8.0 * ITER * 8 * per;          // 8 wavefronts x ITER x 8 instructions
    printf("%-12s %8.3f ms  %6.2f cycles per wave-instruction and SIMD (at %.0f MHz)\n", name, ms, ms * 1e-3 * mhz * 1e6 / instr_per_simd, mhz);
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double mhz = p.clockRate / 1000.0;
    uint32_t *d; hipMalloc(&d, 4096);
    printf("%s, %d CUs, %.0f MHz\n", p.name, p.multiProcessorCount, mhz);
#define R(n) run(#n, k_##n, d, p.multiProcessorCount, mhz);
    R(add) R(mullo) R(mulhi) R(mul24) R(mad24) R(madu24) R(lshladd) R(add3) R(perm) R(align) R(dot4) R(pkadd) R(pkmul) R(pkmad) R(pkmax) R(pkashr)
    R(sdwa) R(bfe) R(med3) R(sad) R(cndmask) R(bitop3) R(ashrpk) R(lshladd64) R(mad64) R(mov) R(dpp)
    R(sub) R(and) R(or) R(xor) R(shl) R(shr) R(shrv) R(ashr) R(min) R(max) R(andor) R(or3) R(bfi) R(addu16) R(pksub) R(pkmin) R(pkshl)
    R(cmp) R(cmpcnd) R(cnd64) R(cmps) R(mbcnt) R(sat) R(addco) R(mov64) R(pkmov) R(ubfe) R(sub16sd)
    R(add) R(mullo)
    return 0;
}
