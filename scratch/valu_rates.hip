// VALU issue-rate probe: which of the instructions the kernels live on run at the full rate?  16 waves per CU, long streams of
// independent instructions; rates relative to v_add_u32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define BODY(ins) asm volatile(REP4(REP4(ins)) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc", "s10", "s11", "s12", "s13");
#define KERNEL(name, ins) __global__ void name(int *out, int iters) { \
    int a = threadIdx.x, b = 1, c = 2, d = 3, e = threadIdx.x * 7, f = 0x01020304; \
    for (int i = 0; i < iters; i++) { BODY(ins) } \
    if (a + b + c + d == 12345) out[0] = a; }
KERNEL(k_add,   "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
KERNEL(k_fma,   "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
KERNEL(k_perm,  "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n")
KERNEL(k_align, "v_alignbyte_b32 %0, %0, %4, 1\n v_alignbyte_b32 %1, %1, %4, 2\n v_alignbyte_b32 %2, %2, %4, 3\n v_alignbyte_b32 %3, %3, %4, 1\n")
KERNEL(k_dot4,  "v_dot4_i32_i8 %0, %4, %5, %0\n v_dot4_i32_i8 %1, %4, %5, %1\n v_dot4_i32_i8 %2, %4, %5, %2\n v_dot4_i32_i8 %3, %4, %5, %3\n")
KERNEL(k_dot4c, "v_dot4c_i32_i8 %0, %4, %5\n v_dot4c_i32_i8 %1, %4, %5\n v_dot4c_i32_i8 %2, %4, %5\n v_dot4c_i32_i8 %3, %4, %5\n")
KERNEL(k_pkadd, "v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4\n")
KERNEL(k_pkmax, "v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_max_i16 %3, %3, %4\n")
KERNEL(k_pkmad, "v_pk_mad_i16 %0, %0, %4, %5\n v_pk_mad_i16 %1, %1, %4, %5\n v_pk_mad_i16 %2, %2, %4, %5\n v_pk_mad_i16 %3, %3, %4, %5\n")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n")
KERNEL(k_mad24, "v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %4, %5\n v_mad_i32_i24 %2, %2, %4, %5\n v_mad_i32_i24 %3, %3, %4, %5\n")
KERNEL(k_lshla, "v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4\n")
KERNEL(k_bfi,   "v_bfi_b32 %0, %0, %4, %5\n v_bfi_b32 %1, %1, %4, %5\n v_bfi_b32 %2, %2, %4, %5\n v_bfi_b32 %3, %3, %4, %5\n")
KERNEL(k_add3,  "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5\n")
KERNEL(k_ashrpk,"v_ashr_pk_u8_i32 %0, %0, %4, %5\n v_ashr_pk_u8_i32 %1, %1, %4, %5\n v_ashr_pk_u8_i32 %2, %2, %4, %5\n v_ashr_pk_u8_i32 %3, %3, %4, %5\n")
KERNEL(k_dpp,   "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_cnd,   "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
KERNEL(k_sad,   "v_sad_u8 %0, %0, %4, %5\n v_sad_u8 %1, %1, %4, %5\n v_sad_u8 %2, %2, %4, %5\n v_sad_u8 %3, %3, %4, %5\n")

KERNEL(k_and,   "v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n")
KERNEL(k_lshl,  "v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n")
KERNEL(k_ashr,  "v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 3, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 3, %3\n")
KERNEL(k_sub,   "v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4\n")
KERNEL(k_maxi,  "v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4\n")
KERNEL(k_max3,  "v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %1, %1, %4, %5\n v_max3_i32 %2, %2, %4, %5\n v_max3_i32 %3, %3, %4, %5\n")
KERNEL(k_med3,  "v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %4, %5\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %4, %5\n")
KERNEL(k_andor, "v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5\n")
KERNEL(k_bfe,   "v_bfe_u32 %0, %0, 8, 8\n v_bfe_u32 %1, %1, 8, 8\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n")
KERNEL(k_lshlor,"v_lshl_or_b32 %0, %0, 8, %4\n v_lshl_or_b32 %1, %1, 8, %4\n v_lshl_or_b32 %2, %2, 8, %4\n v_lshl_or_b32 %3, %3, 8, %4\n")
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4\n")
KERNEL(k_cnd64, "v_cndmask_b32_e64 %0, %0, %4, s[10:11]\n v_cndmask_b32_e64 %1, %1, %4, s[10:11]\n v_cndmask_b32_e64 %2, %2, %4, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]\n")
KERNEL(k_cndb,  "v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n")
KERNEL(k_cmp,   "v_cmp_lt_i32 vcc, %0, %4\n v_cmp_lt_i32 vcc, %1, %4\n v_cmp_lt_i32 vcc, %2, %4\n v_cmp_lt_i32 vcc, %3, %4\n")
KERNEL(k_cmp64, "v_cmp_lt_i32_e64 s[10:11], %0, %4\n v_cmp_lt_i32_e64 s[12:13], %1, %4\n v_cmp_lt_i32_e64 s[10:11], %2, %4\n v_cmp_lt_i32_e64 s[12:13], %3, %4\n")
KERNEL(k_pksub, "v_pk_sub_i16 %0, %0, %4\n v_pk_sub_i16 %1, %1, %4\n v_pk_sub_i16 %2, %2, %4\n v_pk_sub_i16 %3, %3, %4\n")
KERNEL(k_pkashr,"v_pk_ashrrev_i16 %0, 3, %0 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %1, 3, %1 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %2, 3, %2 op_sel_hi:[0,1]\n v_pk_ashrrev_i16 %3, 3, %3 op_sel_hi:[0,1]\n")
KERNEL(k_bitop3,"v_bitop3_b32 %0, %0, %4, %5 bitop3:0x80\n v_bitop3_b32 %1, %1, %4, %5 bitop3:0x80\n v_bitop3_b32 %2, %2, %4, %5 bitop3:0x80\n v_bitop3_b32 %3, %3, %4, %5 bitop3:0x80\n")
KERNEL(k_sdwa,  "v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n")
KERNEL(k_mov,   "v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n")
KERNEL(k_add16, "v_add_u16 %0, %0, %4\n v_add_u16 %1, %1, %4\n v_add_u16 %2, %2, %4\n v_add_u16 %3, %3, %4\n")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4\n")

KERNEL(k_pair32, "v_cmp_lt_i32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_i32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n")
KERNEL(k_pair64, "v_cmp_lt_i32_e64 s[10:11], %0, %4\n v_cndmask_b32_e64 %0, %0, %5, s[10:11]\n v_cmp_lt_i32_e64 s[12:13], %1, %4\n v_cndmask_b32_e64 %1, %1, %5, s[12:13]\n")
KERNEL(k_cndvcc64, "v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n")
KERNEL(k_addci,  "v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n")
KERNEL(k_or,     "v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4\n")
KERNEL(k_xor,    "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n")
KERNEL(k_lshr,   "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n")
KERNEL(k_minu,   "v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4\n")
KERNEL(k_sub16,  "v_sub_u16 %0, %0, %4\n v_sub_u16 %1, %1, %4\n v_sub_u16 %2, %2, %4\n v_sub_u16 %3, %3, %4\n")
KERNEL(k_max16,  "v_max_i16 %0, %0, %4\n v_max_i16 %1, %1, %4\n v_max_i16 %2, %2, %4\n v_max_i16 %3, %3, %4\n")
KERNEL(k_ashr16, "v_ashrrev_i16 %0, 3, %0\n v_ashrrev_i16 %1, 3, %1\n v_ashrrev_i16 %2, 3, %2\n v_ashrrev_i16 %3, 3, %3\n")
KERNEL(k_lshl16, "v_lshlrev_b16 %0, 3, %0\n v_lshlrev_b16 %1, 3, %1\n v_lshlrev_b16 %2, 3, %2\n v_lshlrev_b16 %3, 3, %3\n")
KERNEL(k_mac,    "v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n")
KERNEL(k_addf,   "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
KERNEL(k_cvtpk,  "v_cvt_pk_u8_f32 %0, %0, %4, %5\n v_cvt_pk_u8_f32 %1, %1, %4, %5\n v_cvt_pk_u8_f32 %2, %2, %4, %5\n v_cvt_pk_u8_f32 %3, %3, %4, %5\n")
KERNEL(k_msad,   "v_msad_u8 %0, %0, %4, %5\n v_msad_u8 %1, %1, %4, %5\n v_msad_u8 %2, %2, %4, %5\n v_msad_u8 %3, %3, %4, %5\n")
KERNEL(k_add32e64, "v_add_u32_e64 %0, %0, %4\n v_add_u32_e64 %1, %1, %4\n v_add_u32_e64 %2, %2, %4\n v_add_u32_e64 %3, %3, %4\n")

KERNEL(k_pkaddh, "v_pk_add_f16 %0, %0, %4\n v_pk_add_f16 %1, %1, %4\n v_pk_add_f16 %2, %2, %4\n v_pk_add_f16 %3, %3, %4\n")
KERNEL(k_pkaddhc,"v_pk_add_f16 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1] clamp\n v_pk_add_f16 %1, %1, %4 neg_lo:[0,1] neg_hi:[0,1] clamp\n v_pk_add_f16 %2, %2, %4 neg_lo:[0,1] neg_hi:[0,1] clamp\n v_pk_add_f16 %3, %3, %4 neg_lo:[0,1] neg_hi:[0,1] clamp\n")
KERNEL(k_pkmulh, "v_pk_mul_f16 %0, %0, %4\n v_pk_mul_f16 %1, %1, %4\n v_pk_mul_f16 %2, %2, %4\n v_pk_mul_f16 %3, %3, %4\n")
KERNEL(k_pkfmah, "v_pk_fma_f16 %0, %0, %4, %5\n v_pk_fma_f16 %1, %1, %4, %5\n v_pk_fma_f16 %2, %2, %4, %5\n v_pk_fma_f16 %3, %3, %4, %5\n")
KERNEL(k_pkmaxh, "v_pk_max_f16 %0, %0, %4\n v_pk_max_f16 %1, %1, %4\n v_pk_max_f16 %2, %2, %4\n v_pk_max_f16 %3, %3, %4\n")
KERNEL(k_pkminh, "v_pk_min_f16 %0, %0, %4\n v_pk_min_f16 %1, %1, %4\n v_pk_min_f16 %2, %2, %4\n v_pk_min_f16 %3, %3, %4\n")
KERNEL(k_addh,   "v_add_f16 %0, %0, %4\n v_add_f16 %1, %1, %4\n v_add_f16 %2, %2, %4\n v_add_f16 %3, %3, %4\n")
KERNEL(k_fmah,   "v_fma_f16 %0, %0, %4, %5\n v_fma_f16 %1, %1, %4, %5\n v_fma_f16 %2, %2, %4, %5\n v_fma_f16 %3, %3, %4, %5\n")
KERNEL(k_maxf,   "v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n")
KERNEL(k_med3f,  "v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5\n")
KERNEL(k_floorf, "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n")
KERNEL(k_mulf,   "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n")
KERNEL(k_cvtu8,  "v_cvt_f32_ubyte1 %0, %4\n v_cvt_f32_ubyte2 %1, %4\n v_cvt_f32_ubyte3 %2, %4\n v_cvt_f32_ubyte0 %3, %4\n")
KERNEL(k_pkmul16,"v_pk_mul_lo_u16 %0, %0, %4\n v_pk_mul_lo_u16 %1, %1, %4\n v_pk_mul_lo_u16 %2, %2, %4\n v_pk_mul_lo_u16 %3, %3, %4\n")
KERNEL(k_max16u, "v_max_u16 %0, %0, %4\n v_max_u16 %1, %1, %4\n v_max_u16 %2, %2, %4\n v_max_u16 %3, %3, %4\n")
KERNEL(k_mad16,  "v_mad_u16 %0, %0, %4, %5\n v_mad_u16 %1, %1, %4, %5\n v_mad_u16 %2, %2, %4, %5\n v_mad_u16 %3, %3, %4, %5\n")
KERNEL(k_mul16,  "v_mul_lo_u16 %0, %0, %4\n v_mul_lo_u16 %1, %1, %4\n v_mul_lo_u16 %2, %2, %4\n v_mul_lo_u16 %3, %3, %4\n")
typedef void (*kern_t)(int *, int);
int main()
{
    int *d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    struct { const char *n; kern_t k; } ks[] = { {"v_add_u32", k_add}, {"v_fma_f32", k_fma}, {"v_perm_b32", k_perm}, {"v_alignbyte_b32", k_align}, {"v_dot4_i32_i8", k_dot4},
        {"v_dot4c_i32_i8", k_dot4c}, {"v_pk_add_u16", k_pkadd}, {"v_pk_max_i16", k_pkmax}, {"v_pk_mad_i16", k_pkmad}, {"v_mul_lo_u32", k_mullo}, {"v_mad_i32_i24", k_mad24},
        {"v_lshl_add_u32", k_lshla}, {"v_bfi_b32", k_bfi}, {"v_add3_u32", k_add3}, {"v_ashr_pk_u8_i32", k_ashrpk}, {"v_mov_b32_dpp", k_dpp}, {"v_cndmask_b32", k_cnd}, {"v_sad_u8", k_sad}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_lshl}, {"v_ashrrev_i32", k_ashr}, {"v_sub_u32", k_sub}, {"v_max_i32", k_maxi}, {"v_max3_i32", k_max3}, {"v_med3_i32", k_med3}, {"v_and_or_b32", k_andor}, {"v_bfe_u32", k_bfe}, {"v_lshl_or_b32", k_lshlor}, {"v_mul_u32_u24", k_mul24}, {"v_cndmask_e64 sgpr", k_cnd64}, {"v_cndmask vcc nodep", k_cndb}, {"v_cmp vcc", k_cmp}, {"v_cmp_e64 sgpr", k_cmp64}, {"v_pk_sub_i16", k_pksub}, {"v_pk_ashrrev_i16", k_pkashr}, {"v_bitop3_b32", k_bitop3}, {"v_add_u32_sdwa", k_sdwa}, {"v_mov_b32", k_mov}, {"v_add_u16", k_add16}, {"v_add_co_u32", k_addco}, {"cmp+cnd e32 (per 4)", k_pair32}, {"cmp+cnd e64 (per 4)", k_pair64}, {"v_cndmask_e64 vcc", k_cndvcc64}, {"v_addc_co_u32", k_addci}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_lshrrev_b32", k_lshr}, {"v_min_u32", k_minu}, {"v_sub_u16", k_sub16}, {"v_max_i16", k_max16}, {"v_ashrrev_i16", k_ashr16}, {"v_lshlrev_b16", k_lshl16}, {"v_mac_f32", k_mac}, {"v_add_f32", k_addf}, {"v_cvt_pk_u8_f32", k_cvtpk}, {"v_msad_u8", k_msad}, {"v_add_u32_e64", k_add32e64}, {"v_pk_add_f16", k_pkaddh}, {"v_pk_add_f16 neg clamp", k_pkaddhc}, {"v_pk_mul_f16", k_pkmulh}, {"v_pk_fma_f16", k_pkfmah}, {"v_pk_max_f16", k_pkmaxh}, {"v_pk_min_f16", k_pkminh}, {"v_add_f16", k_addh}, {"v_fma_f16", k_fmah}, {"v_max_f32", k_maxf}, {"v_med3_f32", k_med3f}, {"v_floor_f32", k_floorf}, {"v_mul_f32", k_mulf}, {"v_cvt_f32_ubyteN", k_cvtu8}, {"v_pk_mul_lo_u16", k_pkmul16}, {"v_max_u16", k_max16u}, {"v_mad_u16", k_mad16}, {"v_mul_lo_u16", k_mul16} };
    for (int waves = 16; waves <= 16; waves *= 2)
        for (auto &k : ks) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k.k, dim3(256), dim3(waves * 64), 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            // instructions per SIMD = waves/4 * iters * 64
            printf("waves/CU %2d %-18s %.3f ms  %.2f ns per instruction per SIMD\n", waves, k.n, ms, ms * 1e6 / ((waves / 4) * iters * 64.0));
        }
    return 0;
}
