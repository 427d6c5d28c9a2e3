// microbenchmark: VALU issue rate of the integer / packed / SDWA / permute instructions the DP kernels are made of, at
// 1, 2, 4 and 8 waves per SIMD (is the wave64 rate 4 or 2 cycles per instruction for THESE opcodes?).
//
// A workgroup is 256 threads = 4 waves (one per SIMD of the CU); its dynamic LDS request is sized so that exactly W
// workgroups fit on a CU (160 KiB / W), the grid holds many more workgroups than fit, so every SIMD runs W waves for the
// whole launch.  Each wave executes REPS x 32 independent instructions of one opcode (8 accumulator chains, unrolled by 4)
// and stamps s_memtime around the loop: cycles per instruction per SIMD = wave cycles / (instructions x W ... see main).
//
//   hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue && ./valu_issue > profiles/rNN_valu_issue.md
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>

#define REPS 2048
// one "row" = 8 independent instructions, acc[k] op= (x, y); written per opcode as inline asm so the compiler cannot fold it
#define ROW(INSN)                                                                                         \
    asm volatile(INSN(0) INSN(1) INSN(2) INSN(3) INSN(4) INSN(5) INSN(6) INSN(7)                          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)         \
                 : "v"(x), "v"(y) : "vcc");

#define I_ADD(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_PKADD(k) "v_pk_add_u16 %" #k ", %" #k ", %8\n"
#define I_PKMAX(k) "v_pk_max_i16 %" #k ", %" #k ", %8\n"
#define I_PKSUBC(k) "v_pk_sub_u16 %" #k ", %" #k ", %8 clamp\n"
#define I_PKMUL(k) "v_pk_mul_lo_u16 %" #k ", %" #k ", %8\n"
#define I_PKMAXSEL(k) "v_pk_max_i16 %" #k ", %" #k ", %8 op_sel:[0,1] op_sel_hi:[1,0]\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_SDWA(k) "v_add_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_MAX3(k) "v_max3_i32 %" #k ", %" #k ", %8, %9\n"
#define I_MAX(k) "v_max_i32 %" #k ", %" #k ", %8\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define I_ANDOR(k) "v_and_or_b32 %" #k ", %" #k ", %8, %9\n"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 7\n"
#define I_ALIGN(k) "v_alignbyte_b32 %" #k ", %" #k ", %8, 1\n"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %8\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define I_MADU24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define I_DPP(k) "v_add_u32_dpp %" #k ", %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MOVDPP(k) "v_mov_b32_dpp %" #k ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_PKFMA16(k) "v_pk_fma_f16 %" #k ", %" #k ", %8, %9\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_CMP(k) "v_cmp_lt_u32 vcc, %" #k ", %8\n"
#define I_SADU8(k) "v_sad_u8 %" #k ", %" #k ", %8, %9\n"
#define I_MIN3U(k) "v_min3_u32 %" #k ", %" #k ", %8, %9\n"

#define I_SUB(k) "v_sub_u32 %" #k ", %" #k ", %8\n"
#define I_OR(k) "v_or_b32 %" #k ", %" #k ", %8\n"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %8\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_MINU(k) "v_min_u32 %" #k ", %" #k ", %8\n"
#define I_MAXU(k) "v_max_u32 %" #k ", %" #k ", %8\n"
#define I_LSHR(k) "v_lshrrev_b32 %" #k ", 1, %" #k "\n"
#define I_ADDF(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define I_SUBF(k) "v_sub_f32 %" #k ", %" #k ", %8\n"
#define I_MULF(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define I_MAXF(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define I_MINF(k) "v_min_f32 %" #k ", %" #k ", %8\n"
#define I_MAX3F(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MED3F(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_ADDF16(k) "v_add_f16 %" #k ", %" #k ", %8\n"
#define I_MAXF16(k) "v_max_f16 %" #k ", %" #k ", %8\n"
#define I_PKMAXF16(k) "v_pk_max_f16 %" #k ", %" #k ", %8\n"
#define I_PKADDF16(k) "v_pk_add_f16 %" #k ", %" #k ", %8\n"
#define I_ADDU16(k) "v_add_u16 %" #k ", %" #k ", %8\n"
#define I_MAXI16(k) "v_max_i16 %" #k ", %" #k ", %8\n"
#define I_MUL24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n"
#define I_CVTUB0(k) "v_cvt_f32_ubyte0 %" #k ", %" #k "\n"
#define I_CVTUB2(k) "v_cvt_f32_ubyte2 %" #k ", %" #k "\n"
#define I_CVTI(k) "v_cvt_f32_i32 %" #k ", %" #k "\n"
#define I_CVTU8PK(k) "v_cvt_pk_u8_f32 %" #k ", %8, 1, %" #k "\n"
#define I_BFI(k) "v_bfi_b32 %" #k ", %8, %" #k ", %9\n"
#define I_XNOR(k) "v_xnor_b32 %" #k ", %" #k ", %8\n"
#define I_DOT4(k) "v_dot4_i32_i8 %" #k ", %8, %9, %" #k "\n"
#define I_ADDCO(k) "v_add_co_u32 %" #k ", vcc, %" #k ", %8\n"
#define I_CMPCND(k) "v_cmp_eq_u32 vcc, %" #k ", %8\nv_cndmask_b32 %" #k ", %9, %8, vcc\n"
#define I_SUBREVSDWA(k) "v_sub_u32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_ANDE64(k) "v_and_b32_e64 %" #k ", %" #k ", %8\n"
#define I_ADDE64(k) "v_add_u32_e64 %" #k ", %" #k ", %8\n"
#define I_ADDFE64NEG(k) "v_add_f32_e64 %" #k ", %" #k ", -%8\n"
#define I_MAXFE64ABS(k) "v_max_f32_e64 %" #k ", %" #k ", |%8|\n"
#define I_ANDSGPR(k) "v_and_b32 %" #k ", s4, %" #k "\n"
#define I_ADDLIT(k) "v_add_u32 %" #k ", 0x12345, %" #k "\n"
#define I_ADDINL(k) "v_add_u32 %" #k ", 7, %" #k "\n"
#define I_LSHL8(k) "v_lshlrev_b32 %" #k ", 8, %" #k "\n"
#define I_LSHLV(k) "v_lshlrev_b32 %" #k ", %9, %" #k "\n"
#define I_LSHL16(k) "v_lshlrev_b16 %" #k ", 1, %" #k "\n"
#define I_LSHR16(k) "v_lshrrev_b16 %" #k ", 1, %" #k "\n"
#define I_ASHR(k) "v_ashrrev_i32 %" #k ", 1, %" #k "\n"
#define I_LSHLOR(k) "v_lshl_or_b32 %" #k ", %" #k ", 8, %8\n"
#define I_MINU16(k) "v_min_u16 %" #k ", %" #k ", %8\n"
#define I_MAXU16(k) "v_max_u16 %" #k ", %" #k ", %8\n"
#define I_MINI16(k) "v_min_i16 %" #k ", %" #k ", %8\n"
#define I_SUBU16(k) "v_sub_u16 %" #k ", %" #k ", %8\n"
#define I_SUBU16C(k) "v_sub_u16_e64 %" #k ", %" #k ", %8 clamp\n"
#define I_ADDU16C(k) "v_add_u16_e64 %" #k ", %" #k ", %8 clamp\n"
#define I_SUBU32C(k) "v_sub_u32_e64 %" #k ", %" #k ", %8 clamp\n"
#define I_MULLO16(k) "v_mul_lo_u16 %" #k ", %" #k ", %8\n"
#define I_MADU16(k) "v_mad_u16 %" #k ", %" #k ", %8, %9\n"
#define I_MINF16(k) "v_min_f16 %" #k ", %" #k ", %8\n"
#define I_SUBF16(k) "v_sub_f16 %" #k ", %" #k ", %8\n"
#define I_MULF16(k) "v_mul_f16 %" #k ", %" #k ", %8\n"
#define I_FMAF16(k) "v_fma_f16 %" #k ", %" #k ", %8, %9\n"
#define I_MACF32(k) "v_fmac_f32 %" #k ", %8, %9\n"
#define I_MAXI16S(k) "v_max_i16 %" #k ", s4, %" #k "\n"
#define I_ADDSGPR(k) "v_add_u32 %" #k ", s4, %" #k "\n"
#define I_MAXI16LIT(k) "v_max_i16 %" #k ", 0, %" #k "\n"
#define I_CNDAFTERCMP(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_READLANE(k) "v_readfirstlane_b32 s5, %" #k "\n"
#define I_MBCNT(k) "v_mbcnt_lo_u32_b32 %" #k ", %8, %" #k "\n"

enum { OP_ADD, OP_PKADD, OP_PKMAX, OP_PKSUBC, OP_PKMUL, OP_PKMAXSEL, OP_PERM, OP_SDWA, OP_MAX3, OP_MAX, OP_AND, OP_ANDOR, OP_LSHL, OP_BFE,
       OP_ALIGN, OP_ADD3, OP_LSHLADD, OP_MULLO, OP_MADU24, OP_DPP, OP_MOVDPP, OP_FMA, OP_PKFMA16, OP_CNDMASK, OP_CMP, OP_SADU8, OP_MIN3U, OP_SUB, OP_OR, OP_XOR, OP_MOV, OP_MINU, OP_MAXU, OP_LSHR, OP_ADDF, OP_SUBF, OP_MULF, OP_MAXF, OP_MINF, OP_MAX3F, OP_MED3F, OP_ADDF16, OP_MAXF16, OP_PKMAXF16, OP_PKADDF16, OP_ADDU16, OP_MAXI16, OP_MUL24, OP_CVTUB0, OP_CVTUB2, OP_CVTI, OP_CVTU8PK, OP_BFI, OP_XNOR, OP_DOT4, OP_ADDCO, OP_CMPCND, OP_SUBREVSDWA, OP_ANDE64, OP_ADDE64, OP_ADDFE64NEG, OP_MAXFE64ABS, OP_ANDSGPR, OP_ADDLIT, OP_ADDINL, OP_LSHL8, OP_LSHLV, OP_LSHL16, OP_LSHR16, OP_ASHR, OP_LSHLOR, OP_MINU16, OP_MAXU16, OP_MINI16, OP_SUBU16, OP_SUBU16C, OP_ADDU16C, OP_SUBU32C, OP_MULLO16, OP_MADU16, OP_MINF16, OP_SUBF16, OP_MULF16, OP_FMAF16, OP_MACF32, OP_MAXI16S, OP_ADDSGPR, OP_MAXI16LIT, OP_CNDAFTERCMP, OP_READLANE, OP_MBCNT, OP_COUNT };
static const char *kNames[OP_COUNT] = {"v_add_u32", "v_pk_add_u16", "v_pk_max_i16", "v_pk_sub_u16 clamp", "v_pk_mul_lo_u16", "v_pk_max_i16 op_sel",
    "v_perm_b32", "v_add_u32_sdwa (BYTE_1)", "v_max3_i32", "v_max_i32", "v_and_b32", "v_and_or_b32", "v_lshlrev_b32", "v_bfe_u32",
    "v_alignbyte_b32", "v_add3_u32", "v_lshl_add_u32", "v_mul_lo_u32", "v_mad_u32_u24", "v_add_u32_dpp row_shr:1", "v_mov_b32_dpp row_shr:1",
    "v_fma_f32", "v_pk_fma_f16", "v_cndmask_b32 (vcc)", "v_cmp_lt_u32 (vcc)", "v_sad_u8", "v_min3_u32", "v_sub_u32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_min_u32", "v_max_u32", "v_lshrrev_b32", "v_add_f32", "v_sub_f32", "v_mul_f32", "v_max_f32", "v_min_f32", "v_max3_f32", "v_med3_f32", "v_add_f16", "v_max_f16", "v_pk_max_f16", "v_pk_add_f16", "v_add_u16", "v_max_i16", "v_mul_u32_u24", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte2", "v_cvt_f32_i32", "v_cvt_pk_u8_f32", "v_bfi_b32", "v_xnor_b32", "v_dot4_i32_i8", "v_add_co_u32 (vcc)", "v_cmp_eq_u32 + v_cndmask_b32 (pair = 2 instr)", "v_sub_u32_sdwa (WORD_1)", "v_and_b32_e64 (VOP3 encoding)", "v_add_u32_e64 (VOP3 encoding)", "v_add_f32_e64 with neg modifier", "v_max_f32_e64 with abs modifier", "v_and_b32 with SGPR operand", "v_add_u32 with literal", "v_add_u32 with inline constant", "v_lshlrev_b32 by 8", "v_lshlrev_b32 by VGPR", "v_lshlrev_b16 by 1", "v_lshrrev_b16 by 1", "v_ashrrev_i32 by 1", "v_lshl_or_b32", "v_min_u16", "v_max_u16", "v_min_i16", "v_sub_u16", "v_sub_u16 clamp", "v_add_u16 clamp", "v_sub_u32 clamp", "v_mul_lo_u16", "v_mad_u16", "v_min_f16", "v_sub_f16", "v_mul_f16", "v_fma_f16", "v_fmac_f32", "v_max_i16 with SGPR operand", "v_add_u32 with SGPR operand", "v_max_i16 with inline constant", "v_cndmask_b32 x8 after one v_cmp (8 instr)", "v_readfirstlane_b32 s5 (SALU dest)", "v_mbcnt_lo_u32_b32"};

template <int OP>
__global__ __launch_bounds__(256) void issue(uint32_t *out, unsigned long long *cyc, uint32_t seed) {
    extern __shared__ uint32_t lds[];
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t x = seed * 0x9E3779B9u + threadIdx.x, y = 0x03020100u ^ (seed & 0x01010101u);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if constexpr (OP == OP_ADD) { ROW(I_ADD) }
            else if constexpr (OP == OP_PKADD) { ROW(I_PKADD) }
            else if constexpr (OP == OP_PKMAX) { ROW(I_PKMAX) }
            else if constexpr (OP == OP_PKSUBC) { ROW(I_PKSUBC) }
            else if constexpr (OP == OP_PKMUL) { ROW(I_PKMUL) }
            else if constexpr (OP == OP_PKMAXSEL) { ROW(I_PKMAXSEL) }
            else if constexpr (OP == OP_PERM) { ROW(I_PERM) }
            else if constexpr (OP == OP_SDWA) { ROW(I_SDWA) }
            else if constexpr (OP == OP_MAX3) { ROW(I_MAX3) }
            else if constexpr (OP == OP_MAX) { ROW(I_MAX) }
            else if constexpr (OP == OP_AND) { ROW(I_AND) }
            else if constexpr (OP == OP_ANDOR) { ROW(I_ANDOR) }
            else if constexpr (OP == OP_LSHL) { ROW(I_LSHL) }
            else if constexpr (OP == OP_BFE) { ROW(I_BFE) }
            else if constexpr (OP == OP_ALIGN) { ROW(I_ALIGN) }
            else if constexpr (OP == OP_ADD3) { ROW(I_ADD3) }
            else if constexpr (OP == OP_LSHLADD) { ROW(I_LSHLADD) }
            else if constexpr (OP == OP_MULLO) { ROW(I_MULLO) }
            else if constexpr (OP == OP_MADU24) { ROW(I_MADU24) }
            else if constexpr (OP == OP_DPP) { ROW(I_DPP) }
            else if constexpr (OP == OP_MOVDPP) { ROW(I_MOVDPP) }
            else if constexpr (OP == OP_FMA) { ROW(I_FMA) }
            else if constexpr (OP == OP_PKFMA16) { ROW(I_PKFMA16) }
            else if constexpr (OP == OP_CNDMASK) { ROW(I_CNDMASK) }
            else if constexpr (OP == OP_CMP) { ROW(I_CMP) }
            else if constexpr (OP == OP_SADU8) { ROW(I_SADU8) }
            else if constexpr (OP == OP_MIN3U) { ROW(I_MIN3U) }
            else if constexpr (OP == OP_SUB) { ROW(I_SUB) }
            else if constexpr (OP == OP_OR) { ROW(I_OR) }
            else if constexpr (OP == OP_XOR) { ROW(I_XOR) }
            else if constexpr (OP == OP_MOV) { ROW(I_MOV) }
            else if constexpr (OP == OP_MINU) { ROW(I_MINU) }
            else if constexpr (OP == OP_MAXU) { ROW(I_MAXU) }
            else if constexpr (OP == OP_LSHR) { ROW(I_LSHR) }
            else if constexpr (OP == OP_ADDF) { ROW(I_ADDF) }
            else if constexpr (OP == OP_SUBF) { ROW(I_SUBF) }
            else if constexpr (OP == OP_MULF) { ROW(I_MULF) }
            else if constexpr (OP == OP_MAXF) { ROW(I_MAXF) }
            else if constexpr (OP == OP_MINF) { ROW(I_MINF) }
            else if constexpr (OP == OP_MAX3F) { ROW(I_MAX3F) }
            else if constexpr (OP == OP_MED3F) { ROW(I_MED3F) }
            else if constexpr (OP == OP_ADDF16) { ROW(I_ADDF16) }
            else if constexpr (OP == OP_MAXF16) { ROW(I_MAXF16) }
            else if constexpr (OP == OP_PKMAXF16) { ROW(I_PKMAXF16) }
            else if constexpr (OP == OP_PKADDF16) { ROW(I_PKADDF16) }
            else if constexpr (OP == OP_ADDU16) { ROW(I_ADDU16) }
            else if constexpr (OP == OP_MAXI16) { ROW(I_MAXI16) }
            else if constexpr (OP == OP_MUL24) { ROW(I_MUL24) }
            else if constexpr (OP == OP_CVTUB0) { ROW(I_CVTUB0) }
            else if constexpr (OP == OP_CVTUB2) { ROW(I_CVTUB2) }
            else if constexpr (OP == OP_CVTI) { ROW(I_CVTI) }
            else if constexpr (OP == OP_CVTU8PK) { ROW(I_CVTU8PK) }
            else if constexpr (OP == OP_BFI) { ROW(I_BFI) }
            else if constexpr (OP == OP_XNOR) { ROW(I_XNOR) }
            else if constexpr (OP == OP_DOT4) { ROW(I_DOT4) }
            else if constexpr (OP == OP_ADDCO) { ROW(I_ADDCO) }
            else if constexpr (OP == OP_CMPCND) { ROW(I_CMPCND) }
            else if constexpr (OP == OP_SUBREVSDWA) { ROW(I_SUBREVSDWA) }
            else if constexpr (OP == OP_ANDE64) { ROW(I_ANDE64) }
            else if constexpr (OP == OP_ADDE64) { ROW(I_ADDE64) }
            else if constexpr (OP == OP_ADDFE64NEG) { ROW(I_ADDFE64NEG) }
            else if constexpr (OP == OP_MAXFE64ABS) { ROW(I_MAXFE64ABS) }
            else if constexpr (OP == OP_ANDSGPR) { ROW(I_ANDSGPR) }
            else if constexpr (OP == OP_ADDLIT) { ROW(I_ADDLIT) }
            else if constexpr (OP == OP_ADDINL) { ROW(I_ADDINL) }
            else if constexpr (OP == OP_LSHL8) { ROW(I_LSHL8) }
            else if constexpr (OP == OP_LSHLV) { ROW(I_LSHLV) }
            else if constexpr (OP == OP_LSHL16) { ROW(I_LSHL16) }
            else if constexpr (OP == OP_LSHR16) { ROW(I_LSHR16) }
            else if constexpr (OP == OP_ASHR) { ROW(I_ASHR) }
            else if constexpr (OP == OP_LSHLOR) { ROW(I_LSHLOR) }
            else if constexpr (OP == OP_MINU16) { ROW(I_MINU16) }
            else if constexpr (OP == OP_MAXU16) { ROW(I_MAXU16) }
            else if constexpr (OP == OP_MINI16) { ROW(I_MINI16) }
            else if constexpr (OP == OP_SUBU16) { ROW(I_SUBU16) }
            else if constexpr (OP == OP_SUBU16C) { ROW(I_SUBU16C) }
            else if constexpr (OP == OP_ADDU16C) { ROW(I_ADDU16C) }
            else if constexpr (OP == OP_SUBU32C) { ROW(I_SUBU32C) }
            else if constexpr (OP == OP_MULLO16) { ROW(I_MULLO16) }
            else if constexpr (OP == OP_MADU16) { ROW(I_MADU16) }
            else if constexpr (OP == OP_MINF16) { ROW(I_MINF16) }
            else if constexpr (OP == OP_SUBF16) { ROW(I_SUBF16) }
            else if constexpr (OP == OP_MULF16) { ROW(I_MULF16) }
            else if constexpr (OP == OP_FMAF16) { ROW(I_FMAF16) }
            else if constexpr (OP == OP_MACF32) { ROW(I_MACF32) }
            else if constexpr (OP == OP_MAXI16S) { ROW(I_MAXI16S) }
            else if constexpr (OP == OP_ADDSGPR) { ROW(I_ADDSGPR) }
            else if constexpr (OP == OP_MAXI16LIT) { ROW(I_MAXI16LIT) }
            else if constexpr (OP == OP_CNDAFTERCMP) { ROW(I_CNDAFTERCMP) }
            else if constexpr (OP == OP_READLANE) { ROW(I_READLANE) }
            else if constexpr (OP == OP_MBCNT) { ROW(I_MBCNT) }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const uint32_t acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (acc == 0x12345678u) lds[threadIdx.x] = acc;          // keeps the LDS allocation and the chains alive
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(uint32_t *, unsigned long long *, uint32_t);
template <int OP> static void reg(kern_t *tab) {
    tab[OP] = issue<OP>;
    (void)hipFuncSetAttribute((const void *)issue<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if constexpr (OP + 1 < OP_COUNT) reg<OP + 1>(tab);
}

int main() {
    kern_t tab[OP_COUNT];
    reg<0>(tab);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int waves[4] = {1, 2, 4, 8};
    const size_t lds_for[4] = {96 * 1024, 64 * 1024, 36 * 1024, 19 * 1024};       // exactly W workgroups of 4 waves fit in 160 KiB
    const int rounds = 6;                                                          // workgroups per slot over the launch
    const int maxblocks = cus * 8 * rounds;
    uint32_t *out; unsigned long long *cyc;
    hipMalloc(&out, 4 * 256 * (size_t)maxblocks); hipMalloc(&cyc, 8 * 4 * (size_t)maxblocks);
    std::vector<unsigned long long> h(4 * (size_t)maxblocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# VALU issue rate on %s (%d CUs), wave64, %d independent instructions per wave per launch\n\n", prop.gcnArchName, cus, REPS * 32);
    printf("Columns: median wave cycles (s_memtime) per instruction = what ONE wave sees; x W waves sharing the SIMD gives the SIMD's\n"
           "cycles per wave64 instruction (`simd`), i.e. 4.0 = one instruction every 4 cycles (16 lanes / cycle), 2.0 = every 2 cycles.\n"
           "`T/s` = lane-instructions per second of the whole chip from the launch's wall time (HIP events).\n\n");
    printf("| instruction |");
    for (int w : waves) printf(" W=%d wave | simd | T/s |", w);
    printf("\n|---|");
    for (int i = 0; i < 4; i++) printf("---|---|---|");
    printf("\n");
    for (int op = 0; op < OP_COUNT; op++) {
        printf("| `%s` |", kNames[op]);
        for (int wi = 0; wi < 4; wi++) {
            const int W = waves[wi], blocks = cus * W * rounds;
            hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), lds_for[wi], 0, out, cyc, 1u);       // warm
            hipEventRecord(e0);
            hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), lds_for[wi], 0, out, cyc, 2u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, 8 * 4 * (size_t)blocks, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + 4 * (size_t)blocks);
            const double per_wave = (double)h[2 * (size_t)blocks] / (REPS * 32.0);
            const double lane_instr = (double)blocks * 256.0 * REPS * 32.0;
            printf(" %.2f | %.2f | %.1f |", per_wave, per_wave / W, lane_instr / (ms * 1e-3) / 1e12);
        }
        printf("\n");
        fflush(stdout);
    }
    printf("\nPeak if every SIMD issued one wave64 VALU instruction every 2 cycles at 2.4 GHz: %d CUs x 4 SIMDs x 64 lanes x 1.2 G = %.1f T lane-instr/s;"
           " every 4 cycles: %.1f T.\n", cus, cus * 4 * 64 * 1.2e9 / 1e12, cus * 4 * 64 * 0.6e9 / 1e12);
    return 0;
}
