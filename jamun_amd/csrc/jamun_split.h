// jamun_split.h — THE f16x3 split primitives (one implementation for every kernel of the library).
//
// An fp32 value a is split exactly into two f16 terms, hi = rne16(a) and lo = rne16(a - hi): two values per v_cvt_pk_f16_f32, the residual
// with v_fma_mix_f32 (fma with per-operand f16 / f32 selection; a - hi is exact because hi holds the leading 11 bits of a).
//
// They are COMPILER-VISIBLE instructions, not inline asm, and that is a correctness rule, not a style (round 6, profiles/EXPERIMENTS.md
// "Root cause of the instruction-form miscomputations"):
//   on gfx942 / gfx950 a VGPR written by a vector-ALU instruction must not be read by a v_mfma_* (SrcA / SrcB / SrcC) for TWO wait states.
//   hipcc keeps that distance for the instructions it selects (`x = f(..); mfma(x, ..)` compiles to `v_..; s_waitcnt / s_nop; s_nop 0;
//   v_mfma`), but the body of an asm statement is opaque to its hazard recogniser: behind an asm that defines a VGPR it keeps ONE wait
//   state, its generic rule.  `asm("v_cvt_pk_f16_f32 %0, %1, %2")` whose result is the B operand of the next MFMA therefore reads, now and
//   then, the register's previous contents — k_conv_ml<8> with the asm forms in its T term: features 2e-4 off and different from call to
//   call; rounds 2-5 shipped the same pattern in k_conv_mf / k_conv_mfi / k_conv_mfx, right by the luck of their schedules
//   (profiles/tools/mfma_hazard_scan.py lists the sites; tests/test_abi.py runs it over the shipped build).
// With -1.0 held in a scalar register the optimiser cannot see through, fma(float(h), -1, a) is not folded into a subtraction and the back
// end selects exactly v_cvt_pk_f16_f32 / v_fma_mix_f32 — as instructions it schedules and whose hazards it tracks.
//
// -DJAMUN_SPLIT_ASM_REPRO restores the inline-asm v_cvt_pk_f16_f32 (reproducer builds only: profiles/microbench/README.md).
#pragma once
#include <hip/hip_runtime.h>

namespace {

typedef _Float16 jsp_h2 __attribute__((ext_vector_type(2)));
typedef float jsp_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float opaque_minus_one() {
  float m1 = -1.0f;
  asm("" : "+s"(m1));  // (not volatile: hoisted out of loops and merged like any other value)
  return m1;
}
__device__ __forceinline__ unsigned cvt_pk_f16_c(float a, float b) {
#ifdef JAMUN_SPLIT_ASM_REPRO
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return __builtin_bit_cast(unsigned, __builtin_convertvector(jsp_f2{a, b}, jsp_h2));
#endif
}
__device__ __forceinline__ float resid_lo_c(float a, unsigned pk, float m1) { return __builtin_fmaf((float)__builtin_bit_cast(jsp_h2, pk).x, m1, a); }  // a - float(pk[15:0])
__device__ __forceinline__ float resid_hi_c(float a, unsigned pk, float m1) { return __builtin_fmaf((float)__builtin_bit_cast(jsp_h2, pk).y, m1, a); }  // a - float(pk[31:16])
// the same without the explicit scalar (kernels that split in a few places only)
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) { return cvt_pk_f16_c(a, b); }
__device__ __forceinline__ float resid_lo(float a, unsigned pk) { return resid_lo_c(a, pk, opaque_minus_one()); }
__device__ __forceinline__ float resid_hi(float a, unsigned pk) { return resid_hi_c(a, pk, opaque_minus_one()); }

}  // namespace
