// Device helpers shared by the matrix-formed conv kernels (jamun_conv_mf.hip, jamun_conv_ml.hip): f16x3 split primitives, LDS / constant
// address-space loads, power-of-two scales, the MFMA wrappers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jamun_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
// one f16x3 product: lo hi + hi lo + hi hi (fp32 accumulate; the lo lo term is below 2^-24 of the product)
#define M3(ACC, AH_, AL_, BH_, BL_)   \
  ACC = MFMA32H(AL_, BH_, ACC);       \
  ACC = MFMA32H(AH_, BL_, ACC);       \
  ACC = MFMA32H(AH_, BH_, ACC)
// ... or, in the opt-in reduced-precision mode (X1 in scope: "f16x1"), the hi hi product alone: operands rounded to 11 bits, fp32 accumulation — the
// loads, splits and conversions that only feed the dropped products are dead code for the compiler
#define MX(ACC, AH_, AL_, BH_, BL_)                              \
  do {                                                           \
    if constexpr (X1) { ACC = MFMA32H(AH_, BH_, ACC); }          \
    else { M3(ACC, AH_, AL_, BH_, BL_); }                        \
  } while (0)
#define RFL(v) __builtin_amdgcn_readfirstlane(v)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define MF_SCHED() __builtin_amdgcn_sched_barrier(0)

namespace {

__device__ __forceinline__ float4 lds_f4(int addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 v = *(const __attribute__((address_space(3))) f32x4*)(unsigned)addr;
  return make_float4(v.x, v.y, v.z, v.w);
#else
  (void)addr;
  return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}
__device__ __forceinline__ void lds_st4(int addr, const float4& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  *(__attribute__((address_space(3))) f32x4*)(unsigned)addr = f32x4{v.x, v.y, v.z, v.w};
#else
  (void)addr; (void)v;
#endif
}
__device__ __forceinline__ int lds_addr(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
#else
  (void)p;
  return 0;
#endif
}
// uniform loads through the constant address space: scalar loads into scalar registers
__device__ __forceinline__ int4 ld_const(const int4* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 v = *(const __attribute__((address_space(4))) i32x4*)(uintptr_t)p;
  return make_int4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ int2 ld_const(const int2* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  const i32x2 v = *(const __attribute__((address_space(4))) i32x2*)(uintptr_t)p;
  return make_int2(v.x, v.y);
#else
  return *p;
#endif
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
__device__ __forceinline__ int exp_above(float v) { return (int)((__float_as_uint(v) >> 23) & 0xffu) - 126; }  // v < 2^exp_above(v)
__device__ __forceinline__ int clamp40(int s) { return max(-40, min(40, s)); }
__device__ __forceinline__ int clamp100(int s) { return max(-100, min(100, s)); }

}  // namespace
