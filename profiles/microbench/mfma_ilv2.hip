#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// one wave per SIMD: per iteration NM MFMAs (dependent chain) then NV independent v_pk_fma_f32; KIND 0: 32x32x2 f32, 1: 16x16x4 f32, 2: 4x4x1 f32 (4 passes)
template <int KIND, int NM, int NV>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  const int tid = threadIdx.x;
  f32x16 a32; f32x4 a16;
  for (int q = 0; q < 16; ++q) a32[q] = 0.f;
  for (int q = 0; q < 4; ++q) a16[q] = 0.f;
  const float sgn = (tid & 1) ? 1.f : -1.f;
  f32x2 x[16];
  for (int q = 0; q < 16; ++q) x[q] = f32x2{(float)tid, (float)q};
  const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if (KIND == 0) a32 = __builtin_amdgcn_mfma_f32_32x32x2f32(sgn, sgn, a32, 0, 0, 0);
      else if (KIND == 1) a16 = __builtin_amdgcn_mfma_f32_16x16x4f32(sgn, sgn, a16, 0, 0, 0);
      else a16 = __builtin_amdgcn_mfma_f32_4x4x1f32(sgn, sgn, a16, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NV; ++q) x[q % 16] = __builtin_elementwise_fma(x[q % 16], m, c);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0;
  for (int q = 0; q < 16; ++q) s += a32[q] + x[q].x + x[q].y;
  for (int q = 0; q < 4; ++q) s += a16[q];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND, int NM, int NV>
void run(float* out, unsigned long long* cyc) {
  const int iters = 20000;
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((k<KIND, NM, NV>), dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
  const char* names[] = {"32x32x2 (64 cyc)", "16x16x4 (32 cyc)", "4x4x1"};
  const int mc = KIND == 0 ? 64 : (KIND == 1 ? 32 : 8);
  const double per = (double)cyc[0] / iters;
  printf("%-18s %d MFMA + %2d pk_fma: %.1f cycles per iteration  (MFMA %d + VALU %d -> extra %.1f)\n", names[KIND], NM, NV, per, NM * mc, NV * 4, per - NM * mc - NV * 4);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMallocManaged(&cyc, 64);
  run<0, 4, 0>(out, cyc); run<0, 4, 4>(out, cyc); run<0, 4, 16>(out, cyc); run<0, 1, 4>(out, cyc); run<0, 1, 16>(out, cyc);
  run<1, 8, 0>(out, cyc); run<1, 8, 4>(out, cyc); run<1, 8, 16>(out, cyc); run<1, 1, 4>(out, cyc); run<1, 1, 16>(out, cyc); run<1, 2, 4>(out, cyc);
  return 0;
}
