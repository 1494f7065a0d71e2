#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// waves [0, nm): bf16 32x32x16 MFMA chains; waves [nm, nm+nv): packed-FMA VALU
template <int NACC>
__global__ __launch_bounds__(768) void k(float* out, unsigned long long* cyc, int iters, int nm, int vk) {
  const int tid = threadIdx.x, w = tid >> 6;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float s = 0;
  if (w < nm) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
    bf16x8 av, bv;
    for (int q = 0; q < 8; ++q) { av[q] = (__bf16)((tid & 1) ? 1.f : -1.f); bv[q] = (__bf16)(1.f + q); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[a], 0, 0, 0);
    }
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) s += acc[a][q];
  } else {
    f32x2 x[8];
    for (int q = 0; q < 8; ++q) x[q] = f32x2{(float)tid, (float)q};
    const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    for (int it = 0; it < iters * vk; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = __builtin_elementwise_fma(x[q], m, c);
    }
    for (int q = 0; q < 8; ++q) s += x[q].x + x[q].y;
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * 768 + tid] = s;
  if ((tid & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 768 * 4); hipMallocManaged(&cyc, 8 * 64);
  const int iters = 20000;
  struct Cfg { int nacc, nm, nv, vk; const char* what; };
  Cfg cfgs[] = {{1, 4, 0, 0, "4 bf16 matrix waves, 1 acc"}, {2, 4, 0, 0, "4 bf16 matrix waves, 2 acc"}, {1, 8, 0, 0, "8 bf16 matrix waves, 1 acc"},
                {1, 0, 8, 2, "8 valu waves (vk 2)"}, {1, 4, 8, 2, "4 bf16 matrix + 8 valu (vk 2)"}, {2, 4, 8, 4, "4 bf16 matrix 2acc + 8 valu (vk 4)"},
                {1, 4, 4, 2, "4 bf16 matrix + 4 valu (vk 2)"}};
  for (auto& c : cfgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    const int thr = 64 * (c.nm + c.nv);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (c.nacc == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(thr), 0, 0, out, cyc, iters, c.nm, c.vk);
      else hipLaunchKernelGGL(k<2>, dim3(256), dim3(thr), 0, 0, out, cyc, iters, c.nm, c.vk);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double mf = 32768.0 * 4 * c.nacc * iters * c.nm * 256;
    const double vf = 4.0 * 64 * 8 * (double)iters * c.vk * c.nv * 256;
    printf("%-40s %.3f ms  mfma %.1f TF  valu %.1f TF | wave0 cycles %llu (%.1f per mfma); last wave cycles %llu\n", c.what, ms, mf / ms * 1e-9, vf / ms * 1e-9,
           cyc[0], c.nm ? (double)cyc[0] / (4.0 * c.nacc * iters) : 0.0, cyc[c.nm + c.nv - 1]);
  }
  return 0;
}
