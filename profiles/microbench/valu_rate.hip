// valu_rate.hip — VALU issue rate per SIMD vs number of co-resident waves, with LONG loop bodies (64 instructions per iteration:
// the 8-instruction bodies of mfma_overlap*.hip pay a taken-branch bubble per iteration, which a single wave cannot hide), alone
// and beside 4 matrix waves running v_mfma_f32_32x32x16_f16 back to back.  256 workgroups, one per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define MEMTIME(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
template <int VK>
__device__ __forceinline__ float other_work(int n, int tid) {
  float s = 0;
  if constexpr (VK == 0) {
    f32x2 x[8];
    for (int q = 0; q < 8; ++q) x[q] = f32x2{(float)tid, (float)q};
    const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 64; ++q) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[q & 7]) : "v"(m), "v"(c));
    }
    for (int q = 0; q < 8; ++q) s += x[q].x + x[q].y;
  } else {
    float x[8];
    for (int q = 0; q < 8; ++q) x[q] = (float)(tid + q);
    const float m = 1.0001f, c = 1e-3f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 64; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q & 7]) : "v"(m), "v"(c));
    }
    for (int q = 0; q < 8; ++q) s += x[q];
  }
  return s;
}
template <int VK>
__global__ __launch_bounds__(1024) void ka(float* out, unsigned long long* cyc, int iters, int nm, int vk) {
  const int tid = threadIdx.x, w = tid >> 6;
  unsigned long long t0, t1;
  MEMTIME(t0);
  float s = 0;
  if (w < nm) {
    h8 av, bv;
    for (int q = 0; q < 8; ++q) { av[q] = (_Float16)((tid & 1) ? 1.f : -1.f); bv[q] = (_Float16)(1.f + q); }
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a) for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[u & 1], 0, 0, 0);
    }
    for (int a = 0; a < 2; ++a) for (int q = 0; q < 16; ++q) s += acc[a][q];
  } else {
    s = other_work<VK>(iters * vk / 8, tid);
  }
  MEMTIME(t1);
  out[blockIdx.x * 1024 + tid] = s;
  if ((tid & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}
template <int VK>
void run(float* out, unsigned long long* cyc, int nm, int nv, int vk, const char* what) {
  const int iters = 10000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((ka<VK>), dim3(256), dim3(64 * (nm + nv)), 0, 0, out, cyc, iters, nm, vk);
    (void)hipDeviceSynchronize();
  }
  unsigned long long mm = 0, vv = 0;
  for (int w = 0; w < nm; ++w) mm = cyc[w] > mm ? cyc[w] : mm;
  for (int w = nm; w < nm + nv; ++w) vv = cyc[w] > vv ? cyc[w] : vv;
  const double n_instr = 64.0 * (iters * vk / 8) * (nv / 4.0);  // per SIMD
  printf("%-52s | matrix waves %8llu cyc (%.1f per 8 MFMAs) | other waves %8llu cyc = %.2f cyc / instr / SIMD", what, mm, nm ? (double)mm / iters : 0.0, vv, nv ? vv / n_instr : 0.0);
  if (nm && nv) printf("  [while the MFMAs ran: %.2f cyc / instr / SIMD]", (double)mm / (n_instr - (vv > mm ? (double)(vv - mm) / (VK == 0 ? 4.4 : 4.1) : 0.0)));
  printf("\n");
}
int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMallocManaged(&cyc, 8 * 64);
  run<1>(out, cyc, 0, 4, 4, "4 waves v_fma_f32 (1 per SIMD)");
  run<1>(out, cyc, 0, 8, 4, "8 waves v_fma_f32");
  run<1>(out, cyc, 0, 12, 4, "12 waves v_fma_f32");
  run<1>(out, cyc, 0, 16, 4, "16 waves v_fma_f32");
  run<0>(out, cyc, 0, 4, 4, "4 waves v_pk_fma_f32");
  run<0>(out, cyc, 0, 8, 4, "8 waves v_pk_fma_f32");
  run<0>(out, cyc, 0, 12, 4, "12 waves v_pk_fma_f32");
  run<0>(out, cyc, 0, 16, 4, "16 waves v_pk_fma_f32");
  run<1>(out, cyc, 4, 0, 0, "4 matrix waves f16 32x32x16 alone");
  run<1>(out, cyc, 4, 4, 2, "4 matrix + 4 waves v_fma_f32 (32 instr per 8 MFMAs per SIMD)");
  run<1>(out, cyc, 4, 8, 2, "4 matrix + 8 waves v_fma_f32 (64 per 8 MFMAs)");
  run<1>(out, cyc, 4, 12, 2, "4 matrix + 12 waves v_fma_f32 (96 per 8 MFMAs)");
  run<1>(out, cyc, 4, 8, 1, "4 matrix + 8 waves v_fma_f32 (32 per 8 MFMAs)");
  run<0>(out, cyc, 4, 8, 1, "4 matrix + 8 waves v_pk_fma_f32 (32 per 8 MFMAs)");
  run<0>(out, cyc, 4, 8, 2, "4 matrix + 8 waves v_pk_fma_f32 (64 per 8 MFMAs)");
  run<0>(out, cyc, 4, 12, 2, "4 matrix + 12 waves v_pk_fma_f32 (96 per 8 MFMAs)");
  return 0;
}
