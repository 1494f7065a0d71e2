// mfma_overlap.hip — does work of OTHER instruction classes overlap with fp32 MFMAs on a gfx950 SIMD?  (round 3; extends
// mfma_abs.hip / mfma_ilv2.hip, which only ever paired MFMAs with v_pk_fma_f32 and only with a dependent MFMA chain)
//
// Part A: co-resident waves.  256 workgroups (one per CU) of nm matrix waves + nv "other" waves.  Matrix-wave programs:
//   M0 one accumulator, dependent chain, back to back        M1 two accumulators alternating
//   M2 one accumulator, 3 x s_nop 15 behind every MFMA (the wave has no VALU-class instruction pending while its MFMA runs)
//   M3 four accumulators round robin                          M4 as M2 with 2 x s_nop 15        M5 as M0 at s_setprio 0 / others at 3
// Other-wave programs (8 independent chains per wave, all inline asm so the instruction is what the label says):
//   V0 v_pk_fma_f32   V1 v_fma_f32   V2 v_add_u32   V3 v_mov_b32   V4 ds_read_b128 (4 in flight)   V5 v_pk_fma_f32 at s_setprio 3
// Part B: ONE wave per SIMD, NM MFMAs over NACC independent accumulators, then NV instructions of kind V, repeated.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)
#define NOP16() asm volatile("s_nop 15")
#define MEMTIME(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")

template <int VK>
__device__ __forceinline__ float other_work(int n, int tid, float* lds) {
  float s = 0;
  if constexpr (VK == 0 || VK == 5) {
    if (VK == 5) __builtin_amdgcn_s_setprio(3);
    f32x2 x[8];
    for (int q = 0; q < 8; ++q) x[q] = f32x2{(float)tid, (float)q};
    const f32x2 m = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(m), "v"(c));
    }
    for (int q = 0; q < 8; ++q) s += x[q].x + x[q].y;
  } else if constexpr (VK == 1) {
    float x[8];
    for (int q = 0; q < 8; ++q) x[q] = (float)(tid + q);
    const float m = 1.0001f, c = 1e-3f;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(m), "v"(c));
    }
    for (int q = 0; q < 8; ++q) s += x[q];
  } else if constexpr (VK == 2) {
    int x[8];
    for (int q = 0; q < 8; ++q) x[q] = tid + q;
    const int c = 3;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[q]) : "v"(c));
    }
    for (int q = 0; q < 8; ++q) s += (float)x[q];
  } else if constexpr (VK == 3) {
    int x[8], y = tid;
    for (int q = 0; q < 8; ++q) x[q] = 0;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("v_mov_b32 %0, %1" : "=v"(x[q]) : "v"(y));
    }
    for (int q = 0; q < 8; ++q) s += (float)x[q];
  } else if constexpr (VK == 4) {
    const unsigned addr = (unsigned)(tid & 63) * 16u;
    f32x4 v[4];
    for (int it = 0; it < n; ++it) {
      // 8 reads per iteration (2 rounds of 4 in flight), like the 8 VALU instructions of the other kinds
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[q]) : "v"(addr), "n"(1024 * q));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    for (int q = 0; q < 4; ++q) s += v[q].x + v[q].w;
    (void)lds;
  }
  return s;
}

template <int MK>
__device__ __forceinline__ float matrix_work(int iters, int tid) {
  constexpr int NACC = MK == 1 ? 2 : (MK == 3 ? 4 : 1);
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
  const float sgn = (tid & 1) ? 1.f : -1.f;
  if (MK == 5) __builtin_amdgcn_s_setprio(0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[u % NACC] = MFMA(sgn, sgn, acc[u % NACC]);
      if constexpr (MK == 2 || MK == 4) {
        SB();
        NOP16(); NOP16();
        if (MK == 2) NOP16();
        SB();
      }
    }
  }
  float s = 0;
  for (int a = 0; a < NACC; ++a)
    for (int q = 0; q < 16; ++q) s += acc[a][q];
  return s;
}

template <int MK, int VK>
__global__ __launch_bounds__(768) void ka(float* out, unsigned long long* cyc, int iters, int nm, int vk) {
  __shared__ float lds[64 * 4 * 4 + 1024 * 4];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < 64 * 4 * 4 + 1024 * 4; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  unsigned long long t0, t1;
  MEMTIME(t0);
  float s;
  if (w < nm) s = matrix_work<MK>(iters, tid);
  else s = other_work<VK>(iters * vk, tid, lds);
  MEMTIME(t1);
  out[blockIdx.x * 768 + tid] = s;
  if ((tid & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

template <int MK, int VK>
void runA(float* out, unsigned long long* cyc, int nm, int nv, int vk, const char* what) {
  const int iters = 10000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((ka<MK, VK>), dim3(256), dim3(64 * (nm + nv)), 0, 0, out, cyc, iters, nm, vk);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long mm = 0, vv = 0;
  for (int w = 0; w < nm; ++w) mm = cyc[w] > mm ? cyc[w] : mm;
  for (int w = nm; w < nm + nv; ++w) vv = cyc[w] > vv ? cyc[w] : vv;
  // per-SIMD accounting: cycles per MFMA of a matrix wave, cycles per "other" instruction per SIMD (nv / 4 waves per SIMD)
  printf("A M%d V%d %-44s %7.3f ms | matrix waves %8llu cyc (%.1f / MFMA) | other waves %8llu cyc (%.2f cyc / instr / SIMD)\n", MK, VK, what, ms, mm,
         nm ? (double)mm / (4.0 * iters) : 0.0, vv, nv ? (double)vv / (8.0 * iters * vk * (nv / 4.0)) : 0.0);
}

// ---- part B: one wave per SIMD
template <int NACC, int NM, int NV, int VK, bool FINE>
__global__ __launch_bounds__(256) void kb(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[64 * 4 * 4 + 1024 * 4];
  const int tid = threadIdx.x;
  for (int i = tid; i < 64 * 4 * 4 + 1024 * 4; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
  const float sgn = (tid & 1) ? 1.f : -1.f;
  f32x2 xp[8];
  float xf[8];
  int xi[8];
  f32x4 xl[4];
  for (int q = 0; q < 8; ++q) { xp[q] = f32x2{(float)tid, (float)q}; xf[q] = (float)(tid + q); xi[q] = tid + q; }
  for (int q = 0; q < 4; ++q) xl[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x2 m2 = {1.0001f, 0.9999f}, c2 = {1e-3f, -1e-3f};
  const float m1 = 1.0001f, c1 = 1e-3f;
  const int ci = 3;
  const unsigned addr = (unsigned)(tid & 63) * 16u;
  auto other = [&](int q) {
    if constexpr (VK == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[q % 8]) : "v"(m2), "v"(c2));
    else if constexpr (VK == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xf[q % 8]) : "v"(m1), "v"(c1));
    else if constexpr (VK == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(xi[q % 8]) : "v"(ci));
    else if constexpr (VK == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(xl[q % 4]) : "v"(addr));
  };
  unsigned long long t0, t1;
  MEMTIME(t0);
  for (int it = 0; it < iters; ++it) {
    if constexpr (FINE) {
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        acc[i % NACC] = MFMA(sgn, sgn, acc[i % NACC]);
        SB();
#pragma unroll
        for (int q = 0; q < NV / NM; ++q) other(i * (NV / NM) + q);
        SB();
      }
    } else {
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i % NACC] = MFMA(sgn, sgn, acc[i % NACC]);
      SB();
#pragma unroll
      for (int q = 0; q < NV; ++q) other(q);
      SB();
    }
    if constexpr (VK == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  MEMTIME(t1);
  float s = 0;
  for (int a = 0; a < NACC; ++a)
    for (int q = 0; q < 16; ++q) s += acc[a][q];
  for (int q = 0; q < 8; ++q) s += xp[q].x + xp[q].y + xf[q] + (float)xi[q];
  for (int q = 0; q < 4; ++q) s += xl[q].x + xl[q].w;
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC, int NM, int NV, int VK, bool FINE>
void runB(float* out, unsigned long long* cyc) {
  const int iters = 10000;
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL((kb<NACC, NM, NV, VK, FINE>), dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
  const char* vn[] = {"v_pk_fma_f32", "v_fma_f32", "v_add_u32", "v_mov_b32", "ds_read_b128"};
  const double per = (double)cyc[0] / iters;
  printf("B %d acc  %d MFMA + %2d %-13s %-5s: %7.1f cycles / iteration  (MFMA alone %d; other alone ~%d)  extra over MFMA %.1f\n", NACC, NM, NV, vn[VK], FINE ? "fine" : "block", per,
         NM * 64, NV * 4, per - NM * 64);
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 768 * 4); hipMallocManaged(&cyc, 8 * 64);
  // ---- A: alone
  runA<0, 0>(out, cyc, 4, 0, 0, "4 matrix waves M0 alone");
  runA<1, 0>(out, cyc, 4, 0, 0, "4 matrix waves M1 (2 acc) alone");
  runA<2, 0>(out, cyc, 4, 0, 0, "4 matrix waves M2 (3 s_nop 15 each) alone");
  runA<4, 0>(out, cyc, 4, 0, 0, "4 matrix waves M4 (2 s_nop 15 each) alone");
  runA<3, 0>(out, cyc, 4, 0, 0, "4 matrix waves M3 (4 acc) alone");
  runA<0, 0>(out, cyc, 0, 8, 4, "8 waves v_pk_fma_f32 alone");
  runA<0, 1>(out, cyc, 0, 8, 4, "8 waves v_fma_f32 alone");
  runA<0, 2>(out, cyc, 0, 8, 4, "8 waves v_add_u32 alone");
  runA<0, 3>(out, cyc, 0, 8, 4, "8 waves v_mov_b32 alone");
  runA<0, 4>(out, cyc, 0, 8, 4, "8 waves ds_read_b128 alone");
  runA<0, 1>(out, cyc, 0, 4, 4, "4 waves v_fma_f32 alone");
  // ---- A: together, vk = 4 -> 4 matrix waves owe 64 cyc x 4 x iters per SIMD, 8 other waves owe 2 x 8 x 4 x 4 cyc x iters = the same
  runA<0, 0>(out, cyc, 4, 8, 4, "M0 + 8 waves v_pk_fma_f32");
  runA<0, 1>(out, cyc, 4, 8, 4, "M0 + 8 waves v_fma_f32");
  runA<0, 2>(out, cyc, 4, 8, 4, "M0 + 8 waves v_add_u32");
  runA<0, 3>(out, cyc, 4, 8, 4, "M0 + 8 waves v_mov_b32");
  runA<0, 4>(out, cyc, 4, 8, 4, "M0 + 8 waves ds_read_b128");
  runA<1, 0>(out, cyc, 4, 8, 4, "M1 (2 acc) + 8 waves v_pk_fma_f32");
  runA<1, 1>(out, cyc, 4, 8, 4, "M1 (2 acc) + 8 waves v_fma_f32");
  runA<3, 1>(out, cyc, 4, 8, 4, "M3 (4 acc) + 8 waves v_fma_f32");
  runA<2, 0>(out, cyc, 4, 8, 4, "M2 (nops) + 8 waves v_pk_fma_f32");
  runA<2, 1>(out, cyc, 4, 8, 4, "M2 (nops) + 8 waves v_fma_f32");
  runA<2, 2>(out, cyc, 4, 8, 4, "M2 (nops) + 8 waves v_add_u32");
  runA<2, 4>(out, cyc, 4, 8, 4, "M2 (nops) + 8 waves ds_read_b128");
  runA<4, 0>(out, cyc, 4, 8, 4, "M4 (2 nops) + 8 waves v_pk_fma_f32");
  runA<4, 1>(out, cyc, 4, 8, 4, "M4 (2 nops) + 8 waves v_fma_f32");
  runA<5, 5>(out, cyc, 4, 8, 4, "M5 (prio 0) + 8 waves v_pk_fma_f32 prio 3");
  runA<0, 5>(out, cyc, 4, 8, 4, "M0 + 8 waves v_pk_fma_f32 prio 3");
  runA<0, 1>(out, cyc, 4, 4, 4, "M0 + 4 waves v_fma_f32");
  runA<2, 1>(out, cyc, 4, 4, 4, "M2 (nops) + 4 waves v_fma_f32");
  // ---- B: one wave per SIMD
  runB<1, 4, 0, 0, false>(out, cyc);
  runB<4, 4, 0, 0, false>(out, cyc);
  runB<1, 4, 8, 0, false>(out, cyc);
  runB<4, 4, 8, 0, false>(out, cyc);
  runB<4, 4, 8, 0, true>(out, cyc);
  runB<4, 4, 8, 1, false>(out, cyc);
  runB<4, 4, 8, 1, true>(out, cyc);
  runB<4, 4, 8, 2, true>(out, cyc);
  runB<4, 4, 8, 4, true>(out, cyc);
  runB<4, 4, 32, 0, true>(out, cyc);
  runB<4, 4, 32, 1, true>(out, cyc);
  runB<4, 4, 32, 2, true>(out, cyc);
  runB<2, 4, 32, 1, true>(out, cyc);
  runB<1, 4, 32, 1, true>(out, cyc);
  runB<4, 4, 48, 1, true>(out, cyc);
  runB<4, 4, 16, 4, true>(out, cyc);
  return 0;
}
