// mfma_overlap_bf16.hip — part A of mfma_overlap.hip with bf16 MFMAs on the matrix waves: how much VALU / LDS work of co-resident
// waves runs WHILE v_mfma_f32_32x32x16_bf16 (32 cycles) / v_mfma_f32_16x16x32_bf16 (16 cycles) chains keep the matrix pipe busy?
// 256 workgroups (one per CU): nm matrix waves + nv other waves.  Other-wave programs (8 independent chains, inline asm):
//   V0 v_pk_fma_f32  V1 v_fma_f32  V2 v_add_u32  V3 v_cvt_pk_bf16_f32  V4 ds_read_b128  V6 v_and_b32 + v_sub_f32 pairs (the split)  V7 v_perm_b32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
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
    float x[8]; int y[8];
    for (int q = 0; q < 8; ++q) { x[q] = (float)(tid + q); y[q] = 0; }
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y[q]) : "v"(x[q]), "v"(x[(q + 1) & 7]));
    }
    for (int q = 0; q < 8; ++q) s += (float)y[q];
  } else if constexpr (VK == 4) {
    const unsigned addr = (unsigned)(tid & 63) * 16u;
    f32x4 v[4];
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[q]) : "v"(addr), "n"(1024 * q));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    for (int q = 0; q < 4; ++q) s += v[q].x + v[q].w;
  } else if constexpr (VK == 6) {
    float x[4], y[4];
    for (int q = 0; q < 4; ++q) { x[q] = (float)(tid + q) * 1.37f; y[q] = 0.f; }
    const int mask = 0xffff0000;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        asm volatile("v_and_b32 %0, %1, %2" : "=v"(y[q]) : "v"(mask), "v"(x[q]));
        asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[q]) : "v"(y[q]));
      }
    }
    for (int q = 0; q < 4; ++q) s += x[q] + y[q];
  } else if constexpr (VK == 7) {
    int x[8];
    for (int q = 0; q < 8; ++q) x[q] = tid * 77 + q;
    const int sel = 0x07060302;
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(x[(q + 1) & 7]), "v"(sel));
    }
    for (int q = 0; q < 8; ++q) s += (float)x[q];
  }
  return s;
}

template <int MT, int NACC>
__device__ __forceinline__ float matrix_work(int iters, int tid) {
  bf16x8 av, bv;
  for (int q = 0; q < 8; ++q) { av[q] = (__bf16)((tid & 1) ? 1.f : -1.f); bv[q] = (__bf16)(1.f + q); }
  float s = 0;
  if constexpr (MT == 1) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[u % NACC], 0, 0, 0);
    }
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 16; ++q) s += acc[a][q];
  } else {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 4; ++q) acc[a][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[u % NACC], 0, 0, 0);
    }
    for (int a = 0; a < NACC; ++a) for (int q = 0; q < 4; ++q) s += acc[a][q];
  }
  return s;
}

// per iteration a matrix wave owes 8 x 32 (or 16 x 16) = 256 cycles of matrix pipe
template <int MT, int NACC, int VK>
__global__ __launch_bounds__(1024) void ka(float* out, unsigned long long* cyc, int iters, int nm, int vk) {
  __shared__ float lds[64 * 4 * 4 + 1024 * 4];
  const int tid = threadIdx.x, w = tid >> 6;
  for (int i = tid; i < 64 * 4 * 4 + 1024 * 4; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  unsigned long long t0, t1;
  MEMTIME(t0);
  float s;
  if (w < nm) s = matrix_work<MT, NACC>(iters, tid);
  else s = other_work<VK>(iters * vk, tid);
  MEMTIME(t1);
  out[blockIdx.x * 1024 + tid] = s;
  if ((tid & 63) == 0 && blockIdx.x == 0) cyc[w] = t1 - t0;
}

template <int MT, int NACC, int VK>
void runA(float* out, unsigned long long* cyc, int nm, int nv, int vk, const char* what) {
  const int iters = 10000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((ka<MT, NACC, VK>), dim3(256), dim3(64 * (nm + nv)), 0, 0, out, cyc, iters, nm, vk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  unsigned long long mm = 0, vv = 0;
  for (int w = 0; w < nm; ++w) mm = cyc[w] > mm ? cyc[w] : mm;
  for (int w = nm; w < nm + nv; ++w) vv = cyc[w] > vv ? cyc[w] : vv;
  const int per_it = (VK == 6) ? 8 : 8;
  printf("MT%d acc%d V%d %-46s %7.3f ms | matrix waves %8llu cyc (%.1f / 256-cycle block) | other waves %8llu cyc (%.2f cyc / instr / SIMD)\n", MT, NACC, VK, what, ms, mm,
         nm ? (double)mm / iters : 0.0, vv, nv ? (double)vv / ((double)per_it * iters * vk * (nv / 4.0)) : 0.0);
}

int main() {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMallocManaged(&cyc, 8 * 64);
  runA<1, 1, 1>(out, cyc, 4, 0, 0, "4 waves 32x32x16 bf16, 1 acc, alone");
  runA<1, 2, 1>(out, cyc, 4, 0, 0, "4 waves 32x32x16 bf16, 2 acc, alone");
  runA<2, 2, 1>(out, cyc, 4, 0, 0, "4 waves 16x16x32 bf16, 2 acc, alone");
  runA<2, 4, 1>(out, cyc, 4, 0, 0, "4 waves 16x16x32 bf16, 4 acc, alone");
  runA<1, 2, 0>(out, cyc, 0, 8, 4, "8 waves v_pk_fma_f32 alone");
  runA<1, 2, 1>(out, cyc, 0, 8, 4, "8 waves v_fma_f32 alone");
  runA<1, 2, 1>(out, cyc, 0, 12, 4, "12 waves v_fma_f32 alone");
  runA<1, 2, 1>(out, cyc, 0, 16, 4, "16 waves v_fma_f32 alone");
  runA<1, 2, 2>(out, cyc, 0, 8, 4, "8 waves v_add_u32 alone");
  runA<1, 2, 2>(out, cyc, 0, 16, 4, "16 waves v_add_u32 alone");
  runA<1, 2, 3>(out, cyc, 0, 8, 4, "8 waves v_cvt_pk_bf16_f32 alone");
  runA<1, 2, 6>(out, cyc, 0, 8, 4, "8 waves v_and + v_sub alone");
  runA<1, 2, 7>(out, cyc, 0, 8, 4, "8 waves v_perm_b32 alone");
  runA<1, 2, 4>(out, cyc, 0, 8, 4, "8 waves ds_read_b128 alone");
  // together: vk = 4 -> other waves (2 per SIMD) owe 2 x 8 x 4 = 64 instructions per 256-cycle matrix block per SIMD
  runA<1, 2, 0>(out, cyc, 4, 8, 4, "32x32x16 + 8 waves v_pk_fma_f32");
  runA<1, 2, 1>(out, cyc, 4, 8, 4, "32x32x16 + 8 waves v_fma_f32");
  runA<1, 1, 1>(out, cyc, 4, 8, 4, "32x32x16 (1 acc) + 8 waves v_fma_f32");
  runA<1, 2, 1>(out, cyc, 4, 12, 4, "32x32x16 + 12 waves v_fma_f32");
  runA<1, 2, 2>(out, cyc, 4, 8, 4, "32x32x16 + 8 waves v_add_u32");
  runA<1, 2, 3>(out, cyc, 4, 8, 4, "32x32x16 + 8 waves v_cvt_pk_bf16_f32");
  runA<1, 2, 6>(out, cyc, 4, 8, 4, "32x32x16 + 8 waves v_and + v_sub");
  runA<1, 2, 7>(out, cyc, 4, 8, 4, "32x32x16 + 8 waves v_perm_b32");
  runA<1, 2, 4>(out, cyc, 4, 8, 4, "32x32x16 + 8 waves ds_read_b128");
  runA<1, 2, 1>(out, cyc, 4, 8, 2, "32x32x16 + 8 waves v_fma_f32 (half the VALU work)");
  runA<1, 2, 0>(out, cyc, 4, 8, 2, "32x32x16 + 8 waves v_pk_fma_f32 (half)");
  runA<2, 4, 0>(out, cyc, 4, 8, 4, "16x16x32 + 8 waves v_pk_fma_f32");
  runA<2, 4, 1>(out, cyc, 4, 8, 4, "16x16x32 + 8 waves v_fma_f32");
  runA<2, 4, 4>(out, cyc, 4, 8, 4, "16x16x32 + 8 waves ds_read_b128");
  return 0;
}
