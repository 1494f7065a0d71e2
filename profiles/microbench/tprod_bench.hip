#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define RFL(v) __builtin_amdgcn_readfirstlane(v)
// FLAGS: 1 no stores, 2 no B loads after the first, 4 no MFMA
template <int FLAGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_tprod(
    const float* __restrict__ x, int XS, int n_atoms, int n_k, int kg, const float4* __restrict__ wt, float* __restrict__ T) {
  const int lane = threadIdx.x & 63, wave = RFL(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int gid = blockIdx.x * 4 + wave;
  const int tiles_n = (n_atoms + 31) / 32;
  const int tile = (FLAGS & 16) ? RFL(gid % tiles_n) : RFL(gid / kg), g = (FLAGS & 16) ? RFL(gid / tiles_n) : RFL(gid - tile * kg);
  if (g >= kg) return;
  const int a0 = tile * 32;
  if (a0 >= n_atoms) return;
  const int k_lo = RFL((g * n_k) / kg), k_hi = RFL(((g + 1) * n_k) / kg);
  if (k_lo >= k_hi) return;
  const int row = min(a0 + r, n_atoms - 1);
  float4 av[15], b0[15], b1[15];
  auto load_b = [&](float4 (&bv)[15], int k) {
    const float4* __restrict__ wk = wt + (size_t)min(k, n_k - 1) * 15 * 64;
#pragma unroll
    for (int q = 0; q < 15; ++q) bv[q] = wk[q * 64 + lane];
  };
#pragma unroll
  for (int q = 0; q < 15; ++q) av[q] = *reinterpret_cast<const float4*>(x + (size_t)row * XS + 8 * q + 4 * hh);
  load_b(b0, k_lo);
  if (FLAGS & 2) load_b(b1, k_lo + 1);
  auto step = [&](const float4 (&bv)[15], int k) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    if (FLAGS & 8) {
#pragma unroll
      for (int q = 0; q < 15; ++q) {
        acc = MFMA32(bv[q].x, av[q].x, acc);
        acc = MFMA32(bv[q].y, av[q].y, acc);
        acc = MFMA32(bv[q].z, av[q].z, acc);
        acc = MFMA32(bv[q].w, av[q].w, acc);
      }
    } else if (!(FLAGS & 4)) {
#pragma unroll
      for (int q = 0; q < 15; ++q) {
        acc = MFMA32(av[q].x, bv[q].x, acc);
        acc = MFMA32(av[q].y, bv[q].y, acc);
        acc = MFMA32(av[q].z, bv[q].z, acc);
        acc = MFMA32(av[q].w, bv[q].w, acc);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 15; ++q) acc[q] = av[q].x * bv[q].x;
    }
    float* __restrict__ tk = T + ((size_t)k * n_atoms + a0) * 32;
    if (FLAGS & 8) {
      if (a0 + r < n_atoms) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *reinterpret_cast<float4*>(tk + r * 32 + 8 * g4 + 4 * hh) = make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
      }
    } else if (FLAGS & 1) {
      float s = 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) s += acc[q];
      if (s == 1.2345e30f) tk[r] = s;
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rr = (q & 3) + 8 * (q >> 2) + 4 * hh;
        if (a0 + rr < n_atoms) tk[rr * 32 + r] = acc[q];
      }
    }
  };
  for (int k = k_lo; k < k_hi; k += 2) {
    if (!(FLAGS & 2)) load_b(b1, k + 1);
    step(b0, k);
    if (k + 1 < k_hi) {
      if (!(FLAGS & 2)) load_b(b0, k + 2);
      step(b1, k + 1);
    }
  }
}
template <int F>
void run(const char* what, float* x, float4* wt, float* T, int n_atoms, int n_k, int waves_target) {
  const int tiles = (n_atoms + 31) / 32;
  const int kg = std::min(n_k, std::max(1, waves_target / tiles));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0, best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_tprod<F>, dim3((tiles * kg + 3) / 4), dim3(256), 0, 0, x, 216, n_atoms, n_k, kg, wt, T);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  printf("%-34s atoms %5d kg %2d waves %5d: %.1f us\n", what, n_atoms, kg, tiles * kg, best * 1e3);
}
int main() {
  const int n_k = 65;
  for (int n_atoms : {4352, 10624}) {
    float *x, *T; float4* wt;
    hipMalloc(&x, (size_t)n_atoms * 216 * 4); hipMalloc(&T, (size_t)n_k * n_atoms * 32 * 4 + 65536); hipMalloc(&wt, (size_t)n_k * 15 * 64 * 16);
    hipMemset(x, 0, (size_t)n_atoms * 216 * 4); hipMemset(wt, 0, (size_t)n_k * 15 * 64 * 16);
    for (int wt_target : {512, 1024, 2048}) {
      run<0>("full", x, wt, T, n_atoms, n_k, wt_target);
      run<1>("no stores", x, wt, T, n_atoms, n_k, wt_target);
      run<2>("no B reloads", x, wt, T, n_atoms, n_k, wt_target);
      run<3>("no stores, no B reloads", x, wt, T, n_atoms, n_k, wt_target);
      run<4>("no MFMA", x, wt, T, n_atoms, n_k, wt_target);
      run<8>("transposed tile, float4 stores", x, wt, T, n_atoms, n_k, wt_target);
      run<24>("float4 stores, tile-fastest map", x, wt, T, n_atoms, n_k, wt_target);
    }
    hipFree(x); hipFree(T); hipFree(wt);
  }
  return 0;
}
