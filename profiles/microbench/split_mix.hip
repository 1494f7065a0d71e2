// Checks that the 4-instruction hi/lo split (v_fma_mixlo/hi_f16 with the scale folded in) equals the 6-instruction one bit for bit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) { unsigned r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float resid_lo(float a, unsigned pk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(a)); return r; }
__device__ __forceinline__ float resid_hi(float a, unsigned pk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(a)); return r; }
__global__ void k(const float* f, float rs, unsigned* out_old, unsigned* out_new, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float F0 = f[2 * i], F1 = f[2 * i + 1];
  { float v0 = F0 * rs, v1 = F1 * rs; unsigned ph = cvt_pk_f16(v0, v1); unsigned pl = cvt_pk_f16(resid_lo(v0, ph), resid_hi(v1, ph)); out_old[2 * i] = ph; out_old[2 * i + 1] = pl; }
  {
    unsigned ph = 0u, pl = 0u;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(ph) : "v"(F0), "v"(rs));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(ph) : "v"(F1), "v"(rs));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(pl) : "v"(F0), "v"(rs), "v"(ph));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(pl) : "v"(F1), "v"(rs), "v"(ph));
    out_new[2 * i] = ph; out_new[2 * i + 1] = pl;
  }
}
int main() {
  const int n = 1 << 20;
  float* h = (float*)malloc(2 * n * sizeof(float));
  srand(1);
  for (int i = 0; i < 2 * n; ++i) { float m = (float)rand() / RAND_MAX * 2 - 1; int e = rand() % 60 - 30; h[i] = ldexpf(m, e + 20); if (i % 97 == 0) h[i] = 0.f; }
  float *d; unsigned *a, *b;
  hipMalloc(&d, 2 * n * 4); hipMalloc(&a, 2 * n * 4); hipMalloc(&b, 2 * n * 4);
  hipMemcpy(d, h, 2 * n * 4, hipMemcpyHostToDevice);
  unsigned* ha = (unsigned*)malloc(2 * n * 4); unsigned* hb = (unsigned*)malloc(2 * n * 4);
  long bad = 0;
  for (int e = -40; e <= -4; e += 6) {
    k<<<n / 256, 256>>>(d, ldexpf(1.f, e), a, b, n);
    hipMemcpy(ha, a, 2 * n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 2 * n * 4, hipMemcpyDeviceToHost);
    long bd = 0; for (int i = 0; i < 2 * n; ++i) if (ha[i] != hb[i]) { if (bd < 3) printf("rs 2^%d i %d old %08x new %08x f %g %g\n", e, i, ha[i], hb[i], h[i & ~1], h[i | 1]); ++bd; }
    printf("rs 2^%d: %ld mismatches of %d\n", e, bd, 2 * n); bad += bd;
  }
  printf(bad ? "FAIL\n" : "OK: identical\n");
  return bad != 0;
}
