#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// split 8 floats (two float4) into three bf16x8 planes: x = h + m + l (each RNE)
__device__ __forceinline__ void split8(const float4 a, const float4 b, u32x4& H, u32x4& M, u32x4& L) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const f32x2 v = {x[2 * p], x[2 * p + 1]};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    const f32x2 hf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
    const f32x2 r1 = v - hf;
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const unsigned mb = __builtin_bit_cast(unsigned, m);
    const f32x2 mf = {__builtin_bit_cast(float, mb << 16), __builtin_bit_cast(float, mb & 0xffff0000u)};
    const f32x2 r2 = r1 - mf;
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    H[p] = hb; M[p] = mb; L[p] = __builtin_bit_cast(unsigned, l);
  }
}
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
// C[32][32] = A[32][K] * B[K][32], one wave; assumed fragment layout: lane l: row/col = l % 32, k = 16 s + 8 (l / 32) + i
__global__ void k(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x, r = lane & 31, kb = lane >> 5;
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int s = 0; s < K / 16; ++s) {
    const float* ap = A + r * K + 16 * s + 8 * kb;
    float bq[8];
    for (int i = 0; i < 8; ++i) bq[i] = B[(16 * s + 8 * kb + i) * 32 + r];
    u32x4 ah, am, al, bh, bm, bl;
    split8(*(const float4*)ap, *(const float4*)(ap + 4), ah, am, al);
    split8(make_float4(bq[0], bq[1], bq[2], bq[3]), make_float4(bq[4], bq[5], bq[6], bq[7]), bh, bm, bl);
    acc = MF(al, bh, acc); acc = MF(ah, bl, acc); acc = MF(am, bm, acc);
    acc = MF(am, bh, acc); acc = MF(ah, bm, acc); acc = MF(ah, bh, acc);
  }
  for (int q = 0; q < 16; ++q) C[((q & 3) + 8 * (q >> 2) + 4 * kb) * 32 + r] = acc[q];
}
int main() {
  const int K = 128;
  std::vector<float> A(32 * K), B(K * 32), C(32 * 32);
  srand(1);
  for (auto& v : A) v = (float)((rand() / (double)RAND_MAX - 0.5) * exp(6.0 * (rand() / (double)RAND_MAX - 0.5)));
  for (auto& v : B) v = (float)((rand() / (double)RAND_MAX - 0.5) * 0.2);
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  double emax = 0, smax = 0, e32max = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0, sc = 0; float f = 0.f;
      for (int kk = 0; kk < K; ++kk) { ref += (double)A[i * K + kk] * B[kk * 32 + j]; sc += fabs((double)A[i * K + kk] * B[kk * 32 + j]); f = fmaf(A[i * K + kk], B[kk * 32 + j], f); }
      emax = fmax(emax, fabs(C[i * 32 + j] - ref) / sc); e32max = fmax(e32max, fabs(f - ref) / sc); smax = fmax(smax, sc);
    }
  printf("bf16x6 MFMA: max |err| / sum|a b| = %.3e   (sequential fp32 fma: %.3e)\n", emax, e32max);
  return 0;
}
