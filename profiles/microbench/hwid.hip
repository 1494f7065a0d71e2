#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(768) void k(unsigned* out) {
  extern __shared__ float lds[];
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 12 + (threadIdx.x >> 6)] = id;
  lds[threadIdx.x] = id;
}
int main() {
  unsigned* out; hipMallocManaged(&out, 256 * 12 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(k, dim3(256), dim3(768), 150 * 1024, 0, out);
  hipDeviceSynchronize();
  for (int b : {0, 1, 2, 100, 255}) {
    printf("WG %3d: simd of waves 0..11:", b);
    for (int w = 0; w < 12; ++w) printf(" %u", (out[b * 12 + w] >> 4) & 3);
    printf("   cu %u se %u\n", (out[b * 12] >> 8) & 15, (out[b * 12] >> 13) & 7);
  }
  int hist[256] = {0};
  for (int b = 0; b < 256; ++b) { int m = 0; for (int w = 0; w < 4; ++w) m |= 1 << ((out[b * 12 + w] >> 4) & 3); hist[m]++; }
  for (int m = 0; m < 16; ++m) if (hist[m]) printf("matrix-wave SIMD mask %x: %d WGs\n", m, hist[m]);
  int bal = 0;
  for (int b = 0; b < 256; ++b) { int c[4] = {0, 0, 0, 0}; for (int w = 0; w < 12; ++w) c[(out[b * 12 + w] >> 4) & 3]++; if (c[0] == 3 && c[1] == 3 && c[2] == 3 && c[3] == 3) bal++; }
  printf("WGs with 3 waves on every SIMD: %d of 256\n", bal);
  return 0;
}
