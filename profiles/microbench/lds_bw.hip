#include <hip/hip_runtime.h>
#include <stdio.h>
// LDS read throughput per CU: pattern 0 unique b128 (lane*16), 1 per-half broadcast b128, 2 full broadcast b128,
// 3 two halves read the same 512 B (lane&31)*16, 4 unique b64, 5 unique b32, 6 broadcast b32, 7 row gather: half-uniform row + u*16
template <int PAT>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ float4 lds4[];
  float* lds = (float*)lds4;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 16384; i += blockDim.x) lds[i] = i;
  __syncthreads();
  int off;
  if (PAT == 0) off = lane * 16;
  else if (PAT == 1) off = (lane >> 5) * 16;
  else if (PAT == 2) off = 0;
  else if (PAT == 3) off = (lane & 31) * 16;
  else if (PAT == 4) off = lane * 8;
  else if (PAT == 5 ) off = lane * 4;
  else if (PAT == 6) off = (lane >> 5) * 4;
  else off = (lane >> 5) * 1008 * 3 + (lane & 31) * 16;
  off += w * 2048;
  const char* base = (const char*)lds;
  float4 acc = {0, 0, 0, 0};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  const unsigned ad = (unsigned)off;
  for (int it = 0; it < iters; ++it) {
    float4 v0, v1, v2, v3; float2 d0, d1, d2, d3; float f0, f1, f2, f3;
    if (PAT == 4) {
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:1024\n\tds_read_b64 %2, %4 offset:2048\n\tds_read_b64 %3, %4 offset:3072\n\t"
                   "ds_read_b64 %0, %4 offset:4096\n\tds_read_b64 %1, %4 offset:5120\n\tds_read_b64 %2, %4 offset:6144\n\tds_read_b64 %3, %4 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(ad) : "memory");
      acc.x += d0.x;
    } else if (PAT == 5 || PAT == 6) {
      asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:1024\n\tds_read_b32 %2, %4 offset:2048\n\tds_read_b32 %3, %4 offset:3072\n\t"
                   "ds_read_b32 %0, %4 offset:4096\n\tds_read_b32 %1, %4 offset:5120\n\tds_read_b32 %2, %4 offset:6144\n\tds_read_b32 %3, %4 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(ad) : "memory");
      acc.x += f0;
    } else {
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\t"
                   "ds_read_b128 %0, %4 offset:4096\n\tds_read_b128 %1, %4 offset:5120\n\tds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(ad) : "memory");
      acc.x += v0.x;
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * 1024 + tid] = acc.x + acc.y + acc.z + acc.w;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMallocManaged(&cyc, 8);
  const int iters = 40000;
  const char* names[] = {"unique b128", "per-half broadcast b128", "full broadcast b128", "two halves same 512B b128", "unique b64", "unique b32", "per-half bcast b32", "row gather b128"};
  for (int waves : {4, 8, 16}) {
    for (int p = 0; p < 8; ++p) {
#define L(P) if (p == P) hipLaunchKernelGGL(k<P>, dim3(256), dim3(64 * waves), 64 * 1024, 0, out, cyc, iters)
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      printf("waves %2d  %-28s wave0: %.2f   whole kernel: %.2f cycles per wave-instruction per CU (at 2.4 GHz)\n", waves, names[p], (double)*cyc / ((double)iters * 8 * waves),
             ms * 2.4e6 / ((double)iters * 8 * waves));
    }
  }
  return 0;
}
