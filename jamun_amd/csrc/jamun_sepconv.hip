// jamun_sepconv.hip — SeparableConv (src/jamun/e3tools/nn/_conv.py:122-135, _tensor_product.py:8-58): the conv of a ConvBlock
// with the tensor product factored into a depth-wise part ("uvu": one weight per input channel and (input x sh -> output) triple,
// 2 n0 + 3 n1 = 336 weights per edge for 120x0e + 32x1e) and a point-wise o3.Linear shared by all edges.
//
// Unlike the fully connected product (28 992 weights per edge, re-associated by destination in jamun_conv_dg.hip) the per-edge
// weights are cheap here, so the REFERENCE association is the right one:
//   k_sep_weights   w[slot][336] = h~[slot][0..64] . W2~  (+ bias row)                one fp32-MFMA GEMM, M = edge slots, K = 65
//   k_sep_apply     per destination: D = sum_e dtp(x_src, v_e; w_e)  (704 values), then m = Linear(D)  (the Linear commutes with
//                   the sum over edges); m goes to the partial-slab buffers of k_node_update (one slab), which divides by the
//                   in-degree (the mean), gates, and applies self-interaction + skip as for every other conv path.
// Constants folded into W2~ on the host: input noise scaling, path weights sqrt(2 l + 1), Clebsch-Gordan factors and the sqrt(3)
// of the l = 1 spherical harmonic, so that with the unit vector v of an edge
//   D0[u]   += wA x0[u]           D1[u][m] += wB x0[u] v[m]        D2[u'][m] += wC x1[u'][m]
//   D3[u']  += wD (x1[u'] . v)    D4[u'][m] += wE (x1[u'] x v)[m]
// Both kernels stream: HBM / L2-bound (the weights table is written once and read once per layer).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jamun_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define RFL(v) __builtin_amdgcn_readfirstlane(v)

// One wave = 32 consecutive edge slots x all weight columns.  A operand: h~ (k-major table: a K step is one coalesced 128-byte read
// per half-wave), held in 33 registers for the whole wave; B operand: W2~ packed [33 K pairs][column tile][64 lanes].
__global__ __launch_bounds__(256) void k_sep_weights(SepArgs a) {
  const int lane = threadIdx.x & 63, wave = RFL(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int64_t slot0 = tile * 32;
  if (slot0 >= a.n_slots) return;
  const int64_t slot = slot0 + r < a.n_slots ? slot0 + r : a.n_slots - 1;
  float av[33];
#pragma unroll
  for (int s = 0; s < 33; ++s) {
    const int k = 2 * s + hh;
    av[s] = k < JAMUN_HROWS ? a.h[(size_t)k * a.h_kstride + slot] : 0.f;
  }
  const int n_ct = a.NWp / 32;
  for (int ct = 0; ct < n_ct; ++ct) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const float* __restrict__ bp = a.w2p + ((size_t)ct * 64 + lane);
#pragma unroll
    for (int s = 0; s < 33; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bp[(size_t)s * n_ct * 64], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int64_t row = slot0 + (q & 3) + 8 * (q >> 2) + 4 * hh;
      if (row < a.n_slots) a.w[(size_t)row * a.NWp + ct * 32 + r] = acc[q];
    }
  }
}

// One workgroup = 16 destination atoms, 4 waves x 4 atoms.  Edge phase: one wave per destination, lanes = input channels
// (scalar channels lane and lane + 64, vector channel lane < n1): coalesced reads of the edge's weight row and of the source's
// feature row, sums in registers in edge order (fixed order: bit-reproducible).  Then the 704 sums of the wave's four atoms go to
// LDS and the point-wise Linear runs with lanes = output columns, every weight read shared by the four atoms.
#define SEP_APW 4  // atoms per wave
__global__ __launch_bounds__(256) void k_sep_apply(SepArgs a) {
  extern __shared__ float sep_lds[];  // [16 atoms][K0 + 3 K1]
  const int lane = threadIdx.x & 63, wave = RFL(threadIdx.x >> 6);
  const int n0 = a.n0, n1 = a.n1, K0 = n0 + n1, K1 = n0 + 2 * n1, DW = K0 + 3 * K1;
  const int atom0 = (blockIdx.x * 4 + wave) * SEP_APW;
  float* __restrict__ dl = sep_lds + (size_t)wave * SEP_APW * DW;
  for (int ai = 0; ai < SEP_APW; ++ai) {
    const int i = atom0 + ai;
    float d0[2] = {0.f, 0.f}, d1[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, d2[3] = {0.f, 0.f, 0.f}, d3 = 0.f, d4[3] = {0.f, 0.f, 0.f};
    if (i < a.n_atoms) {
      const int dg = a.deg[i];
      for (int t = 0; t < dg; ++t) {
        const size_t slot = (size_t)i * a.S + t;
        const int j = a.esrc[slot] & 0x7fffffff;
        const float4 ge = a.egeo[slot];  // unit vector (x, y, z), distance
        const float* __restrict__ wr = a.w + slot * a.NWp;
        const float* __restrict__ xr = a.x + (size_t)j * a.XS;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int u = lane + 64 * c;
          if (u < n0) {
            const float x0 = xr[u], wa = wr[u], wb = wr[n0 + u];
            d0[c] = fmaf(wa, x0, d0[c]);
            const float bx = wb * x0;
            d1[c][0] = fmaf(bx, ge.x, d1[c][0]);
            d1[c][1] = fmaf(bx, ge.y, d1[c][1]);
            d1[c][2] = fmaf(bx, ge.z, d1[c][2]);
          }
        }
        if (lane < n1) {
          const float* __restrict__ x1 = xr + n0 + 3 * lane;
          const float vx = x1[0], vy = x1[1], vz = x1[2];
          const float wc = wr[2 * n0 + lane], wd = wr[2 * n0 + n1 + lane], we = wr[2 * n0 + 2 * n1 + lane];
          d2[0] = fmaf(wc, vx, d2[0]);
          d2[1] = fmaf(wc, vy, d2[1]);
          d2[2] = fmaf(wc, vz, d2[2]);
          d3 = fmaf(wd, fmaf(vz, ge.z, fmaf(vy, ge.y, vx * ge.x)), d3);
          // (x1 x v)[m] = x1[m+1] v[m+2] - x1[m+2] v[m+1]
          d4[0] = fmaf(we, vy * ge.z - vz * ge.y, d4[0]);
          d4[1] = fmaf(we, vz * ge.x - vx * ge.z, d4[1]);
          d4[2] = fmaf(we, vx * ge.y - vy * ge.x, d4[2]);
        }
      }
    }
    // rows of the Linear's inputs: scalars [D0 (n0) | D3 (n1)], vectors per component m: [D1 (n0) | D2 (n1) | D4 (n1)]
    float* __restrict__ d = dl + (size_t)ai * DW;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int u = lane + 64 * c;
      if (u < n0) {
        d[u] = d0[c];
#pragma unroll
        for (int m = 0; m < 3; ++m) d[K0 + m * K1 + u] = d1[c][m];
      }
    }
    if (lane < n1) {
      d[n0 + lane] = d3;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        d[K0 + m * K1 + n0 + lane] = d2[m];
        d[K0 + m * K1 + n0 + n1 + lane] = d4[m];
      }
    }
  }
  __syncthreads();
  // point-wise Linear (normalisation folded into the weights): lanes = output columns, four atoms per weight read
  const int G0 = a.G0, G1 = a.G1;
  for (int c0 = 0; c0 < a.nt0 * 32; c0 += 64) {
    const int col = c0 + lane;
    float acc[SEP_APW] = {0.f, 0.f, 0.f, 0.f};
    if (col < G0) {
      for (int rr = 0; rr < K0; ++rr) {
        const float wv = a.wl0[(size_t)rr * G0 + col];
#pragma unroll
        for (int ai = 0; ai < SEP_APW; ++ai) acc[ai] = fmaf(wv, dl[(size_t)ai * DW + rr], acc[ai]);
      }
    }
    if (col < a.nt0 * 32) {
#pragma unroll
      for (int ai = 0; ai < SEP_APW; ++ai)
        if (atom0 + ai < a.n_atoms) a.partial0[(size_t)(atom0 + ai) * (a.nt0 * 32) + col] = acc[ai];
    }
  }
  for (int o0 = 0; o0 < 3 * a.nt1 * 32; o0 += 64) {
    const int o = o0 + lane;  // (m, column) of the vector outputs
    if (o >= 3 * a.nt1 * 32) break;
    const int m = o / (a.nt1 * 32), col = o - m * (a.nt1 * 32);
    float acc[SEP_APW] = {0.f, 0.f, 0.f, 0.f};
    if (col < G1) {
      for (int rr = 0; rr < K1; ++rr) {
        const float wv = a.wl1[(size_t)rr * G1 + col];
#pragma unroll
        for (int ai = 0; ai < SEP_APW; ++ai) acc[ai] = fmaf(wv, dl[(size_t)ai * DW + K0 + m * K1 + rr], acc[ai]);
      }
    }
#pragma unroll
    for (int ai = 0; ai < SEP_APW; ++ai)
      if (atom0 + ai < a.n_atoms) a.partial1[((size_t)(atom0 + ai) * 3 + m) * (a.nt1 * 32) + col] = acc[ai];
  }
}

int launch_sep_conv(const SepArgs& a, hipStream_t st) {
  if (a.n0 > 128 || a.n1 > 32 || a.NWp % 32 != 0 || a.NWp < 2 * a.n0 + 3 * a.n1) return -1;
  const size_t smem = (size_t)16 * (a.n0 + a.n1 + 3 * (a.n0 + 2 * a.n1)) * sizeof(float);
  if (smem > 64 * 1024) return -1;
  const int64_t tiles = (a.n_slots + 31) / 32;
  hipLaunchKernelGGL(k_sep_weights, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_sep_apply, dim3((a.n_atoms + 15) / 16), dim3(256), smem, st, a);
  return 0;
}
