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
      // edge slot t of this destination lives in lane t (source index, unit vector): one coalesced read each, then broadcasts — and
      // the operands of edge t + 1 are requested before the FMAs of edge t (the loop is a chain of dependent global reads otherwise)
      const size_t slot_l = (size_t)i * a.S + (lane < dg ? lane : 0);
      const int j_l = a.esrc[slot_l] & 0x7fffffff;
      const float4 ge_l = a.egeo[slot_l];
      struct Ops { float x0[2], wa[2], wb[2], x1[3], wc, wd, we, gx, gy, gz; };
      auto fetch = [&](int t) {
        Ops o;
        const int tt = t < dg ? t : 0;
        const int j = __shfl(j_l, tt, 64);
        o.gx = __shfl(ge_l.x, tt, 64); o.gy = __shfl(ge_l.y, tt, 64); o.gz = __shfl(ge_l.z, tt, 64);
        const float* __restrict__ wr = a.w + ((size_t)i * a.S + tt) * a.NWp;
        const float* __restrict__ xr = a.x + (size_t)j * a.XS;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int u = lane + 64 * c;
          const bool ok = u < n0;
          o.x0[c] = ok ? xr[u] : 0.f; o.wa[c] = ok ? wr[u] : 0.f; o.wb[c] = ok ? wr[n0 + u] : 0.f;
        }
        const bool ok1 = lane < n1;
        const float* __restrict__ x1 = xr + n0 + 3 * (ok1 ? lane : 0);
        o.x1[0] = ok1 ? x1[0] : 0.f; o.x1[1] = ok1 ? x1[1] : 0.f; o.x1[2] = ok1 ? x1[2] : 0.f;
        o.wc = ok1 ? wr[2 * n0 + lane] : 0.f; o.wd = ok1 ? wr[2 * n0 + n1 + lane] : 0.f; o.we = ok1 ? wr[2 * n0 + 2 * n1 + lane] : 0.f;
        return o;
      };
      Ops cur{};
      if (dg > 0) cur = fetch(0);  // (the slots of an atom without in-edges hold no valid source index)
      for (int t = 0; t < dg; ++t) {
        const Ops nxt = fetch(t + 1);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          d0[c] = fmaf(cur.wa[c], cur.x0[c], d0[c]);
          const float bx = cur.wb[c] * cur.x0[c];
          d1[c][0] = fmaf(bx, cur.gx, d1[c][0]);
          d1[c][1] = fmaf(bx, cur.gy, d1[c][1]);
          d1[c][2] = fmaf(bx, cur.gz, d1[c][2]);
        }
        {
          const float vx = cur.x1[0], vy = cur.x1[1], vz = cur.x1[2];
          d2[0] = fmaf(cur.wc, vx, d2[0]);
          d2[1] = fmaf(cur.wc, vy, d2[1]);
          d2[2] = fmaf(cur.wc, vz, d2[2]);
          d3 = fmaf(cur.wd, fmaf(vz, cur.gz, fmaf(vy, cur.gy, vx * cur.gx)), d3);
          // (x1 x v)[m] = x1[m+1] v[m+2] - x1[m+2] v[m+1]
          d4[0] = fmaf(cur.we, vy * cur.gz - vz * cur.gy, d4[0]);
          d4[1] = fmaf(cur.we, vz * cur.gx - vx * cur.gz, d4[1]);
          d4[2] = fmaf(cur.we, vx * cur.gy - vy * cur.gx, d4[2]);
        }
        cur = nxt;
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

// What the SeparableConv kernels can run: checked once in jamun_sampler_create (a model / topology outside these limits is rejected
// there with this message, not at the first forward) and again at launch.
const char* sep_conv_unsupported(int n0, int n1, int NWp, int S) {
  if (n0 > 128 || n1 > 32) return "SeparableConv: input irreps wider than 128x0e + 32x1e";
  if (NWp % 32 != 0 || NWp < 2 * n0 + 3 * n1) return "SeparableConv: internal weight padding";
  if (S > 64) return "SeparableConv: more than 64 edge slots per destination (32 radial neighbours + bonded in-edges; repeated bond listings count)";  // edge slot t lives in lane t
  if ((size_t)16 * (n0 + n1 + 3 * (n0 + 2 * n1)) * sizeof(float) > 64 * 1024) return "SeparableConv: per-destination sums exceed 64 KiB of LDS";
  return nullptr;
}

int launch_sep_conv(const SepArgs& a, hipStream_t st) {
  if (sep_conv_unsupported(a.n0, a.n1, a.NWp, a.S)) return -1;
  const size_t smem = (size_t)16 * (a.n0 + a.n1 + 3 * (a.n0 + 2 * a.n1)) * sizeof(float);
  const int64_t tiles = (a.n_slots + 31) / 32;
  hipLaunchKernelGGL(k_sep_weights, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_sep_apply, dim3((a.n_atoms + 15) / 16), dim3(256), smem, st, a);
  return 0;
}
