// jamun_node.hip — node update of a ConvBlock with both o3.Linear layers as f16x3 MFMAs.
//
// Same arithmetic as k_node_update (jamun_kernels.hip; src/jamun/e3tools/nn/_conv.py:117, _gate.py:53-64, _interaction.py:26-30,
// model/noise_conditioning.py:69-73): sum the partial slabs of the conv (fixed order), mean over the in-edges, leaky-relu / sigmoid gate,
// out = [act | x_in] . [W_self ; W_skip] (one contraction per irrep block), noise-conditional skip mix.  What changes is the shape:
//   * phase 1 works on 16-byte pieces (16 threads per atom; 19 loads per thread instead of 62) and writes the Linear inputs ROW-major
//     ([atom][K] halves, hi and lo planes: the A fragments of v_mfma_f32_32x32x16_f16 are 16-byte reads of a row), scaled per atom
//     by a power of two so that its largest input sits below 2^14 (one scale for the scalar K range, one for the vector planes);
//   * phase 2 runs 15 (scalar tile, K = 240) or 4 (vector plane, K = 64) K-steps of three f16 MFMAs — 1.4 k matrix cycles for a scalar
//     tile against 7.7 k with v_mfma_f32_32x32x2_f32 — against weights split hi + lo on the host;
//   * the weights are balanced by exact powers of two per ROW (K index: the input is multiplied by the inverse when it is staged, before
//     the per-atom scale is chosen) and per COLUMN (undone in the epilogue): with one scale per tensor, weights 2^-14 below the largest
//     have denormal lo halves, and an input channel that is small because its weights are large would lose its low bits to the
//     atom's largest input (NodeArgs::kga0 ...; round 4: a checkpoint with a per-channel spread of 2^+-10 came out 0.9 off per channel).
// Measured on MI355X (cfg2, 4352 atoms, 3 slabs): k_node_update 24.4 us = 13.8 phase 1 + 10.6 phase 2; see DESIGN.md 3.4 for this kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdio.h>

#include "jamun_internal.h"
#include "jamun_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define NH_T 512
#define NH_S 3  // partial slabs fetched at once (more are summed in a loop)
#define NH_W 8  // K-steps of weight blocks in flight per wave (the first NH_W are requested before phase 1)
#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
#define RFL(v) __builtin_amdgcn_readfirstlane(v)

#ifdef NH_TRACE  // per-wave timeline of every workgroup (diagnostic builds): [workgroup * 8 + wave][stamp]
__device__ unsigned long long g_nhtrace[2400][8];
#define NSTAMP(slot) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0 && blockIdx.x < 300) g_nhtrace[blockIdx.x * 8 + wave][slot] = t_; } while (0)
#define NSTAMP_RT(slot) do { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0 && blockIdx.x < 300) g_nhtrace[blockIdx.x * 8 + wave][slot] = t_; } while (0)
#define NSTAMP_V(slot, v) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(v) :: "memory"); if (lane == 0 && blockIdx.x < 300) g_nhtrace[blockIdx.x * 8 + wave][slot] = t_; } while (0)
#else
#define NSTAMP(slot) do { } while (0)
#define NSTAMP_RT(slot) do { } while (0)
#define NSTAMP_V(slot, v) do { } while (0)
#endif

namespace {

__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
__device__ __forceinline__ int scale_of(float mx) {  // largest magnitude -> [2^13, 2^14)
  int s = 0;
  if (mx > 0.f) s = 14 - ((int)((__float_as_uint(mx) >> 23) & 0xffu) - 126);
  return max(-60, min(60, s));
}
__device__ __forceinline__ float max4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
// four consecutive K values of one row -> their hi halves (8 bytes) and lo halves, scaled by sc
__device__ __forceinline__ void put4(char* hi_row, int lo_off, int k, float4 v, float sc) {
  const float e0 = v.x * sc, e1 = v.y * sc, e2 = v.z * sc, e3 = v.w * sc;
  const unsigned p0 = cvt_pk_f16(e0, e1), p1 = cvt_pk_f16(e2, e3);
  const unsigned q0 = cvt_pk_f16(resid_lo(e0, p0), resid_hi(e1, p0)), q1 = cvt_pk_f16(resid_lo(e2, p1), resid_hi(e3, p1));
  *reinterpret_cast<uint2*>(hi_row + 2 * k) = make_uint2(p0, p1);
  *reinterpret_cast<uint2*>(hi_row + lo_off + 2 * k) = make_uint2(q0, q1);
}
__device__ __forceinline__ void put1(char* hi_row, int lo_off, int k, float v, float sc) {
  const float e = v * sc;
  const unsigned p = cvt_pk_f16(e, 0.f), q = cvt_pk_f16(resid_lo(e, p), 0.f);
  *reinterpret_cast<unsigned short*>(hi_row + 2 * k) = (unsigned short)(p & 0xffffu);
  *reinterpret_cast<unsigned short*>(hi_row + lo_off + 2 * k) = (unsigned short)(q & 0xffffu);
}

}  // namespace

// One workgroup = 32 atoms, 8 waves; 16 threads per atom in phase 1.  Requires nt0 <= 5 (partial0 rows of <= 160 floats), nt1 == 1,
// XSin a multiple of 4 and <= 224, in0 a multiple of 4, mul0 a multiple of 4, K0h / K1h multiples of 16.
// LR = the low-register variant (<= 128 registers: two workgroups per CU; chosen by the launcher when the grid exceeds the CUs, where a
// second round of a few workgroups would double the launch: 33 x 256 atoms = 264 workgroups took 36 us against 21 for 256): nothing of
// phase 2 is requested before phase 1 — the other workgroup of the CU covers those round trips — and the weight ring is 4 steps deep.
// FX = the contraction depths are the standard ones (K0h = 240, K1h = 64: 15 and 4 K-steps): phase 2 is unrolled with compile-time slots of
// the weight ring.  (With run-time depths the compiler rotates the ring through register COPIES behind `s_waitcnt vmcnt(0)` — every K-step
// waited for the request it had just issued, ~500 cycles each, 6-9 k cycles for the 45 MFMAs of a scalar tile: profiles/r6_cfg2_tprod_node_trace.txt.)
template <bool LR, bool FX>
__global__ __launch_bounds__(NH_T, LR ? 4 : 2) void k_node_update_h(NodeArgs a) {
  constexpr int NW = LR ? 4 : NH_W;
  extern __shared__ float4 nh_lds[];
  char* __restrict__ sm = reinterpret_cast<char*>(nh_lds);
  const int RB0 = RFL(a.K0h * 2 + 16), RB1 = RFL(a.K1h * 2 + 16);  // row bytes: K halves + 16 (rows 16 B apart mod 32: conflict-free b128 reads)
  const int L0 = 32 * RB0, L1 = 96 * RB1;                          // hi -> lo plane
  char* __restrict__ A0 = sm;                                      // [32 atoms][K0h] hi | lo
  char* __restrict__ A1 = A0 + 2 * L0;                             // [3 planes][32 atoms][K1h] hi | lo
  float* __restrict__ s_gate = reinterpret_cast<float*>(A1 + 2 * L1);  // [32][mul1 <= 32]
  float* __restrict__ isc0 = s_gate + 32 * 32;                     // [32] 2^-(sS_atom + sW0)
  float* __restrict__ isc1 = isc0 + 32;                            // [32] 2^-(sV_atom + sW1)
  const int n0 = blockIdx.x * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = RFL(tid >> 6);
  const int il = tid >> 4, c16 = tid & 15;
  const int i = n0 + il;
  NSTAMP_RT(0);
  NSTAMP(1);
  const bool ok = i < a.n_atoms;
  const int ns = ok ? (a.atom_nslab ? a.atom_nslab[i] : a.n_slices) : 0;
  const int ns_max = a.atom_nslab ? a.max_slabs : a.n_slices;
  const int dgi = ok ? a.deg[i] : 1;
  const float degf = (float)(dgi < 1 ? 1 : dgi);
  // mean over the in-edges as a product with 1 / deg, sigmoid with v_exp_f32 / v_rcp_f32 (each within 1 ulp of the IEEE forms the fp32
  // kernel k_node_update keeps; the results are split into 22-bit operands a few lines below): 32 ten-instruction divisions per thread less
  const float rdeg = 1.f / degf;
  const int w0 = a.nt0 * 32, G0 = a.mul0 + a.mul1;
  const size_t slab0 = (size_t)a.n_pad * w0, slab1 = (size_t)a.n_pad * 96;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // this wave's job of phase 2: scalar-output tile `job` (32 columns, K0h / 16 steps) or vector plane job - nts (K1h / 16 steps)
  const int nts = (a.mul0 + 31) >> 5, nst0 = a.K0h >> 4, nst1 = a.K1h >> 4;
  const int job = wave;
  const bool has_job = job < nts + 3, scalar_j = job < nts;
  const int nst_j = scalar_j ? nst0 : nst1;
  const float4* __restrict__ wp_j = (scalar_j ? a.wh0 + (size_t)job * nst0 * 128 : a.wh1) + lane;
  const int r_j = lane & 31, hh_j = lane >> 5;
  const int col_j = scalar_j ? job * 32 + r_j : r_j;                  // output channel within the irrep block
  const bool col_ok_j = has_job && (scalar_j ? col_j < a.mul0 : col_j < a.mul1);
  const int o_j = scalar_j ? col_j : a.mul0 + 3 * col_j + (job - nts);  // column of x_out
  float4 wh[NW], wl[NW];
  float xo[LR ? 1 : 16];

  // ---- loads: scalar rows (<= 40 pieces of 16 bytes per atom and slab), vector rows (24), input features (<= 56); the first NH_S
  // slabs of everything are in flight together, summation order stays s = 0, 1, 2, ...
  float4 ms[3], mv[2], xv[4];
  // (row / column factors of the balanced weights and the skip-mix weight of this lane's output column: requested here, with the slabs —
  // as loads at their points of use they were round trips of their own in front of phase 1a, phase 1b and the epilogue)
  // (the low-register variant keeps them at their points of use: it has no registers to hold them, and a second workgroup on its CU to wait behind)
  auto ld_kg0 = [&](int q) { return *reinterpret_cast<const float4*>(a.kga0 + min(4 * (c16 + 16 * q), a.mul0 - 4)); };
  auto ld_kg1 = [&](int q) { return *reinterpret_cast<const float4*>(a.kga1 + min(4 * ((c16 + 16 * q) & 7), a.mul1 - 4)); };
  auto ld_kgx = [&](int q) { return *reinterpret_cast<const float4*>(a.kgx + min(4 * (c16 + 16 * q), a.XSin - 4)); };
  auto ld_mw = [&]() { return (a.mix && col_ok_j) ? a.mix[scalar_j ? col_j : a.mul0 + col_j] : 0.f; };
  auto ld_cinv = [&]() { return col_ok_j ? (scalar_j ? a.cg0[col_j] : a.cg1[col_j]) : 0.f; };  // 2^-sW of this lane's output column
  float4 kg0[LR ? 1 : 3], kg1[LR ? 1 : 2], kgxv[LR ? 1 : 4];
  float mw_j = 0.f, cinv_j = 0.f;
  if constexpr (!LR) {
#pragma unroll
    for (int q = 0; q < 3; ++q) kg0[q] = ld_kg0(q);
#pragma unroll
    for (int q = 0; q < 2; ++q) kg1[q] = ld_kg1(q);
#pragma unroll
    for (int q = 0; q < 4; ++q) kgxv[q] = ld_kgx(q);
    mw_j = ld_mw();
    cinv_j = ld_cinv();
  }
  {
    float4 ls[NH_S][3], lv[NH_S][2];
    const float* __restrict__ p0 = a.partial0 + (size_t)(n0 + il) * w0;
    const float* __restrict__ p1 = a.partial1 + (size_t)(n0 + il) * 96;
#pragma unroll
    for (int s = 0; s < NH_S; ++s) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int j = c16 + 16 * q;
        ls[s][q] = *reinterpret_cast<const float4*>(p0 + (s < ns_max ? s : 0) * slab0 + min(4 * j, w0 - 4));  // (clamped, masked below:
        // a conditional load of this form compiles to a FLAT load through a pointer select with the zero constant in scratch)
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = c16 + 16 * q;
        lv[s][q] = *reinterpret_cast<const float4*>(p1 + (s < ns_max ? s : 0) * slab1 + 4 * min(j, 23));
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = c16 + 16 * q;
      xv[q] = *reinterpret_cast<const float4*>(a.x_in + (size_t)min(i, a.n_atoms - 1) * a.XSin + min(4 * j, a.XSin - 4));
    }
    // (the loads above do not wait for this atom's slab count: slab indices are clamped to the batch's maximum, results masked here)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const bool in = 4 * (c16 + 16 * q) < w0;
      ms[q] = (in && 0 < ns) ? ls[0][q] : z4;
#pragma unroll
      for (int s = 1; s < NH_S; ++s) ms[q] = f4add(ms[q], (in && s < ns) ? ls[s][q] : z4);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const bool in = c16 + 16 * q < 24;
      mv[q] = (in && 0 < ns) ? lv[0][q] : z4;
#pragma unroll
      for (int s = 1; s < NH_S; ++s) mv[q] = f4add(mv[q], (in && s < ns) ? lv[s][q] : z4);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (!(ok && 4 * (c16 + 16 * q) < a.XSin)) xv[q] = z4;
    for (int s = NH_S; s < ns_max; ++s) {
      if (s < ns) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int j = c16 + 16 * q;
          if (4 * j < w0) ms[q] = f4add(ms[q], *reinterpret_cast<const float4*>(p0 + s * slab0 + 4 * j));
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int j = c16 + 16 * q;
          if (j < 24) mv[q] = f4add(mv[q], *reinterpret_cast<const float4*>(p1 + s * slab1 + 4 * j));
        }
      }
    }
  }

#if defined(NH_EXP) && (NH_EXP & 2)
  if (a.mix) { if (ms[0].x + ms[1].x + ms[2].x + mv[0].x + mv[1].x + xv[0].x + xv[1].x + xv[2].x + xv[3].x == 1.2345f) a.x_out[0] = 0.f; return; }
#endif
  // phase 2's operands that do not depend on phase 1 — the first NH_W K-steps of this wave's weight blocks and the x_old values of its
  // output tile — are requested here, once the slabs have ARRIVED: their round trips run during phase 1 instead of after its barrier.
  // (Requested together with the slabs they delayed them: a CU's eight waves move 126 KB of slabs and 160 KB of weights / x_old through
  // one 64 B/clk vector-memory path, and waves 4..7's slabs queued behind waves 0..3's weights — slabs at ~10 k cycles, the barrier waiting
  // for the late half: profiles/r6_cfg2_tprod_node_trace.txt)
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (!LR) {
#pragma unroll
    for (int t = 0; t < NW; ++t) {
      const int sw = t < nst_j ? t : nst_j - 1;
      wh[t] = wp_j[(2 * sw) * 64];  // (a wave without a job reads the blocks of plane 0: valid addresses, results unused)
      wl[t] = wp_j[(2 * sw + 1) * 64];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int rl = (q & 3) + 8 * (q >> 2) + 4 * hh_j, ii = n0 + rl;
      xo[q] = (a.mix && col_ok_j && ii < a.n_atoms) ? a.x_in[(size_t)ii * a.XSin + o_j] : 0.f;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  NSTAMP_V(2, ms[0].x);
  // ---- phase 1a: mean over in-edges, activation (scalar K range) / gate; the scalar channels of x_in extend the scalar K range
  float mxS = 0.f;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int c = 4 * (c16 + 16 * q);
    float e[4] = {ms[q].x * rdeg, ms[q].y * rdeg, ms[q].z * rdeg, ms[q].w * rdeg};
    const float4 k4 = LR ? ld_kg0(q) : kg0[LR ? 0 : q];
    const float kg[4] = {k4.x, k4.y, k4.z, k4.w};  // (a piece is all activated scalars or all gate pre-activations: mul0 % 4 == 0)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int w = c + t;
      if (w < a.mul0) { e[t] = (a.cL * (e[t] > 0.f ? e[t] : 0.01f * e[t])) * kg[t]; mxS = fmaxf(mxS, fabsf(e[t])); }
      else if (w < G0) s_gate[il * 32 + (w - a.mul0)] = a.cS * __builtin_amdgcn_rcpf(1.f + __expf(-e[t]));
    }
    ms[q] = make_float4(e[0], e[1], e[2], e[3]);
  }
  float mxV = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 4 * (c16 + 16 * q);
    if (c < a.XSin) {  // x_in channels times the row factors of the skip weights (exact powers of two)
      const float4 kg = LR ? ld_kgx(q) : kgxv[LR ? 0 : q];
      xv[q] = make_float4(xv[q].x * kg.x, xv[q].y * kg.y, xv[q].z * kg.z, xv[q].w * kg.w);
    }
    if (c < a.in0) mxS = fmaxf(mxS, max4(xv[q]));          // (in0 is a multiple of 4: a piece is all scalar or all vector)
    else if (c < a.XSin) mxV = fmaxf(mxV, max4(xv[q]));
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) mxS = fmaxf(mxS, __shfl_xor(mxS, o, 16));
  const int sS = scale_of(mxS);
  {
    const float sc = pow2f(sS);
    char* __restrict__ row = A0 + il * RB0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int c = 4 * (c16 + 16 * q);
      if (c < a.mul0) put4(row, L0, c, ms[q], sc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * (c16 + 16 * q);
      if (c < a.in0) put4(row, L0, a.mul0 + c, xv[q], sc);
    }
    for (int k = a.mul0 + a.in0 + 4 * c16; k < a.K0h; k += 64) put4(row, L0, k, z4, 1.f);  // pad rows of the K range
    if (c16 == 0) isc0[il] = pow2f(-sS);
  }
  NSTAMP(3);
  __syncthreads();  // gates
  // (standard depths: the K-steps of a scalar tile beyond the ring are requested here — the slab registers are free — and arrive during phase 1b)
  float4 wh2[(FX && !LR) ? 7 : 1], wl2[(FX && !LR) ? 7 : 1];
  if constexpr (FX && !LR) {
    if (scalar_j) {
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        wh2[t] = wp_j[(2 * (NW + t)) * 64];
        wl2[t] = wp_j[(2 * (NW + t) + 1) * 64];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- phase 1b: gated vectors and the vector channels of x_in (K range of a plane: [gate * m1 (mul1) | x_in vectors (in1) | pad])
  float gv[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int j = c16 + 16 * q, c = 4 * (j & 7);  // piece j: plane j / 8, channels c .. c + 3
    const float e[4] = {mv[q].x * rdeg, mv[q].y * rdeg, mv[q].z * rdeg, mv[q].w * rdeg};
    const float4 k4 = LR ? ld_kg1(q) : kg1[LR ? 0 : q];
    const float kg[4] = {k4.x, k4.y, k4.z, k4.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      gv[q][t] = (j < 24 && c + t < a.mul1) ? (e[t] * s_gate[il * 32 + c + t]) * kg[t] : 0.f;
      mxV = fmaxf(mxV, fabsf(gv[q][t]));
    }
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) mxV = fmaxf(mxV, __shfl_xor(mxV, o, 16));
  const int sV = scale_of(mxV);
  {
    const float sc = pow2f(sV);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = c16 + 16 * q, m = j >> 3, c = 4 * (j & 7);
      if (j < 24 && c < a.mul1) put4(A1 + (m * 32 + il) * RB1, L1, c, make_float4(gv[q][0], gv[q][1], gv[q][2], gv[q][3]), sc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = 4 * (c16 + 16 * q);
      if (c >= a.in0 && c < a.XSin) {
        const float e[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int v = c + t - a.in0, u = v / 3, m = v - 3 * u;  // x_in vectors are stored [u][m]
          put1(A1 + (m * 32 + il) * RB1, L1, a.mul1 + u, e[t], sc);
        }
      }
    }
    for (int idx = c16; idx < 3 * ((a.K1h - a.mul1 - a.in1) >> 2); idx += 16) {  // pad rows of the planes' K range
      const int per = (a.K1h - a.mul1 - a.in1) >> 2, m = idx / per, k = a.mul1 + a.in1 + 4 * (idx - m * per);
      put4(A1 + (m * 32 + il) * RB1, L1, k, z4, 1.f);
    }
    if (c16 == 0) isc1[il] = pow2f(-sV);
  }
  __syncthreads();
  NSTAMP(4);
  // (the operands requested in front of phase 1 are consumed here once: first used inside the epilogue's predicated stores, the compiler's
  // wait for them is `s_waitcnt vmcnt(0)` in front of EVERY store block — sixteen serialised store round trips, ~5 k cycles)
  if constexpr (!LR) {
#pragma unroll
    for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(xo[q]));
    asm volatile("" : "+v"(mw_j), "+v"(cinv_j));
  }

  // ---- phase 2: one job per wave: scalar-output tile (32 columns, K0h / 16 steps) or vector plane (K1h / 16 steps); weight blocks of
  // 64 lanes x 8 halves, (hi, lo) per step, NH_W steps in flight (the first NH_W requested before phase 1)
#if defined(NH_EXP) && (NH_EXP & 1)
  if (a.mix) return;
#endif
  const int XSo = a.mul0 + 3 * a.mul1;
  if (has_job) {
    const bool scalar = scalar_j;
    const int nst = nst_j, r = r_j, hh = hh_j;
    const float4* __restrict__ wp = wp_j;
    const char* __restrict__ ap = (scalar ? A0 + r * RB0 : A1 + ((job - nts) * 32 + r) * RB1) + 16 * hh;
    const int lo = scalar ? L0 : L1;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    if constexpr (LR) {
#pragma unroll
      for (int t = 0; t < NW; ++t) {
        const int sw = t < nst ? t : nst - 1;
        wh[t] = wp[(2 * sw) * 64];
        wl[t] = wp[(2 * sw + 1) * 64];
      }
    }
    if constexpr (FX) {
      auto run = [&](auto nst_c) {
        constexpr int NST = decltype(nst_c)::value;
        float4 ah = *reinterpret_cast<const float4*>(ap), al = *reinterpret_cast<const float4*>(ap + lo);
#pragma unroll
        for (int s_ = 0; s_ < NST; ++s_) {
          float4 nah = ah, nal = al;
          if (s_ + 1 < NST) {
            nah = *reinterpret_cast<const float4*>(ap + 32 * (s_ + 1));
            nal = *reinterpret_cast<const float4*>(ap + lo + 32 * (s_ + 1));
          }
          float4 bh = wh[s_ % NW], bl = wl[s_ % NW];
          if constexpr (!LR) {
            if (s_ >= NW) { bh = wh2[s_ >= NW ? s_ - NW : 0]; bl = wl2[s_ >= NW ? s_ - NW : 0]; }
          }
          acc = MFMA32H(al, bh, acc);
          acc = MFMA32H(ah, bl, acc);
          acc = MFMA32H(ah, bh, acc);
          if constexpr (LR) {
            if (s_ + NW < NST) {  // (compile-time slot: no copies)
              wh[s_ % NW] = wp[(2 * (s_ + NW)) * 64];
              wl[s_ % NW] = wp[(2 * (s_ + NW) + 1) * 64];
            }
          }
          ah = nah; al = nal;
          __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks every reload to just in front of its use)
        }
      };
      if (scalar) run(std::integral_constant<int, 15>{});
      else run(std::integral_constant<int, 4>{});
    } else
    for (int s0 = 0; s0 < nst; s0 += NW) {
#pragma unroll
      for (int t = 0; t < NW; ++t) {
        if (s0 + t < nst) {  // wave-uniform
          const float4 bh = wh[t], bl = wl[t];
          if (s0 + t + NW < nst) {
            wh[t] = wp[(2 * (s0 + t + NW)) * 64];
            wl[t] = wp[(2 * (s0 + t + NW) + 1) * 64];
          }
          const float4 ah = *reinterpret_cast<const float4*>(ap + 32 * (s0 + t)), al = *reinterpret_cast<const float4*>(ap + lo + 32 * (s0 + t));
          acc = MFMA32H(al, bh, acc);
          acc = MFMA32H(ah, bl, acc);
          acc = MFMA32H(ah, bh, acc);
        }
      }
    }
    NSTAMP_V(5, acc[0]);
    float mw = LR ? ld_mw() : mw_j, cinv = LR ? ld_cinv() : cinv_j;
    const float* __restrict__ isc = scalar ? isc0 : isc1;
    // x_old of this lane's sixteen outputs and the tile's row scales, all requested before the first predicated store (inside the store
    // blocks every one of them was a round trip of its own: the compiler's wait in a predicated block is vmcnt(0), stores included)
    float xold[16], iscr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int rl = (q & 3) + 8 * (q >> 2) + 4 * hh;
      iscr[q] = isc[rl];
      if constexpr (LR) xold[q] = a.x_in[(size_t)min(n0 + rl, a.n_atoms - 1) * a.XSin + ((a.mix && col_ok_j) ? o_j : 0)];  // (unconditional, clamped)
      else xold[q] = xo[LR ? 0 : q];
    }
    if constexpr (LR) {
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(xold[q]));
      asm volatile("" : "+v"(mw), "+v"(cinv));
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int rl = (q & 3) + 8 * (q >> 2) + 4 * hh, ii = n0 + rl;
      if (col_ok_j && ii < a.n_atoms) {
        float v = (acc[q] * iscr[q]) * cinv;
        if (a.mix) v = mw * xold[q] + (1.f - mw) * v;  // hidden layers: XSin == XSo, x_in is x_old
        a.x_out[(size_t)ii * XSo + o_j] = v;
      }
    }
  }
  NSTAMP(6);
  NSTAMP_RT(7);
}

void node_print_stamps() {
#ifdef NH_TRACE
  static unsigned long long tr[2400][8];
  if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_nhtrace), sizeof(tr)) != hipSuccess) return;
  unsigned long long rt0 = ~0ull, rt1 = 0;
  int nw = 0;
  for (int g = 0; g < 2400; ++g)
    if (tr[g][0]) { rt0 = rt0 < tr[g][0] ? rt0 : tr[g][0]; rt1 = rt1 > tr[g][7] ? rt1 : tr[g][7]; ++nw; }
  if (!nw) return;
  fprintf(stderr, "node update trace: %d waves; first entry -> last exit %llu ticks of s_memrealtime (100 MHz)\n", nw, rt1 - rt0);
  fprintf(stderr, "  wg.wave: entry (ticks after first) | cycles after entry: loads arrived, phase 1a done, phase 1 done (2nd barrier passed), MFMAs done, exit | exit (ticks)\n");
  double sum[8] = {};
  for (int g = 0; g < 2400; ++g) {
    if (!tr[g][0]) continue;
    for (int i = 2; i < 7; ++i) sum[i] += (double)(tr[g][i] - tr[g][1]);
    sum[0] += (double)(tr[g][0] - rt0); sum[7] += (double)(tr[g][7] - rt0);
    if ((g >> 3) % 23 != 0) continue;
    fprintf(stderr, "  %3d.%d: %4llu |", g >> 3, g & 7, tr[g][0] - rt0);
    for (int i = 2; i < 7; ++i) fprintf(stderr, " %6lld", tr[g][i] ? (long long)(tr[g][i] - tr[g][1]) : 0ll);
    fprintf(stderr, " | %4llu\n", tr[g][7] - rt0);
  }
  fprintf(stderr, "  means: entry %.1f ticks |", sum[0] / nw);
  for (int i = 2; i < 7; ++i) fprintf(stderr, " %.0f", sum[i] / nw);
  fprintf(stderr, " | exit %.1f ticks\n", sum[7] / nw);
#endif
}

size_t node_update_h_lds_bytes(const NodeArgs& a) {
  return (size_t)2 * 32 * (a.K0h * 2 + 16) + (size_t)2 * 96 * (a.K1h * 2 + 16) + sizeof(float) * (32 * 32 + 64);
}
bool node_update_h_supported(const NodeArgs& a) {
  return ((a.mul0 + 31) >> 5) + 3 <= NH_T / 64 && a.wh0 != nullptr && a.wh1 != nullptr && a.kgx != nullptr && a.nt0 <= 5 && a.nt1 == 1 && a.mul1 <= 32 && (a.mul1 & 3) == 0 && a.mul1 >= 4 && (a.mul0 & 3) == 0 && a.mul0 >= 4 && (a.in0 & 3) == 0 && (a.XSin & 3) == 0 &&
         a.XSin <= 256 && (a.K0h & 15) == 0 && (a.K1h & 15) == 0 && a.K0h >= a.mul0 + a.in0 && a.K1h >= a.mul1 + a.in1 &&
         ((a.K0h - a.mul0 - a.in0) & 3) == 0 && ((a.K1h - a.mul1 - a.in1) & 3) == 0 && node_update_h_lds_bytes(a) <= 64 * 1024;
}
void launch_node_update_h(const NodeArgs& a, int cus, hipStream_t st) {
  const int grid = a.n_pad / 32;
  const bool lr = grid > cus && 2 * node_update_h_lds_bytes(a) <= 150 * 1024, fx = a.K0h == 240 && a.K1h == 64;
  const size_t lds = node_update_h_lds_bytes(a);
  if (lr && fx) hipLaunchKernelGGL((k_node_update_h<true, true>), dim3(grid), dim3(NH_T), lds, st, a);
  else if (lr) hipLaunchKernelGGL((k_node_update_h<true, false>), dim3(grid), dim3(NH_T), lds, st, a);
  else if (fx) hipLaunchKernelGGL((k_node_update_h<false, true>), dim3(grid), dim3(NH_T), lds, st, a);
  else hipLaunchKernelGGL((k_node_update_h<false, false>), dim3(grid), dim3(NH_T), lds, st, a);
}
