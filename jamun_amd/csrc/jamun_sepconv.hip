// jamun_sepconv.hip — SeparableConv (src/jamun/e3tools/nn/_conv.py:122-135, _tensor_product.py:8-58): the conv of a ConvBlock
// with the tensor product factored into a depth-wise part ("uvu": one weight per input channel and (input x sh -> output) triple,
// 2 n0 + 3 n1 = 336 weights per edge for 120x0e + 32x1e) and a point-wise o3.Linear shared by all edges.
//
// Unlike the fully connected product (28 992 weights per edge, re-associated by destination in jamun_conv_mf.hip) the per-edge
// weights are cheap here, so the REFERENCE association is the right one — but the [edge slots, 336] weight tensor never reaches
// memory (round 3 wrote and re-read 116 MB of it per layer):
//   k_sep_fused   one wave per destination: the weights of its <= 32 edge slots, w[slot][col] = h~[slot] . W2~[:, col] + b~[col], as ONE
//                 f16x3 MFMA tile per column tile (M = edge slots, K = 64 hidden units, N = 32 columns; h~ scaled by a static bound,
//                 W2~ balanced per column, both split hi + lo; 12 MFMAs), consumed straight from the accumulator: lane = channel,
//                 registers = edge slots, times the gathered source features and the edge's unit vector, summed over the slots:
//                     D0[u]   += wA x0[u]           D1[u][m] += wB x0[u] v[m]        D2[u'][m] += wC x1[u'][m]
//                     D3[u']  += wD (x1[u'] . v)    D4[u'][m] += wE (x1[u'] x v)[m]
//                 -> D [atom][K0 + 3 K1] (704 floats per atom).  W2~ sits in LDS (88 KB) for the whole launch.
//   k_sep_linear  m = Linear(D) (it commutes with the sum over edges): fp32 MFMAs, 32 atoms per workgroup, one output tile per wave;
//                 m goes to the partial-slab buffers of the node update (one slab), which divides by the in-degree (the mean), gates,
//                 and applies self-interaction + skip as for every other conv path.
// Constants folded into W2~ on the host: input noise scaling, path weights sqrt(2 l + 1), Clebsch-Gordan factors and the sqrt(3)
// of the l = 1 spherical harmonic.  Column order of W2~ (host: build_layer_separable): tiles of 32 columns — A (x0 -> 0e) in tiles
// 0..3, B (x0 x v -> 1e) in 4..7, C (x1 -> 1e) 8, D (x1 . v -> 0e) 9, E (x1 x v -> 1e) 10.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "jamun_internal.h"
#include "jamun_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define RFL(v) __builtin_amdgcn_readfirstlane(v)
#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
#define M3(ACC, AH_, AL_, BH_, BL_)   \
  ACC = MFMA32H(AL_, BH_, ACC);       \
  ACC = MFMA32H(AH_, BL_, ACC);       \
  ACC = MFMA32H(AH_, BH_, ACC)

#define SF_THREADS 512
#define SF_NCT 11                                  // column tiles of W2~
#define SF_W2_BYTES (4 * SF_NCT * 2 * 1024)        // [4 K-steps][11 column tiles][hi, lo][64 lanes x 16 B]
#define SF_LDS_BYTES (SF_W2_BYTES + 8 * 64 * 16 + 2 * 352 * 4)  // + slot records [8 waves][64] float4 + column factors and biases [2][352]

namespace {
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
}  // namespace

__global__ __launch_bounds__(SF_THREADS) void k_sep_fused(SepArgs a) {
  extern __shared__ float4 sf_lds[];
  const int tid = threadIdx.x, lane0 = tid & 63, wave = RFL(tid >> 6);
  for (int idx = tid; idx < SF_W2_BYTES / 16; idx += SF_THREADS) sf_lds[idx] = a.w2b[idx];
  float4* __restrict__ rec = sf_lds + SF_W2_BYTES / 16 + wave * 64;  // this wave's edge slots: (vx, vy, vz, source atom)
  // per column: inverse column scale (times the inverse of the static scale of h~) and the bias row of W2~ (read from LDS at the point
  // of use: 22 loop-invariant registers otherwise)
  float* __restrict__ colc = reinterpret_cast<float*>(sf_lds + SF_W2_BYTES / 16 + 8 * 64);
  {
    const float ish0 = pow2f(-a.sH);
    for (int idx = tid; idx < 352; idx += SF_THREADS) { colc[idx] = a.cfw[idx] * ish0; colc[352 + idx] = a.bias[idx]; }
  }
  __syncthreads();
  const int n0 = a.n0, n1 = a.n1, K0 = n0 + n1, K1 = n0 + 2 * n1, DW = K0 + 3 * K1;
  const int nA = RFL((n0 + 31) >> 5);
  const bool hasV = n1 > 0;
  const float sch = pow2f(a.sH);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0x7fffffff, 0x00020000);

  for (int d = blockIdx.x * 8 + wave; d < a.n_atoms; d += gridDim.x * 8) {
    int lane = lane0;
    asm volatile("" : "+v"(lane));  // (keeps ~40 lane-derived 64-bit addresses from being hoisted out of this loop and spilled)
    const int c = lane & 31, hh = lane >> 5;
    const bool vok = c < n1;
    auto bfrag = [&](int s, int ct, int p) { return sf_lds[((s * SF_NCT + ct) * 2 + p) * 64 + lane]; };
    const int dg = RFL(min(a.deg[d], a.S));
    {
      float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < dg) {
        const size_t slot = (size_t)d * a.S + lane;
        const float4 ge = a.egeo[slot];
        r4 = make_float4(ge.x, ge.y, ge.z, __int_as_float(a.esrc[slot] & 0x7fffffff));
      }
      rec[lane] = r4;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    float D0[4] = {0.f, 0.f, 0.f, 0.f}, D1[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    float D2[3] = {0.f, 0.f, 0.f}, D3 = 0.f, D4[3] = {0.f, 0.f, 0.f};
    const int n_mt = (dg + 31) >> 5;
    for (int mt = 0; mt < n_mt; ++mt) {
      // A fragments: h~ of slot 32 mt + c (the tile's row c), hidden units 16 s + 8 hh + j; scaled by the static 2^sH, split hi + lo
      float4 Ah[4], Al[4];
      {
        const int trow = 32 * mt + c;
        const bool rowok = trow < dg;
        // (one lane-dependent base, then wave-uniform strides: scalar offsets)
        const float* __restrict__ hp = a.h + (size_t)d * a.S + (rowok ? trow : 0) + (size_t)(8 * hh) * a.h_kstride;
        float hv[4][8];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) hv[s][j] = hp[(size_t)(16 * s + j) * a.h_kstride];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          unsigned ph[4], pl[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float e0 = rowok ? hv[s][2 * i] * sch : 0.f, e1 = rowok ? hv[s][2 * i + 1] * sch : 0.f;
            ph[i] = cvt_pk_f16(e0, e1);
            pl[i] = cvt_pk_f16(resid_lo(e0, ph[i]), resid_hi(e1, ph[i]));
          }
          Ah[s] = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
          Al[s] = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
        }
      }
      // the 16 accumulator rows of this lane: edge slots 32 mt + (q & 3) + 8 (q >> 2) + 4 hh
      int roff[16];
      unsigned rmask = 0u;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int t = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * hh;
        const bool ok = t < dg;
        roff[q] = ok ? __float_as_int(rec[t & 63].w) * a.XS : 0;
        rmask |= ok ? (1u << q) : 0u;
      }
      // (software pipeline over the tile pairs: the gathers of pair ct + 1 are requested before the products of pair ct are applied;
      // requesting the vector channels' 48 values behind the last pair as well spilled 240 bytes per lane)
      float x2[2][16], x1[16][3];  // (x2: ping-pong buffers of the scalar channels' gathers, indexed by the compile-time tile counter)
      // (buffer loads: one 32-bit byte offset per gather instead of a 64-bit address pair — sixteen gathers in flight twice over would
      // otherwise hold 64 registers of addresses)
      auto gather0 = [&](int ct, float (&dst)[16]) {
        const int u = 32 * ct + c;
        const bool uok = ct < nA && u < n0;
#pragma unroll
        for (int q = 0; q < 16; ++q) dst[q] = uok ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, 4 * (roff[q] + u), 0, 0)) : 0.f;
      };
      auto gather1 = [&]() {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int o = 4 * (roff[q] + n0 + 3 * (vok ? c : 0));
          x1[q][0] = vok ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, o, 0, 0)) : 0.f;
          x1[q][1] = vok ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, o + 4, 0, 0)) : 0.f;
          x1[q][2] = vok ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, o + 8, 0, 0)) : 0.f;
        }
      };
      gather0(0, x2[0]);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {  // (compile-time ct: the per-tile sums stay in registers)
        __builtin_amdgcn_sched_barrier(0);
        if (ct < nA) {
          f32x16 accA, accB;
#pragma unroll
          for (int q = 0; q < 16; ++q) { accA[q] = 0.f; accB[q] = 0.f; }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float4 ah = bfrag(s, ct, 0), al = bfrag(s, ct, 1), bh = bfrag(s, 4 + ct, 0), bl = bfrag(s, 4 + ct, 1);
            M3(accA, Ah[s], Al[s], ah, al);
            M3(accB, Ah[s], Al[s], bh, bl);
          }
          if (ct + 1 < 4 && ct + 1 < nA) gather0(ct + 1, x2[(ct + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          int z0 = 0;
          asm volatile("" : "+v"(z0));  // (opaque zero: the slot records are re-read per phase instead of living in 64 registers)
          const float cA = colc[32 * ct + c], oA = colc[352 + 32 * ct + c], cB = colc[128 + 32 * ct + c], oB = colc[352 + 128 + 32 * ct + c];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const bool ok = (rmask >> q) & 1u;
            const float4 r4 = rec[((32 * mt + (q & 3) + 8 * (q >> 2) + 4 * hh) & 63) + z0];
            const float wa = ok ? accA[q] * cA + oA : 0.f, wb = ok ? accB[q] * cB + oB : 0.f;
            D0[ct] = fmaf(wa, x2[ct & 1][q], D0[ct]);
            const float bx = wb * x2[ct & 1][q];
            D1[ct][0] = fmaf(bx, r4.x, D1[ct][0]);
            D1[ct][1] = fmaf(bx, r4.y, D1[ct][1]);
            D1[ct][2] = fmaf(bx, r4.z, D1[ct][2]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (hasV) {
        gather1();
        // C, D, E one after the other (one accumulator tile live at a time)
        __builtin_amdgcn_sched_barrier(0);
        {
          f32x16 acc;
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) { const float4 bh = bfrag(s, 8, 0), bl = bfrag(s, 8, 1); M3(acc, Ah[s], Al[s], bh, bl); }
          const float cf = colc[256 + c], bb = colc[352 + 256 + c];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float wc = ((rmask >> q) & 1u) ? acc[q] * cf + bb : 0.f;
            D2[0] = fmaf(wc, x1[q][0], D2[0]);
            D2[1] = fmaf(wc, x1[q][1], D2[1]);
            D2[2] = fmaf(wc, x1[q][2], D2[2]);
          }
        }
        {
          f32x16 acc;
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) { const float4 bh = bfrag(s, 9, 0), bl = bfrag(s, 9, 1); M3(acc, Ah[s], Al[s], bh, bl); }
          const float cf = colc[288 + c], bb = colc[352 + 288 + c];
          int z0 = 0;
          asm volatile("" : "+v"(z0));
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float4 r4 = rec[((32 * mt + (q & 3) + 8 * (q >> 2) + 4 * hh) & 63) + z0];
            const float wd = ((rmask >> q) & 1u) ? acc[q] * cf + bb : 0.f;
            D3 = fmaf(wd, fmaf(x1[q][2], r4.z, fmaf(x1[q][1], r4.y, x1[q][0] * r4.x)), D3);
          }
        }
        {
          f32x16 acc;
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) { const float4 bh = bfrag(s, 10, 0), bl = bfrag(s, 10, 1); M3(acc, Ah[s], Al[s], bh, bl); }
          const float cf = colc[320 + c], bb = colc[352 + 320 + c];
          int z0 = 0;
          asm volatile("" : "+v"(z0));
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float4 r4 = rec[((32 * mt + (q & 3) + 8 * (q >> 2) + 4 * hh) & 63) + z0];
            const float we = ((rmask >> q) & 1u) ? acc[q] * cf + bb : 0.f;
            const float vx = x1[q][0], vy = x1[q][1], vz = x1[q][2];
            // (x1 x v)[m] = x1[m+1] v[m+2] - x1[m+2] v[m+1]
            D4[0] = fmaf(we, vy * r4.z - vz * r4.y, D4[0]);
            D4[1] = fmaf(we, vz * r4.x - vx * r4.z, D4[1]);
            D4[2] = fmaf(we, vx * r4.y - vy * r4.x, D4[2]);
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();  // (the records are rewritten for the next destination)
    // the two half-waves hold the even / odd groups of four slots: one sum per channel, stored by the lower half.
    // Rows of the Linear's input: scalars [D0 (n0) | D3 (n1)], vectors per component m: [D1 (n0) | D2 (n1) | D4 (n1)]
    float* __restrict__ dr = a.D + (size_t)d * DW;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int u = 32 * ct + c;
      const float s0 = D0[ct] + __shfl_xor(D0[ct], 32, 64);
      const float s1 = D1[ct][0] + __shfl_xor(D1[ct][0], 32, 64), s2 = D1[ct][1] + __shfl_xor(D1[ct][1], 32, 64), s3 = D1[ct][2] + __shfl_xor(D1[ct][2], 32, 64);
      if (ct < nA && hh == 0 && u < n0) {
        dr[u] = s0;
        dr[K0 + u] = s1; dr[K0 + K1 + u] = s2; dr[K0 + 2 * K1 + u] = s3;
      }
    }
    if (hasV) {
      const float s3 = D3 + __shfl_xor(D3, 32, 64);
      float s2[3], s4[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) { s2[m] = D2[m] + __shfl_xor(D2[m], 32, 64); s4[m] = D4[m] + __shfl_xor(D4[m], 32, 64); }
      if (hh == 0 && vok) {
        dr[n0 + c] = s3;
#pragma unroll
        for (int m = 0; m < 3; ++m) { dr[K0 + m * K1 + n0 + c] = s2[m]; dr[K0 + m * K1 + n0 + n1 + c] = s4[m]; }
      }
    }
  }
}

// m = Linear(D): one workgroup = 32 atoms, its D tile in LDS (row stride DW + 1 floats: the A operand of v_mfma_f32_32x32x2_f32 is
// one float per lane, row = atom), wave j < nt0: scalar-output tile j (K0 rows of wl0), wave nt0 + m: vector plane m (K1 rows of wl1).
__global__ __launch_bounds__(512) void k_sep_linear(SepArgs a) {
  extern __shared__ float sl_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = RFL(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int n0 = a.n0, n1 = a.n1, K0 = n0 + n1, K1 = n0 + 2 * n1, DW = K0 + 3 * K1, LD = DW + 1;
  const int a0 = blockIdx.x * 32;
  // (the tile's 32 rows are contiguous in D: 16-byte loads, sixteen in flight per thread before the first LDS store — one load per loop
  // iteration with its store behind it was a chain of 64 round trips)
  {
    const int total4 = 32 * DW / 4;  // DW = 4 (n0 + 2 n1 + ...) is a multiple of 4 for the supported irreps (checked at launch)
    const float4* __restrict__ src = reinterpret_cast<const float4*>(a.D + (size_t)a0 * DW);
    const int lim4 = (int)(((size_t)min(32, a.n_atoms - a0) * DW) / 4);
    for (int base = tid; base < total4; base += 512 * 8) {
      float4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = base + 512 * i;
        v[i] = idx < lim4 ? src[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = base + 512 * i;
        if (idx < total4) {
          const int e = 4 * idx, row = e / DW, col = e - row * DW;
          float* __restrict__ d = sl_lds + row * LD + col;
          d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
      }
    }
  }
  __syncthreads();
  const int G0 = a.G0, G1 = a.G1;
  // K loop in groups of eight MFMAs with their sixteen operand loads issued first: one weight load per MFMA in program order is a chain of
  // ~90 dependent L2 round trips (first version: 43 us, as long as k_sep_fused)
  auto gemm = [&](const float* __restrict__ arow, const float* __restrict__ wcol, int K, int wstride, bool col_ok) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int s0 = 0; 2 * s0 < K; s0 += 8) {
      float av[8], bv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 2 * (s0 + i) + hh;
        const bool ok = k < K;
        av[i] = ok ? arow[k] : 0.f;
        bv[i] = (ok && col_ok) ? wcol[(size_t)k * wstride] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
    }
    return acc;
  };
  if (wave < a.nt0) {
    const int col = 32 * wave + r;
    const f32x16 acc = gemm(sl_lds + r * LD, a.wl0 + col, K0, G0, col < G0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
      if (a0 + row < a.n_atoms) a.partial0[(size_t)(a0 + row) * (a.nt0 * 32) + col] = acc[q];
    }
  } else if (wave < a.nt0 + 3 && a.nt1 == 1) {
    const int m = wave - a.nt0;
    const f32x16 acc = gemm(sl_lds + r * LD + K0 + m * K1, a.wl1 + r, K1, G1, r < G1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
      if (a0 + row < a.n_atoms) a.partial1[((size_t)(a0 + row) * 3 + m) * 32 + r] = acc[q];
    }
  }
}

// What the SeparableConv kernels can run: checked once in jamun_sampler_create (a model / topology outside these limits is rejected
// there with this message, not at the first forward) and again at launch.
const char* sep_conv_unsupported(int n0, int n1, int nt0, int nt1, int S, int edge_attr_dim) {
  if (n0 > 128 || n1 > 32) return "SeparableConv: input irreps wider than 128x0e + 32x1e";
  if (nt0 + 3 > 8 || nt1 != 1) return "SeparableConv: output irreps wider than five scalar tiles / 32 vector channels";
  if (edge_attr_dim != 64) return "SeparableConv: radial MLP with other than 64 hidden units";
  if (S > 64) return "SeparableConv: more than 64 edge slots per destination (32 radial neighbours + bonded in-edges; repeated bond listings count)";
  if ((size_t)32 * (n0 + n1 + 3 * (n0 + 2 * n1) + 1) * sizeof(float) > 150 * 1024) return "SeparableConv: a 32-atom tile of per-destination sums exceeds the LDS";
  if ((n0 + n1 + 3 * (n0 + 2 * n1)) % 4 != 0) return "SeparableConv: input irreps whose per-destination sums are not a multiple of four floats";
  return nullptr;
}

int sep_conv_set_max_lds() {
  if (hipFuncSetAttribute((const void*)k_sep_fused, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  if (hipFuncSetAttribute((const void*)k_sep_linear, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  return 0;
}

int launch_sep_conv(const SepArgs& a, int cus, hipStream_t st) {
  if (sep_conv_unsupported(a.n0, a.n1, a.nt0, a.nt1, a.S, 64)) return -1;
  const int grid = std::max(1, std::min(cus, (a.n_atoms + 7) / 8));
  hipLaunchKernelGGL(k_sep_fused, dim3(grid), dim3(SF_THREADS), SF_LDS_BYTES, st, a);
  const int DW = a.n0 + a.n1 + 3 * (a.n0 + 2 * a.n1);
  hipLaunchKernelGGL(k_sep_linear, dim3((a.n_atoms + 31) / 32), dim3(512), (size_t)32 * (DW + 1) * sizeof(float), st, a);
  return 0;
}
