// jamun_conv_mf.hip — conv contraction of a hidden layer with the A operand formed ON THE MATRIX CORES and chained, in registers,
// into the contraction with the weights (f16x3 throughout).  For tiles whose source span fits one K = 64 window (molecules up to
// ~30 atoms: two or three per 32-destination tile).
//
// Same mathematics as jamun_conv_dg.hip (src/jamun/e3tools/nn/_conv.py:93-119, re-associated by destination):
//   m[(i,c)][w] = sum_k sum_u A_k[(i,c)][u] W~[(k,u)][w],   A_k[(i,c)][u] = sum_{e = (j -> i)} h~_e[k] zeta_e[c][u].
// k_conv_dg forms A_k edge by edge with vector FMAs: 19 FMAs per (edge, channel lane) and hidden unit, ~1800 vector instructions
// per SIMD and k-step against 78 MFMAs — since the contraction moved to f16 MFMAs (16x the fp32 matrix rate) the forming waves are
// the whole k-step (measured: profiles/r3b_*).  Here forming is a dense product over the source rows of the tile's span,
//   A_k[(i,c)][u] = sum_j C_k[c][i][j] x_j[u],    C_k[c][i][j] = h~_(j->i)[k] (1, vx, vy, vz)_c   (zero where there is no edge),
// i.e. GEMMs with M = channels, N = 32 destinations, K = 64 source rows: 4 K-steps of v_mfma_f32_32x32x16_f16 per 32 x 32 tile,
// x 3 for the f16x3 split of both operands.  A quarter of C_k is non-zero for 17-atom molecules, and the matrix pipe is still 5x
// faster at it than the VALU at the sparse form: 414 MFMAs (13.2 k matrix cycles over 4 SIMDs) per (tile, k) for forming AND
// contraction.
//
// Chaining.  The forming MFMA computes the TRANSPOSED tile D^T[u][i] (A operand = x^T rows from LDS, B operand = C_k rows from
// LDS), whose accumulator layout — lane = destination i, registers = channels u — IS the A-operand layout of the contraction
// (lane = row i, 8 K values per lane) up to a fixed permutation of the K index, which the host folds into the weight blocks.  So
// the formed tile never visits LDS: split into hi + lo halves in registers (2 v_cvt_pk + 2 v_fma_mix per pair), then multiplied
// with the weight fragments streamed from L2.  The contraction is K-split over the four matrix waves (each wave owns the channels
// it formed and accumulates ALL 152 scalar-output columns; the four partial tiles are summed through LDS at the segment end):
//   wave 0      scalar channels  0..31  | dot(x1, v) (32 channels, K = 3 x 64)            -> 5 scalar-output tiles
//   wave 1 + m  scalar channels 32(1+m).. | x1[m] and (x1 x v)[m] (K = 64, 2 x 64)        -> 5 scalar-output tiles | vector plane m
//               + the T term  out_m[i][w'] += sum_j C_k[v_m][i][j] T_k[j][w']  straight into the plane accumulator
// Helper waves 4..7 build C_k (one edge per lane: coefficient x geometry, scaled, split, 8 two-byte LDS stores) and stage T_k
// (transposed by the pre-pass k_tprod_h) for the NEXT hidden unit into the other half of a double buffer; one barrier per k.
//
// Scales (all powers of two, exact).  An f16 pair hi + lo carries 22 bits only while lo stays a NORMAL half (>= 2^-14), so every
// operand is scaled to the TOP of the f16 range: x rows of the span by 2^sX (largest |x| -> [2^13, 2^14)), coefficients by 2^sC
// (static: 3 max|h~| -> 2^14), weights by 2^sB (host: largest -> 2^14), T by 2^sT (bound |x| x largest column sum of the T
// weights -> 2^14).  A formed tile is then below in-degree x 2^28 in its fp32 accumulator and is brought below 2^14 by a
// per-destination power of two (one v_mul per value, in the lane that owns the destination) before its own split.  The T term has
// its own accumulator (its scale does not depend on the destination); both are scaled back at the segment end.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdio.h>

#include "jamun_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MF_THREADS 512
#define MF_ROWB 144                      // bytes per row of a K = 64 plane of halves: 128 + 16 (rows 16 B apart mod 256: conflict-free b128 reads)
#define MF_X0H 0                         // [128 channels][64 j] hi | lo
#define MF_X0L (128 * MF_ROWB)
#define MF_X1H (2 * 128 * MF_ROWB)       // [3 m][32 u][64 j] hi | lo
#define MF_X1L (MF_X1H + 96 * MF_ROWB)
#define MF_TT (MF_X1L + 96 * MF_ROWB)    // [2 buffers][hi, lo][32 w'][64 j]
#define MF_TTB (2 * 32 * MF_ROWB)
#define MF_C (MF_TT + 2 * MF_TTB)        // [2 buffers][4 components][hi, lo][32 i][64 j]
#define MF_CB (4 * 2 * 32 * MF_ROWB)
#define MF_MISC (MF_C + 2 * MF_CB)       // deg[32] | xmax
#define MF_LDS_BYTES (MF_MISC + 144)
#define MF_PL (32 * MF_ROWB)             // hi -> lo plane of a T / C tile

#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
#define M3(ACC, AH_, AL_, BH_, BL_)   \
  ACC = MFMA32H(AL_, BH_, ACC);       \
  ACC = MFMA32H(AH_, BL_, ACC);       \
  ACC = MFMA32H(AH_, BH_, ACC)
#define RFL(v) __builtin_amdgcn_readfirstlane(v)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

namespace {

__device__ __forceinline__ float4 lds_f4(int addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 v = *(const __attribute__((address_space(3))) f32x4*)(unsigned)addr;
  return make_float4(v.x, v.y, v.z, v.w);
#else
  (void)addr;
  return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}
__device__ __forceinline__ int lds_addr(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
#else
  (void)p;
  return 0;
#endif
}
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float resid_lo(float a, unsigned pk) {  // a - float(pk[15:0])
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(a));
  return r;
}
__device__ __forceinline__ float resid_hi(float a, unsigned pk) {  // a - float(pk[31:16])
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pk), "v"(a));
  return r;
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
__device__ __forceinline__ int exp_above(float v) { return (int)((__float_as_uint(v) >> 23) & 0xffu) - 126; }  // v < 2^exp_above(v)
__device__ __forceinline__ int clamp40(int s) { return max(-40, min(40, s)); }
__device__ __forceinline__ int clamp100(int s) { return max(-100, min(100, s)); }

}  // namespace

// SPD: edge slots per destination handled by one lane group of the helper waves (32 for edge strides up to 32, else 64)
template <int SPD>
__global__ __launch_bounds__(MF_THREADS) void k_conv_mf(MfArgs a) {
  extern __shared__ float4 lds4[];
  char* __restrict__ lds = reinterpret_cast<char*>(lds4);
  const int L0 = lds_addr(lds);
  int* __restrict__ deg_lds = reinterpret_cast<int*>(lds + MF_MISC);
  unsigned* __restrict__ xmax_lds = reinterpret_cast<unsigned*>(lds + MF_MISC + 128);
  float* __restrict__ ST0 = reinterpret_cast<float*>(lds + MF_TT);  // segment end: [4 waves][32][160] partial scalar-output tiles
  float* __restrict__ ST1 = reinterpret_cast<float*>(lds);          //              [32][96] vector planes

  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = RFL(tid0 >> 6);
  const bool is_mat = wave < 4;

  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int4 sg0 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2];
    const int4 sg1 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2 + 1];
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    int lane = lane0;
    asm volatile("" : "+v"(lane));  // (keeps lane-derived addresses from being hoisted out of the segment loop and spilled)
    const int tid = wave * 64 + lane;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) { kk = kk < 0 ? 0 : (kk < nk ? kk : nk - 1); return kk < k_run ? k_begin + kk : k_extra; };
    const int2 t_at = a.tile_atoms[tile];
    const int n0 = RFL(t_at.x), n_dst = RFL(t_at.y);
    const int2 span = a.tile_span[tile];
    const int s_lo = RFL(span.x), rows = RFL(span.y) - s_lo;
    const int s_base = s_lo & ~1, off = s_lo - s_base;  // window of 64 source rows from an even atom (8-byte T loads)

    // ---- segment prologue (all threads): zero the coefficient tiles, stage the span's rows transposed and split
    for (int idx = tid; idx < 2 * MF_CB / 16; idx += MF_THREADS) reinterpret_cast<float4*>(lds + MF_C)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 144) {  // channel rows 120..127 of the scalar block (inputs of the last K-step that do not exist)
      const int pl = tid / 72, q = tid - pl * 72;
      *reinterpret_cast<float4*>(lds + (pl ? MF_X0L : MF_X0H) + 120 * MF_ROWB + 16 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < 32) deg_lds[tid] = (tid < n_dst) ? a.deg[n0 + tid] : 0;
    if (tid == 0) *xmax_lds = 0u;
    float4 va[4], vb[4];
    float mx = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // item = (pair of source rows jp, float4 column c4): 32 x 54 items
      const int it = tid + MF_THREADS * q, c4 = it >> 5, jp = it & 31;
      const int j0 = 2 * jp - off, j1 = j0 + 1;  // rows relative to the span
      va[q] = vb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c4 < 54) {
        if (j0 >= 0 && j0 < rows) va[q] = *reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + j0) * a.XS + 4 * c4);
        if (j1 >= 0 && j1 < rows) vb[q] = *reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + j1) * a.XS + 4 * c4);
      }
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(va[q].x), fabsf(va[q].y)), fmaxf(fabsf(va[q].z), fabsf(va[q].w))));
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(vb[q].x), fabsf(vb[q].y)), fmaxf(fabsf(vb[q].z), fabsf(vb[q].w))));
    }
    LDS_BARRIER();  // xmax zeroed (and the previous segment's staging tile read)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) atomicMax(xmax_lds, __float_as_uint(mx));
    LDS_BARRIER();
    const float xm = __uint_as_float(*xmax_lds);
    const int sX = RFL(xm > 0.f ? clamp40(14 - exp_above(xm)) : 0);
    {
      const float scx = pow2f(sX);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int it = tid + MF_THREADS * q, c4 = it >> 5, jp = it & 31;
        if (c4 < 54) {
          const float ea[4] = {va[q].x, va[q].y, va[q].z, va[q].w}, eb[4] = {vb[q].x, vb[q].y, vb[q].z, vb[q].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int f = 4 * c4 + e;
            int rowb;  // byte offset of the channel's hi row
            if (f < 120) rowb = MF_X0H + f * MF_ROWB;
            else { const int e1 = f - 120, u = e1 / 3, m = e1 - 3 * u; rowb = MF_X1H + (m * 32 + u) * MF_ROWB; }
            const int lo_off = f < 120 ? (MF_X0L - MF_X0H) : (MF_X1L - MF_X1H);
            const float a0 = ea[e] * scx, b0 = eb[e] * scx;
            const unsigned ph = cvt_pk_f16(a0, b0), pl = cvt_pk_f16(resid_lo(a0, ph), resid_hi(b0, ph));
            *reinterpret_cast<unsigned*>(lds + rowb + 4 * jp) = ph;
            *reinterpret_cast<unsigned*>(lds + rowb + lo_off + 4 * jp) = pl;
          }
        }
      }
    }

    const int r = lane & 31, hh = lane >> 5;
    if (is_mat) {
      // =========================================== MATRIX waves ===========================================
      const int w = wave;
      f32x16 accS[5], accP, accT;
#pragma unroll
      for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) accS[n][q] = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) accP[q] = accT[q] = 0.f;
      // formed values are below in-degree x 2^28: x 2^-(14 + e(in-degree)) of this lane's destination -> below 2^14
      const int edeg_r = deg_lds[r] > 0 ? exp_above((float)deg_lds[r]) : 1;
      const float rs = pow2f(-14 - edeg_r);
      const int fo = r * MF_ROWB + 16 * hh + L0;            // this lane's fragment offset inside a 32-row tile
      const int x0a = MF_X0H + 32 * w * MF_ROWB + fo;       // its row of the wave's 32 scalar channels (lo plane: + MF_X0L)
      // weight blocks: buffer loads with ONE address register (the lane's 16 bytes) and the block offset in a scalar register
      const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.wm), 0, 0x7fffffff, 0x00020000);
      const int wvo = lane * 16;
      auto wstream = [&](int k) { return (k * 4 + w) * (40 * 1024); };  // byte offset of this wave's stream of hidden unit k
      auto wload = [&](int so) { return __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so, 0); };
      // forming: F[u][i] += sum_j x^T[u][j] C[i][j] over the four K-steps (x fragment at xa / xa + xlo, C at ca / ca + MF_PL)
      auto form = [&](f32x16& F, int xa, int xlo, int ca) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float4 ah = lds_f4(xa + 32 * s), al = lds_f4(xa + xlo + 32 * s), bh = lds_f4(ca + 32 * s), bl = lds_f4(ca + MF_PL + 32 * s);
          M3(F, ah, al, bh, bl);
        }
      };
      auto split = [&](const f32x16& F, float4 (&Ah)[2], float4 (&Al)[2]) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          unsigned ph[4], pl[4];
#pragma unroll
          for (int p2 = 0; p2 < 4; ++p2) {
            const float v0 = F[8 * s2 + 2 * p2] * rs, v1 = F[8 * s2 + 2 * p2 + 1] * rs;
            ph[p2] = cvt_pk_f16(v0, v1);
            pl[p2] = cvt_pk_f16(resid_lo(v0, ph[p2]), resid_hi(v1, ph[p2]));
          }
          Ah[s2] = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
          Al[s2] = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
        }
      };
      const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

      if (w == 0) {
        // ---- wave 0: scalar channels 0..31 and the dot inputs; weight stream of 40 blocks per k, ring of 20
        constexpr int NB = 40, R = 20;
        u32x4 RB[R];
        {
          const int c0 = wstream(k_of(0));
#pragma unroll
          for (int p = 0; p < R; ++p) RB[p] = wload(c0 + p * 1024);
        }
        LDS_BARRIER();  // C(k0), T(k0) complete
        for (int it = 0; it < nk; ++it) {
          const int cur = wstream(k_of(it)), nxt = wstream(k_of(it + 1));
          const int cb = MF_C + (it & 1) * MF_CB + fo;
          auto refill = [&](int p) {  // stream position p + R into the slot of position p
            RB[p % R] = (p + R < NB) ? wload(cur + (p + R) * 1024) : wload(nxt + (p + R - NB) * 1024);
          };
          float4 Ah[2], Al[2];
          {
            f32x16 FX = zero16;
            form(FX, x0a, MF_X0L - MF_X0H, cb);
            split(FX, Ah, Al);
          }
#pragma unroll
          for (int n = 0; n < 5; ++n)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const int p = 2 * (2 * n + s2);
              M3(accS[n], Ah[s2], Al[s2], RB[p % R], RB[(p + 1) % R]);
              refill(p); refill(p + 1);
            }
          {
            f32x16 FD = zero16;
#pragma unroll
            for (int m = 0; m < 3; ++m) form(FD, MF_X1H + (m * 32) * MF_ROWB + fo, MF_X1L - MF_X1H, cb + (1 + m) * 2 * MF_PL);
            split(FD, Ah, Al);
          }
#pragma unroll
          for (int n = 0; n < 5; ++n)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const int p = 20 + 2 * (2 * n + s2);
              M3(accS[n], Ah[s2], Al[s2], RB[p % R], RB[(p + 1) % R]);
              refill(p); refill(p + 1);
            }
          LDS_BARRIER();
        }
      } else {
        // ---- waves 1..3: scalar channels 32 w.., vector plane m = w - 1; weight stream of 28 blocks per k, ring of 14
        constexpr int NB = 28, R = 14;
        const int m = w - 1, m1 = (m + 1) % 3, m2 = (m + 2) % 3;
        u32x4 RB[R];
        {
          const int c0 = wstream(k_of(0));
#pragma unroll
          for (int p = 0; p < R; ++p) RB[p] = wload(c0 + p * 1024);
        }
        const int x1a = MF_X1H + fo;  // + plane * 32 rows
        LDS_BARRIER();  // C(k0), T(k0) complete
        for (int it = 0; it < nk; ++it) {
          const int cur = wstream(k_of(it)), nxt = wstream(k_of(it + 1));
          const int cb = MF_C + (it & 1) * MF_CB + fo;
          const int tb = MF_TT + (it & 1) * MF_TTB + fo;
          auto refill = [&](int p) { RB[p % R] = (p + R < NB) ? wload(cur + (p + R) * 1024) : wload(nxt + (p + R - NB) * 1024); };
          float4 Ah[2], Al[2];
          {
            f32x16 FX = zero16;
            form(FX, x0a, MF_X0L - MF_X0H, cb);
            split(FX, Ah, Al);
          }
#pragma unroll
          for (int n = 0; n < 5; ++n)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              const int p = 2 * (2 * n + s2);
              M3(accS[n], Ah[s2], Al[s2], RB[p % R], RB[(p + 1) % R]);
              refill(p); refill(p + 1);
            }
          {
            f32x16 FA = zero16;
            form(FA, x1a + m * 32 * MF_ROWB, MF_X1L - MF_X1H, cb);
            split(FA, Ah, Al);
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int p = 20 + 2 * s2;
            M3(accP, Ah[s2], Al[s2], RB[p % R], RB[(p + 1) % R]);
            refill(p); refill(p + 1);
          }
          {
            // (x1 x v)[m] = x1[m+1] v[m+2] - x1[m+2] v[m+1]
            f32x16 F1 = zero16, F2 = zero16;
            form(F1, x1a + m1 * 32 * MF_ROWB, MF_X1L - MF_X1H, cb + (1 + m2) * 2 * MF_PL);
            form(F2, x1a + m2 * 32 * MF_ROWB, MF_X1L - MF_X1H, cb + (1 + m1) * 2 * MF_PL);
#pragma unroll
            for (int q = 0; q < 16; ++q) F1[q] -= F2[q];
            split(F1, Ah, Al);
          }
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int p = 24 + 2 * s2;
            M3(accP, Ah[s2], Al[s2], RB[p % R], RB[(p + 1) % R]);
            refill(p); refill(p + 1);
          }
          // T term: out_m[i][w'] += sum_j C[v_m][i][j] T_k[j][w']  (A = coefficient rows, B = T^T rows; own accumulator: scale 2^(sC + sT))
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int ca = cb + (1 + m) * 2 * MF_PL;
            const float4 ch = lds_f4(ca + 32 * s), cl = lds_f4(ca + MF_PL + 32 * s), th = lds_f4(tb + 32 * s), tl = lds_f4(tb + MF_PL + 32 * s);
            M3(accT, ch, cl, th, tl);
          }
          LDS_BARRIER();
        }
      }
      // ---- segment end: back to true scale, partial tiles -> LDS staging (the coefficient / T / x tiles are dead)
      // (row i of the contraction carries 2^(sX + sC + sB - 14 - e(in-degree_i)), the T term 2^(sX + sC + sTw); two factors each, so
      // that no intermediate power of two leaves the fp32 range)
      const float i1 = pow2f(clamp100(-(sX + a.sC))), iT2 = pow2f(clamp100(-a.sTw));
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        const int edeg = deg_lds[row] > 0 ? exp_above((float)deg_lds[row]) : 1;
        const float i2 = pow2f(clamp100(14 + edeg - a.sB));
#pragma unroll
        for (int n = 0; n < 5; ++n) ST0[(w * 32 + row) * 160 + 32 * n + r] = (accS[n][q] * i1) * i2;
        if (w > 0) ST1[row * 96 + (w - 1) * 32 + r] = (accP[q] * i1) * i2 + (accT[q] * i1) * iT2;
      }
    } else {
      // =========================================== HELPER waves ===========================================
      constexpr int DPP = 256 / SPD, NP = 32 / DPP;  // destinations per pass, passes
      const int hl = tid - 256, grp = hl / SPD, t = hl % SPD;
      const int gsh = (SPD == 32) ? 32 * (lane >> 5) : 0;  // this group's bit offset inside the wave's ballot
      int coff[NP], hsl[NP], tw0[NP], tw1[NP];
      float evx[NP], evy[NP], evz[NP];
      bool any_tw = false;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int i = p * DPP + grp;
        const int dg = deg_lds[i];
        const int slot = (n0 + i) * a.S + t;
        const bool in = t < dg && t < a.S;
        const int sj = in ? a.esrc[slot] : 0;
        float4 ge = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) ge = a.egeo[slot];
        const bool bonded = in && sj < 0;  // bit 31
        const int jraw = sj & 0x7fffffff, jl = jraw - s_base;
        const bool valid = in && jl >= 0 && jl < 64;
        bool active = valid;
        tw0[p] = tw1[p] = -1;
        // Several edges of one (source, destination) pair (a bonded pair inside the cutoff has a radial and a bonded edge:
        // src/jamun/model/denoiser.py:152) share ONE entry of the coefficient tile: the first slot of the pair owns it and adds
        // the h~ of the others (bonded edges are the last slots of a destination: k_geom).
        const unsigned long long balb = __ballot(bonded);
        const unsigned long long gmask = (SPD == 32) ? 0xffffffffull : ~0ull;
        const int nb = __popcll((balb >> gsh) & gmask);
        const int nb_max = RFL(max(__popcll(balb & 0xffffffffull), SPD == 32 ? __popcll(balb >> 32) : __popcll(balb)));
        for (int b = 0; b < nb_max; ++b) {
          const int lb = dg - nb + b;  // slot of this group's b-th bonded edge
          const int jb = __shfl(jraw, (lb & (SPD - 1)) + gsh, 64);
          const bool match = b < nb && valid && t < lb && jraw == jb;
          const unsigned long long mb = (__ballot(match) >> gsh) & gmask;
          if (b < nb && mb != 0ull) {
            const int first = __ffsll((long long)mb) - 1;
            if (t == lb) active = false;
            if (t == first) {
              const int sl = (n0 + i) * a.S + lb;
              if (tw0[p] < 0) tw0[p] = sl;
              else if (tw1[p] < 0) tw1[p] = sl;
              else atomicOr(a.err, 1);  // more than three edges of one pair: not representable here
            }
          }
        }
        any_tw = any_tw || tw0[p] >= 0;
        coff[p] = active ? i * MF_ROWB + 2 * jl : -1;
        hsl[p] = active ? slot : n0 * a.S;
        evx[p] = ge.x; evy[p] = ge.y; evz[p] = ge.z;
      }
      const bool wave_tw = __ballot(any_tw) != 0ull;
      const float scC = pow2f(a.sC), scT = pow2f(clamp100(sX + a.sTw));
      float hv[NP];
      float2 tv[4];
      auto load_k = [&](int k) {  // h~ of this lane's edges and this lane's T elements of hidden unit k
        const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          float c = hk[hsl[p]];
          if (wave_tw) {
            if (tw0[p] >= 0) c += hk[tw0[p]];
            if (tw1[p] >= 0) c += hk[tw1[p]];
          }
          hv[p] = c;
        }
        const float* __restrict__ tk = a.Tt + (size_t)k * 32 * a.t_stride + s_base;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int item = hl + 256 * q, wp = item >> 5, jp = item & 31;
          tv[q] = *reinterpret_cast<const float2*>(tk + (size_t)wp * a.t_stride + 2 * jp);
        }
      };
      auto build = [&](int buf) {  // the loaded hidden unit -> coefficient tiles and T tile of buffer `buf`
        char* __restrict__ cbuf = lds + MF_C + buf * MF_CB;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if (coff[p] >= 0) {
            const float c0 = hv[p] * scC, c1 = c0 * evx[p], c2 = c0 * evy[p], c3 = c0 * evz[p];
            const unsigned h01 = cvt_pk_f16(c0, c1), h23 = cvt_pk_f16(c2, c3);
            const unsigned l01 = cvt_pk_f16(resid_lo(c0, h01), resid_hi(c1, h01)), l23 = cvt_pk_f16(resid_lo(c2, h23), resid_hi(c3, h23));
            char* __restrict__ d = cbuf + coff[p];
            *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h01 & 0xffffu);
            *reinterpret_cast<unsigned short*>(d + MF_PL) = (unsigned short)(l01 & 0xffffu);
            *reinterpret_cast<unsigned short*>(d + 2 * MF_PL) = (unsigned short)(h01 >> 16);
            *reinterpret_cast<unsigned short*>(d + 3 * MF_PL) = (unsigned short)(l01 >> 16);
            *reinterpret_cast<unsigned short*>(d + 4 * MF_PL) = (unsigned short)(h23 & 0xffffu);
            *reinterpret_cast<unsigned short*>(d + 5 * MF_PL) = (unsigned short)(l23 & 0xffffu);
            *reinterpret_cast<unsigned short*>(d + 6 * MF_PL) = (unsigned short)(h23 >> 16);
            *reinterpret_cast<unsigned short*>(d + 7 * MF_PL) = (unsigned short)(l23 >> 16);
          }
        }
        char* __restrict__ tbuf = lds + MF_TT + buf * MF_TTB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int item = hl + 256 * q, wp = item >> 5, jp = item & 31;
          const int j0 = 2 * jp - off;
          const float t0 = (j0 >= 0 && j0 < rows) ? tv[q].x * scT : 0.f, t1 = (j0 + 1 >= 0 && j0 + 1 < rows) ? tv[q].y * scT : 0.f;
          const unsigned ph = cvt_pk_f16(t0, t1), pl = cvt_pk_f16(resid_lo(t0, ph), resid_hi(t1, ph));
          *reinterpret_cast<unsigned*>(tbuf + wp * MF_ROWB + 4 * jp) = ph;
          *reinterpret_cast<unsigned*>(tbuf + MF_PL + wp * MF_ROWB + 4 * jp) = pl;
        }
      };
      load_k(k_of(0));
      build(0);
      load_k(k_of(1));
      LDS_BARRIER();  // C(k0), T(k0) complete
      for (int it = 0; it < nk; ++it) {
        if (it + 1 < nk) {
          build((it + 1) & 1);
          load_k(k_of(it + 2));
        }
        LDS_BARRIER();
      }
    }
    // ---- all threads: sum of the four K-partial tiles -> partial slab of this segment, coalesced 16-byte stores
    LDS_BARRIER();
    {
      float* __restrict__ p0 = a.partial0 + ((size_t)slab * a.n_pad + n0) * (size_t)(a.nt0 * 32);
      float* __restrict__ p1 = a.partial1 + ((size_t)slab * a.n_pad + n0) * 96;
      for (int idx = tid; idx < 32 * 40; idx += MF_THREADS) {
        const int row = idx / 40, c4 = idx - row * 40;
        const float* __restrict__ q0 = ST0 + row * 160 + 4 * c4;
        const float4 a0 = *reinterpret_cast<const float4*>(q0), a1 = *reinterpret_cast<const float4*>(q0 + 5120),
                     a2 = *reinterpret_cast<const float4*>(q0 + 10240), a3 = *reinterpret_cast<const float4*>(q0 + 15360);
        const float4 v = make_float4(((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y, ((a0.z + a1.z) + a2.z) + a3.z,
                                     ((a0.w + a1.w) + a2.w) + a3.w);
        if (row < n_dst) *reinterpret_cast<float4*>(p0 + row * 160 + 4 * c4) = v;
      }
      for (int idx = tid; idx < 32 * 24; idx += MF_THREADS) {
        const int row = idx / 24, c4 = idx - row * 24;
        if (row < n_dst) *reinterpret_cast<float4*>(p1 + row * 96 + 4 * c4) = *reinterpret_cast<const float4*>(ST1 + row * 96 + 4 * c4);
      }
    }
    LDS_BARRIER();  // the next segment rewrites the tiles
  }
}

size_t conv_mf_lds_bytes() { return MF_LDS_BYTES; }

int launch_conv_mf(const MfArgs& a, int grid, hipStream_t st) {
  if (a.XS != 216 || a.nt0 != 5 || a.S > 64 || (a.t_stride & 1)) return -1;
  if (a.S <= 32) hipLaunchKernelGGL((k_conv_mf<32>), dim3(grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((k_conv_mf<64>), dim3(grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
  return 0;
}

int conv_mf_set_max_lds() {
  const void* fns[2] = {(const void*)k_conv_mf<32>, (const void*)k_conv_mf<64>};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  return 0;
}
