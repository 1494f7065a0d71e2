// jamun_conv_dg.hip — conv contraction of a hidden layer, destination-grouped, with the A operand formed on the VECTOR
// ALUs by dedicated waves while other waves run the contraction on the matrix cores.
//
// Same mathematics as jamun_conv.hip / jamun_conv_fused.hip (src/jamun/e3tools/nn/_conv.py:93-119, re-associated by
// destination):   m[(i,c)][w] = sum_k sum_u A_k[(i,c)][u] W~[(k,u)][w],   A_k[(i,c)][u] = sum_{e->i} h~_e[k] zeta_e[c][u].
// jamun_conv_fused.hip forms A_k with MFMAs from dense (source x destination) coefficient tiles; for small molecules those
// tiles are block diagonal and 2/3 of the forming MFMAs multiply structural zeros (and forming cost grows with the number of
// source rows of a tile).  Here A_k is formed EDGE BY EDGE on the VALU — work proportional to the real edges — and the
// matrix cores run only the contraction with the weights:
//
//   12 waves per workgroup (one persistent workgroup per CU), two roles:
//   * waves 0..3  (one per SIMD)  MATRIX waves.  Wave w owns scalar-output tile w (32 of the 152 columns), a quarter of the
//     K range of scalar-output tile 4 and the 16x16 sub-tiles (rh = w&1, ch = w>>1) of the three vector planes.  All four
//     run the same program on different addresses.
//   * waves 4..11 (two per SIMD)  FORMING waves.  Each owns 4 destination atoms, two at a time: lane = (half h -> which of
//     the two destinations, u = channel).  Per edge e = (j -> i) of its destination the half-wave reads the source row
//     x_j from LDS (coalesced: the lanes of a half read consecutive channels of ONE row, so no bank conflict and no
//     per-lane gather) and accumulates  c x0_j[u],  c x1_j[u][m],  sum_m (c v_m) x1_j[u][m],  c (x1_j[u] x v)[m]  with the
//     per-edge coefficients c = h~_e[k], c v_m broadcast from a small per-wave LDS table, plus the vector-row
//     contribution of the scalar inputs through the re-association  out_m[i][w] += sum_e (c v_m) T_k[j][w],
//     T_k[j][w] = sum_u x0_j[u] W~[(k,u)][w]  (T_k comes from the pre-pass k_tprod at the end of this file: 2 x 120 x 32 FLOP
//     per source atom and k, once per layer, instead of 3 x per destination).
//   MFMA and VALU instructions of a SIMD exclude each other in time on gfx950 (measured: DESIGN.md 3.3), so the forming work is
//   NOT hidden behind the contraction — a k-step costs the sum of both — but it is proportional to the real edges, and the
//   LDS latency of the forming waves is filled with MFMAs.
//
//   Four schedules of the hidden units of a segment (template parameter MODE, chosen by the host from the tile spans):
//     0  two phases per k, single-buffered A tiles (spans up to ~80 rows):
//          P1(k): matrix waves  X(k) x W  (scalar inputs -> scalar rows, K = 120);        forming waves write Y(k)
//          P2(k): matrix waves  Y(k) x W  (dot -> scalar rows; x1, cross -> vector rows);   forming waves write X(k+1)
//     2  one phase per k, double-buffered A tiles, X and Y formed in ONE pass over the edges (spans up to ~52 rows)
//     3  as 2 with a double-buffered X tile and ONE Y tile: Y(k) x W first, mid-step barrier, then X(k) x W (up to ~73 rows)
//     1  large molecules (up to ~176 rows): two passes over the k run, vector blocks resident, then scalar channels
//   MFMA work per (32-destination tile, k): 476 units of v_mfma_f32_32x32x2 (64 cycles), against 637 algorithmic.
//
// Work distribution, partial slabs, fixed summation order: as jamun_conv_fused.hip (host-built segment lists).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <type_traits>
#include <stdio.h>

#include "jamun_internal.h"
#include "jamun_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

#define DG_WAVES 12   // 4 matrix waves + 8 forming waves (two forming waves per SIMD co-issue VALU and LDS instructions; with one
                      // per SIMD the forming phases took 25 % longer)
#define DG_NDP 2     // destination pairs per forming wave: 32 / (2 * forming waves)
#define DG_THREADS (64 * DG_WAVES)
#define DG_XST_RES 252  // resident mode, xs row: x0 [0,120) | [u][4] = (x1_j[u].x, T_k[j][u], x1_j[u].y, .z) [120,248) | pad: 4 * 63 floats
#define DG_XST_ALT 132  // alternating mode, xs row = EITHER x0 [0,120) + zeros (phase P2) OR the [u][4] block [0,128) (phase P1): 4 * 33 floats
#define DG_RS_MAX_RES 128
#define DG_RS_MAX_ALT 192
#define DG_XS0 124   // X tile row stride (120 + 4):  4 * 31
#define DG_XS0H 124  // f16x3 path: 120 inputs as 15 chunks of [8 hi halves | 8 lo halves] = 480 B, + 16 B of zeros: rows 16 B apart mod 256.  The
                     // last group of 16 inputs reads "chunk 15" = the zero pad + the first 16 B of the next row (finite halves; zero weights)
#define DG_YD 36     // Yd tile row stride (32 + 4):   4 * 9
#define DG_YV 68     // Yv tile row stride (64 + 4):   4 * 17
#define DG_U 4       // edges per batch of the forming loops
#define DG_SB 8      // elements per thread in flight in the staging loop of the segment prologue
#define DG_SBA 4     // ... and in the per-phase re-staging of the alternating mode (accumulators are live there)

#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
#define MFMA16H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, (a)), __builtin_bit_cast(h8, (b)), (c), 0, 0, 0)
#define RFL(v) __builtin_amdgcn_readfirstlane(v)

#ifdef JAMUN_STAMP
__device__ unsigned long long g_dgstamp[2][8];  // [role][prologue, P1 work, P1 wait, P2 work, P2 wait, epilogue]
#define DSTAMP(t) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); } while (0)
__device__ unsigned long long g_dgtrace[12][48][4];  // workgroup 7, SP mode: per wave and k-step: start, arrival at barrier A, release, end
#define DTRACE(slot, t) do { if (blockIdx.x == 7 && trc < 48) g_dgtrace[wave][trc][slot] = (t); } while (0)
#define DACC(slot, t1, t0) do { st_acc[slot] += (t1) - (t0); } while (0)
#else
#define DSTAMP(t) do { } while (0)
#define DTRACE(slot, t) do { } while (0)
#define DACC(slot, t1, t0) do { } while (0)
#endif

// v_pk_fma_f32 with the two source-1 selections LLVM does not emit (it materialises the pair with v_mov instead, 3 extra VALU
// instructions per edge in the forming loops): both result lanes read the HIGH dword of b / the lanes read b swapped.
__device__ __forceinline__ f32x2 pk_bhi(f32x2 a2, f32x2 b2, f32x2 c2) {  // (a.x b.y + c.x, a.y b.y + c.y)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(c2) : "v"(a2), "v"(b2));
  return c2;
}
__device__ __forceinline__ f32x2 pk_bswap(f32x2 a2, f32x2 b2, f32x2 c2) {  // (a.x b.y + c.x, a.y b.x + c.y)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(c2) : "v"(a2), "v"(b2));
  return c2;
}

__device__ __forceinline__ float4 lds_read_f4(int addr) {  // ds_read_b128 from an absolute LDS address
#if defined(__HIP_DEVICE_COMPILE__)
  const f32x4 v = *(const __attribute__((address_space(3))) f32x4*)(unsigned)addr;
  return make_float4(v.x, v.y, v.z, v.w);
#else
  (void)addr;
  return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}
__device__ __forceinline__ int lds_address(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
#else
  (void)p;
  return 0;
#endif
}

// f16x3 split helpers: (a, b) -> packed halves with round-to-nearest-even; residual of a value against one half of a packed pair
// (v_fma_mix_f32: fma with per-operand f16 / f32 selection; a - hi is exact: hi holds the leading 11 bits of a)
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }  // 2^e, -126 <= e <= 127

__host__ __device__ constexpr int dg_xs0(bool h) { return h ? DG_XS0H : DG_XS0; }
__host__ __device__ constexpr int dg_xbuf(bool h) { return 32 * dg_xs0(h) + 8; }                         // floats of one X tile (mode 3: a second one behind the A tiles)
__host__ __device__ constexpr int dg_abuf(bool h) { return dg_xbuf(h) + 32 * DG_YD + 3 * 32 * DG_YV; }  // floats of one set of A tiles (X | Yd | Yv)
// mode: 0 two-phase with resident source rows, 1 two passes (large molecules), 2 single phase (double-buffered A tiles),
// 3 single phase with a double-buffered X tile and a single Y tile (mid-size spans)
__host__ __device__ inline size_t dg_lds_floats(int rs, int pmax, int mode, bool h) {
  // xs | A tiles: X (+8: the dummy quarter group reads 4 floats past the last row) | Yd | Yv | tabA | tabJ | deg | f16x3: per-destination
  // scales (2 x 32) and the per-source-row magnitudes
  return (size_t)rs * (mode == 1 ? DG_XST_ALT : DG_XST_RES) + (size_t)(mode == 2 ? 2 : 1) * dg_abuf(h) + (mode == 3 ? dg_xbuf(h) : 0) +
         (size_t)32 * pmax * 4 + (size_t)32 * pmax + 32 + (h ? 64 + rs : 0);
}

// ALT = false: the source rows of a tile stay in LDS for the whole segment (spans up to ~80 rows).
// ALT = true (molecules above that): LDS holds HALF of every source row at a time and the segment makes TWO PASSES over its
// hidden units (the sum over k is linear; the accumulators stay in the matrix waves' registers across both):
//   pass A, vector blocks (x, T, y, z)[32] resident:  Y(k) formed | barrier | Y(k) x W while T(k+1) is staged | barrier
//   pass B, the 120 scalar channels resident (one re-staging per segment):  X(k+1) formed while X(k) x W runs, X double
//           buffered in the dead Y tiles, one barrier per k
// instead of re-staging both halves at the phase boundaries of every hidden unit (measured on the 166-atom batch: 1.47 ms per
// launch with per-k re-staging, 1.21 ms with two passes), and instead of cutting the sources into row blocks whose
// (destination chunk, block) tiles each pay the full contraction.  Y is single buffered in pass A (LDS is full): forming and
// contraction alternate there, which costs little because fp32 MFMA and VALU work of a SIMD do not overlap anyway.
// MODE 2 (SP, spans up to ~52 rows: 2AA-size molecules): ONE phase per hidden unit.  The A tiles are double buffered (93 KB), the
// matrix waves run X(k) W and Y(k) W back to back while the forming waves build X(k+1) and Y(k+1) in ONE pass over the edges
// (one table read, one row address and 9 packed + 1 scalar FMA per edge instead of two passes with their own reads), one
// barrier per k plus a short second one around the copy of T(k+2) into its slots.
// MODE 3 (SPH, spans up to ~73 rows: 33-atom molecules, two per tile): the single-phase schedule with what LDS allows — X double
// buffered (16 KB), Y single.  The matrix waves contract Y(k) FIRST; the forming waves run the merged edge pass of their first
// destination pair meanwhile, store its X rows, and meet the matrix waves at a mid-step barrier before any Y(k+1) row is
// written; then X(k) x W runs against the rest of the forming work.  Three barriers per k instead of two phases of unequal
// weight with two passes over the edges (measured on 33 atoms x 256: 25.2 k -> see DESIGN.md cycles per k-step).
// H (f16x3 contraction): the forming waves write every A tile as two f16 planes (hi = rne16(a), lo = rne16(a - hi); chunks of
// [8 hi | 8 lo] halves = one lane's fragments of 8 consecutive inputs) and the matrix waves run three v_mfma_f32_*_f16 per
// product (lo hi + hi lo + hi hi, fp32 accumulate) against weights split the same way on the host: the error of a product is
// ~3 x 2^-24 of its magnitude, the level of one fp32 rounding, at 3/16 of the matrix-pipe time of v_mfma_f32_32x32x2_f32 — and,
// unlike fp32 MFMAs (which run on the vector FMA lanes and exclude EVERY VALU instruction of the SIMD while they execute;
// profiles/microbench/mfma_overlap.hip), f16 MFMAs leave ~3/4 of the vector issue slots to the forming waves.  To stay inside
// the f16 range the coefficients of destination i carry a power of two 2^sA_i chosen from a rigorous bound of its A rows
// (in-degree x max |h~| (static, a.hmax2) x max |x| over its source rows (measured while the rows are staged)), the weights carry
// 2^sB, and the accumulators are scaled back by 2^-(sA_i + sB) at the segment end: all exact.
template <int MODE, bool H>
__global__ __launch_bounds__(DG_THREADS) void k_conv_dg(DgArgs a) {
  constexpr bool ALT = MODE == 1, SPH = MODE == 3, SP = MODE == 2 || SPH;  // (SP: the single-phase schedule, both variants)
  constexpr int XS0 = dg_xs0(H), DG_ABUF = dg_abuf(H), DG_XBUF = dg_xbuf(H);
  constexpr int AH = H ? 8 : 4;  // floats between the two K halves (32x32) / four K quarters (16x16) of a fragment row
  constexpr int DG_XST = ALT ? DG_XST_ALT : DG_XST_RES;
  constexpr int OFFY = ALT ? 0 : 480;    // byte offset of the vector block inside a row
  constexpr int TCOL = ALT ? 0 : 120;    // float column of the vector block
  extern __shared__ float4 lds4[];
  float* __restrict__ lds = reinterpret_cast<float*>(lds4);
  float* __restrict__ xs = lds;                                  // [RS][DG_XST]
  float* __restrict__ Xt = xs + (size_t)a.RS * DG_XST;           // [32][XS0] (+8)
  float* __restrict__ Yd = Xt + DG_XBUF;                         // [32][DG_YD]
  float* __restrict__ Yv = Yd + 32 * DG_YD;                      // [3][32][DG_YV]
  float* __restrict__ X1 = Xt + DG_ABUF;                        // second X tile (SPH) / second set of A tiles (MODE 2)
  float4* __restrict__ tabA = reinterpret_cast<float4*>(Xt + (SPH ? DG_ABUF + DG_XBUF : (SP ? 2 : 1) * DG_ABUF));  // [32][PMAX] (c, c vx, c vy, c vz)
  int* __restrict__ tabJ = reinterpret_cast<int*>(tabA + 32 * a.PMAX);          // [32][PMAX] byte offset of the source row in xs
  int* __restrict__ deg_lds = tabJ + 32 * a.PMAX;                               // [32]
  float* __restrict__ sc_lds = reinterpret_cast<float*>(deg_lds + 32);          // H: [32] 2^sA_i
  float* __restrict__ isc_lds = sc_lds + 32;                                    // H: [32] 2^-(sA_i + sB)
  unsigned* __restrict__ rowmax = reinterpret_cast<unsigned*>(isc_lds + 32);    // H: [RS] max |x| of each source row (float bits)
  // segment end: the output tile is staged in LDS (dead A tiles / source rows) and stored by ALL threads as coalesced float4 rows
  float* __restrict__ OM = Xt;               // [32][128]    scalar tiles 0..3
  float* __restrict__ OQ = OM + 32 * 128;    // [4][32][32]  the four K-quarter partials of scalar tile 4
  float* __restrict__ OP = OQ + 4 * 32 * 32; // [32][96]     vector planes from the matrix waves
  float* __restrict__ OL = xs;               // [32][96]     vector rows accumulated by the forming waves (T term)

  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = RFL(tid0 >> 6);
#if defined(DG_STUB) && DG_STUB == 1  // register-pressure experiments: one role compiled out
  const bool is_mat = false;
#elif defined(DG_STUB) && DG_STUB == 2
  const bool is_mat = true;
#else
  const bool is_mat = wave < 4;
#endif
  const int PMAX = a.PMAX;
#if defined(JAMUN_STAMP) || defined(JAMUN_DGDBG)  // timing experiments (JAMUN_DG_DBG) exist in the diagnostic builds only
  const bool dbg_noform = a.dbg & 1, dbg_nomfma = a.dbg & 2, dbg_noweights = a.dbg & 4, dbg_notouch = a.dbg & 16, dbg_noprio = a.dbg & 32, dbg_nostage = a.dbg & 8;
#else
  constexpr bool dbg_noform = false, dbg_nomfma = false, dbg_noweights = false, dbg_notouch = false, dbg_noprio = false, dbg_nostage = false;
#endif
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, ta = 0, tb = 0;
  int trc = 0;
  (void)trc;
  (void)st_acc; (void)ta; (void)tb;

  if constexpr (H) {
    for (int i = tid0; i < a.RS; i += DG_THREADS) rowmax[i] = 0u;
    LDS_BARRIER();
  }
  auto absmax4 = [](float4 v) { return __float_as_uint(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))); };
  (void)absmax4;
  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int4 sg0 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2];
    const int4 sg1 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2 + 1];
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    // Everything derived from the lane id is recomputed per segment: an opaque copy keeps the compiler from hoisting dozens of
    // lane-only address expressions out of this loop and spilling them (each reload then costs a full memory round trip).
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) { kk = kk < 0 ? 0 : (kk < nk ? kk : nk - 1); return kk < k_run ? k_begin + kk : k_extra; };
    const int2 t_at = a.tile_atoms[tile];
    const int n0 = RFL(t_at.x), n_dst = RFL(t_at.y);
    const int2 span = a.tile_span[tile];
    const int s_lo = RFL(span.x), rows = RFL(span.y) - s_lo;
    const int rows16 = rows;
    DSTAMP(ta);

    // ---- segment prologue: source rows -> LDS (x1 re-laid as [u][4] with the T slot second); zero the A tiles
    // (staging loops: DG_SB global loads are issued before the first LDS store — a load-store-load chain would serialise one
    // L2 round trip per element and the re-staging of the alternating mode sits on the critical path of every hidden unit)
    auto stage_x0 = [&]() {  // ALT: scalar channels of every source row (+ zero padding)
      const int total = rows16 * (DG_XST_ALT / 4);
      for (int base = tid; base < total; base += DG_SBA * DG_THREADS) {
        float4 v[DG_SBA];
#pragma unroll
        for (int q8 = 0; q8 < DG_SBA; ++q8) {
          const int idx = base + q8 * DG_THREADS;
          const int j = idx / (DG_XST_ALT / 4), q = idx - j * (DG_XST_ALT / 4);
          v[q8] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < total && j < rows && q < 30) {
            v[q8] = *reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + j) * a.XS + 4 * q);
            if constexpr (H) { const float4 g4 = reinterpret_cast<const float4*>(a.gx)[q]; v[q8] = make_float4(v[q8].x * g4.x, v[q8].y * g4.y, v[q8].z * g4.z, v[q8].w * g4.w); }
          }
        }
#pragma unroll
        for (int q8 = 0; q8 < DG_SBA; ++q8) {
          const int idx = base + q8 * DG_THREADS;
          const int j = idx / (DG_XST_ALT / 4), q = idx - j * (DG_XST_ALT / 4);
          if (idx < total) *reinterpret_cast<float4*>(xs + (size_t)j * DG_XST_ALT + 4 * q) = v[q8];
        }
      }
      if constexpr (H)  // row pads of the second X buffer (it aliases the Y tiles) are read as inputs 120..127: zero halves, not stale bits
        for (int i = tid; i < 33 * 4; i += DG_THREADS) Yd[i < 128 ? (i >> 2) * XS0 + 120 + (i & 3) : 32 * XS0 + (i & 3)] = 0.f;
    };
    auto stage_x1 = [&](int k) {  // ALT: vector block (x, T_k, y, z) of every source row (T from the pre-pass buffer)
      const int total = rows * 32;
      const float* __restrict__ tk = a.T + ((size_t)k * a.n_atoms + s_lo) * 32;
      for (int base = tid; base < total; base += DG_SBA * DG_THREADS) {
        float4 v[DG_SBA];
#pragma unroll
        for (int q8 = 0; q8 < DG_SBA; ++q8) {
          const int idx = min(base + q8 * DG_THREADS, total - 1);
          const float* __restrict__ p1 = a.x + (size_t)(s_lo + (idx >> 5)) * a.XS + 120 + 3 * (idx & 31);
          const float g1 = H ? a.gx[120 + 3 * (idx & 31)] : 1.f;  // (f16x3: the channel's power of two, DgArgs::gx; T stays in true units)
          v[q8] = make_float4(p1[0] * g1, tk[idx], p1[1] * g1, p1[2] * g1);
        }
#pragma unroll
        for (int q8 = 0; q8 < DG_SBA; ++q8) {
          const int idx = base + q8 * DG_THREADS;
          if (idx < total) *reinterpret_cast<float4*>(xs + (size_t)(idx >> 5) * DG_XST_ALT + 4 * (idx & 31)) = v[q8];
        }
      }
    };
    auto stage_t = [&](int k) {  // ALT, forming waves only (512 threads): T_k of every source row into the T slots
      const int total = rows * 32, ftid = tid - 4 * 64;
      const float* __restrict__ tk = a.T + ((size_t)k * a.n_atoms + s_lo) * 32;
      constexpr int NB = 11;  // 512 x 11 elements = 176 rows in ONE round trip (the forming waves hold nothing else here)
      for (int base = ftid; base < total; base += NB * 512) {
        float v[NB];
#pragma unroll
        for (int q8 = 0; q8 < NB; ++q8) v[q8] = tk[min(base + q8 * 512, total - 1)];
#pragma unroll
        for (int q8 = 0; q8 < NB; ++q8) {
          const int idx = base + q8 * 512;
          if (idx < total) xs[(size_t)(idx >> 5) * DG_XST_ALT + 4 * (idx & 31) + 1] = v[q8];
        }
      }
    };
    if constexpr (ALT) {
      if constexpr (H) {  // magnitudes of the source rows (both halves of a row share the destination's scale)
        const int total = rows * 54;  // 216 floats = 54 x 16 bytes per row
        for (int idx = tid; idx < total; idx += DG_THREADS) {
          const int j = idx / 54, q = idx - j * 54;
          const float4 xv_ = *reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + j) * a.XS + 4 * q), g4 = reinterpret_cast<const float4*>(a.gx)[q];
          atomicMax(&rowmax[j], absmax4(make_float4(xv_.x * g4.x, xv_.y * g4.y, xv_.z * g4.z, xv_.w * g4.w)));
        }
      }
      stage_x1(k_of(0));
    } else {
      const int total = rows16 * (DG_XST / 4);
      for (int base = tid; base < total; base += DG_SB * DG_THREADS) {
        float4 v[DG_SB];
#pragma unroll
        for (int q8 = 0; q8 < DG_SB; ++q8) {
          const int idx = base + q8 * DG_THREADS;
          const int j = idx / (DG_XST / 4), q = idx - j * (DG_XST / 4);
          v[q8] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (idx < total && j < rows) {
            const float* __restrict__ xr = a.x + (size_t)(s_lo + j) * a.XS;
            if (q < 30) {  // x0 (120 = 30 x 4)
              v[q8] = *reinterpret_cast<const float4*>(xr + 4 * q);
              if constexpr (H) { const float4 g4 = reinterpret_cast<const float4*>(a.gx)[q]; v[q8] = make_float4(v[q8].x * g4.x, v[q8].y * g4.y, v[q8].z * g4.z, v[q8].w * g4.w); }
            } else if (q < 62) {  // x1[u] -> (x, T = 0, y, z)
              const float* p = xr + 120 + 3 * (q - 30);
              const float g1 = H ? a.gx[120 + 3 * (q - 30)] : 1.f;
              v[q8] = make_float4(p[0] * g1, 0.f, p[1] * g1, p[2] * g1);
            }
          }
        }
#pragma unroll
        for (int q8 = 0; q8 < DG_SB; ++q8) {
          const int idx = base + q8 * DG_THREADS;
          const int j = idx / (DG_XST / 4), q = idx - j * (DG_XST / 4);
          if (idx < total) *reinterpret_cast<float4*>(xs + (size_t)j * DG_XST + 4 * q) = v[q8];
          if constexpr (H) {
            if (idx < total && j < rows && q < 62) atomicMax(&rowmax[j], absmax4(v[q8]));
          }
        }
      }
    }
    for (int idx = tid; idx < (SPH ? DG_ABUF + DG_XBUF : (SP ? 2 : 1) * DG_ABUF); idx += DG_THREADS) Xt[idx] = 0.f;
    if (tid < 32) deg_lds[tid] = (tid < n_dst) ? a.deg[n0 + tid] : 0;
    LDS_BARRIER();  // (LDS-only: the previous segment's slab stores keep draining in the background)

    if (is_mat) {
      // =========================================== MATRIX waves ===========================================
      const int w = wave;
#if defined(JAMUN_STAMP) || defined(JAMUN_DGDBG)
      if (a.dbg & 1024) __builtin_amdgcn_s_setprio(3);
      if (a.dbg & 2048) __builtin_amdgcn_s_setprio(2);
#endif
      const int r = lane & 31, hh = lane >> 5;      // 32x32x2 fragments: row / column r, K half hh
      const int r16 = lane & 15, kq = lane >> 4;    // 16x16x4 fragments: row / column r16, K quarter kq
      const int rh = w & 1, ch = w >> 1;            // this wave's 16x16 sub-tiles of the vector planes
      f32x16 accM, acc4;
      f32x4 accP[3];
#pragma unroll
      for (int q = 0; q < 16; ++q) { accM[q] = 0.f; acc4[q] = 0.f; }
#pragma unroll
      for (int m = 0; m < 3; ++m) accP[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      // fragment rows of this lane.  fp32: 4 consecutive K steps per 16-byte read; H: chunk (2 g + hh) / (4 G + kq) of 8 inputs,
      // hi halves at +0, lo halves at +4 floats
      const float* Xa = Xt + r * XS0 + AH * hh;  // (SP: re-pointed to the A buffer of the hidden unit at hand)
      const float* Da = Yd + r * DG_YD + AH * hh;
      const float* Va = Yv + (16 * rh + r16) * DG_YV + AH * kq;

      // weight blocks (64 lanes x float4) of one hidden unit, in the order this wave consumes them:
      //   P1: WX[w][g], g = 0..14, with WX[4][4 s + w] after g = 3, 7, 11, 14          (19 blocks: chunks 5 5 5 4)
      //   P2: WD[w][0..3], WD[4][w] | WV[ch][0..3]                                       (5 | 4)
      // (uniform block base in scalar registers + the lane as a 32-bit offset: one address register per load, not two)
      auto WX = [&](int k, int t, int g) { return (a.wx + ((size_t)k * 5 * 16 + t * 16 + g) * 64)[lane]; };
      auto WD = [&](int k, int t, int g) { return (a.wd + ((size_t)k * 5 * 4 + t * 4 + g) * 64)[lane]; };
      auto WV = [&](int k, int c, int g) { return (a.wv + ((size_t)k * 2 * 4 + c * 4 + g) * 64)[lane]; };

      // H: blocks of 8 halves per lane; chunk c of the scalar inputs = groups 2c, 2c+1 of 16 inputs (own tile: hi, lo, hi, lo) + this
      // wave's share of scalar tile 4 — groups w and w + 4, i.e. one group in the chunks with (c & 1) == (w >> 1)
      auto WH = [&](int k, int n) { return (a.wh + (((size_t)k * 4 + w) * 34 + n) * 64)[lane]; };  // block n of this wave's stream of hidden unit k
      auto LDF = [](const float* p_) { return *reinterpret_cast<const float4*>(p_); };
      float4 B0[6], B1[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) B0[i] = B1[i] = make_float4(1.f, 1.f, 1.f, 1.f);  // (defined values for the JAMUN_DG_DBG=4 experiment)
      auto load_x = [&](float4 (&B)[6], int k, int c) {  // chunk c of P1: groups 4c..4c+3 (+ the quarter group 4c + w)
        if (dbg_noweights) return;
        if constexpr (H) {
#pragma unroll
          for (int i = 0; i < 4; ++i) B[i] = WH(k, 6 * c + i);
          if ((w >> 1) == (c & 1)) { B[4] = WH(k, 6 * c + 4); B[5] = WH(k, 6 * c + 5); }
          return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) B[i] = WX(k, w, (4 * c + i) < 15 ? 4 * c + i : 15);
        B[4] = WX(k, 4, 4 * c + w);  // group 15 (wave 3, chunk 3) is a zero block
      };
      // (A fragments are requested two groups ahead of the MFMAs that consume them and the scheduler is told to keep it that
      // way: left alone, the compiler reads each fragment into the register of the previous one right before its use, and
      // the matrix pipe drains at every group of four MFMAs while the LDS read returns — about a quarter of the matrix
      // waves' time.  Two fragments in flight is what the register budget of 168 allows without spilling weight pointers.)
#define DG_SCHB() __builtin_amdgcn_sched_barrier(0)
#define DG_M4(ACC, A, Bv)                                                                                                    \
  ACC = MFMA32(A.x, Bv.x, ACC);                                                                                              \
  ACC = MFMA32(A.y, Bv.y, ACC);                                                                                              \
  ACC = MFMA32(A.z, Bv.z, ACC);                                                                                              \
  ACC = MFMA32(A.w, Bv.w, ACC)
#define DG_M3(ACC, AH_, AL_, BH_, BL_)                                                                                         \
  ACC = MFMA32H(AL_, BH_, ACC);                                                                                              \
  ACC = MFMA32H(AH_, BL_, ACC);                                                                                              \
  ACC = MFMA32H(AH_, BH_, ACC)
      auto run_x = [&](const float4 (&B)[6], int c) {
        if (dbg_nomfma) return;
        if constexpr (H) {
          const float4 h0 = LDF(Xa + 32 * c), l0 = LDF(Xa + 32 * c + 4), h1 = LDF(Xa + 32 * c + 16), l1 = LDF(Xa + 32 * c + 20);
          DG_SCHB();
          DG_M3(accM, h0, l0, B[0], B[1]);
          DG_M3(accM, h1, l1, B[2], B[3]);
          if ((w >> 1) == (c & 1)) {  // this wave's group of scalar tile 4: the (w & 1)-th group of the chunk
            if (w & 1) { DG_M3(acc4, h1, l1, B[4], B[5]); }
            else { DG_M3(acc4, h0, l0, B[4], B[5]); }
          }
          return;
        }
        const bool last = (c == 3);  // chunk 3 holds groups 12..14 only
        float4 a0 = *reinterpret_cast<const float4*>(Xa + 8 * (4 * c));
        float4 a1 = *reinterpret_cast<const float4*>(Xa + 8 * (4 * c + 1));
        DG_SCHB();
        DG_M4(accM, a0, B[0]);
        a0 = *reinterpret_cast<const float4*>(Xa + 8 * (4 * c + 2));
        DG_SCHB();
        DG_M4(accM, a1, B[1]);
        a1 = *reinterpret_cast<const float4*>(Xa + 8 * (last ? 4 * c + w : 4 * c + 3));
        DG_SCHB();
        DG_M4(accM, a0, B[2]);
        if (!last) {
          a0 = *reinterpret_cast<const float4*>(Xa + 8 * (4 * c + w));  // this wave's quarter group of scalar tile 4
          DG_SCHB();
          DG_M4(accM, a1, B[3]);
          DG_M4(acc4, a0, B[4]);
        } else {
          DG_M4(acc4, a1, B[4]);
        }
      };
      auto load_d = [&](float4 (&B)[6], int k) {
        if (dbg_noweights) return;
        if constexpr (H) {
#pragma unroll
          for (int i = 0; i < 4; ++i) B[i] = WH(k, 24 + i);
          if (w < 2) { B[4] = WH(k, 28); B[5] = WH(k, 29); }  // scalar tile 4: group w
          return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) B[i] = WD(k, w, i);
        B[4] = WD(k, 4, w);
      };
      auto run_d = [&](const float4 (&B)[6]) {
        if (dbg_nomfma) return;
        if constexpr (H) {
          // (the dot-input tile is fp32: 8 consecutive inputs of a row = two 16-byte reads, split here into hi / lo halves)
          auto split8 = [&](float4 f0, float4 f1, float4& hi, float4& lo) {
            const unsigned p0 = cvt_pk_f16(f0.x, f0.y), p1 = cvt_pk_f16(f0.z, f0.w), p2 = cvt_pk_f16(f1.x, f1.y), p3 = cvt_pk_f16(f1.z, f1.w);
            const unsigned q0 = cvt_pk_f16(resid_lo(f0.x, p0), resid_hi(f0.y, p0)), q1 = cvt_pk_f16(resid_lo(f0.z, p1), resid_hi(f0.w, p1)),
                           q2 = cvt_pk_f16(resid_lo(f1.x, p2), resid_hi(f1.y, p2)), q3 = cvt_pk_f16(resid_lo(f1.z, p3), resid_hi(f1.w, p3));
            hi = make_float4(__uint_as_float(p0), __uint_as_float(p1), __uint_as_float(p2), __uint_as_float(p3));
            lo = make_float4(__uint_as_float(q0), __uint_as_float(q1), __uint_as_float(q2), __uint_as_float(q3));
          };
          const float4 f00 = LDF(Da), f01 = LDF(Da + 4), f10 = LDF(Da + 16), f11 = LDF(Da + 20);
          float4 h0, l0, h1, l1;
          split8(f00, f01, h0, l0);
          split8(f10, f11, h1, l1);
          DG_SCHB();
          DG_M3(accM, h0, l0, B[0], B[1]);
          DG_M3(accM, h1, l1, B[2], B[3]);
          if (w == 0) { DG_M3(acc4, h0, l0, B[4], B[5]); }
          else if (w == 1) { DG_M3(acc4, h1, l1, B[4], B[5]); }
          return;
        }
        float4 a0 = *reinterpret_cast<const float4*>(Da);
        float4 a1 = *reinterpret_cast<const float4*>(Da + 8);
        DG_SCHB();
        DG_M4(accM, a0, B[0]);
        a0 = *reinterpret_cast<const float4*>(Da + 16);
        DG_SCHB();
        DG_M4(accM, a1, B[1]);
        a1 = *reinterpret_cast<const float4*>(Da + 24);
        DG_SCHB();
        DG_M4(accM, a0, B[2]);
        a0 = *reinterpret_cast<const float4*>(Da + 8 * w);  // this wave's K quarter of scalar tile 4
        DG_SCHB();
        DG_M4(accM, a1, B[3]);
        DG_M4(acc4, a0, B[4]);
      };
      auto load_v = [&](float4 (&B)[6], int k) {
        if (dbg_noweights) return;
        if constexpr (H) {
#pragma unroll
          for (int i = 0; i < 4; ++i) B[i] = WH(k, 30 + i);
          return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) B[i] = WV(k, ch, i);
      };
      auto run_v = [&](const float4 (&B)[6]) {
        if (dbg_nomfma) return;
        if constexpr (H) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {  // (the three planes are independent accumulators: their MFMAs alternate)
            float4 fh[3], fl[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
              fh[m] = LDF(Va + m * 32 * DG_YV + 32 * g);
              fl[m] = LDF(Va + m * 32 * DG_YV + 32 * g + 4);
            }
            DG_SCHB();
#pragma unroll
            for (int m = 0; m < 3; ++m) accP[m] = MFMA16H(fl[m], B[2 * g], accP[m]);
#pragma unroll
            for (int m = 0; m < 3; ++m) accP[m] = MFMA16H(fh[m], B[2 * g + 1], accP[m]);
#pragma unroll
            for (int m = 0; m < 3; ++m) accP[m] = MFMA16H(fh[m], B[2 * g], accP[m]);
            DG_SCHB();
          }
          return;
        }
        float4 av[2][3];
#pragma unroll
        for (int m = 0; m < 3; ++m) av[0][m] = *reinterpret_cast<const float4*>(Va + m * 32 * DG_YV);
        DG_SCHB();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < 3) {
#pragma unroll
            for (int m = 0; m < 3; ++m) av[(i + 1) & 1][m] = *reinterpret_cast<const float4*>(Va + m * 32 * DG_YV + 16 * (i + 1));
            DG_SCHB();
          }
          float4(&A)[3] = av[i & 1];
#pragma unroll
          for (int m = 0; m < 3; ++m) accP[m] = MFMA16(A[m].x, B[i].x, accP[m]);
#pragma unroll
          for (int m = 0; m < 3; ++m) accP[m] = MFMA16(A[m].y, B[i].y, accP[m]);
#pragma unroll
          for (int m = 0; m < 3; ++m) accP[m] = MFMA16(A[m].z, B[i].z, accP[m]);
#pragma unroll
          for (int m = 0; m < 3; ++m) accP[m] = MFMA16(A[m].w, B[i].w, accP[m]);
          if (i < 3) DG_SCHB();
        }
      };
      // T_k[j][w'] = sum_u x0_j[u] W[(k,u)][w'] comes from the pre-pass k_tprod (once per source atom and layer, not once
      // per tile span).  Resident mode: the matrix waves copy the rows of the span for hidden unit k+1 into the T slots of the
      // source rows during P2(k) (loads issued at the start of the phase, stores at its end).  Alternating mode: T is staged
      // together with the vector block by all threads (stage_x1).
      constexpr int TLD = ALT ? 1 : 7;  // T elements per matrix-wave thread held in registers: 256 threads x 7 = 56 rows x 32
      float tpre[TLD];
      auto load_tslots = [&](int k) {
        if constexpr (!ALT) {
          const float* __restrict__ tk = a.T + ((size_t)k * a.n_atoms + s_lo) * 32;
#pragma unroll
          for (int q = 0; q < TLD; ++q) {
            const int idx = min(w * 64 + lane + 256 * q, rows * 32 - 1);
            tpre[q] = tk[idx];
          }
        }
      };
      auto store_tslots = [&](int k) {
        if constexpr (!ALT) {
#pragma unroll
          for (int q = 0; q < TLD; ++q) {
            const int idx = w * 64 + lane + 256 * q;
            if (idx < rows * 32) xs[(size_t)(idx >> 5) * DG_XST + TCOL + 4 * (idx & 31) + 1] = tpre[q];
          }
          // spans above 56 rows: the rest in a plain loop (two 29..40-atom molecules per tile)
          const float* __restrict__ tk = a.T + ((size_t)k * a.n_atoms + s_lo) * 32;
          for (int idx = w * 64 + lane + 256 * TLD; idx < rows * 32; idx += 256) xs[(size_t)(idx >> 5) * DG_XST + TCOL + 4 * (idx & 31) + 1] = tk[idx];
        }
      };

      if (a.row_blocks && !__syncthreads_or(0)) {  // a (destination chunk x source row block) tile without any edge: zero slab
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
          if (row < n_dst) {
            a.partial0[((size_t)slab * a.n_pad + n0 + row) * (a.nt0 * 32) + w * 32 + r] = 0.f;
            if (w == 0) a.partial0[((size_t)slab * a.n_pad + n0 + row) * (a.nt0 * 32) + 4 * 32 + r] = 0.f;
            if (w < 3) a.partial1[(((size_t)slab * a.n_pad + n0 + row) * 3 + w) * 32 + r] = 0.f;
          }
        }
        if constexpr (H)
          for (int i = tid; i < a.RS; i += DG_THREADS) rowmax[i] = 0u;
        __syncthreads();
        continue;
      }
      if constexpr (SPH) {
        // ---- single phase per hidden unit, Y single buffered: Y(k) x W first, mid-step barrier, then X(k) x W
        {
          const int k0 = k_of(0);
          load_tslots(k0);
          load_d(B0, k0);
          load_v(B1, k0);
          store_tslots(k0);
        }
        LDS_BARRIER();                    // T(k0) in its slots: the forming waves build A(k0) (X into buffer 0)
        if (nk > 1) load_tslots(k_of(1));
        LDS_BARRIER();                    // A(k0) complete
        if (nk > 1) store_tslots(k_of(1));
        LDS_BARRIER();                    // T(k1) in its slots
        DSTAMP(tb); DACC(0, tb, ta);
        for (int it = 0; it < nk; ++it) {
          const int k = k_of(it), kn = k_of(it + 1);
          DSTAMP(ta);
          Xa = ((it & 1) ? X1 : Xt) + r * XS0 + AH * hh;
          if (it + 2 < nk) load_tslots(k_of(it + 2));
          run_d(B0); load_x(B0, k, 0);
          run_v(B1); load_x(B1, k, 1);
          DTRACE(0, ta);
          DSTAMP(tb); DACC(3, tb, ta);
          DTRACE(1, tb);
          LDS_BARRIER();                  // Y(k) consumed: the forming waves may write Y(k+1)
          DSTAMP(ta); DACC(4, ta, tb);
          run_x(B0, 0); load_x(B0, k, 2);
          run_x(B1, 1); load_x(B1, k, 3);
          run_x(B0, 2); load_d(B0, kn);
          run_x(B1, 3); load_v(B1, kn);
          DSTAMP(tb); DACC(1, tb, ta);
          DTRACE(2, tb);
          LDS_BARRIER();                  // A(k+1) complete; nobody reads X(k) or T(k+1) any more
          DSTAMP(ta); DACC(2, ta, tb);
          if (it + 2 < nk) store_tslots(k_of(it + 2));
          LDS_BARRIER();
          DSTAMP(tb);
          DTRACE(3, tb);
          ++trc;
        }
      } else if constexpr (SP) {
        // ---- single phase per hidden unit
        {
          const int k0 = k_of(0);
          load_tslots(k0);
          load_x(B0, k0, 0);
          load_x(B1, k0, 1);
          store_tslots(k0);
        }
        LDS_BARRIER();                    // T(k0) in its slots: the forming waves build A(k0) into buffer 0
        if (nk > 1) load_tslots(k_of(1));
        LDS_BARRIER();                    // A(k0) complete
        if (nk > 1) store_tslots(k_of(1));
        LDS_BARRIER();                    // T(k1) in its slots
        DSTAMP(tb); DACC(0, tb, ta);
        for (int it = 0; it < nk; ++it) {
          const int k = k_of(it), kn = k_of(it + 1);
          DSTAMP(ta);
          const int bo = (it & 1) * DG_ABUF;
          Xa = Xt + bo + r * XS0 + AH * hh;
          Da = Yd + bo + r * DG_YD + AH * hh;
          Va = Yv + bo + (16 * rh + r16) * DG_YV + AH * kq;
          if (it + 2 < nk) load_tslots(k_of(it + 2));
          run_x(B0, 0); load_x(B0, k, 2);
          run_x(B1, 1); load_x(B1, k, 3);
          run_x(B0, 2); load_d(B0, k);
          run_x(B1, 3); load_v(B1, k);
          run_d(B0); load_x(B0, kn, 0);
          run_v(B1); load_x(B1, kn, 1);
          DTRACE(0, ta);
          DSTAMP(tb); DACC(1, tb, ta);
          DTRACE(1, tb);
          LDS_BARRIER();                  // A(k+1) complete; nobody reads A(k) or T(k+1) any more
#ifdef JAMUN_DUMP
          if (a.dump && blockIdx.x < 32 && sgi == 0 && it == a.dbg) {
            const float* src = Xt + ((it + 1) & 1) * DG_ABUF;
            float* dst = a.dump + (size_t)blockIdx.x * 12288;
            for (int idx = tid; idx < DG_ABUF; idx += 256) dst[idx] = src[idx];
            if (tid < 32) dst[DG_ABUF + tid] = H ? sc_lds[tid] : 1.f;
            if (tid == 0) { dst[DG_ABUF + 32] = (float)k_of(it + 1); dst[DG_ABUF + 33] = (float)n0; dst[DG_ABUF + 34] = (float)n_dst; dst[DG_ABUF + 35] = (float)nk; }
          }
#endif
          DSTAMP(ta); DACC(2, ta, tb);
          DTRACE(2, ta);
          if (it + 2 < nk) store_tslots(k_of(it + 2));
          DSTAMP(tb); DACC(3, tb, ta);
          LDS_BARRIER();
          DSTAMP(ta); DACC(4, ta, tb);
          DTRACE(3, ta);
          ++trc;
        }
      } else if constexpr (ALT) {
        // ---- two passes over the hidden units of the segment (the sum over k is linear, the accumulators stay in registers):
        // pass A with the VECTOR blocks resident:  Y(k) formed | barrier | Y(k) x W, T(k+1) staged by the forming waves | barrier
        load_d(B0, k_of(0));
        load_v(B1, k_of(0));
        DSTAMP(tb); DACC(0, tb, ta);
        float t_sink = 0.f;
        for (int it = 0; it < nk; ++it) {
          const int kn = k_of(it + 1);
          DSTAMP(ta);
          if (it + 1 < nk && !dbg_notouch) {
            // The matrix waves idle while Y(k) is formed: they pull T(k+1) of the span (HBM / Infinity Cache: the T buffer of
            // a large batch is tens of MB) into L2, where the forming waves' staging loads find it after the barrier.
            const float* __restrict__ tk = a.T + ((size_t)kn * a.n_atoms + s_lo) * 32;
            for (int idx = 32 * (w * 64 + lane); idx < rows * 32; idx += 32 * 256) t_sink += tk[idx];  // one load per 128-byte line
          }
          LDS_BARRIER();                  // Y(k) complete
          DSTAMP(tb); DACC(2, tb, ta);
          run_d(B0);
          run_v(B1);
          if (it + 1 < nk) { load_d(B0, kn); load_v(B1, kn); }
          DSTAMP(ta); DACC(3, ta, tb);
          LDS_BARRIER();                  // Y(k) consumed; T(k+1) in its slots
          DSTAMP(tb); DACC(4, tb, ta);
        }
        if (t_sink == 1.2345e30f) accM[0] += t_sink;  // (keeps the touch loads alive)
        // pass B with the SCALAR channels resident (one re-staging per segment): X double buffered (second buffer = the dead Y
        // tiles), X(k+1) formed while X(k) x W runs, one barrier per hidden unit
        DSTAMP(ta);
        stage_x0();
        load_x(B0, k_of(0), 0);
        load_x(B1, k_of(0), 1);
        LDS_BARRIER();                    // scalar channels resident
        LDS_BARRIER();                    // X(k0) complete
        DSTAMP(tb); DACC(0, tb, ta);
        for (int it = 0; it < nk; ++it) {
          const int k = k_of(it), kn = k_of(it + 1);
          DSTAMP(ta);
          Xa = ((it & 1) ? Yd : Xt) + r * XS0 + AH * hh;
          run_x(B0, 0); load_x(B0, k, 2);
          run_x(B1, 1); load_x(B1, k, 3);
          run_x(B0, 2); load_x(B0, kn, 0);
          run_x(B1, 3); load_x(B1, kn, 1);
          DSTAMP(tb); DACC(1, tb, ta);
          LDS_BARRIER();                  // X(k+1) complete; X(k) consumed
          DSTAMP(ta); DACC(2, ta, tb);
        }
      } else {
      // "P2(-1)": T slots of the first hidden unit (no Y yet); first weight chunks
        {
          const int k0 = k_of(0);
          load_tslots(k0);
          load_x(B0, k0, 0);
          load_x(B1, k0, 1);
          store_tslots(k0);
        }
        LDS_BARRIER();
        DSTAMP(tb); DACC(0, tb, ta);
        for (int it = 0; it < nk; ++it) {
          const int k = k_of(it), kn = k_of(it + 1);
          DSTAMP(ta);
          const bool has_next = it + 1 < nk;
          // ---- P1: X(k) x W  (B0 = chunk 0, B1 = chunk 1 already in flight)
          run_x(B0, 0); load_x(B0, k, 2);
          run_x(B1, 1); load_x(B1, k, 3);
          run_x(B0, 2); load_d(B0, k);
          run_x(B1, 3); load_v(B1, k);
          DSTAMP(tb); DACC(1, tb, ta);
          LDS_BARRIER();
          DSTAMP(ta); DACC(2, ta, tb);
          // ---- P2: Y(k) x W; T slots of k+1 (nobody reads T during P2)
          if (has_next) load_tslots(kn);
          run_d(B0); load_x(B0, kn, 0);
          run_v(B1); load_x(B1, kn, 1);
          if (has_next) store_tslots(kn);
          DSTAMP(tb); DACC(3, tb, ta);
          LDS_BARRIER();
          DSTAMP(ta); DACC(4, ta, tb);
        }
      }
      DSTAMP(ta);
      if constexpr (H) {  // back to true scale: row i carries 2^(sA_i + sB)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float sI = isc_lds[(q & 3) + 8 * (q >> 2) + 4 * hh];
          accM[q] *= sI;
          acc4[q] *= sI;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float sI = isc_lds[16 * rh + 4 * kq + q];
#pragma unroll
          for (int m = 0; m < 3; ++m) accP[m][q] *= sI;
        }
      }
      // ---- segment end: accumulators -> LDS staging tile (the A tiles are dead after the last barrier)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        OM[row * 128 + w * 32 + r] = accM[q];
        OQ[(w * 32 + row) * 32 + r] = acc4[q];
      }
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) OP[(16 * rh + 4 * kq + q) * 96 + m * 32 + 16 * ch + r16] = accP[m][q];
      DSTAMP(tb); DACC(5, tb, ta);
    } else {
      // =========================================== FORMING waves ===========================================
      // Forming waves run at raised priority: left to the default oldest-first arbitration the matrix wave of a SIMD issues its
      // whole MFMA chain first (fp32 MFMA and VALU of a SIMD exclude each other) and the two forming waves then finish one
      // after the other, bound by LDS latency; with priority their VALU instructions go first and the MFMAs fill their
      // latency gaps (measured timeline: matrix waves done at 10.2k cycles of a 15.1k step, forming waves at 11.7k .. 13.9k).
      if (!dbg_noprio) __builtin_amdgcn_s_setprio(3);
#if defined(JAMUN_STAMP) || defined(JAMUN_DGDBG)
      if (a.dbg & 1024) __builtin_amdgcn_s_setprio(1);
      if (a.dbg & 2048) __builtin_amdgcn_s_setprio(2);
#endif
      const int fw = wave - 4;                    // forming wave index
      const int h = lane >> 5, u = lane & 31;
      const int xs_lds = lds_address(xs);  // LDS address of the source rows
      // per destination pair dp: this lane's edge slots t = u (page 0) and u + 32 (page 1) of destination i = 4 fw + 2 dp + h
      float evx[DG_NDP][2], evy[DG_NDP][2], evz[DG_NDP][2];
      float scA[DG_NDP], iscA[DG_NDP];  // H: 2^sA_i of this lane's destination and its inverse
      int hidx[DG_NDP][2];
      int hidx2[DG_NDP];  // page 0 only: slot of the bonded edge merged into this lane's radial edge (same source), or -1
      int P[DG_NDP];
      const bool two_pages = a.S > 32;
#pragma unroll
      for (int dp = 0; dp < DG_NDP; ++dp) {
        const int i = 2 * DG_NDP * fw + 2 * dp + h;
        const int dg = deg_lds[i];
        hidx2[dp] = -1;
        int sj0 = 0;
        float xm = 0.f;  // H: largest |x| over the source rows of this lane's edges
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
          evx[dp][pg] = evy[dp][pg] = evz[dp][pg] = 0.f;
          hidx[dp][pg] = -1;
          if (pg == 1 && !two_pages) continue;
          const int t = u + 32 * pg;
          int jofs = 0;
          if (t < dg) {
            const int slot = (n0 + i) * a.S + t;
            const int sj = a.esrc[slot];
            if (pg == 0) sj0 = sj;
            const int jl = (sj & 0x7fffffff) - s_lo;
            if (jl >= 0 && jl < rows) {  // (a source outside the span belongs to another row-block tile of these destinations)
              const float4 ge = a.egeo[slot];
              evx[dp][pg] = ge.x; evy[dp][pg] = ge.y; evz[dp][pg] = ge.z;
              hidx[dp][pg] = slot;
              jofs = jl * DG_XST * 4;  // byte offset of the source row inside xs
              if constexpr (H) xm = fmaxf(xm, __uint_as_float(rowmax[jl]));
            }
          }
          if (t < PMAX) tabJ[i * PMAX + t] = jofs + xs_lds;
        }
        scA[dp] = iscA[dp] = 1.f;
        if constexpr (H) {
          // |A rows of destination i| <= sum_e |c_e| max |zeta_e| <= deg_i * 2 max|h~| * max|x| (dot and cross products of a unit
          // vector with x1: <= sqrt(3) max|x|): scaled below 2^14, a factor 4 inside the f16 range
#pragma unroll
          for (int off = 16; off >= 1; off >>= 1) xm = fmaxf(xm, __shfl_xor(xm, off, 32));
          const float bound = (float)dg * a.hmax2 * xm;
          int sA = 0;
          if (bound > 0.f) sA = 14 - ((int)((__float_as_uint(bound) >> 23) & 0xffu) - 126);  // bound < 2^(exponent field - 126)
          sA = max(-60, min(60, sA));
          scA[dp] = pow2f(sA);
          iscA[dp] = pow2f(-sA);
          if (u == 0) { sc_lds[i] = scA[dp]; isc_lds[i] = pow2f(-sA - a.sB); }
        }
        if (!two_pages) {
          // A bonded pair inside the cutoff appears twice in the edge table (radial edge + bonded edge, same source and unit
          // vector; src/jamun/model/denoiser.py:152): the bonded edge is folded into its radial twin (coefficients add), which
          // shortens the edge loops — for a 17-atom molecule from 17 slots (5 batches of DG_U) to 16 (4 batches).  The bonded
          // in-edges are the LAST slots of a destination (k_geom), so their lanes are dg - nb .. dg - 1 of the half.
          const bool is_b = u < dg && sj0 < 0;  // bit 31 = bonded
          const unsigned long long bal = __ballot(is_b);
          const int nb = __popc((unsigned)(bal >> (32 * h)));
          const int nb_max = RFL(max(__popc((unsigned)bal), __popc((unsigned)(bal >> 32))));
          const int jraw = sj0 & 0x7fffffff;
          bool merged = false;
          for (int b = 0; b < nb_max; ++b) {
            const int lb = dg - nb + b;  // lane (slot) of this half's b-th bonded edge
            const int jb = __shfl(jraw, lb & 31, 32);
            const bool match = b < nb && u < dg && !is_b && jraw == jb && hidx[dp][0] >= 0 && hidx2[dp] < 0;  // (a second bond of the same pair stays its own edge)
            if (match) hidx2[dp] = (n0 + i) * a.S + lb;
            const unsigned long long mb = __ballot(match);
            if (is_b && u == lb && (unsigned)(mb >> (32 * h)) != 0u) merged = true;
          }
          if (merged) hidx[dp][0] = -1;  // (its table entry becomes zero; its geometry is the twin's)
          const unsigned long long act = __ballot(hidx[dp][0] >= 0);
          const int p_lo = 32 - __clz((unsigned)act), p_hi = 32 - __clz((unsigned)(act >> 32));  // __clz(0) = 32
          P[dp] = RFL(max(p_lo, p_hi));
        } else {
          P[dp] = RFL(max(deg_lds[2 * DG_NDP * fw + 2 * dp], deg_lds[2 * DG_NDP * fw + 2 * dp + 1]));
        }
      }
      // A tile of a large molecule (destination chunk x source row block) may have no edge at all: checked below by all waves
      int any_edge = 0;
#pragma unroll
      for (int dp = 0; dp < DG_NDP; ++dp)
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) any_edge |= hidx[dp][pg] >= 0;
      if (a.row_blocks && !__syncthreads_or(any_edge)) {  // (only batches with row-block tiles pay for the vote)
        if constexpr (H)
          for (int i = tid; i < a.RS; i += DG_THREADS) rowmax[i] = 0u;
        __syncthreads();
        continue;
      }

      float O[DG_NDP][3];  // vector rows from scalar inputs, accumulated over the hidden units of the segment: lane = (h, w' = u)
#pragma unroll
      for (int dp = 0; dp < DG_NDP; ++dp) O[dp][0] = O[dp][1] = O[dp][2] = 0.f;
      float hv[DG_NDP][2];  // h~ of this lane's edges for the NEXT hidden unit (prefetched)
      float hv2[DG_NDP];
      auto load_h = [&](int k) {
        const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
        for (int dp = 0; dp < DG_NDP; ++dp) {
#pragma unroll
          for (int pg = 0; pg < 2; ++pg) hv[dp][pg] = hk[hidx[dp][pg] >= 0 ? hidx[dp][pg] : n0 * a.S];
          hv2[dp] = hk[hidx2[dp] >= 0 ? hidx2[dp] : n0 * a.S];
        }
      };
      auto write_tab = [&]() {  // coefficients of the prefetched hidden unit -> this wave's private table rows
#pragma unroll
        for (int dp = 0; dp < DG_NDP; ++dp) {
          const int i = 2 * DG_NDP * fw + 2 * dp + h;
#pragma unroll
          for (int pg = 0; pg < 2; ++pg) {
            if (pg == 1 && !two_pages) continue;
            const int t = u + 32 * pg;
            float c = hidx[dp][pg] >= 0 ? hv[dp][pg] : 0.f;
            if (pg == 0 && hidx2[dp] >= 0) c += hv2[dp];  // radial + bonded edge of the same pair
            if constexpr (H) c *= scA[dp];
            if (t < PMAX) tabA[i * PMAX + t] = make_float4(c, c * evx[dp][pg], c * evy[dp][pg], c * evz[dp][pg]);
          }
        }
      };
      // (row addresses in tabJ are absolute LDS addresses, so that the row reads are one VGPR + an immediate offset: with a
      // generic base pointer the compiler adds the base of the dynamic LDS block to every address, one v_add per read)
      auto LDSF4 = [](int addr) -> float4 { return lds_read_f4(addr); };
      // The edge loops run in batches of DG_U edges, software pipelined by one batch: while the FMAs of batch b issue, the
      // coefficients and source-row fragments of batch b+1 and the row offsets of batch b+2 are in flight.  Table entries past
      // a destination's degree are zero (PMAX is a multiple of DG_U), so the loads are unconditional and the tail batch needs
      // no predicate.  The VALU of a SIMD issues one wave64 instruction per 4 cycles and the two forming waves of a SIMD
      // saturate it, so the arithmetic is written as PACKED FMAs (v_pk_fma_f32: two products per lane per instruction) on the
      // register pairs the loads deliver: coefficients (c, c vx | c vy, c vz), row fragment (x, T | y, z) — 15 products per edge
      // in 7 packed + 1 scalar FMA with no sign flips inside the loop (negative terms have their own accumulators), and one address add per edge and phase.
// H: plain v_fma_f32, NOT v_pk_fma_f32.  Measured on MI355X (profiles/microbench/README.md, "packed fp32 beside f16 MFMAs"): packed
      // fp32 FMAs of a forming wave occasionally return wrong values in lanes 48..63 while another wave of the SIMD runs
      // v_mfma_f32_*_f16 (about one forward in two had one wrong A-tile row; never with plain FMAs, never beside fp32 MFMAs, which
      // exclude VALU work in time) — and they would not overlap with the MFMAs anyway (valu_rate.hip: zero progress while the
      // matrix pipe is busy, against one plain VALU instruction per ~5.6 cycles).
      auto pk = [](f32x2 a2, f32x2 b2, f32x2 c2) -> f32x2 {
        if constexpr (H) return f32x2{fmaf(a2.x, b2.x, c2.x), fmaf(a2.y, b2.y, c2.y)};
        else return __builtin_elementwise_fma(a2, b2, c2);
      };
      auto pkbhi = [](f32x2 a2, f32x2 b2, f32x2 c2) -> f32x2 {  // (a.x b.y + c.x, a.y b.y + c.y)
        if constexpr (H) return f32x2{fmaf(a2.x, b2.y, c2.x), fmaf(a2.y, b2.y, c2.y)};
        else return pk_bhi(a2, b2, c2);
      };
      auto pkbswap = [](f32x2 a2, f32x2 b2, f32x2 c2) -> f32x2 {  // (a.x b.y + c.x, a.y b.x + c.y)
        if constexpr (H) return f32x2{fmaf(a2.x, b2.y, c2.x), fmaf(a2.y, b2.x, c2.y)};
        else return pk_bswap(a2, b2, c2);
      };
      const int offx = u * 16, offy = OFFY + u * 16;
      // A-tile stores.  fp32: as they are.  H: every value is split hi = rne16(v), lo = rne16(v - hi) (two values per v_cvt_pk) and
      // written to its chunk of [8 hi | 8 lo] halves
      auto store_x = [&](float* __restrict__ Xo, int i, f32x2 a01, f32x2 a23) {
        if (u < 30) {
          if constexpr (H) {
            const unsigned p01 = cvt_pk_f16(a01.x, a01.y), p23 = cvt_pk_f16(a23.x, a23.y);
            const unsigned q01 = cvt_pk_f16(resid_lo(a01.x, p01), resid_hi(a01.y, p01)), q23 = cvt_pk_f16(resid_lo(a23.x, p23), resid_hi(a23.y, p23));
            float* __restrict__ d = Xo + i * XS0 + 8 * (u >> 1) + 2 * (u & 1);  // chunk u / 2; this lane's 4 halves of its hi and lo parts
            *reinterpret_cast<uint2*>(d) = make_uint2(p01, p23);
            *reinterpret_cast<uint2*>(d + 4) = make_uint2(q01, q23);
          } else {
            *reinterpret_cast<float4*>(Xo + i * XS0 + 4 * u) = make_float4(a01.x, a01.y, a23.x, a23.y);
          }
        }
      };
      auto store_y = [&](float* __restrict__ Ydo, float* __restrict__ Yvo, int i, float dv, float x1x, float x1y, float x1z, float cx, float cy, float cz) {
        if constexpr (H) {
          // dot inputs: fp32 as they are (the matrix waves split their fragments of this small tile themselves); vector planes:
          // input order kappa = 2 u + {x1, cross} (the weights are packed in the same order), so that lane u owns ONE dword of
          // the hi part and one of the lo part of chunk u / 4 — whole-dword stores only
          Ydo[i * DG_YD + u] = dv;
          unsigned* __restrict__ yv = reinterpret_cast<unsigned*>(Yvo) + i * DG_YV + 8 * (u >> 2) + (u & 3);
          const float xs_[3] = {x1x, x1y, x1z}, cs_[3] = {cx, cy, cz};
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const unsigned ph = cvt_pk_f16(xs_[m], cs_[m]), pl = cvt_pk_f16(resid_lo(xs_[m], ph), resid_hi(cs_[m], ph));
            yv[m * 32 * DG_YV] = ph;
            yv[m * 32 * DG_YV + 4] = pl;
          }
        } else {
          Ydo[i * DG_YD + u] = dv;
          Yvo[(0 * 32 + i) * DG_YV + u] = x1x; Yvo[(1 * 32 + i) * DG_YV + u] = x1y; Yvo[(2 * 32 + i) * DG_YV + u] = x1z;
          Yvo[(0 * 32 + i) * DG_YV + 32 + u] = cx; Yvo[(1 * 32 + i) * DG_YV + 32 + u] = cy; Yvo[(2 * 32 + i) * DG_YV + 32 + u] = cz;
        }
      };
      // X(k): scalar inputs, lanes u < 30 own channels 4u..4u+3 (lanes 30, 31 compute on x1 data and are not stored)
      auto form_x = [&](float* __restrict__ Xo) {
#pragma unroll
        for (int dp = 0; dp < DG_NDP; ++dp) {
          const int i = 2 * DG_NDP * fw + 2 * dp + h;
          f32x2 ax01 = {0.f, 0.f}, ax23 = {0.f, 0.f};
          const float4* __restrict__ ta = tabA + i * PMAX;
          const int* __restrict__ tj = tabJ + i * PMAX;
          const int nb = dbg_noform ? 0 : (P[dp] + DG_U - 1) / DG_U;
          float cA[DG_U], cB[DG_U];
          float4 xA[DG_U], xB[DG_U];
          int jA[DG_U], jB[DG_U];
          auto ld_j = [&](int (&J)[DG_U], int bb) {
            bb = bb < nb ? bb : nb - 1;
#pragma unroll
            for (int q = 0; q < DG_U; ++q) J[q] = tj[DG_U * bb + q];
          };
          auto ld_cx = [&](float (&c)[DG_U], float4 (&x)[DG_U], const int (&J)[DG_U], int bb) {
            bb = bb < nb ? bb : nb - 1;
#pragma unroll
            for (int q = 0; q < DG_U; ++q) {
              c[q] = ta[DG_U * bb + q].x;
              x[q] = LDSF4(J[q] + offx);
            }
          };
          auto fm = [&](const float (&c)[DG_U], const float4 (&x)[DG_U]) {
#pragma unroll
            for (int q = 0; q < DG_U; ++q) {
              const f32x2 cc = {c[q], c[q]};
              ax01 = pk(cc, f32x2{x[q].x, x[q].y}, ax01);
              ax23 = pk(cc, f32x2{x[q].z, x[q].w}, ax23);
            }
          };
          if (nb > 0) {
            ld_j(jA, 0); ld_cx(cA, xA, jA, 0); ld_j(jB, 1);
            for (int bb = 0; bb < nb; bb += 2) {
              ld_cx(cB, xB, jB, bb + 1); ld_j(jA, bb + 2);
              fm(cA, xA);
              if (bb + 1 < nb) {
                ld_cx(cA, xA, jA, bb + 2); ld_j(jB, bb + 3);
                fm(cB, xB);
              }
            }
          }
          store_x(Xo, i, ax01, ax23);
        }
      };
      // Y(k): vector inputs (lane u = channel u) + the T term (lane u = output channel w')
      auto form_y = [&]() {
#pragma unroll
        for (int dp = 0; dp < DG_NDP; ++dp) {
          const int i = 2 * DG_NDP * fw + 2 * dp + h;
          // accumulator pairs: (a1x, o0) (a1y, a1z) (o1, o2) (d0, d1), scalar d2, and the cross-product terms
          // n3 = (cx y, cx z), n5 = (cy x, cz x), n7 = (cy z, cz y):  acx = n7.y - n7.x, acy = n3.y - n5.y, acz = n5.x - n3.x
          f32x2 p_a1x_o0 = {0.f, O[dp][0]}, p_a1yz = {0.f, 0.f}, p_o12 = {O[dp][1], O[dp][2]}, n3 = {0.f, 0.f}, n5 = {0.f, 0.f},
                n7 = {0.f, 0.f}, p_d = {0.f, 0.f};
          float d2 = 0.f;
          const float4* __restrict__ ta = tabA + i * PMAX;
          const int* __restrict__ tj = tabJ + i * PMAX;
          const int nb = dbg_noform ? 0 : (P[dp] + DG_U - 1) / DG_U;
          float4 cA[DG_U], cB[DG_U], xA[DG_U], xB[DG_U];
          int jA[DG_U], jB[DG_U];
          auto ld_j = [&](int (&J)[DG_U], int bb) {
            bb = bb < nb ? bb : nb - 1;
#pragma unroll
            for (int q = 0; q < DG_U; ++q) J[q] = tj[DG_U * bb + q];
          };
          auto ld_cx = [&](float4 (&c)[DG_U], float4 (&x)[DG_U], const int (&J)[DG_U], int bb) {
            bb = bb < nb ? bb : nb - 1;
#pragma unroll
            for (int q = 0; q < DG_U; ++q) {
              c[q] = ta[DG_U * bb + q];                                            // c, c vx | c vy, c vz
              x[q] = LDSF4(J[q] + offy);         // T_k[j][u], x1_j[u].x | .y, .z
            }
          };
          auto fm = [&](const float4 (&c)[DG_U], const float4 (&x)[DG_U]) {
#pragma unroll
            for (int q = 0; q < DG_U; ++q) {
              const float4 cf = c[q], xv = x[q];  // cf = (c, cx | cy, cz), xv = (x, T | y, z)
              p_a1x_o0 = pk(f32x2{cf.x, cf.y}, f32x2{xv.x, xv.y}, p_a1x_o0);    // (c x, cx T)
              p_a1yz = pk(f32x2{cf.x, cf.x}, f32x2{xv.z, xv.w}, p_a1yz);        // (c y, c z)
              p_o12 = pkbhi(f32x2{cf.z, cf.w}, f32x2{xv.x, xv.y}, p_o12);          // (cy T, cz T)
              n3 = pk(f32x2{cf.y, cf.y}, f32x2{xv.z, xv.w}, n3);                // (cx y, cx z)
              n5 = pk(f32x2{cf.z, cf.w}, f32x2{xv.x, xv.x}, n5);                // (cy x, cz x)
              n7 = pkbswap(f32x2{cf.z, cf.w}, f32x2{xv.z, xv.w}, n7);                // (cy z, cz y)
              p_d = pk(f32x2{cf.z, cf.w}, f32x2{xv.z, xv.w}, p_d);              // (cy y, cz z)
              d2 = fmaf(cf.y, xv.x, d2);                                          // cx x
            }
          };
          if (nb > 0) {
            ld_j(jA, 0); ld_cx(cA, xA, jA, 0); ld_j(jB, 1);
            for (int bb = 0; bb < nb; bb += 2) {
              ld_cx(cB, xB, jB, bb + 1); ld_j(jA, bb + 2);
              fm(cA, xA);
              if (bb + 1 < nb) {
                ld_cx(cA, xA, jA, bb + 2); ld_j(jB, bb + 3);
                fm(cB, xB);
              }
            }
          }
          O[dp][0] = p_a1x_o0.y; O[dp][1] = p_o12.x; O[dp][2] = p_o12.y;
          // (x1 x v)[m] = x1[m+1] v[m+2] - x1[m+2] v[m+1]
          store_y(Yd, Yv, i, (p_d.x + p_d.y) + d2, p_a1x_o0.x, p_a1yz.x, p_a1yz.y, n7.y - n7.x, n3.y - n5.y, n5.x - n3.x);
        }
      };

      // SP: X and Y of one hidden unit in ONE pass over the edges, into A buffer `buf`
      auto form_xy = [&](int buf, bool mid) {
        float* __restrict__ Xo = SPH ? (buf ? X1 : Xt) : Xt + buf * DG_ABUF;
        float* __restrict__ Ydo = SPH ? Yd : Yd + buf * DG_ABUF;
        float* __restrict__ Yvo = SPH ? Yv : Yv + buf * DG_ABUF;
#pragma unroll
        for (int dp = 0; dp < DG_NDP; ++dp) {
          const int i = 2 * DG_NDP * fw + 2 * dp + h;
#if defined(JAMUN_STAMP) || defined(JAMUN_DGDBG)
          if (a.dbg & 64) { if ((dp ^ (fw >> 2)) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
          if (a.dbg & 128) { if ((dp ^ (fw >> 2)) & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1); }
#endif
          f32x2 ax01 = {0.f, 0.f}, ax23 = {0.f, 0.f};
          f32x2 p_a1x_o0 = {0.f, O[dp][0]}, p_a1yz = {0.f, 0.f}, p_o12 = {O[dp][1], O[dp][2]}, n3 = {0.f, 0.f}, n5 = {0.f, 0.f},
                n7 = {0.f, 0.f}, p_d = {0.f, 0.f};
          float d2 = 0.f;
          const float4* __restrict__ ta = tabA + i * PMAX;
          const int* __restrict__ tj = tabJ + i * PMAX;
          const int nb = dbg_noform ? 0 : (P[dp] + DG_U - 1) / DG_U;
          float4 cA[DG_U], cB[DG_U], xA[DG_U], xB[DG_U], z[DG_U];
#if defined(JAMUN_STAMP) || defined(JAMUN_DGDBG)
          if (a.dbg & 256)
            for (int q = 0; q < DG_U; ++q) cA[q] = cB[q] = xA[q] = xB[q] = z[q] = make_float4(1.f + q, 0.5f, 0.25f, 2.f);
#endif
          int jA[DG_U], jB[DG_U];
          auto ld_j = [&](int (&J)[DG_U], int bb) {
            bb = bb < nb ? bb : nb - 1;
#pragma unroll
            for (int q = 0; q < DG_U; ++q) J[q] = tj[DG_U * bb + q] + offx;  // ONE row address per edge: scalar channels at +0, vector block at +480
          };
          auto ld_cx = [&](float4 (&c)[DG_U], float4 (&x)[DG_U], const int (&J)[DG_U], int bb) {
            bb = bb < nb ? bb : nb - 1;
#if defined(JAMUN_STAMP) || defined(JAMUN_DGDBG)
            if (a.dbg & 256) return;
#endif
#pragma unroll
            for (int q = 0; q < DG_U; ++q) {
              c[q] = ta[DG_U * bb + q];                                             // c, c vx | c vy, c vz
              x[q] = LDSF4(J[q] + 480);            // x1_j[u].x, T_k[j][u] | .y, .z
            }
          };
          // the scalar channels of a batch are requested at the top of its FMA block and consumed at its end (behind the 32
          // packed FMAs of the vector part): single-buffered, 16 registers instead of 32
          // (H: plain C++ FMAs leave the compiler free to reorder, and it moves the four FMAs that consume z right behind the z
          // loads — a full LDS round trip exposed per batch; scheduling barriers keep "all loads of the next batch, then the 60
          // FMAs of the vector part, then the 16 of the scalar part")
          auto fm = [&](const float4 (&c)[DG_U], const float4 (&x)[DG_U], const int (&J)[DG_U], int (&Jn)[DG_U], int bn) {
#if defined(JAMUN_STAMP) || defined(JAMUN_DGDBG)
            if (!(a.dbg & 256))
#endif
#pragma unroll
            for (int q = 0; q < DG_U; ++q) z[q] = LDSF4(J[q]);  // x0_j[4u..4u+3]
            int raw[DG_U];
            if constexpr (H) {  // row offsets of the batch after next (J itself is dead after the z requests); the lane's column
                                // offset is added behind the FMAs — adding it here would wait for every read issued above
              const int bc = bn < nb ? bn : nb - 1;
#pragma unroll
              for (int q = 0; q < DG_U; ++q) raw[q] = tj[DG_U * bc + q];
              __builtin_amdgcn_sched_barrier(0);
            }
#if defined(JAMUN_STAMP) || defined(JAMUN_DGDBG)
            if (a.dbg & 512) {
#pragma unroll
              for (int q = 0; q < DG_U; ++q) { d2 += c[q].x + x[q].y + z[q].z; }
#pragma unroll
              for (int q = 0; q < DG_U; ++q) Jn[q] = raw[q] + offx;
              return;
            }
#endif
#pragma unroll
            for (int q = 0; q < DG_U; ++q) {
              const float4 cf = c[q], xv = x[q];
              p_a1x_o0 = pk(f32x2{cf.x, cf.y}, f32x2{xv.x, xv.y}, p_a1x_o0);    // (c x, cx T)
              p_a1yz = pk(f32x2{cf.x, cf.x}, f32x2{xv.z, xv.w}, p_a1yz);        // (c y, c z)
              p_o12 = pkbhi(f32x2{cf.z, cf.w}, f32x2{xv.x, xv.y}, p_o12);          // (cy T, cz T)
              n3 = pk(f32x2{cf.y, cf.y}, f32x2{xv.z, xv.w}, n3);                // (cx y, cx z)
              n5 = pk(f32x2{cf.z, cf.w}, f32x2{xv.x, xv.x}, n5);                // (cy x, cz x)
              n7 = pkbswap(f32x2{cf.z, cf.w}, f32x2{xv.z, xv.w}, n7);                // (cy z, cz y)
              p_d = pk(f32x2{cf.z, cf.w}, f32x2{xv.z, xv.w}, p_d);              // (cy y, cz z)
              d2 = fmaf(cf.y, xv.x, d2);                                          // cx x
            }
            if constexpr (H) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < DG_U; ++q) {
              const f32x2 cc = {c[q].x, c[q].x};
              ax01 = pk(cc, f32x2{z[q].x, z[q].y}, ax01);
              ax23 = pk(cc, f32x2{z[q].z, z[q].w}, ax23);
            }
            if constexpr (H) {
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int q = 0; q < DG_U; ++q) Jn[q] = raw[q] + offx;
              __builtin_amdgcn_sched_barrier(0);
            }
          };
          if (nb > 0) {
            ld_j(jA, 0); ld_cx(cA, xA, jA, 0); ld_j(jB, 1);
            for (int bb = 0; bb < nb; bb += 2) {
              ld_cx(cB, xB, jB, bb + 1);
              fm(cA, xA, jA, jA, bb + 2);
              if constexpr (!H) ld_j(jA, bb + 2);
              if (bb + 1 < nb) {
                ld_cx(cA, xA, jA, bb + 2);
                fm(cB, xB, jB, jB, bb + 3);
                if constexpr (!H) ld_j(jB, bb + 3);
              }
            }
          }
          O[dp][0] = p_a1x_o0.y; O[dp][1] = p_o12.x; O[dp][2] = p_o12.y;
          store_x(Xo, i, ax01, ax23);
          if constexpr (SPH) {
            if (dp == 0 && mid) {
              DSTAMP(tb); DTRACE(1, tb);
              LDS_BARRIER();  // the matrix waves have finished Y(k) x W: the single Y tile may be rewritten
            }
          }
          store_y(Ydo, Yvo, i, (p_d.x + p_d.y) + d2, p_a1x_o0.x, p_a1yz.x, p_a1yz.y, n7.y - n7.x, n3.y - n5.y, n5.x - n3.x);
        }
      };

      if constexpr (SP) {
        load_h(k_of(0));
        write_tab();
        load_h(k_of(1));
        LDS_BARRIER();                    // T(k0) in its slots
        form_xy(0, false);
        if (nk > 1) { write_tab(); load_h(k_of(2)); }
        LDS_BARRIER();                    // A(k0) complete
        LDS_BARRIER();                    // T(k1) in its slots
        DSTAMP(tb); DACC(0, tb, ta);
        for (int it = 0; it < nk; ++it) {
          DSTAMP(ta);
          if (it + 1 < nk) {
            form_xy((it + 1) & 1, true);
            if (it + 2 < nk) { write_tab(); load_h(k_of(it + 3)); }
          } else if constexpr (SPH) {
            LDS_BARRIER();                // (the mid-step barrier of the last hidden unit: nothing left to form)
          }
          DTRACE(0, ta);
          DSTAMP(tb); DACC(1, tb, ta);
          DTRACE(SPH ? 2 : 1, tb);
          LDS_BARRIER();
          DSTAMP(ta); DACC(2, ta, tb);
          if constexpr (!SPH) DTRACE(2, ta);
          LDS_BARRIER();
          DSTAMP(tb); DACC(4, tb, ta);
          DTRACE(3, tb);
          ++trc;
        }
      } else if constexpr (ALT) {
        // ---- pass A (vector blocks resident)
        load_h(k_of(0));
        write_tab();
        load_h(k_of(1));
        DSTAMP(tb); DACC(0, tb, ta);
        for (int it = 0; it < nk; ++it) {
          DSTAMP(ta);
          form_y();                       // Y(k): table and T slots of k
          DSTAMP(tb); DACC(1, tb, ta);
          LDS_BARRIER();
          DSTAMP(ta); DACC(2, ta, tb);
          if (it + 1 < nk) {              // (the matrix waves contract Y(k) meanwhile)
            if (!dbg_nostage) stage_t(k_of(it + 1));        // (first: its wait would otherwise cover the h loads as well)
            write_tab();
            load_h(k_of(it + 2));
          }
          DSTAMP(tb); DACC(3, tb, ta);
          LDS_BARRIER();
          DSTAMP(ta); DACC(4, ta, tb);
        }
        // ---- pass B (scalar channels resident)
        DSTAMP(ta);
        load_h(k_of(0));
        stage_x0();
        write_tab();
        load_h(k_of(1));
        LDS_BARRIER();                    // scalar channels resident
        form_x(Xt);
        if (nk > 1) { write_tab(); load_h(k_of(2)); }
        LDS_BARRIER();                    // X(k0) complete
        DSTAMP(tb); DACC(0, tb, ta);
        for (int it = 0; it < nk; ++it) {
          DSTAMP(ta);
          if (it + 1 < nk) {
            form_x((it & 1) ? Xt : Yd);   // X(k+1) into the other buffer
            if (it + 2 < nk) { write_tab(); load_h(k_of(it + 3)); }
          }
          DSTAMP(tb); DACC(1, tb, ta);
          LDS_BARRIER();
          DSTAMP(ta); DACC(2, ta, tb);
        }
      } else {
      // "P2(-1)": coefficients and X of the first hidden unit
        load_h(k_of(0));
        write_tab();
        load_h(k_of(1));
        form_x(Xt);
        LDS_BARRIER();
        DSTAMP(tb); DACC(0, tb, ta);
        for (int it = 0; it < nk; ++it) {
          const bool has_next = it + 1 < nk;
          // ---- P1: Y(k) with the table of k (written in the previous P2) and T(k)
          DSTAMP(ta);
          form_y();
          DSTAMP(tb); DACC(1, tb, ta);
          LDS_BARRIER();
          DSTAMP(ta); DACC(2, ta, tb);
          // ---- P2: table of k+1, X(k+1)
          if (has_next) {
            write_tab();
            load_h(k_of(it + 2));
            form_x(Xt);
          }
          DSTAMP(tb); DACC(3, tb, ta);
          LDS_BARRIER();
          DSTAMP(ta); DACC(4, ta, tb);
        }
      }
      DSTAMP(ta);
      // ---- segment end: the accumulated vector rows (T term) join the staging tile (OL aliases the source rows: dead by now)
#pragma unroll
      for (int dp = 0; dp < DG_NDP; ++dp) {
        const int i = 2 * DG_NDP * fw + 2 * dp + h;
#pragma unroll
        for (int m = 0; m < 3; ++m) OL[i * 96 + m * 32 + u] = H ? O[dp][m] * iscA[dp] : O[dp][m];
      }
      DSTAMP(tb); DACC(5, tb, ta);
    }
    // ---- all threads: staged tile -> partial slab of this segment, coalesced 16-byte stores in fixed summation order
    LDS_BARRIER();
    {
      float* __restrict__ p0 = a.partial0 + ((size_t)slab * a.n_pad + n0) * (size_t)(a.nt0 * 32);
      float* __restrict__ p1 = a.partial1 + ((size_t)slab * a.n_pad + n0) * 96;
      for (int idx = tid; idx < 32 * 40; idx += DG_THREADS) {
        const int row = idx / 40, c4 = idx - row * 40;
        float4 v;
        if (c4 < 32) v = *reinterpret_cast<const float4*>(OM + row * 128 + 4 * c4);
        else {
          const float* __restrict__ q0 = OQ + row * 32 + 4 * (c4 - 32);
          const float4 a0 = *reinterpret_cast<const float4*>(q0), a1 = *reinterpret_cast<const float4*>(q0 + 1024),
                       a2 = *reinterpret_cast<const float4*>(q0 + 2048), a3 = *reinterpret_cast<const float4*>(q0 + 3072);
          v = make_float4(((a0.x + a1.x) + a2.x) + a3.x, ((a0.y + a1.y) + a2.y) + a3.y, ((a0.z + a1.z) + a2.z) + a3.z,
                          ((a0.w + a1.w) + a2.w) + a3.w);
        }
        if constexpr (H) {  // the column scales of the balanced weights (exact powers of two)
          const float4 cf = reinterpret_cast<const float4*>(a.cf0)[c4];
          v = make_float4(v.x * cf.x, v.y * cf.y, v.z * cf.z, v.w * cf.w);
        }
        if (row < n_dst) *reinterpret_cast<float4*>(p0 + row * 160 + 4 * c4) = v;
      }
      for (int idx = tid; idx < 32 * 24; idx += DG_THREADS) {
        const int row = idx / 24, c4 = idx - row * 24;
        float4 pv = *reinterpret_cast<const float4*>(OP + row * 96 + 4 * c4);
        const float4 ov = *reinterpret_cast<const float4*>(OL + row * 96 + 4 * c4);
        if constexpr (H) {
          const float4 cf = reinterpret_cast<const float4*>(a.cf1)[c4 & 7];
          pv = make_float4(pv.x * cf.x, pv.y * cf.y, pv.z * cf.z, pv.w * cf.w);
        }
        if (row < n_dst) *reinterpret_cast<float4*>(p1 + row * 96 + 4 * c4) = make_float4(pv.x + ov.x, pv.y + ov.y, pv.z + ov.z, pv.w + ov.w);
      }
    }
    if constexpr (H)
      for (int i = tid; i < a.RS; i += DG_THREADS) rowmax[i] = 0u;  // (its readers finished in the segment prologue)
    LDS_BARRIER();  // the next segment rewrites the tiles
  }
#ifdef JAMUN_STAMP
  if (lane0 == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_dgstamp[is_mat ? 0 : 1][i], st_acc[i]);
#endif
}

#ifdef TP_TRACE
__device__ unsigned long long g_tptrace[1024][16];
#endif
void conv_dg_print_stamps() {
#ifdef TP_TRACE
  {
    static unsigned long long tp[1024][16];
    if (hipMemcpyFromSymbol(tp, HIP_SYMBOL(g_tptrace), sizeof(tp)) == hipSuccess) {
      unsigned long long rt0 = ~0ull, rt1 = 0;
      int nw = 0;
      for (int g = 0; g < 1024; ++g)
        if (tp[g][0]) { rt0 = std::min(rt0, tp[g][0]); rt1 = std::max(rt1, tp[g][14]); ++nw; }
      fprintf(stderr, "tprod trace: %d waves; first entry -> last exit %llu ticks of s_memrealtime (100 MHz)\n", nw, rt1 - rt0);
      fprintf(stderr, "  wave: entry (rt ticks after first) | cycles after entry: rows arrived, split done, units..., exit | exit (rt ticks after first entry)\n");
      for (int g = 0; g < 1024; ++g) {
        if (!tp[g][0] || !(g % 37 == 0 || g >= 945)) continue;
        fprintf(stderr, "  %4d: %4llu |", g, tp[g][0] - rt0);
        fprintf(stderr, " %6lld %6lld |", (long long)(tp[g][15] - tp[g][1]), (long long)(tp[g][2] - tp[g][1]));
        for (int i = 3; i < 13; ++i) fprintf(stderr, " %6lld", tp[g][i] ? (long long)(tp[g][i] - tp[g][1]) : 0ll);
        fprintf(stderr, " | %6lld | %4llu\n", (long long)(tp[g][13] - tp[g][1]), tp[g][14] - rt0);
      }
      // distribution of entry and exit times
      unsigned long long emax = 0, xmin = ~0ull; double esum = 0, xsum = 0, dsum = 0;
      for (int g = 0; g < 1024; ++g) if (tp[g][0]) { emax = std::max(emax, tp[g][0] - rt0); xmin = std::min(xmin, tp[g][14] - rt0); esum += tp[g][0] - rt0; xsum += tp[g][14] - rt0; dsum += tp[g][13] - tp[g][1]; }
      fprintf(stderr, "  entry: mean %.1f max %llu ticks; exit: min %llu mean %.1f ticks; mean cycles per wave %.0f\n", esum / nw, emax, xmin, xsum / nw, dsum / nw);
    }
  }
#endif
#ifdef JAMUN_STAMP
  unsigned long long v[2][8], z[2][8] = {};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_dgstamp), sizeof(v)) != hipSuccess) return;
  const char* names[6] = {"prologue", "P1 work", "P1 wait", "P2 work", "P2 wait", "epilogue"};
  for (int r = 0; r < 2; ++r) {
    fprintf(stderr, "dg stamps %s waves (cycles summed over waves):", r == 0 ? "matrix" : "forming");
    for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %llu", names[i], v[r][i]);
    fprintf(stderr, "\n");
  }
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dgstamp), z, sizeof(z));
  static unsigned long long tr[12][48][4];
  if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_dgtrace), sizeof(tr)) == hipSuccess && tr[0][0][0] != 0) {
    const unsigned long long t0 = tr[0][0][0];
    fprintf(stderr, "dg trace (workgroup 7): per k-step, per wave (0-3 matrix, 4-11 forming): start / arrive A / release A / end (mode 3: start / arrive mid / arrive A / end), cycles since first start\n");
    for (int st = 0; st < 12; ++st) {
      for (int w = 0; w < 12; ++w)
        fprintf(stderr, "  step %2d wave %2d: %7lld %7lld %7lld %7lld\n", st, w, (long long)(tr[w][st][0] - t0), (long long)(tr[w][st][1] - t0),
                (long long)(tr[w][st][2] - t0), (long long)(tr[w][st][3] - t0));
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// k_tprod — pre-pass of a hidden layer: T[k][j][w'] = sum_u x0_j[u] W~[(k,u)][w'] for every atom j and hidden unit k (the
// scalar-input -> vector-row weights; k_conv_dg applies  out_m[i][w'] += sum_e (c_e v_m) T_k[j_e][w']).  Computing it once per
// source atom costs 2*120*32 FLOP per (atom, k); inside k_conv_dg it was recomputed for every tile whose span contains the atom
// (x1.6 for 17-atom molecules, x5 for a 166-atom molecule: 352 of 828 MFMA units per (tile, k)).
// One wave = 32 atoms x a contiguous run of hidden units: the inputs (32 atoms x 120 channels) are loaded into registers once,
// the 15 weight blocks of hidden unit k+1 are in flight (second register buffer) while the 60 v_mfma_f32_32x32x2 of k run.
// The MFMA computes the TRANSPOSED tile (weights as the A operand, inputs as B: both fragments have the same lane layout, so
// this is only the operand order): lane (atom r, half hh) then holds 4 consecutive output channels per accumulator quad and
// the 32 x 32 tile goes to HBM as four 16-byte stores per lane instead of sixteen 4-byte stores with their address and
// predicate arithmetic (VALU work that the matrix pipe of the SIMD waits for).  No LDS, no barriers.
// The launch is sized to ONE wave per SIMD (1024 waves: k-groups = 1024 / row tiles).  Measured on MI355X, 4352 atoms:
// 4352 short waves of ~2 hidden units 36 us; 2040 waves (two per SIMD) 38 us; 952 waves 27 us (standalone: profiles/microbench/tprod_bench.hip).
#define TP_WAVES 4
#define TP_LD 36  // floats per row of a wave's output staging tile (144 bytes: 16-byte aligned, rows 4 banks apart)
__global__ __launch_bounds__(64 * TP_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_tprod(
    const float* __restrict__ x, int XS, int n_atoms, int n_k, int kg, const float4* __restrict__ wt, float* __restrict__ T) {
  const int lane = threadIdx.x & 63, wave = RFL(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int gid = blockIdx.x * TP_WAVES + wave;
  const int tile = RFL(gid / kg), g = RFL(gid - tile * kg);
  const int a0 = tile * 32;
  if (a0 >= n_atoms) return;
  const int k_lo = RFL((g * n_k) / kg), k_hi = RFL(((g + 1) * n_k) / kg);
  if (k_lo >= k_hi) return;
  const int row = min(a0 + r, n_atoms - 1);
  float4 xv[15], w0[15], w1[15];
  auto load_w = [&](float4 (&wv)[15], int k) {
    const float4* __restrict__ wk = wt + (size_t)min(k, n_k - 1) * 15 * 64;
#pragma unroll
    for (int q = 0; q < 15; ++q) wv[q] = wk[q * 64 + lane];
  };
#pragma unroll
  for (int q = 0; q < 15; ++q) xv[q] = *reinterpret_cast<const float4*>(x + (size_t)row * XS + 8 * q + 4 * hh);
  load_w(w0, k_lo);
  auto step = [&](const float4 (&wv)[15], int k) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int q = 0; q < 15; ++q) {  // acc[row = output channel][column = atom]
      acc = MFMA32(wv[q].x, xv[q].x, acc);
      acc = MFMA32(wv[q].y, xv[q].y, acc);
      acc = MFMA32(wv[q].z, xv[q].z, acc);
      acc = MFMA32(wv[q].w, xv[q].w, acc);
    }
    float* __restrict__ tk = T + ((size_t)k * n_atoms + a0) * 32;
    if (a0 + r < n_atoms) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)  // accumulator register 4 g4 + i  <->  output channel 8 g4 + 4 hh + i
        *reinterpret_cast<float4*>(tk + r * 32 + 8 * g4 + 4 * hh) = make_float4(acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]);
    }
  };
  for (int k = k_lo; k < k_hi; k += 2) {
    load_w(w1, k + 1);
    step(w0, k);
    if (k + 1 < k_hi) {
      load_w(w0, k + 2);
      step(w1, k + 1);
    }
  }
}

// The same product as f16x3 (as the contraction of k_conv_dg<.., true>): the 120 scalar inputs of the wave's 32 atoms are scaled by a
// power of two per atom (largest |x| below 2^14), split ONCE into hi + lo halves and held in registers as the B operands of
// v_mfma_f32_32x32x16_f16 (lane (atom, hh): inputs 16 g + 8 hh + j); per hidden unit the 16 weight blocks (8 groups x hi, lo; split
// on the host after scaling by 2^sBt) stream through a double buffer and 24 MFMAs of 32 cycles replace 60 of 64.  The kernel turns
// from MFMA-bound to bound by its weight stream (16 KB per wave and hidden unit from L2).
#ifdef TP_TRACE
#define TSTAMP(slot) do { if (gid < 1024) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0) g_tptrace[gid][slot] = t_; } } while (0)
#define TSTAMP_RT(slot) do { if (gid < 1024) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); if (lane == 0) g_tptrace[gid][slot] = t_; } } while (0)
#define TSTAMP_V(slot, v) do { if (gid < 1024) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(v) :: "memory"); if (lane == 0) g_tptrace[gid][slot] = t_; } } while (0)
#else
#define TSTAMP(slot) do { } while (0)
#define TSTAMP_RT(slot) do { } while (0)
#define TSTAMP_V(slot, v) do { } while (0)
#endif
__global__ __launch_bounds__(64 * TP_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_tprod_h(
    const float* __restrict__ x, int XS, int n_atoms, int n_k, int kg, const float4* __restrict__ wth, const float* __restrict__ gT, const float* __restrict__ cfT,
    float* __restrict__ T, int t_stride) {
  const int lane = threadIdx.x & 63, wave = RFL(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int gid = blockIdx.x * TP_WAVES + wave;
  const int tile = RFL(gid / kg), g = RFL(gid - tile * kg);
  const int a0 = tile * 32;
  if (a0 >= n_atoms) return;
  const int k_lo = RFL((g * n_k) / kg), k_hi = RFL(((g + 1) * n_k) / kg);
  if (k_lo >= k_hi) return;
  const int row = min(a0 + r, n_atoms - 1);
  TSTAMP_RT(0);
  TSTAMP(1);
  float4 xh[8], xl[8], w0[16], w1[16];
  float isc;
  auto load_w = [&](float4 (&wv)[16], int k) {
    const float4* __restrict__ wk = wth + (size_t)min(k, n_k - 1) * 16 * 64;
#pragma unroll
    for (int q = 0; q < 16; ++q) wv[q] = wk[q * 64 + lane];
  };
  load_w(w0, k_lo);  // (requested with the feature rows, not behind their split: one round trip less per wave)
  {
    float4 xf[8][2];
    float mx = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
        const int u = 16 * q + 8 * hh + 4 * p2;  // inputs u .. u + 3 (120..127: beyond the scalar block -> zeros)
        float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * XS + u);
        if (u >= 120) v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 gv = *reinterpret_cast<const float4*>(gT + u);  // 2^e_u of the input channels (the host took it out of their weight rows)
        v = make_float4(v.x * gv.x, v.y * gv.y, v.z * gv.z, v.w * gv.w);
        xf[q][p2] = v;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // both input halves of the atom
    TSTAMP_V(15, mx);
    if (t_stride > 0) {  // transposed output: one scale for the wave's 32 atoms (a register of the accumulator then holds four ATOMS)
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    int sA = 0;
    if (mx > 0.f) sA = 14 - ((int)((__float_as_uint(mx) >> 23) & 0xffu) - 126);
    sA = max(-60, min(60, sA));
    const float sc = pow2f(sA);
    isc = pow2f(-sA);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 a4 = xf[q][0], b4 = xf[q][1];
      const float e[8] = {a4.x * sc, a4.y * sc, a4.z * sc, a4.w * sc, b4.x * sc, b4.y * sc, b4.z * sc, b4.w * sc};
      unsigned ph[4], pl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ph[i] = cvt_pk_f16(e[2 * i], e[2 * i + 1]);
        pl[i] = cvt_pk_f16(resid_lo(e[2 * i], ph[i]), resid_hi(e[2 * i + 1], ph[i]));
      }
      xh[q] = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
      xl[q] = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
    }
  }
  { float v_ = xl[7].w; TSTAMP_V(2, v_); xl[7].w = v_; }
  __shared__ float tp_tile[TP_WAVES][32 * TP_LD];
  float* __restrict__ tt = tp_tile[wave];
  // (inverse column scales of the weights: transposed output — lane = output channel r: one factor; else register 4 g4 + i = channel 8 g4 + 4 hh + i)
  const float cfr = cfT[r];
  float4 cfq[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) cfq[g4] = *reinterpret_cast<const float4*>(cfT + 8 * g4 + 4 * hh);
  auto stage = [&](const f32x16& acc) {  // row r of the tile <- this lane's four quads (columns 8 g4 + 4 hh ..), scaled
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 c = t_stride > 0 ? make_float4(cfr, cfr, cfr, cfr) : cfq[g4];
      *reinterpret_cast<float4*>(tt + r * TP_LD + 8 * g4 + 4 * hh) =
          make_float4((acc[4 * g4] * isc) * c.x, (acc[4 * g4 + 1] * isc) * c.y, (acc[4 * g4 + 2] * isc) * c.z, (acc[4 * g4 + 3] * isc) * c.w);
    }
  };
  auto step = [&](const float4 (&wv)[16], int k) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    if (t_stride > 0) {
      // transposed for k_conv_mf, [k][w'][atom]: the operands swap roles (both fragments have the same lane layout), acc[row = atom][column
      // = output channel], so lane (channel r, hh) holds four consecutive atoms per accumulator quad: 16-byte stores (rows past the
      // last atom land in the slack of the buffer)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        acc = MFMA32H(xh[q], wv[2 * q + 1], acc);
        acc = MFMA32H(xl[q], wv[2 * q], acc);
        acc = MFMA32H(xh[q], wv[2 * q], acc);
      }
      // lane (r, hh) holds 16-byte pieces of 32 different rows: through the wave's LDS tile, so that a store instruction writes eight
      // whole 128-byte row segments (eight lanes per row) instead of 64 half sectors — the 36 MB of T were bound by the L2 request rate
      stage(acc);
      const int R = lane >> 3, c = lane & 7;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<float4*>(T + ((size_t)k * 32 + 8 * j + R) * t_stride + a0 + 4 * c) = *reinterpret_cast<const float4*>(tt + (8 * j + R) * TP_LD + 4 * c);
      return;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // acc[row = output channel][column = atom]; blocks 2q (hi), 2q + 1 (lo) of the weights
      acc = MFMA32H(wv[2 * q + 1], xh[q], acc);
      acc = MFMA32H(wv[2 * q], xl[q], acc);
      acc = MFMA32H(wv[2 * q], xh[q], acc);
    }
    float* __restrict__ tk = T + ((size_t)k * n_atoms + a0) * 32;
    stage(acc);  // (accumulator register 4 g4 + i  <->  output channel 8 g4 + 4 hh + i of atom r: whole 128-byte atom rows per eight lanes)
    const int R = lane >> 3, c = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (a0 + 8 * j + R < n_atoms) *reinterpret_cast<float4*>(tk + (8 * j + R) * 32 + 4 * c) = *reinterpret_cast<const float4*>(tt + (8 * j + R) * TP_LD + 4 * c);
  };
  for (int k = k_lo; k < k_hi; k += 2) {
    load_w(w1, k + 1);
    step(w0, k);
    if (k - k_lo < 10) TSTAMP(3 + k - k_lo);
    if (k + 1 < k_hi) {
      load_w(w0, k + 2);
      step(w1, k + 1);
      if (k + 1 - k_lo < 10) TSTAMP(3 + k + 1 - k_lo);
    }
  }
  TSTAMP(13);
  TSTAMP_RT(14);
}

// k_tprod_t — the transposed pre-pass ([k][w'][atom], what k_conv_mf / k_conv_ml stage) as a kernel of its own (round 6).
// Per-wave timeline of k_tprod_h on cfg2 (-DTP_TRACE, profiles/r6_cfg2_tprod_node_trace.txt): ~3.4 us until the feature rows have arrived
// (every lane fetches ITS row in sixteen 16-byte pieces: 32 cache lines per load instruction, seven k-groups fetch every tile), 0.45 us of
// split, then 9-10 hidden units of ~1750 cycles each around a chain of 24 MFMAs (768 cycles).  What holds a unit is the CU's vector-memory
// path: four waves x (16 KB of weight blocks + 4 KB of stores) per unit is ~47 B/clk of its 64 — a first rewrite with two tiles per wave
// and the stores between the MFMAs (24 KB per wave and unit) ran at the same rate per byte.  Here the weights go through LDS:
//   * a workgroup = 4 waves x 64 atoms walking the SAME run of hidden units; a unit's 16 weight blocks are fetched ONCE per workgroup (each
//     wave a quarter, a unit ahead, through registers into the other half of a 2 x 16 KB LDS buffer; one barrier per unit) and read from
//     LDS as A fragments, each read feeding six MFMAs (two 32-atom tiles x f16x3);
//   * the rows are fetched whole (a load instruction = two rows of 512 bytes), scaled by the tile's power of two, split once, transposed
//     through LDS ([plane][2 q + hh][atom] x 16 bytes, blocks 528 bytes apart: the 8-byte writes of a row and the 16-byte reads of a
//     fragment are conflict-free) and then HELD IN REGISTERS as B fragments (128 of them);
//   * the weights are the A operand, so a lane of the accumulator is an ATOM and a register a channel: a row of T is 32 consecutive lanes —
//     two whole 128-byte lines per dword store, no staging tile; the stores of a finished unit (32 products with the column factors, 32
//     dword stores) are placed between the MFMAs of the next one, two sets of accumulators taking turns.
// Per CU and unit: 16 KB of loads + 32 KB of stores (was 64 + 16) against 48 x 4 MFMAs.
// Same products in the same order as k_tprod_h's transposed branch (operand roles swapped: each output element sums the same K sequence);
// all scale factors are powers of two: T is bit-identical.
#define TT_WAVES 4
#define TT_BLK 528                      // bytes of a fragment block: 32 lanes x 16 + 16
#define TT_PLANE (16 * TT_BLK)          // hi -> lo
#define TT_TILE (2 * TT_PLANE)          // 32 atoms: a wave's staging area
#define TT_WBUF (16 * 1024)             // a unit's weight blocks
#define TT_LDS_BYTES (TT_WAVES * TT_TILE + 2 * TT_WBUF)
__global__ __launch_bounds__(64 * TT_WAVES) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_tprod_t(
    const float* __restrict__ x, int XS, int n_atoms, int n_k, int kg, const float4* __restrict__ wth, const float* __restrict__ gT, const float* __restrict__ cfT,
    float* __restrict__ T, int t_stride) {
  extern __shared__ float4 tt_lds4[];
  const int lane = threadIdx.x & 63, wave = RFL(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int rowg = RFL((int)blockIdx.x / kg), g = RFL((int)blockIdx.x - rowg * kg);
  const int a0 = (rowg * TT_WAVES + wave) * 64;
  const bool active = a0 < n_atoms;  // wave-uniform; a wave without atoms still fetches its share of the weights and joins the barriers
  const int k_lo = RFL((g * n_k) / kg), k_hi = RFL(((g + 1) * n_k) / kg);
  if (k_lo >= k_hi) return;  // (workgroup-uniform)
  char* __restrict__ xs = reinterpret_cast<char*>(tt_lds4) + wave * TT_TILE;
  char* __restrict__ wl = reinterpret_cast<char*>(tt_lds4) + TT_WAVES * TT_TILE;  // [2][16 blocks][64 lanes] x 16 bytes
#ifdef TP_TRACE
  const int gid = blockIdx.x * TT_WAVES + wave;
#endif
  TSTAMP_RT(0);
  TSTAMP(1);
  // this wave's quarter of a unit's weight blocks: 4 wave .. 4 wave + 3
  struct Quarter { float4 b0, b1, b2, b3; };
  auto load_w = [&](int k) {
    const float4* __restrict__ wk = wth + ((size_t)min(k, n_k - 1) * 16 + 4 * wave) * 64 + lane;
    return Quarter{wk[0], wk[64], wk[128], wk[192]};
  };
  auto put_w = [&](int buf, const Quarter& w) {
    char* __restrict__ d = wl + buf * TT_WBUF + 4 * wave * 1024 + 16 * lane;
    *reinterpret_cast<float4*>(d) = w.b0;
    *reinterpret_cast<float4*>(d + 1024) = w.b1;
    *reinterpret_cast<float4*>(d + 2048) = w.b2;
    *reinterpret_cast<float4*>(d + 3072) = w.b3;
  };
  const Quarter wfirst = load_w(k_lo);  // (requested with the rows)
  // column factors of this lane's sixteen accumulator registers (register 4 g4 + i <-> channel 8 g4 + 4 hh + i)
  float cf[16];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const float4 c = *reinterpret_cast<const float4*>(cfT + 8 * g4 + 4 * hh);
    cf[4 * g4] = c.x; cf[4 * g4 + 1] = c.y; cf[4 * g4 + 2] = c.z; cf[4 * g4 + 3] = c.w;
  }
  float isc[2] = {0.f, 0.f};
  float4 xh[2][8], xl[2][8];  // B fragments of the two tiles: lane (atom r, hh), halves <-> inputs 16 q + 8 hh + j
  if (active) {
    // rows: load instruction n of tile t = rows 2n, 2n + 1 (lanes 0..31 / 32..63), inputs 4 r .. 4 r + 3 (120..127: beyond the scalar block -> zeros)
    // 2^e_u of the input channels (the host took it out of their weight rows); zero for the lanes beyond the scalar block (a select on the
    // loaded value instead would be compiled into a branch around the load: sixteen serialised round trips)
    float4 gv = *reinterpret_cast<const float4*>(gT + min(4 * r, 116));
    const float km = 4 * r < 120 ? 1.f : 0.f;
    gv = make_float4(gv.x * km, gv.y * km, gv.z * km, gv.w * km);
    const int wofs = (r >> 1) * TT_BLK + 8 * (r & 1);  // block 2 q + hh of this lane's inputs (q = r / 4, hh = (r / 2) & 1), first or second half of its 16 bytes
    const float m1 = opaque_minus_one();
    float4 xa[2][16];  // (all 32 requests of the wave's 64 rows in flight together)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        // (rows past the last atom repeat it: the tile's maximum is unchanged; their columns of T land in the slack of the rows, as k_tprod_h's)
        const int row = min(a0 + 32 * t + 2 * n + hh, n_atoms - 1);
        xa[t][n] = *reinterpret_cast<const float4*>(x + (size_t)row * XS + min(4 * r, 116));
      }
    __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks the second tile's requests behind the first tile's split: a second round trip)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float4 xf[16];
      float mx = 0.f;
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        float4 v = xa[t][n];
        v = make_float4(v.x * gv.x, v.y * gv.y, v.z * gv.z, v.w * gv.w);
        xf[n] = v;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));  // one scale for the tile's 32 atoms
      if (t == 0) TSTAMP_V(15, mx);
      int sA = 0;
      if (mx > 0.f) sA = 14 - ((int)((__float_as_uint(mx) >> 23) & 0xffu) - 126);
      sA = max(-60, min(60, sA));
      const float sc = pow2f(sA);
      isc[t] = pow2f(-sA);
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        const float e0 = xf[n].x * sc, e1 = xf[n].y * sc, e2 = xf[n].z * sc, e3 = xf[n].w * sc;
        const unsigned p0 = cvt_pk_f16_c(e0, e1), p1 = cvt_pk_f16_c(e2, e3);
        const unsigned q0 = cvt_pk_f16_c(resid_lo_c(e0, p0, m1), resid_hi_c(e1, p0, m1)), q1 = cvt_pk_f16_c(resid_lo_c(e2, p1, m1), resid_hi_c(e3, p1, m1));
        *reinterpret_cast<uint2*>(xs + wofs + 16 * (2 * n + hh)) = make_uint2(p0, p1);
        *reinterpret_cast<uint2*>(xs + wofs + TT_PLANE + 16 * (2 * n + hh)) = make_uint2(q0, q1);
      }
      // (the staging area is this wave's own: its LDS writes and reads are ordered by the counters, no barrier)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        xh[t][q] = *reinterpret_cast<const float4*>(xs + (2 * q + hh) * TT_BLK + 16 * r);
        xl[t][q] = *reinterpret_cast<const float4*>(xs + TT_PLANE + (2 * q + hh) * TT_BLK + 16 * r);
      }
    }
  }
  // (consumed here once: first used inside the loop, the compiler's wait for these loads would sit at the loop's top and drain the
  // requests and stores in flight there in EVERY iteration — vmcnt counts in order)
#pragma unroll
  for (int i_ = 0; i_ < 16; ++i_) asm volatile("" : "+v"(cf[i_]));
  put_w(0, wfirst);
  TSTAMP(2);
  const unsigned lofs = (unsigned)(4 * hh * t_stride + a0 + r);  // T[(32 k + 8 g4 + 4 hh + i) t_stride + atom]: a wave-uniform row base + this lane's 32-bit offset
  const char* __restrict__ wr = wl + 16 * lane;
  auto store2 = [&](const f32x16& p0, const f32x16& p1, int kp, int i_) {  // register i_ of a finished unit's two tiles
    float* __restrict__ tp_ = T + ((size_t)kp * 32 + 8 * (i_ >> 2) + (i_ & 3)) * t_stride;
    tp_[lofs] = (p0[i_] * isc[0]) * cf[i_];
    (tp_ + 32)[lofs] = (p1[i_] * isc[1]) * cf[i_];
  };
  // one unit: barrier (its weights are in buffer `buf`, the other buffer is free), request the next unit's quarter, 8 x (one pair of A
  // fragments -> six MFMAs; two registers of the unit before stored), write the quarter into the other buffer
  auto unit = [&](f32x16& c0, f32x16& c1, const f32x16& p0, const f32x16& p1, auto have_pend, int k, int buf) {
    __syncthreads();
    const Quarter wn = load_w(k + 1);
    if (active) {
      const char* __restrict__ wb = wr + buf * TT_WBUF;
      float4 a0f = *reinterpret_cast<const float4*>(wb), a1f = *reinterpret_cast<const float4*>(wb + 1024);
#pragma unroll
      for (int q_ = 0; q_ < 16; ++q_) { c0[q_] = 0.f; c1[q_] = 0.f; }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float4 n0f = a0f, n1f = a1f;
        if (q + 1 < 8) {
          n0f = *reinterpret_cast<const float4*>(wb + (2 * q + 2) * 1024);
          n1f = *reinterpret_cast<const float4*>(wb + (2 * q + 3) * 1024);
        }
        c0 = MFMA32H(a1f, xh[0][q], c0);
        c1 = MFMA32H(a1f, xh[1][q], c1);
        c0 = MFMA32H(a0f, xl[0][q], c0);
        c1 = MFMA32H(a0f, xl[1][q], c1);
        c0 = MFMA32H(a0f, xh[0][q], c0);
        c1 = MFMA32H(a0f, xh[1][q], c1);
        if constexpr (decltype(have_pend)::value) {
          store2(p0, p1, k - 1, 2 * q);
          store2(p0, p1, k - 1, 2 * q + 1);
        }
        a0f = n0f; a1f = n1f;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    put_w(buf ^ 1, wn);
  };
  f32x16 A0, A1, B0, B1;
  int k = k_lo;
  unit(A0, A1, B0, B1, std::false_type{}, k, 0);
  if (k_lo == k_hi - 1) TSTAMP(3);
  for (++k; k + 1 < k_hi; k += 2) {
    unit(B0, B1, A0, A1, std::true_type{}, k, 1);
    if (k - k_lo < 10) TSTAMP(3 + k - k_lo);
    unit(A0, A1, B0, B1, std::true_type{}, k + 1, 0);
    if (k + 1 - k_lo < 10) TSTAMP(3 + k + 1 - k_lo);
  }
  if (k < k_hi) {
    unit(B0, B1, A0, A1, std::true_type{}, k, 1);
    if (active) {
#pragma unroll
      for (int i_ = 0; i_ < 16; ++i_) store2(B0, B1, k, i_);
    }
  } else if (active) {
#pragma unroll
    for (int i_ = 0; i_ < 16; ++i_) store2(A0, A1, k - 1, i_);
  }
  TSTAMP(13);
  TSTAMP_RT(14);
}

void launch_tprod(const float* x, int XS, int n_atoms, int n_k, const float4* wt, const float4* wth, const float* gT, const float* cfT, float* T, int t_stride,
                  hipStream_t st, bool no_tprod_t) {
  const int tiles = (n_atoms + 31) / 32;
  // waves = tiles x kg runs of hidden units.  One wave per SIMD (1024 waves) is best while a wave's run stays below ~10 hidden units
  // (136 tiles: 18.0 us against 20.0 with 2048 waves); a wave walks its units one after the other behind a one-unit weight prefetch, so
  // with more tiles the runs are cut to ~9 units again, up to two waves per SIMD — the register budget of the kernel (33 x 256 atoms,
  // 264 tiles: 36.7 -> 29.7 us; 296 tiles: 29.8 -> 27.4)
  if (wth && t_stride > 0 && !no_tprod_t) {  // k_tprod_t: workgroups of 4 waves x 64 atoms, one per CU; the shortest runs of hidden units that fit
    const int rows = (n_atoms + 64 * TT_WAVES - 1) / (64 * TT_WAVES);
    const int kg_max = std::min(n_k, std::max(1, 256 / rows));
    const int units = (n_k + kg_max - 1) / kg_max, kg_t = (n_k + units - 1) / units;
    hipLaunchKernelGGL(k_tprod_t, dim3(rows * kg_t), dim3(64 * TT_WAVES), TT_LDS_BYTES, st, x, XS, n_atoms, n_k, kg_t, wth, gT, cfT, T, t_stride);
    return;
  }
  const int kg = std::min(n_k, std::max(1, std::min(std::max(1024 / tiles, 7), 2048 / tiles)));
  if (wth)
    hipLaunchKernelGGL(k_tprod_h, dim3((tiles * kg + TP_WAVES - 1) / TP_WAVES), dim3(64 * TP_WAVES), 0, st, x, XS, n_atoms, n_k, kg, wth, gT, cfT, T, t_stride);
  else
    hipLaunchKernelGGL(k_tprod, dim3((tiles * kg + TP_WAVES - 1) / TP_WAVES), dim3(64 * TP_WAVES), 0, st, x, XS, n_atoms, n_k, kg, wt, T);
}

size_t conv_dg_lds_bytes(int rs, int pmax, int mode, int emu) { return sizeof(float) * ((dg_lds_floats(rs, pmax, mode, emu != 0) + 3) & ~(size_t)3); }

int launch_conv_dg(const DgArgs& a, int grid, hipStream_t st) {
  const size_t smem = conv_dg_lds_bytes(a.RS, a.PMAX, a.alt, a.emu);
  if (smem > JAMUN_MAX_DYN_LDS) return -2;
  if (a.RS > (a.alt == 1 ? DG_RS_MAX_ALT : DG_RS_MAX_RES) || a.XS != 216 || a.nt0 != 5) return -1;
  if (a.emu) {
    if (a.alt == 3) hipLaunchKernelGGL((k_conv_dg<3, true>), dim3(grid), dim3(DG_THREADS), smem, st, a);
    else if (a.alt == 1) hipLaunchKernelGGL((k_conv_dg<1, true>), dim3(grid), dim3(DG_THREADS), smem, st, a);
    else if (a.alt == 2) hipLaunchKernelGGL((k_conv_dg<2, true>), dim3(grid), dim3(DG_THREADS), smem, st, a);
    else hipLaunchKernelGGL((k_conv_dg<0, true>), dim3(grid), dim3(DG_THREADS), smem, st, a);
  } else {
    if (a.alt == 3) hipLaunchKernelGGL((k_conv_dg<3, false>), dim3(grid), dim3(DG_THREADS), smem, st, a);
    else if (a.alt == 1) hipLaunchKernelGGL((k_conv_dg<1, false>), dim3(grid), dim3(DG_THREADS), smem, st, a);
    else if (a.alt == 2) hipLaunchKernelGGL((k_conv_dg<2, false>), dim3(grid), dim3(DG_THREADS), smem, st, a);
    else hipLaunchKernelGGL((k_conv_dg<0, false>), dim3(grid), dim3(DG_THREADS), smem, st, a);
  }
  return 0;
}

int conv_dg_set_max_lds() {
  const void* fns[8] = {(const void*)k_conv_dg<0, false>, (const void*)k_conv_dg<1, false>, (const void*)k_conv_dg<2, false>, (const void*)k_conv_dg<3, false>,
                        (const void*)k_conv_dg<0, true>,  (const void*)k_conv_dg<1, true>,  (const void*)k_conv_dg<2, true>,  (const void*)k_conv_dg<3, true>};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  if (hipFuncSetAttribute((const void*)k_tprod_t, hipFuncAttributeMaxDynamicSharedMemorySize, TT_LDS_BYTES) != hipSuccess) return -1;
  return 0;
}
