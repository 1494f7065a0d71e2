// jamun_conv_ml.hip — the matrix-formed conv contraction (jamun_conv_mf.hip) for LARGE source spans: molecules of 63 .. 174 atoms
// (chignolin with / without hydrogens, 5AA peptides; BASELINE configs[4]), whose feature rows do not fit the LDS next to the coefficient tiles.
//
// Same mathematics (src/jamun/e3tools/nn/_conv.py:93-119, re-associated by destination), same weight stream (MfArgs::wm), same f16x3
// arithmetic and scales as k_conv_mf:
//   A_k[(i,c)][u] = sum_j C_k[c][i][j] x_j[u]   (dense GEMMs over the source rows of the tile's span, on the matrix cores),
//   m[(i,c)][w]   = sum_k sum_u A_k[(i,c)][u] W~[(k,u)][w]   (the formed tile chained in registers into the contraction).
// What is different:
//  * TWO PASSES over the segment's hidden units.  The contraction is K-split over the input channels and its accumulators live in
//    registers, so the k loop can run twice with a different operand set resident:
//      pass V   x1^T (96 rows) + all four coefficient components:  dot(x1, v) -> scalar outputs (wave 4);  x1[m], (x1 x v)[m] -> vector
//               plane m (waves 5 + m);  the T term  out_m += C[v_m] T_k  (waves 1 + m, T_k straight from global memory: no T tile);
//               waves 0..3 build the coefficient tiles
//      pass S   x0^T (128 rows) + component 0, double buffered:  scalar channels 32 w .. -> the five scalar-output tiles (waves w < 4);
//               waves 4..7 build
//    176 rows of x^T alone would be 165 KB; per pass 65 - 94 KB.
//  * BLOCK-SPARSE forming.  torch_cluster's radius graph keeps the FIRST 32 hits in index order (denoiser.py:149), so above ~40 atoms the
//    sources of a 32-destination tile cluster in index space: on a 166-atom chain half of the (tile, 16-row source block) pairs carry
//    no edge at all.  The builder lanes OR the blocks their edges touch into a mask (segment prologue); every forming product then walks
//    the OCCUPIED blocks only (compile-time unrolled list positions, uniform branches).  Empty blocks would multiply zeros.
//  * Rows: x^T planes ROWX bytes per row, coefficient planes ROWC = 32 NKS + 16 (both odd multiples of 16 B: conflict-free 16-byte
//    fragment reads).  HALF: the span has 16 NKS - 8 rows and ROWX = 32 NKS - 16 — the last block's upper half reads the first bytes of the
//    NEXT x^T row (finite) against coefficient columns that are never written (zero): 168 rows fit where 176 do not.
//  * Pass V is single buffered when two sets of four coefficient components do not fit (DBV = false: forming | barrier | split, contraction
//    and build | barrier), double buffered otherwise (one barrier per hidden unit, as k_conv_mf).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdio.h>
#include <type_traits>

#include "jamun_internal.h"
#include "jamun_mf_dev.h"

#define ML_THREADS 512
#if defined(ML_EXP)  // timing experiments, compile-time (-DML_EXP=bits; results are wrong): 1 no T term, 2 no coefficient build in the loops, 4 no forming
                     // MFMAs, 8 no pass S loop, 16 no pass V loop, 32 no contraction MFMAs
constexpr int ml_dbg = ML_EXP;
#else
constexpr int ml_dbg = 0;
#endif

#ifdef ML_TRACE  // per-wave timeline of workgroup 7 (diagnostic builds): [wave][pass V, S][k-step][stamp], [wave][segment][stamp]
__device__ unsigned long long g_mltrace[8][2][24][6];
__device__ unsigned long long g_mlseg[8][3][10];
#define MLSTAMP(pass, slot) do { if (blockIdx.x == 7 && sgi == 0 && it < 24) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_mltrace[wave][pass][it][slot] = t_; } } while (0)
#define MLSEG(slot) do { if (blockIdx.x == 7 && sgi < 3) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_mlseg[wave][sgi][slot] = t_; } } while (0)
#else
#define MLSTAMP(pass, slot) do { } while (0)
#define MLSEG(slot) do { } while (0)
#endif

namespace {

template <int NKS, bool HALF, bool DBV>
struct MlGeo {
  static constexpr int RH = 16 * NKS - (HALF ? 8 : 0);       // source rows of the window
  static constexpr int ROWX = HALF ? 32 * NKS - 16 : 32 * NKS + 16;
  static constexpr int ROWC = 32 * NKS + 16;
  static constexpr int PL = 32 * ROWC;                       // hi -> lo plane of a coefficient component
  static constexpr int CC = 2 * PL;                          // one component
  // pass V: x1^T hi [3 m][32 u] | lo | coefficient tiles [DBV ? 2 : 1][4 components][hi, lo][32 i]
  static constexpr int V_X1H = 0, V_X1L = 96 * ROWX, V_C = 2 * 96 * ROWX, V_CB = 4 * CC;
  static constexpr int V_HB = V_C + (DBV ? 2 : 1) * V_CB;  // hand-over of the split dot tile (wave 4 -> wave 0): [s2][hi, lo][64 lanes] x 16 B
  static constexpr int V_END = V_HB + (DBV ? 2 : 1) * 4096;  // (double buffered: no barrier separates wave 0's read of unit k - 1 from wave 4's write of unit k)
  // pass S: x0^T hi [128 u] | lo | component 0 [2 buffers][hi, lo][32 i]
  static constexpr int S_X0H = 0, S_X0L = 128 * ROWX, S_C = 2 * 128 * ROWX, S_CB = CC;
  static constexpr int S_END = S_C + 2 * S_CB;
  static constexpr int EP_END = 5 * 32 * 160 * 4 + 2 * 32 * 96 * 4;  // segment-end staging tiles (alias everything)
  static constexpr int M1 = V_END > S_END ? V_END : S_END;
  static constexpr int MISC = ((M1 > EP_END ? M1 : EP_END) + 15) & ~15;  // deg[32] | xmax | mask
  static constexpr int LDS_BYTES = MISC + 160;
};

}  // namespace

// X1: the opt-in reduced-precision mode (jamun_tuning.f16x1): every product is its hi hi MFMA alone
template <int NKS, bool HALF, bool DBV, bool X1>
__global__ __launch_bounds__(ML_THREADS) void k_conv_ml(MlArgs a) {
  using G = MlGeo<NKS, HALF, DBV>;
  constexpr int ROWX = G::ROWX, ROWC = G::ROWC, PL = G::PL, CC = G::CC, RH = G::RH;
  extern __shared__ float4 lds4[];
  char* __restrict__ lds = reinterpret_cast<char*>(lds4);
  const int L0 = lds_addr(lds);
  int* __restrict__ deg_lds = reinterpret_cast<int*>(lds + G::MISC);
  unsigned* __restrict__ xmax_lds = reinterpret_cast<unsigned*>(lds + G::MISC + 128);
  unsigned* __restrict__ mask_lds = reinterpret_cast<unsigned*>(lds + G::MISC + 132);

  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = RFL(tid0 >> 6);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float m1 = opaque_minus_one();  // (jamun_mf_dev.h: the split primitives as compiler-visible instructions)

  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int4 sg0 = ld_const(a.segs + ((size_t)blockIdx.x * a.max_segs + sgi) * 2);
    const int4 sg1 = ld_const(a.segs + ((size_t)blockIdx.x * a.max_segs + sgi) * 2 + 1);
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    int lane = lane0;
    asm volatile("" : "+v"(lane));  // (keeps lane-derived addresses from being hoisted out of the segment loop and spilled)
    const int tid = wave * 64 + lane;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) __attribute__((always_inline)) { kk = kk < 0 ? 0 : (kk < nk ? kk : nk - 1); return kk < k_run ? k_begin + kk : k_extra; };
    const int2 t_at = make_int2(sg1.y, sg1.z & 255), span = make_int2(sg1.w, sg1.w + (sg1.z >> 8));  // (descriptor inside the segment record, jamun_api.cpp: embed)
    const int n0 = RFL(t_at.x), n_dst = RFL(t_at.y);
    const int s_lo = RFL(span.x), rows = RFL(span.y) - s_lo;
    const int s_base = s_lo & ~1, off = s_lo - s_base;  // window from an even atom (8-byte T loads, row pairs)
    const int r = lane & 31, hh = lane >> 5;
    MLSEG(0);

    // ---- builder layout (both halves of the workgroup hold it: waves 0..3 build in pass V, waves 4..7 in pass S): four passes of 8
    // destinations x 32 slots over 256 lanes + ONE pass for slots 32..39 of all 32 destinations (k_geom caps the radial neighbours at 32: those
    // slots hold bonded edges only); lane = (q-th destination of this wave, slot - 32) keeps a destination's slots inside one wave
    constexpr int BT = 256, SPL = 32, LG = 5, DPP = BT / SPL, NPM = 32 / DPP, NP = NPM + 1, GPW = 64 / SPL;
    const int bt = tid & (BT - 1), bw = bt >> 6;
    auto dst_of = [&](int p) __attribute__((always_inline)) { return p == NPM ? DPP * ((lane >> 3) / GPW) + GPW * bw + ((lane >> 3) % GPW) : (bt + BT * p) / SPL; };
    auto slt_of = [&](int p) __attribute__((always_inline)) { return p == NPM ? SPL + (lane & 7) : (bt + BT * p) % SPL; };
    const int slot0 = (n0 + dst_of(0)) * a.S + slt_of(0), pstride = DPP * a.S, slot_x = (n0 + dst_of(NPM)) * a.S + slt_of(NPM);
    auto slot_of = [&](int p) __attribute__((always_inline)) { return p == NPM ? slot_x : slot0 + p * pstride; };
    const int dgl = a.deg[n0 + min(tid & 31, n_dst - 1)];
    const int dgv = dgl * ((tid < 32 && tid < n_dst) ? 1 : 0);  // (a product, not a select: a select's load is sunk into a branch)
    float hv[NP];
    auto load_h = [&](int k) __attribute__((always_inline)) {  // h~ of this lane's edge slots of hidden unit k (lanes past the in-degree read a neighbouring slot: never used)
      const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
      for (int p = 0; p < NP; ++p) hv[p] = hk[slot_of(p)];
    };
    load_h(k_of(0));
    float evx[NP], evy[NP], evz[NP];
    int sjv[NP], epv[NP], ent[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = dst_of(p), t = slt_of(p);
      const bool in = i < n_dst && t < a.S;
      // (unconditional requests at a clamped slot — a destination of the tile, a slot below the stride — masked after: as `in ? a.esrc[slot] : 0`
      // every load was a branch and some of the branches carried an `s_waitcnt vmcnt(0)`, round 6)
      const int sc_ = (n0 + min(i, n_dst - 1)) * a.S + min(t, a.S - 1);
      const int sj_ = a.esrc[sc_], ep_ = a.epair[sc_];
      const float4 ge = a.egeo[sc_];
      const float kin = in ? 1.f : 0.f;
      sjv[p] = in ? sj_ : 0;
      epv[p] = in ? ep_ : 0;
      evx[p] = ge.x * kin; evy[p] = ge.y * kin; evz[p] = ge.z * kin;
    }

    // ---- prologue 1: zero the coefficient tiles of pass V; largest |x g| over the span's rows (all 216 channels: ONE scale 2^sX for both passes)
    for (int idx = tid; idx < (DBV ? 2 : 1) * G::V_CB / 16; idx += ML_THREADS) reinterpret_cast<float4*>(lds + G::V_C)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == 0) { *xmax_lds = 0u; *mask_lds = 0u; }
    {
      // thread = (float4 column c4 = tid % 54, row group tid / 54 of 9): the channel factors once per thread, five row loads in flight
      float mx = 0.f;
      const int c4 = tid % 54, jg = tid / 54;
      if (tid < 486) {
        const float4 g4 = reinterpret_cast<const float4*>(a.gx)[c4];
        const float4* __restrict__ px = reinterpret_cast<const float4*>(a.x + (size_t)s_lo * a.XS) + c4;
        for (int j0 = jg; j0 < rows; j0 += 45) {
          float4 v[5];
#pragma unroll
          for (int q = 0; q < 5; ++q) v[q] = px[(size_t)min(j0 + 9 * q, rows - 1) * (a.XS / 4)];  // (clamped: a repeated row does not change the maximum; unconditional requests)
#pragma unroll
          for (int q = 0; q < 5; ++q) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[q].x * g4.x), fabsf(v[q].y * g4.y)), fmaxf(fabsf(v[q].z * g4.z), fabsf(v[q].w * g4.w))));
        }
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      LDS_BARRIER();  // xmax / mask zeroed
      if (lane == 0) atomicMax(xmax_lds, __float_as_uint(mx));
    }
    if (tid < 32) deg_lds[tid] = dgv;
    LDS_BARRIER();
    MLSEG(1);
    const float xm = __uint_as_float(*xmax_lds);
    const int sX = RFL(xm > 0.f ? clamp40(14 - exp_above(xm)) : 0);

    // ---- builder state: entry offsets, pair partners (k_geom's pair table: MfArgs::epair), the mask of occupied 16-row source blocks
    {
      unsigned bm = 0u;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int i = dst_of(p), t = slt_of(p);
        const int dg = deg_lds[i];
        const bool in = t < dg && t < a.S;
        const int jl = (sjv[p] & 0x7fffffff) - s_base;  // (bit 31: bonded)
        const bool valid = in && jl >= 0 && jl < RH;
        if (in && (jl < 0 || jl >= RH)) atomicOr(a.err, 2);  // a source outside the window: the host's plan is wrong
        const int ep = in ? epv[p] : 0;
        if (ep & JAMUN_EP_OVERFLOW) atomicOr(a.err, 1);
        const int pa = (ep & 127) - 1, pb = ((ep >> 7) & 127) - 1;  // slots of the pair's other edges (-1: none); always behind this one
        auto ref = [&](int ps) __attribute__((always_inline)) {  // where this lane finds the h~ of slot ps of its destination: lane distance, or (| 64) a lane of the last pass
          if (ps < 0) return 0;
          if (p < NPM && ps >= SPL) return (((GPW * p + (lane >> LG)) << 3) + (ps - SPL)) | 64;
          return ps - t;
        };
        const int r0 = ref(pa), r1 = ref(pb);
        const bool active = valid && !((unsigned)ep & JAMUN_EP_OWNED);
        // bits 0..13 byte offset inside a coefficient plane, 14..19 / 20..25 the pair's other edges, 26 / 27: they sit in the last pass
        ent[p] = active ? (i * ROWC + 2 * jl) | ((r0 & 63) << 14) | ((r1 & 63) << 20) | ((r0 >> 6) << 26) | ((r1 >> 6) << 27) : (ROWC - 16);
        if (active) bm |= 1u << (jl >> 4);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) bm |= (unsigned)__shfl_xor((int)bm, o, 64);
      if (lane == 0 && wave < 4) atomicOr(mask_lds, bm);
    }
    const float scC = pow2f(a.sC), scx = pow2f(sX), scT = pow2f(clamp100(sX + a.sTw));
    auto coef = [&](int p) __attribute__((always_inline)) {  // h~ of pass p's entry: this lane's edge + the pair's other edges
      const int d0 = (ent[p] >> 14) & 63, d1 = (ent[p] >> 20) & 63;
      float t0 = __shfl(hv[p], lane + d0, 64), t1 = __shfl(hv[p], lane + d1, 64);
      bool u0 = d0 != 0, u1 = d1 != 0;
      if (p < NPM) {  // (the other edge in the extra pass: absolute lane)
        const float q0 = __shfl(hv[NP - 1], d0, 64), q1 = __shfl(hv[NP - 1], d1, 64);
        if ((ent[p] >> 26) & 1) { t0 = q0; u0 = true; }
        if ((ent[p] >> 27) & 1) { t1 = q1; u1 = true; }
      }
      return (hv[p] + (u0 ? t0 : 0.f)) + (u1 ? t1 : 0.f);
    };
    // all four components (pass V), in two halves: prep4 computes the entries of the NEXT hidden unit into registers (coefficient x geometry, scaled,
    // split: four packed words per builder pass) — vector work that runs beside the other waves' forming MFMAs, while the single coefficient
    // buffer is still being read — and flush4 stores them behind the barrier that ends the forming phase: sixteen-bit LDS stores only, so the
    // phase in which the matrix pipes idle is short ("double buffering in registers": two LDS buffers of four components do not fit)
    // PM: the builder passes this wave computes and stores — both halves of the workgroup hold the same lane layout (waves w and w + 4), so
    // the work is shared: waves 1..3, which also carry the T term, take passes 0, 1 and the extra pass; waves 5..7 take passes 2, 3 behind
    // their forming; waves 0 and 4 take their share the same way
    unsigned stg[NP][4];
    auto prep4 = [&](auto pm_) __attribute__((always_inline)) {
      constexpr int PM = decltype(pm_)::value;
      float cc[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p)
        if ((PM >> p) & 1) cc[p] = coef(p) * scC;  // (all passes' lane shuffles in flight before the first is consumed)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (!((PM >> p) & 1)) continue;
        const float c0 = cc[p], c1 = c0 * evx[p], c2 = c0 * evy[p], c3 = c0 * evz[p];
        const unsigned h01 = cvt_pk_f16_c(c0, c1), h23 = cvt_pk_f16_c(c2, c3);
        stg[p][0] = h01; stg[p][1] = h23;
        stg[p][2] = cvt_pk_f16_c(resid_lo_c(c0, h01, m1), resid_hi_c(c1, h01, m1));
        stg[p][3] = cvt_pk_f16_c(resid_lo_c(c2, h23, m1), resid_hi_c(c3, h23, m1));
      }
    };
    auto flush4 = [&](char* __restrict__ cbuf, auto pm_) __attribute__((always_inline)) {
      constexpr int PM = decltype(pm_)::value;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (!((PM >> p) & 1)) continue;
        const unsigned h01 = stg[p][0], h23 = stg[p][1], l01 = stg[p][2], l23 = stg[p][3];
        char* __restrict__ d = cbuf + (ent[p] & 0x3fff);
        *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h01 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + PL) = (unsigned short)(l01 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 2 * PL) = (unsigned short)(h01 >> 16);
        *reinterpret_cast<unsigned short*>(d + 3 * PL) = (unsigned short)(l01 >> 16);
        *reinterpret_cast<unsigned short*>(d + 4 * PL) = (unsigned short)(h23 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 5 * PL) = (unsigned short)(l23 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 6 * PL) = (unsigned short)(h23 >> 16);
        *reinterpret_cast<unsigned short*>(d + 7 * PL) = (unsigned short)(l23 >> 16);
      }
    };
    const std::integral_constant<int, 0x13> PM_LO{};  // passes 0, 1, 4
    const std::integral_constant<int, 0x0c> PM_HI{};  // passes 2, 3
    auto build1 = [&](char* __restrict__ cbuf) __attribute__((always_inline)) {  // component 0 only (pass S)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float c0 = coef(p) * scC;
        const unsigned h0 = cvt_pk_f16_c(c0, 0.f);
        const unsigned l0 = cvt_pk_f16_c(resid_lo_c(c0, h0, m1), 0.f);
        char* __restrict__ d = cbuf + (ent[p] & 0x3fff);
        *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h0 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + PL) = (unsigned short)(l0 & 0xffffu);
      }
    };
    // ---- staging of the span's rows, transposed and split: lane = row pair (coalesced LDS words), one float4 column per (wave, round)
    //   first channel float4 column c0 (30: the vector block), NC columns, destination planes by the row of each channel
    auto stage = [&](auto row_of, auto c0_, auto nc_, int lo_off) __attribute__((always_inline)) {
      constexpr int c0 = decltype(c0_)::value, NC = decltype(nc_)::value;
      constexpr int NPAIR = RH / 2, NR = (NPAIR + 63) / 64, NIT = (NC * NR + 7) / 8;
      float4 va[NIT], vb[NIT], gq[NIT];
#pragma unroll
      for (int q = 0; q < NIT; ++q) {  // (every round's rows requested before the first is split and stored)
        const int cw = wave + 8 * q, c4 = c0 + cw % NC, jp = 64 * (cw / NC) + lane;
        const int j0 = 2 * jp - off, j1 = j0 + 1;  // rows relative to the span
        // (the channel factors with the rows: as loads at their use, under the round's predicate, each was a request and an `s_waitcnt vmcnt(0)` of
        // its own — up to four serialised round trips per staging pass, round 6)
        gq[q] = reinterpret_cast<const float4*>(a.gx)[c4];
        va[q] = vb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cw < NC * NR && jp < NPAIR) {
          if (j0 >= 0 && j0 < rows) va[q] = reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + j0) * a.XS)[c4];
          if (j1 >= 0 && j1 < rows) vb[q] = reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + j1) * a.XS)[c4];
        }
      }
#pragma unroll
      for (int q = 0; q < NIT; ++q) {
        const int cw = wave + 8 * q, c4 = c0 + cw % NC, jp = 64 * (cw / NC) + lane;
        if (cw < NC * NR && jp < NPAIR) {
          const float4 g4 = gq[q];
          const float ea[4] = {va[q].x * g4.x, va[q].y * g4.y, va[q].z * g4.z, va[q].w * g4.w}, eb[4] = {vb[q].x * g4.x, vb[q].y * g4.y, vb[q].z * g4.z, vb[q].w * g4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int rowb = row_of(4 * c4 + e) * ROWX;
            const float a0 = ea[e] * scx, b0 = eb[e] * scx;
            const unsigned ph = cvt_pk_f16_c(a0, b0), pl = cvt_pk_f16_c(resid_lo_c(a0, ph, m1), resid_hi_c(b0, ph, m1));
            *reinterpret_cast<unsigned*>(lds + rowb + 4 * jp) = ph;
            *reinterpret_cast<unsigned*>(lds + rowb + lo_off + 4 * jp) = pl;
          }
        }
      }
    };
    // pass V: vector element 120 + 3 u + m -> row (m, u) of the x1^T planes
    stage([](int ch) { const int e1 = ch - 120, u = e1 / 3; return (e1 - 3 * u) * 32 + u; }, std::integral_constant<int, 30>{}, std::integral_constant<int, 24>{}, G::V_X1L - G::V_X1H);
    MLSEG(2);
    LDS_BARRIER();  // mask, x1^T complete
    MLSEG(3);

    // ---- the occupied source blocks of this tile: a list of 4-bit block indices in ONE 64-bit scalar (position i at bits 4 i ..); the forming
    // loops are run-time loops over the list (positions past the end clamp to the last entry: prefetches need no predicate)
    unsigned long long blist = 0ull;
    int nb = 0;
    {
      unsigned m = (unsigned)RFL((int)*mask_lds);
      while (m) {
        blist |= (unsigned long long)__builtin_ctz(m) << (4 * nb);
        ++nb;
        m &= m - 1u;
      }
      nb = RFL(nb);
    }
    auto boff = [&](int i) __attribute__((always_inline)) {  // byte offset of list position i inside a row of halves
      const int ic = min(i, nb - 1);
      return nb > 0 ? (int)((blist >> (4 * ic)) & 15ull) << 5 : 0;
    };
    if (tid == 0 && a.mfma_count) atomicAdd(a.mfma_count, (unsigned long long)nk * (unsigned long long)((X1 ? 1 : 3) * (19 * nb + 62)));

    // ---- common matrix-wave pieces
    const int edeg_r = deg_lds[r] > 0 ? exp_above((float)deg_lds[r]) : 1;
    const float rs = pow2f(-14 - edeg_r);  // formed values are below in-degree x 2^28 -> below 2^14
    const int fox = r * ROWX + 16 * hh + L0, foc = r * ROWC + 16 * hh + L0;  // this lane's fragment offsets inside a 32-row x^T / coefficient plane
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.wm), 0, 0x7fffffff, 0x00020000);
    const int wvo = lane * 16;
    auto wload = [&](int so) __attribute__((always_inline)) { return __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so, 0); };
    struct Frag { float4 ah, al, bh, bl; };
    auto ldf = [&](int xa, int xlo, int ca, int clo) __attribute__((always_inline)) -> Frag { return Frag{lds_f4(xa), lds_f4(xa + xlo), lds_f4(ca), lds_f4(ca + clo)}; };
    auto mm = [&](f32x16& F, const Frag& f) __attribute__((always_inline)) { MX(F, f.ah, f.al, f.bh, f.bl); };
    auto split16 = [&](const f32x16& F, float4 (&Ah)[2], float4 (&Al)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        unsigned ph[4], pl[4];
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          const float v0 = F[8 * s2 + 2 * p2] * rs, v1 = F[8 * s2 + 2 * p2 + 1] * rs;
          ph[p2] = cvt_pk_f16_c(v0, v1);
          pl[p2] = cvt_pk_f16_c(resid_lo_c(v0, ph[p2], m1), resid_hi_c(v1, ph[p2], m1));
        }
        Ah[s2] = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
        Al[s2] = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
      }
    };
    // Each role keeps its state inside its own branch (a value that lives across a branch of another role would hold its registers
    // there too); the roles meet at the barriers only: V0 | per hidden unit of pass V: (A unless DBV) B | S0 S1 | per hidden unit of pass S: one
    constexpr int NB20 = 20, RD = 10;
    // scalar-output contraction of one formed tile: 20 weight blocks per k ((hi, lo) per (output tile n, K-step s2)), ring half a unit deep
    auto contract5 = [&](f32x16 (&accS)[5], u32x4 (&RB)[RD], const float4 (&Ah)[2], const float4 (&Al)[2], int cur, int nxt) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int p = 2 * (2 * n + s2);
          MX(accS[n], Ah[s2], Al[s2], RB[p % RD], RB[(p + 1) % RD]);
#pragma unroll
          for (int e = 0; e < 2; ++e) RB[(p + e) % RD] = (p + e + RD < NB20) ? wload(cur + (p + e + RD) * 1024) : wload(nxt + (p + e + RD - NB20) * 1024);
          MF_SCHED();  // (pins the ring: left alone, the scheduler sinks each load to just before its use)
        }
    };
    // between the passes (all threads; every wave left pass V through its last barrier: nothing reads the x1^T planes or the coefficient
    // tiles any more): zero component 0's two buffers, stage x0^T
    auto prep_S = [&]() __attribute__((always_inline)) {
      for (int idx = tid; idx < 2 * G::S_CB / 16; idx += ML_THREADS) reinterpret_cast<float4*>(lds + G::S_C)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tid < 2 * 8 * (ROWX / 16)) {  // channel rows 120..127 of the scalar block (inputs of the last K-step that do not exist)
        const int pl = tid / (8 * (ROWX / 16)), q = tid - pl * 8 * (ROWX / 16);
        *reinterpret_cast<float4*>(lds + (pl ? G::S_X0L : G::S_X0H) + 120 * ROWX + 16 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      stage([](int ch) { return ch; }, std::integral_constant<int, 0>{}, std::integral_constant<int, 30>{}, G::S_X0L - G::S_X0H);
    };
    float* __restrict__ ST0 = reinterpret_cast<float*>(lds);                     // [5 waves][32][160] partial scalar-output tiles (segment end)
    float* __restrict__ ST1 = reinterpret_cast<float*>(lds + 5 * 32 * 160 * 4);  // [2][32][96] vector planes: contraction, T term
    auto cbuf_rd = [&](int it) __attribute__((always_inline)) { return G::V_C + (DBV ? (it & 1) * G::V_CB : 0); };
    auto cbuf_wr = [&](int it) __attribute__((always_inline)) { return G::V_C + (DBV ? ((it + 1) & 1) * G::V_CB : 0); };

    // pass S of a scalar wave (waves 0..3): channels 32 wave .. -> the five scalar-output tiles; barriers S0, S1, one per hidden unit
    auto pass_S_scalar = [&](f32x16 (&accS)[5]) __attribute__((always_inline)) {
      u32x4 RB[RD];
      auto wstreamS = [&](int k) __attribute__((always_inline)) { return (k * 124 + 20 * wave) * 1024; };
      {
        const int c0 = wstreamS(k_of(0));
#pragma unroll
        for (int p = 0; p < RD; ++p) RB[p] = wload(c0 + p * 1024);
      }
      LDS_BARRIER();  // S0: x0^T complete, tiles zeroed
      LDS_BARRIER();  // S1: C0(k0) complete
      MLSEG(5);
      // the formed tile of hidden unit k is split behind its forming and CONTRACTED at the start of unit k + 1 (registers and weights only),
      // behind that unit's first fragment reads: the contraction's MFMAs cover the LDS round trip the forming chain starts with
      float4 sAh[2], sAl[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) sAh[s2] = sAl[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int it = 0; it < ((ml_dbg & 8) ? 0 : nk); ++it) {
        int fox_ = fox, foc_ = foc;
        asm volatile("" : "+v"(fox_), "+v"(foc_));
        const int cb = G::S_C + (it & 1) * G::S_CB + foc_, xa = G::S_X0H + 32 * wave * ROWX + fox_;
        MLSTAMP(1, 0);
        f32x16 F = zero16;
        auto fr = [&](int i) __attribute__((always_inline)) { const int bo = boff(i); return ldf(xa + bo, G::S_X0L - G::S_X0H, cb + bo, PL); };
        Frag fA = fr(0), fB = fr(1);
        MF_SCHED();
        if (it > 0 && !(ml_dbg & 32)) contract5(accS, RB, sAh, sAl, wstreamS(k_of(it - 1)), wstreamS(k_of(it)));
        MLSTAMP(1, 1);
        for (int i = 0; i < nb; i += 2) {  // two blocks per round, the reads one round ahead
          MF_SCHED();
          if (!(ml_dbg & 4)) mm(F, fA);
          MF_SCHED();
          fA = fr(i + 2);
          MF_SCHED();
          if (i + 1 < nb && !(ml_dbg & 4)) mm(F, fB);
          MF_SCHED();
          fB = fr(i + 3);
        }
        MLSTAMP(1, 2);
        split16(F, sAh, sAl);
        MF_SCHED();
        MLSTAMP(1, 3);
        LDS_BARRIER();
        MLSTAMP(1, 4);
      }
      if (!(ml_dbg & (8 | 32))) contract5(accS, RB, sAh, sAl, wstreamS(k_of(nk - 1)), wstreamS(k_of(nk - 1)));
      MLSEG(6);
    };
    // pass S of a builder wave (waves 4..7)
    auto pass_S_builder = [&]() __attribute__((always_inline)) {
      prep_S();
      load_h(k_of(0));
      LDS_BARRIER();  // S0
      build1(lds + G::S_C);
      load_h(k_of(1));
      LDS_BARRIER();  // S1
      for (int it = 0; it < ((ml_dbg & 8) ? 0 : nk); ++it) {
        MLSTAMP(1, 0);
        if (!(ml_dbg & 2)) build1(lds + G::S_C + ((it + 1) & 1) * G::S_CB);
        load_h(k_of(it + 2));
        MF_SCHED();
        MLSTAMP(1, 3);
        LDS_BARRIER();
        MLSTAMP(1, 4);
      }
    };
    const int hb = G::V_HB + 16 * lane + L0;  // this lane's slot of the hand-over buffer (+ 1024 per fragment: Ah[0], Al[0], Ah[1], Al[1])

    if (wave == 0) {
      // =================================================================================================================
      // wave 0.  Pass V: contracts the dot(x1, v) tile wave 4 formed one hidden unit earlier (handed over split, through LDS) into ITS
      //          five scalar-output tiles — partial sums of the same outputs as its own channels in pass S: one set of accumulators —
      //          while the other waves form; then builds the next unit's coefficient tiles.  Pass S: scalar channels 0..31.
      // =================================================================================================================
      f32x16 accS[5];
#pragma unroll
      for (int n = 0; n < 5; ++n) accS[n] = zero16;
      {
        u32x4 RV[RD];  // the dot stream: the ring of contract5, half a hidden unit ahead
        auto wstreamD = [&](int k) __attribute__((always_inline)) { return (k * 124 + 80) * 1024; };
        auto contract_dot = [&](int cur, int nxt, int hbk) __attribute__((always_inline)) {
          float4 Ah[2], Al[2];
          Ah[0] = lds_f4(hbk); Al[0] = lds_f4(hbk + 1024); Ah[1] = lds_f4(hbk + 2048); Al[1] = lds_f4(hbk + 3072);
          if (!(ml_dbg & 32)) contract5(accS, RV, Ah, Al, cur, nxt);
        };
        {
          const int c0 = wstreamD(k_of(0));
#pragma unroll
          for (int p = 0; p < RD; ++p) RV[p] = wload(c0 + p * 1024);
        }
        prep4(PM_LO);
        flush4(lds + G::V_C, PM_LO);
        load_h(k_of(1));
        LDS_BARRIER();  // V0: C(k0) complete
        for (int it = 0; it < ((ml_dbg & 16) ? 0 : nk); ++it) {
          MLSTAMP(0, 0);
          if (it > 0) contract_dot(wstreamD(k_of(it - 1)), wstreamD(k_of(it)), hb + (DBV ? ((it - 1) & 1) * 4096 : 0));  // the tile of hidden unit it - 1; the ring moves on to unit it
          if (!(ml_dbg & 2)) prep4(PM_LO);  // (on the last hidden unit: from clamped loads, never read)
          load_h(k_of(it + 2));        // (behind the entries computed from unit it + 1's: the phase between the barriers is stores only)
          MLSTAMP(0, 1);
          if (!DBV) LDS_BARRIER();  // A: every read of C(k) done
          MLSTAMP(0, 2);
          if (!(ml_dbg & 2)) flush4(lds + cbuf_wr(it), PM_LO);
          MF_SCHED();
          MLSTAMP(0, 3);
          LDS_BARRIER();  // B: C(k + 1) complete, the dot tile of unit k handed over
          MLSTAMP(0, 4);
        }
        if (!(ml_dbg & 16)) contract_dot(wstreamD(k_of(nk - 1)), wstreamD(k_of(nk - 1)), hb + (DBV ? ((nk - 1) & 1) * 4096 : 0));  // the last unit's tile (the ring's reload is not used)
        MLSEG(4);
      }
      prep_S();
      pass_S_scalar(accS);
#pragma unroll
      for (int q = 0; q < 16; ++q) {  // (raw accumulators: the power-of-two factors are applied to the SUM of the partial tiles below)
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
          ST0[(0 * 32 + row) * 160 + 32 * n + r] = accS[n][q];
          ST0[(4 * 32 + row) * 160 + 32 * n + r] = 0.f;  // (the dot tile's slot of the five-way sum: already inside accS)
        }
      }
    } else if (wave < 4) {
      // =================================================================================================================
      // waves 1..3.  Pass V: the T term of plane wave - 1, then the coefficient tiles of the next hidden unit.
      //              Pass S: scalar channels 32 wave .. -> the five scalar-output tiles.
      // =================================================================================================================
      const int vm = wave - 1;
      f32x16 accT = zero16;
      {
        // T_k rows of the span (fp32, global, transposed by the pre-pass: MfArgs::Tt): lane (w' = r, hh) needs T[j][w'], j = 16 b + 8 hh .. + 7, of
        // every occupied block — two 16-byte buffer loads per block, a ring of four blocks in flight (the first four of a hidden unit are
        // requested during the previous unit's build)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const int tvo = (r * a.t_stride + s_base + 8 * hh) * 4;  // this lane's byte offset inside the [32 w'][t_stride] table of a hidden unit
        constexpr int TD = NKS < 8 ? NKS : 8;  // blocks of T in flight: all of a hidden unit's up to eight occupied blocks, requested behind the previous unit's build
        f32x4 tq[TD][2];
        auto t_rsrc = [&](int k) __attribute__((always_inline)) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Tt + (size_t)k * 32 * a.t_stride), 0, 0x7fffffff, 0x00020000); };
        auto load_T = [&](f32x4 (&t)[2], __amdgpu_buffer_rsrc_t trs, int i) __attribute__((always_inline)) {
          const int so = 2 * boff(i);
          t[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trs, tvo, so, 0));
          t[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(trs, tvo + 16, so, 0));
        };
        auto load_TD = [&](int k) __attribute__((always_inline)) {
          const __amdgpu_buffer_rsrc_t trs = t_rsrc(k);
#pragma unroll
          for (int q = 0; q < TD; ++q) load_T(tq[q], trs, q);
        };
        prep4(PM_LO);
        flush4(lds + G::V_C, PM_LO);
        load_h(k_of(1));
        load_TD(k_of(0));
        LDS_BARRIER();  // V0: C(k0) complete
        for (int it = 0; it < ((ml_dbg & 16) ? 0 : nk); ++it) {
          const int cb = cbuf_rd(it);
          int foc_ = foc;
          asm volatile("" : "+v"(foc_));  // (per-iteration copy: the block addresses are not hoisted out of the k loop and spilled)
          MLSTAMP(0, 0);
          if (!(ml_dbg & 1)) {
            // out_m[i][w'] += sum_j C[v_m][i][j] T_k[j][w']: A = coefficient rows (LDS), B = T_k^T rows, scaled and split here
            const __amdgpu_buffer_rsrc_t trs = t_rsrc(k_of(it));
            auto tstep = [&](f32x4 (&t)[2], int i) __attribute__((always_inline)) {
              const int ca = cb + (1 + vm) * CC + foc_ + boff(i);
              const float4 ah = lds_f4(ca), al = lds_f4(ca + PL);
              unsigned ph[4], pl[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float t0 = t[q >> 1][2 * (q & 1)] * scT, t1 = t[q >> 1][2 * (q & 1) + 1] * scT;
                ph[q] = cvt_pk_f16_c(t0, t1);
                pl[q] = cvt_pk_f16_c(resid_lo_c(t0, ph[q], m1), resid_hi_c(t1, ph[q], m1));
              }
              const float4 bh = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
              const float4 bl = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
              MX(accT, ah, al, bh, bl);
              if (NKS > TD && i + TD < nb) load_T(t, trs, i + TD);  // (more than eight occupied blocks: the ring refills as it drains)
            };
            for (int i = 0; i < nb; i += TD) {
#pragma unroll
              for (int q = 0; q < TD; ++q)
                if (i + q < nb) tstep(tq[q], i + q);
            }
          }
          load_TD(k_of(it + 1));  // (the ring is drained: the next unit's T rows are requested before this unit's barrier, a whole phase ahead of their use)
          if (!(ml_dbg & 2)) prep4(PM_LO);
          load_h(k_of(it + 2));
          MLSTAMP(0, 1);
          if (!DBV) LDS_BARRIER();  // A: every read of C(k) done
          MLSTAMP(0, 2);
          if (!(ml_dbg & 2)) flush4(lds + cbuf_wr(it), PM_LO);
          MF_SCHED();
          MLSTAMP(0, 3);
          LDS_BARRIER();  // B: C(k + 1) complete (double buffered: and every read of C(k) done)
          MLSTAMP(0, 4);
        }
        MLSEG(4);
      }
      prep_S();
      f32x16 accS[5];
#pragma unroll
      for (int n = 0; n < 5; ++n) accS[n] = zero16;
      pass_S_scalar(accS);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
#pragma unroll
        for (int n = 0; n < 5; ++n) ST0[(wave * 32 + row) * 160 + 32 * n + r] = accS[n][q];
        ST1[3072 + row * 96 + vm * 32 + r] = accT[q];
      }
    } else if (wave == 4) {
      // =================================================================================================================
      // wave 4.  Pass V: forms dot(x1, v) = sum_m x1[m] C[v_m] over the occupied blocks, splits it and hands it to wave 0.  Pass S: builder.
      // =================================================================================================================
      prep4(PM_HI);
      flush4(lds + G::V_C, PM_HI);
      load_h(k_of(1));
      LDS_BARRIER();  // V0
      for (int it = 0; it < ((ml_dbg & 16) ? 0 : nk); ++it) {
        const int cb = cbuf_rd(it);
        int fox_ = fox, foc_ = foc;
        asm volatile("" : "+v"(fox_), "+v"(foc_));
        MLSTAMP(0, 0);
        f32x16 F = zero16;
        auto fr = [&](int bo, int m) __attribute__((always_inline)) { return ldf(G::V_X1H + m * 32 * ROWX + fox_ + bo, G::V_X1L - G::V_X1H, cb + (1 + m) * CC + foc_ + bo, PL); };
        int bo = boff(0);
        Frag f0 = fr(bo, 0), f1 = fr(bo, 1), f2;
        for (int i = 0; i < nb; ++i) {  // (the reads run two products ahead of the MFMAs)
          const int bn = boff(i + 1);
          f2 = fr(bo, 2);
          MF_SCHED();
          if (!(ml_dbg & 4)) mm(F, f0);
          MF_SCHED();
          f0 = fr(bn, 0);
          MF_SCHED();
          if (!(ml_dbg & 4)) mm(F, f1);
          MF_SCHED();
          f1 = fr(bn, 1);
          MF_SCHED();
          if (!(ml_dbg & 4)) mm(F, f2);
          MF_SCHED();
          bo = bn;
        }
        if (!(ml_dbg & 2)) prep4(PM_HI);
        load_h(k_of(it + 2));
        MLSTAMP(0, 1);
        if (!DBV) LDS_BARRIER();  // A
        MLSTAMP(0, 2);
        if (!(ml_dbg & 2)) flush4(lds + cbuf_wr(it), PM_HI);
        float4 Ah[2], Al[2];
        split16(F, Ah, Al);
        const int hbk = hb + (DBV ? (it & 1) * 4096 : 0);
        lds_st4(hbk, Ah[0]); lds_st4(hbk + 1024, Al[0]); lds_st4(hbk + 2048, Ah[1]); lds_st4(hbk + 3072, Al[1]);
        MLSTAMP(0, 3);
        LDS_BARRIER();  // B
        MLSTAMP(0, 4);
      }
      pass_S_builder();
    } else {
      // =================================================================================================================
      // waves 5..7.  Pass V: vector plane vm: FA = x1[m] C[0];  F1 = x1[m+1] C[v_(m+2)], F2 = x1[m+2] C[v_(m+1)]  ((x1 x v)[m] = F1 - F2);
      //              both tiles are split behind the forming barrier and contracted into the plane at the START of the next unit's forming
      //              phase (registers and weights only), next to the first fragment reads.  Pass S: builder.
      // =================================================================================================================
      const int vm = wave - 5, m1 = (vm + 1) % 3, m2 = (vm + 2) % 3;
      f32x16 accP = zero16;
      {
        u32x4 RB[8];
        auto wstreamV = [&](int k) __attribute__((always_inline)) { return (k * 124 + 100 + 8 * vm) * 1024; };
        {
          const int c0 = wstreamV(k_of(0));
#pragma unroll
          for (int p = 0; p < 8; ++p) RB[p] = wload(c0 + p * 1024);
        }
        float4 sAh[2], sAl[2], sCh[2], sCl[2];  // the split tiles of the previous hidden unit: x1[m], cross
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) sAh[s2] = sAl[s2] = sCh[s2] = sCl[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
        auto contract_plane = [&](int nxt) __attribute__((always_inline)) {  // blocks 0..3: x1 inputs ((hi, lo) per K-step), 4..7: cross inputs; the ring moves on to the next unit
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int s2 = g & 1, p = 2 * g;
            if (!(ml_dbg & 32)) { MX(accP, (g < 2 ? sAh[s2] : sCh[s2]), (g < 2 ? sAl[s2] : sCl[s2]), RB[p], RB[p + 1]); }
            RB[p] = wload(nxt + p * 1024);
            RB[p + 1] = wload(nxt + (p + 1) * 1024);
            MF_SCHED();
          }
        };
        prep4(PM_HI);
        flush4(lds + G::V_C, PM_HI);
        load_h(k_of(1));
        LDS_BARRIER();  // V0
        f32x16 FA = zero16, F1 = zero16, F2 = zero16;
        auto split_plane = [&]() __attribute__((always_inline)) {
          split16(FA, sAh, sAl);
#pragma unroll
          for (int q = 0; q < 16; ++q) F1[q] -= F2[q];
          split16(F1, sCh, sCl);
        };
        for (int it = 0; it < ((ml_dbg & 16) ? 0 : nk); ++it) {
          const int cb = cbuf_rd(it);
          int fox_ = fox, foc_ = foc;
          asm volatile("" : "+v"(fox_), "+v"(foc_));
          MLSTAMP(0, 0);
          auto fr = [&](int bo, int g) __attribute__((always_inline)) {
            const int xm_ = g == 0 ? vm : g == 1 ? m1 : m2, cc = g == 0 ? 0 : g == 1 ? 1 + m2 : 1 + m1;
            return ldf(G::V_X1H + xm_ * 32 * ROWX + fox_ + bo, G::V_X1L - G::V_X1H, cb + cc * CC + foc_ + bo, PL);
          };
          int bo = boff(0);
          Frag f0 = fr(bo, 0), f1 = fr(bo, 1), f2;
          MF_SCHED();
          if (it > 0) {  // hidden unit it - 1: split and contracted behind the first fragment reads of unit it (registers and weights only)
            split_plane();
            contract_plane(wstreamV(k_of(it)));
          }
          FA = zero16; F1 = zero16; F2 = zero16;
          for (int i = 0; i < nb; ++i) {
            const int bn = boff(i + 1);
            f2 = fr(bo, 2);
            MF_SCHED();
            if (!(ml_dbg & 4)) mm(FA, f0);
            MF_SCHED();
            f0 = fr(bn, 0);
            MF_SCHED();
            if (!(ml_dbg & 4)) mm(F1, f1);
            MF_SCHED();
            f1 = fr(bn, 1);
            MF_SCHED();
            if (!(ml_dbg & 4)) mm(F2, f2);
            MF_SCHED();
            bo = bn;
          }
          if (!(ml_dbg & 2)) prep4(PM_HI);
          load_h(k_of(it + 2));
          MLSTAMP(0, 1);
          if (!DBV) LDS_BARRIER();  // A
          MLSTAMP(0, 2);
          if (!(ml_dbg & 2)) flush4(lds + cbuf_wr(it), PM_HI);
          MLSTAMP(0, 3);
          LDS_BARRIER();  // B
          MLSTAMP(0, 4);
        }
        if (!(ml_dbg & 16)) split_plane();
        if (!(ml_dbg & 16)) contract_plane(wstreamV(k_of(nk - 1)));  // the last unit (the ring's reload is not used)
      }
      pass_S_builder();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        ST1[row * 96 + vm * 32 + r] = accP[q];
      }
    }

    // ---- segment end: partial tiles -> LDS (raw accumulators), summed and scaled on the way to the slab (as k_conv_mf)
    const float i1 = pow2f(clamp100(-(sX + a.sC))), iT2 = pow2f(clamp100(-a.sTw));
    auto i2_of = [&](int row) __attribute__((always_inline)) {
      const int edeg = deg_lds[row] > 0 ? exp_above((float)deg_lds[row]) : 1;
      return pow2f(clamp100(14 + edeg - a.sB));
    };
    // (the inverse column scales of this thread's slab pieces, requested in front of the barrier: not a global round trip inside the epilogue)
    constexpr int NE0 = (32 * 40 + ML_THREADS - 1) / ML_THREADS, NE1 = (32 * 24 + ML_THREADS - 1) / ML_THREADS;
    float4 cfa[NE0], cfb[NE1];
#pragma unroll
    for (int j = 0; j < NE0; ++j) cfa[j] = reinterpret_cast<const float4*>(a.cf0)[(tid + j * ML_THREADS) % 40];
#pragma unroll
    for (int j = 0; j < NE1; ++j) cfb[j] = reinterpret_cast<const float4*>(a.cf1)[((tid + j * ML_THREADS) % 24) & 7];
    LDS_BARRIER();
    {
      const float* __restrict__ ST0 = reinterpret_cast<const float*>(lds);
      const float* __restrict__ ST1 = reinterpret_cast<const float*>(lds + 5 * 32 * 160 * 4);
      float* __restrict__ p0 = a.partial0 + ((size_t)slab * a.n_pad + n0) * (size_t)(a.nt0 * 32);
      float* __restrict__ p1 = a.partial1 + ((size_t)slab * a.n_pad + n0) * 96;
#pragma unroll
      for (int j = 0; j < NE0; ++j) {
        const int idx = tid + j * ML_THREADS;
        if (idx >= 32 * 40) break;
        const int row = idx / 40, c4 = idx - row * 40;
        const float* __restrict__ q0 = ST0 + row * 160 + 4 * c4;
        const float4 a0 = *reinterpret_cast<const float4*>(q0), a1 = *reinterpret_cast<const float4*>(q0 + 5120),
                     a2 = *reinterpret_cast<const float4*>(q0 + 10240), a3 = *reinterpret_cast<const float4*>(q0 + 15360),
                     a4 = *reinterpret_cast<const float4*>(q0 + 20480);
        const float i2 = i2_of(row);
        const float4 cf = cfa[j];
        const float4 v = make_float4(((((((a0.x + a1.x) + a2.x) + a3.x) + a4.x) * i1) * i2) * cf.x, ((((((a0.y + a1.y) + a2.y) + a3.y) + a4.y) * i1) * i2) * cf.y,
                                     ((((((a0.z + a1.z) + a2.z) + a3.z) + a4.z) * i1) * i2) * cf.z, ((((((a0.w + a1.w) + a2.w) + a3.w) + a4.w) * i1) * i2) * cf.w);
        if (row < n_dst) *reinterpret_cast<float4*>(p0 + row * 160 + 4 * c4) = v;
      }
#pragma unroll
      for (int j = 0; j < NE1; ++j) {
        const int idx = tid + j * ML_THREADS;
        if (idx >= 32 * 24) break;
        const int row = idx / 24, c4 = idx - row * 24;
        const float i2 = i2_of(row);
        const float4 c = *reinterpret_cast<const float4*>(ST1 + row * 96 + 4 * c4), t = *reinterpret_cast<const float4*>(ST1 + 3072 + row * 96 + 4 * c4);
        const float4 cf = cfb[j];  // (column 4 (c4 % 8) .. of plane c4 / 8)
        const float4 v = make_float4(((c.x * i1) * i2) * cf.x + (t.x * i1) * iT2, ((c.y * i1) * i2) * cf.y + (t.y * i1) * iT2, ((c.z * i1) * i2) * cf.z + (t.z * i1) * iT2,
                                     ((c.w * i1) * i2) * cf.w + (t.w * i1) * iT2);
        if (row < n_dst) *reinterpret_cast<float4*>(p1 + row * 96 + 4 * c4) = v;
      }
    }
    MLSEG(7);
    LDS_BARRIER();  // the next segment rewrites the tiles
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// k_conv_mlx — the INITIAL projector for the same large spans: k_conv_mfx (jamun_conv_mf.hip: the layer as a hidden layer with 64 scalar
// input channels, formed from the host-split embedding rows) with the window, the block-sparse forming and the register-staged coefficient
// entries of k_conv_ml.  One pass: x^T of 64 channels is 43 KB, the four coefficient components (single buffered) 94 KB.
//   wave = 2 c + t: coefficient component c (0: scalar outputs, 1 + m: x0 (x) v_m -> vector plane m), channel tile t: forms its tile over
//   the occupied blocks, splits it behind the forming barrier and contracts it at the start of the next hidden unit's forming phase
//   (waves 0, 1 into the five scalar-output tiles, the others into their plane; K-split over t, summed at the segment end).
//   All eight waves prepare the next unit's entries in registers during the forming phase and store them between the two barriers.
template <int NKS, bool HALF>
__global__ __launch_bounds__(ML_THREADS) void k_conv_mlx(MlxArgs a) {
  constexpr int RH = 16 * NKS - (HALF ? 8 : 0);
  constexpr int ROWX = HALF ? 32 * NKS - 16 : 32 * NKS + 16, ROWC = 32 * NKS + 16, PL = 32 * ROWC, CC = 2 * PL;
  constexpr int X_H = 0, X_L = 64 * ROWX, X_C = 2 * 64 * ROWX, X_END = X_C + 4 * CC;
  constexpr int EP_END = 2 * 32 * 160 * 4 + 2 * 32 * 96 * 4;
  constexpr int MISC = ((X_END > EP_END ? X_END : EP_END) + 15) & ~15;
  static_assert(MISC + 160 <= JAMUN_MAX_DYN_LDS, "k_conv_mlx: LDS budget");
  extern __shared__ float4 lds4[];
  char* __restrict__ lds = reinterpret_cast<char*>(lds4);
  const int L0 = lds_addr(lds);
  int* __restrict__ deg_lds = reinterpret_cast<int*>(lds + MISC);
  unsigned* __restrict__ mask_lds = reinterpret_cast<unsigned*>(lds + MISC + 132);
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = RFL(tid0 >> 6);
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float m1 = opaque_minus_one();

  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int4 sg0 = ld_const(a.segs + ((size_t)blockIdx.x * a.max_segs + sgi) * 2);
    const int4 sg1 = ld_const(a.segs + ((size_t)blockIdx.x * a.max_segs + sgi) * 2 + 1);
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) __attribute__((always_inline)) { kk = kk < 0 ? 0 : (kk < nk ? kk : nk - 1); return kk < k_run ? k_begin + kk : k_extra; };
    const int2 t_at = make_int2(sg1.y, sg1.z & 255), span = make_int2(sg1.w, sg1.w + (sg1.z >> 8));  // (descriptor inside the segment record, jamun_api.cpp: embed)
    const int n0 = RFL(t_at.x), n_dst = RFL(t_at.y);
    const int s_base = RFL(span.x) & ~1;  // window from an even atom (the rows are stored in pairs)
    const int r = lane & 31, hh = lane >> 5;

    // ---- builder layout: four passes of 8 destinations x 64 slots over the 512 lanes (a destination's slots inside one wave)
    constexpr int NP = 4;
    const int slot0 = (n0 + tid / 64) * a.S + (tid & 63), pstride = 8 * a.S;
    const int dgl = a.deg[n0 + min(tid & 31, n_dst - 1)];
    const int dgv = dgl * ((tid < 32 && tid < n_dst) ? 1 : 0);
    float hv[NP];
    auto load_h = [&](int k) __attribute__((always_inline)) {
      const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
      for (int p = 0; p < NP; ++p) hv[p] = hk[slot0 + p * pstride];
    };
    load_h(k_of(0));
    float evx[NP], evy[NP], evz[NP];
    int sjv[NP], epv[NP], ent[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = tid / 64 + 8 * p, t = tid & 63;
      const bool in = i < n_dst && t < a.S;
      const int sc_ = (n0 + min(i, n_dst - 1)) * a.S + min(t, a.S - 1);  // (unconditional, clamped, masked: as k_conv_ml)
      const int sj_ = a.esrc[sc_], ep_ = a.epair[sc_];
      const float4 ge = a.egeo[sc_];
      const float kin = in ? 1.f : 0.f;
      sjv[p] = in ? sj_ : 0;
      epv[p] = in ? ep_ : 0;
      evx[p] = ge.x * kin; evy[p] = ge.y * kin; evz[p] = ge.z * kin;
    }
    // ---- prologue: zero the coefficient tiles; the window's rows (already split, two atoms per word): a copy, transposed
    for (int idx = tid; idx < 4 * CC / 16; idx += ML_THREADS) reinterpret_cast<float4*>(lds + X_C)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid == 0) *mask_lds = 0u;
    {
      const unsigned* __restrict__ ph = a.xph + (size_t)(s_base >> 1) * 64;
      const unsigned* __restrict__ pl = a.xpl + (size_t)(s_base >> 1) * 64;
      constexpr int NW = (RH / 2) * 64, NQ = (NW + ML_THREADS - 1) / ML_THREADS;
      unsigned vh[NQ], vl[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = tid + ML_THREADS * q;
        vh[q] = ph[min(idx, NW - 1)];  // (unconditional, clamped; pieces past the end are not stored)
        vl[q] = pl[min(idx, NW - 1)];
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int idx = tid + ML_THREADS * q, ch = idx & 63, jp = idx >> 6;
        if (idx < NW) {
          *reinterpret_cast<unsigned*>(lds + X_H + ch * ROWX + 4 * jp) = vh[q];
          *reinterpret_cast<unsigned*>(lds + X_L + ch * ROWX + 4 * jp) = vl[q];
        }
      }
    }
    if (tid < 32) deg_lds[tid] = dgv;
    LDS_BARRIER();
    {
      unsigned bm = 0u;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int i = tid / 64 + 8 * p, t = tid & 63;
        const int dg = deg_lds[i];
        const bool in = t < dg && t < a.S;
        const int jl = (sjv[p] & 0x7fffffff) - s_base;  // (bit 31: bonded)
        const bool valid = in && jl >= 0 && jl < RH;
        if (in && (jl < 0 || jl >= RH)) atomicOr(a.err, 2);
        const int ep = in ? epv[p] : 0;
        if (ep & JAMUN_EP_OVERFLOW) atomicOr(a.err, 1);
        const int pa = (ep & 127) - 1, pb = ((ep >> 7) & 127) - 1;
        const int d0 = pa >= 0 ? pa - t : 0, d1 = pb >= 0 ? pb - t : 0;  // lane distance to the pair's other edges (same lane group)
        const bool active = valid && !((unsigned)ep & JAMUN_EP_OWNED);
        ent[p] = active ? (i * ROWC + 2 * jl) | (d0 << 14) | (d1 << 20) : (ROWC - 16);
        if (active) bm |= 1u << (jl >> 4);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) bm |= (unsigned)__shfl_xor((int)bm, o, 64);
      if (lane == 0) atomicOr(mask_lds, bm);
    }
    const float scC = pow2f(a.sC);
    unsigned stg[NP][4];
    auto prep = [&]() __attribute__((always_inline)) {
      float cc[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int d0 = (ent[p] >> 14) & 63, d1 = (ent[p] >> 20) & 63;
        const float t0 = __shfl(hv[p], lane + d0, 64), t1 = __shfl(hv[p], lane + d1, 64);
        cc[p] = ((hv[p] + (d0 ? t0 : 0.f)) + (d1 ? t1 : 0.f)) * scC;
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float c0 = cc[p], c1 = c0 * evx[p], c2 = c0 * evy[p], c3 = c0 * evz[p];
        const unsigned h01 = cvt_pk_f16_c(c0, c1), h23 = cvt_pk_f16_c(c2, c3);
        stg[p][0] = h01; stg[p][1] = h23;
        stg[p][2] = cvt_pk_f16_c(resid_lo_c(c0, h01, m1), resid_hi_c(c1, h01, m1));
        stg[p][3] = cvt_pk_f16_c(resid_lo_c(c2, h23, m1), resid_hi_c(c3, h23, m1));
      }
    };
    auto flush = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned h01 = stg[p][0], h23 = stg[p][1], l01 = stg[p][2], l23 = stg[p][3];
        char* __restrict__ d = lds + X_C + (ent[p] & 0x3fff);
        *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h01 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + PL) = (unsigned short)(l01 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 2 * PL) = (unsigned short)(h01 >> 16);
        *reinterpret_cast<unsigned short*>(d + 3 * PL) = (unsigned short)(l01 >> 16);
        *reinterpret_cast<unsigned short*>(d + 4 * PL) = (unsigned short)(h23 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 5 * PL) = (unsigned short)(l23 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 6 * PL) = (unsigned short)(h23 >> 16);
        *reinterpret_cast<unsigned short*>(d + 7 * PL) = (unsigned short)(l23 >> 16);
      }
    };
    LDS_BARRIER();  // mask, x^T complete, tiles zeroed
    unsigned long long blist = 0ull;
    int nb = 0;
    {
      unsigned m = (unsigned)RFL((int)*mask_lds);
      while (m) {
        blist |= (unsigned long long)__builtin_ctz(m) << (4 * nb);
        ++nb;
        m &= m - 1u;
      }
      nb = RFL(nb);
    }
    auto boff = [&](int i) __attribute__((always_inline)) {
      const int ic = min(i, nb - 1);
      return nb > 0 ? (int)((blist >> (4 * ic)) & 15ull) << 5 : 0;
    };
    if (tid == 0 && a.mfma_count) atomicAdd(a.mfma_count, (unsigned long long)nk * (unsigned long long)(24 * nb + 96));

    const int comp = wave >> 1, ct = wave & 1;
    const int edeg_r = deg_lds[r] > 0 ? exp_above((float)deg_lds[r]) : 1;
    const float rs = pow2f(-14 - edeg_r);  // formed values are below in-degree x 2^28 -> below 2^14
    const int fox = r * ROWX + 16 * hh + L0 + X_H + 32 * ct * ROWX, foc = r * ROWC + 16 * hh + L0 + X_C + comp * CC;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.wx), 0, 0x7fffffff, 0x00020000);
    const int wvo = lane * 16;
    auto wload = [&](int so) __attribute__((always_inline)) { return __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so, 0); };
    struct Frag { float4 ah, al, bh, bl; };
    auto split16 = [&](const f32x16& F, float4 (&Ah)[2], float4 (&Al)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        unsigned ph[4], pl[4];
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          const float v0 = F[8 * s2 + 2 * p2] * rs, v1 = F[8 * s2 + 2 * p2 + 1] * rs;
          ph[p2] = cvt_pk_f16_c(v0, v1);
          pl[p2] = cvt_pk_f16_c(resid_lo_c(v0, ph[p2], m1), resid_hi_c(v1, ph[p2], m1));
        }
        Ah[s2] = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
        Al[s2] = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
      }
    };
    float* __restrict__ ST0 = reinterpret_cast<float*>(lds);                     // [2 channel tiles][32][160]
    float* __restrict__ ST1 = reinterpret_cast<float*>(lds + 2 * 32 * 160 * 4);  // [2][32][96]
    float4 sAh[2], sAl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) sAh[s2] = sAl[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
    prep();
    flush();
    load_h(k_of(1));

    auto k_loop = [&](auto contract) __attribute__((always_inline)) {
      for (int it = 0; it < nk; ++it) {
        int fox_ = fox, foc_ = foc;
        asm volatile("" : "+v"(fox_), "+v"(foc_));
        auto fr = [&](int i) __attribute__((always_inline)) { const int bo = boff(i); return Frag{lds_f4(fox_ + bo), lds_f4(fox_ + bo + (X_L - X_H)), lds_f4(foc_ + bo), lds_f4(foc_ + bo + PL)}; };
        Frag fA = fr(0), fB = fr(1);
        MF_SCHED();
        if (it > 0) contract(it - 1);  // the previous unit's tile (registers and weights only), behind this unit's first fragment reads
        f32x16 F = zero16;
        for (int i = 0; i < nb; i += 2) {
          MF_SCHED();
          M3(F, fA.ah, fA.al, fA.bh, fA.bl);
          MF_SCHED();
          fA = fr(i + 2);
          MF_SCHED();
          if (i + 1 < nb) { M3(F, fB.ah, fB.al, fB.bh, fB.bl); }
          MF_SCHED();
          fB = fr(i + 3);
        }
        prep();  // (the entries of unit it + 1, in registers; on the last unit from clamped loads, never read)
        load_h(k_of(it + 2));
        LDS_BARRIER();  // A: every read of C(k) done
        flush();
        split16(F, sAh, sAl);
        MF_SCHED();
        LDS_BARRIER();  // B: C(k + 1) complete
      }
      contract(nk - 1);
    };
    if (comp == 0) {
      // scalar outputs: 20 weight blocks per hidden unit ((hi, lo) per (output tile n, K-step s2)), ring of 10
      constexpr int NB = 20, RD = 10;
      f32x16 accS[5];
#pragma unroll
      for (int n = 0; n < 5; ++n) accS[n] = zero16;
      auto wstream = [&](int k) __attribute__((always_inline)) { return (k * 48 + 20 * ct) * 1024; };
      u32x4 RB[RD];
      {
        const int c0 = wstream(k_of(0));
#pragma unroll
        for (int p = 0; p < RD; ++p) RB[p] = wload(c0 + p * 1024);
      }
      LDS_BARRIER();  // C(k0) complete
      k_loop([&](int u) __attribute__((always_inline)) {
        const int cur = wstream(k_of(u)), nxt = wstream(k_of(u + 1));
#pragma unroll
        for (int n = 0; n < 5; ++n)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int p = 2 * (2 * n + s2);
            M3(accS[n], sAh[s2], sAl[s2], RB[p % RD], RB[(p + 1) % RD]);
#pragma unroll
            for (int e = 0; e < 2; ++e) RB[(p + e) % RD] = (p + e + RD < NB) ? wload(cur + (p + e + RD) * 1024) : wload(nxt + (p + e + RD - NB) * 1024);
            MF_SCHED();
          }
      });
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
#pragma unroll
        for (int n = 0; n < 5; ++n) ST0[(ct * 32 + row) * 160 + 32 * n + r] = accS[n][q];
      }
    } else {
      // vector plane comp - 1: 4 weight blocks per hidden unit ((hi, lo) per K-step), the next unit's requested as they are used
      f32x16 accP = zero16;
      auto wstream = [&](int k) __attribute__((always_inline)) { return (k * 48 + 40 + 4 * ct) * 1024; };
      u32x4 RB[4];
      {
        const int c0 = wstream(k_of(0));
#pragma unroll
        for (int p = 0; p < 4; ++p) RB[p] = wload(c0 + p * 1024);
      }
      LDS_BARRIER();
      k_loop([&](int u) __attribute__((always_inline)) {
        const int nxt = wstream(k_of(u + 1));
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          M3(accP, sAh[s2], sAl[s2], RB[2 * s2], RB[2 * s2 + 1]);
          RB[2 * s2] = wload(nxt + (2 * s2) * 1024);
          RB[2 * s2 + 1] = wload(nxt + (2 * s2 + 1) * 1024);
          MF_SCHED();
        }
      });
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        ST1[(ct * 32 + row) * 96 + (comp - 1) * 32 + r] = accP[q];
      }
    }
    LDS_BARRIER();
    {
      const float i1 = pow2f(clamp100(-(a.sX + a.sC)));
      auto i2_of = [&](int row) __attribute__((always_inline)) {
        const int edeg = deg_lds[row] > 0 ? exp_above((float)deg_lds[row]) : 1;
        return pow2f(14 + edeg);
      };
      float* __restrict__ p0 = a.partial0 + ((size_t)slab * a.n_pad + n0) * (size_t)(a.nt0 * 32);
      float* __restrict__ p1 = a.partial1 + ((size_t)slab * a.n_pad + n0) * 96;
      for (int idx = tid; idx < 32 * 40; idx += ML_THREADS) {
        const int row = idx / 40, c4 = idx - row * 40;
        const float4 a0 = *reinterpret_cast<const float4*>(ST0 + row * 160 + 4 * c4), a1 = *reinterpret_cast<const float4*>(ST0 + 5120 + row * 160 + 4 * c4);
        const float4 cf = reinterpret_cast<const float4*>(a.cf0)[c4];
        const float i2 = i2_of(row);
        if (row < n_dst)
          *reinterpret_cast<float4*>(p0 + row * 160 + 4 * c4) = make_float4((((a0.x + a1.x) * i1) * i2) * cf.x, (((a0.y + a1.y) * i1) * i2) * cf.y,
                                                                             (((a0.z + a1.z) * i1) * i2) * cf.z, (((a0.w + a1.w) * i1) * i2) * cf.w);
      }
      for (int idx = tid; idx < 32 * 24; idx += ML_THREADS) {
        const int row = idx / 24, c4 = idx - row * 24;
        const float4 a0 = *reinterpret_cast<const float4*>(ST1 + row * 96 + 4 * c4), a1 = *reinterpret_cast<const float4*>(ST1 + 3072 + row * 96 + 4 * c4);
        const float4 cf = reinterpret_cast<const float4*>(a.cf1)[c4 & 7];
        const float i2 = i2_of(row);
        if (row < n_dst)
          *reinterpret_cast<float4*>(p1 + row * 96 + 4 * c4) = make_float4((((a0.x + a1.x) * i1) * i2) * cf.x, (((a0.y + a1.y) * i1) * i2) * cf.y,
                                                                            (((a0.z + a1.z) * i1) * i2) * cf.z, (((a0.w + a1.w) * i1) * i2) * cf.w);
      }
    }
    LDS_BARRIER();  // the next segment rewrites the tiles
  }
}

void conv_ml_print_stamps() {
#ifdef ML_TRACE
  static unsigned long long tr[8][2][24][6], sg[8][3][10];
  if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_mltrace), sizeof(tr)) != hipSuccess || hipMemcpyFromSymbol(sg, HIP_SYMBOL(g_mlseg), sizeof(sg)) != hipSuccess || sg[0][0][0] == 0) return;
  const unsigned long long t0 = sg[0][0][0];
  fprintf(stderr, "ml segments (workgroup 7): start / max barrier / staged V / loop V entered / loop V left / loop S entered / loop S left / slab stored\n");
  for (int q = 0; q < 3; ++q)
    for (int w = 0; w < 8; ++w) {
      if (sg[w][q][0] == 0) continue;
      fprintf(stderr, "  seg %d wave %d:", q, w);
      for (int i = 0; i < 8; ++i) fprintf(stderr, " %8lld", (long long)(sg[w][q][i] - t0));
      fprintf(stderr, "\n");
    }
  for (int ps = 0; ps < 2; ++ps) {
    fprintf(stderr, "ml pass %s, first segment: per k-step and wave: top / formed / after barrier A / phase B done / after barrier B (cycles since segment start)\n", ps ? "S" : "V");
    for (int st = 0; st < 24; ++st)
      for (int w = 0; w < 8; ++w) {
        if (tr[w][ps][st][0] == 0) continue;
        fprintf(stderr, "  step %2d wave %d:", st, w);
        for (int i = 0; i < 5; ++i) fprintf(stderr, " %8lld", tr[w][ps][st][i] ? (long long)(tr[w][ps][st][i] - t0) : -1LL);
        fprintf(stderr, "\n");
      }
  }
#endif
}

namespace {
template <int NKS, bool HALF, bool DBV>
int launch_ml(const MlArgs& a, int grid, hipStream_t st) {
  static_assert(MlGeo<NKS, HALF, DBV>::LDS_BYTES <= JAMUN_MAX_DYN_LDS, "k_conv_ml: LDS budget");
  constexpr size_t smem = MlGeo<NKS, HALF, DBV>::LDS_BYTES;
  if (a.x1) hipLaunchKernelGGL((k_conv_ml<NKS, HALF, DBV, true>), dim3(grid), dim3(ML_THREADS), smem, st, a);
  else hipLaunchKernelGGL((k_conv_ml<NKS, HALF, DBV, false>), dim3(grid), dim3(ML_THREADS), smem, st, a);
  return 0;
}
template <int NKS, bool HALF, bool DBV>
int set_lds() {
  if (hipFuncSetAttribute((const void*)k_conv_ml<NKS, HALF, DBV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  return hipFuncSetAttribute((const void*)k_conv_ml<NKS, HALF, DBV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) == hipSuccess ? 0 : -1;
}
}  // namespace

// rows of the window the instantiation for a span of `rows_needed` rows (counted from the even atom below the span's first) multiplies; 0: none
int conv_ml_window(int rows_needed) {
  if (rows_needed <= 96) return 96;
  if (rows_needed <= 128) return 128;
  if (rows_needed <= 168) return 168;
  return 0;
}

int launch_conv_ml(const MlArgs& a, int grid, hipStream_t st) {
  if (a.XS != 216 || a.nt0 != 5 || a.S > 40 || a.S < 33 || (a.t_stride & 1) || !a.epair) return -1;
  switch (a.window) {
    case 96: return launch_ml<6, false, true>(a, grid, st);
    case 128: return launch_ml<8, false, false>(a, grid, st);
    case 168: return launch_ml<11, true, false>(a, grid, st);
    default: return -1;
  }
}

namespace {
template <int NKS, bool HALF>
int launch_mlx(const MlxArgs& a, int grid, hipStream_t st) {
  constexpr int ROWX = HALF ? 32 * NKS - 16 : 32 * NKS + 16, ROWC = 32 * NKS + 16, X_END = 2 * 64 * ROWX + 4 * 2 * 32 * ROWC;
  constexpr int EP_END = 2 * 32 * 160 * 4 + 2 * 32 * 96 * 4;
  constexpr size_t smem = (((X_END > EP_END ? X_END : EP_END) + 15) & ~15) + 160;
  hipLaunchKernelGGL((k_conv_mlx<NKS, HALF>), dim3(grid), dim3(ML_THREADS), smem, st, a);
  return 0;
}
}  // namespace

int launch_conv_mlx(const MlxArgs& a, int grid, hipStream_t st) {
  if (a.nt0 != 5 || a.S > 64 || !a.epair) return -1;
  switch (a.window) {
    case 96: return launch_mlx<6, false>(a, grid, st);
    case 128: return launch_mlx<8, false>(a, grid, st);
    case 168: return launch_mlx<11, true>(a, grid, st);
    default: return -1;
  }
}

int conv_ml_set_max_lds() {
  if (hipFuncSetAttribute((const void*)k_conv_mlx<6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  if (hipFuncSetAttribute((const void*)k_conv_mlx<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  if (hipFuncSetAttribute((const void*)k_conv_mlx<11, true>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  if (set_lds<6, false, true>() != 0) return -1;
  if (set_lds<8, false, false>() != 0) return -1;
  if (set_lds<11, true, false>() != 0) return -1;
  return 0;
}
