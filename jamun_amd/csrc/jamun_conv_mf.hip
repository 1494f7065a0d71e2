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

#include "jamun_mf_dev.h"

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

#ifdef MF_TRACE  // per-wave timeline of workgroup 7 (diagnostic builds): [wave][k-step][stamp]
__device__ unsigned long long g_mftrace[8][40][8];
#define MSTAMP(slot) do { if (blockIdx.x == 7 && trc < 40) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_mftrace[wave][trc][slot] = t_; } } while (0)
__device__ unsigned long long g_mfseg[8][4][8];  // [wave][segment][stamp]: segment prologue / epilogue
#define SSTAMP(slot) do { if (blockIdx.x == 7 && sgi < 4) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_mfseg[wave][sgi][slot] = t_; } } while (0)
#else
#define MSTAMP(slot) do { } while (0)
#define SSTAMP(slot) do { } while (0)
#endif


// SPD: the builder's lane layout.  32: edge strides up to 32 — four passes of 8 destinations x 32 slots over the 256 builder lanes;
// 64: eight passes of 4 destinations x 64 slots; 40: edge strides 33..40 — the four passes of layout 32 plus ONE pass for slots 32..39 of
// all 32 destinations (lane = (destination of this wave, slot - 32): k_geom caps the radial neighbours at 32, so those slots hold bonded
// edges only): five passes instead of eight for a 33-atom molecule with every atom at the neighbour cap.
// NKS: K-steps of 16 source rows per forming product (4 = the whole 64-row window; 3 when EVERY tile of the launch has its sources
// in the window's first 48 rows — one 33-atom molecule per tile: 57 instead of 76 forming products per hidden unit; chosen by the host,
// MfArgs::nks: a per-step branch on the span breaks the pinned schedule, a compile-time count does not)
// X1: the opt-in reduced-precision mode (jamun_tuning.f16x1): every product is its hi hi MFMA alone (operands rounded to 11 bits, fp32 accumulation)
template <int SPD, int NKS, bool X1>
__global__ __launch_bounds__(MF_THREADS) void k_conv_mf(MfArgs a) {
  extern __shared__ float4 lds4[];
  char* __restrict__ lds = reinterpret_cast<char*>(lds4);
  const int L0 = lds_addr(lds);
  int* __restrict__ deg_lds = reinterpret_cast<int*>(lds + MF_MISC);
  unsigned* __restrict__ xmax_lds = reinterpret_cast<unsigned*>(lds + MF_MISC + 128);

  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = RFL(tid0 >> 6);
#if defined(MF_EXP)  // timing experiments, compile-time (-DMF_EXP=bits; results are wrong): 1 no weight loads in the k loop, 2 no forming
                     // MFMAs, 4 no contraction MFMAs, 8 helpers build nothing, 16 no split, 32 no forming LDS reads
  constexpr int dbg = MF_EXP;
#else
  constexpr int dbg = 0;
#endif

#ifdef MF_TRACE
  int trc = 0;
#endif
  // (the descriptors of segment s + 1 — list entry, then its tile's atoms and span — are requested during segment s: three dependent
  // global round trips off the critical path of every segment but the first)
  const int4* __restrict__ my_segs = a.segs + (size_t)blockIdx.x * a.max_segs * 2;
  int4 sg0 = ld_const(my_segs), sg1 = ld_const(my_segs + 1);
  // (the record carries its tile's descriptor, jamun_api.cpp: embed — no second, dependent trip to the tile tables)
  int2 t_at = make_int2(sg1.y, sg1.z & 255), span = make_int2(sg1.w, sg1.w + (sg1.z >> 8));
  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    int4 nsg0 = make_int4(-1, 0, 0, 0), nsg1 = make_int4(-1, 0, 0, 0);
    if (sgi + 1 < a.max_segs) { nsg0 = ld_const(my_segs + 2 * (sgi + 1)); nsg1 = ld_const(my_segs + 2 * (sgi + 1) + 1); }
    int lane = lane0;
    asm volatile("" : "+v"(lane));  // (keeps lane-derived addresses from being hoisted out of the segment loop and spilled)
    const int tid = wave * 64 + lane;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) { kk = kk < 0 ? 0 : (kk < nk ? kk : nk - 1); return kk < k_run ? k_begin + kk : k_extra; };
    const int n0 = RFL(t_at.x), n_dst = RFL(t_at.y);
    const int s_lo = RFL(span.x), rows = RFL(span.y) - s_lo;
    const int s_base = s_lo & ~1, off = s_lo - s_base;  // window of 64 source rows from an even atom (8-byte T loads)
    SSTAMP(0);

    // (in-degrees of the tile: the FIRST load of the segment, stored after the first barrier — waiting for it before that barrier
    // would wait for every load issued in front of it, i.e. hold all eight waves for a global round trip)
    int dgv = 0;
    if (tid < 32 && tid < n_dst) dgv = a.deg[n0 + tid];
    // ---- builder lanes (waves 0..3, the lighter matrix role): their share of the coefficient-tile entries (edge slots) and of the T
    // tile.  Edge strides up to 32: four passes of 8 destinations x 32 slots; above: eight passes of 4 destinations x 64 slots (a
    // destination's slots stay within one wave: the pair bookkeeping uses ballots and shuffles)
    constexpr int BT = 256;                          // builder threads
    constexpr bool XP = SPD == 40;                   // layout with an extra pass of eight slots per destination behind the main passes
    // (a 16 + 8 layout for strides up to 24 — three passes instead of four on 17-atom molecules — was measured and dropped: cfg2 0.1132 ->
    // 0.1133 ms per launch; there the builder waves wait at the barrier anyway, the plane waves' streams set the step)
    constexpr int SPL = XP ? SPD - 8 : SPD;          // slots per destination of the main passes
    constexpr int LG = SPL == 16 ? 4 : 5;            // log2 of a main pass' lane group
    constexpr int DPP = BT / SPL, NPM = 32 / DPP;    // destinations per main pass, main passes
    constexpr int NP = NPM + (XP ? 1 : 0);           // + the pass of slots SPL .. SPL + 7
    constexpr int NTV = 1024 / BT;                   // T elements (pairs of source rows) per builder lane
    const bool builder = tid < BT;
    // (destination, slot) of this lane in pass p; layout 40, last pass: wave w owns the destinations of its main passes —
    // 8 (q >> 1) + 2 w + (q & 1), q = lane >> 3 — so that every edge of a destination stays inside one wave (lane shuffles below)
    // (extra pass: lane = (q, slot - SPL), q = lane >> 3 = the wave's q-th destination: main pass q / GPW, lane group q % GPW)
    constexpr int GPW = 64 / SPL;  // lane groups (destinations) of a wave per main pass
    auto dst_of = [&](int p) { return (XP && p == NPM) ? DPP * ((lane >> 3) / GPW) + GPW * wave + ((lane >> 3) % GPW) : ((tid & (BT - 1)) + BT * p) / SPL; };
    auto slt_of = [&](int p) { return (XP && p == NPM) ? SPL + (lane & 7) : ((tid & (BT - 1)) + BT * p) % SPL; };
    // per pass ONE packed word: bits 0..12 byte offset of the entry inside a coefficient plane (i * 144 + 2 jl), 13..18 / 19..24 where a
    // second / third edge of the same pair sits (0: none): the LANE DISTANCE inside this pass, or — bits 25 / 26 set, layout 40 — the lane
    // of the last pass.  Lanes without an entry point at the pad bytes of row 0 (offset 128 of every plane; never read), so that build()
    // is straight-line code.
    int ent[NP];
    const int slot0 = (n0 + dst_of(0)) * a.S + slt_of(0), pstride = DPP * a.S, slot_x = (n0 + dst_of(NP - 1)) * a.S + slt_of(NP - 1);
    auto slot_of = [&](int p) { return (XP && p == NPM) ? slot_x : slot0 + p * pstride; };
    float hv[NP];
    float2 tv[NTV];
    // (loads only, nothing consumed here: a use would wait for EVERY vector load in flight, the weight ring included)
    // Every lane loads the h~ of its OWN slot, the lanes of a pair's second / third edge included: the owner of the entry fetches
    // theirs with a lane shuffle (same wave: slot distance = lane distance) instead of two more loads per pass.
    auto load_k = [&](int k) {  // h~ of this lane's edges and this lane's T elements of hidden unit k
      if (!builder) return;
      const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
      for (int p = 0; p < NP; ++p) hv[p] = hk[slot_of(p)];  // (lanes past the in-degree read a neighbouring slot or the table's slack: never used)
      const float* __restrict__ tk = a.Tt + (size_t)k * 32 * a.t_stride + s_base;
#pragma unroll
      for (int q = 0; q < NTV; ++q) {
        const int item = tid + BT * q, wp = item >> 5, jp = item & 31;
        tv[q] = *reinterpret_cast<const float2*>(tk + (size_t)wp * a.t_stride + 2 * jp);
      }
    };
    load_k(k_of(0));
    float evx[NP], evy[NP], evz[NP];
    int sjv[NP], epv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {  // (all passes' edge records requested at once, and before the rows are staged: their round trip hides behind that)
      const int i = dst_of(p), t = slt_of(p);
      const int slot = slot_of(p);
      const bool in = builder && i < n_dst && t < a.S;  // (the in-degree is applied below: slots past it hold stale records)
      sjv[p] = in ? a.esrc[slot] : 0;
      epv[p] = in ? a.epair[slot] : 0;
      float4 ge = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in) ge = a.egeo[slot];
      evx[p] = ge.x; evy[p] = ge.y; evz[p] = ge.z;
    }

    // ---- segment prologue (all threads): zero the coefficient tiles, stage the span's rows transposed and split
    for (int idx = tid; idx < 2 * MF_CB / 16; idx += MF_THREADS) reinterpret_cast<float4*>(lds + MF_C)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 144) {  // channel rows 120..127 of the scalar block (inputs of the last K-step that do not exist)
      const int pl = tid / 72, q = tid - pl * 72;
      *reinterpret_cast<float4*>(lds + (pl ? MF_X0L : MF_X0H) + 120 * MF_ROWB + 16 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid == 0) *xmax_lds = 0u;
    // The span's rows are loaded, scaled, split and stored transposed by waves 4..7 alone (seven float4 columns of a row pair per lane)
    // WHILE waves 0..3 — their SIMD partners — examine the edge records and build the first hidden unit's tiles: both are vector work
    // of a few hundred instructions per lane that used to run one after the other on all eight waves.
    constexpr int NXQ = 7;  // 32 row pairs x 54 float4 columns over 256 lanes
    float4 va[NXQ], vb[NXQ];
    float mx = 0.f;
    const bool stager = wave >= 4;
    const int xt = tid - 256;
#pragma unroll
    for (int q = 0; q < NXQ; ++q) va[q] = vb[q] = make_float4(0.f, 0.f, 0.f, 0.f);  // (defined on every path: no value carried around the segment loop)
    // lane = (row pair jp, float4 column c40 + 8 q): the two row pointers and their predicates do not depend on q — one address pair, the
    // columns by immediate offsets
    const int xjp = xt & 31, xc40 = (xt >> 5) & 7;
    // (per-channel powers of two 2^e_u, MfArgs::gx: the inverse of what the host took out of the channel's weight rows — applied BEFORE the
    // span's maximum is measured, so that every channel enters the f16 split with the weight of its contribution)
    float4 gq[NXQ];
#pragma unroll
    for (int q = 0; q < NXQ; ++q) gq[q] = (stager && xc40 + 8 * q < 54) ? reinterpret_cast<const float4*>(a.gx)[xc40 + 8 * q] : make_float4(1.f, 1.f, 1.f, 1.f);
    {
      const int j0 = 2 * xjp - off, j1 = j0 + 1;  // rows relative to the span
      const bool in0 = stager && j0 >= 0 && j0 < rows, in1 = stager && j1 >= 0 && j1 < rows;
      const float4* __restrict__ pa = reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + j0) * a.XS) + xc40;
      const float4* __restrict__ pb = reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + j1) * a.XS) + xc40;
      if (in0) {
#pragma unroll
        for (int q = 0; q < NXQ - 1; ++q) va[q] = pa[8 * q];
        if (xc40 < 6) va[NXQ - 1] = pa[8 * (NXQ - 1)];  // (columns 48..53)
      }
      if (in1) {
#pragma unroll
        for (int q = 0; q < NXQ - 1; ++q) vb[q] = pb[8 * q];
        if (xc40 < 6) vb[NXQ - 1] = pb[8 * (NXQ - 1)];
      }
    }
    SSTAMP(1);
    LDS_BARRIER();  // tiles zeroed, xmax = 0 (and the previous segment's staging tile read)
    if (tid < 32) deg_lds[tid] = dgv;
    if (stager) {
#pragma unroll
      for (int q = 0; q < NXQ; ++q) {
        va[q] = make_float4(va[q].x * gq[q].x, va[q].y * gq[q].y, va[q].z * gq[q].z, va[q].w * gq[q].w);
        vb[q] = make_float4(vb[q].x * gq[q].x, vb[q].y * gq[q].y, vb[q].z * gq[q].z, vb[q].w * gq[q].w);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(va[q].x), fabsf(va[q].y)), fmaxf(fabsf(va[q].z), fabsf(va[q].w))));
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(vb[q].x), fabsf(vb[q].y)), fmaxf(fabsf(vb[q].z), fabsf(vb[q].w))));
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      if (lane == 0) atomicMax(xmax_lds, __float_as_uint(mx));
    }
    LDS_BARRIER();
    const float xm = __uint_as_float(*xmax_lds);
    const int sX = RFL(xm > 0.f ? clamp40(14 - exp_above(xm)) : 0);
    if (stager) {
      const float scx = pow2f(sX);
#pragma unroll
      for (int q = 0; q < NXQ; ++q) {
        const int c4 = xc40 + 8 * q, jp = xjp;
        if (c4 < 54) {
          const float ea[4] = {va[q].x, va[q].y, va[q].z, va[q].w}, eb[4] = {vb[q].x, vb[q].y, vb[q].z, vb[q].w};
          // byte offsets of the four channels' hi rows: float4 columns below 30 are scalar channels (q < 3: always, q > 3: never,
          // q = 3: per wave), the others vector elements 3 u + m -> row (m, u); the division is done once per column
          const bool sc4 = q < 3 || (q == 3 && c4 < 30);
          const int e1 = 4 * c4 - 120, u0 = e1 / 3, m0 = e1 - 3 * u0;
          const int lo_off = sc4 ? (MF_X0L - MF_X0H) : (MF_X1L - MF_X1H);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int mm_ = m0 + e, wr = mm_ >= 3 ? 1 : 0;
            const int rowb = sc4 ? MF_X0H + (4 * c4 + e) * MF_ROWB : MF_X1H + ((mm_ - 3 * wr) * 32 + u0 + wr) * MF_ROWB;
            const float a0 = ea[e] * scx, b0 = eb[e] * scx;
            const unsigned ph = cvt_pk_f16(a0, b0), pl = cvt_pk_f16(resid_lo(a0, ph), resid_hi(b0, ph));
            *reinterpret_cast<unsigned*>(lds + rowb + 4 * jp) = ph;
            *reinterpret_cast<unsigned*>(lds + rowb + lo_off + 4 * jp) = pl;
          }
        }
      }
    }

    const int r = lane & 31, hh = lane >> 5;
    SSTAMP(2);

#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int i = dst_of(p), t = slt_of(p);
      const int dg = deg_lds[i];
      const bool in = builder && t < dg && t < a.S;
      const int jl = (sjv[p] & 0x7fffffff) - s_base;  // (bit 31: bonded)
      const bool valid = in && jl >= 0 && jl < 16 * NKS;
      if (in && jl >= 16 * NKS) atomicOr(a.err, 2);  // a source outside the rows this instantiation multiplies: the host's plan is wrong
      // Several edges of one (source, destination) pair (a bonded pair inside the cutoff has a radial and a bonded edge:
      // src/jamun/model/denoiser.py:152) share ONE entry of the coefficient tile: the first slot of the pair owns it and adds the
      // h~ of the others.  k_geom has matched them (JAMUN_EP_*: an owned slot carries bit 31, an owner the slots of up to two others).
      const int ep = in ? epv[p] : 0;
      if (ep & JAMUN_EP_OVERFLOW) atomicOr(a.err, 1);  // more than three edges of one pair: not representable here (excluded by the host)
      const int pa = (ep & 127) - 1, pb = ((ep >> 7) & 127) - 1;  // slots of the pair's other edges (-1: none); always behind this one
      auto ref = [&](int ps) {  // where this lane finds the h~ of slot ps of its destination
        if (ps < 0) return 0;
        if (XP && p < NPM && ps >= SPL) return (((GPW * p + (lane >> LG)) << 3) + (ps - SPL)) | 64;
        return ps - t;
      };
      const int r0 = ref(pa), r1 = ref(pb);
      const bool active = valid && !((unsigned)ep & JAMUN_EP_OWNED);
      ent[p] = active ? (i * MF_ROWB + 2 * jl) | ((r0 & 63) << 13) | ((r1 & 63) << 19) | ((r0 >> 6) << 25) | ((r1 >> 6) << 26) : 128;
    }
    SSTAMP(3);
    const float scC = pow2f(a.sC), scT = pow2f(clamp100(sX + a.sTw));
    auto coef = [&](int p) {  // h~ of pass p's entry: this lane's edge + the pair's other edges
      const int d0 = (ent[p] >> 13) & 63, d1 = (ent[p] >> 19) & 63;
      float t0 = __shfl(hv[p], lane + d0, 64), t1 = __shfl(hv[p], lane + d1, 64);
      bool u0 = d0 != 0, u1 = d1 != 0;
      if constexpr (XP) {
        if (p < NPM) {  // (the other edge in the extra pass: absolute lane)
          const float q0 = __shfl(hv[NP - 1], d0, 64), q1 = __shfl(hv[NP - 1], d1, 64);
          if ((ent[p] >> 25) & 1) { t0 = q0; u0 = true; }
          if ((ent[p] >> 26) & 1) { t1 = q1; u1 = true; }
        }
      }
      return (hv[p] + (u0 ? t0 : 0.f)) + (u1 ? t1 : 0.f);
    };
    auto build = [&](int buf) {  // the loaded hidden unit -> coefficient tiles and T tile of buffer `buf`
      if constexpr ((dbg & 8) != 0) return;
      if (!builder) return;
      char* __restrict__ cbuf = lds + MF_C + buf * MF_CB;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        {
          const float c0 = coef(p) * scC, c1 = c0 * evx[p], c2 = c0 * evy[p], c3 = c0 * evz[p];
          const unsigned h01 = cvt_pk_f16(c0, c1), h23 = cvt_pk_f16(c2, c3);
          const unsigned l01 = cvt_pk_f16(resid_lo(c0, h01), resid_hi(c1, h01)), l23 = cvt_pk_f16(resid_lo(c2, h23), resid_hi(c3, h23));
          char* __restrict__ d = cbuf + (ent[p] & 0x1fff);
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
      for (int q = 0; q < NTV; ++q) {
        const int item = tid + BT * q, wp = item >> 5, jp = item & 31;
        const int j0 = 2 * jp - off;
        const float t0 = (j0 >= 0 && j0 < rows) ? tv[q].x * scT : 0.f, t1 = (j0 + 1 >= 0 && j0 + 1 < rows) ? tv[q].y * scT : 0.f;
        const unsigned ph = cvt_pk_f16(t0, t1), pl = cvt_pk_f16(resid_lo(t0, ph), resid_hi(t1, ph));
        *reinterpret_cast<unsigned*>(tbuf + wp * MF_ROWB + 4 * jp) = ph;
        *reinterpret_cast<unsigned*>(tbuf + MF_PL + wp * MF_ROWB + 4 * jp) = pl;
      }
    };

    // ---- matrix work.  Waves w and w + 4 share a SIMD (waves are dealt to the SIMDs cyclically): two independent streams per
    // matrix pipe, so that the LDS reads, the splits and the weight loads of one fill behind the MFMAs of the other.
    //   wave w < 4:  scalar channels 32 w .. 32 w + 31: form, split, contract into all five scalar-output tiles   (42 MFMAs per k)
    //   wave 4:      dot(x1, v): form over 3 x 64 source rows, split, contract into the five scalar-output tiles (66)
    //   wave 5 + m:  vector plane m: x1[m] and (x1 x v)[m] -> contract into the plane; T term                     (60)
    // Formed values are below in-degree x 2^28: x 2^-(14 + e(in-degree)) of this lane's destination -> below 2^14.
    const int edeg_r = deg_lds[r] > 0 ? exp_above((float)deg_lds[r]) : 1;
    const float rs = pow2f(-14 - edeg_r);
    const int fo = r * MF_ROWB + 16 * hh + L0;  // this lane's fragment offset inside a 32-row tile
    // weight blocks: buffer loads with ONE address register (the lane's 16 bytes) and the block offset in a scalar register
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.wm), 0, 0x7fffffff, 0x00020000);
    const int wvo = lane * 16;
    auto wload = [&](int so) { return __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so, 0); };
    // forming: F[u][i] += sum_j x^T[u][j] C[i][j] (M3: f16x3) over K-steps of 16 source rows; one step reads four fragments (x hi, x lo
    // at xa, xa + xlo; C hi, C lo at ca, ca + MF_PL).  The reads of a step are issued TWO steps ahead of its MFMAs (an LDS round trip is
    // ~2 steps of 3 MFMAs) and pinned there with scheduling barriers.
    struct Frag { float4 ah, al, bh, bl; };
    auto ldf = [&](int xa, int xlo, int ca) -> Frag {
      if constexpr ((dbg & 32) != 0) { const float4 c1 = make_float4(1.f, 1.f, 1.f, 1.f); return Frag{c1, c1, c1, c1}; }
      return Frag{lds_f4(xa), lds_f4(xa + xlo), lds_f4(ca), lds_f4(ca + MF_PL)};
    };
    auto mm = [&](f32x16& F, const Frag& f) {
      if constexpr (!(dbg & 2)) { MX(F, f.ah, f.al, f.bh, f.bl); }
      else { F[0] += f.ah.x + f.al.x + f.bh.x + f.bl.x; }
    };
    auto split = [&](const f32x16& F, float4 (&Ah)[2], float4 (&Al)[2]) {
      if constexpr ((dbg & 16) != 0) { Ah[0] = Ah[1] = Al[0] = Al[1] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
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
    // segment end: row i of the contraction carries 2^(sX + sC - 14 - e(in-degree_i)) and column w 2^sB_w (the host's column scale of the
    // weights, undone by cf0 / cf1), the T term 2^(sX + sC + sTw); several factors each, so that no intermediate power of two leaves
    // the fp32 range
    const float i1 = pow2f(clamp100(-(sX + a.sC))), iT2 = pow2f(clamp100(-a.sTw));
    auto i2_of = [&](int row) {
      const int edeg = deg_lds[row] > 0 ? exp_above((float)deg_lds[row]) : 1;
      return pow2f(clamp100(14 + edeg - a.sB));
    };

    if (wave < 5) {
      // ---- scalar-output waves: 20 weight blocks per k ((hi, lo) per (output tile n, K-step s2)); ring: half a hidden unit ahead
      // (edge strides above 32: eight builder passes per lane instead of four — their registers come out of the ring of waves 0..3,
      // whose contraction is bound by the vector work riding on it, not by the weight latency)
      constexpr int NB = 20, R = SPD == 64 ? 5 : SPD == 40 ? 10 : 10, RD = 10;  // (R divides NB: block b of every hidden unit lives in RB[b % R])
      const int w = wave;
      f32x16 accS[5];
#pragma unroll
      for (int n = 0; n < 5; ++n) accS[n] = zero16;
      auto wstream = [&](int k) { return (k * 124 + 20 * w) * 1024; };  // byte offset of this wave's stream of hidden unit k
      u32x4 RB[RD];
      auto contract = [&](const float4 (&Ah)[2], const float4 (&Al)[2], int cur, int nxt) {
#pragma unroll
        for (int n = 0; n < 5; ++n)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int p = 2 * (2 * n + s2);
            if constexpr (!(dbg & 4)) { MX(accS[n], Ah[s2], Al[s2], RB[p % RD], RB[(p + 1) % RD]); }
            if constexpr (!(dbg & 1)) {
#pragma unroll
              for (int e = 0; e < 2; ++e) RB[(p + e) % RD] = (p + e + RD < NB) ? wload(cur + (p + e + RD) * 1024) : wload(nxt + (p + e + RD - NB) * 1024);
              MF_SCHED();  // (pins the ring: left alone, the scheduler sinks each load to just before its use)
            }
          }
      };
      if (w < 4) {
        // (issue order as in the loop — h~ / T of the next hidden unit, then weight blocks — so that the wait in front of build()
        // counts the same loads on the way into the loop as around it)
        build(0);
        load_k(k_of(1));
        MF_SCHED();
        {
          const int c0 = wstream(k_of(0));
#pragma unroll
          for (int p = 0; p < R; ++p) RB[p] = wload(c0 + p * 1024);
        }
        LDS_BARRIER();  // x rows, C(k0), T(k0) complete
        SSTAMP(4);
        const int xa = MF_X0H + 32 * w * MF_ROWB + fo;
        for (int it = 0; it < nk; ++it) {
          const int cur = wstream(k_of(it)), nxt = wstream(k_of(it + 1));
          const int cb = MF_C + (it & 1) * MF_CB + fo;
          MSTAMP(0);
          float4 Ah[2], Al[2];
          if constexpr (SPD == 32) {
            Frag f0 = ldf(xa, MF_X0L - MF_X0H, cb), f1 = ldf(xa + 32, MF_X0L - MF_X0H, cb + 32);
            MF_SCHED();
            MSTAMP(1);
            f32x16 F = zero16;
            mm(F, f0);
            MF_SCHED();
            f0 = ldf(xa + 64, MF_X0L - MF_X0H, cb + 64);
            MF_SCHED();
            mm(F, f1);
            MF_SCHED();
            if constexpr (NKS == 4) {
              f1 = ldf(xa + 96, MF_X0L - MF_X0H, cb + 96);
              MF_SCHED();
            }
            mm(F, f0);
            if constexpr (NKS == 4) mm(F, f1);
            MSTAMP(2);
            split(F, Ah, Al);
          } else {  // (one fragment set: the eight builder passes need the registers; this wave is not the longest stream of its SIMD)
            MSTAMP(1);
            f32x16 F = zero16;
#pragma unroll
            for (int st = 0; st < NKS; ++st) {
              const Frag f0 = ldf(xa + 32 * st, MF_X0L - MF_X0H, cb + 32 * st);
              mm(F, f0);
              MF_SCHED();
            }
            MSTAMP(2);
            split(F, Ah, Al);
          }
          MSTAMP(3);
          // The coefficient and T tiles of the next hidden unit are built BETWEEN the MFMAs of the contraction: an MFMA holds the
          // vector issue of the SIMD for 8 of its 32 cycles, so ~5 vector instructions per MFMA ride along.  build() cut into 28 pieces
          // of 4..6 instructions (bstep), one behind each MFMA, the order pinned with scheduling barriers (the split helpers are inline
          // assembly, which scheduling GROUP barriers cannot classify).  On the last hidden unit the unused buffer is rebuilt from
          // clamped loads: straight-line code throughout.
          {
            char* __restrict__ cbuf = lds + MF_C + ((it + 1) & 1) * MF_CB;
            char* __restrict__ tbuf = lds + MF_TT + ((it + 1) & 1) * MF_TTB;
            float cc = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f, t0 = 0.f, t1 = 0.f;
            unsigned h01 = 0u, h23 = 0u, l01 = 0u, l23 = 0u, ph = 0u;
            char* __restrict__ d = cbuf;
            auto bstep = [&](int i) {
              if constexpr ((dbg & 8) != 0) return;
              if (i < 5 * NP) {
                const int p = i / 5, j = i % 5;
                if (j == 0) cc = coef(p) * scC;
                else if (j == 1) { c1 = cc * evx[p]; c2 = cc * evy[p]; c3 = cc * evz[p]; h01 = cvt_pk_f16(cc, c1); h23 = cvt_pk_f16(c2, c3); }
                else if (j == 2) { l01 = cvt_pk_f16(resid_lo(cc, h01), resid_hi(c1, h01)); l23 = cvt_pk_f16(resid_lo(c2, h23), resid_hi(c3, h23)); }
                else if (j == 3) {
                  d = cbuf + (ent[p] & 0x1fff);
                  *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h01 & 0xffffu);
                  *reinterpret_cast<unsigned short*>(d + MF_PL) = (unsigned short)(l01 & 0xffffu);
                  *reinterpret_cast<unsigned short*>(d + 2 * MF_PL) = (unsigned short)(h01 >> 16);
                  *reinterpret_cast<unsigned short*>(d + 3 * MF_PL) = (unsigned short)(l01 >> 16);
                } else {
                  *reinterpret_cast<unsigned short*>(d + 4 * MF_PL) = (unsigned short)(h23 & 0xffffu);
                  *reinterpret_cast<unsigned short*>(d + 5 * MF_PL) = (unsigned short)(l23 & 0xffffu);
                  *reinterpret_cast<unsigned short*>(d + 6 * MF_PL) = (unsigned short)(h23 >> 16);
                  *reinterpret_cast<unsigned short*>(d + 7 * MF_PL) = (unsigned short)(l23 >> 16);
                }
              } else if (i < 5 * NP + 2 * NTV) {
                const int q = (i - 5 * NP) / 2, j = (i - 5 * NP) % 2;
                const int item = tid + BT * q, wp = item >> 5, jp = item & 31, j0 = 2 * jp - off;
                if (j == 0) {
                  t0 = (j0 >= 0 && j0 < rows) ? tv[q].x * scT : 0.f;
                  t1 = (j0 + 1 >= 0 && j0 + 1 < rows) ? tv[q].y * scT : 0.f;
                  ph = cvt_pk_f16(t0, t1);
                } else {
                  const unsigned pl = cvt_pk_f16(resid_lo(t0, ph), resid_hi(t1, ph));
                  *reinterpret_cast<unsigned*>(tbuf + wp * MF_ROWB + 4 * jp) = ph;
                  *reinterpret_cast<unsigned*>(tbuf + MF_PL + wp * MF_ROWB + 4 * jp) = pl;
                }
              }
            };
            // (30 MFMA slots; with eight builder passes there are 48 pieces: the first 18 slots carry two CONSECUTIVE pieces — the pieces of
            // a pass share their temporaries and must run in order)
            auto bslot = [&](int sl) {
              constexpr int EXTRA = 5 * NP + 2 * NTV > 30 ? 5 * NP + 2 * NTV - 30 : 0;
              if (sl < EXTRA) { bstep(2 * sl); bstep(2 * sl + 1); }
              else bstep(sl + EXTRA);
            };
            MF_SCHED();
#pragma unroll
            for (int n = 0; n < 5; ++n)
#pragma unroll
              for (int s2 = 0; s2 < 2; ++s2) {
                const int g = 2 * n + s2, p = 2 * g;
                if constexpr (!(dbg & 4) && !X1) accS[n] = MFMA32H(Al[s2], RB[p % R], accS[n]);
                MF_SCHED();
                bslot(3 * g);
                MF_SCHED();
                if constexpr (!(dbg & 4) && !X1) accS[n] = MFMA32H(Ah[s2], RB[(p + 1) % R], accS[n]);
                MF_SCHED();
                bslot(3 * g + 1);
                MF_SCHED();
                if constexpr (!(dbg & 4)) accS[n] = MFMA32H(Ah[s2], RB[p % R], accS[n]);
                if constexpr (!(dbg & 1)) {
#pragma unroll
                  for (int e = 0; e < 2; ++e) RB[(p + e) % R] = (p + e + R < NB) ? wload(cur + (p + e + R) * 1024) : wload(nxt + (p + e + R - NB) * 1024);
                }
                MF_SCHED();
                bslot(3 * g + 2);
                MF_SCHED();
              }
          }
          MSTAMP(4);
          load_k(k_of(it + 2));
          MF_SCHED();
          LDS_BARRIER();
          MSTAMP(5);
#ifdef MF_TRACE
          ++trc;
#endif
        }
      } else {
        {
          const int c0 = wstream(k_of(0));
#pragma unroll
          for (int p = 0; p < RD; ++p) RB[p] = wload(c0 + p * 1024);
        }
        LDS_BARRIER();  // x rows, C(k0), T(k0) complete
        SSTAMP(4);
        // The barrier of a step sits right after this wave's LAST read of the step's coefficient tile (end of forming), not at the end
        // of its work: split and contraction need registers only and run past it, next to the scalar waves' start of the next step
        // (whose first fragment reads would otherwise leave the matrix pipe idle).  The barrier still orders both hazards: the
        // builders arrive with C(k + 1) complete, every reader arrives after its last read of C(k).
        constexpr int QD = SPD == 64 ? 2 : 3;  // fragment sets in flight (edge strides above 32: register allocation of the kernel is at its limit)
        auto fr = [&](int st, int cb) { const int m = st / NKS, s4 = st % NKS; return ldf(MF_X1H + (m * 32) * MF_ROWB + fo + 32 * s4, MF_X1L - MF_X1H, cb + (1 + m) * 2 * MF_PL + 32 * s4); };
        Frag fq[QD];
#pragma unroll
        for (int st = 0; st < QD - 1; ++st) fq[st] = fr(st, MF_C + fo);
        for (int it = 0; it < nk; ++it) {
          const int cur = wstream(k_of(it)), nxt = wstream(k_of(it + 1));
          const int cb = MF_C + (it & 1) * MF_CB + fo, cbn = MF_C + ((it + 1) & 1) * MF_CB + fo;
          MSTAMP(0);
          MSTAMP(1);
          float4 Ah[2], Al[2];
          f32x16 F = zero16;
#pragma unroll
          for (int st = 0; st < 3 * NKS; ++st) {
            if (st + QD - 1 < 3 * NKS) fq[(st + QD - 1) % QD] = fr(st + QD - 1, cb);
            MF_SCHED();
            mm(F, fq[st % QD]);
            MF_SCHED();
          }
          MSTAMP(2);
          LDS_BARRIER();
#pragma unroll
          for (int st = 0; st < QD - 1; ++st) fq[st] = fr(st, cbn);  // (first reads of the next step; unused after the last)
          MF_SCHED();
          split(F, Ah, Al);
          MSTAMP(3);
          contract(Ah, Al, cur, nxt);
          MSTAMP(4);
          MSTAMP(5);
#ifdef MF_TRACE
          ++trc;
#endif
        }
      }
      SSTAMP(5);
      float* __restrict__ ST0 = reinterpret_cast<float*>(lds);  // [5 waves][32][160] partial scalar-output tiles (every tile of the segment is dead)
#pragma unroll
      for (int q = 0; q < 16; ++q) {  // (raw accumulators: the power-of-two factors i1, i2 are applied to the SUM of the five partial tiles below)
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
#pragma unroll
        for (int n = 0; n < 5; ++n) ST0[(w * 32 + row) * 160 + 32 * n + r] = accS[n][q];
      }
    } else {
      // ---- plane waves: 8 weight blocks per k (x1 inputs, cross inputs: (hi, lo) per K-step), ring of 4
      constexpr int NB = 8, R = 8;
      const int m = wave - 5, m1 = (m + 1) % 3, m2 = (m + 2) % 3;
      f32x16 accP = zero16, accT = zero16;
      auto wstream = [&](int k) { return (k * 124 + 100 + 8 * m) * 1024; };
      u32x4 RB[R];
      const int x1a = MF_X1H + fo;  // + plane * 32 rows
      {
        const int c0 = wstream(k_of(0));
#pragma unroll
        for (int p = 0; p < R; ++p) RB[p] = wload(c0 + p * 1024);
      }
      LDS_BARRIER();  // x rows, C(k0), T(k0) complete
        SSTAMP(4);
      // steps 0..3 x1[m] C[1] | 4..7 x1[m+1] C[v_(m+2)] | 8..11 x1[m+2] C[v_(m+1)] ((x1 x v)[m] = x1[m+1] v[m+2] - x1[m+2] v[m+1]) |
      // 12..15 the T term: out_m[i][w'] += sum_j C[v_m][i][j] T_k[j][w'] (A = coefficient rows, B = T^T rows; own accumulator: scale 2^(sC + sT))
      auto fr = [&](int st, int cb, int tb) {
        const int g = st / NKS, s4 = 32 * (st % NKS);
        if (g == 0) return ldf(x1a + m * 32 * MF_ROWB + s4, MF_X1L - MF_X1H, cb + s4);
        if (g == 1) return ldf(x1a + m1 * 32 * MF_ROWB + s4, MF_X1L - MF_X1H, cb + (1 + m2) * 2 * MF_PL + s4);
        if (g == 2) return ldf(x1a + m2 * 32 * MF_ROWB + s4, MF_X1L - MF_X1H, cb + (1 + m1) * 2 * MF_PL + s4);
        return ldf(cb + (1 + m) * 2 * MF_PL + s4, MF_PL, tb + s4);
      };
      // (the step's barrier sits after the T term — this wave's last read of the step's tiles — and the contraction, registers only,
      // runs past it with the first fragments of the next step already requested: see wave 4)
      Frag fq[3];
      fq[0] = fr(0, MF_C + fo, MF_TT + fo); fq[1] = fr(1, MF_C + fo, MF_TT + fo);
      for (int it = 0; it < nk; ++it) {
        const int cur = wstream(k_of(it)), nxt = wstream(k_of(it + 1));
        const int cb = MF_C + (it & 1) * MF_CB + fo, cbn = MF_C + ((it + 1) & 1) * MF_CB + fo;
        const int tb = MF_TT + (it & 1) * MF_TTB + fo, tbn = MF_TT + ((it + 1) & 1) * MF_TTB + fo;
        MSTAMP(0);
        MSTAMP(1);
        f32x16 FA = zero16, F1 = zero16, F2 = zero16;
#pragma unroll
        for (int st = 0; st < 3 * NKS; ++st) {
          fq[(st + 2) % 3] = fr(st + 2, cb, tb);
          MF_SCHED();
          mm(st < NKS ? FA : st < 2 * NKS ? F1 : F2, fq[st % 3]);
          MF_SCHED();
          if (st == NKS - 1) MSTAMP(2);
        }
        // The 12 MFMAs of the T term (steps 12..15; their own accumulator) carry the vector work of BOTH splits between them — 16 pairs
        // of values (x1[m] tile, then the cross-product tile F1 - F2), one or two pairs behind each MFMA, pinned with scheduling
        // barriers: left in program order the two splits are ~130 vector instructions with an idle matrix pipe.
        unsigned ahA[2][4], alA[2][4], ahC[2][4], alC[2][4];
        auto split_pair = [&](int i) {  // i = 0..7: x1[m] tile, 8..15: cross-product tile; pair (s2, p2) of the tile
          const int s2 = (i >> 2) & 1, p2 = i & 3, q = 8 * s2 + 2 * p2;
          if constexpr ((dbg & 16) != 0) { ahA[s2][p2] = alA[s2][p2] = ahC[s2][p2] = alC[s2][p2] = 0u; return; }
          if (i < 8) {
            const float v0 = FA[q] * rs, v1 = FA[q + 1] * rs;
            const unsigned ph = cvt_pk_f16(v0, v1);
            ahA[s2][p2] = ph;
            alA[s2][p2] = cvt_pk_f16(resid_lo(v0, ph), resid_hi(v1, ph));
          } else {
            const float v0 = (F1[q] - F2[q]) * rs, v1 = (F1[q + 1] - F2[q + 1]) * rs;
            const unsigned ph = cvt_pk_f16(v0, v1);
            ahC[s2][p2] = ph;
            alC[s2][p2] = cvt_pk_f16(resid_lo(v0, ph), resid_hi(v1, ph));
          }
        };
        // (MFMA slot i of the 3 NKS carries pairs [16 i / (3 NKS), 16 (i + 1) / (3 NKS)): 1 + 1 + 2 per step for four steps)
        auto ride = [&](int slot) {
#pragma unroll
          for (int i = 16 * slot / (3 * NKS); i < 16 * (slot + 1) / (3 * NKS); ++i) split_pair(i);
        };
#pragma unroll
        for (int st = 3 * NKS; st < 4 * NKS; ++st) {
          if (st + 2 < 4 * NKS) fq[(st + 2) % 3] = fr(st + 2, cb, tb);
          MF_SCHED();
          const Frag& f = fq[st % 3];
          const int e = 3 * (st - 3 * NKS);
          if constexpr (!(dbg & 2) && !X1) accT = MFMA32H(f.al, f.bh, accT);
          MF_SCHED();
          ride(e);
          MF_SCHED();
          if constexpr (!(dbg & 2) && !X1) accT = MFMA32H(f.ah, f.bl, accT);
          MF_SCHED();
          ride(e + 1);
          MF_SCHED();
          if constexpr (!(dbg & 2)) accT = MFMA32H(f.ah, f.bh, accT);
          MF_SCHED();
          ride(e + 2);
          MF_SCHED();
        }
        MSTAMP(3);
        LDS_BARRIER();
        fq[0] = fr(0, cbn, tbn); fq[1] = fr(1, cbn, tbn);  // (unused after the last step)
        MF_SCHED();
        auto f4 = [](const unsigned (&v)[4]) { return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])); };
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // x1 inputs (blocks 0..3), then cross inputs (4..7): (hi, lo) per K-step
          const int s2 = g & 1, p = 2 * g;
          const float4 Ah_ = g < 2 ? f4(ahA[s2]) : f4(ahC[s2]), Al_ = g < 2 ? f4(alA[s2]) : f4(alC[s2]);
          if constexpr (!(dbg & 4)) { MX(accP, Ah_, Al_, RB[p % R], RB[(p + 1) % R]); }
          if constexpr (!(dbg & 1)) {
#pragma unroll
            for (int e = 0; e < 2; ++e) RB[(p + e) % R] = (p + e + R < NB) ? wload(cur + (p + e + R) * 1024) : wload(nxt + (p + e + R - NB) * 1024);
            MF_SCHED();
          }
        }
        MSTAMP(4);
        MSTAMP(5);
#ifdef MF_TRACE
        ++trc;
#endif
      }
      SSTAMP(5);
      float* __restrict__ ST1 = reinterpret_cast<float*>(lds + 5 * 32 * 160 * 4);  // [2][32][96] vector planes: contraction, T term (raw)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        ST1[row * 96 + m * 32 + r] = accP[q];
        ST1[3072 + row * 96 + m * 32 + r] = accT[q];
      }
    }
    // ---- all threads: sum of the five K-partial tiles -> partial slab of this segment, coalesced 16-byte stores
    sg0 = nsg0; sg1 = nsg1;
    t_at = make_int2(sg1.y, sg1.z & 255); span = make_int2(sg1.w, sg1.w + (sg1.z >> 8));
    // (the inverse column scales of this thread's slab pieces: requested in FRONT of the barrier — as loads at their points of use they were a
    // global round trip of ~2 k cycles inside every segment's epilogue, round 6)
    constexpr int NE0 = (32 * 40 + MF_THREADS - 1) / MF_THREADS, NE1 = (32 * 24 + MF_THREADS - 1) / MF_THREADS;
    float4 cfa[NE0], cfb[NE1];
#pragma unroll
    for (int j = 0; j < NE0; ++j) cfa[j] = reinterpret_cast<const float4*>(a.cf0)[(tid + j * MF_THREADS) % 40];
#pragma unroll
    for (int j = 0; j < NE1; ++j) cfb[j] = reinterpret_cast<const float4*>(a.cf1)[((tid + j * MF_THREADS) % 24) & 7];
    LDS_BARRIER();
    SSTAMP(6);
    {
      const float* __restrict__ ST0 = reinterpret_cast<const float*>(lds);
      const float* __restrict__ ST1 = reinterpret_cast<const float*>(lds + 5 * 32 * 160 * 4);
      float* __restrict__ p0 = a.partial0 + ((size_t)slab * a.n_pad + n0) * (size_t)(a.nt0 * 32);
      float* __restrict__ p1 = a.partial1 + ((size_t)slab * a.n_pad + n0) * 96;
#pragma unroll
      for (int j = 0; j < NE0; ++j) {
        const int idx = tid + j * MF_THREADS;
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
        const int idx = tid + j * MF_THREADS;
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
    LDS_BARRIER();  // the next segment rewrites the tiles
    SSTAMP(7);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// k_conv_mfi — the INITIAL projector on the same scheme.  Its inputs are atom embeddings: a handful of distinct rows (U <= 32 per
// batch), so the input-times-weight table Tab_k[uid][w] is static (host: jamun_api.cpp, as for jamun_conv_init.hip) and the layer is
//   m[i][w] = sum_k sum_uid G_k[c][i][uid] Tab_k[uid][w],   G_k[c][i][uid] = sum_{j -> i, uid_j = uid} h~_(j->i)[k] (1, vx, vy, vz)_c
// (component 0 for the 152 scalar outputs, v_m for vector plane m).  G_k is FORMED like the hidden layers' A tiles, with the one-hot
// selector S[uid][j] = (uid_j == uid) in place of the feature rows (exact in f16: two MFMAs per K-step, C hi and C lo), split in
// registers and contracted with the table blocks streamed from L2: per (tile, k) 8 x (8 + 6) MFMAs against 414 of a hidden layer.
// Eight equal waves: wave r < 5 owns scalar-output tile r, wave 5 + m vector plane m; every wave forms the G it needs itself (8
// MFMAs: cheaper than sharing it), holds ONE 32 x 32 accumulator for the whole segment and stores it straight into the slab.
// All eight waves build the coefficient tiles of the next hidden unit (same code as k_conv_mf, no T tile).
// UT: tiles of 32 distinct embedding rows (1, 2 or 4)
template <int SPD, int UT>
__global__ __launch_bounds__(MF_THREADS) void k_conv_mfi(MfiArgs a) {
  extern __shared__ float4 lds4[];
  char* __restrict__ lds = reinterpret_cast<char*>(lds4);
  const int L0 = lds_addr(lds);
  // LDS: selector [32 uid][64 j] halves (one plane) | coefficient tiles as k_conv_mf | deg
  constexpr int I_S = 0, I_C = UT * 32 * MF_ROWB, I_MISC = I_C + 2 * MF_CB;
  int* __restrict__ deg_lds = reinterpret_cast<int*>(lds + I_MISC);
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = RFL(tid0 >> 6);
  // (the record of segment s + 1 is requested during segment s, as in k_conv_mf: one dependent round trip less per segment but the first)
  const int4* __restrict__ my_segs = a.segs + (size_t)blockIdx.x * a.max_segs * 2;
  int4 sg0 = ld_const(my_segs), sg1 = ld_const(my_segs + 1);
  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    int4 nsg0 = make_int4(-1, 0, 0, 0), nsg1 = make_int4(-1, 0, 0, 0);
    if (sgi + 1 < a.max_segs) { nsg0 = ld_const(my_segs + 2 * (sgi + 1)); nsg1 = ld_const(my_segs + 2 * (sgi + 1) + 1); }
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) { kk = kk < 0 ? 0 : (kk < nk ? kk : nk - 1); return kk < k_run ? k_begin + kk : k_extra; };
    const int2 t_at = make_int2(sg1.y, sg1.z & 255);  // (the record carries its tile's descriptor, jamun_api.cpp: embed — one dependent round trip less per segment)
    const int n0 = RFL(t_at.x), n_dst = RFL(t_at.y);
    const int2 span = make_int2(sg1.w, sg1.w + (sg1.z >> 8));
    const int s_lo = RFL(span.x), rows = RFL(span.y) - s_lo;
    const int s_base = s_lo;

    // ---- everything the builder needs from global memory is requested first (in-degrees, h~ of the first hidden unit, the edge records
    // of all passes without the degree predicate — slots past it hold stale records and are masked below), so that these round trips
    // and the selector's overlap instead of following one another.  Every request is UNCONDITIONAL at a clamped address and masked after
    // (round 6: as `in ? a.esrc[slot] : 0` each load was a branch, and the compiler put `s_waitcnt vmcnt(0)` into some of the branches —
    // five serialised round trips in front of every tile of this kernel)
    const int dgl = a.deg[n0 + min(tid & 31, n_dst - 1)];
    const int dgv = dgl * ((tid < 32 && tid < n_dst) ? 1 : 0);  // (a product, not a select: a select's load is sunk into a branch, with a wait of its own)
    constexpr int BT = MF_THREADS, DPP = BT / SPD, NP = 32 / DPP;
    int ent[NP];
    const int slot0 = (n0 + tid / SPD) * a.S + tid % SPD, pstride = DPP * a.S;
    const int tcl = min(tid % SPD, a.S - 1);  // (clamped slot of the records: a destination of the tile, a slot below the stride)
    auto slot_c = [&](int p) { return (n0 + min((tid + BT * p) / SPD, n_dst - 1)) * a.S + tcl; };
    float hv[NP];
    // (every lane loads the h~ of its OWN slot; the owner of a coefficient entry fetches the pair's other edges by lane shuffle, as in k_conv_mf)
    auto load_k = [&](int k) {
      const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
      for (int p = 0; p < NP; ++p) hv[p] = hk[slot0 + p * pstride];
    };
    load_k(k_of(0));
    float evx[NP], evy[NP], evz[NP];
    int sjv[NP], epv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int g = tid + BT * p, i = g / SPD, t = g % SPD;
      const bool in = i < n_dst && t < a.S;
      const int sc_ = slot_c(p);
      const int sj_ = a.esrc[sc_], ep_ = a.epair[sc_];
      const float4 ge = a.egeo[sc_];
      const float kin = in ? 1.f : 0.f;
      sjv[p] = in ? sj_ : 0;
      epv[p] = in ? ep_ : 0;
      evx[p] = ge.x * kin; evy[p] = ge.y * kin; evz[p] = ge.z * kin;
    }
    // (the embedding-row ids of the span: a thread's pair of source rows is the same in every round of the selector loop below)
    const int jp_u = tid & 31;
    const int uu0 = a.atom_uid[s_lo + min(2 * jp_u, rows - 1)], uu1 = a.atom_uid[s_lo + min(2 * jp_u + 1, rows - 1)];
    const int u0 = 2 * jp_u < rows ? uu0 : -1, u1 = 2 * jp_u + 1 < rows ? uu1 : -1;

    // ---- segment prologue: zero the coefficient tiles, the selector
    for (int idx = tid; idx < 2 * MF_CB / 16; idx += MF_THREADS) reinterpret_cast<float4*>(lds + I_C)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int idx = tid; idx < UT * 32 * 32; idx += MF_THREADS) {  // (uid, pair of source rows; idx & 31 == tid & 31: MF_THREADS is a multiple of 32)
      const int uid = idx >> 5, jp = idx & 31;
      *reinterpret_cast<unsigned*>(lds + I_S + uid * MF_ROWB + 4 * jp) = (u0 == uid ? 0x3c00u : 0u) | (u1 == uid ? 0x3c000000u : 0u);
    }
    if (tid < 32) deg_lds[tid] = dgv;
    LDS_BARRIER();

    // ---- builder state (all eight waves): as k_conv_mf
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int g = tid + BT * p, i = g / SPD, t = g % SPD;
      const int dg = deg_lds[i];
      const bool in = t < dg && t < a.S;
      const int sj = sjv[p], ep = in ? epv[p] : 0;
      const int jl = (sj & 0x7fffffff) - s_base;  // (bit 31: bonded)
      const bool valid = in && jl >= 0 && jl < 64;
      // (edges of one ordered pair share one entry: matched by k_geom — MfArgs::epair, JAMUN_EP_* — as in k_conv_mf)
      if (ep & JAMUN_EP_OVERFLOW) atomicOr(a.err, 1);
      const int pa = (ep & 127) - 1, pb = ((ep >> 7) & 127) - 1;
      const int d0 = pa >= 0 ? pa - t : 0, d1 = pb >= 0 ? pb - t : 0;  // lane distance to the pair's other edges (same lane group)
      const bool active = valid && !((unsigned)ep & JAMUN_EP_OWNED);
      ent[p] = active ? (i * MF_ROWB + 2 * jl) | (d0 << 13) | (d1 << 19) : 128;
    }
    const float scC = pow2f(a.sC);
    auto coef = [&](int p) {  // h~ of pass p's entry: this lane's edge + the pair's other edges
      const int d0 = (ent[p] >> 13) & 63, d1 = (ent[p] >> 19) & 63;
      const float t0 = __shfl(hv[p], lane + d0, 64), t1 = __shfl(hv[p], lane + d1, 64);
      return (hv[p] + (d0 ? t0 : 0.f)) + (d1 ? t1 : 0.f);
    };
    auto build = [&](int buf) {
      char* __restrict__ cbuf = lds + I_C + buf * MF_CB;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float c0 = coef(p) * scC, c1 = c0 * evx[p], c2 = c0 * evy[p], c3 = c0 * evz[p];
        const unsigned h01 = cvt_pk_f16(c0, c1), h23 = cvt_pk_f16(c2, c3);
        const unsigned l01 = cvt_pk_f16(resid_lo(c0, h01), resid_hi(c1, h01)), l23 = cvt_pk_f16(resid_lo(c2, h23), resid_hi(c3, h23));
        char* __restrict__ d = cbuf + (ent[p] & 0x1fff);
        *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h01 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + MF_PL) = (unsigned short)(l01 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 2 * MF_PL) = (unsigned short)(h01 >> 16);
        *reinterpret_cast<unsigned short*>(d + 3 * MF_PL) = (unsigned short)(l01 >> 16);
        *reinterpret_cast<unsigned short*>(d + 4 * MF_PL) = (unsigned short)(h23 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 5 * MF_PL) = (unsigned short)(l23 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 6 * MF_PL) = (unsigned short)(h23 >> 16);
        *reinterpret_cast<unsigned short*>(d + 7 * MF_PL) = (unsigned short)(l23 >> 16);
      }
    };

    // ---- matrix work: role = wave
    const int r = lane & 31, hh = lane >> 5;
    const int comp = wave < 5 ? 0 : wave - 4;  // coefficient component this wave contracts
    const int edeg_r = deg_lds[r] > 0 ? exp_above((float)deg_lds[r]) : 1;
    const float rs = pow2f(-edeg_r);            // formed values (a sum of <= in-degree coefficients below 2^14) -> below 2^14
    const int fo = r * MF_ROWB + 16 * hh + L0;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.tabw), 0, 0x7fffffff, 0x00020000);
    const int wvo = lane * 16;
    auto wload = [&](int so) { return __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so, 0); };
    constexpr int NBW = 4 * UT;  // blocks per role and hidden unit: (hi, lo) x 2 UT K-steps of 16 uids
    auto wstream = [&](int k) { return (k * 6 + (wave < 5 ? wave : 5)) * NBW * 1024; };
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    u32x4 RB[NBW];
    build(0);
    load_k(k_of(1));
    __builtin_amdgcn_sched_barrier(0);
    {
      const int c0 = wstream(k_of(0));
#pragma unroll
      for (int p = 0; p < NBW; ++p) RB[p] = wload(c0 + p * 1024);
    }
    LDS_BARRIER();  // selector, C(k0) complete
    // (the selector does not depend on the hidden unit: its fragments are read once per segment, not once per step)
    float4 sa[UT][4];
#pragma unroll
    for (int ut = 0; ut < UT; ++ut)
#pragma unroll
      for (int s = 0; s < 4; ++s) sa[ut][s] = lds_f4(I_S + ut * 32 * MF_ROWB + fo + 32 * s);
    for (int it = 0; it < nk; ++it) {
      const int nxt = wstream(k_of(it + 1));
      const int cb = I_C + (it & 1) * MF_CB + comp * 2 * MF_PL + fo;
      // G^T[uid][i] = sum_j S[uid][j] C[i][j]: A = selector rows (exact halves), B = coefficient rows hi, lo
      float4 bh[4], bl[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { bh[s] = lds_f4(cb + 32 * s); bl[s] = lds_f4(cb + MF_PL + 32 * s); }
#pragma unroll
      for (int ut = 0; ut < UT; ++ut) {
        f32x16 G;
#pragma unroll
        for (int q = 0; q < 16; ++q) G[q] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) { G = MFMA32H(sa[ut][s], bl[s], G); G = MFMA32H(sa[ut][s], bh[s], G); }
        float4 Ah[2], Al[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          unsigned ph[4], pl[4];
#pragma unroll
          for (int p2 = 0; p2 < 4; ++p2) {
            const float v0 = G[8 * s2 + 2 * p2] * rs, v1 = G[8 * s2 + 2 * p2 + 1] * rs;
            ph[p2] = cvt_pk_f16(v0, v1);
            pl[p2] = cvt_pk_f16(resid_lo(v0, ph[p2]), resid_hi(v1, ph[p2]));
          }
          Ah[s2] = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
          Al[s2] = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int b = 4 * ut + 2 * s2;
          M3(acc, Ah[s2], Al[s2], RB[b], RB[b + 1]);
          RB[b] = wload(nxt + b * 1024);
          RB[b + 1] = wload(nxt + (b + 1) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      build((it + 1) & 1);  // (on the last hidden unit: the unused buffer, from clamped loads)
      load_k(k_of(it + 2));
      __builtin_amdgcn_sched_barrier(0);
      LDS_BARRIER();
    }
    // ---- segment end: this wave's tile straight into the slab (rows carry 2^(sC - e(in-degree) + sB))
    {
      const float i1 = pow2f(clamp100(-(a.sC + a.sB)));
      float* __restrict__ p0 = a.partial0 + ((size_t)slab * a.n_pad + n0) * (size_t)(a.nt0 * 32);
      float* __restrict__ p1 = a.partial1 + ((size_t)slab * a.n_pad + n0) * 96;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        const int edeg = deg_lds[row] > 0 ? exp_above((float)deg_lds[row]) : 1;
        const float v = (acc[q] * i1) * pow2f(edeg);
        if (row < n_dst) {
          if (wave < 5) p0[row * 160 + 32 * wave + r] = v;
          else p1[row * 96 + (wave - 5) * 32 + r] = v;
        }
      }
    }
    LDS_BARRIER();  // the next segment rewrites the tiles
    sg0 = nsg0; sg1 = nsg1;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// k_conv_mfx — the initial projector for batches with MANY distinct embedding rows (k_conv_mfi's one-hot selector costs 8 + 6 MFMAs per
// wave, hidden unit and 32 distinct rows: 143 rows — every realistic peptide batch: 20 residue types x their heavy-atom names — would be
// 5 selector tiles).  Here the layer runs like a hidden layer with 64 scalar input channels (the 56 noise-scaled embedding channels,
// padded): the aggregated operand is FORMED from the feature rows,
//   A_k[(i,c)][u] = sum_j C_k[c][i][j] x_j[u]     (c = 0: scalar outputs;  c = 1 + m: x0 (x) v_m -> vector plane m),
// 4 components x 2 channel tiles of 32 = 8 formed tiles per (tile, k), one per wave (wave = 2 c + t), 12 MFMAs each, split in
// registers and contracted with weight blocks streamed from L2: waves 0, 1 into the five scalar-output tiles (K-split over the two
// channel tiles: 30 MFMAs each), waves 2 + 2 m + t into vector plane m (6 MFMAs); 192 MFMAs per (tile, k) whatever the number of
// distinct rows.  The feature rows are constant per (topology, sigma): the host balances the weights per channel and column as for
// the hidden layers, folds the channel factors and ONE static scale 2^sX into the rows and stores them split (hi / lo halves, two atoms
// per word): staging is a copy.  All eight waves build the coefficient tiles of the next hidden unit (as k_conv_mfi).
template <int SPD>
__global__ __launch_bounds__(MF_THREADS) void k_conv_mfx(MfxArgs a) {
  extern __shared__ float4 lds4[];
  char* __restrict__ lds = reinterpret_cast<char*>(lds4);
  const int L0 = lds_addr(lds);
  constexpr int X_H = 0, X_L = 64 * MF_ROWB, X_C = 2 * 64 * MF_ROWB, X_MISC = X_C + 2 * MF_CB;  // x^T hi | lo ([64 ch][64 j]) | coefficient tiles | deg
  int* __restrict__ deg_lds = reinterpret_cast<int*>(lds + X_MISC);
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = RFL(tid0 >> 6);
  const int4* __restrict__ my_segs = a.segs + (size_t)blockIdx.x * a.max_segs * 2;  // (next record requested during this segment, as k_conv_mfi)
  int4 sg0 = ld_const(my_segs), sg1 = ld_const(my_segs + 1);
  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    int4 nsg0 = make_int4(-1, 0, 0, 0), nsg1 = make_int4(-1, 0, 0, 0);
    if (sgi + 1 < a.max_segs) { nsg0 = ld_const(my_segs + 2 * (sgi + 1)); nsg1 = ld_const(my_segs + 2 * (sgi + 1) + 1); }
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) { kk = kk < 0 ? 0 : (kk < nk ? kk : nk - 1); return kk < k_run ? k_begin + kk : k_extra; };
    const int2 t_at = make_int2(sg1.y, sg1.z & 255);  // (as k_conv_mfi: descriptor inside the segment record)
    const int n0 = RFL(t_at.x), n_dst = RFL(t_at.y);
    const int2 span = make_int2(sg1.w, sg1.w + (sg1.z >> 8));
    const int s_base = RFL(span.x) & ~1;  // window of 64 source rows from an even atom (the rows are stored in pairs)

    // ---- loads first: in-degrees, h~ of the first hidden unit, the edge records of all passes (as k_conv_mfi: unconditional, clamped, masked)
    const int dgl = a.deg[n0 + min(tid & 31, n_dst - 1)];
    const int dgv = dgl * ((tid < 32 && tid < n_dst) ? 1 : 0);
    constexpr int BT = MF_THREADS, DPP = BT / SPD, NP = 32 / DPP;
    int ent[NP];
    const int slot0 = (n0 + tid / SPD) * a.S + tid % SPD, pstride = DPP * a.S;
    const int tcl = min(tid % SPD, a.S - 1);
    auto slot_c = [&](int p) { return (n0 + min((tid + BT * p) / SPD, n_dst - 1)) * a.S + tcl; };
    float hv[NP];
    auto load_k = [&](int k) {
      const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
      for (int p = 0; p < NP; ++p) hv[p] = hk[slot0 + p * pstride];
    };
    load_k(k_of(0));
    float evx[NP], evy[NP], evz[NP];
    int sjv[NP], epv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int g = tid + BT * p, i = g / SPD, t = g % SPD;
      const bool in = i < n_dst && t < a.S;
      const int sc_ = slot_c(p);
      const int sj_ = a.esrc[sc_], ep_ = a.epair[sc_];
      const float4 ge = a.egeo[sc_];
      const float kin = in ? 1.f : 0.f;
      sjv[p] = in ? sj_ : 0;
      epv[p] = in ? ep_ : 0;
      evx[p] = ge.x * kin; evy[p] = ge.y * kin; evz[p] = ge.z * kin;
    }
    // ---- segment prologue: zero the coefficient tiles; the window's rows (already split, two atoms per word): a copy, transposed
    for (int idx = tid; idx < 2 * MF_CB / 16; idx += MF_THREADS) reinterpret_cast<float4*>(lds + X_C)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      const unsigned* __restrict__ ph = a.xph + (size_t)(s_base >> 1) * 64;
      const unsigned* __restrict__ pl = a.xpl + (size_t)(s_base >> 1) * 64;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // 32 pairs x 64 channels over 512 threads
        const int idx = tid + MF_THREADS * q, ch = idx & 63, jp = idx >> 6;
        *reinterpret_cast<unsigned*>(lds + X_H + ch * MF_ROWB + 4 * jp) = ph[idx];
        *reinterpret_cast<unsigned*>(lds + X_L + ch * MF_ROWB + 4 * jp) = pl[idx];
      }
    }
    if (tid < 32) deg_lds[tid] = dgv;
    LDS_BARRIER();

    // ---- builder state (all eight waves): as k_conv_mfi
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int g = tid + BT * p, i = g / SPD, t = g % SPD;
      const int dg = deg_lds[i];
      const bool in = t < dg && t < a.S;
      const int sj = sjv[p], ep = in ? epv[p] : 0;
      const int jl = (sj & 0x7fffffff) - s_base;  // (bit 31: bonded)
      const bool valid = in && jl >= 0 && jl < 64;
      // (edges of one ordered pair share one entry: matched by k_geom — MfArgs::epair, JAMUN_EP_* — as in k_conv_mf)
      if (ep & JAMUN_EP_OVERFLOW) atomicOr(a.err, 1);
      const int pa = (ep & 127) - 1, pb = ((ep >> 7) & 127) - 1;
      const int d0 = pa >= 0 ? pa - t : 0, d1 = pb >= 0 ? pb - t : 0;  // lane distance to the pair's other edges (same lane group)
      const bool active = valid && !((unsigned)ep & JAMUN_EP_OWNED);
      ent[p] = active ? (i * MF_ROWB + 2 * jl) | (d0 << 13) | (d1 << 19) : 128;
    }
    const float scC = pow2f(a.sC);
    auto coef = [&](int p) {
      const int d0 = (ent[p] >> 13) & 63, d1 = (ent[p] >> 19) & 63;
      const float t0 = __shfl(hv[p], lane + d0, 64), t1 = __shfl(hv[p], lane + d1, 64);
      return (hv[p] + (d0 ? t0 : 0.f)) + (d1 ? t1 : 0.f);
    };
    auto build = [&](int buf) {
      char* __restrict__ cbuf = lds + X_C + buf * MF_CB;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const float c0 = coef(p) * scC, c1 = c0 * evx[p], c2 = c0 * evy[p], c3 = c0 * evz[p];
        const unsigned h01 = cvt_pk_f16(c0, c1), h23 = cvt_pk_f16(c2, c3);
        const unsigned l01 = cvt_pk_f16(resid_lo(c0, h01), resid_hi(c1, h01)), l23 = cvt_pk_f16(resid_lo(c2, h23), resid_hi(c3, h23));
        char* __restrict__ d = cbuf + (ent[p] & 0x1fff);
        *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h01 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + MF_PL) = (unsigned short)(l01 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 2 * MF_PL) = (unsigned short)(h01 >> 16);
        *reinterpret_cast<unsigned short*>(d + 3 * MF_PL) = (unsigned short)(l01 >> 16);
        *reinterpret_cast<unsigned short*>(d + 4 * MF_PL) = (unsigned short)(h23 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 5 * MF_PL) = (unsigned short)(l23 & 0xffffu);
        *reinterpret_cast<unsigned short*>(d + 6 * MF_PL) = (unsigned short)(h23 >> 16);
        *reinterpret_cast<unsigned short*>(d + 7 * MF_PL) = (unsigned short)(l23 >> 16);
      }
    };

    // ---- matrix work: wave = 2 c + t: coefficient component c, channel tile t
    const int r = lane & 31, hh = lane >> 5;
    const int comp = wave >> 1, ct = wave & 1;
    const int edeg_r = deg_lds[r] > 0 ? exp_above((float)deg_lds[r]) : 1;
    const float rs = pow2f(-14 - edeg_r);  // formed values are below in-degree x 2^28 -> below 2^14
    const int fo = r * MF_ROWB + 16 * hh + L0;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.wx), 0, 0x7fffffff, 0x00020000);
    const int wvo = lane * 16;
    auto wload = [&](int so) { return __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo, so, 0); };
    // the wave's x^T fragments do not depend on the hidden unit: read once per segment (A operand: channels 32 t + r, K-steps of 16 rows)
    float4 xh[4], xl[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { xh[s4] = lds_f4(X_H + 32 * ct * MF_ROWB + fo + 32 * s4); xl[s4] = lds_f4(X_L + 32 * ct * MF_ROWB + fo + 32 * s4); }
    auto form_split = [&](int cb, float4 (&Ah)[2], float4 (&Al)[2]) {
      f32x16 F;
#pragma unroll
      for (int q = 0; q < 16; ++q) F[q] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float4 bh = lds_f4(cb + 32 * s4), bl = lds_f4(cb + MF_PL + 32 * s4);
        M3(F, xh[s4], xl[s4], bh, bl);
      }
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
    build(0);
    load_k(k_of(1));
    __builtin_amdgcn_sched_barrier(0);
    float* __restrict__ ST0 = reinterpret_cast<float*>(lds);                   // [2 channel tiles][32][160] partial scalar-output tiles
    float* __restrict__ ST1 = reinterpret_cast<float*>(lds + 2 * 32 * 160 * 4);  // [2][32][96] partial vector planes
    if (comp == 0) {
      // scalar outputs: 20 weight blocks per hidden unit ((hi, lo) per (output tile n, K-step s2)), ring of 10
      constexpr int NB = 20, R = 10;
      f32x16 accS[5];
#pragma unroll
      for (int n = 0; n < 5; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) accS[n][q] = 0.f;
      auto wstream = [&](int k) { return (k * 48 + 20 * ct) * 1024; };
      u32x4 RB[R];
      {
        const int c0 = wstream(k_of(0));
#pragma unroll
        for (int p = 0; p < R; ++p) RB[p] = wload(c0 + p * 1024);
      }
      LDS_BARRIER();  // C(k0) complete
      for (int it = 0; it < nk; ++it) {
        const int cur = wstream(k_of(it)), nxt = wstream(k_of(it + 1));
        float4 Ah[2], Al[2];
        form_split(X_C + (it & 1) * MF_CB + fo, Ah, Al);
#pragma unroll
        for (int n = 0; n < 5; ++n)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int p = 2 * (2 * n + s2);
            M3(accS[n], Ah[s2], Al[s2], RB[p % R], RB[(p + 1) % R]);
#pragma unroll
            for (int e = 0; e < 2; ++e) RB[(p + e) % R] = (p + e + R < NB) ? wload(cur + (p + e + R) * 1024) : wload(nxt + (p + e + R - NB) * 1024);
            __builtin_amdgcn_sched_barrier(0);
          }
        build((it + 1) & 1);
        load_k(k_of(it + 2));
        __builtin_amdgcn_sched_barrier(0);
        LDS_BARRIER();
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
#pragma unroll
        for (int n = 0; n < 5; ++n) ST0[(ct * 32 + row) * 160 + 32 * n + r] = accS[n][q];
      }
    } else {
      // vector plane m = comp - 1: 4 weight blocks per hidden unit ((hi, lo) per K-step), the next unit's requested as they are used
      f32x16 accP;
#pragma unroll
      for (int q = 0; q < 16; ++q) accP[q] = 0.f;
      auto wstream = [&](int k) { return (k * 48 + 40 + 4 * ct) * 1024; };
      u32x4 RB[4];
      {
        const int c0 = wstream(k_of(0));
#pragma unroll
        for (int p = 0; p < 4; ++p) RB[p] = wload(c0 + p * 1024);
      }
      LDS_BARRIER();
      for (int it = 0; it < nk; ++it) {
        const int nxt = wstream(k_of(it + 1));
        float4 Ah[2], Al[2];
        form_split(X_C + (it & 1) * MF_CB + comp * 2 * MF_PL + fo, Ah, Al);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          M3(accP, Ah[s2], Al[s2], RB[2 * s2], RB[2 * s2 + 1]);
          RB[2 * s2] = wload(nxt + (2 * s2) * 1024);
          RB[2 * s2 + 1] = wload(nxt + (2 * s2 + 1) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
        build((it + 1) & 1);
        load_k(k_of(it + 2));
        __builtin_amdgcn_sched_barrier(0);
        LDS_BARRIER();
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        ST1[(ct * 32 + row) * 96 + (comp - 1) * 32 + r] = accP[q];
      }
    }
    // (the column factors of this thread's pieces of the slab, requested in front of the barrier: as loads at their points of use every
    // round of the two loops below was a round trip of its own — five per segment; as k_conv_mf's cfa / cfb)
    float4 cfa[3], cfb[2];
#pragma unroll
    for (int n = 0; n < 3; ++n) cfa[n] = reinterpret_cast<const float4*>(a.cf0)[(tid + MF_THREADS * n) % 40];
#pragma unroll
    for (int n = 0; n < 2; ++n) cfb[n] = reinterpret_cast<const float4*>(a.cf1)[((tid + MF_THREADS * n) % 24) & 7];
    // (the staging tiles alias the x^T rows and the first coefficient buffer: every wave left the loop through its last barrier, after
    // which nothing reads those; the stores above of a fast wave may precede a slow wave's last barrier only by program order of ITS
    // OWN reads, all finished — the k loop ends with a barrier on every path)
    LDS_BARRIER();
    {
      const float i1 = pow2f(clamp100(-(a.sX + a.sC)));
      auto i2_of = [&](int row) {
        const int edeg = deg_lds[row] > 0 ? exp_above((float)deg_lds[row]) : 1;
        return pow2f(14 + edeg);
      };
      float* __restrict__ p0 = a.partial0 + ((size_t)slab * a.n_pad + n0) * (size_t)(a.nt0 * 32);
      float* __restrict__ p1 = a.partial1 + ((size_t)slab * a.n_pad + n0) * 96;
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const int idx = tid + MF_THREADS * n;
        if (idx >= 32 * 40) break;
        const int row = idx / 40, c4 = idx - row * 40;
        const float4 a0 = *reinterpret_cast<const float4*>(ST0 + row * 160 + 4 * c4), a1 = *reinterpret_cast<const float4*>(ST0 + 5120 + row * 160 + 4 * c4);
        const float4 cf = cfa[n];
        const float i2 = i2_of(row);
        if (row < n_dst)
          *reinterpret_cast<float4*>(p0 + row * 160 + 4 * c4) = make_float4((((a0.x + a1.x) * i1) * i2) * cf.x, (((a0.y + a1.y) * i1) * i2) * cf.y,
                                                                             (((a0.z + a1.z) * i1) * i2) * cf.z, (((a0.w + a1.w) * i1) * i2) * cf.w);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int idx = tid + MF_THREADS * n;
        if (idx >= 32 * 24) break;
        const int row = idx / 24, c4 = idx - row * 24;
        const float4 a0 = *reinterpret_cast<const float4*>(ST1 + row * 96 + 4 * c4), a1 = *reinterpret_cast<const float4*>(ST1 + 3072 + row * 96 + 4 * c4);
        const float4 cf = cfb[n];
        const float i2 = i2_of(row);
        if (row < n_dst)
          *reinterpret_cast<float4*>(p1 + row * 96 + 4 * c4) = make_float4((((a0.x + a1.x) * i1) * i2) * cf.x, (((a0.y + a1.y) * i1) * i2) * cf.y,
                                                                            (((a0.z + a1.z) * i1) * i2) * cf.z, (((a0.w + a1.w) * i1) * i2) * cf.w);
      }
    }
    LDS_BARRIER();  // the next segment rewrites the tiles
    sg0 = nsg0; sg1 = nsg1;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Tail tiles.  A tile with FEW destinations (a 33-atom molecule cuts into 32 + 1) costs k_conv_mf a whole tile: 414 MFMAs per hidden unit
// for one useful column of the forming GEMMs and one useful row of the contraction.  For such tiles the two halves separate:
//   k_tail_form      one workgroup per tail tile: the forming GEMMs with the HIDDEN UNIT in the column index — column (d, kk) of a step
//                    = destination d of the tile, hidden unit k0 + kk; 32 / ndp hidden units per step (ndp = destinations rounded up to a
//                    power of two) — against the same resident x^T window: ceil(65 / KB) steps instead of 65.  All 23 formed tiles of a
//                    column (scalar channels, dot, x1, cross — and x0 (x) v_m in place of the T pre-pass) are scaled, split and PARKED in
//                    global memory in the A-fragment layout of the contraction (the accumulator layout of the forming MFMA).
//   k_tail_contract  tail destinations of MANY molecules gathered 32 to a tile: the contraction alone, A fragments from the parked buffer,
//                    weights from the same stream as k_conv_mf (scalar outputs) / a small stream of their own (vector outputs: x1, cross
//                    and the x0 (x) v weights under one column scale); one output tile per wave, the hidden units cut into R runs = R
//                    partial slabs for the node update.
// Cost per tail destination: ~3 forming steps + 1/32 of a contraction tile, instead of 65 full steps: 33-atom molecules 2 tiles -> 1.03.
template <int SPD>
__global__ __launch_bounds__(MF_THREADS) void k_tail_form(TailArgs a) {
  extern __shared__ float4 lds4[];
  char* __restrict__ lds = reinterpret_cast<char*>(lds4);
  const int L0 = lds_addr(lds);
  int* __restrict__ deg_lds = reinterpret_cast<int*>(lds + MF_MISC);
  unsigned* __restrict__ xmax_lds = reinterpret_cast<unsigned*>(lds + MF_MISC + 128);
  const int tid = threadIdx.x, lane = tid & 63, wave = RFL(tid >> 6);
  const int4 tt = a.tail_tiles[blockIdx.x];  // {first tail-destination index, first atom, atoms | source rows << 8, first source row}
  const int td0 = RFL(tt.x), n0 = RFL(tt.y), n_dst = RFL(tt.z & 255);
  const int s_lo = RFL(tt.w), rows = RFL(tt.z >> 8);
  const int s_base = s_lo & ~1, off = s_lo - s_base;
  const int ndp = n_dst <= 1 ? 1 : (n_dst <= 2 ? 2 : (n_dst <= 4 ? 4 : 8));
  const int KB = 32 / ndp, kb_sh = ndp == 1 ? 5 : (ndp == 2 ? 4 : (ndp == 4 ? 3 : 2));  // hidden units per step = 1 << kb_sh
  const int n_k = a.n_k, nsteps = (n_k + KB - 1) / KB;

  // ---- prologue: zero the coefficient tiles, stage the span's rows transposed, scaled (channel factors, then the span's power of two)
  // and split — as k_conv_mf, without its role split (this kernel's time does not matter)
  for (int idx = tid; idx < 2 * MF_CB / 16; idx += MF_THREADS) reinterpret_cast<float4*>(lds + MF_C)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < 144) {
    const int pl = tid / 72, q = tid - pl * 72;
    *reinterpret_cast<float4*>(lds + (pl ? MF_X0L : MF_X0H) + 120 * MF_ROWB + 16 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (tid == 0) *xmax_lds = 0u;
  if (tid < 32) deg_lds[tid] = tid < n_dst ? a.deg[n0 + tid] : 0;
  float4 va[4], vb[4];
  float mx = 0.f;
  {
    // (every request unconditional, at a clamped address, the channel factors zeroed where the row or the channel group does not exist:
    // as loads under `if (row in the span)` each of the eight was a branch, a request and an `s_waitcnt vmcnt(0)` — eight serialised round
    // trips in front of a kernel of three steps, round 6)
    float4 ra[4], rb[4], ga[4], gb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int item = tid + MF_THREADS * q, jp = item & 31, c4 = item >> 5, c4c = min(c4, 53);
      const int j0 = 2 * jp - off, j1 = j0 + 1;
      const float4 g4 = reinterpret_cast<const float4*>(a.gx)[c4c];
      const float k0 = (c4 < 54 && j0 >= 0 && j0 < rows) ? 1.f : 0.f, k1 = (c4 < 54 && j1 >= 0 && j1 < rows) ? 1.f : 0.f;
      ga[q] = make_float4(g4.x * k0, g4.y * k0, g4.z * k0, g4.w * k0);
      gb[q] = make_float4(g4.x * k1, g4.y * k1, g4.z * k1, g4.w * k1);
      ra[q] = reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + max(0, min(j0, rows - 1))) * a.XS)[c4c];
      rb[q] = reinterpret_cast<const float4*>(a.x + (size_t)(s_lo + max(0, min(j1, rows - 1))) * a.XS)[c4c];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      va[q] = make_float4(ra[q].x * ga[q].x, ra[q].y * ga[q].y, ra[q].z * ga[q].z, ra[q].w * ga[q].w);
      vb[q] = make_float4(rb[q].x * gb[q].x, rb[q].y * gb[q].y, rb[q].z * gb[q].z, rb[q].w * gb[q].w);
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(va[q].x), fabsf(va[q].y)), fmaxf(fabsf(va[q].z), fabsf(va[q].w))));
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(vb[q].x), fabsf(vb[q].y)), fmaxf(fabsf(vb[q].z), fabsf(vb[q].w))));
    }
  }
  LDS_BARRIER();
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
      const int item = tid + MF_THREADS * q, jp = item & 31, c4 = item >> 5;
      if (c4 < 54) {
        const float ea[4] = {va[q].x, va[q].y, va[q].z, va[q].w}, eb[4] = {vb[q].x, vb[q].y, vb[q].z, vb[q].w};
        const bool sc4 = c4 < 30;  // scalar channels 4 c4 .. (120 = 30 x 4), else vector elements 3 u + m -> row (m, u)
        const int e1 = 4 * c4 - 120, u0 = e1 / 3, m0 = e1 - 3 * u0;
        const int lo_off = sc4 ? (MF_X0L - MF_X0H) : (MF_X1L - MF_X1H);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int mm_ = m0 + e, wr = mm_ >= 3 ? 1 : 0;
          const int rowb = sc4 ? MF_X0H + (4 * c4 + e) * MF_ROWB : MF_X1H + ((mm_ - 3 * wr) * 32 + u0 + wr) * MF_ROWB;
          const float a0 = ea[e] * scx, b0 = eb[e] * scx;
          const unsigned ph = cvt_pk_f16(a0, b0), pl = cvt_pk_f16(resid_lo(a0, ph), resid_hi(b0, ph));
          *reinterpret_cast<unsigned*>(lds + rowb + 4 * jp) = ph;
          *reinterpret_cast<unsigned*>(lds + rowb + lo_off + 4 * jp) = pl;
        }
      }
    }
  }
  // ---- builder state (all eight waves): entry (column i, edge slot t); column i = (destination d = i >> kb_sh, hidden unit k0 + (i & (KB - 1)))
  constexpr int BT = MF_THREADS, DPP = BT / SPD, NP = 32 / DPP;
  int ent[NP], slotv[NP];
  float hv[NP], evx[NP], evy[NP], evz[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int g = tid + BT * p, i = g / SPD, t = g % SPD;
    const int d = i >> kb_sh;
    const int dg = d < n_dst ? deg_lds[d] : 0;
    const bool in = d < n_dst && t < dg && t < a.S;
    const int slot = (n0 + (d < n_dst ? d : 0)) * a.S + (t < a.S ? t : 0);
    slotv[p] = slot;
    // (the slot is clamped: unconditional requests, all passes' in flight together, masked after — under `in ? .. : 0` each was a branch, a
    // request and a wait of its own)
    const int sjv = a.esrc[slot], epv = a.epair[slot];
    const float4 gev = a.egeo[slot];
    const int sj = in ? sjv : 0, ep = in ? epv : 0;
    const float kin = in ? 1.f : 0.f;
    evx[p] = gev.x * kin; evy[p] = gev.y * kin; evz[p] = gev.z * kin;
    const int jl = (sj & 0x7fffffff) - s_base;  // (bit 31: bonded)
    const bool valid = in && jl >= 0 && jl < 64;
    // (edges of one ordered pair share one entry: matched by k_geom — MfArgs::epair, JAMUN_EP_* — as in k_conv_mf)
    if (ep & JAMUN_EP_OVERFLOW) atomicOr(a.err, 1);
    const int pa = (ep & 127) - 1, pb = ((ep >> 7) & 127) - 1;
    const int d0 = pa >= 0 ? pa - t : 0, d1 = pb >= 0 ? pb - t : 0;  // lane distance to the pair's other edges (same lane group)
    const bool active = valid && !((unsigned)ep & JAMUN_EP_OWNED);
    ent[p] = active ? (i * MF_ROWB + 2 * jl) | (d0 << 13) | (d1 << 19) : 128;
  }
  auto load_k = [&](int st) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int g = tid + BT * p, i = g / SPD;
      const int k = (st << kb_sh) + (i & (KB - 1));
      hv[p] = a.h[(size_t)min(k, n_k - 1) * a.h_kstride + slotv[p]] * ((st < nsteps && k < n_k) ? 1.f : 0.f);  // (unconditional request, clamped)
    }
  };
  const float scC = pow2f(a.sC);
  auto coef = [&](int p) {
    const int d0 = (ent[p] >> 13) & 63, d1 = (ent[p] >> 19) & 63;
    const float t0 = __shfl(hv[p], lane + d0, 64), t1 = __shfl(hv[p], lane + d1, 64);
    return (hv[p] + (d0 ? t0 : 0.f)) + (d1 ? t1 : 0.f);
  };
  auto build = [&](int buf) {
    char* __restrict__ cbuf = lds + MF_C + buf * MF_CB;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float c0 = coef(p) * scC, c1 = c0 * evx[p], c2 = c0 * evy[p], c3 = c0 * evz[p];
      const unsigned h01 = cvt_pk_f16(c0, c1), h23 = cvt_pk_f16(c2, c3);
      const unsigned l01 = cvt_pk_f16(resid_lo(c0, h01), resid_hi(c1, h01)), l23 = cvt_pk_f16(resid_lo(c2, h23), resid_hi(c3, h23));
      char* __restrict__ d = cbuf + (ent[p] & 0x1fff);
      *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h01 & 0xffffu);
      *reinterpret_cast<unsigned short*>(d + MF_PL) = (unsigned short)(l01 & 0xffffu);
      *reinterpret_cast<unsigned short*>(d + 2 * MF_PL) = (unsigned short)(h01 >> 16);
      *reinterpret_cast<unsigned short*>(d + 3 * MF_PL) = (unsigned short)(l01 >> 16);
      *reinterpret_cast<unsigned short*>(d + 4 * MF_PL) = (unsigned short)(h23 & 0xffffu);
      *reinterpret_cast<unsigned short*>(d + 5 * MF_PL) = (unsigned short)(l23 & 0xffffu);
      *reinterpret_cast<unsigned short*>(d + 6 * MF_PL) = (unsigned short)(h23 >> 16);
      *reinterpret_cast<unsigned short*>(d + 7 * MF_PL) = (unsigned short)(l23 >> 16);
    }
  };
  load_k(0);
  build(0);
  load_k(1);
  LDS_BARRIER();  // x rows, C(step 0)

  // ---- forming, one step = 32 columns; this lane: column c = lane & 31 = (destination dc, hidden-unit offset kkc)
  const int r = lane & 31, hh = lane >> 5;
  const int dc = r >> kb_sh, kkc = r & (KB - 1);
  const int edeg_c = (dc < n_dst && deg_lds[dc] > 0) ? exp_above((float)deg_lds[dc]) : 1;
  const float rs = pow2f(-14 - edeg_c);
  const int fo = r * MF_ROWB + 16 * hh + L0;
  if (kkc == 0 && hh == 0 && dc < n_dst) a.tail_scale[td0 + dc] = pow2f(clamp100(-(sX + a.sC)));
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // F (+ optional second tile to subtract) -> scaled, split, parked as formed tile `ft` of (tail destination td0 + dc, hidden unit k)
  auto park = [&](const f32x16& F, int ft, int k) {
    if (dc >= n_dst || k >= n_k) return;
    // parked layout [hidden unit][tile of 32 tail destinations][formed tile][K-step][row][lane half][hi, lo] x 16 B: what a wave of the
    // contraction reads for one (k, formed tile, K-step) is 2 KB contiguous
    const int tdc = td0 + dc, n_ct = (a.n_tail + 31) >> 5;
    float4* __restrict__ dst = a.P + (((((size_t)k * n_ct + (tdc >> 5)) * TAIL_NFT + ft) * 2) * 32 + (tdc & 31)) * 4 + 2 * hh;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      unsigned ph[4], pl[4];
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        const float v0 = F[8 * s2 + 2 * p2] * rs, v1 = F[8 * s2 + 2 * p2 + 1] * rs;
        ph[p2] = cvt_pk_f16(v0, v1);
        pl[p2] = cvt_pk_f16(resid_lo(v0, ph[p2]), resid_hi(v1, ph[p2]));
      }
      dst[128 * s2] = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
      dst[128 * s2 + 1] = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
    }
  };
  auto form = [&](int xa, int xlo, int ca) {  // one 32 x 32 tile: 4 K-steps of 16 source rows
    f32x16 F = zero16;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float4 ah = lds_f4(xa + 32 * s4), al = lds_f4(xa + xlo + 32 * s4), bh = lds_f4(ca + 32 * s4), bl = lds_f4(ca + MF_PL + 32 * s4);
      M3(F, ah, al, bh, bl);
    }
    return F;
  };
  for (int st = 0; st < nsteps; ++st) {
    const int k = (st << kb_sh) + kkc;
    const int cb = MF_C + (st & 1) * MF_CB + fo;
    if (wave < 4) {  // scalar channels 32 w ..: with the plain coefficients (formed tile w) and with each component of v (x0 (x) v_m: 11 + 3 w + m)
      const int xa = MF_X0H + 32 * wave * MF_ROWB + fo;
      park(form(xa, MF_X0L - MF_X0H, cb), wave, k);
#pragma unroll
      for (int m = 0; m < 3; ++m) park(form(xa, MF_X0L - MF_X0H, cb + (1 + m) * 2 * MF_PL), 11 + 3 * wave + m, k);
    } else if (wave == 4) {  // dot(x1, v)
      f32x16 F = zero16;
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int xa = MF_X1H + m * 32 * MF_ROWB + fo + 32 * s4, ca = cb + (1 + m) * 2 * MF_PL + 32 * s4;
          const float4 ah = lds_f4(xa), al = lds_f4(xa + (MF_X1L - MF_X1H)), bh = lds_f4(ca), bl = lds_f4(ca + MF_PL);
          M3(F, ah, al, bh, bl);
        }
      park(F, 4, k);
    } else {  // plane m: x1[m]; (x1 x v)[m] = x1[m+1] v[m+2] - x1[m+2] v[m+1]
      const int m = wave - 5, m1 = (m + 1) % 3, m2 = (m + 2) % 3;
      const int x1a = MF_X1H + fo;
      park(form(x1a + m * 32 * MF_ROWB, MF_X1L - MF_X1H, cb), 5 + 2 * m, k);
      const f32x16 F1 = form(x1a + m1 * 32 * MF_ROWB, MF_X1L - MF_X1H, cb + (1 + m2) * 2 * MF_PL);
      const f32x16 F2 = form(x1a + m2 * 32 * MF_ROWB, MF_X1L - MF_X1H, cb + (1 + m1) * 2 * MF_PL);
      f32x16 Fc;
#pragma unroll
      for (int q = 0; q < 16; ++q) Fc[q] = F1[q] - F2[q];
      park(Fc, 6 + 2 * m, k);
    }
    build((st + 1) & 1);  // (after the last step: the unused buffer, from zeros)
    load_k(st + 2);
    LDS_BARRIER();
  }
}

// The same for the INITIAL projector (the scheme of k_conv_mfx: 64 scalar channels of the noise-scaled embedding, stored split): formed
// tile 2 c + t = coefficient component c (0: scalar outputs, 1 + m: vector plane m) x channel tile t; wave = formed tile.
template <int SPD>
__global__ __launch_bounds__(MF_THREADS) void k_tail_form_init(TailArgs a) {
  extern __shared__ float4 lds4[];
  char* __restrict__ lds = reinterpret_cast<char*>(lds4);
  const int L0 = lds_addr(lds);
  constexpr int X_H = 0, X_L = 64 * MF_ROWB, X_C = 2 * 64 * MF_ROWB, X_MISC = X_C + 2 * MF_CB;
  int* __restrict__ deg_lds = reinterpret_cast<int*>(lds + X_MISC);
  const int tid = threadIdx.x, lane = tid & 63, wave = RFL(tid >> 6);
  const int4 tt = a.tail_tiles[blockIdx.x];  // (the tile's descriptor: see k_tail_form)
  const int td0 = RFL(tt.x), n0 = RFL(tt.y), n_dst = RFL(tt.z & 255);
  const int s_base = RFL(tt.w) & ~1;
  const int ndp = n_dst <= 1 ? 1 : (n_dst <= 2 ? 2 : (n_dst <= 4 ? 4 : 8));
  const int KB = 32 / ndp, kb_sh = ndp == 1 ? 5 : (ndp == 2 ? 4 : (ndp == 4 ? 3 : 2));
  const int n_k = a.n_k, nsteps = (n_k + KB - 1) / KB;
  for (int idx = tid; idx < 2 * MF_CB / 16; idx += MF_THREADS) reinterpret_cast<float4*>(lds + X_C)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const unsigned* __restrict__ ph = a.xph + (size_t)(s_base >> 1) * 64;
    const unsigned* __restrict__ pl = a.xpl + (size_t)(s_base >> 1) * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + MF_THREADS * q, ch = idx & 63, jp = idx >> 6;
      *reinterpret_cast<unsigned*>(lds + X_H + ch * MF_ROWB + 4 * jp) = ph[idx];
      *reinterpret_cast<unsigned*>(lds + X_L + ch * MF_ROWB + 4 * jp) = pl[idx];
    }
  }
  if (tid < 32) deg_lds[tid] = tid < n_dst ? a.deg[n0 + tid] : 0;
  LDS_BARRIER();
  constexpr int BT = MF_THREADS, DPP = BT / SPD, NP = 32 / DPP;
  int ent[NP], slotv[NP];
  float hv[NP], evx[NP], evy[NP], evz[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int g = tid + BT * p, i = g / SPD, t = g % SPD;
    const int d = i >> kb_sh;
    const int dg = d < n_dst ? deg_lds[d] : 0;
    const bool in = d < n_dst && t < dg && t < a.S;
    const int slot = (n0 + (d < n_dst ? d : 0)) * a.S + (t < a.S ? t : 0);
    slotv[p] = slot;
    // (the slot is clamped: unconditional requests, all passes' in flight together, masked after — under `in ? .. : 0` each was a branch, a
    // request and a wait of its own)
    const int sjv = a.esrc[slot], epv = a.epair[slot];
    const float4 gev = a.egeo[slot];
    const int sj = in ? sjv : 0, ep = in ? epv : 0;
    const float kin = in ? 1.f : 0.f;
    evx[p] = gev.x * kin; evy[p] = gev.y * kin; evz[p] = gev.z * kin;
    const int jl = (sj & 0x7fffffff) - s_base;  // (bit 31: bonded)
    const bool valid = in && jl >= 0 && jl < 64;
    // (edges of one ordered pair share one entry: matched by k_geom — MfArgs::epair, JAMUN_EP_* — as in k_conv_mf)
    if (ep & JAMUN_EP_OVERFLOW) atomicOr(a.err, 1);
    const int pa = (ep & 127) - 1, pb = ((ep >> 7) & 127) - 1;
    const int d0 = pa >= 0 ? pa - t : 0, d1 = pb >= 0 ? pb - t : 0;  // lane distance to the pair's other edges (same lane group)
    const bool active = valid && !((unsigned)ep & JAMUN_EP_OWNED);
    ent[p] = active ? (i * MF_ROWB + 2 * jl) | (d0 << 13) | (d1 << 19) : 128;
  }
  auto load_k = [&](int st) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int g = tid + BT * p, i = g / SPD;
      const int k = (st << kb_sh) + (i & (KB - 1));
      hv[p] = a.h[(size_t)min(k, n_k - 1) * a.h_kstride + slotv[p]] * ((st < nsteps && k < n_k) ? 1.f : 0.f);  // (unconditional request, clamped)
    }
  };
  const float scC = pow2f(a.sC);
  auto coef = [&](int p) {
    const int d0 = (ent[p] >> 13) & 63, d1 = (ent[p] >> 19) & 63;
    const float t0 = __shfl(hv[p], lane + d0, 64), t1 = __shfl(hv[p], lane + d1, 64);
    return (hv[p] + (d0 ? t0 : 0.f)) + (d1 ? t1 : 0.f);
  };
  auto build = [&](int buf) {
    char* __restrict__ cbuf = lds + X_C + buf * MF_CB;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const float c0 = coef(p) * scC, c1 = c0 * evx[p], c2 = c0 * evy[p], c3 = c0 * evz[p];
      const unsigned h01 = cvt_pk_f16(c0, c1), h23 = cvt_pk_f16(c2, c3);
      const unsigned l01 = cvt_pk_f16(resid_lo(c0, h01), resid_hi(c1, h01)), l23 = cvt_pk_f16(resid_lo(c2, h23), resid_hi(c3, h23));
      char* __restrict__ d = cbuf + (ent[p] & 0x1fff);
      *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h01 & 0xffffu);
      *reinterpret_cast<unsigned short*>(d + MF_PL) = (unsigned short)(l01 & 0xffffu);
      *reinterpret_cast<unsigned short*>(d + 2 * MF_PL) = (unsigned short)(h01 >> 16);
      *reinterpret_cast<unsigned short*>(d + 3 * MF_PL) = (unsigned short)(l01 >> 16);
      *reinterpret_cast<unsigned short*>(d + 4 * MF_PL) = (unsigned short)(h23 & 0xffffu);
      *reinterpret_cast<unsigned short*>(d + 5 * MF_PL) = (unsigned short)(l23 & 0xffffu);
      *reinterpret_cast<unsigned short*>(d + 6 * MF_PL) = (unsigned short)(h23 >> 16);
      *reinterpret_cast<unsigned short*>(d + 7 * MF_PL) = (unsigned short)(l23 >> 16);
    }
  };
  load_k(0);
  build(0);
  load_k(1);
  LDS_BARRIER();
  const int r = lane & 31, hh = lane >> 5;
  const int dc = r >> kb_sh, kkc = r & (KB - 1);
  const int edeg_c = (dc < n_dst && deg_lds[dc] > 0) ? exp_above((float)deg_lds[dc]) : 1;
  const float rs = pow2f(-14 - edeg_c);
  const int fo = r * MF_ROWB + 16 * hh + L0;
  if (kkc == 0 && hh == 0 && dc < n_dst) a.tail_scale[td0 + dc] = pow2f(clamp100(-(a.sX + a.sC)));
  const int comp = wave >> 1, ct = wave & 1;
  float4 xh[4], xl[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) { xh[s4] = lds_f4(X_H + 32 * ct * MF_ROWB + fo + 32 * s4); xl[s4] = lds_f4(X_L + 32 * ct * MF_ROWB + fo + 32 * s4); }
  const int n_ct = (a.n_tail + 31) >> 5;
  for (int st = 0; st < nsteps; ++st) {
    const int k = (st << kb_sh) + kkc;
    const int cb = X_C + (st & 1) * MF_CB + comp * 2 * MF_PL + fo;
    f32x16 F;
#pragma unroll
    for (int q = 0; q < 16; ++q) F[q] = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const float4 bh = lds_f4(cb + 32 * s4), bl = lds_f4(cb + MF_PL + 32 * s4);
      M3(F, xh[s4], xl[s4], bh, bl);
    }
    if (dc < n_dst && k < n_k) {
      const int tdc = td0 + dc;
      float4* __restrict__ dst = a.P + (((((size_t)k * n_ct + (tdc >> 5)) * TAIL_NFT_INIT + wave) * 2) * 32 + (tdc & 31)) * 4 + 2 * hh;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        unsigned ph[4], pl[4];
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
          const float v0 = F[8 * s2 + 2 * p2] * rs, v1 = F[8 * s2 + 2 * p2 + 1] * rs;
          ph[p2] = cvt_pk_f16(v0, v1);
          pl[p2] = cvt_pk_f16(resid_lo(v0, ph[p2]), resid_hi(v1, ph[p2]));
        }
        dst[128 * s2] = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
        dst[128 * s2 + 1] = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
      }
    }
    build((st + 1) & 1);
    load_k(st + 2);
    LDS_BARRIER();
  }
}

// The contraction of 32 gathered tail destinations: one workgroup per (tile of 32 tail destinations, run of hidden units, output job) —
// job n < 5: scalar-output tile n (K = the five scalar formed tiles), job 5 + m: vector plane m (x1[m], cross[m], x0 (x) v_m: six formed
// tiles).  The run's hidden units are dealt to the EIGHT WAVES (wave w: k_lo + w, + 8, ...), every wave accumulating its own 32 x 32 tile
// with all loads of a hidden unit in flight at once; the eight partial tiles are summed through LDS in wave order: one partial slab per
// run for the node update.
template <bool INIT>  // INIT: the initial projector (formed tiles 2 c + t of k_tail_form_init, weight stream of k_conv_mfx)
#define TC_WAVES 8   // hidden units of a run in flight at once (one per wave; a wave's units run one after the other: 40-48 loads each,
                     // ALL requested before the first MFMA — two waves per SIMD leave the 200 registers that takes.  With 16 waves (128
                     // registers) the compiler kept 60 and issued the loads next to their uses: `3 loads, s_waitcnt vmcnt(0), 1 MFMA, 1 load,
                     // s_waitcnt vmcnt(0), 2 MFMAs, ...` — twenty serialised round trips per hidden unit, 31 us per launch on cfg3)
__global__ __launch_bounds__(64 * TC_WAVES) void k_tail_contract(TailArgs a) {
  __shared__ float red[TC_WAVES][32][33];
  constexpr int NFT = INIT ? TAIL_NFT_INIT : TAIL_NFT;
  const int tid = threadIdx.x, lane = tid & 63, wave = RFL(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int job = blockIdx.x & 7, cr = blockIdx.x >> 3;
  const int ctile = cr / a.n_runs, run = cr - ctile * a.n_runs;
  const int k_lo = (run * a.n_k) / a.n_runs, k_hi = ((run + 1) * a.n_k) / a.n_runs;
  const int n_ct = (a.n_tail + 31) >> 5;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  // A fragments of (hidden unit k, formed tile ft, K-step s2): row = this lane's tail destination
  auto aptr = [&](int k, int ft, int s2) { return a.P + ((((((size_t)k * n_ct + ctile) * NFT + ft) * 2 + s2) * 32 + r) * 4 + 2 * hh); };
  if constexpr (INIT) {
    // scalar outputs: formed tiles t = 0, 1 (component 0), blocks 20 t + 2 (2 n + s2); plane m: formed tiles 2 (1 + m) + t, blocks 40 + 4 t + 2 s2
    for (int k = k_lo + wave; k < k_hi; k += TC_WAVES) {
      const float4* __restrict__ wk = a.wx + (size_t)k * 48 * 64 + lane;
      float4 ah[4], al[4], bh[4], bl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // (t, s2) = (i >> 1, i & 1)
        const int t = i >> 1, s2 = i & 1;
        const float4* __restrict__ p = aptr(k, job < 5 ? t : 2 * (job - 4) + t, s2);
        ah[i] = p[0]; al[i] = p[1];
        const int blk = job < 5 ? 20 * t + 2 * (2 * job + s2) : 40 + 4 * t + 2 * s2;
        bh[i] = wk[blk * 64]; bl[i] = wk[(blk + 1) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);  // (the requests of a unit stay in front of its MFMAs)
#pragma unroll
      for (int i = 0; i < 4; ++i) { M3(acc, ah[i], al[i], bh[i], bl[i]); }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (job < 5) {
    const int n = job;
    for (int k = k_lo + wave; k < k_hi; k += TC_WAVES) {
      const float4* __restrict__ wk = a.wm + (size_t)k * 124 * 64 + lane;
      float4 ah[10], al[10], bh[10], bl[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) {  // (w, s2) = (i >> 1, i & 1)
        const float4* __restrict__ p = aptr(k, i >> 1, i & 1);
        ah[i] = p[0]; al[i] = p[1];
        bh[i] = wk[(20 * (i >> 1) + 2 * (2 * n + (i & 1))) * 64]; bl[i] = wk[(20 * (i >> 1) + 2 * (2 * n + (i & 1)) + 1) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 10; ++i) { M3(acc, ah[i], al[i], bh[i], bl[i]); }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    const int m = job - 5;
    for (int k = k_lo + wave; k < k_hi; k += TC_WAVES) {
      const float4* __restrict__ wk = a.wmt + (size_t)k * 24 * 64 + lane;
      float4 ah[12], al[12], bh[12], bl[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) {  // (g, s2) = (i >> 1, i & 1): x1[m], cross[m], then x0 (x) v_m of the four scalar channel tiles
        const int g = i >> 1, ft = g == 0 ? 5 + 2 * m : (g == 1 ? 6 + 2 * m : 11 + 3 * (g - 2) + m);
        const float4* __restrict__ p = aptr(k, ft, i & 1);
        ah[i] = p[0]; al[i] = p[1];
        bh[i] = wk[(4 * g + 2 * (i & 1)) * 64]; bl[i] = wk[(4 * g + 2 * (i & 1) + 1) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 12; ++i) { M3(acc, ah[i], al[i], bh[i], bl[i]); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) red[wave][(q & 3) + 8 * (q >> 2) + 4 * hh][r] = acc[q];
  __syncthreads();
  // rows carry 2^(sX + sC - 14 - e(in-degree)) (tail_scale holds 2^-(sX + sC)), columns the weights' 2^sB_w
  for (int idx = tid; idx < 32 * 32; idx += 64 * TC_WAVES) {
    const int row = idx >> 5, col = idx & 31, t2 = 32 * ctile + row;
    if (t2 >= a.n_tail) continue;
    float v = red[0][row][col];
#pragma unroll
    for (int w = 1; w < TC_WAVES; ++w) v += red[w][row][col];
    const int atom = a.tail_atom[t2];
    const int dg = a.deg[atom];
    const int edeg = dg > 0 ? exp_above((float)dg) : 1;
    const float cf = job < 5 ? a.cf0[32 * job + col] : a.cf1t[col];  // (INIT: the column factors of k_conv_mfx's weights)
    v = ((v * a.tail_scale[t2]) * pow2f(14 + edeg)) * cf;
    if (job < 5) a.partial0[((size_t)run * a.n_pad + atom) * (size_t)(a.nt0 * 32) + 32 * job + col] = v;
    else a.partial1[((size_t)run * a.n_pad + atom) * 96 + (job - 5) * 32 + col] = v;
  }
}

int launch_conv_tail(const TailArgs& a, hipStream_t st) {
  if (a.XS != 216 || a.nt0 != 5 || a.S > 64 || a.n_tail_tiles < 1 || !a.epair) return -1;
  if (a.S <= 32) hipLaunchKernelGGL((k_tail_form<32>), dim3(a.n_tail_tiles), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((k_tail_form<64>), dim3(a.n_tail_tiles), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
  hipLaunchKernelGGL(k_tail_contract<false>, dim3(((a.n_tail + 31) / 32) * a.n_runs * 8), dim3(64 * TC_WAVES), 0, st, a);
  return 0;
}

int launch_conv_tail_init(const TailArgs& a, hipStream_t st) {
  if (a.nt0 != 5 || a.S > 64 || a.n_tail_tiles < 1 || !a.epair) return -1;
  const size_t smem = 2 * 64 * MF_ROWB + 2 * MF_CB + 144;
  if (a.S <= 32) hipLaunchKernelGGL((k_tail_form_init<32>), dim3(a.n_tail_tiles), dim3(MF_THREADS), smem, st, a);
  else hipLaunchKernelGGL((k_tail_form_init<64>), dim3(a.n_tail_tiles), dim3(MF_THREADS), smem, st, a);
  hipLaunchKernelGGL(k_tail_contract<true>, dim3(((a.n_tail + 31) / 32) * a.n_runs * 8), dim3(64 * TC_WAVES), 0, st, a);
  return 0;
}

void conv_mf_print_stamps() {
#ifdef MF_TRACE
  static unsigned long long tr[8][40][8];
  if (hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_mftrace), sizeof(tr)) != hipSuccess || tr[0][0][0] == 0) return;
  const unsigned long long t0 = tr[0][0][0];
  fprintf(stderr, "mf trace (workgroup 7): per k-step and wave: start / built / formed / split / contracted / released, cycles since first start\n");
  for (int st = 0; st < 40; ++st)
    for (int w = 0; w < 8; ++w) {
      fprintf(stderr, "  step %2d wave %d:", st, w);
      for (int i = 0; i < 6; ++i) fprintf(stderr, " %7lld", (long long)(tr[w][st][i] - t0));
      fprintf(stderr, "\n");
    }
  static unsigned long long sg[8][4][8];
  if (hipMemcpyFromSymbol(sg, HIP_SYMBOL(g_mfseg), sizeof(sg)) != hipSuccess) return;
  fprintf(stderr, "segments: start / loads issued, tiles zeroed (before the first barrier) / rows staged (waves 4..7) / edges examined (waves 0..3) / loop entered / loop left / tiles staged / slab stored\n");
  for (int q = 0; q < 4; ++q)
    for (int w = 0; w < 8; ++w) {
      if (sg[w][q][0] == 0) continue;
      fprintf(stderr, "  seg %d wave %d:", q, w);
      for (int i = 0; i < 8; ++i) fprintf(stderr, " %7lld", (long long)(sg[w][q][i] - t0));
      fprintf(stderr, "\n");
    }
#endif
}

size_t conv_mf_lds_bytes() { return MF_LDS_BYTES; }

namespace {
template <bool X1>
int launch_mf(const MfArgs& a, int grid, hipStream_t st) {
  if (a.S <= 32) {
    if (a.nks == 3) hipLaunchKernelGGL((k_conv_mf<32, 3, X1>), dim3(grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((k_conv_mf<32, 4, X1>), dim3(grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
  } else if (a.S <= 40) {
    if (a.nks == 3) hipLaunchKernelGGL((k_conv_mf<40, 3, X1>), dim3(grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((k_conv_mf<40, 4, X1>), dim3(grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
  } else {
    if (a.nks == 3) hipLaunchKernelGGL((k_conv_mf<64, 3, X1>), dim3(grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
    else hipLaunchKernelGGL((k_conv_mf<64, 4, X1>), dim3(grid), dim3(MF_THREADS), MF_LDS_BYTES, st, a);
  }
  return 0;
}
}  // namespace

int launch_conv_mf(const MfArgs& a, int grid, hipStream_t st) {
  if (a.XS != 216 || a.nt0 != 5 || a.S > 64 || (a.t_stride & 1)) return -1;
  if (a.nks != 3 && a.nks != 4) return -1;
  if (!a.epair) return -1;
  return a.x1 ? launch_mf<true>(a, grid, st) : launch_mf<false>(a, grid, st);
}

int launch_conv_mfi(const MfiArgs& a, int grid, hipStream_t st) {
  if (a.nt0 != 5 || a.S > 64 || (a.ut != 1 && a.ut != 2 && a.ut != 4) || !a.epair) return -1;
  const size_t smem = a.ut * 32 * MF_ROWB + 2 * MF_CB + 144;
  if (a.S <= 32) {
    if (a.ut == 1) hipLaunchKernelGGL((k_conv_mfi<32, 1>), dim3(grid), dim3(MF_THREADS), smem, st, a);
    else if (a.ut == 2) hipLaunchKernelGGL((k_conv_mfi<32, 2>), dim3(grid), dim3(MF_THREADS), smem, st, a);
    else hipLaunchKernelGGL((k_conv_mfi<32, 4>), dim3(grid), dim3(MF_THREADS), smem, st, a);
  } else {
    if (a.ut == 1) hipLaunchKernelGGL((k_conv_mfi<64, 1>), dim3(grid), dim3(MF_THREADS), smem, st, a);
    else if (a.ut == 2) hipLaunchKernelGGL((k_conv_mfi<64, 2>), dim3(grid), dim3(MF_THREADS), smem, st, a);
    else hipLaunchKernelGGL((k_conv_mfi<64, 4>), dim3(grid), dim3(MF_THREADS), smem, st, a);
  }
  return 0;
}

int launch_conv_mfx(const MfxArgs& a, int grid, hipStream_t st) {
  if (a.nt0 != 5 || a.S > 64 || !a.epair) return -1;
  const size_t smem = 2 * 64 * MF_ROWB + 2 * MF_CB + 144;
  if (a.S <= 32) hipLaunchKernelGGL((k_conv_mfx<32>), dim3(grid), dim3(MF_THREADS), smem, st, a);
  else hipLaunchKernelGGL((k_conv_mfx<64>), dim3(grid), dim3(MF_THREADS), smem, st, a);
  return 0;
}

int conv_mf_set_max_lds() {
  const void* ft[4] = {(const void*)k_tail_form<32>, (const void*)k_tail_form<64>, (const void*)k_tail_form_init<32>, (const void*)k_tail_form_init<64>};
  for (const void* f : ft)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  const void* fx[2] = {(const void*)k_conv_mfx<32>, (const void*)k_conv_mfx<64>};
  for (const void* f : fx)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  const void* fi[6] = {(const void*)k_conv_mfi<32, 1>, (const void*)k_conv_mfi<64, 1>, (const void*)k_conv_mfi<32, 2>, (const void*)k_conv_mfi<64, 2>,
                       (const void*)k_conv_mfi<32, 4>, (const void*)k_conv_mfi<64, 4>};
  for (const void* f : fi)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  const void* fns[12] = {(const void*)k_conv_mf<32, 4, false>, (const void*)k_conv_mf<64, 4, false>, (const void*)k_conv_mf<32, 3, false>,
                         (const void*)k_conv_mf<64, 3, false>, (const void*)k_conv_mf<40, 4, false>, (const void*)k_conv_mf<40, 3, false>,
                         (const void*)k_conv_mf<32, 4, true>,  (const void*)k_conv_mf<64, 4, true>,  (const void*)k_conv_mf<32, 3, true>,
                         (const void*)k_conv_mf<64, 3, true>,  (const void*)k_conv_mf<40, 4, true>,  (const void*)k_conv_mf<40, 3, true>};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) != hipSuccess) return -1;
  return 0;
}
