// jamun_conv_fused.hip — fused conv contraction (hidden layers): the A operand is formed ON THE MATRIX CORES and every
// wave OWNS one output tile.
//
// Same mathematics as jamun_conv.hip (destination-grouped association of src/jamun/e3tools/nn/_conv.py:93-119):
//     m[(i,c)][w] = sum_k sum_u ( sum_{e->i} h~_e[k] zeta_e[c][u] ) W~[(k,u)][w]
// but the inner edge sum is written as a dense product over the source atoms j of the tile's span (whole small molecules,
// or one row block of a large molecule — the product is linear in C, so blocks simply add partial slabs; JR <= 64 rows):
//     D'_k[u][i] = sum_j X[j][u] * C_k[j][i]          C_k[j][i] = sum over the (<= 2) edges j->i of h~_e[k] * f_e
// with f = 1 or a component of the edge unit vector.  D' is a 32x32 MFMA accumulator whose COLUMN is the destination
// atom, i.e. exactly the lane layout of the A operand of the main product  out[i][w] += sum_u D'[u][i] W[(k,u)][w]:
// register q of the accumulator is K-step q of the main MFMA (rows u = (q&3) + 8(q>>2) + 4(lane>>5)).  The weights are
// packed in that K order.
//
// One persistent workgroup per CU = 8 waves.  Work items are (tile of 32 destination atoms, hidden unit k); a workgroup
// walks a host-built list of segments (tile, run of k): the (tile, k) list (optionally sliced in k over XCD groups) is cut
// evenly — by item count or by modelled cost — over the workgroups, so there is no tail round.
//
// Every wave OWNS one output tile: wave w < NT0 the scalar-row tile w, waves NT0..NT0+2 the vector planes.  No accumulator
// is shared between waves: no cross-wave reduction, 16 accumulator registers per wave.  The k of a segment are software
// pipelined with ONE barrier per k; in the interval of k
//   * the scalar-row tiles D' of k+1 (4 scalar-input u-tiles, the dot-product u-tile) and the T tiles of k+2
//     (T[j][w] = sum_u x0_j[u] W[(k,u)][w], vector rows from scalar inputs) are formed by statically assigned waves and
//     parked in LDS in register order (conflict-free b128, same lane / same register when read back as an operand),
//   * a scalar-row owner runs the main K-steps of k against the tiles formed one interval earlier,
//   * a plane owner forms its x1 / cross tiles of k+1 in registers, runs their main K-steps at once, and applies
//     out_m[i][w] += sum_j C^{h v_m}[j][i] T[j][w]  with the T tile of k+1 as the B operand,
//   * all threads build the coefficient tiles of k+2 (double buffered; h~ values prefetched one interval ahead).
// The forming units are dealt so that every SIMD (waves w and w+4) carries the same number of MFMAs per interval.
// Feature rows and coefficient tiles are stored TRANSPOSED ([column][source row], row stride JR = 4 * odd) so the four
// K-steps a lane needs are one ds_read_b128.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "jamun_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FW JAMUN_FUSED_WAVES
#define FT (64 * FW)
#define FPAIRS ((64 * 32 + FT - 1) / FT)  // (source row, destination) pairs per thread, JR <= 64
#define MAXB JAMUN_FUSED_MAX_B
#ifndef TR
#define TR 8  // depth of the T-tile weight ring (blocks of 4 K-steps): 32 MFMAs of prefetch distance
#endif

#ifdef JAMUN_STAMP
__device__ unsigned long long g_fstamp[8];
__device__ unsigned long long g_wstamp[32];
#define FSTAMP(t) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); } while (0)
#else
#define FSTAMP(t) do { } while (0)
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, which would
// expose the latency of every global prefetch (h~ of the next k, weights) issued before it.
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define RFL(v) __builtin_amdgcn_readfirstlane(v)

__global__ __launch_bounds__(FT, 1) void k_conv_fused(FusedArgs a) {
  extern __shared__ float4 lds4[];
  float* __restrict__ lds = reinterpret_cast<float*>(lds4);
  const int JR = a.JR;
  const int CT = 32 * JR;                                                   // floats per coefficient tile
  float* __restrict__ xT = lds;                                             // [XS][JR]
  float* __restrict__ cT = xT + a.XS * JR;                                  // [2][4][32][JR]
  float4* __restrict__ dP = reinterpret_cast<float4*>(cT + 2 * 4 * CT);     // [2][n_p][4][64] formed scalar-row tiles
  float4* __restrict__ dT = dP + 2 * a.n_p * 256;                           // [2][n_t][4][64] T tiles
  int* __restrict__ pm = reinterpret_cast<int*>(dP);                        // [32][JR] slot map (prologue only; aliases dP)
  int* __restrict__ ua = reinterpret_cast<int*>(dT + 2 * a.n_t * 256);      // [FW][max_a][8] forming units
  int* __restrict__ ub = ua + FW * a.max_a * 8;                             // [FW][MAXB][8] owner entries
  int* __restrict__ deg_lds = ub + FW * MAXB * 8;                           // [32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = RFL(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;

  // ---- once per workgroup: the unit tables
  for (int idx = tid; idx < FW * a.max_a * 8; idx += FT) ua[idx] = reinterpret_cast<const int*>(a.a_units)[idx];
  for (int idx = tid; idx < FW * MAXB * 8; idx += FT) ub[idx] = reinterpret_cast<const int*>(a.b_units)[idx];
  // tiles start finite: stand-in reads (padding K-groups, ramp intervals) must never see NaN bit patterns
  for (int idx = tid; idx < 2 * 4 * CT + 2 * (a.n_p + a.n_t) * 1024; idx += FT) cT[idx] = 0.f;
  const int4* __restrict__ my_a = reinterpret_cast<const int4*>(ua + wave * a.max_a * 8);
  const int4* __restrict__ my_b = reinterpret_cast<const int4*>(ub + wave * MAXB * 8);
  const int4 own = a.owner[wave];  // {kind: -1 none / 0 scalar-row tile / 1 vector plane, index}
  const int own_kind = RFL(own.x), own_idx = RFL(own.y), own_t = RFL(own.z);  // own_t: index of this wave's first T unit, -1 if none
  const int n_ent = RFL(own.w) & 0xff, n_self = RFL(own.w) >> 8;  // owner entries; of which formed in registers (plane owners, <= 2)

  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, s_pro = 0, s_wait = 0, s_a = 0, s_b = 0, s_build = 0, s_epi = 0;
  (void)t0; (void)t1; (void)t2; (void)t3; (void)t4; (void)s_pro; (void)s_wait; (void)s_a; (void)s_b; (void)s_build; (void)s_epi;

  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int4 sg0 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2];
    const int4 sg1 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2 + 1];
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) { return kk < k_run ? k_begin + kk : k_extra; };
    FSTAMP(t0);
    const int2 t_at = a.tile_atoms[tile];
    const int n0 = t_at.x, n_dst = t_at.y;  // first destination atom and number of destination atoms (<= 32) of the tile
    const int2 span = a.tile_span[tile];
    const int rows = span.y - span.x;
    const int Jt4 = (rows + 3) & ~3;  // K extent of the forming products for this tile
    const int n8 = Jt4 >> 3, tail4 = Jt4 & 4;

    // D'[u][i] += sum_j X[j][u] C[j][i] over the terms of a forming unit (transposed tiles: four K-steps per b128 read)
    // (No sign handling here: the K-steps feed the operands straight from LDS.  A unit's negative terms — the second
    // term of a cross product — are formed into a second accumulator by a second call and subtracted by the caller; one
    // v_mul per MFMA in this loop costs ~10 % of the forming rate, profiles/r1e micro-benchmark.)
    auto form = [&](f32x16& af, int n_terms, const int4& ut, const float* __restrict__ cb) {
      // operand pointers of all terms first; the K-groups of all terms then run as ONE flat loop whose body always
      // fetches the next group (next term's first group at a term boundary, a harmless re-read at the very end): no
      // branches around the LDS reads, so the waits can be counted exactly
      const float* xq[3];
      const float* cq[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int td = RFL(t == 0 ? ut.x : (t == 1 ? ut.y : ut.z));
        const int xcol0 = td & 0xfff, stride = (td >> 12) & 0xf, ctype = (td >> 16) & 0xf;
        xq[t] = xT + (xcol0 + r * stride) * JR;
        cq[t] = cb + ctype * CT + r * JR;
      }
      if (tail4) {  // four more source rows per term: two K-steps, rows 8 n8 + 2s + hh (all fetched before the first MFMA)
        float ta[3][2], tb[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int tt = t < n_terms ? t : 0;
          const float* __restrict__ xs = tt == 0 ? xq[0] : (tt == 1 ? xq[1] : xq[2]);
          const float* __restrict__ cs = tt == 0 ? cq[0] : (tt == 1 ? cq[1] : cq[2]);
          ta[t][0] = xs[8 * n8 + hh]; tb[t][0] = cs[8 * n8 + hh];
          ta[t][1] = xs[8 * n8 + 2 + hh]; tb[t][1] = cs[8 * n8 + 2 + hh];
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          if (t >= n_terms) break;  // wave-uniform
          af = MFMA(ta[t][0], tb[t][0], af);
          af = MFMA(ta[t][1], tb[t][1], af);
        }
      }
      if (n8 > 0) {
        const float4* __restrict__ xp = reinterpret_cast<const float4*>(xq[0] + 4 * hh);  // rows 8g + 4hh + (0..3)
        const float4* __restrict__ cp = reinterpret_cast<const float4*>(cq[0] + 4 * hh);
        float4 av = xp[0], bv = cp[0];
        int t = 0, g = 0;
        const int total = n_terms * n8;
        for (int i = 0; i < total; ++i) {
          int gn = g + 1, tn = t;
          if (gn == n8) { gn = 0; tn = t + 1; }
          const int tl = tn < n_terms ? tn : t;  // past the end: re-read the last group's term
          const float* __restrict__ xs = tl == 0 ? xq[0] : (tl == 1 ? xq[1] : xq[2]);
          const float* __restrict__ cs = tl == 0 ? cq[0] : (tl == 1 ? cq[1] : cq[2]);
          const float4 an = reinterpret_cast<const float4*>(xs + 4 * hh)[2 * gn];
          const float4 bn = reinterpret_cast<const float4*>(cs + 4 * hh)[2 * gn];
          af = MFMA(av.x, bv.x, af);
          af = MFMA(av.y, bv.y, af);
          af = MFMA(av.z, bv.z, af);
          af = MFMA(av.w, bv.w, af);
          av = an; bv = bn;
          t = tn; g = gn;
        }
      }
    };
    // a unit = positive terms, then negative terms (n_terms = positive | negative << 4)
    auto form_unit = [&](f32x16& af, int n_terms, const int4& ut, const float* __restrict__ cb) {
      const int n_pos = n_terms & 0xf, n_neg = n_terms >> 4;
      form(af, n_pos, ut, cb);
      if (n_neg > 0) {  // wave-uniform
        f32x16 afn;
#pragma unroll
        for (int q = 0; q < 16; ++q) afn[q] = 0.f;
        const int4 un = n_pos == 1 ? make_int4(ut.y, ut.z, 0, 0) : (n_pos == 2 ? make_int4(ut.z, 0, 0, 0) : make_int4(ut.x, ut.y, ut.z, 0));
        form(afn, n_neg, un, cb);
#pragma unroll
        for (int q = 0; q < 16; ++q) af[q] -= afn[q];
      }
    };

    // The weight-group count is padded to a multiple of TR on the host (zero weights; the feature columns read for the
    // padding are finite stand-ins), so the loop body is straight-line: unconditional ring loads let the compiler count
    // vmcnt exactly instead of draining the whole queue.
    auto t_form = [&](const int4& ud, const int4& ut, const float4* __restrict__ wk, float4 (&ring)[TR], int dbuf) {
      const int tix = RFL(ud.y), wofs = RFL(ud.z), jt = RFL(ut.x), nsg = RFL(ut.y), xcol0 = RFL(ut.z);
      if (32 * jt >= Jt4) return;  // wave-uniform: the tile sees no source atom in this row tile
      int row = 32 * jt + r;
      row = row < JR ? row : JR - 1;  // rows past the tile hold finite stand-ins; their coefficients are zero
      const float* __restrict__ xr = xT + (xcol0 + hh) * JR + row;
      f32x16 af;
#pragma unroll
      for (int q = 0; q < 16; ++q) af[q] = 0.f;
      float xa[4], xn[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) xa[st] = xr[2 * st * JR];
      for (int sg = 0; sg < nsg; sg += TR) {
#pragma unroll
        for (int i = 0; i < TR; ++i) {
          // block sg+i reads ring[i] IN PLACE; the slot the previous block consumed is refilled with the block TR-1 ahead
          // (a copy of ring[i] followed by a refill of the same slot makes the compiler wait for the load it just issued)
          const float4 (&w) = ring[i];
          int nxt = sg + i == 0 ? TR - 1 : sg + i - 1 + TR;
          nxt = nxt < nsg ? nxt : nsg - 1;
          ring[(i + TR - 1) % TR] = wk[(wofs + nxt) * 64];
          const int nb = sg + i + 1 < nsg ? sg + i + 1 : nsg - 1;
          const float* __restrict__ xs = xr + 8 * nb * JR;
#pragma unroll
          for (int st = 0; st < 4; ++st) xn[st] = xs[2 * st * JR];
          af = MFMA(xa[0], w.x, af);
          af = MFMA(xa[1], w.y, af);
          af = MFMA(xa[2], w.z, af);
          af = MFMA(xa[3], w.w, af);
#pragma unroll
          for (int st = 0; st < 4; ++st) xa[st] = xn[st];
        }
      }
#pragma unroll
      for (int qg = 0; qg < 4; ++qg)
        dT[((dbuf * a.n_t + tix) * 4 + qg) * 64 + lane] = make_float4(af[4 * qg], af[4 * qg + 1], af[4 * qg + 2], af[4 * qg + 3]);
    };
    auto t_ring = [&](const int4& ud, const float4* __restrict__ wk, float4 (&ring)[TR]) {
      const int wofs = RFL(ud.z);
#pragma unroll
      for (int i = 0; i < TR; ++i) ring[i] = wk[(wofs + i) * 64];
    };

    // ---- per segment: transposed feature rows, (source, destination) -> edge-slot map, per-pair edge geometry.
    // (The previous segment ended with a barrier: nobody reads the tiles any more.)
    {
      const int ncg = (a.XS + 3) >> 2;  // 16-byte column groups; lane = source row
      const float* __restrict__ xrow = a.x + (size_t)(span.x + (lane < rows ? lane : 0)) * a.XS;
      const bool vec_ok = (a.XS & 3) == 0;
      for (int cg0 = wave; cg0 < ncg; cg0 += 4 * FW) {
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cg = cg0 + i * FW;
          v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (cg < ncg && lane < rows) {
            if (vec_ok) v[i] = reinterpret_cast<const float4*>(xrow)[cg];
            else {
              const int c = 4 * cg;
              v[i].x = xrow[c];
              if (c + 1 < a.XS) v[i].y = xrow[c + 1];
              if (c + 2 < a.XS) v[i].z = xrow[c + 2];
              if (c + 3 < a.XS) v[i].w = xrow[c + 3];
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cg = cg0 + i * FW;
          if (cg < ncg && lane < JR) {
            const int c = 4 * cg;
            xT[c * JR + lane] = v[i].x;
            if (c + 1 < a.XS) xT[(c + 1) * JR + lane] = v[i].y;
            if (c + 2 < a.XS) xT[(c + 2) * JR + lane] = v[i].z;
            if (c + 3 < a.XS) xT[(c + 3) * JR + lane] = v[i].w;
          }
        }
      }
    }
    for (int idx = tid; idx < CT; idx += FT) pm[idx] = 0;
    if (tid < 32) deg_lds[tid] = (tid < n_dst) ? a.deg[n0 + tid] : 0;
    __syncthreads();
    for (int idx = tid; idx < 32 * a.S; idx += FT) {
      const int il = idx / a.S, t = idx - il * a.S;
      if (t < deg_lds[il]) {
        const int sj = a.esrc[(size_t)(n0 + il) * a.S + t];
        const int jl = (sj & 0x7fffffff) - span.x;
        // radial and bonded edges of one (source, destination) pair go to different byte fields: no write conflict
        // within a field because a destination's radial neighbours (and its bonded sources) are distinct atoms.
        // (Sources outside the span belong to another tile of the same destinations: large molecules are cut into row blocks.)
        if (jl >= 0 && jl < rows) atomicOr(&pm[il * JR + jl], (sj < 0) ? ((t + 1) << 8) : (t + 1));
      }
    }
    // the T tiles of the first k need only the feature rows: formed here, behind the latency of the map build
    for (int ui = 0; ui < a.max_a; ++ui) {
      const int4 ud = my_a[2 * ui], ut = my_a[2 * ui + 1];
      const int kind = RFL(ud.x);
      if (kind < 0) break;
      if (kind != 1) continue;
      const float4* __restrict__ wk = a.wpack + (size_t)k_of(0) * a.k_stride + lane;
      float4 ring[TR];
      t_ring(ud, wk, ring);
      t_form(ud, ut, wk, ring, 0);
    }
    __syncthreads();
    // Each thread owns up to FPAIRS (destination il, source row j) pairs, p = il * JR + j: slots, unit vectors and h~
    // addresses live in registers for the whole segment.
    int p_oa[FPAIRS], p_ob[FPAIRS];  // edge slot of the radial / bonded edge of the pair (a stand-in if none)
    int p_has = 0;                   // bit 2i: pair i has a radial edge, bit 2i+1: a bonded edge
    float p_g[FPAIRS][3];            // unit vector of the pair (its radial and bonded edge point the same way)
    float p_ha[FPAIRS], p_hb[FPAIRS];
#pragma unroll
    for (int i = 0; i < FPAIRS; ++i) {
      const int p = tid + FT * i;
      const int pt = (p < CT) ? pm[p] : 0;
      const int il = p / JR;
      const int ta = pt & 0xff, tb = (pt >> 8) & 0xff;
      // loads are unconditional (slot 0 of atom n0 stands in for absent edges): no branches around them
      const int ea = ta ? (n0 + il) * a.S + ta - 1 : n0 * a.S, eb = tb ? (n0 + il) * a.S + tb - 1 : n0 * a.S;
      p_oa[i] = ea;
      p_ob[i] = eb;
      p_has |= (ta ? 1 : 0) << (2 * i) | (tb ? 1 : 0) << (2 * i + 1);
      const float4 ge = a.egeo[ta ? ea : eb];
      const bool any = ta || tb;
      p_g[i][0] = any ? ge.x : 0.f; p_g[i][1] = any ? ge.y : 0.f; p_g[i][2] = any ? ge.z : 0.f;
    }
    // A tile of a large molecule (destination chunk x source row block) may have no edge at all: its slab is zero
    if (a.row_blocks && !__syncthreads_or(p_has)) {  // (only batches with row-block tiles pay for the vote)
      if (own_kind == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
          if (row < n_dst) a.partial0[((size_t)slab * a.n_pad + n0 + row) * (a.nt0 * 32) + own_idx * 32 + r] = 0.f;
        }
      } else if (own_kind == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
          if (row < n_dst) a.partial1[(((size_t)slab * a.n_pad + n0 + row) * 3 + own_idx) * 32 + r] = 0.f;
        }
      }
      continue;
    }
    auto load_h = [&](int k) {  // raw loads (row k of the [k][slot] table: neighbouring slots share cache lines); write_c masks out the stand-ins of absent edges
      const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
      for (int i = 0; i < FPAIRS; ++i) {
        p_ha[i] = hk[p_oa[i]];
        p_hb[i] = hk[p_ob[i]];
      }
    };
    auto write_c = [&](float* __restrict__ cbuf) {
#pragma unroll
      for (int i = 0; i < FPAIRS; ++i) {
        const int p = tid + FT * i;
        if (p < CT) {  // every (destination, row < JR) entry is written: absent pairs and rows >= the span get zeros
          const float ha = (p_has >> (2 * i)) & 1 ? p_ha[i] : 0.f, hb = (p_has >> (2 * i + 1)) & 1 ? p_hb[i] : 0.f;
          cbuf[p] = ha + hb;
          cbuf[CT + p] = fmaf(hb, p_g[i][0], ha * p_g[i][0]);
          cbuf[2 * CT + p] = fmaf(hb, p_g[i][1], ha * p_g[i][1]);
          cbuf[3 * CT + p] = fmaf(hb, p_g[i][2], ha * p_g[i][2]);
        }
      }
    };

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    load_h(k_of(0));
    write_c(cT);  // (the slot map, aliased on dP, is dead after the barrier below)
    FSTAMP(t1);
#ifdef JAMUN_STAMP
    s_pro += t1 - t0;
#endif
    LDS_BARRIER();

    // ---- software pipeline over the k of the segment, ONE barrier per interval.  Interval `it` runs
    //   T tiles of k(it+2)                     (feature rows x weights; no coefficient tile)        -> dT[(it+2)&1]
    //   forming of k(it+1) from C(k(it+1))     (scalar-row tiles -> dP[(it+1)&1]; plane owners form their x1 / cross
    //                                           tiles in registers and run the main K-steps at once; apply T(k(it+1)))
    //   main K-steps of k(it) from dP[it&1]    (scalar-row owners)
    //   build of C(k(it+2)) into the buffer C(k(it)) occupied (last read in interval it-1)
    for (int it = -1; it < nk; ++it) {
      const bool v_main = it >= 0, v_next = it + 1 < nk, v_t = it + 2 < nk;
      // k of the three pipeline stages, clamped into the segment: loads are issued unconditionally (ramp intervals fetch a
      // valid stand-in they never use), only the arithmetic is skipped
      const int k0 = k_of(it < 0 ? 0 : it), k1 = k_of(it + 1 < nk ? it + 1 : nk - 1), k2 = k_of(it + 2 < nk ? it + 2 : nk - 1);
      const float* __restrict__ cb = cT + ((it + 1) & 1) * 4 * CT;  // C(k(it+1))
      FSTAMP(t0);
      const float4* __restrict__ wkb = a.wpack + (size_t)(own_kind == 0 ? k0 : k1) * a.k_stride + lane;
      const float4* __restrict__ wkt = a.wpack + (size_t)k2 * a.k_stride + lane;
      // weights of the first two owner entries: in flight behind the forming units.  (Entry fields z of non-weight and
      // terminator entries are small valid block offsets, so every fetch is in range.)  Three buffers rotate: entry e reads
      // B[e % 3] in place while B[(e + 2) % 3] — consumed by entry e-1 — is refilled with the weights of entry e+2: no
      // register copies, so the compiler never has to wait for a load it has just issued.
      float4 B0[4], B1[4], B2[4];
      auto wload = [&](float4 (&B)[4], int e) {
        const int wofs = RFL(my_b[2 * (e < MAXB ? e : MAXB - 1)].z);
#pragma unroll
        for (int q = 0; q < 4; ++q) B[q] = wkb[(wofs + q) * 64];
      };
      wload(B0, 0);
      wload(B1, 1);
      // weight ring of this wave's T tile (if it has one)
      float4 tring[TR];
      if (own_t >= 0) t_ring(my_a[2 * own_t], wkt, tring);

      // ================= forming units of this wave =================
      for (int ui = 0; ui < a.max_a; ++ui) {
        const int4 ud = my_a[2 * ui];
        const int4 ut = my_a[2 * ui + 1];
        const int kind = RFL(ud.x);
        if (kind < 0) break;
        if (kind == 1) {
          if (!v_t) continue;
          if (ui != own_t) t_ring(ud, wkt, tring);  // (a second T tile on the same wave: not prefetched)
          t_form(ud, ut, wkt, tring, (it + 2) & 1);
        } else {
          if (!v_next) continue;
          const int tix = RFL(ud.y);
          f32x16 af;
#pragma unroll
          for (int q = 0; q < 16; ++q) af[q] = 0.f;
          form_unit(af, RFL(ud.z), ut, cb);
#pragma unroll
          for (int qg = 0; qg < 4; ++qg)
            dP[((((it + 1) & 1) * a.n_p + tix) * 4 + qg) * 64 + lane] =
                make_float4(af[4 * qg], af[4 * qg + 1], af[4 * qg + 2], af[4 * qg + 3]);
        }
      }
      FSTAMP(t1);
      // h~ of k(it+2): in flight behind the owner entries, written to the free buffer at the end of the interval.  (Issued
      // here rather than at the top so the waves do not all hit the texture addresser right after the barrier.)
      load_h(k2);

      // ================= owner entries: K-steps into the tile this wave owns =================
      if (own_kind == 0) {
        // ---- scalar-row owner: main K-steps of k(it) against the tiles parked in the previous interval
        auto main_entry = [&](const float4 (&w)[4], int e) {
          const int tix = RFL(my_b[2 * e].y);
          float4 av[4];
#pragma unroll
          for (int qg = 0; qg < 4; ++qg) av[qg] = dP[(((it & 1) * a.n_p + tix) * 4 + qg) * 64 + lane];
#pragma unroll
          for (int qg = 0; qg < 4; ++qg) {
            acc = MFMA(av[qg].x, w[qg].x, acc);
            acc = MFMA(av[qg].y, w[qg].y, acc);
            acc = MFMA(av[qg].z, w[qg].z, acc);
            acc = MFMA(av[qg].w, w[qg].w, acc);
          }
        };
        if (v_main) {
          wload(B2, 2); if (0 < n_ent) main_entry(B0, 0);
          wload(B0, 3); if (1 < n_ent) main_entry(B1, 1);
          wload(B1, 4); if (2 < n_ent) main_entry(B2, 2);
          wload(B2, 5); if (3 < n_ent) main_entry(B0, 3);
          if (4 < n_ent) main_entry(B1, 4);
          if (5 < n_ent) main_entry(B2, 5);
        }
      } else if (own_kind == 1 && v_next) {
        // ---- plane owner: form its x1 / cross tiles of k(it+1) in registers, feed the accumulator registers straight into
        // the main MFMAs, then apply the T tiles of k(it+1)
        auto self_entry = [&](const float4 (&w)[4], int e) {
          f32x16 af;
#pragma unroll
          for (int q = 0; q < 16; ++q) af[q] = 0.f;
          form_unit(af, RFL(my_b[2 * e].y), my_b[2 * e + 1], cb);
#pragma unroll
          for (int qg = 0; qg < 4; ++qg) {
            acc = MFMA(af[4 * qg + 0], w[qg].x, acc);
            acc = MFMA(af[4 * qg + 1], w[qg].y, acc);
            acc = MFMA(af[4 * qg + 2], w[qg].z, acc);
            acc = MFMA(af[4 * qg + 3], w[qg].w, acc);
          }
        };
        if (0 < n_self) self_entry(B0, 0);
        if (1 < n_self) self_entry(B1, 1);
        for (int e = n_self; e < n_ent; ++e) {
          // out_m[i][w] += sum_j C^{h v_m}[j][i] T[j][w]: A operand = coefficient tile rows in T's register order
          const int4 bd = my_b[2 * e];
          const int tix = RFL(bd.y), m = RFL(bd.z), jt = RFL(bd.w);
          if (32 * jt >= Jt4) continue;
          const float* __restrict__ cm = cb + (1 + m) * CT + r * JR + 32 * jt + 4 * hh;
          float4 tv[4], cv[4];  // all eight operand fragments are requested before the first MFMA
#pragma unroll
          for (int qg = 0; qg < 4; ++qg) {
            tv[qg] = dT[((((it + 1) & 1) * a.n_t + tix) * 4 + qg) * 64 + lane];
            cv[qg] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (32 * jt + 4 * hh + 8 * qg < JR) cv[qg] = *reinterpret_cast<const float4*>(cm + 8 * qg);
          }
#pragma unroll
          for (int qg = 0; qg < 4; ++qg) {
            acc = MFMA(cv[qg].x, tv[qg].x, acc);
            acc = MFMA(cv[qg].y, tv[qg].y, acc);
            acc = MFMA(cv[qg].z, tv[qg].z, acc);
            acc = MFMA(cv[qg].w, tv[qg].w, acc);
          }
        }
      }
      FSTAMP(t2);
      if (v_t) write_c(cT + (it & 1) * 4 * CT);
      FSTAMP(t3);
      LDS_BARRIER();
#ifdef JAMUN_STAMP
      FSTAMP(t4);
      s_a += t1 - t0; s_b += t2 - t1; s_build += t3 - t2; s_wait += t4 - t3;
#endif
    }

    // ---- the owner stores its tile of the segment's partial slab
    FSTAMP(t0);
    if (own_kind == 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        if (row < n_dst) a.partial0[((size_t)slab * a.n_pad + n0 + row) * (a.nt0 * 32) + own_idx * 32 + r] = acc[q];
      }
    } else if (own_kind == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        if (row < n_dst) a.partial1[(((size_t)slab * a.n_pad + n0 + row) * 3 + own_idx) * 32 + r] = acc[q];
      }
    }
#ifdef JAMUN_STAMP
    FSTAMP(t1);
    s_epi += t1 - t0;
#endif
  }
#ifdef JAMUN_STAMP
  if (lane == 0) {
    atomicAdd(&g_fstamp[0], s_build); atomicAdd(&g_fstamp[1], s_wait); atomicAdd(&g_fstamp[2], s_a); atomicAdd(&g_fstamp[3], s_b);
    atomicAdd(&g_fstamp[4], s_pro); atomicAdd(&g_fstamp[6], s_epi);
    atomicAdd(&g_wstamp[wave], s_a); atomicAdd(&g_wstamp[8 + wave], s_b); atomicAdd(&g_wstamp[16 + wave], s_wait); atomicAdd(&g_wstamp[24 + wave], s_build);
  }
#endif
}

size_t fused_lds_bytes(int XS, int JR, int n_p, int n_t, int max_a) {
  const size_t fl = (size_t)XS * JR + 2 * 4 * 32 * (size_t)JR + 2 * (size_t)(n_p + n_t) * 1024 + (size_t)FW * max_a * 8 +
                    (size_t)FW * MAXB * 8 + 32;
  return sizeof(float) * ((fl + 3) & ~(size_t)3);
}

int launch_conv_fused(const FusedArgs& a, int grid, hipStream_t st) {
  const size_t smem = fused_lds_bytes(a.XS, a.JR, a.n_p, a.n_t, a.max_a);
  if (smem > JAMUN_MAX_DYN_LDS) return -2;
  if (32 * a.JR > 2 * a.n_p * 1024) return -1;  // the slot map aliases the formed scalar-row tiles
  hipLaunchKernelGGL(k_conv_fused, dim3(grid), dim3(FT), smem, st, a);
  return 0;
}

int conv_fused_read_stamps(unsigned long long* out8) {
#ifdef JAMUN_STAMP
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_fstamp), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
  {
    unsigned long long w[32], z32[32] = {0};
    if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_wstamp), sizeof(w)) == hipSuccess) {
      for (int i = 0; i < 32; ++i) fprintf(stderr, "%s%llu%s", i % 8 == 0 ? (i == 0 ? "wstamp A: " : i == 8 ? "wstamp B: " : i == 16 ? "wstamp wait: " : "wstamp build: ") : "", w[i] / 1000, i % 8 == 7 ? "\n" : " ");
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wstamp), z32, sizeof(z32));
    }
  }
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_fstamp), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
#else
  (void)out8;
  return -2;
#endif
}

int conv_fused_set_max_lds() {
  return hipFuncSetAttribute((const void*)k_conv_fused, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) ==
                 hipSuccess
             ? 0
             : -1;
}
