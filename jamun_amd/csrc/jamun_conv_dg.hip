// jamun_conv_dg.hip — conv contraction of a hidden layer, destination-grouped, with the A operand formed on the VECTOR
// ALUs by dedicated waves while other waves keep the matrix cores busy.
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
//     K range of scalar-output tile 4, the 16x16 sub-tiles (rh = w&1, ch = w>>1) of the three vector planes, and the
//     row tiles {w, w+4, ..} of the T product.  All four run the same program on different addresses.
//   * waves 4..11 (two per SIMD)  FORMING waves.  Each owns 4 destination atoms, two at a time: lane = (half h -> which of
//     the two destinations, u = channel).  Per edge e = (j -> i) of its destination the half-wave reads the source row
//     x_j from LDS (coalesced: the lanes of a half read consecutive channels of ONE row, so no bank conflict and no
//     per-lane gather) and accumulates  c x0_j[u],  c x1_j[u][m],  sum_m (c v_m) x1_j[u][m],  c (x1_j[u] x v)[m]  with the
//     per-edge coefficients c = h~_e[k], c v_m broadcast from a small per-wave LDS table, plus the vector-row
//     contribution of the scalar inputs through the re-association  out_m[i][w] += sum_e (c v_m) T_k[j][w],
//     T_k[j][w] = sum_u x0_j[u] W~[(k,u)][w]  (T_k is formed by the matrix waves: 2 x 120 x 32 FLOP per source atom instead
//     of 3 x per destination).
//   The VALU of a SIMD is otherwise idle under an MFMA-bound kernel, so the forming work hides behind the contraction.
//
//   Per hidden unit k, two phases separated by LDS-only barriers (single-buffered A tiles: 46 KB instead of 92 KB, which
//   leaves LDS for source spans of ~90 atoms):
//     P1(k): matrix waves  X(k) x W  (scalar inputs -> scalar rows, K = 120);        forming waves write Y(k)
//     P2(k): matrix waves  Y(k) x W  (dot -> scalar rows; x1, cross -> vector rows) and T(k+1);   forming waves write X(k+1)
//   MFMA work per (32-destination tile, k): 300 + 304 units of v_mfma_f32_32x32x2 (64 cycles), against 637 algorithmic.
//
// Work distribution, partial slabs, fixed summation order: as jamun_conv_fused.hip (host-built segment lists).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jamun_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DG_WAVES 12
#define DG_THREADS (64 * DG_WAVES)
#define DG_XST 284   // xs row: x0 [0,120) | x1 as [u][4] (x,y,z,0) [120,248) | T_k[j][w] [248,280) | pad: 4 * 71 floats
#define DG_XS0 124   // X tile row stride (120 + 4):  4 * 31
#define DG_YD 36     // Yd tile row stride (32 + 4):   4 * 9
#define DG_YV 68     // Yv tile row stride (64 + 4):   4 * 17

#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define RFL(v) __builtin_amdgcn_readfirstlane(v)

__host__ __device__ inline size_t dg_lds_floats(int rs, int pmax) {
  // xs | X (+4: the dummy quarter group reads 4 floats past the last row) | Yd | Yv | tabA | tabJ | deg
  return (size_t)rs * DG_XST + 32 * DG_XS0 + 8 + 32 * DG_YD + 3 * 32 * DG_YV + (size_t)32 * pmax * 4 + (size_t)32 * pmax + 32;
}

__global__ __launch_bounds__(DG_THREADS) void k_conv_dg(DgArgs a) {
  extern __shared__ float4 lds4[];
  float* __restrict__ lds = reinterpret_cast<float*>(lds4);
  float* __restrict__ xs = lds;                                  // [RS][DG_XST]
  float* __restrict__ Xt = xs + (size_t)a.RS * DG_XST;           // [32][DG_XS0] (+8)
  float* __restrict__ Yd = Xt + 32 * DG_XS0 + 8;                 // [32][DG_YD]
  float* __restrict__ Yv = Yd + 32 * DG_YD;                      // [3][32][DG_YV]
  float4* __restrict__ tabA = reinterpret_cast<float4*>(Yv + 3 * 32 * DG_YV);  // [32][PMAX] (c, c vx, c vy, c vz)
  int* __restrict__ tabJ = reinterpret_cast<int*>(tabA + 32 * a.PMAX);          // [32][PMAX] byte offset of the source row in xs
  int* __restrict__ deg_lds = tabJ + 32 * a.PMAX;                               // [32]
  float* __restrict__ OL = Xt;   // [3][32][32]   segment end: vector rows accumulated by the forming waves (aliases X, Yd)
  float* __restrict__ QL = Yv;   // [4][16][64]   segment end: the four K-quarter partials of scalar tile 4 (aliases Yv)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = RFL(tid >> 6);
  const bool is_mat = wave < 4;
  const int PMAX = a.PMAX;

  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int4 sg0 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2];
    const int4 sg1 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2 + 1];
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) { kk = kk < 0 ? 0 : (kk < nk ? kk : nk - 1); return kk < k_run ? k_begin + kk : k_extra; };
    const int2 t_at = a.tile_atoms[tile];
    const int n0 = RFL(t_at.x), n_dst = RFL(t_at.y);
    const int2 span = a.tile_span[tile];
    const int s_lo = RFL(span.x), rows = RFL(span.y) - s_lo;
    const int rows16 = (rows + 15) & ~15;   // rows of the T product (multiples of the 16-row MFMA tile), <= RS
    const int n_rt = rows16 >> 4;

    // ---- segment prologue: source rows -> LDS (x1 re-laid as [u][4]); zero the A tiles and the T columns
    for (int idx = tid; idx < rows16 * (DG_XST / 4); idx += DG_THREADS) {
      const int j = idx / (DG_XST / 4), q = idx - j * (DG_XST / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < rows) {
        const float* __restrict__ xr = a.x + (size_t)(s_lo + j) * a.XS;
        if (q < 30) v = *reinterpret_cast<const float4*>(xr + 4 * q);             // x0 (120 = 30 x 4)
        else if (q < 62) { const float* p = xr + 120 + 3 * (q - 30); v = make_float4(p[0], p[1], p[2], 0.f); }  // x1[u] -> (x, y, z, 0)
      }
      *reinterpret_cast<float4*>(xs + (size_t)j * DG_XST + 4 * q) = v;
    }
    for (int idx = tid; idx < 32 * DG_XS0 + 8 + 32 * DG_YD + 3 * 32 * DG_YV; idx += DG_THREADS) Xt[idx] = 0.f;
    if (tid < 32) deg_lds[tid] = (tid < n_dst) ? a.deg[n0 + tid] : 0;
    __syncthreads();

    if (is_mat) {
      // =========================================== MATRIX waves ===========================================
      const int w = wave;
      const int r = lane & 31, hh = lane >> 5;      // 32x32x2 fragments: row / column r, K half hh
      const int r16 = lane & 15, kq = lane >> 4;    // 16x16x4 fragments: row / column r16, K quarter kq
      const int rh = w & 1, ch = w >> 1;            // this wave's 16x16 sub-tiles of the vector planes
      f32x16 accM, acc4;
      f32x4 accP[3];
#pragma unroll
      for (int q = 0; q < 16; ++q) { accM[q] = 0.f; acc4[q] = 0.f; }
#pragma unroll
      for (int m = 0; m < 3; ++m) accP[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* __restrict__ Xa = Xt + r * DG_XS0 + 4 * hh;
      const float* __restrict__ Da = Yd + r * DG_YD + 4 * hh;
      const float* __restrict__ Va = Yv + (16 * rh + r16) * DG_YV + 4 * kq;

      // weight blocks (64 lanes x float4) of one hidden unit, in the order this wave consumes them:
      //   P1: WX[w][g], g = 0..14, with WX[4][4 s + w] after g = 3, 7, 11, 14          (19 blocks: chunks 5 5 5 4)
      //   P2: WD[w][0..3], WD[4][w] | WV[ch][0..3] | WT[g][0..1], g = 0..7              (5 | 4 | 4 x 4)
      // (uniform block base in scalar registers + the lane as a 32-bit offset: one address register per load, not two)
      auto WX = [&](int k, int t, int g) { return (a.wx + ((size_t)k * 5 * 16 + t * 16 + g) * 64)[lane]; };
      auto WD = [&](int k, int t, int g) { return (a.wd + ((size_t)k * 5 * 4 + t * 4 + g) * 64)[lane]; };
      auto WV = [&](int k, int c, int g) { return (a.wv + ((size_t)k * 2 * 4 + c * 4 + g) * 64)[lane]; };
      auto WT = [&](int k, int g, int c) { return (a.wt + ((size_t)k * 8 * 2 + g * 2 + c) * 64)[lane]; };

      float4 B0[5], B1[5];
      auto load_x = [&](float4 (&B)[5], int k, int c) {  // chunk c of P1: groups 4c..4c+3 (+ the quarter group 4c + w)
#pragma unroll
        for (int i = 0; i < 4; ++i) B[i] = WX(k, w, (4 * c + i) < 15 ? 4 * c + i : 15);
        B[4] = WX(k, 4, 4 * c + w);  // group 15 (wave 3, chunk 3) is a zero block
      };
      auto run_x = [&](const float4 (&B)[5], int c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (4 * c + i >= 15) break;
          const float4 av = *reinterpret_cast<const float4*>(Xa + 8 * (4 * c + i));
          accM = MFMA32(av.x, B[i].x, accM);
          accM = MFMA32(av.y, B[i].y, accM);
          accM = MFMA32(av.z, B[i].z, accM);
          accM = MFMA32(av.w, B[i].w, accM);
        }
        const float4 aq = *reinterpret_cast<const float4*>(Xa + 8 * (4 * c + w));
        acc4 = MFMA32(aq.x, B[4].x, acc4);
        acc4 = MFMA32(aq.y, B[4].y, acc4);
        acc4 = MFMA32(aq.z, B[4].z, acc4);
        acc4 = MFMA32(aq.w, B[4].w, acc4);
      };
      auto load_d = [&](float4 (&B)[5], int k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) B[i] = WD(k, w, i);
        B[4] = WD(k, 4, w);
      };
      auto run_d = [&](const float4 (&B)[5]) {
        const float4 aq = *reinterpret_cast<const float4*>(Da + 8 * w);  // this wave's K quarter of scalar tile 4: group w
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 av = *reinterpret_cast<const float4*>(Da + 8 * i);
          accM = MFMA32(av.x, B[i].x, accM);
          accM = MFMA32(av.y, B[i].y, accM);
          accM = MFMA32(av.z, B[i].z, accM);
          accM = MFMA32(av.w, B[i].w, accM);
        }
        acc4 = MFMA32(aq.x, B[4].x, acc4);
        acc4 = MFMA32(aq.y, B[4].y, acc4);
        acc4 = MFMA32(aq.z, B[4].z, acc4);
        acc4 = MFMA32(aq.w, B[4].w, acc4);
      };
      auto load_v = [&](float4 (&B)[5], int k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) B[i] = WV(k, ch, i);
      };
      auto run_v = [&](const float4 (&B)[5]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float4 av[3];
#pragma unroll
          for (int m = 0; m < 3; ++m) av[m] = *reinterpret_cast<const float4*>(Va + m * 32 * DG_YV + 16 * i);
#pragma unroll
          for (int m = 0; m < 3; ++m) accP[m] = MFMA16(av[m].x, B[i].x, accP[m]);
#pragma unroll
          for (int m = 0; m < 3; ++m) accP[m] = MFMA16(av[m].y, B[i].y, accP[m]);
#pragma unroll
          for (int m = 0; m < 3; ++m) accP[m] = MFMA16(av[m].z, B[i].z, accP[m]);
#pragma unroll
          for (int m = 0; m < 3; ++m) accP[m] = MFMA16(av[m].w, B[i].w, accP[m]);
        }
      };
      // T product of hidden unit k: T[j][w'] = sum_u x0_j[u] W[(k,u)][w'], row tiles rt = w, w + 4 (16 rows each), both
      // 16-column halves per row tile (two independent accumulators keep the 16x16x4 MFMA at its issue rate)
      f32x4 accT[2][2];
      auto load_t = [&](float4 (&B)[5], int k, int c) {  // chunk c: groups 2c, 2c+1, both column halves
#pragma unroll
        for (int i = 0; i < 4; ++i) B[i] = WT(k, 2 * c + (i >> 1), i & 1);
      };
      auto run_t = [&](const float4 (&B)[5], int c) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (w + 4 * p >= n_rt) break;  // wave-uniform
          const float* __restrict__ Ta = xs + (size_t)(16 * (w + 4 * p) + r16) * DG_XST + 4 * kq;
#pragma unroll
          for (int gi = 0; gi < 2; ++gi) {
            const float4 av = *reinterpret_cast<const float4*>(Ta + 16 * (2 * c + gi));
            accT[p][0] = MFMA16(av.x, B[2 * gi].x, accT[p][0]);
            accT[p][1] = MFMA16(av.x, B[2 * gi + 1].x, accT[p][1]);
            accT[p][0] = MFMA16(av.y, B[2 * gi].y, accT[p][0]);
            accT[p][1] = MFMA16(av.y, B[2 * gi + 1].y, accT[p][1]);
            accT[p][0] = MFMA16(av.z, B[2 * gi].z, accT[p][0]);
            accT[p][1] = MFMA16(av.z, B[2 * gi + 1].z, accT[p][1]);
            accT[p][0] = MFMA16(av.w, B[2 * gi].w, accT[p][0]);
            accT[p][1] = MFMA16(av.w, B[2 * gi + 1].w, accT[p][1]);
          }
        }
      };
      auto zero_t = [&]() {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int c = 0; c < 2; ++c) accT[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      };
      auto store_t = [&]() {  // D[row = 4 kq + q][col = r16] of each 16x16 sub-tile -> xs[row][248 + col]
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (w + 4 * p >= n_rt) break;
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[(size_t)(16 * (w + 4 * p) + 4 * kq + q) * DG_XST + 248 + 16 * c + r16] = accT[p][c][q];
        }
      };
      // (row tiles beyond w + 4 — spans above 128 rows — do not occur: the host caps the span at RS <= 128)

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
        __syncthreads();
        continue;
      }
      // "P2(-1)": T of the first hidden unit (no Y yet)
      {
        const int k0 = k_of(0);
        zero_t();
        load_t(B0, k0, 0);
        load_t(B1, k0, 1);
        run_t(B0, 0); load_t(B0, k0, 2);
        run_t(B1, 1); load_t(B1, k0, 3);
        run_t(B0, 2); load_x(B0, k0, 0);
        run_t(B1, 3); load_x(B1, k0, 1);
        store_t();
      }
      LDS_BARRIER();
      for (int it = 0; it < nk; ++it) {
        const int k = k_of(it), kn = k_of(it + 1);
        const bool has_next = it + 1 < nk;
        // ---- P1: X(k) x W  (B0 = chunk 0, B1 = chunk 1 already in flight)
        run_x(B0, 0); load_x(B0, k, 2);
        run_x(B1, 1); load_x(B1, k, 3);
        run_x(B0, 2); load_d(B0, k);
        run_x(B1, 3); load_v(B1, k);
        LDS_BARRIER();
        // ---- P2: Y(k) x W, then T(k+1)
        run_d(B0); load_t(B0, kn, 0);
        run_v(B1); load_t(B1, kn, 1);
        if (has_next) {
          zero_t();
          run_t(B0, 0); load_t(B0, kn, 2);
          run_t(B1, 1); load_t(B1, kn, 3);
          run_t(B0, 2); load_x(B0, kn, 0);
          run_t(B1, 3); load_x(B1, kn, 1);
          store_t();
        }
        LDS_BARRIER();
      }
      // ---- segment end: scalar tile w straight to the slab; tile 4 and the vector planes through LDS
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        if (row < n_dst) a.partial0[((size_t)slab * a.n_pad + n0 + row) * (a.nt0 * 32) + w * 32 + r] = accM[q];
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) QL[(w * 16 + q) * 64 + lane] = acc4[q];
      __syncthreads();  // (forming waves have written OL)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {  // wave w sums registers 4w..4w+3 of the four partials, in wave order
        const int q = 4 * w + qq;
        float s = QL[(0 * 16 + q) * 64 + lane];
        s += QL[(1 * 16 + q) * 64 + lane];
        s += QL[(2 * 16 + q) * 64 + lane];
        s += QL[(3 * 16 + q) * 64 + lane];
        const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
        if (row < n_dst && 4 * 32 + r < a.nt0 * 32) a.partial0[((size_t)slab * a.n_pad + n0 + row) * (a.nt0 * 32) + 4 * 32 + r] = s;
      }
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 16 * rh + 4 * kq + q, col = 16 * ch + r16;
          if (row < n_dst) a.partial1[(((size_t)slab * a.n_pad + n0 + row) * 3 + m) * 32 + col] = accP[m][q] + OL[(m * 32 + row) * 32 + col];
        }
      __syncthreads();  // the next segment rewrites the tiles
    } else {
      // =========================================== FORMING waves ===========================================
      const int fw = wave - 4;                    // 0..7: destinations 4 fw .. 4 fw + 3
      const int h = lane >> 5, u = lane & 31;
      // per destination pair dp: this lane's edge slots t = u (page 0) and u + 32 (page 1) of destination i = 4 fw + 2 dp + h
      float evx[2][2], evy[2][2], evz[2][2];
      int hidx[2][2];
      int P[2];
      const bool two_pages = a.S > 32;
#pragma unroll
      for (int dp = 0; dp < 2; ++dp) {
        const int i = 4 * fw + 2 * dp + h;
        const int dg = deg_lds[i];
        P[dp] = RFL(max(deg_lds[4 * fw + 2 * dp], deg_lds[4 * fw + 2 * dp + 1]));
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
          evx[dp][pg] = evy[dp][pg] = evz[dp][pg] = 0.f;
          hidx[dp][pg] = -1;
          if (pg == 1 && !two_pages) continue;
          const int t = u + 32 * pg;
          int jofs = 0;
          if (t < dg) {
            const int slot = (n0 + i) * a.S + t;
            const int jl = (a.esrc[slot] & 0x7fffffff) - s_lo;
            if (jl >= 0 && jl < rows) {  // (a source outside the span belongs to another row-block tile of these destinations)
              const float4 ge = a.egeo[slot];
              evx[dp][pg] = ge.x; evy[dp][pg] = ge.y; evz[dp][pg] = ge.z;
              hidx[dp][pg] = slot;
              jofs = jl * DG_XST * 4;
            }
          }
          if (t < PMAX) tabJ[i * PMAX + t] = jofs;
        }
      }
      // A tile of a large molecule (destination chunk x source row block) may have no edge at all: checked below by all waves
      int any_edge = 0;
#pragma unroll
      for (int dp = 0; dp < 2; ++dp)
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) any_edge |= hidx[dp][pg] >= 0;
      if (a.row_blocks && !__syncthreads_or(any_edge)) {  // (only batches with row-block tiles pay for the vote)
        __syncthreads();
        continue;
      }

      float O[2][3];  // vector rows from scalar inputs, accumulated over the hidden units of the segment: lane = (h, w' = u)
#pragma unroll
      for (int dp = 0; dp < 2; ++dp) O[dp][0] = O[dp][1] = O[dp][2] = 0.f;
      float hv[2][2];  // h~ of this lane's edges for the NEXT hidden unit (prefetched)
      auto load_h = [&](int k) {
        const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
        for (int dp = 0; dp < 2; ++dp)
#pragma unroll
          for (int pg = 0; pg < 2; ++pg) hv[dp][pg] = hk[hidx[dp][pg] >= 0 ? hidx[dp][pg] : n0 * a.S];
      };
      auto write_tab = [&]() {  // coefficients of the prefetched hidden unit -> this wave's private table rows
#pragma unroll
        for (int dp = 0; dp < 2; ++dp) {
          const int i = 4 * fw + 2 * dp + h;
#pragma unroll
          for (int pg = 0; pg < 2; ++pg) {
            if (pg == 1 && !two_pages) continue;
            const int t = u + 32 * pg;
            const float c = hidx[dp][pg] >= 0 ? hv[dp][pg] : 0.f;
            if (t < PMAX) tabA[i * PMAX + t] = make_float4(c, c * evx[dp][pg], c * evy[dp][pg], c * evz[dp][pg]);
          }
        }
      };
      const char* __restrict__ xsb = reinterpret_cast<const char*>(xs);
      // X(k): scalar inputs, lanes u < 30 own channels 4u..4u+3 (lanes 30, 31 compute on x1 data and are not stored)
      auto form_x = [&]() {
#pragma unroll
        for (int dp = 0; dp < 2; ++dp) {
          const int i = 4 * fw + 2 * dp + h;
          float ax0 = 0.f, ax1 = 0.f, ax2 = 0.f, ax3 = 0.f;
          const float4* __restrict__ ta = tabA + i * PMAX;
          const int* __restrict__ tj = tabJ + i * PMAX;
          const int n = P[dp];
#pragma unroll 2
          for (int t = 0; t < n; ++t) {
            const float c = ta[t].x;
            const float4 xv = *reinterpret_cast<const float4*>(xsb + tj[t] + u * 16);
            ax0 = fmaf(c, xv.x, ax0); ax1 = fmaf(c, xv.y, ax1); ax2 = fmaf(c, xv.z, ax2); ax3 = fmaf(c, xv.w, ax3);
          }
          if (u < 30) *reinterpret_cast<float4*>(Xt + i * DG_XS0 + 4 * u) = make_float4(ax0, ax1, ax2, ax3);
        }
      };
      // Y(k): vector inputs (lane u = channel u) + the T term (lane u = output channel w')
      auto form_y = [&]() {
#pragma unroll
        for (int dp = 0; dp < 2; ++dp) {
          const int i = 4 * fw + 2 * dp + h;
          float a1x = 0.f, a1y = 0.f, a1z = 0.f, ad = 0.f, acx = 0.f, acy = 0.f, acz = 0.f;
          float o0 = O[dp][0], o1 = O[dp][1], o2 = O[dp][2];
          const float4* __restrict__ ta = tabA + i * PMAX;
          const int* __restrict__ tj = tabJ + i * PMAX;
          const int n = P[dp];
#pragma unroll 2
          for (int t = 0; t < n; ++t) {
            const float4 cf = ta[t];  // c, c vx, c vy, c vz
            const char* __restrict__ row = xsb + tj[t];
            const float4 xv = *reinterpret_cast<const float4*>(row + 480 + u * 16);  // x1_j[u] = (x, y, z, 0)
            const float tw = *reinterpret_cast<const float*>(row + 992 + u * 4);     // T_k[j][w' = u]
            a1x = fmaf(cf.x, xv.x, a1x); a1y = fmaf(cf.x, xv.y, a1y); a1z = fmaf(cf.x, xv.z, a1z);
            ad = fmaf(cf.y, xv.x, ad); ad = fmaf(cf.z, xv.y, ad); ad = fmaf(cf.w, xv.z, ad);
            // (x1 x v)[m] = x1[m+1] v[m+2] - x1[m+2] v[m+1]
            acx = fmaf(cf.w, xv.y, acx); acx = fmaf(-cf.z, xv.z, acx);
            acy = fmaf(cf.y, xv.z, acy); acy = fmaf(-cf.w, xv.x, acy);
            acz = fmaf(cf.z, xv.x, acz); acz = fmaf(-cf.y, xv.y, acz);
            o0 = fmaf(cf.y, tw, o0); o1 = fmaf(cf.z, tw, o1); o2 = fmaf(cf.w, tw, o2);
          }
          O[dp][0] = o0; O[dp][1] = o1; O[dp][2] = o2;
          Yd[i * DG_YD + u] = ad;
          Yv[(0 * 32 + i) * DG_YV + u] = a1x; Yv[(1 * 32 + i) * DG_YV + u] = a1y; Yv[(2 * 32 + i) * DG_YV + u] = a1z;
          Yv[(0 * 32 + i) * DG_YV + 32 + u] = acx; Yv[(1 * 32 + i) * DG_YV + 32 + u] = acy; Yv[(2 * 32 + i) * DG_YV + 32 + u] = acz;
        }
      };

      // "P2(-1)": coefficients and X of the first hidden unit
      load_h(k_of(0));
      write_tab();
      load_h(k_of(1));
      form_x();
      LDS_BARRIER();
      for (int it = 0; it < nk; ++it) {
        // ---- P1: Y(k) with the table of k (written in the previous P2) and T(k)
        form_y();
        LDS_BARRIER();
        // ---- P2: table of k+1, X(k+1)
        if (it + 1 < nk) {
          write_tab();
          load_h(k_of(it + 2));
          form_x();
        }
        LDS_BARRIER();
      }
      // ---- segment end: hand the accumulated vector rows to the plane owners (OL aliases X / Yd: dead after the last barrier)
#pragma unroll
      for (int dp = 0; dp < 2; ++dp) {
        const int i = 4 * fw + 2 * dp + h;
#pragma unroll
        for (int m = 0; m < 3; ++m) OL[(m * 32 + i) * 32 + u] = O[dp][m];
      }
      __syncthreads();
      __syncthreads();
    }
  }
}

size_t conv_dg_lds_bytes(int rs, int pmax) { return sizeof(float) * ((dg_lds_floats(rs, pmax) + 3) & ~(size_t)3); }

int launch_conv_dg(const DgArgs& a, int grid, hipStream_t st) {
  const size_t smem = conv_dg_lds_bytes(a.RS, a.PMAX);
  if (smem > JAMUN_MAX_DYN_LDS) return -2;
  if (a.RS > 128 || (a.RS & 15) || a.XS != 216 || a.nt0 != 5) return -1;
  hipLaunchKernelGGL(k_conv_dg, dim3(grid), dim3(DG_THREADS), smem, st, a);
  return 0;
}

int conv_dg_set_max_lds() {
  return hipFuncSetAttribute((const void*)k_conv_dg, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) == hipSuccess ? 0 : -1;
}
