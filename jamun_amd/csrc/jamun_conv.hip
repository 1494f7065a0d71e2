// jamun_conv.hip — the conv contraction kernel (dominant kernel of the path), gfx950.
//
// For one ConvBlock of the reference (src/jamun/e3tools/nn/_conv.py:93-119) the per-destination message is
//     m_i[w] = 1/deg_i * sum_{e->i} sum_k h~_e[k] sum_u W~2[k,(path,u,w)] * zeta_e[u]
// with h~_e = [SiLU(W1 a_e + b1), 1] (65 values), zeta_e[u] the CG-contracted product of the source features and the
// edge unit vector, and all path / Clebsch-Gordan / noise-scaling constants folded into the packed weights.
// Destination-grouped association: A[(i,plane)][(k,u)] = sum_{e->i} h~_e[k] zeta_e[plane][u] is formed on chip and
// contracted with the packed weights on the matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
// Work decomposition
//   workgroup = 4 waves; tile = 32 consecutive destination atoms; K-slice = a range of hidden units k (8 slices, so
//   that under round-robin dispatch all workgroups of one slice share one XCD's L2 and its 1/8 of the weights stays
//   resident there — a speed choice only).  RC = 1 plane for scalar outputs, 3 planes (x,y,z) for vector outputs.
//   Once per workgroup the tile's edge table (source atom, unit vector) is staged in LDS.  Per chunk = (u-block of
//   <= 64 input channels) x (k-subgroup of <= KSUB hidden units):
//     form : each wave builds the A rows of 8 atoms, lanes = u.  Per edge it needs three kinds of operand: the source
//            feature row segment (one coalesced 256-B global load, L1/L2 resident), and two wave-uniform records —
//            (source, vhat) and h~[k-subgroup] — which are LDS broadcast reads (in-order, so they pipeline under
//            counted waits; scalar-memory loads would force lgkmcnt(0) on every edge).  Rows are written transposed
//            into LDS with row stride 33 (conflict-free for the write and for the MFMA A-fragment read).
//     mma  : the 4 waves split the chunk's K-steps; each streams its own 16-byte weight fragments (4 K-steps per
//            lane) from L2 and accumulates all RC x NT 32x32 tiles.  The next chunk's h~ records are staged meanwhile.
//   Partial sums are reduced across waves through LDS in a fixed order and written to a per-slice slab that the
//   node-update kernel sums (no atomics: bitwise reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jamun_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define A_ROW 33

template <int RC, int KSUB>
struct ConvLds {
  static constexpr int A_PLANE = (KSUB * 64 + 8) * A_ROW;  // floats per plane of the transposed A tile
  static constexpr int HST = KSUB <= 2 ? 2 : (KSUB <= 4 ? 4 : 8);  // record stride (floats) of the staged h~ values
};

// One batch = up to 4 in-edges of one destination atom.  load_batch() issues every memory operation of the batch
// (LDS broadcast records + one coalesced global feature load per edge); compute_batch() consumes registers only, so
// two batches can be kept in flight (double buffering below) and the loads of batch b+1 overlap the FMAs of batch b.
template <int RC, int KS, int TYPE>
struct EdgeBatch {
  static constexpr int NX = (TYPE == JAMUN_T_X0 || TYPE == JAMUN_T_X0V) ? 1 : 3;
  float4 gj[4];
  float hk[4][KS];
  float xv[4][NX];
  int il, last;
};

template <int RC, int KSUB, int KS, int TYPE>
__device__ __forceinline__ void load_batch(EdgeBatch<RC, KS, TYPE>& B, int entry, const float4* __restrict__ g_lds,
                                           const float* __restrict__ h_lds, const ConvArgs& a, int xc) {
  using L = ConvLds<RC, KSUB>;
  // entry: il | t0 << 8 | last << 24   (wave-uniform).  Edge slots are padded to a multiple of 4 per atom: padding
  // slots carry h~ = 0 and a valid source row, so no per-edge predicate is needed.
  B.il = entry & 0xff;
  B.last = entry >> 24;
  const int slot0 = B.il * a.S4 + ((entry >> 8) & 0xff);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    B.gj[u] = g_lds[slot0 + u];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) B.hk[u][kk] = h_lds[(size_t)(slot0 + u) * L::HST + kk];
  }
  const char* __restrict__ xb = reinterpret_cast<const char*>(a.x) + xc * 4;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int off = __builtin_amdgcn_readfirstlane(__float_as_int(B.gj[u].x));  // byte offset of the source row
    const float* __restrict__ xp = reinterpret_cast<const float*>(xb + off);
#pragma unroll
    for (int q = 0; q < EdgeBatch<RC, KS, TYPE>::NX; ++q) B.xv[u][q] = xp[q];
  }
}

template <int RC, int KSUB, int KS, int TYPE>
__device__ __forceinline__ void compute_batch(const EdgeBatch<RC, KS, TYPE>& B, float (&g)[RC][KS], float* __restrict__ A_lds,
                                              int nu, int lane, bool active, bool is_cross) {
  using L = ConvLds<RC, KSUB>;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    {
      const float4 gj = B.gj[u];
      float z[RC];
      if (TYPE == JAMUN_T_X0) {
        z[0] = B.xv[u][0];
      } else if (TYPE == JAMUN_T_DOT) {
        z[0] = B.xv[u][0] * gj.y + B.xv[u][EdgeBatch<RC, KS, TYPE>::NX > 1 ? 1 : 0] * gj.z + B.xv[u][EdgeBatch<RC, KS, TYPE>::NX > 2 ? 2 : 0] * gj.w;
      } else if (TYPE == JAMUN_T_X0V) {
        const float x0 = B.xv[u][0];
        z[0] = x0 * gj.y;
        if (RC == 3) { z[1 % RC] = x0 * gj.z; z[2 % RC] = x0 * gj.w; }
      } else {  // JAMUN_T_X1C: first half of the lanes x1[u'][m], second half (x1[u'] x vhat)[m]
        const float x0 = B.xv[u][0], x1 = B.xv[u][EdgeBatch<RC, KS, TYPE>::NX > 1 ? 1 : 0], x2 = B.xv[u][EdgeBatch<RC, KS, TYPE>::NX > 2 ? 2 : 0];
        const float cx = x1 * gj.w - x2 * gj.z, cy = x2 * gj.y - x0 * gj.w, cz = x0 * gj.z - x1 * gj.y;
        z[0] = is_cross ? cx : x0;
        if (RC == 3) { z[1 % RC] = is_cross ? cy : x1; z[2 % RC] = is_cross ? cz : x2; }
      }
#pragma unroll
      for (int c = 0; c < RC; ++c)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) g[c][kk] = fmaf(B.hk[u][kk], z[c], g[c][kk]);
    }
  }
  if (B.last) {  // wave-uniform: this batch closes its destination atom
    if (active) {
#pragma unroll
      for (int c = 0; c < RC; ++c)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) A_lds[c * L::A_PLANE + (kk * nu + lane) * A_ROW + B.il] = g[c][kk];
    }
#pragma unroll
    for (int c = 0; c < RC; ++c)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) g[c][kk] = 0.f;
  }
}

template <int RC, int KSUB, int KS, int TYPE>
__device__ __forceinline__ void form_rows(float* __restrict__ A_lds, const float4* __restrict__ g_lds,
                                          const float* __restrict__ h_lds, const int* __restrict__ blist, int nb,
                                          const ConvArgs& a, int lane, int nu, int xcol) {
  const bool active = lane < nu;
  const bool is_cross = (xcol & JAMUN_XOFF_CROSS) != 0;
  const int xc = xcol & 0xffff;
  float g[RC][KS];
#pragma unroll
  for (int c = 0; c < RC; ++c)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) g[c][kk] = 0.f;
  EdgeBatch<RC, KS, TYPE> B0, B1;
  load_batch<RC, KSUB, KS, TYPE>(B0, __builtin_amdgcn_readfirstlane(blist[0]), g_lds, h_lds, a, xc);
  for (int b = 0; b < nb; b += 2) {
    if (b + 1 < nb) load_batch<RC, KSUB, KS, TYPE>(B1, __builtin_amdgcn_readfirstlane(blist[b + 1]), g_lds, h_lds, a, xc);
    compute_batch<RC, KSUB, KS, TYPE>(B0, g, A_lds, nu, lane, active, is_cross);
    if (b + 2 < nb) load_batch<RC, KSUB, KS, TYPE>(B0, __builtin_amdgcn_readfirstlane(blist[b + 2]), g_lds, h_lds, a, xc);
    if (b + 1 < nb) compute_batch<RC, KSUB, KS, TYPE>(B1, g, A_lds, nu, lane, active, is_cross);
  }
}

template <int RC, int KSUB, int KS>
__device__ __forceinline__ void form_dispatch(int type, float* A_lds, const float4* g_lds, const float* h_lds,
                                              const int* blist, int nb, const ConvArgs& a, int lane, int nu, int xcol) {
  if (RC == 1) {
    if (type == JAMUN_T_X0) form_rows<RC, KSUB, KS, JAMUN_T_X0>(A_lds, g_lds, h_lds, blist, nb, a, lane, nu, xcol);
    else form_rows<RC, KSUB, KS, JAMUN_T_DOT>(A_lds, g_lds, h_lds, blist, nb, a, lane, nu, xcol);
  } else {
    if (type == JAMUN_T_X0V) form_rows<RC, KSUB, KS, JAMUN_T_X0V>(A_lds, g_lds, h_lds, blist, nb, a, lane, nu, xcol);
    else form_rows<RC, KSUB, KS, JAMUN_T_X1C>(A_lds, g_lds, h_lds, blist, nb, a, lane, nu, xcol);
  }
}

// stage the h~ values of one chunk's k-subgroup for every edge slot of the tile
template <int RC, int KSUB>
__device__ __forceinline__ void stage_h(float* __restrict__ h_lds, const ConvArgs& a, int n0, int k0, int ks, int tid) {
  using L = ConvLds<RC, KSUB>;
  const int per_node = a.S4 * ks;
  for (int idx = tid; idx < 32 * per_node; idx += 256) {
    const int il = idx / per_node, rem = idx - il * per_node;
    const int t = rem / ks, kk = rem - t * ks;
    const int i = n0 + il;
    float v = 0.f;  // padding slots (t >= deg) contribute nothing
    if (i < a.n_atoms && t < a.deg[i]) v = a.h[(size_t)(k0 + kk) * a.h_kstride + (size_t)i * a.S + t];
    h_lds[((size_t)il * a.S4 + t) * L::HST + kk] = v;
  }
}

template <int RC, int NT, int KSUB>
__global__ __launch_bounds__(256, 2) void k_conv(ConvArgs a) {
  using L = ConvLds<RC, KSUB>;
  extern __shared__ float4 lds4[];
  float* __restrict__ lds = reinterpret_cast<float*>(lds4);
  float* __restrict__ A_lds = lds;                                           // [RC][KSUB*64+8][33]
  float4* __restrict__ g_lds = reinterpret_cast<float4*>(lds + ((RC * L::A_PLANE + 3) & ~3));  // [32][S]
  float* __restrict__ h_lds = reinterpret_cast<float*>(g_lds + 32 * a.S4);    // [32][S4][HST]
  int* __restrict__ b_lds = reinterpret_cast<int*>(h_lds + 32 * a.S4 * L::HST); // [4 waves][JAMUN_MAX_BATCH] batch list
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction; tell the compiler
  const int r = lane & 31, hh = lane >> 5;
  const int slice = blockIdx.x % a.n_slices;
  const int tile = blockIdx.x / a.n_slices;
  const int n0 = tile * 32;

  f32x16 acc[RC][NT];
#pragma unroll
  for (int c = 0; c < RC; ++c)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[c][nt][q] = 0.f;

  const int c_begin = a.slice_ptr[slice], c_end = a.slice_ptr[slice + 1];
  // edge table of the tile: (source, vhat) per slot
  for (int idx = tid; idx < 32 * a.S4; idx += 256) {
    const int il = idx / a.S4, t = idx - il * a.S4;
    const int i = n0 + il;
    float4 rec = make_float4(__int_as_float(0), 0.f, 0.f, 0.f);  // padding: row 0 (valid memory), zero geometry
    if (i < a.n_atoms && t < a.deg[i]) {
      const size_t e = (size_t)i * a.S + t;
      const float4 geo = a.egeo[e];
      rec = make_float4(__int_as_float((a.esrc[e] & 0x7fffffff) * a.XS * 4), geo.x, geo.y, geo.z);
    }
    g_lds[idx] = rec;
  }
  if (c_begin < c_end) {
    const int4 cd = a.chunks[c_begin];
    stage_h<RC, KSUB>(h_lds, a, n0, cd.y & 0xffff, cd.y >> 16, tid);
  }
  // batch list of this wave: its 8 destination atoms' in-edges in groups of <= 4 (an atom without edges gets one empty
  // batch so that its A rows are still written as zeros)
  int* __restrict__ blist = b_lds + wave * JAMUN_MAX_BATCH;
  int nb;
  {
    const int i_l = n0 + wave * 8 + (lane & 7);
    const int dg = (i_l < a.n_atoms) ? a.deg[i_l] : 0;
    const int nbat = dg > 0 ? (dg + 3) >> 2 : 1;
    int pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int v = __shfl(nbat, k);
      if (k < (lane & 7)) pre += v;
      tot += v;
    }
    nb = __builtin_amdgcn_readfirstlane(tot);
    int pks[8], nks[8], dks[8];  // gathered with all lanes active (a shuffle inside the divergent loop below would not be)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      pks[k] = __shfl(pre, k);
      nks[k] = __shfl(nbat, k);
      dks[k] = __shfl(dg, k);
    }
    for (int b = lane; b < nb; b += 64) {
      int ent = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int pk = pks[k], nk = nks[k], dk = dks[k];
        if (b >= pk && b < pk + nk) ent = (wave * 8 + k) | (((b - pk) * 4) << 8) | ((b == pk + nk - 1 ? 1 : 0) << 24);
        (void)dk;
      }
      blist[b] = ent;
    }
  }
  __syncthreads();

  for (int ci = c_begin; ci < c_end; ++ci) {
    const int4 cd = a.chunks[ci];  // {ublk, k0 | ks << 16, first weight group, number of weight groups}
    const int4 ub = a.ublk[cd.x];  // {type, nu, xcol0, width}
    const int type = ub.x, nu = ub.y;
    const int ks = cd.y >> 16;
    const int xcol = a.lane_xoff[cd.x * 64 + lane] + ub.z;
    if (ks == KSUB) form_dispatch<RC, KSUB, KSUB>(type, A_lds, g_lds, h_lds, blist, nb, a, lane, nu, xcol);
    else form_dispatch<RC, KSUB, KSUB - 1>(type, A_lds, g_lds, h_lds, blist, nb, a, lane, nu, xcol);
    {  // zero the K-step padding rows (K-steps are issued in groups of 4 = 8 rows of A)
      const int kused = ks * nu, kpad = cd.w * 8;
#pragma unroll
      for (int c = 0; c < RC; ++c)
        for (int idx = kused * A_ROW + tid; idx < kpad * A_ROW; idx += 256) A_lds[c * L::A_PLANE + idx] = 0.f;
    }
    __syncthreads();
    if (ci + 1 < c_end) {  // h~ of the next chunk lands in LDS while the matrix cores work on this one
      const int4 nx = a.chunks[ci + 1];
      stage_h<RC, KSUB>(h_lds, a, n0, nx.y & 0xffff, nx.y >> 16, tid);
    }
    // K-steps of this chunk, in groups of 4, split contiguously over the 4 waves
    const int G = cd.w;
    const int g_begin = (G * wave) >> 2, g_end = (G * (wave + 1)) >> 2;
    const float4* __restrict__ wp = a.wpack + ((size_t)cd.z * NT) * 64 + lane;
    // two weight buffers used in place, alternately: the buffer the next group needs is fetched before this group's MFMAs
    // (a copy `b = bn; bn = load` makes the compiler wait for the load it has just issued)
    float4 b0[NT], b1[NT];
    auto wfetch = [&](float4 (&b)[NT], int g) {
      const int gc = g < g_end ? g : g_end - 1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = wp[((size_t)gc * NT + nt) * 64];
    };
    auto kgroup = [&](const float4 (&b)[NT], int g) {
      const int q0 = g * 4;
#pragma unroll
      for (int c = 0; c < RC; ++c) {
        const float* __restrict__ Ap = A_lds + c * L::A_PLANE + (2 * q0 + hh) * A_ROW + r;
        const float a0 = Ap[0], a1 = Ap[2 * A_ROW], a2 = Ap[4 * A_ROW], a3 = Ap[6 * A_ROW];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[c][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[nt].x, acc[c][nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[c][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[nt].y, acc[c][nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[c][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b[nt].z, acc[c][nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[c][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b[nt].w, acc[c][nt], 0, 0, 0);
      }
    };
    if (g_begin < g_end) {
      wfetch(b0, g_begin);
      int g = g_begin;
      for (; g + 1 < g_end; g += 2) {
        wfetch(b1, g + 1);
        kgroup(b0, g);
        wfetch(b0, g + 2);
        kgroup(b1, g + 1);
      }
      if (g < g_end) kgroup(b0, g);
    }
    __syncthreads();
  }

  // cross-wave reduction in a fixed order ((w0 + w2) + (w1 + w3)) through LDS, then wave 0 stores the slab rows
  constexpr int NACC = RC * NT * 16;
  float* red = lds;  // 2 * NACC * 64 floats, fits inside the A region (checked by static_assert below)
  static_assert(2 * NACC * 64 <= RC * L::A_PLANE, "cross-wave reduction buffer does not fit in the A tile");
  if (wave >= 2) {
#pragma unroll
    for (int c = 0; c < RC; ++c)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[((wave - 2) * NACC + (c * NT + nt) * 16 + q) * 64 + lane] = acc[c][nt][q];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int c = 0; c < RC; ++c)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[c][nt][q] += red[(wave * NACC + (c * NT + nt) * 16 + q) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int c = 0; c < RC; ++c)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[((c * NT + nt) * 16 + q) * 64 + lane] = acc[c][nt][q];
  }
  __syncthreads();
  if (wave == 0) {
    const size_t row_w = (size_t)RC * NT * 32;
#pragma unroll
    for (int c = 0; c < RC; ++c)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float v = acc[c][nt][q] + red[((c * NT + nt) * 16 + q) * 64 + lane];
          const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
          a.partial[((size_t)slice * a.n_pad + n0 + row) * row_w + (size_t)c * NT * 32 + nt * 32 + r] = v;
        }
  }
}

template <int RC, int KSUB>
static size_t conv_lds_bytes(int S) {
  using L = ConvLds<RC, KSUB>;
  const int S4 = (S + 3) & ~3;
  return sizeof(float) * (size_t)(((RC * L::A_PLANE + 3) & ~3) + 32 * S4 * 4 + 32 * S4 * L::HST + 4 * JAMUN_MAX_BATCH);
}

int launch_conv(const ConvArgs& a, int rc, int nt, hipStream_t st) {
  const int grid = a.n_tiles * a.n_slices;
  if (rc == 1) {
    const size_t smem = conv_lds_bytes<1, JAMUN_KSUB0>(a.S);
    if (smem > JAMUN_MAX_DYN_LDS) return -2;
    switch (nt) {
      case 1: hipLaunchKernelGGL((k_conv<1, 1, JAMUN_KSUB0>), dim3(grid), dim3(256), smem, st, a); return 0;
      case 2: hipLaunchKernelGGL((k_conv<1, 2, JAMUN_KSUB0>), dim3(grid), dim3(256), smem, st, a); return 0;
      case 3: hipLaunchKernelGGL((k_conv<1, 3, JAMUN_KSUB0>), dim3(grid), dim3(256), smem, st, a); return 0;
      case 4: hipLaunchKernelGGL((k_conv<1, 4, JAMUN_KSUB0>), dim3(grid), dim3(256), smem, st, a); return 0;
      case 5: hipLaunchKernelGGL((k_conv<1, 5, JAMUN_KSUB0>), dim3(grid), dim3(256), smem, st, a); return 0;
      default: return -1;
    }
  } else if (rc == 3) {
    const size_t smem = conv_lds_bytes<3, JAMUN_KSUB1>(a.S);
    if (smem > JAMUN_MAX_DYN_LDS) return -2;
    switch (nt) {
      case 1: hipLaunchKernelGGL((k_conv<3, 1, JAMUN_KSUB1>), dim3(grid), dim3(256), smem, st, a); return 0;
      case 2: hipLaunchKernelGGL((k_conv<3, 2, JAMUN_KSUB1>), dim3(grid), dim3(256), smem, st, a); return 0;
      default: return -1;
    }
  }
  return -1;
}

int conv_set_max_lds() {
  // opt in to > 64 KiB of dynamic LDS for every instantiation
  hipError_t e = hipSuccess;
#define SETATTR(RC, NT, KS)                                                                                      \
  if (e == hipSuccess)                                                                                           \
    e = hipFuncSetAttribute((const void*)k_conv<RC, NT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS);
  SETATTR(1, 1, JAMUN_KSUB0) SETATTR(1, 2, JAMUN_KSUB0) SETATTR(1, 3, JAMUN_KSUB0) SETATTR(1, 4, JAMUN_KSUB0)
  SETATTR(1, 5, JAMUN_KSUB0) SETATTR(3, 1, JAMUN_KSUB1) SETATTR(3, 2, JAMUN_KSUB1)
#undef SETATTR
  return e == hipSuccess ? 0 : -1;
}
