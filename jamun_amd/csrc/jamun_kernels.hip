// jamun_kernels.hip — gfx950 (MI355X / CDNA4) kernels of the JAMUN walk-jump sampling path.
//
// Layout conventions (all fp32 unless noted):
//   node features x      [n_atoms][XS]  XS = mul0 + 3*mul1, e3nn layout: mul0 scalars, then mul1 vectors [u][m]
//   edge table           fixed stride S per destination atom: slot e = i*S + t, t < deg[i];
//                        radial neighbours first (ascending source index), then bonded in-edges (list order)
//                        == the reference's edge order restricted to one destination
//                        (src/jamun/model/denoiser.py:149-152)
//   esrc[e]              source atom | (bonded << 31)
//   egeo[e]              (vhat.x, vhat.y, vhat.z, d)  in c_in-scaled units (src/jamun/model/arch/e3conv.py:114-116)
//   h[e][HS]             radial-MLP hidden activations SiLU(W1 a_e + b1) [64], then 1 (bias row), then zeros
//
// The conv contraction itself lives in jamun_conv.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jamun_internal.h"
#include "jamun_split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FMUL(a, b) __fmul_rn((a), (b))
#define FADD(a, b) __fadd_rn((a), (b))
#define FSUB(a, b) __fsub_rn((a), (b))

// ------------------------------------------------------------------------------------------------
// mean_center (src/jamun/utils/mean_center.py:7-12) — one workgroup per graph, sequential sum in atom
// order (same order as index_add_ on the CPU reference).
// ------------------------------------------------------------------------------------------------
__global__ void k_mean_center(const float* __restrict__ pos, const int* __restrict__ ptr, float* __restrict__ out) {
  __shared__ float cen[3];
  const int g = blockIdx.x;
  const int lo = ptr[g], hi = ptr[g + 1];
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int a = lo; a < hi; ++a) s = FADD(s, pos[a * 3 + threadIdx.x]);
    float cnt = (float)(hi - lo);
    cen[threadIdx.x] = s / (cnt < 1.f ? 1.f : cnt);
  }
  __syncthreads();
  for (int a = lo * 3 + threadIdx.x; a < hi * 3; a += blockDim.x) out[a] = FSUB(pos[a], cen[a % 3]);
}

// ------------------------------------------------------------------------------------------------
// radius graph, stand-alone form (torch_cluster CUDA semantics; call site src/jamun/model/denoiser.py:149)
// ------------------------------------------------------------------------------------------------
__global__ void k_radius_graph(const float* __restrict__ pos, const int* __restrict__ ptr, float r2, int stride,
                               int* __restrict__ nbr, int* __restrict__ deg) {
  const int g = blockIdx.x;
  const int lo = ptr[g], hi = ptr[g + 1];
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float px = pos[i * 3], py = pos[i * 3 + 1], pz = pos[i * 3 + 2];
    int count = 0, nr = 0;
    for (int j = lo; j < hi; ++j) {
      float dx = FSUB(pos[j * 3], px), dy = FSUB(pos[j * 3 + 1], py), dz = FSUB(pos[j * 3 + 2], pz);
      float d2 = FADD(FADD(FMUL(dx, dx), FMUL(dy, dy)), FMUL(dz, dz));
      if (d2 < r2) {
        if (j != i) nbr[(size_t)i * stride + nr++] = j;
        if (++count >= JAMUN_MAX_NEIGHBORS + 1) break;
      }
    }
    deg[i] = nr;
  }
}

// ------------------------------------------------------------------------------------------------
// geometry kernel: centre y, build the per-destination edge table (radial + bonded), edge unit vectors
// and scaled lengths.  One workgroup per walker.  (src/jamun/model/denoiser.py:138-166,188-192;
// src/jamun/model/arch/e3conv.py:114-116)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_normal3(uint64_t seed, uint32_t iter, uint32_t atom, float out[3]);

// Optional BAOAB first half (B, A, O, A) fused in front of the geometry: the walker's workgroup first advances its own atoms
//   v += u(d/2) psi ; y += (d/2) v ; vhat = a v + z R ; y += (d/2) vhat ; v <- vhat      (functional/_splitting.py:158-163)
// with exactly the arithmetic of k_baoab_pre, then centres and builds the edge table from the new y.
struct GeomArgs {
  float* y;
  const int* ptr;
  float c_in, r2;
  int S;
  const int *bond_in_ptr, *bond_in_src;
  int mean_center;
  float* yc;
  int *deg, *esrc;
  float4* egeo;
  int* epair;
};
// (cen [3] and s_pos [3 GEOM_LDS_ATOMS]: the workgroup's shared arrays)
__device__ __forceinline__ void geom_body(float* __restrict__ y, const int* __restrict__ ptr, float c_in, float r2, int S,
                                          const int* __restrict__ bond_in_ptr, const int* __restrict__ bond_in_src, int mean_center,
                                          float* __restrict__ yc, int* __restrict__ deg, int* __restrict__ esrc,
                                          float4* __restrict__ egeo, int* __restrict__ epair, const LangevinPre& pre, float* cen, float* s_pos) {
  // The walker's coordinates through LDS (molecules up to GEOM_LDS_ATOMS): the centre is a sum in atom order by three threads and the
  // neighbour search reads every atom of the molecule per thread — as global loads both are chains of dependent round trips (a
  // 166-atom molecule: 50 us of a launch that computes almost nothing).  Same values, same operations, same order: bit-identical.
  const int g = blockIdx.x;
  const int lo = ptr[g], hi = ptr[g + 1];
  const bool in_lds = hi - lo <= GEOM_LDS_ATOMS;
  // (the bond list of this thread's first atom — topology, independent of the positions — is requested here, with the integrator's inputs:
  // at its point of use, behind three barriers, it was two more dependent round trips of a kernel that is a chain of them)
  const int i_first = lo + (int)threadIdx.x;
  int pf_b0 = 0, pf_b1 = 0, pf_src[4] = {-1, -1, -1, -1};
  if (i_first < hi) {
    pf_b0 = bond_in_ptr[i_first];
    pf_b1 = bond_in_ptr[i_first + 1];
#pragma unroll
    for (int q = 0; q < 4; ++q) pf_src[q] = q < pf_b1 - pf_b0 ? bond_in_src[pf_b0 + q] : -1;
  }
  if (pre.v) {
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
      // (the atom's nine inputs requested before its first store: pre.v / pre.psi may alias y for all the compiler knows, and between the stores
      // every load was a round trip of its own)
      float v0[3], p0[3], y0[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) { v0[c] = pre.v[i * 3 + c]; p0[c] = pre.psi[i * 3 + c]; y0[c] = y[i * 3 + c]; }
      float R[3];
      if (pre.noise) { R[0] = pre.noise[i * 3]; R[1] = pre.noise[i * 3 + 1]; R[2] = pre.noise[i * 3 + 2]; }
      else philox_normal3(pre.seed, pre.iter, (uint32_t)i, R);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float vv = FADD(v0[c], FMUL(pre.k.u_half_delta, p0[c]));
        float yy = FADD(y0[c], FMUL(pre.k.half_delta, vv));
        float vh = FADD(FMUL(pre.k.exp_mg, vv), FMUL(pre.k.zeta_sqrt_u, R[c]));
        yy = FADD(yy, FMUL(pre.k.half_delta, vh));
        pre.v[i * 3 + c] = vh;
        y[i * 3 + c] = yy;
        if (in_lds) s_pos[(i - lo) * 3 + c] = yy;
      }
    }
    __syncthreads();  // (this walker's y is read below by this workgroup only)
  } else if (in_lds) {
    for (int a = lo * 3 + threadIdx.x; a < hi * 3; a += blockDim.x) s_pos[a - lo * 3] = y[a];
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    float s = 0.f;
    if (mean_center) {
      if (in_lds) for (int a = 0; a < hi - lo; ++a) s = FADD(s, s_pos[a * 3 + threadIdx.x]);
      else for (int a = lo; a < hi; ++a) s = FADD(s, y[a * 3 + threadIdx.x]);
      float cnt = (float)(hi - lo);
      s = s / (cnt < 1.f ? 1.f : cnt);
    }
    cen[threadIdx.x] = s;
  }
  __syncthreads();
  if (in_lds) {
    for (int a = lo * 3 + threadIdx.x; a < hi * 3; a += blockDim.x) {
      const float v = FSUB(s_pos[a - lo * 3], cen[a % 3]);
      yc[a] = v;
      s_pos[a - lo * 3] = v;  // (from here on: the centred coordinates)
    }
  } else {
    for (int a = lo * 3 + threadIdx.x; a < hi * 3; a += blockDim.x) yc[a] = FSUB(y[a], cen[a % 3]);
  }
  __syncthreads();  // yc of this graph is only read by this workgroup (LDS, or global memory on the same CU)
  // (the per-atom body once per address space: a pointer chosen at run time between LDS and global memory would compile to flat loads)
  auto atoms = [&](auto rd) {  // rd(k): element k of the centred coordinates [atom][3]
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const float px = rd(i * 3), py = rd(i * 3 + 1), pz = rd(i * 3 + 2);
    const float sx = FMUL(px, c_in), sy = FMUL(py, c_in), sz = FMUL(pz, c_in);
    int count = 0, nr = 0;
    const size_t base = (size_t)i * S;
    // pair table (jamun_internal.h: JAMUN_EP_*): the radial slot of each bonded neighbour is noted while the radial edges are written
    // (atoms with up to four bonds; more: a scan over the slots below)
    const bool pf = i == i_first;  // (always, for molecules up to the workgroup's 128 threads)
    const int bnd0 = pf ? pf_b0 : bond_in_ptr[i], n_bnd = (pf ? pf_b1 : bond_in_ptr[i + 1]) - bnd0;
    const bool ep_fast = epair != nullptr && n_bnd <= 4;
    int bsrc[4], twin[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { bsrc[q] = (ep_fast && q < n_bnd) ? (pf ? pf_src[q] : bond_in_src[bnd0 + q]) : -1; twin[q] = -1; }
    for (int j = lo; j < hi; ++j) {
      const float qx = rd(j * 3), qy = rd(j * 3 + 1), qz = rd(j * 3 + 2);
      float dx = FSUB(qx, px), dy = FSUB(qy, py), dz = FSUB(qz, pz);
      float d2 = FADD(FADD(FMUL(dx, dx), FMUL(dy, dy)), FMUL(dz, dz));
      if (d2 < r2) {
        if (j != i) {
          // edge_vec = pos_scaled[src] - pos_scaled[dst]
          float ex = FSUB(FMUL(qx, c_in), sx), ey = FSUB(FMUL(qy, c_in), sy), ez = FSUB(FMUL(qz, c_in), sz);
          float d = sqrtf(fmaf(ez, ez, fmaf(ey, ey, FMUL(ex, ex))));  // edge_vec.norm(dim=1), CPU fma order
          float dn = d < 1e-12f ? 1e-12f : d;
          esrc[base + nr] = j;
          egeo[base + nr] = make_float4(ex / dn, ey / dn, ez / dn, d);
          if (epair) epair[base + nr] = 0;  // (owners of a pair are rewritten below)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (j == bsrc[q]) twin[q] = nr;
          ++nr;
        }
        if (++count >= JAMUN_MAX_NEIGHBORS + 1) break;
      }
    }
    const int nr_rad = nr;
    for (int b = bnd0; b < bnd0 + n_bnd; ++b) {
      int j = -1;
      if (pf && b - bnd0 < 4) {  // (a compare per element: a run-time index would put the array into scratch)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q == b - bnd0) j = pf_src[q];
      } else {
        j = bond_in_src[b];
      }
      const float qx = rd(j * 3), qy = rd(j * 3 + 1), qz = rd(j * 3 + 2);  // (bonds stay inside the walker's molecule)
      float ex = FSUB(FMUL(qx, c_in), sx), ey = FSUB(FMUL(qy, c_in), sy), ez = FSUB(FMUL(qz, c_in), sz);
      float d = sqrtf(fmaf(ez, ez, fmaf(ey, ey, FMUL(ex, ex))));
      float dn = d < 1e-12f ? 1e-12f : d;
      esrc[base + nr] = (int)((unsigned)j | 0x80000000u);
      egeo[base + nr] = make_float4(ex / dn, ey / dn, ez / dn, d);
      ++nr;
    }
    deg[i] = nr;
    // a bonded edge whose source already has a slot — its radial twin inside the cutoff, or an earlier copy of the same bond — is owned by
    // the FIRST such slot, which lists it (up to two; more: JAMUN_EP_OVERFLOW)
    if (epair) {
      int wrd[4] = {0, 0, 0, 0};  // words of the twins' slots (fast path: written once, below)
      for (int t = nr_rad; t < nr; ++t) {
        int j = -1, owner = -1;
        if (ep_fast) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q == t - nr_rad) { owner = twin[q]; j = bsrc[q]; }  // (the bond's source from the register, not read back from the slot)
          // (several bonds from one source share its twin: twin[] holds the same slot for each; no radial twin: an earlier bonded slot)
          for (int o = nr_rad; owner < 0 && o < t; ++o)
            if ((esrc[base + o] & 0x7fffffff) == j) owner = o;
        } else {
          j = esrc[base + t] & 0x7fffffff;
          for (int o = 0; owner < 0 && o < t; ++o)
            if ((esrc[base + o] & 0x7fffffff) == j) owner = o;
        }
        epair[base + t] = owner >= 0 ? (int)JAMUN_EP_OWNED : 0;
        if (owner >= 0) {
          auto add = [&](int w) { return (w & 127) == 0 ? (w | (t + 1)) : ((w >> 7) & 127) == 0 ? (w | ((t + 1) << 7)) : (w | JAMUN_EP_OVERFLOW); };
          bool kept = false;
          if (ep_fast && owner < nr_rad) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (!kept && twin[q] == owner) { wrd[q] = add(wrd[q]); kept = true; }  // (the first bond of this source keeps the word)
          }
          if (!kept) epair[base + owner] = add(epair[base + owner]);  // (bonded owners, and the scan path: written above, same thread)
        }
      }
      if (ep_fast) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bool first = twin[q] >= 0;
#pragma unroll
          for (int q2 = 0; q2 < q; ++q2)
            if (twin[q2] == twin[q]) first = false;
          if (first && wrd[q] != 0) epair[base + twin[q]] = wrd[q];
        }
      }
    }
  }
  };
  if (in_lds) atoms([&](int k) { return s_pos[k - lo * 3]; });
  else atoms([&](int k) { return yc[k]; });
}
__global__ void k_geom(float* __restrict__ y, const int* __restrict__ ptr, float c_in, float r2, int S,
                       const int* __restrict__ bond_in_ptr, const int* __restrict__ bond_in_src, int mean_center,
                       float* __restrict__ yc, int* __restrict__ deg, int* __restrict__ esrc,
                       float4* __restrict__ egeo, int* __restrict__ epair, LangevinPre pre) {
  __shared__ float cen[3];
  __shared__ float s_pos[3 * GEOM_LDS_ATOMS];
  geom_body(y, ptr, c_in, r2, S, bond_in_ptr, bond_in_src, mean_center, yc, deg, esrc, egeo, epair, pre, cen, s_pos);
}

// ------------------------------------------------------------------------------------------------
// radial-MLP hidden layer per edge: h = SiLU(W1 [E_bond[mask] | radial(d)] + b1)
// (src/jamun/model/arch/e3conv.py:118-127, src/jamun/e3tools/nn/_mlp.py:10-34).
// One wave per destination atom, lane = hidden unit k.  cmask[mask][k] = b1[k] + W1[k,:32].E_bond[mask].
// ------------------------------------------------------------------------------------------------
// On the matrix cores, transposed: D^T[k][e] = sum_r W1[k][r] rad_e[r] for a tile of 32 edge slots, so that the result
// lands as rows of hidden units over consecutive slots — the `[k][slot]` layout the conv kernels read with coalesced
// loads.  One wave = 32 consecutive slots of the fixed-stride edge table, ALL layers (the radial basis of a slot is
// computed once, in the lane that owns it, directly in B-operand layout: lane (slot, hh) holds basis 2s+hh).
typedef float eh_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_edge_h(const int* __restrict__ deg, const int* __restrict__ esrc,
                                                const float4* __restrict__ egeo, int n_atoms, int S,
                                                const float* __restrict__ w1r_all,    // [layers][32][64] radial part of W1, transposed
                                                const float* __restrict__ cmask_all,  // [layers][2][64]
                                                int n_layers,
                                                const float* __restrict__ mu,         // [32] basis centres
                                                float step, float* __restrict__ h_all, size_t h_layer_stride,
                                                size_t h_kstride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e_l = lane & 31, hh = lane >> 5;
  const long n_slots = (long)n_atoms * S;
  const long slot = ((long)blockIdx.x * 4 + wave) * 32 + e_l;
  const int i = slot < n_slots ? (int)(slot / S) : 0;
  const int t = (int)(slot - (long)i * S);
  const bool valid = slot < n_slots && t < deg[i];
  if (__ballot(valid) == 0) return;  // wave-uniform: no edge in these 32 slots
  float d = 0.f;
  int bonded = 0;
  if (valid) {
    d = egeo[slot].w;
    bonded = esrc[slot] < 0 ? 1 : 0;
  }
  float rad[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const float diff = FSUB(d, mu[2 * s + hh]) / step;
    rad[s] = expf(-FMUL(diff, diff)) / 1.12f;
  }
  // (layer, k-tile) pairs run as one flat loop; the A fragments of the next pair are fetched before the MFMAs of the current
  // blockIdx.y picks a group of layers (more waves in flight; the radial basis is recomputed per group, 16 expf)
  const int l_per = (n_layers + gridDim.y - 1) / gridDim.y, l_begin = blockIdx.y * l_per;
  const int l_end = l_begin + l_per < n_layers ? l_begin + l_per : n_layers;
  const float* __restrict__ wbase = w1r_all + hh * 64 + e_l;  // A operand: W1[k = 32 mt + (lane & 31)][r = 2s + hh]
  float av[16], an[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) av[s] = wbase[(size_t)l_begin * 64 * 32 + 2 * s * 64];
  const int n_jobs = 2 * l_end;
  for (int j = 2 * l_begin; j < n_jobs; ++j) {
    const int l = j >> 1, mt = j & 1;
    const int jn = j + 1 < n_jobs ? j + 1 : j;
    const float* __restrict__ wn = wbase + (size_t)(jn >> 1) * 64 * 32 + 32 * (jn & 1);
#pragma unroll
    for (int s = 0; s < 16; ++s) an[s] = wn[2 * s * 64];
    const float* __restrict__ c = cmask_all + (size_t)l * 128 + bonded * 64 + 4 * hh;  // c_mask[mask of this lane's slot][k]
    float* __restrict__ h = h_all + (size_t)l * h_layer_stride + slot;
    eh_f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = c[32 * mt + (q & 3) + 8 * (q >> 2)];  // accumulator row of register q: hidden unit
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], rad[s], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * hh;
      const float pre = acc[q];
      // SiLU with the hardware exponential and reciprocal (v_exp_f32 / v_rcp_f32: ~1 ulp each; expf and the IEEE division were 2/3 of
      // this kernel's vector instructions: 62 -> see DESIGN.md 3.2)
      const float hv = pre * __frcp_rn(1.f + __expf(-pre));
      if (valid) h[(size_t)k * h_kstride] = hv;
    }
    if (mt == 1 && valid && hh == 0) h[(size_t)64 * h_kstride] = 1.f;  // bias row of the second radial-MLP layer
#pragma unroll
    for (int s = 0; s < 16; ++s) av[s] = an[s];
  }
}

// The same product as f16x3 (three v_mfma_f32_32x32x16_f16 per K-step of 16 basis functions instead of eight v_mfma_f32_32x32x2_f32:
// 192 matrix cycles per (layer, k-tile) instead of 1024 — the fp32 MFMAs were most of this kernel).  The Gaussian basis values
// (<= 1 / 1.12) are scaled by 2^14 and split hi + lo in the lane that computes them; W1's radial part is scaled by 2^sW per layer and
// split on the host (w1h: [layer][k-tile][K-step][hi, lo][64 lanes] A fragments: lane (k, hh), halves p <-> basis 16 s2 + 8 hh + p);
// the bias (c_mask) joins after the accumulator is scaled back (isc[layer] = 2^-(14 + sW)).
typedef _Float16 eh_h8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_edge_h16(const int* __restrict__ deg, const int* __restrict__ esrc,
                                                  const float4* __restrict__ egeo, int n_atoms, int S,
                                                  const float4* __restrict__ w1h, const float* __restrict__ isc_all,
                                                  const float* __restrict__ cmask_all,  // [layers][2][64]
                                                  int n_layers, const float* __restrict__ mu, float step,
                                                  float* __restrict__ h_all, size_t h_layer_stride, size_t h_kstride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int e_l = lane & 31, hh = lane >> 5;
  const long n_slots = (long)n_atoms * S;
  const long slot = ((long)blockIdx.x * 4 + wave) * 32 + e_l;
  const int i = slot < n_slots ? (int)(slot / S) : 0;
  const int t = (int)(slot - (long)i * S);
  // (the slot's record requested with the in-degree, not behind it: unconditional at a clamped slot, masked by `valid` — one dependent round trip less)
  const long slot_c = slot < n_slots ? slot : n_slots - 1;
  const float d_raw = egeo[slot_c].w;
  const int src_raw = esrc[slot_c];
  const bool valid = slot < n_slots && t < deg[i];
  if (__ballot(valid) == 0) return;  // wave-uniform: no edge in these 32 slots
  const float d = valid ? d_raw : 0.f;
  const int bonded = (valid && src_raw < 0) ? 1 : 0;
  float4 Rh[2], Rl[2];  // B fragments: lane (slot, hh), halves p <-> basis 16 s2 + 8 hh + p
  const float inv_step = 1.f / step;
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    unsigned ph[4], pl[4];
#pragma unroll
    for (int p2 = 0; p2 < 4; ++p2) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        // (reciprocal step, hardware exponential, one constant for / 1.12 * 2^14: the IEEE divisions and expf were 45 % of this kernel's
        // vector instructions; the value is split into two f16 terms just below — 2^-22 of it — and these forms are within 1e-7 absolute)
        const float diff = FSUB(d, mu[16 * s2 + 8 * hh + 2 * p2 + e]) * inv_step;
        v[e] = __expf(-FMUL(diff, diff)) * (float)(16384.0 / 1.12);
      }
      ph[p2] = cvt_pk_f16(v[0], v[1]);  // (jamun_split.h: compiler-visible instructions, never inline asm in front of an MFMA)
      pl[p2] = cvt_pk_f16(resid_lo(v[0], ph[p2]), resid_hi(v[1], ph[p2]));
    }
    Rh[s2] = make_float4(__uint_as_float(ph[0]), __uint_as_float(ph[1]), __uint_as_float(ph[2]), __uint_as_float(ph[3]));
    Rl[s2] = make_float4(__uint_as_float(pl[0]), __uint_as_float(pl[1]), __uint_as_float(pl[2]), __uint_as_float(pl[3]));
  }
  const int l_per = (n_layers + gridDim.y - 1) / gridDim.y, l_begin = blockIdx.y * l_per;
  const int l_end = l_begin + l_per < n_layers ? l_begin + l_per : n_layers;
  float4 av[4], an[4];  // (s2, hi / lo)
#pragma unroll
  for (int b = 0; b < 4; ++b) av[b] = w1h[((size_t)(2 * l_begin) * 4 + b) * 64 + lane];
  const int n_jobs = 2 * l_end;
  for (int j = 2 * l_begin; j < n_jobs; ++j) {
    const int l = j >> 1, mt = j & 1;
    const int jn = j + 1 < n_jobs ? j + 1 : j;
#pragma unroll
    for (int b = 0; b < 4; ++b) an[b] = w1h[((size_t)jn * 4 + b) * 64 + lane];
    const float* __restrict__ c = cmask_all + (size_t)l * 128 + bonded * 64 + 4 * hh;  // c_mask[mask of this lane's slot][k]
    const float isc = isc_all[l];
    // (this lane's row 32 mt + 4 hh; the rows of its 16 accumulator registers then lie at wave-uniform distances: one 64-bit add per store)
    float* __restrict__ h = h_all + (size_t)l * h_layer_stride + slot + (size_t)(32 * mt + 4 * hh) * h_kstride;
    eh_f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(eh_h8, av[2 * s2 + 1]), __builtin_bit_cast(eh_h8, Rh[s2]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(eh_h8, av[2 * s2]), __builtin_bit_cast(eh_h8, Rl[s2]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(eh_h8, av[2 * s2]), __builtin_bit_cast(eh_h8, Rh[s2]), acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float pre = fmaf(acc[q], isc, c[32 * mt + (q & 3) + 8 * (q >> 2)]);
      // (v_rcp_f32, 1 ulp: __frcp_rn is the correctly rounded reciprocal — a ten-instruction IEEE division per output)
      const float hv = pre * __builtin_amdgcn_rcpf(1.f + __expf(-pre));
      if (valid) h[(size_t)((q & 3) + 8 * (q >> 2)) * h_kstride] = hv;
    }
    if (mt == 1 && valid && hh == 0) h[(size_t)32 * h_kstride] = 1.f;  // bias row (64) of the second radial-MLP layer
#pragma unroll
    for (int b = 0; b < 4; ++b) av[b] = an[b];
  }
}

// ------------------------------------------------------------------------------------------------
// node update: sum the K-slice partials (fixed order), mean over in-edges, gate, self-interaction
// Linear, skip Linear, noise-conditional skip mix.
// (src/jamun/e3tools/nn/_conv.py:117, _gate.py:53-64, _interaction.py:26-30, model/noise_conditioning.py:69-73)
// ------------------------------------------------------------------------------------------------
// One workgroup = 32 atoms, 8 waves.  Phase 1 (all threads) builds the two Linear inputs TRANSPOSED in LDS
// ([K][33]: lane = atom, conflict-free): scalars [act(m0) | x_in scalars], per vector plane [gate * m1 | x_in vectors].
// Phase 2 runs both o3.Linear layers (self-interaction + skip, concatenated along K) on the matrix cores:
// one job per wave: scalar output tile nt (32 columns) or one vector plane.  Weights are host-packed
// 16-byte fragments in MFMA operand order (4 consecutive K-steps of one lane).
typedef float nu_f32x16 __attribute__((ext_vector_type(16)));
#define NU_LD 33
#define NU_T 512  // threads per workgroup
#define NU_U 5    // 32-column tiles of the scalar rows per thread (nt0 <= 5)
#define NU_X 7    // 32-column tiles of the input features per thread (XSin <= 224)
#define NU_S 3    // partial slabs fetched at once (more are summed in a loop)
#define NU_V 6    // (atom, plane) rows per thread: 32 atoms x 3 planes / 16 rows per pass (nt1 == 1)
__global__ __launch_bounds__(NU_T) void k_node_update(NodeArgs a) {
  extern __shared__ float sm[];
  const int G0 = a.mul0 + a.mul1;
  const int K0 = a.K0p, K1 = a.K1p;                 // padded contraction depths (multiples of 8)
  float* __restrict__ A0 = sm;                      // [K0][33]
  float* __restrict__ A1 = A0 + K0 * NU_LD;         // [3][K1][33]
  float* __restrict__ s_gate = A1 + 3 * K1 * NU_LD; // [32][mul1]
  const int n0 = blockIdx.x * 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int w0 = a.nt0 * 32;  // padded row width of partial0
  const int w1 = a.nt1 * 32;
  const int ns_max = a.atom_nslab ? a.max_slabs : a.n_slices;
  __shared__ int s_ns[32];
  __shared__ float s_deg[32];
  if (tid < 32) {
    const int i = n0 + tid;
    s_ns[tid] = i < a.n_atoms ? (a.atom_nslab ? a.atom_nslab[i] : a.n_slices) : 0;
    const int d = i < a.n_atoms ? a.deg[i] : 1;
    s_deg[tid] = (float)(d < 1 ? 1 : d);
  }
  __syncthreads();
  // Thread -> element maps without integer division: a wave row of 32 lanes covers 32 consecutive columns (one 128-byte
  // segment) of one atom; 16 rows per pass.  All global loads of phase 1 are issued before the first use.
  const int col = tid & 31, row = tid >> 5;  // row < 16
  // ---- vector-row slab sums: 96 (atom, plane) rows x 32 channels (nt1 == 1)
  // The first NU_S slabs are fetched unconditionally (slab index clamped, result masked) so that all their loads are in
  // flight together; a loop over s would wait for slab s before asking for slab s+1.  Summation order stays s = 0, 1, ...
  float mv[NU_V];
  {
    float lv[NU_S][NU_V];
#pragma unroll
    for (int s = 0; s < NU_S; ++s)
#pragma unroll
      for (int u = 0; u < NU_V; ++u) {
        const int rr = row + 16 * u, il = rr / 3;  // rr = il * 3 + plane
        const int sc = s < s_ns[il] ? s : 0;
        lv[s][u] = a.partial1[((size_t)sc * a.n_pad + n0) * 3 * w1 + rr * w1 + col];
      }
#pragma unroll
    for (int u = 0; u < NU_V; ++u) {
      const int rr = row + 16 * u, il = rr / 3;
      mv[u] = 0.f;
#pragma unroll
      for (int s = 0; s < NU_S; ++s) mv[u] += s < s_ns[il] ? lv[s][u] : 0.f;
    }
  }
  for (int s = NU_S; s < ns_max; ++s) {
#pragma unroll
    for (int u = 0; u < NU_V; ++u) {
      const int rr = row + 16 * u, il = rr / 3;
      if (s < s_ns[il]) mv[u] += a.partial1[((size_t)s * a.n_pad + n0) * 3 * w1 + rr * w1 + col];
    }
  }
  // ---- scalar-row slab sums: 32 atoms x w0 columns
  float m0[2][NU_U];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int il = row + 16 * v;
    float ls[NU_S][NU_U];
#pragma unroll
    for (int s = 0; s < NU_S; ++s) {
      const int sc = s < s_ns[il] ? s : 0;
#pragma unroll
      for (int u = 0; u < NU_U; ++u) ls[s][u] = u < a.nt0 ? a.partial0[((size_t)sc * a.n_pad + n0 + il) * w0 + col + 32 * u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NU_U; ++u) {
      m0[v][u] = 0.f;
#pragma unroll
      for (int s = 0; s < NU_S; ++s) m0[v][u] += s < s_ns[il] ? ls[s][u] : 0.f;
    }
  }
  for (int s = NU_S; s < ns_max; ++s) {
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int il = row + 16 * v;
#pragma unroll
      for (int u = 0; u < NU_U; ++u) {
        const int w = col + 32 * u;
        if (u < a.nt0 && s < s_ns[il]) m0[v][u] += a.partial0[((size_t)s * a.n_pad + n0 + il) * w0 + w];
      }
    }
  }
  // ---- input features
  float xv[2][NU_X];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int il = row + 16 * v;
#pragma unroll
    for (int u = 0; u < NU_X; ++u) {
      const int c = col + 32 * u;
      xv[v][u] = (c < a.XSin && n0 + il < a.n_atoms) ? a.x_in[(size_t)(n0 + il) * a.XSin + c] : 0.f;
    }
  }
  // ---- phase 1a: mean over in-edges, activation / gate
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int il = row + 16 * v;
#pragma unroll
    for (int u = 0; u < NU_U; ++u) {
      const int w = col + 32 * u;
      if (u >= a.nt0 || w >= G0) continue;
      const float val = m0[v][u] / s_deg[il];
      if (w < a.mul0) A0[w * NU_LD + il] = a.cL * (val > 0.f ? val : 0.01f * val);
      else s_gate[il * a.mul1 + (w - a.mul0)] = a.cS / (1.f + expf(-val));
    }
    // x_in: scalars extend the scalar K range, vectors extend each plane's K range; pad rows are zero
#pragma unroll
    for (int u = 0; u < NU_X; ++u) {
      const int c = col + 32 * u;
      if (c >= a.XSin) continue;
      if (c < a.in0) A0[(a.mul0 + c) * NU_LD + il] = xv[v][u];
      else {
        const int uu = (c - a.in0) / 3, mm = (c - a.in0) - 3 * uu;
        A1[(mm * K1 + a.mul1 + uu) * NU_LD + il] = xv[v][u];
      }
    }
  }
  for (int idx = tid; idx < (K0 - a.mul0 - a.in0) * 32; idx += NU_T) A0[(a.mul0 + a.in0 + (idx >> 5)) * NU_LD + (idx & 31)] = 0.f;
  for (int idx = tid; idx < 3 * (K1 - a.mul1 - a.in1) * 32; idx += NU_T) {
    const int per = (K1 - a.mul1 - a.in1) * 32;
    const int mm = idx >= 2 * per ? 2 : (idx >= per ? 1 : 0), rem = idx - mm * per;
    A1[(mm * K1 + a.mul1 + a.in1 + (rem >> 5)) * NU_LD + (rem & 31)] = 0.f;
  }
  __syncthreads();
  // ---- phase 1b: gated vectors
#pragma unroll
  for (int u = 0; u < NU_V; ++u) {
    const int rr = row + 16 * u, il = rr / 3, mm = rr - 3 * il;
    if (col < a.mul1) A1[(mm * K1 + col) * NU_LD + il] = (mv[u] / s_deg[il]) * s_gate[il * a.mul1 + col];
  }
  __syncthreads();
  // ---- phase 2: out = [act | x_in] . [W_self ; W_skip] on the matrix cores, then the noise-conditional skip mix
  const int XSo = a.mul0 + 3 * a.mul1;
  const int nts = (a.mul0 + 31) >> 5, nsg0 = K0 >> 3, nsg1 = K1 >> 3;
#if defined(NU_EXP) && (NU_EXP & 2)
  if (a.mix) return;
#endif
  for (int job = wave; job < nts + 3; job += NU_T / 64) {
    const bool scalar = job < nts;
    const int nsg = scalar ? nsg0 : nsg1;
    const float4* __restrict__ wp = (scalar ? a.wcat0 + (size_t)job * nsg0 * 64 : a.wcat1) + lane;
    const float* __restrict__ ap = (scalar ? A0 : A1 + (job - nts) * K1 * NU_LD) + hh * NU_LD + r;
    nu_f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float4 ring[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) ring[i] = wp[(i < nsg ? i : nsg - 1) * 64];
    for (int sg = 0; sg < nsg; sg += 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (sg + i < nsg) {  // wave-uniform
          const float4 w = ring[i];
          if (sg + i + 4 < nsg) ring[i] = wp[(sg + i + 4) * 64];
          const float* __restrict__ as = ap + 8 * (sg + i) * NU_LD;
#if defined(NU_EXP) && (NU_EXP & 1)
          continue;
#endif
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[0], w.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[2 * NU_LD], w.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[4 * NU_LD], w.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as[6 * NU_LD], w.w, acc, 0, 0, 0);
        }
      }
    }
    const int col = scalar ? job * 32 + r : r;                 // output channel within the irrep block
    const bool col_ok = scalar ? col < a.mul0 : col < a.mul1;
    const int o = scalar ? col : a.mul0 + 3 * col + (job - nts);  // column of x_out
    const float mw = (a.mix && col_ok) ? a.mix[scalar ? col : a.mul0 + col] : 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int i = n0 + (q & 3) + 8 * (q >> 2) + 4 * hh;
      if (col_ok && i < a.n_atoms) {
        float v = acc[q];
        if (a.mix) v = mw * a.x_in[(size_t)i * a.XSin + o] + (1.f - mw) * v;  // hidden layers: XSin == XSo, x_in is x_old
        a.x_out[(size_t)i * XSo + o] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// output head: Linear(hidden -> gate_in) . Gate . Linear(hidden -> 1x1e) * output_gain
// (src/jamun/e3tools/nn/_mlp.py:76-81,109; src/jamun/model/arch/e3conv.py:134-135).  Only the gated
// vectors reach the 1x1e output, so the mul0 activated scalars are never formed.
// 32 lanes per atom (lane = vector channel w'), 8 atoms per workgroup.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head(HeadArgs a) {
  extern __shared__ float hs[];
  const int XS = a.mul0 + 3 * a.mul1;
  float* __restrict__ s_wg = hs;                         // [mul0][mul1]
  float* __restrict__ s_wv = s_wg + a.mul0 * a.mul1;     // [mul1][mul1]
  float* __restrict__ s_x = s_wv + a.mul1 * a.mul1;      // [8][XS]
  const int wv = threadIdx.x & 31;
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
  // weights and the 8 feature rows go through LDS once: the contraction loops below then run at LDS latency instead of
  // one dependent global load per iteration
  // (16-byte pieces, all of a thread's loads in flight before its first LDS store: as 4-byte loads in three loops the staging was a chain of
  // ~26 round trips per thread — most of this kernel's 12.7 us; mul1 and XS are multiples of 4 for every supported irreps string)
  if ((a.mul0 | a.mul1) & 3) {  // (odd channel counts: 4-byte pieces)
    for (int idx = threadIdx.x; idx < a.mul0 * a.mul1; idx += 256) s_wg[idx] = a.w_gate[idx];
    for (int idx = threadIdx.x; idx < a.mul1 * a.mul1; idx += 256) s_wv[idx] = a.w_vec[idx];
    for (int idx = threadIdx.x; idx < 8 * XS; idx += 256) {
      const size_t gidx = (size_t)blockIdx.x * 8 * XS + idx;
      s_x[idx] = gidx < (size_t)a.n_atoms * XS ? a.x[gidx] : 0.f;
    }
  } else {
    const int n_wg4 = (a.mul0 * a.mul1) >> 2, n_wv4 = (a.mul1 * a.mul1) >> 2, n_x4 = (8 * XS) >> 2;
    const size_t x4_total = ((size_t)a.n_atoms * XS) >> 2;
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(a.w_gate);
    const float4* __restrict__ v4 = reinterpret_cast<const float4*>(a.w_vec);
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(a.x) + (size_t)blockIdx.x * n_x4;
    constexpr int HQ = 8;  // pieces per thread and round (256 threads x 8 x 16 B = 32 KB per round)
    const int n_all = n_wg4 + n_wv4 + n_x4;
    for (int base = 0; base < n_all; base += 256 * HQ) {
      float4 v[HQ];
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        const int idx = base + threadIdx.x + 256 * q;
        // (ONE unconditional request per piece, from a selected address — as three loads under an if / else chain each piece was a branch and,
        // for the feature rows, a wait of its own; pieces past the end read the first weight piece and are not stored, rows past the last atom
        // are multiplied by zero)
        const int ix = idx - n_wg4 - n_wv4;
        const bool is_g = idx < n_wg4, is_v = idx < n_wg4 + n_wv4;
        const bool x_ok = idx < n_all && (size_t)blockIdx.x * n_x4 + (size_t)(ix < 0 ? 0 : ix) < x4_total;
        const float4* __restrict__ src = is_g ? g4 + idx : (is_v ? v4 + (idx - n_wg4) : (x_ok ? x4 + ix : g4));
        const float km = (is_v || x_ok) ? 1.f : 0.f;
        const float4 t = *src;
        v[q] = make_float4(t.x * km, t.y * km, t.z * km, t.w * km);
      }
#pragma unroll
      for (int q = 0; q < HQ; ++q) {
        const int idx = base + threadIdx.x + 256 * q;
        if (idx < n_all) reinterpret_cast<float4*>(hs)[idx] = v[q];  // (s_wg | s_wv | s_x are contiguous)
      }
    }
  }
  __syncthreads();
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (i < a.n_atoms) {
    const float* __restrict__ xi = s_x + (threadIdx.x >> 5) * XS;
    for (int w0 = wv; w0 < a.mul1; w0 += 32) {
      float gp = 0.f;
      for (int u = 0; u < a.mul0; ++u) gp = fmaf(s_wg[u * a.mul1 + w0], xi[u], gp);
      float hx = 0.f, hy = 0.f, hz = 0.f;
      for (int u = 0; u < a.mul1; ++u) {
        const float ww = s_wv[u * a.mul1 + w0];
        hx = fmaf(ww, xi[a.mul0 + u * 3 + 0], hx);
        hy = fmaf(ww, xi[a.mul0 + u * 3 + 1], hy);
        hz = fmaf(ww, xi[a.mul0 + u * 3 + 2], hz);
      }
      const float gate = a.cS / (1.f + expf(-gp));
      const float wo = a.w_out[w0];
      gx = fmaf(wo, hx * gate, gx);
      gy = fmaf(wo, hy * gate, gy);
      gz = fmaf(wo, hz * gate, gz);
    }
  }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {
    gx += __shfl_xor(gx, off, 32);
    gy += __shfl_xor(gy, off, 32);
    gz += __shfl_xor(gz, off, 32);
  }
  if (wv == 0 && i < a.n_atoms) {
    a.g[i * 3 + 0] = gx;
    a.g[i * 3 + 1] = gy;
    a.g[i * 3 + 2] = gz;
  }
}

// ------------------------------------------------------------------------------------------------
// finalize: xhat = mean_center(c_skip * yc + c_out * g) ; score = (xhat - y) / sigma^2
// (src/jamun/model/denoiser.py:200,213-215,111-114).  One workgroup per walker.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void process_score(const float s[3], const LangevinConsts& k, float psi[3]);

// Optional BAOAB second half fused behind it:  psi = clip(score) * beta ; v = vhat + (d/2) psi (no u, _splitting.py:166) ; the
// saved frame (y, score, xhat) — exactly the arithmetic of k_baoab_post.
__device__ __forceinline__ void finalize_body(const float* __restrict__ y, const float* __restrict__ yc, const float* __restrict__ g,
                                              const int* __restrict__ ptr, float c_skip, float c_out, float sigma2, int mean_center,
                                              float* __restrict__ tmp, float* __restrict__ xhat, float* __restrict__ score, const LangevinPost& post,
                                              float* cen, float* s_tmp) {  // (s_tmp [3 GEOM_LDS_ATOMS], as k_geom: the centre is a sum in atom order by three threads)
  const int gi = blockIdx.x;
  const int lo = ptr[gi], hi = ptr[gi + 1];
  const bool in_lds = hi - lo <= GEOM_LDS_ATOMS;
  for (int a = lo * 3 + threadIdx.x; a < hi * 3; a += blockDim.x) {
    const float v = FADD(FMUL(c_skip, yc[a]), FMUL(c_out, g[a]));
    if (in_lds) s_tmp[a - lo * 3] = v;
    else tmp[a] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    if (mean_center) {
      if (in_lds) for (int a = 0; a < hi - lo; ++a) s = FADD(s, s_tmp[a * 3 + threadIdx.x]);
      else for (int a = lo; a < hi; ++a) s = FADD(s, tmp[a * 3 + threadIdx.x]);
      float cnt = (float)(hi - lo);
      s = s / (cnt < 1.f ? 1.f : cnt);
    }
    cen[threadIdx.x] = s;
  }
  __syncthreads();
  auto atoms = [&](auto rd) {  // (once per address space, as k_geom)
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    // (everything the atom reads is requested before its first store: between the stores — the output pointers of LangevinPost may alias for all
    // the compiler knows — every load was a round trip of its own, seven per atom in a kernel that is nothing but latency)
    float yv[3], vv[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) yv[c] = y[i * 3 + c];
    if (post.psi_out && post.update_v) {
#pragma unroll
      for (int c = 0; c < 3; ++c) vv[c] = post.v[i * 3 + c];
    }
    float xh[3], sc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xh[c] = FSUB(rd(i * 3 + c), cen[c]);
      sc[c] = FSUB(xh[c], yv[c]) / sigma2;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (xhat) xhat[i * 3 + c] = xh[c];
      if (score) score[i * 3 + c] = sc[c];
    }
    if (post.psi_out) {
      float p[3];
      process_score(sc, post.k, p);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        post.psi_out[i * 3 + c] = p[c];
        if (post.update_v) post.v[i * 3 + c] = FADD(vv[c], FMUL(post.k.half_delta, p[c]));
        if (post.y_frame) post.y_frame[i * 3 + c] = yv[c];
        if (post.score_frame) post.score_frame[i * 3 + c] = sc[c];
        if (post.xhat_frame) post.xhat_frame[i * 3 + c] = xh[c];
      }
    }
  }
  };
  if (in_lds) atoms([&](int k) { return s_tmp[k - lo * 3]; });
  else atoms([&](int k) { return tmp[k]; });
}
__global__ void k_finalize(const float* __restrict__ y, const float* __restrict__ yc, const float* __restrict__ g,
                           const int* __restrict__ ptr, float c_skip, float c_out, float sigma2, int mean_center,
                           float* __restrict__ tmp, float* __restrict__ xhat, float* __restrict__ score, LangevinPost post) {
  __shared__ float cen[3];
  __shared__ float s_tmp[3 * GEOM_LDS_ATOMS];
  finalize_body(y, yc, g, ptr, c_skip, c_out, sigma2, mean_center, tmp, xhat, score, post, cen, s_tmp);
}
// The LAST kernel of walk iteration i and the FIRST of iteration i + 1 in one launch (round 6): both are one workgroup per walker, and
// everything the geometry of the next iteration reads — v, psi, y of this walker — was written by this workgroup a moment ago (by the same
// threads: both loops deal the walker's atoms to the threads the same way; the barrier and fence below make that independent of the layout).
// Same arithmetic in the same order as k_finalize followed by k_geom: bit-identical trajectories, one launch boundary less per step.
__global__ void k_finalize_geom(const float* __restrict__ yc_in, const float* __restrict__ g, float c_skip, float c_out, float sigma2,
                                float* __restrict__ tmp, float* __restrict__ xhat, float* __restrict__ score, LangevinPost post, GeomArgs ga,
                                LangevinPre pre) {
  __shared__ float cen[3];
  __shared__ float s_buf[3 * GEOM_LDS_ATOMS];
  finalize_body(ga.y, yc_in, g, ga.ptr, c_skip, c_out, sigma2, ga.mean_center, tmp, xhat, score, post, cen, s_buf);
  __threadfence_block();
  __syncthreads();  // every read of y / yc / cen / s_buf of the finalize half is done; its v and psi are visible to the workgroup
  geom_body(ga.y, ga.ptr, ga.c_in, ga.r2, ga.S, ga.bond_in_ptr, ga.bond_in_src, ga.mean_center, ga.yc, ga.deg, ga.esrc, ga.egeo, ga.epair, pre, cen, s_buf);
}

// ------------------------------------------------------------------------------------------------
// Langevin updates (src/jamun/sampling/mcmc/functional/_splitting.py:26-41,80-101,148-170).
// Arithmetic order and rounding points follow the reference line by line (python-float constants are
// rounded to fp32 once, every tensor op rounds separately: no FMA contraction).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mulhilo(uint32_t a, uint32_t b, uint32_t* hi) {
  uint64_t p = (uint64_t)a * b;
  *hi = (uint32_t)(p >> 32);
  return (uint32_t)p;
}
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, hi1;
    uint32_t lo0 = mulhilo(0xD2511F53u, c[0], &hi0);
    uint32_t lo1 = mulhilo(0xCD9E8D57u, c[2], &hi1);
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ void philox_normal3(uint64_t seed, uint32_t iter, uint32_t atom, float out[3]) {
  uint32_t c[4] = {atom, iter, 0x4a414d55u, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const float inv = 2.3283064365386963e-10f;  // 2^-32
  float u0 = ((float)c[0] + 0.5f) * inv, u1 = ((float)c[1] + 0.5f) * inv;
  float u2 = ((float)c[2] + 0.5f) * inv, u3 = ((float)c[3] + 0.5f) * inv;
  u0 = fminf(fmaxf(u0, 1e-10f), 1.f);
  u2 = fminf(fmaxf(u2, 1e-10f), 1.f);
  float r0 = sqrtf(-2.f * logf(u0)), r1 = sqrtf(-2.f * logf(u2));
  out[0] = r0 * cosf(6.283185307179586f * u1);
  out[1] = r0 * sinf(6.283185307179586f * u1);
  out[2] = r1 * cosf(6.283185307179586f * u3);
}

// psi = clip(score) * beta  (create_score_fn, _splitting.py:26-41).  A zero-norm row gives NaN, as the reference.
__device__ __forceinline__ void process_score(const float s[3], const LangevinConsts& k, float psi[3]) {
  float p0 = s[0], p1 = s[1], p2 = s[2];
  if (k.has_clip) {
    // torch.linalg.vector_norm on CPU accumulates with fused multiply-adds: fma(z,z,fma(y,y,x*x))
    float norm = sqrtf(fmaf(p2, p2, fmaf(p1, p1, FMUL(p0, p0))));
    float clip = fminf(norm, k.clip);
    p0 = FMUL(p0 / norm, clip);
    p1 = FMUL(p1 / norm, clip);
    p2 = FMUL(p2 / norm, clip);
  }
  psi[0] = FMUL(p0, k.beta);
  psi[1] = FMUL(p1, k.beta);
  psi[2] = FMUL(p2, k.beta);
}

// BAOAB first half (B, A, O, A):  v += u(d/2) psi ; y += (d/2) v ; vhat = a v + z R ; y += (d/2) vhat.  v <- vhat.
__global__ void k_baoab_pre(float* __restrict__ y, float* __restrict__ v, const float* __restrict__ psi,
                            const float* __restrict__ noise, uint64_t seed, uint32_t iter, int n, LangevinConsts k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float R[3];
  if (noise) { R[0] = noise[i * 3]; R[1] = noise[i * 3 + 1]; R[2] = noise[i * 3 + 2]; }
  else philox_normal3(seed, iter, (uint32_t)i, R);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float vv = FADD(v[i * 3 + c], FMUL(k.u_half_delta, psi[i * 3 + c]));
    float yy = FADD(y[i * 3 + c], FMUL(k.half_delta, vv));
    float vh = FADD(FMUL(k.exp_mg, vv), FMUL(k.zeta_sqrt_u, R[c]));
    yy = FADD(yy, FMUL(k.half_delta, vh));
    v[i * 3 + c] = vh;
    y[i * 3 + c] = yy;
  }
}

// BAOAB second half-kick + trajectory save:  psi = process(score) ; v = vhat + (d/2) psi  (no u, _splitting.py:166)
__global__ void k_baoab_post(float* __restrict__ v, float* __restrict__ psi_out, const float* __restrict__ score,
                             const float* __restrict__ y, const float* __restrict__ xhat, int n, LangevinConsts k,
                             int update_v, float* __restrict__ y_frame, float* __restrict__ score_frame,
                             float* __restrict__ xhat_frame) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s[3] = {score[i * 3], score[i * 3 + 1], score[i * 3 + 2]};
  float p[3];
  process_score(s, k, p);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    psi_out[i * 3 + c] = p[c];
    if (update_v) v[i * 3 + c] = FADD(v[i * 3 + c], FMUL(k.half_delta, p[c]));
    if (y_frame) y_frame[i * 3 + c] = y[i * 3 + c];
    if (score_frame) score_frame[i * 3 + c] = s[c];
    if (xhat_frame) xhat_frame[i * 3 + c] = xhat[i * 3 + c];
  }
}

// The same update for FOUR consecutive atoms per thread: 12 floats = three 16-byte accesses per array instead of twelve 4-byte
// accesses at a 12-byte lane stride (the per-atom kernel reaches half of the HBM rate; arithmetic per element is identical).
__global__ __launch_bounds__(256) void k_baoab_post4(float* __restrict__ v, float* __restrict__ psi_out, const float* __restrict__ score,
                                                     const float* __restrict__ y, const float* __restrict__ xhat, int n4, LangevinConsts k,
                                                     int update_v, float* __restrict__ y_frame, float* __restrict__ score_frame,
                                                     float* __restrict__ xhat_frame) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n4) return;
  auto ld = [&](const float* p, float (&o)[12]) {
    const float4* q = reinterpret_cast<const float4*>(p) + (size_t)t * 3;
    const float4 a = q[0], b = q[1], c = q[2];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w; o[8] = c.x; o[9] = c.y; o[10] = c.z; o[11] = c.w;
  };
  auto st = [&](float* p, const float (&o)[12]) {
    float4* q = reinterpret_cast<float4*>(p) + (size_t)t * 3;
    q[0] = make_float4(o[0], o[1], o[2], o[3]); q[1] = make_float4(o[4], o[5], o[6], o[7]); q[2] = make_float4(o[8], o[9], o[10], o[11]);
  };
  float s[12], p[12], vv[12];
  ld(score, s);
  if (update_v) ld(v, vv);
#pragma unroll
  for (int a = 0; a < 4; ++a) process_score(s + 3 * a, k, p + 3 * a);
  st(psi_out, p);
  if (update_v) {
#pragma unroll
    for (int c = 0; c < 12; ++c) vv[c] = FADD(vv[c], FMUL(k.half_delta, p[c]));
    st(v, vv);
  }
  if (score_frame) st(score_frame, s);
  if (y_frame) { float tt[12]; ld(y, tt); st(y_frame, tt); }
  if (xhat_frame) { float tt[12]; ld(xhat, tt); st(xhat_frame, tt); }
}

// ABOBA first A:  y += (d/2) v
__global__ void k_aboba_a(float* __restrict__ y, const float* __restrict__ v, int n3, float half_delta) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n3) y[i] = FADD(y[i], FMUL(half_delta, v[i]));
}

// ABOBA  B O B A:  psi = process(score) ; v += u(d/2) psi ; vhat = a v + z R ; v = vhat + (d/2) psi ; y += (d/2) v
__global__ void k_aboba_b(float* __restrict__ y, float* __restrict__ v, const float* __restrict__ score,
                          const float* __restrict__ noise, uint64_t seed, uint32_t iter, int n, LangevinConsts k,
                          float* __restrict__ y_frame, float* __restrict__ score_frame) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s[3] = {score[i * 3], score[i * 3 + 1], score[i * 3 + 2]};
  float p[3];
  process_score(s, k, p);
  float R[3];
  if (noise) { R[0] = noise[i * 3]; R[1] = noise[i * 3 + 1]; R[2] = noise[i * 3 + 2]; }
  else philox_normal3(seed, iter, (uint32_t)i, R);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float vv = FADD(v[i * 3 + c], FMUL(k.u_half_delta, p[c]));
    float vh = FADD(FMUL(k.exp_mg, vv), FMUL(k.zeta_sqrt_u, R[c]));
    vv = FADD(vh, FMUL(k.half_delta, p[c]));
    float yy = FADD(y[i * 3 + c], FMUL(k.half_delta, vv));
    v[i * 3 + c] = vv;
    y[i * 3 + c] = yy;
    if (y_frame) y_frame[i * 3 + c] = yy;
    if (score_frame) score_frame[i * 3 + c] = s[c];
  }
}

__global__ void k_copy(const float* __restrict__ src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ void k_deg_to_float(const int* __restrict__ deg, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)deg[i];
}

__global__ void k_count_edges(const int* __restrict__ deg, int n, unsigned long long* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v = (i < n) ? (unsigned long long)deg[i] : 0ull;
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

// ------------------------------------------------------------------------------------------------
// scatter-mean over destination-sorted rows (torch_scatter "mean"; call site _conv.py:117).
// One wave per destination; lanes stride the feature width with 16-byte loads when width % 4 == 0;
// rows are added in row order (bitwise reproducible, no atomics).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter_mean(const float* __restrict__ src, const int* __restrict__ seg_ptr,
                                                      int n_out, int width, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= n_out) return;
  const int lo = seg_ptr[d], hi = seg_ptr[d + 1];
  const float inv_cnt_den = (float)((hi - lo) < 1 ? 1 : (hi - lo));
  if ((width & 3) == 0) {
    const int w4 = width >> 2;
    for (int c = lane; c < w4; c += 64) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* p = reinterpret_cast<const float4*>(src) + (size_t)lo * w4 + c;
      int e = lo;
      for (; e + 4 <= hi; e += 4) {  // four independent 16-B loads in flight, added in row order
        float4 a0 = p[0], a1 = p[w4], a2 = p[2 * (size_t)w4], a3 = p[3 * (size_t)w4];
        p += 4 * (size_t)w4;
        acc.x = FADD(FADD(FADD(FADD(acc.x, a0.x), a1.x), a2.x), a3.x);
        acc.y = FADD(FADD(FADD(FADD(acc.y, a0.y), a1.y), a2.y), a3.y);
        acc.z = FADD(FADD(FADD(FADD(acc.z, a0.z), a1.z), a2.z), a3.z);
        acc.w = FADD(FADD(FADD(FADD(acc.w, a0.w), a1.w), a2.w), a3.w);
      }
      for (; e < hi; ++e) {
        float4 a0 = p[0];
        p += w4;
        acc.x = FADD(acc.x, a0.x); acc.y = FADD(acc.y, a0.y); acc.z = FADD(acc.z, a0.z); acc.w = FADD(acc.w, a0.w);
      }
      acc.x /= inv_cnt_den; acc.y /= inv_cnt_den; acc.z /= inv_cnt_den; acc.w /= inv_cnt_den;
      reinterpret_cast<float4*>(out)[(size_t)d * w4 + c] = acc;
    }
  } else {
    for (int c = lane; c < width; c += 64) {
      float acc = 0.f;
      for (int e = lo; e < hi; ++e) acc = FADD(acc, src[(size_t)e * width + c]);
      out[(size_t)d * width + c] = acc / inv_cnt_den;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launchers (called from jamun_api.cpp)
// ------------------------------------------------------------------------------------------------
void launch_mean_center(const float* pos, const int* ptr, int n_graphs, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_mean_center, dim3(n_graphs), dim3(128), 0, st, pos, ptr, out);
}
void launch_radius_graph(const float* pos, const int* ptr, int n_graphs, float r2, int stride, int* nbr, int* deg,
                         hipStream_t st) {
  hipLaunchKernelGGL(k_radius_graph, dim3(n_graphs), dim3(128), 0, st, pos, ptr, r2, stride, nbr, deg);
}
void launch_geom(float* y, const int* ptr, int n_graphs, float c_in, float r2, int S, const int* bip,
                 const int* bis, int mean_center, float* yc, int* deg, int* esrc, float4* egeo, int* epair, const LangevinPre& pre,
                 hipStream_t st) {
  hipLaunchKernelGGL(k_geom, dim3(n_graphs), dim3(128), 0, st, y, ptr, c_in, r2, S, bip, bis, mean_center, yc, deg,
                     esrc, egeo, epair, pre);
}
void launch_edge_h(const int* deg, const int* esrc, const float4* egeo, int n_atoms, int S, const float* w1r_all,
                   const float* cmask_all, int n_layers, const float* mu, float step, float* h_all, size_t h_layer_stride,
                   size_t h_kstride, hipStream_t st, const float4* w1h, const float* isc_all) {
  const long tiles = ((long)n_atoms * S + 31) / 32;
  const int groups = n_layers >= 6 ? 3 : (n_layers >= 2 ? 2 : 1);
  if (w1h) {
    hipLaunchKernelGGL(k_edge_h16, dim3((unsigned)((tiles + 3) / 4), groups), dim3(256), 0, st, deg, esrc, egeo, n_atoms, S, w1h, isc_all,
                       cmask_all, n_layers, mu, step, h_all, h_layer_stride, h_kstride);
    return;
  }
  hipLaunchKernelGGL(k_edge_h, dim3((unsigned)((tiles + 3) / 4), groups), dim3(256), 0, st, deg, esrc, egeo, n_atoms, S, w1r_all,
                     cmask_all, n_layers, mu, step, h_all, h_layer_stride, h_kstride);
}

size_t node_update_lds_bytes(const NodeArgs& a) {
  return sizeof(float) * ((size_t)a.K0p * NU_LD + 3 * (size_t)a.K1p * NU_LD + 32 * (size_t)a.mul1);
}
void launch_node_update(const NodeArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_node_update, dim3(a.n_pad / 32), dim3(NU_T), node_update_lds_bytes(a), st, a);
}
int node_update_set_max_lds() {
  return hipFuncSetAttribute((const void*)k_node_update, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) ==
                 hipSuccess ? 0 : -1;
}
void launch_head(const HeadArgs& a, hipStream_t st) {
  const size_t sm = sizeof(float) * ((size_t)a.mul0 * a.mul1 + (size_t)a.mul1 * a.mul1 + 8 * (size_t)(a.mul0 + 3 * a.mul1));
  hipLaunchKernelGGL(k_head, dim3((a.n_atoms + 7) / 8), dim3(256), sm, st, a);
}
void launch_finalize(const float* y, const float* yc, const float* g, const int* ptr, int n_graphs, float c_skip,
                     float c_out, float sigma2, int mean_center, float* tmp, float* xhat, float* score,
                     const LangevinPost& post, hipStream_t st) {
  hipLaunchKernelGGL(k_finalize, dim3(n_graphs), dim3(128), 0, st, y, yc, g, ptr, c_skip, c_out, sigma2, mean_center,
                     tmp, xhat, score, post);
}
void launch_finalize_geom(float* y, float* yc, const float* g, const int* ptr, int n_graphs, float c_skip, float c_out, float sigma2, int mean_center,
                          float* tmp, float* xhat, float* score, const LangevinPost& post, float c_in, float r2, int S, const int* bip, const int* bis,
                          int* deg, int* esrc, float4* egeo, int* epair, const LangevinPre& pre, hipStream_t st) {
  GeomArgs ga{y, ptr, c_in, r2, S, bip, bis, mean_center, yc, deg, esrc, egeo, epair};
  hipLaunchKernelGGL(k_finalize_geom, dim3(n_graphs), dim3(128), 0, st, yc, g, c_skip, c_out, sigma2, tmp, xhat, score, post, ga, pre);
}
void launch_baoab_pre(float* y, float* v, const float* psi, const float* noise, uint64_t seed, uint32_t iter, int n,
                      const LangevinConsts& k, hipStream_t st) {
  hipLaunchKernelGGL(k_baoab_pre, dim3((n + 255) / 256), dim3(256), 0, st, y, v, psi, noise, seed, iter, n, k);
}
void launch_baoab_post(float* v, float* psi, const float* score, const float* y, const float* xhat, int n,
                       const LangevinConsts& k, int update_v, float* yf, float* sf, float* xf, hipStream_t st) {
  auto al16 = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15u) == 0; };
  const int n4 = (n / 4);
  const bool vec = n4 > 0 && al16(v) && al16(psi) && al16(score) && al16(y) && al16(xhat) && al16(yf) && al16(sf) && al16(xf);
  if (vec) hipLaunchKernelGGL(k_baoab_post4, dim3((n4 + 255) / 256), dim3(256), 0, st, v, psi, score, y, xhat, n4, k, update_v, yf, sf, xf);
  const int done = vec ? 4 * n4 : 0, rest = n - done;  // the last n % 4 atoms (or everything, for unaligned buffers): one atom per thread
  if (rest > 0) {
    auto off = [&](auto* p) { return p ? p + (size_t)3 * done : p; };
    hipLaunchKernelGGL(k_baoab_post, dim3((rest + 255) / 256), dim3(256), 0, st, off(v), off(psi), off(score), off(y), off(xhat), rest, k, update_v,
                       off(yf), off(sf), off(xf));
  }
}
void launch_aboba_a(float* y, const float* v, int n, float half_delta, hipStream_t st) {
  hipLaunchKernelGGL(k_aboba_a, dim3((n * 3 + 255) / 256), dim3(256), 0, st, y, v, n * 3, half_delta);
}
void launch_aboba_b(float* y, float* v, const float* score, const float* noise, uint64_t seed, uint32_t iter, int n,
                    const LangevinConsts& k, float* yf, float* sf, hipStream_t st) {
  hipLaunchKernelGGL(k_aboba_b, dim3((n + 255) / 256), dim3(256), 0, st, y, v, score, noise, seed, iter, n, k, yf, sf);
}
void launch_copy(const float* src, float* dst, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, st, src, dst, n);
}
void launch_deg_to_float(const int* deg, float* out, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_deg_to_float, dim3((n + 255) / 256), dim3(256), 0, st, deg, out, n);
}
void launch_count_edges(const int* deg, int n, unsigned long long* out, hipStream_t st) {
  hipLaunchKernelGGL(k_count_edges, dim3((n + 255) / 256), dim3(256), 0, st, deg, n, out);
}
void launch_scatter_mean(const float* src, const int* seg_ptr, int n_out, int width, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_scatter_mean, dim3((n_out + 3) / 4), dim3(256), 0, st, src, seg_ptr, n_out, width, out);
}

// ------------------------------------------------------------------------------------------------------------------------
// Stand-alone operators of the path for foreign callers (include/jamun_hip.h): edge geometry and o3.Linear on node features.
// Inside the sampler the same arithmetic lives in k_geom / k_edge_h and in k_node_update / k_head.

// jamun_edge_geometry: per edge (src -> dst) of an edge list, from already scaled positions (src/jamun/model/arch/e3conv.py:114-123):
//   sh = [1, sqrt(3) v / max(|v|, 1e-12)]   (o3.SphericalHarmonics "1x0e+1x1e", normalize=True, component normalisation)
//   radial[k] = exp(-((|v| - mu_k) / step)^2) / 1.12,  mu = linspace(0, cutoff, n_basis + 2)[1:-1]   (soft_one_hot_linspace, gaussian)
// One thread per edge; the basis row is written by the same thread (n_basis floats: 128 B for the default 32).
__global__ __launch_bounds__(256) void k_edge_geometry(const float* __restrict__ pos, const long long* __restrict__ src, const long long* __restrict__ dst,
                                                       int n_edges, int n_atoms, float cutoff, int n_basis, float* __restrict__ sh, float* __restrict__ radial) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_edges) return;
  const long long j = src[e], i = dst[e];
  if (j < 0 || j >= n_atoms || i < 0 || i >= n_atoms) {  // an index outside the position array: nothing is read, the edge's outputs are NaN
    const float q = __int_as_float(0x7fc00000);
    reinterpret_cast<float4*>(sh)[e] = make_float4(q, q, q, q);
    for (int k = 0; k < n_basis; ++k) radial[(size_t)e * n_basis + k] = q;
    return;
  }
  const float vx = pos[3 * j] - pos[3 * i], vy = pos[3 * j + 1] - pos[3 * i + 1], vz = pos[3 * j + 2] - pos[3 * i + 2];
  const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz)));
  const float inv = 1.f / fmaxf(d, 1e-12f), s3 = 1.7320508075688772f;
  float4 o = make_float4(1.f, s3 * (vx * inv), s3 * (vy * inv), s3 * (vz * inv));
  reinterpret_cast<float4*>(sh)[e] = o;
  const float step = cutoff / (float)(n_basis + 1);
  for (int k = 0; k < n_basis; ++k) {
    const int i1 = k + 1;  // torch.linspace: lower half counted from the start, upper half from the end
    const float mu = i1 < (n_basis + 2) / 2 ? __fmul_rn(step, (float)i1) : __fsub_rn(cutoff, __fmul_rn(step, (float)(n_basis + 1 - i1)));
    const float t = __fdiv_rn(__fsub_rn(d, mu), step);
    radial[(size_t)e * n_basis + k] = expf(-(t * t)) / 1.12f;
  }
}
void launch_edge_geometry(const float* pos, const long long* src, const long long* dst, int n_edges, int n_atoms, float cutoff, int n_basis, float* sh,
                          float* radial, hipStream_t st) {
  hipLaunchKernelGGL(k_edge_geometry, dim3((n_edges + 255) / 256), dim3(256), 0, st, pos, src, dst, n_edges, n_atoms, cutoff, n_basis, sh, radial);
}

// jamun_philox_normal: the standard-normal draws the walks use when no noise tensor is supplied (R of iteration `iter` for atoms
// atom0 .. atom0 + n - 1), written out as [n, 3] — the counterpart of the reference's torch.randn_like (functional/_splitting.py:161)
__global__ __launch_bounds__(256) void k_philox_normal(float* __restrict__ out, int n, uint64_t seed, uint32_t iter, uint32_t atom0) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float R[3];
  philox_normal3(seed, iter, atom0 + (uint32_t)i, R);
  out[(size_t)i * 3] = R[0]; out[(size_t)i * 3 + 1] = R[1]; out[(size_t)i * 3 + 2] = R[2];
}
void launch_philox_normal(float* out, int n, uint64_t seed, uint32_t iter, uint32_t atom0, hipStream_t st) {
  hipLaunchKernelGGL(k_philox_normal, dim3((n + 255) / 256), dim3(256), 0, st, out, n, seed, iter, atom0);
}

// jamun_node_linear: e3nn o3.Linear between irreps (in0 x0e + in1 x1e) and (out0 x0e + out1 x1e) with the flat e3nn weight
// [in0 x out0 | in1 x out1], path normalisation 1 / sqrt(fan_in) (src/jamun/e3tools/nn/_interaction.py:23-24, _mlp.py:69,109).
// One workgroup = 8 atoms; the feature rows are staged in LDS, thread t owns output element t of every atom of the group.
__global__ __launch_bounds__(256) void k_node_linear(const float* __restrict__ x, int n_atoms, int in0, int in1, int out0, int out1,
                                                     const float* __restrict__ w, float* __restrict__ out) {
  extern __shared__ float xs_nl[];  // [8][in0 + 3 in1]
  const int win = in0 + 3 * in1, wout = out0 + 3 * out1, a0 = blockIdx.x * 8;
  for (int idx = threadIdx.x; idx < 8 * win; idx += 256) {
    const int a = a0 + idx / win;
    xs_nl[idx] = a < n_atoms ? x[(size_t)a * win + idx % win] : 0.f;
  }
  __syncthreads();
  const float n0 = in0 > 0 ? 1.f / sqrtf((float)in0) : 0.f, n1 = in1 > 0 ? 1.f / sqrtf((float)in1) : 0.f;
  const float* __restrict__ w1 = w + (size_t)in0 * out0;
  for (int o = threadIdx.x; o < wout; o += 256) {
    float acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) acc[a] = 0.f;
    if (o < out0) {
      for (int u = 0; u < in0; ++u) {
        const float wv = w[(size_t)u * out0 + o];
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = fmaf(wv, xs_nl[a * win + u], acc[a]);
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) acc[a] *= n0;
    } else {
      const int wv_i = (o - out0) / 3, m = (o - out0) % 3;
      for (int u = 0; u < in1; ++u) {
        const float wv = w1[(size_t)u * out1 + wv_i];
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = fmaf(wv, xs_nl[a * win + in0 + 3 * u + m], acc[a]);
      }
#pragma unroll
      for (int a = 0; a < 8; ++a) acc[a] *= n1;
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
      if (a0 + a < n_atoms) out[(size_t)(a0 + a) * wout + o] = acc[a];
  }
}
int launch_node_linear(const float* x, int n_atoms, int in0, int in1, int out0, int out1, const float* w, float* out, hipStream_t st) {
  const size_t smem = (size_t)8 * (in0 + 3 * in1) * sizeof(float);
  if (smem > 60 * 1024) return -1;
  hipLaunchKernelGGL(k_node_linear, dim3((n_atoms + 7) / 8), dim3(256), smem, st, x, n_atoms, in0, in1, out0, out1, w, out);
  return 0;
}
