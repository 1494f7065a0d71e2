// jamun_conv_fused.hip — fused conv contraction for small molecules: the A operand is formed ON THE MATRIX CORES.
//
// Same mathematics as jamun_conv.hip (destination-grouped association of src/jamun/e3tools/nn/_conv.py:93-119):
//     m[(i,c)][w] = sum_k sum_u ( sum_{e->i} h~_e[k] zeta_e[c][u] ) W~[(k,u)][w]
// but the inner edge sum is written as a dense product over the source atoms j the tile can see (whole molecules,
// J <= 64 rows):
//     D'_k[u][i] = sum_j X[j][u] * C_k[j][i]          C_k[j][i] = sum over the (<= 2) edges j->i of h~_e[k] * f_e
// with f = 1 or a component of the edge unit vector.  D' is a 32x32 MFMA accumulator whose COLUMN is the destination
// atom, i.e. exactly the lane layout of the A operand of the main product  out[i][w] += sum_u D'[u][i] W[(k,u)][w]:
// register q of the accumulator is K-step q of the main MFMA (rows u = (q&3) + 8(q>>2) + 4(lane>>5)), so the formed
// tile never leaves the register file — no LDS transpose, no VALU forming loop, no per-edge broadcast traffic (the
// LDS-broadcast-bound part of jamun_conv.hip, profiles/r1b).  The weights are packed in that K order.
//
// One workgroup = 8 waves = one tile of 32 destination atoms x one K-slice of hidden units k, for BOTH output kinds
// (scalar rows and the three vector planes).  Per k: all threads build the four coefficient tiles C^h, C^{h vx},
// C^{h vy}, C^{h vz} in LDS (a gather through a (source, destination) -> edge-slot map, no read-modify-write), one
// barrier (tiles are double buffered), then every wave runs its statically assigned units.  A unit = (u-tile of 32
// input channels, up to 3 "terms" of (feature column set, coefficient tile, sign)) -> forming MFMAs over J/2 K-steps
// per term -> 16 main K-steps into the scalar-row accumulators (NT0 tiles) or one vector-plane accumulator.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jamun_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FUSED_WAVES 8

template <int NT>
__device__ __forceinline__ void unit_main(f32x16 (&acc)[NT], const f32x16& af, const float4* __restrict__ wp, int lane) {
  // wp: [qg 0..3][nt][64 lanes] float4 = weights of main K-steps 4qg..4qg+3
  float4 bn[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bn[nt] = wp[nt * 64 + lane];
#pragma unroll
  for (int qg = 0; qg < 4; ++qg) {
    float4 b[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
    if (qg < 3) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bn[nt] = wp[((qg + 1) * NT + nt) * 64 + lane];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[4 * qg + 0], b[nt].x, acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[4 * qg + 1], b[nt].y, acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[4 * qg + 2], b[nt].z, acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[4 * qg + 3], b[nt].w, acc[nt], 0, 0, 0);
  }
}

template <int NT0>
__global__ __launch_bounds__(512, 1) void k_conv_fused(FusedArgs a) {
  extern __shared__ float4 lds4[];
  float* __restrict__ lds = reinterpret_cast<float*>(lds4);
  const int J = a.J;                                  // rows of the x / coefficient tiles (even, >= every tile span)
  float* __restrict__ x_lds = lds;                    // [J][XR]
  float* __restrict__ c_lds = x_lds + J * a.XR;       // [2][4][J][32]
  int* __restrict__ pm_lds = reinterpret_cast<int*>(c_lds + 2 * 4 * J * 32);  // [J][32] (slot_a+1) | (slot_b+1) << 8
  float4* __restrict__ g_lds = reinterpret_cast<float4*>(pm_lds + J * 32);     // [32][S4] vhat per edge slot
  int* __restrict__ deg_lds = reinterpret_cast<int*>(g_lds + 32 * a.S4);        // [32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int slice = blockIdx.x % a.n_slices;
  const int tile = blockIdx.x / a.n_slices;
  const int n0 = tile * 32;
  const int2 span = a.tile_span[tile];
  const int rows = span.y - span.x;
  const int Jt = (rows + 1) & ~1;  // K extent of the forming products for this tile

  // ---- once per workgroup: feature rows, edge geometry, (source, destination) -> slot map
  for (int idx = tid; idx < Jt * a.XS; idx += 512) {
    const int jl = idx / a.XS, c = idx - jl * a.XS;
    x_lds[jl * a.XR + c] = (jl < rows) ? a.x[(size_t)(span.x + jl) * a.XS + c] : 0.f;
  }
  for (int idx = tid; idx < J * 32; idx += 512) pm_lds[idx] = 0;
  if (tid < 32) deg_lds[tid] = (n0 + tid < a.n_atoms) ? a.deg[n0 + tid] : 0;
  __syncthreads();
  for (int idx = tid; idx < 32 * a.S4; idx += 512) {
    const int il = idx / a.S4, t = idx - il * a.S4;
    if (t < deg_lds[il]) {
      const size_t e = (size_t)(n0 + il) * a.S + t;
      const int sj = a.esrc[e];
      const float4 geo = a.egeo[e];
      g_lds[idx] = make_float4(0.f, geo.x, geo.y, geo.z);
      const int jl = (sj & 0x7fffffff) - span.x;
      // radial and bonded edges of one (source, destination) pair go to different byte fields: no write conflict
      // within a field because a destination's radial neighbours (and its bonded sources) are distinct atoms
      atomicOr(&pm_lds[jl * 32 + il], (sj < 0) ? ((t + 1) << 8) : (t + 1));
    }
  }

  f32x16 acc0[NT0];
  f32x16 acc1[3];
#pragma unroll
  for (int nt = 0; nt < NT0; ++nt)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc0[nt][q] = 0.f;
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1[m][q] = 0.f;
  __syncthreads();

  const int k_begin = a.slice_k[slice], k_end = a.slice_k[slice + 1];
  const int* __restrict__ my_units = a.wave_units + wave * a.max_units;
  for (int k = k_begin; k < k_end; ++k) {
    float* __restrict__ cb = c_lds + ((k - k_begin) & 1) * 4 * J * 32;
    // ---- coefficient tiles of hidden unit k: gather through the slot map
    for (int p = tid; p < Jt * 32; p += 512) {
      const int il = p & 31;
      const int pmv = pm_lds[p];
      const int ta = pmv & 0xff, tb = (pmv >> 8) & 0xff;
      float c0 = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
      if (ta) {
        const float hv = a.h[((size_t)(n0 + il) * a.S + (ta - 1)) * JAMUN_HS + k];
        const float4 g = g_lds[il * a.S4 + ta - 1];
        c0 = hv; cx = hv * g.y; cy = hv * g.z; cz = hv * g.w;
      }
      if (tb) {
        const float hv = a.h[((size_t)(n0 + il) * a.S + (tb - 1)) * JAMUN_HS + k];
        const float4 g = g_lds[il * a.S4 + tb - 1];
        c0 += hv; cx = fmaf(hv, g.y, cx); cy = fmaf(hv, g.z, cy); cz = fmaf(hv, g.w, cz);
      }
      cb[p] = c0;
      cb[J * 32 + p] = cx;
      cb[2 * J * 32 + p] = cy;
      cb[3 * J * 32 + p] = cz;
    }
    __syncthreads();  // tiles are double buffered: one barrier per k orders build(k) before use(k) and use(k-1) before build(k+1)

    const float4* __restrict__ wk = a.wpack + (size_t)k * a.k_stride;
    for (int ui = 0; ui < a.max_units; ++ui) {
      const int uid = my_units[ui];
      if (uid < 0) break;
      const int4 ud = a.units[2 * uid];       // {n_terms, out, wofs (float4 units / 64), 0}
      const int4 ut = a.units[2 * uid + 1];   // up to 3 terms, each packed: xcol0 | stride << 12 | ctype << 16 | neg << 20
      f32x16 af;
#pragma unroll
      for (int q = 0; q < 16; ++q) af[q] = 0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (t >= ud.x) break;  // wave-uniform
        const int td = t == 0 ? ut.x : (t == 1 ? ut.y : ut.z);
        const int xcol0 = td & 0xfff, stride = (td >> 12) & 0xf, ctype = (td >> 16) & 0xf;
        const float sgn = (td >> 20) & 1 ? -1.f : 1.f;
        const float* __restrict__ xp = x_lds + hh * a.XR + xcol0 + r * stride;
        const float* __restrict__ cp = cb + ctype * J * 32 + hh * 32 + r;
        for (int s = 0; s < (Jt >> 1); ++s) {
          const float av = xp[2 * s * a.XR];
          const float bv = cp[2 * s * 32] * sgn;
          af = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, af, 0, 0, 0);
        }
      }
      const float4* __restrict__ wp = wk + (size_t)ud.z * 64;
      if (ud.y == 0) unit_main<NT0>(acc0, af, wp, lane);
      else if (ud.y == 1) unit_main<1>(reinterpret_cast<f32x16(&)[1]>(acc1[0]), af, wp, lane);
      else if (ud.y == 2) unit_main<1>(reinterpret_cast<f32x16(&)[1]>(acc1[1]), af, wp, lane);
      else unit_main<1>(reinterpret_cast<f32x16(&)[1]>(acc1[2]), af, wp, lane);
    }
  }
  __syncthreads();

  // ---- fixed-order cross-wave reduction through LDS (wave 0 += wave 1, 2, ... 7), then wave 0 stores the slab rows
  float* __restrict__ red = lds;  // (NT0 + 3) * 16 * 64 floats
  for (int w = 1; w < FUSED_WAVES; ++w) {
    if (wave == w) {
#pragma unroll
      for (int nt = 0; nt < NT0; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[(nt * 16 + q) * 64 + lane] = acc0[nt][q];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[((NT0 + m) * 16 + q) * 64 + lane] = acc1[m][q];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int nt = 0; nt < NT0; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc0[nt][q] += red[(nt * 16 + q) * 64 + lane];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc1[m][q] += red[((NT0 + m) * 16 + q) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave == 0) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = (q & 3) + 8 * (q >> 2) + 4 * hh;
      const size_t base = (size_t)slice * a.n_pad + n0 + row;
#pragma unroll
      for (int nt = 0; nt < NT0; ++nt) a.partial0[base * (NT0 * 32) + nt * 32 + r] = acc0[nt][q];
#pragma unroll
      for (int m = 0; m < 3; ++m) a.partial1[(base * 3 + m) * 32 + r] = acc1[m][q];
    }
  }
}

size_t fused_lds_bytes(int J, int XR, int S4, int nt0) {
  size_t fl = (size_t)J * XR + 2 * 4 * (size_t)J * 32 + (size_t)J * 32 + 32 * (size_t)S4 * 4 + 32;
  const size_t red = (size_t)(nt0 + 3) * 16 * 64;
  if (fl < red) fl = red;
  return sizeof(float) * ((fl + 3) & ~(size_t)3);
}

int launch_conv_fused(const FusedArgs& a, int nt0, hipStream_t st) {
  const int grid = a.n_tiles * a.n_slices;
  const size_t smem = fused_lds_bytes(a.J, a.XR, a.S4, nt0);
  if (smem > JAMUN_MAX_DYN_LDS) return -2;
  switch (nt0) {
    case 1: hipLaunchKernelGGL((k_conv_fused<1>), dim3(grid), dim3(512), smem, st, a); return 0;
    case 2: hipLaunchKernelGGL((k_conv_fused<2>), dim3(grid), dim3(512), smem, st, a); return 0;
    case 3: hipLaunchKernelGGL((k_conv_fused<3>), dim3(grid), dim3(512), smem, st, a); return 0;
    case 4: hipLaunchKernelGGL((k_conv_fused<4>), dim3(grid), dim3(512), smem, st, a); return 0;
    case 5: hipLaunchKernelGGL((k_conv_fused<5>), dim3(grid), dim3(512), smem, st, a); return 0;
    default: return -1;
  }
}

int conv_fused_set_max_lds() {
  hipError_t e = hipSuccess;
#define SETATTR(NT)                                                                                                \
  if (e == hipSuccess)                                                                                             \
    e = hipFuncSetAttribute((const void*)k_conv_fused<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS);
  SETATTR(1) SETATTR(2) SETATTR(3) SETATTR(4) SETATTR(5)
#undef SETATTR
  return e == hipSuccess ? 0 : -1;
}
