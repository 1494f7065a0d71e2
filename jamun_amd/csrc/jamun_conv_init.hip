// jamun_conv_init.hip — the conv contraction of the INITIAL projector (first ConvBlock of the denoiser).
//
// Its node inputs are the noise-scaled atom embeddings (src/jamun/model/atom_embedding.py:58-76,
// noise_conditioning.py:50-54): constant per (topology, sigma), and identical for all atoms that share the four
// embedding indices.  So the product of the inputs with the per-hidden-unit weights is precomputed once per sampler,
//     TT[k][uid][w] = sum_u x_uid[u] * W~[(k,u)][w]        (w < 152: scalar rows, 160 <= w < 192: vector rows)
// (uid = distinct embedding row), and the layer reduces to the "apply" half of jamun_conv_fused.hip:
//     out0[i][w]    = sum_k sum_j C^h_k[j][i]      TT[k][uid_j][w]
//     out1[i][m][w] = sum_k sum_j C^{h v_m}_k[j][i] TT[k][uid_j][160 + w]
// with C the same dense coefficient tiles (src/jamun/e3tools/nn/_conv.py:93-119 re-associated by destination, see
// jamun_conv_fused.hip).  No forming stage, no parked tiles: LDS holds only coefficient tiles, so TWO hidden units are
// processed per barrier interval (the fixed per-interval cost dominates this small layer), and every wave has the same
// work: wave w < NT0 owns scalar-row tile w, waves NT0..NT0+2 the vector planes, 16 MFMAs per (k, 32-row tile of sources).
// A operand: coefficient tile rows (transposed tiles, one ds_read_b128 per 4 K-steps); B operand: TT rows gathered from
// the L2-resident table straight into accumulator-register order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jamun_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define IW 8
#define IT (64 * IW)
#define IPAIRS ((64 * 32 + IT - 1) / IT)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define RFL(v) __builtin_amdgcn_readfirstlane(v)

__global__ __launch_bounds__(IT, 1) void k_conv_init(InitArgs a) {
  extern __shared__ float4 lds4[];
  float* __restrict__ lds = reinterpret_cast<float*>(lds4);
  const int JR = a.JR, CT = 32 * JR;
  float* __restrict__ cT = lds;                                  // [2 buffers][2 k][4 types][32][JR]
  int* __restrict__ pm = reinterpret_cast<int*>(cT + 8 * CT);    // [32][JR] slot map (prologue; aliases the second buffer)
  int* __restrict__ deg_lds = reinterpret_cast<int*>(cT + 16 * CT);  // [32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = RFL(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  // owner role of this wave and the column of its output tile in the TT table
  const int own_kind = wave < a.nt0 ? 0 : (wave < a.nt0 + 3 ? 1 : -1);
  const int own_idx = own_kind == 0 ? wave : wave - a.nt0;
  const int tt_col = (own_kind == 0 ? 32 * wave : 32 * a.nt0) + r;
  const int c_type = own_kind == 0 ? 0 : 1 + own_idx;  // coefficient tile this owner contracts with

  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int4 sg0 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2];
    const int4 sg1 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2 + 1];
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) { return kk < k_run ? k_begin + kk : k_extra; };
    const int2 t_at = a.tile_atoms[tile];
    const int n0 = t_at.x, n_dst = t_at.y;
    const int2 span = a.tile_span[tile];
    const int rows = span.y - span.x;
    const int Jt4 = (rows + 3) & ~3;
    const int n_jt = (Jt4 + 31) >> 5;  // 32-row tiles of source atoms (<= 2)

    // ---- per segment: (source, destination) -> edge-slot map, per-pair slots and unit vector, TT row offsets
    for (int idx = tid; idx < CT; idx += IT) pm[idx] = 0;
    if (tid < 32) deg_lds[tid] = (tid < n_dst) ? a.deg[n0 + tid] : 0;
    __syncthreads();
    for (int idx = tid; idx < 32 * a.S; idx += IT) {
      const int il = idx / a.S, t = idx - il * a.S;
      if (t < deg_lds[il]) {
        const int sj = a.esrc[(size_t)(n0 + il) * a.S + t];
        const int jl = (sj & 0x7fffffff) - span.x;
        // radial / bonded edge of the pair: separate bytes; sources outside the span belong to another tile (row blocks)
        if (jl >= 0 && jl < rows) atomicOr(&pm[il * JR + jl], (sj < 0) ? ((t + 1) << 8) : (t + 1));
      }
    }
    // TT row offsets of the source rows this lane's B fragments come from: register q of row tile jt <-> source row
    // 32 jt + (q & 3) + 8 (q >> 2) + 4 hh  (rows past the span read row 0 of the table; their coefficients are zero)
    int uoff[2][16];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 32 * jt + (q & 3) + 8 * (q >> 2) + 4 * hh;
        uoff[jt][q] = (row < rows ? a.atom_uid[span.x + row] : 0) * a.tt_row + tt_col;
      }
    __syncthreads();
    int p_oa[IPAIRS], p_ob[IPAIRS], p_has = 0;
    float p_g[IPAIRS][3];
    float p_h[2][IPAIRS][2];  // h~ of the two k of the next interval: [k][pair][radial, bonded]
#pragma unroll
    for (int i = 0; i < IPAIRS; ++i) {
      const int p = tid + IT * i;
      const int pt = (p < CT) ? pm[p] : 0;
      const int il = p / JR;
      const int ta = pt & 0xff, tb = (pt >> 8) & 0xff;
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
    auto load_h = [&](int slot, int k) {
      const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
      for (int i = 0; i < IPAIRS; ++i) {
        p_h[slot][i][0] = hk[p_oa[i]];
        p_h[slot][i][1] = hk[p_ob[i]];
      }
    };
    auto write_c = [&](int slot, float* __restrict__ cbuf) {  // cbuf: [4 types][CT] of one k
#pragma unroll
      for (int i = 0; i < IPAIRS; ++i) {
        const int p = tid + IT * i;
        if (p < CT) {
          const float ha = (p_has >> (2 * i)) & 1 ? p_h[slot][i][0] : 0.f, hb = (p_has >> (2 * i + 1)) & 1 ? p_h[slot][i][1] : 0.f;
          cbuf[p] = ha + hb;
          cbuf[CT + p] = fmaf(hb, p_g[i][0], ha * p_g[i][0]);
          cbuf[2 * CT + p] = fmaf(hb, p_g[i][1], ha * p_g[i][1]);
          cbuf[3 * CT + p] = fmaf(hb, p_g[i][2], ha * p_g[i][2]);
        }
      }
    };
    const int n_iv = (nk + 1) >> 1;  // intervals of two k
    auto kc = [&](int kk) { return k_of(kk < nk ? kk : nk - 1); };  // clamped: loads are unconditional

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    load_h(0, kc(0));
    load_h(1, kc(1));
    __syncthreads();  // (every thread has read its slot-map entries before buffer 1 is written below)
    write_c(0, cT);
    write_c(1, cT + 4 * CT);
    LDS_BARRIER();

    for (int iv = 0; iv < n_iv; ++iv) {
      const float* __restrict__ cb = cT + (iv & 1) * 8 * CT;
      load_h(0, kc(2 * iv + 2));  // the two k of the next interval: in flight behind this interval's MFMAs
      load_h(1, kc(2 * iv + 3));
      if (own_kind >= 0) {
        // (k, row tile) steps of the interval in a fixed order; B fragments are gathered two steps ahead of their MFMAs
        float tb0[16], tb1[16], tb2[16];
        auto gather = [&](float (&tb)[16], int step) {
          const int kk = step >> 1, jt = step & 1;
          const float* __restrict__ tk = a.tt + (size_t)kc(2 * iv + kk) * a.tt_kstride;
#pragma unroll
          for (int q = 0; q < 16; ++q) tb[q] = tk[uoff[jt][q]];
        };
        auto apply = [&](const float (&tb)[16], int step) {
          const int kk = step >> 1, jt = step & 1;
          if (2 * iv + kk >= nk || jt >= n_jt) return;  // wave-uniform
          const float* __restrict__ cm = cb + (kk * 4 + c_type) * CT + r * JR + 32 * jt + 4 * hh;
          float4 cv[4];
#pragma unroll
          for (int qg = 0; qg < 4; ++qg) {
            cv[qg] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (32 * jt + 4 * hh + 8 * qg < JR) cv[qg] = *reinterpret_cast<const float4*>(cm + 8 * qg);
          }
#pragma unroll
          for (int qg = 0; qg < 4; ++qg) {
            acc = MFMA(cv[qg].x, tb[4 * qg + 0], acc);
            acc = MFMA(cv[qg].y, tb[4 * qg + 1], acc);
            acc = MFMA(cv[qg].z, tb[4 * qg + 2], acc);
            acc = MFMA(cv[qg].w, tb[4 * qg + 3], acc);
          }
        };
        // three register sets rotate: a set is refilled one full step after the MFMAs that read it were issued
        gather(tb0, 0);
        gather(tb1, 1);
        gather(tb2, 2); apply(tb0, 0);
        gather(tb0, 3); apply(tb1, 1);
        apply(tb2, 2);
        apply(tb0, 3);
      }
      if (iv + 1 < n_iv) {
        float* __restrict__ nb = cT + ((iv + 1) & 1) * 8 * CT;
        write_c(0, nb);
        write_c(1, nb + 4 * CT);
      }
      LDS_BARRIER();
    }

    // ---- the owner stores its tile of the segment's partial slab
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
    __syncthreads();  // the next segment rewrites the slot map over coefficient buffer 1
  }
}

size_t conv_init_lds_bytes(int JR) { return sizeof(float) * ((size_t)16 * 32 * JR + 32); }

int launch_conv_init(const InitArgs& a, int grid, hipStream_t st) {
  const size_t smem = conv_init_lds_bytes(a.JR);
  if (smem > JAMUN_MAX_DYN_LDS) return -2;
  hipLaunchKernelGGL(k_conv_init, dim3(grid), dim3(IT), smem, st, a);
  return 0;
}

int conv_init_set_max_lds() {
  return hipFuncSetAttribute((const void*)k_conv_init, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) ==
                 hipSuccess
             ? 0
             : -1;
}
