// jamun_conv_initv.hip — conv contraction of the INITIAL projector, edge by edge on the vector ALUs.
//
// Same mathematics as jamun_conv_init.hip (src/jamun/e3tools/nn/_conv.py:93-119 re-associated by destination; the node inputs
// are the constant scaled embeddings, src/jamun/model/atom_embedding.py:58-76, so input x weight is tabulated per distinct
// embedding row):
//     out0[i][w]    = sum_k sum_{e->i} h~_e[k]        TT[k][uid_j][w]          (w < 152 scalar rows)
//     out1[i][m][w] = sum_k sum_{e->i} h~_e[k] v_e[m] TT[k][uid_j][160 + w]   (w < 32  vector rows)
// jamun_conv_init.hip applies the table with MFMAs from dense (source x destination) coefficient tiles, most of whose entries are
// structural zeros for small molecules.  Here the layer is what it is per (edge, k): ONE 184-float row gather and 248 FMAs —
// the forming loop of jamun_conv_dg.hip with table rows in place of feature rows and no contraction behind it:
//   * work item, tiles, segment lists, partial slabs: those of k_conv_dg (the node update sums the slabs of a tile);
//   * 8 waves, each owns 4 destinations, two at a time: lane = (half h -> destination, u = 0..31).  Lane u accumulates the
//     scalar columns 4u..4u+3 and 128+u and the vector column u (three planes): 4 packed FMAs per edge and destination pair on
//     the register pairs the loads deliver (coefficients (c, c vx | c vy, c vz) broadcast from the per-wave LDS table of
//     jamun_conv_dg.hip; row fragment (s0 s1 s2 s3) + (s4, v));
//   * per k the table rows of the tile's source atoms, TT2[k][uid_j][192] ([u][4] = s0 s1 s2 s3, then [u][2] = s4 v: both reads
//     of a half-wave are contiguous — a [u][8] row read at a 32-byte lane stride was bank-conflict bound: 0.152 instead of 0.113 ms on cfg2),
//     are staged into LDS by all threads, double buffered: the loads of k+1 are issued before the edge loops of k and stored
//     after them; one LDS-only barrier per k;
//   * the accumulators stay in registers over the k run of the segment and go straight to the partial slab.
// Two row buffers when they fit (spans up to ~90 source rows), one buffer above that (up to ~170 rows: a 166-atom molecule);
// only batches with source row blocks keep jamun_conv_init.hip.
// Measured on MI355X (initial-projector launch): 17 atoms x 256 walkers 0.123 -> 0.113 ms, 33 x 256: 0.398 -> 0.328 ms.  The kernel is
// LDS-bound (3 reads per edge, destination pair and k); the staged rows are a native vector array — as HIP float4 structs kept
// across the loop they were not promoted to registers and went through scratch memory.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jamun_internal.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define IV_WAVES 8
#define IV_THREADS (64 * IV_WAVES)
#define IV_NDP 2
#define IV_ROW 192      // floats per staged table row: [u][4] scalar columns 4u..4u+3 | [u][2] (column 128+u, vector column u)
#define IV_LROW 196     // LDS row stride in floats (784 B: consecutive rows start 4 banks apart; at 768 B every row starts on bank 0 and
                        // the two halves of a wave, which read different rows, collide)
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define RFL(v) __builtin_amdgcn_readfirstlane(v)

__device__ __forceinline__ f32x2 iv_pk_bhi(f32x2 a2, f32x2 b2, f32x2 c2) {  // (a.x b.y + c.x, a.y b.y + c.y)
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(c2) : "v"(a2), "v"(b2));
  return c2;
}
__device__ __forceinline__ float4 iv_lds_f4(int addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  const f32x4 v = *(const __attribute__((address_space(3))) f32x4*)(unsigned)addr;
  return make_float4(v.x, v.y, v.z, v.w);
#else
  (void)addr;
  return make_float4(0.f, 0.f, 0.f, 0.f);
#endif
}
__device__ __forceinline__ float2 iv_lds_f2(int addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  const f32x2 v = *(const __attribute__((address_space(3))) f32x2*)(unsigned)addr;
  return make_float2(v.x, v.y);
#else
  (void)addr;
  return make_float2(0.f, 0.f);
#endif
}
__device__ __forceinline__ int iv_lds_address(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
#else
  (void)p;
  return 0;
#endif
}

__host__ __device__ inline size_t iv_lds_floats(int rs, int pmax, int nbuf) {
  // rows [nbuf][rs][196] | tabA [32][pmax] float4 | tabJ [32][pmax] | deg [32] | uid [rs]
  return (size_t)nbuf * rs * IV_LROW + (size_t)32 * pmax * 4 + (size_t)32 * pmax + 32 + rs;
}

#ifdef JAMUN_STAMP
#include <stdio.h>
__device__ unsigned long long g_ivstamp[8];  // load issue, edge loops, wait + store, tables, barrier, prologue, epilogue
#define ISTAMP(t) do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); } while (0)
#define IACC(slot, t1, t0) do { iv_acc[slot] += (t1) - (t0); } while (0)
#else
#define ISTAMP(t) do { } while (0)
#define IACC(slot, t1, t0) do { } while (0)
#endif
// IV_SB: staged float4 per thread held in registers over the edge loops (8 x 512 x 16 B = 85 rows, 16: 170 rows).
// DBL: two row buffers (one barrier per k); otherwise ONE buffer (molecules whose rows fill LDS: the store of the next rows
// waits behind a second barrier, the loads are still issued before the edge loops).
template <int IV_SB, bool DBL>
__global__ __launch_bounds__(IV_THREADS) void k_conv_init_v(InitVArgs a) {
  constexpr int IV_U = IV_SB > 8 ? 2 : 4;  // edges per batch of the edge loops (the 16 staging registers of large spans leave room for 2)
  unsigned long long iv_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0, t1 = 0;
  (void)iv_acc; (void)t0; (void)t1;
  extern __shared__ float4 lds4[];
  float* __restrict__ lds = reinterpret_cast<float*>(lds4);
  float* __restrict__ rows_lds = lds;                                                       // [2][RS][196]
  float4* __restrict__ tabA = reinterpret_cast<float4*>(rows_lds + (size_t)(DBL ? 2 : 1) * a.RS * IV_LROW);  // [32][PMAX]
  int* __restrict__ tabJ = reinterpret_cast<int*>(tabA + 32 * a.PMAX);                        // [32][PMAX] LDS address of the row in buffer 0
  int* __restrict__ deg_lds = tabJ + 32 * a.PMAX;                                            // [32]
  int* __restrict__ uid_lds = deg_lds + 32;                                                  // [RS]
  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = RFL(tid0 >> 6);
  const int PMAX = a.PMAX;
  const int buf_bytes = a.RS * IV_LROW * 4;

  for (int sgi = 0; sgi < a.max_segs; ++sgi) {
    const int4 sg0 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2];
    const int4 sg1 = a.segs[((size_t)blockIdx.x * a.max_segs + sgi) * 2 + 1];
    const int tile = RFL(sg0.x);
    if (tile < 0) break;
    int lane = lane0;  // (opaque per-segment copy: see jamun_conv_dg.hip)
    asm volatile("" : "+v"(lane));
    const int tid = wave * 64 + lane;
    const int slab = RFL(sg0.y), k_begin = RFL(sg0.z), k_run = RFL(sg0.w) - k_begin, k_extra = RFL(sg1.x);
    const int nk = k_run + (k_extra >= 0 ? 1 : 0);
    auto k_of = [&](int kk) { kk = kk < 0 ? 0 : (kk < nk ? kk : nk - 1); return kk < k_run ? k_begin + kk : k_extra; };
    const int2 t_at = a.tile_atoms[tile];
    const int n0 = RFL(t_at.x), n_dst = RFL(t_at.y);
    const int2 span = a.tile_span[tile];
    const int s_lo = RFL(span.x), rows = RFL(span.y) - s_lo;

    if (tid < 32) deg_lds[tid] = (tid < n_dst) ? a.deg[n0 + tid] : 0;
    for (int j = tid; j < rows; j += IV_THREADS) uid_lds[j] = a.atom_uid[s_lo + j];
    LDS_BARRIER();

    // ---- staging of the table rows of hidden unit k: loads into registers, stores into row buffer b
    const int total = rows * (IV_ROW / 4);  // float4 per buffer
    // (macros, not lambdas: an array captured by reference in a closure is not promoted to registers — the staged rows went
    // through scratch memory, with the global-load latency exposed at every hidden unit)
    f32x4 sv[IV_SB];
#define IV_STAGE_LOAD(K)                                                                                             \
  do {                                                                                                               \
    const float* __restrict__ tk_ = a.tt2 + (size_t)(K) * a.tt2_kstride;                                            \
    _Pragma("unroll") for (int q = 0; q < IV_SB; ++q) {                                                              \
      const int idx = min(tid + q * IV_THREADS, total - 1);                                                          \
      const int jr = idx / (IV_ROW / 4);                                                                             \
      sv[q] = *reinterpret_cast<const f32x4*>(tk_ + (size_t)uid_lds[jr] * IV_ROW + 4 * (idx - jr * (IV_ROW / 4)));    \
    }                                                                                                                \
  } while (0)
#define IV_STAGE_STORE(K, B)                                                                                         \
  do {                                                                                                               \
    float* __restrict__ dst_ = rows_lds + (size_t)(B) * a.RS * IV_LROW;                                              \
    _Pragma("unroll") for (int q = 0; q < IV_SB; ++q) {                                                              \
      const int idx = tid + q * IV_THREADS;                                                                          \
      const int jr = idx / (IV_ROW / 4);                                                                             \
      if (idx < total) *reinterpret_cast<f32x4*>(dst_ + jr * IV_LROW + 4 * (idx - jr * (IV_ROW / 4))) = sv[q];        \
    }                                                                                                                \
    const float* __restrict__ tk_ = a.tt2 + (size_t)(K) * a.tt2_kstride; /* spans above 64 rows: the rest in a plain loop */ \
    for (int idx = tid + IV_SB * IV_THREADS; idx < total; idx += IV_THREADS) {                                       \
      const int jr = idx / (IV_ROW / 4);                                                                             \
      *reinterpret_cast<float4*>(dst_ + jr * IV_LROW + 4 * (idx - jr * (IV_ROW / 4))) =                              \
          *reinterpret_cast<const float4*>(tk_ + (size_t)uid_lds[jr] * IV_ROW + 4 * (idx - jr * (IV_ROW / 4)));      \
    }                                                                                                                \
  } while (0)
    IV_STAGE_LOAD(k_of(0));

    // ---- per-wave edge tables (as the forming waves of k_conv_dg)
    const int fw = wave;
    const int h = lane >> 5, u = lane & 31;
    const int rows_addr = iv_lds_address(rows_lds);
    float evx[IV_NDP][2], evy[IV_NDP][2], evz[IV_NDP][2];
    int hidx[IV_NDP][2], hidx2[IV_NDP], P[IV_NDP];
    const bool two_pages = a.S > 32;
#pragma unroll
    for (int dp = 0; dp < IV_NDP; ++dp) {
      const int i = 2 * IV_NDP * fw + 2 * dp + h;
      const int dg = deg_lds[i];
      hidx2[dp] = -1;
      int sj0 = 0;
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
          if (jl >= 0 && jl < rows) {
            const float4 ge = a.egeo[slot];
            evx[dp][pg] = ge.x; evy[dp][pg] = ge.y; evz[dp][pg] = ge.z;
            hidx[dp][pg] = slot;
            jofs = jl * IV_LROW * 4;
          }
        }
        if (t < PMAX) tabJ[i * PMAX + t] = jofs + rows_addr;
      }
      if (!two_pages) {
        // bonded edge folded into its radial twin (same source and unit vector; see jamun_conv_dg.hip)
        const bool is_b = u < dg && sj0 < 0;
        const unsigned long long bal = __ballot(is_b);
        const int nb = __popc((unsigned)(bal >> (32 * h)));
        const int nb_max = RFL(max(__popc((unsigned)bal), __popc((unsigned)(bal >> 32))));
        const int jraw = sj0 & 0x7fffffff;
        bool merged = false;
        for (int b = 0; b < nb_max; ++b) {
          const int lb = dg - nb + b;
          const int jb = __shfl(jraw, lb & 31, 32);
          const bool match = b < nb && u < dg && !is_b && jraw == jb && hidx[dp][0] >= 0 && hidx2[dp] < 0;
          if (match) hidx2[dp] = (n0 + i) * a.S + lb;
          const unsigned long long mb = __ballot(match);
          if (is_b && u == lb && (unsigned)(mb >> (32 * h)) != 0u) merged = true;
        }
        if (merged) hidx[dp][0] = -1;
        const unsigned long long act = __ballot(hidx[dp][0] >= 0);
        const int p_lo = 32 - __clz((unsigned)act), p_hi = 32 - __clz((unsigned)(act >> 32));
        P[dp] = RFL(max(p_lo, p_hi));
      } else {
        P[dp] = RFL(max(deg_lds[2 * IV_NDP * fw + 2 * dp], deg_lds[2 * IV_NDP * fw + 2 * dp + 1]));
      }
    }
    float hv[IV_NDP][2], hv2[IV_NDP];
    auto load_h = [&](int k) __attribute__((always_inline)) {
      const float* __restrict__ hk = a.h + (size_t)k * a.h_kstride;
#pragma unroll
      for (int dp = 0; dp < IV_NDP; ++dp) {
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) hv[dp][pg] = hk[hidx[dp][pg] >= 0 ? hidx[dp][pg] : n0 * a.S];
        hv2[dp] = hk[hidx2[dp] >= 0 ? hidx2[dp] : n0 * a.S];
      }
    };
    auto write_tab = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int dp = 0; dp < IV_NDP; ++dp) {
        const int i = 2 * IV_NDP * fw + 2 * dp + h;
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
          if (pg == 1 && !two_pages) continue;
          const int t = u + 32 * pg;
          float c = hidx[dp][pg] >= 0 ? hv[dp][pg] : 0.f;
          if (pg == 0 && hidx2[dp] >= 0) c += hv2[dp];
          if (t < PMAX) tabA[i * PMAX + t] = make_float4(c, c * evx[dp][pg], c * evy[dp][pg], c * evz[dp][pg]);
        }
      }
    };
    auto pk = [](f32x2 a2, f32x2 b2, f32x2 c2) -> f32x2 { return __builtin_elementwise_fma(a2, b2, c2); };

    // accumulators over the hidden units of the segment, per destination pair: scalar columns 4u..4u+3, (column 128+u, vector
    // plane 0), (vector planes 1, 2)
    f32x2 s01[IV_NDP], s23[IV_NDP], se[IV_NDP], so[IV_NDP];
#pragma unroll
    for (int dp = 0; dp < IV_NDP; ++dp) s01[dp] = s23[dp] = se[dp] = so[dp] = f32x2{0.f, 0.f};

    auto form = [&](int b) __attribute__((always_inline)) {
      const int offa = u * 16 + b * buf_bytes, offb = 512 + u * 8 + b * buf_bytes;  // this lane's 16 + 8 bytes of a row, in row buffer b
#pragma unroll
      for (int dp = 0; dp < IV_NDP; ++dp) {
        const int i = 2 * IV_NDP * fw + 2 * dp + h;
        const float4* __restrict__ ta = tabA + i * PMAX;
        const int* __restrict__ tj = tabJ + i * PMAX;
        const int nb = (a.dbg & 1) ? 0 : (P[dp] + IV_U - 1) / IV_U;
        float4 cA[IV_U], cB[IV_U], xA[IV_U], xB[IV_U];
        float2 yA[IV_U], yB[IV_U];
        int jA[IV_U], jB[IV_U];
        auto ld_j = [&](int (&J)[IV_U], int bb) {
          bb = bb < nb ? bb : nb - 1;
#pragma unroll
          for (int q = 0; q < IV_U; ++q) J[q] = tj[IV_U * bb + q];
        };
        auto ld_cx = [&](float4 (&c)[IV_U], float4 (&x)[IV_U], float2 (&y)[IV_U], const int (&J)[IV_U], int bb) {
          bb = bb < nb ? bb : nb - 1;
#pragma unroll
          for (int q = 0; q < IV_U; ++q) {
            c[q] = ta[IV_U * bb + q];       // c, c vx | c vy, c vz
            x[q] = iv_lds_f4(J[q] + offa);  // s0 s1 | s2 s3
            y[q] = iv_lds_f2(J[q] + offb);  // s4, v
          }
        };
        auto fm = [&](const float4 (&c)[IV_U], const float4 (&x)[IV_U], const float2 (&y)[IV_U]) {
#pragma unroll
          for (int q = 0; q < IV_U; ++q) {
            const float4 cf = c[q];
            s01[dp] = pk(f32x2{cf.x, cf.x}, f32x2{x[q].x, x[q].y}, s01[dp]);
            s23[dp] = pk(f32x2{cf.x, cf.x}, f32x2{x[q].z, x[q].w}, s23[dp]);
            se[dp] = pk(f32x2{cf.x, cf.y}, f32x2{y[q].x, y[q].y}, se[dp]);          // (c s4, c vx v)
            so[dp] = iv_pk_bhi(f32x2{cf.z, cf.w}, f32x2{y[q].x, y[q].y}, so[dp]);   // (c vy v, c vz v)
          }
        };
        if (nb > 0) {
          ld_j(jA, 0); ld_cx(cA, xA, yA, jA, 0); ld_j(jB, 1);
          for (int bb = 0; bb < nb; bb += 2) {
            ld_cx(cB, xB, yB, jB, bb + 1); ld_j(jA, bb + 2);
            fm(cA, xA, yA);
            if (bb + 1 < nb) {
              ld_cx(cA, xA, yA, jA, bb + 2); ld_j(jB, bb + 3);
              fm(cB, xB, yB);
            }
          }
        }
      }
    };

    // ---- the k run of the segment
    load_h(k_of(0));
    IV_STAGE_STORE(k_of(0), 0);
    write_tab();
    load_h(k_of(1));
    LDS_BARRIER();  // rows of the first hidden unit and every wave's tables in place
    for (int it = 0; it < nk; ++it) {
      ISTAMP(t0);
      if (it + 1 < nk) IV_STAGE_LOAD(k_of(it + 1));
      ISTAMP(t1); IACC(0, t1, t0);
      form(DBL ? (it & 1) : 0);
      ISTAMP(t0); IACC(1, t0, t1);
      if (it + 1 < nk) {
        if constexpr (!DBL) LDS_BARRIER();  // every wave is done with the rows of k
        // (DBL: nobody reads the other buffer: its hidden unit was finished before the last barrier)
        IV_STAGE_STORE(k_of(it + 1), DBL ? ((it + 1) & 1) : 0);
        ISTAMP(t1); IACC(2, t1, t0);
        write_tab();                               // this wave's private rows, read only by itself
        load_h(k_of(it + 2));
        ISTAMP(t0); IACC(3, t0, t1);
      }
      ISTAMP(t0);
      LDS_BARRIER();
      ISTAMP(t1); IACC(4, t1, t0);
    }

    // ---- segment end: accumulators -> partial slab (every lane owns its columns: no staging)
    {
      float* __restrict__ p0 = a.partial0 + ((size_t)slab * a.n_pad + n0) * (size_t)(a.nt0 * 32);
      float* __restrict__ p1 = a.partial1 + ((size_t)slab * a.n_pad + n0) * 96;
#pragma unroll
      for (int dp = 0; dp < IV_NDP; ++dp) {
        const int i = 2 * IV_NDP * fw + 2 * dp + h;
        if (i < n_dst) {
          *reinterpret_cast<float4*>(p0 + (size_t)i * (a.nt0 * 32) + 4 * u) = make_float4(s01[dp].x, s01[dp].y, s23[dp].x, s23[dp].y);
          p0[(size_t)i * (a.nt0 * 32) + 128 + u] = u < 24 ? se[dp].x : 0.f;
          p1[i * 96 + u] = se[dp].y;
          p1[i * 96 + 32 + u] = so[dp].x;
          p1[i * 96 + 64 + u] = so[dp].y;
        }
      }
    }
    LDS_BARRIER();  // the next segment rewrites the tables and row buffers
  }
#ifdef JAMUN_STAMP
  if (lane0 == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_ivstamp[i], iv_acc[i]);
#endif
}

void conv_initv_print_stamps() {
#ifdef JAMUN_STAMP
  unsigned long long v[8], z[8] = {};
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_ivstamp), sizeof(v)) != hipSuccess) return;
  fprintf(stderr, "init-v stamps (cycles summed over waves): load issue %llu, edge loops %llu, wait+store %llu, tables %llu, barrier %llu\n", v[0], v[1], v[2],
          v[3], v[4]);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ivstamp), z, sizeof(z));
#endif
}

size_t conv_initv_lds_bytes(int rs, int pmax, int nbuf) { return sizeof(float) * ((iv_lds_floats(rs, pmax, nbuf) + 3) & ~(size_t)3); }

int launch_conv_initv(const InitVArgs& a, int grid, hipStream_t st) {
  const size_t smem = conv_initv_lds_bytes(a.RS, a.PMAX, a.nbuf);
  if (smem > JAMUN_MAX_DYN_LDS || a.nt0 != 5) return -1;
  if (a.nbuf == 2 && a.RS <= 85) hipLaunchKernelGGL((k_conv_init_v<8, true>), dim3(grid), dim3(IV_THREADS), smem, st, a);
  else if (a.nbuf == 2) hipLaunchKernelGGL((k_conv_init_v<16, true>), dim3(grid), dim3(IV_THREADS), smem, st, a);
  else if (a.RS <= 85) hipLaunchKernelGGL((k_conv_init_v<8, false>), dim3(grid), dim3(IV_THREADS), smem, st, a);
  else hipLaunchKernelGGL((k_conv_init_v<16, false>), dim3(grid), dim3(IV_THREADS), smem, st, a);
  return 0;
}

int conv_initv_set_max_lds() {
  return (hipFuncSetAttribute((const void*)k_conv_init_v<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) == hipSuccess &&
          hipFuncSetAttribute((const void*)k_conv_init_v<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) == hipSuccess &&
          hipFuncSetAttribute((const void*)k_conv_init_v<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) == hipSuccess &&
          hipFuncSetAttribute((const void*)k_conv_init_v<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, JAMUN_MAX_DYN_LDS) == hipSuccess)
             ? 0
             : -1;
}
