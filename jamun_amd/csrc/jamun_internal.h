// Internal structures shared by jamun_kernels.hip (device code) and jamun_api.cpp (host runtime).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GEOM_LDS_ATOMS 1024  // k_geom / k_finalize keep a walker's coordinates in LDS up to this many atoms (12 KB)
#define JAMUN_MAX_NEIGHBORS 32  // torch_geometric.nn.radius_graph default (src/jamun/model/denoiser.py:149)
#define JAMUN_HROWS 65          // rows of the per-layer radial-MLP activation table: 64 hidden units + the bias row

// conv kernel geometry (jamun_conv.hip)
#define JAMUN_KSUB0 5  // hidden units per k-subgroup, scalar-output rows (subgroups of 4 or 5)
#define JAMUN_KSUB1 2  // hidden units per k-subgroup, vector-output rows (3 planes per workgroup; subgroups of 1 or 2)
#define JAMUN_MAX_DYN_LDS (160 * 1024 - 256)  // dynamic LDS per workgroup: 160 KiB minus the kernels' static words (at most 256 B: the vote of __syncthreads_or)
#define JAMUN_MAX_BATCH 128  // edge batches (of 4) per wave: 8 atoms x ceil(S / 4); limits the edge stride S to 64

// zeta types of a u-block (what the edge feeds into the contraction)
#define JAMUN_T_X0 0   // x0_j[u]                      -> scalar rows
#define JAMUN_T_DOT 1  // x1_j[u'] . vhat              -> scalar rows
#define JAMUN_T_X0V 2  // x0_j[u] * vhat[m]            -> vector rows
#define JAMUN_T_X1C 3  // lanes<32: x1_j[u'][m] ; lanes>=32: (x1_j[u'] x vhat)[m]  -> vector rows
#define JAMUN_XOFF_CROSS 0x10000

struct ConvArgs {
  // graph
  const int* deg;
  const int* esrc;
  const float4* egeo;
  const float* h;  // [hidden unit k (65 rows)][h_kstride >= n_atoms*S]: radial-MLP activations per edge slot
  size_t h_kstride;
  const float* x;  // [n_atoms][XS]
  int n_atoms, n_pad, n_tiles, S, S4, XS;  // S4 = S rounded up to a multiple of 4 (edge batches)
  // problem
  const float4* wpack;   // 16-byte weight fragments: 4 consecutive K-steps of one lane
  const int4* chunks;    // {ublk, k0 | ks << 16, first weight group, number of weight groups (of 4 K-steps)}
  const int* slice_ptr;  // [n_slices+1] chunk ranges
  const int4* ublk;      // {type, nu, xcol0, width}
  const int* lane_xoff;  // [n_ublk][64] column relative to xcol0 | JAMUN_XOFF_CROSS
  float* partial;        // [n_slices][n_pad][RC][NT*32]
  int n_slices;
};

// destination-grouped conv with VALU forming (jamun_conv_dg.hip): hidden layers with irreps 120x0e + 32x1e
struct DgArgs {
  const int* deg;
  const int* esrc;
  const float4* egeo;
  const float* h;  // [hidden unit k (65 rows)][h_kstride]
  size_t h_kstride;
  const float* x;  // [n_atoms][XS = 216]
  int n_pad, S, XS;
  int RS;    // rows of the LDS source-row tile: multiple of 16, >= every tile span
  int PMAX;  // per-destination capacity of the coefficient tables (= edge stride S)
  const int2* tile_span;   // [n_tiles] {lo, hi} source atoms of the tile
  const int2* tile_atoms;  // [n_tiles] {first destination atom, number of destination atoms (<= 32)}
  const int4* segs;        // [grid][max_segs][2]: {tile (-1 end), slab, k_begin, k_end}, {k_extra (-1 none), 0, 0, 0}
  int max_segs, row_blocks, nt0;
  int alt;  // kernel mode: 0 two-phase resident, 1 alternating residency of the source rows (large molecules), 2 single phase (small spans)
  int dbg;  // tuning aid (JAMUN_DG_DBG): 1 forming waves skip their edge loops, 2 matrix waves skip their MFMAs
  float* dump;  // diagnostic builds (-DJAMUN_DUMP): workgroup 0 copies the A tiles of its second hidden unit here
  // weights, 64-lane x float4 blocks (one block = the B operand of 4 consecutive MFMAs):
  const float4* wx;  // [k][5 output tiles][16 groups (15 = zeros)]   scalar inputs -> scalar rows   (32x32x2: u = 8g + 4hh + st)
  const float4* wd;  // [k][5][4]                                      dot(x1, v)    -> scalar rows
  const float4* wv;  // [k][2 column halves][4]   x1 | cross -> vector rows                          (16x16x4: kappa = 16g + 4kq + st)
  const float* T;    // [k][n_atoms][32]  T_k[j][w'] = sum_u x0_j[u] W[(k,u)][w'] from the pre-pass k_tprod (scalar inputs -> vector rows)
  int n_atoms;
  // f16x3 contraction (emu = 1): every fp32 operand is split into two f16 terms (hi + lo, 22+ bits) and a product is three
  // v_mfma_f32_*_f16 (hi hi + hi lo + lo hi, fp32 accumulate).  Weights are split on the host after scaling by 2^sB; 64-lane x
  // 16-byte blocks (8 halves = one lane's B fragment of one 32x32x16 / 16x16x32 MFMA), planes p = 0 (hi), 1 (lo):
  int emu;
  // wh [k][matrix wave w][34 blocks] in the wave's consumption order:
  //   6 c + {0..3}  scalar inputs, chunk c = 0..3: own output tile w, groups 2c and 2c+1 of 16 inputs (hi, lo, hi, lo; inputs 120..127 zeros)
  //   6 c + {4, 5}  the wave's group w + 4 (c >> 1) of output tile 4 when (c & 1) == (w >> 1) (else unused)
  //   24..27 dot(x1, v) inputs, own tile, groups 0, 1;  28, 29 output tile 4, group w (w < 2)
  //   30..33 x1 | cross inputs of the vector planes (16x16x32: kappa = 32 G + 8 kq + j), column half w >> 1, groups 0, 1
  const float4* wh;
  int sB;             // (0: the weights are balanced per input channel and per output column, see gx / cf0 / cf1)
  // f16x3 balancing (jamun_api.cpp: build_layer): the feature rows are multiplied by gx [216] (2^e_u per element, layout of a row) when
  // they are staged; the weights carry 2^-e_u per input and 2^sB_w per output column; cf0 [160] / cf1 [32] = 2^-sB_w undo the latter
  const float *gx, *cf0, *cf1;
  float hmax2;        // 2 * upper bound of |h~| over the layer (static, from the radial MLP's first layer): bounds the A tiles
  float* partial0;   // [slab][n_pad][nt0*32]
  float* partial1;   // [slab][n_pad][3][32]
};

// conv with the A operand formed on the matrix cores and chained into the contraction (jamun_conv_mf.hip): hidden layers with
// irreps 120x0e + 32x1e, tiles whose source span fits a K = 64 window
// k_geom's pair table (one word per edge slot; several edges of one ordered (source, destination) pair — a radial edge and the bonded
// edge(s) of the same atoms — share one entry of the matrix-formed kernels' coefficient tiles): the FIRST slot of the pair owns the
// entry and lists the slots of up to two others (bits 0..6 and 7..13: slot + 1, 0 = none); the others carry JAMUN_EP_OWNED.
#define JAMUN_EP_OWNED 0x80000000u
#define JAMUN_EP_OVERFLOW 0x40000000  // more than three edges of one pair

struct MfArgs {
  const int* deg;
  const int* epair;  // [n_atoms * S] pair table (above)
  const int* esrc;
  const float4* egeo;
  const float* h;  // [hidden unit k (65 rows)][h_kstride]
  size_t h_kstride;
  const float* x;  // [n_atoms][XS = 216]
  int n_pad, S, XS, n_atoms;
  const int2* tile_span;   // [n_tiles] {lo, hi} source atoms of the tile (hi - lo <= 62)
  const int2* tile_atoms;  // [n_tiles] {first destination atom, number of destination atoms (<= 32)}
  const int4* segs;        // [grid][max_segs][2]: {tile (-1 end), slab, k_begin, k_end}, {k_extra (-1 none), 0, 0, 0}
  int max_segs, nt0;
  int nks;                 // forming K-steps of 16 source rows: 4, or 3 when every tile's sources lie in the first 48 rows of its window
  // wm [k][124 blocks of 64 lanes x 8 halves] in consumption order, (hi, lo) pairs; B fragments of v_mfma_f32_32x32x16_f16 with the
  // K index of a step permuted to the accumulator layout of the forming MFMA: half p of lane (column c, hh) <-> input
  // 16 s2 + (p & 3) + 8 (p >> 2) + 4 hh of the wave's 32 channels:
  //   20 w + 2 (2 n + s2) + {0, 1}   wave w < 4: scalar channels 32 w .. 32 w + 31 -> scalar-output tile n;  w = 4: dot(x1, v) inputs
  //   100 + 8 m + 2 s2 + {0, 1}      wave 5 + m: x1 inputs -> vector rows;   104 + 8 m + 2 s2: cross inputs -> vector rows
  const float4* wm;
  const float* Tt;   // [k][32 w'][t_stride] TRANSPOSED pre-pass product (k_tprod_h): T_k[j][w'] at Tt[(k * 32 + w') * t_stride + j]
  int t_stride;      // even, >= n_atoms + 64
  int sB;            // (0: balanced weights, as DgArgs)
  const float *gx, *cf0, *cf1;  // as DgArgs
  int sC;            // coefficients are scaled by 2^sC: 3 max|h~| (an entry may sum three edges of one pair) below 2^14
  int sTw;           // 2^sTw x (largest column sum of |T weights|) < 1: T_k = x0 W is scaled by 2^(sX + sTw)
  float* partial0;   // [slab][n_pad][nt0*32]
  float* partial1;   // [slab][n_pad][3][32]
  int* err;          // device flag: bit 0 = more than three edges of one (source, destination) pair
  int x1;            // 1: the reduced-precision instantiation (f16x1: hi hi products only; jamun_tuning.f16x1)
};

// the matrix-formed conv for LARGE source spans (jamun_conv_ml.hip): two passes over the hidden units (vector channels, then scalar
// channels), block-sparse forming over the occupied 16-row source blocks; weight stream, T buffer, scales and epilogue of k_conv_mf
struct MlArgs {
  const int* deg;
  const int* epair;  // k_geom's pair table (MfArgs::epair)
  const int* esrc;
  const float4* egeo;
  const float* h;  // [hidden unit k (65 rows)][h_kstride]
  size_t h_kstride;
  const float* x;  // [n_atoms][XS = 216]
  int n_pad, S, XS, n_atoms;
  const int2* tile_span;   // [n_tiles] {lo, hi} source atoms of the tile (hi - (lo & ~1) <= window)
  const int2* tile_atoms;  // [n_tiles] {first destination atom, number of destination atoms (<= 32)}
  const int4* segs;        // [grid][max_segs][2], as MfArgs
  int max_segs, nt0;
  int window;              // source rows of the instantiation: 96, 128 or 168 (conv_ml_window)
  const float4* wm;        // MfArgs::wm
  const float* Tt;         // MfArgs::Tt (read as global fp32, scaled and split in registers)
  int t_stride;
  int sB;
  const float *gx, *cf0, *cf1;
  int sC, sTw;
  float* partial0;
  float* partial1;
  int* err;                // as MfArgs::err (bit 1: a source outside the window)
  unsigned long long* mfma_count;  // += v_mfma_f32_32x32x16_f16 executed (depends on the occupied source blocks of every tile), or null
  int x1;                  // as MfArgs::x1
};
// the initial projector on the same spans (k_conv_mlx in jamun_conv_ml.hip): MfxArgs + the window
struct MlxArgs {
  const int* deg;
  const int* epair;
  const int* esrc;
  const float4* egeo;
  const float* h;
  size_t h_kstride;
  int n_pad, S;
  const int2* tile_span;
  const int2* tile_atoms;
  const int4* segs;
  int max_segs, nt0;
  int window;  // as MlArgs
  const unsigned* xph;  // MfxArgs::xph / xpl (zero rows behind the batch: a window reads window / 2 pairs from the pair of its first atom)
  const unsigned* xpl;
  const float4* wx;     // MfxArgs::wx
  int sX, sC;
  const float *cf0, *cf1;
  float* partial0;
  float* partial1;
  int* err;
  unsigned long long* mfma_count;
};
int launch_conv_mlx(const MlxArgs& a, int grid, hipStream_t st);
int launch_conv_ml(const MlArgs& a, int grid, hipStream_t st);
int conv_ml_set_max_lds();
int conv_ml_window(int rows_needed);
void conv_ml_print_stamps();

// initial projector on the scheme of k_conv_mf (k_conv_mfi in jamun_conv_mf.hip): selector-formed coefficient sums per distinct
// embedding row, contracted with the input-times-weight table
struct MfiArgs {
  const int* deg;
  const int* epair;  // k_geom's pair table (MfArgs::epair)
  const int* esrc;
  const float4* egeo;
  const float* h;  // [hidden unit k (65 rows)][h_kstride]
  size_t h_kstride;
  int n_pad, S;
  const int2* tile_span;
  const int2* tile_atoms;
  const int4* segs;
  int max_segs, nt0;
  const int* atom_uid;  // [n_atoms] index of the atom's distinct embedding row (< 32 ut)
  int ut;               // tiles of 32 distinct embedding rows: 1, 2 or 4
  // tabw [k][6 roles][4 ut blocks of 64 lanes x 8 halves]: the table Tab_k[uid][w] scaled by 2^sB, split hi + lo, as B fragments with the
  // K index (uid) permuted as in MfArgs::wm: block 4 t + 2 s2 + {0 hi, 1 lo} of role r = 0..4 (scalar-output tile r, columns 32 r ..)
  // or r = 5 (the vector rows): uids 32 t + 16 s2 + ...
  const float4* tabw;
  int sB, sC;
  float* partial0;  // [slab][n_pad][nt0*32]
  float* partial1;  // [slab][n_pad][3][32]
  int* err;
};

// tail tiles of k_conv_mf (k_tail_form + k_tail_contract in jamun_conv_mf.hip): tiles with few destinations, formed with the hidden unit in
// the column index, parked, and contracted 32 gathered destinations at a time
#define TAIL_NFT 23  // formed tiles of 32 channels per (destination, hidden unit): 0..3 scalar channels, 4 dot, 5 + 2 m x1[m], 6 + 2 m cross[m],
                     // 11 + 3 w + m scalar channels 32 w .. times v_m (in place of the T pre-pass)
#define TAIL_NFT_INIT 8  // initial projector: formed tile 2 c + t (coefficient component c, channel tile t of the 64 embedding channels)
struct TailArgs {
  const int* deg;
  const int* epair;  // k_geom's pair table (MfArgs::epair)
  const int* esrc;
  const float4* egeo;
  const float* h;  // [hidden unit k][h_kstride]
  size_t h_kstride;
  const float* x;  // [n_atoms][XS = 216]
  int n_pad, S, XS, n_k, nt0;
  const int2* tile_span;   // the tile arrays of k_conv_mf
  const int2* tile_atoms;
  const int4* tail_tiles;  // [n_tail_tiles] {index of its first tail destination, first atom, atoms | source rows << 8, first source row}
  int n_tail_tiles, n_tail, n_runs;  // tail tiles, tail destinations, runs of hidden units (= partial slabs) of the contraction
  const int* tail_atom;    // [n_tail] atom of each tail destination
  float* tail_scale;       // [n_tail] 2^-(sX + sC) of the destination's forming (work buffer)
  float4* P;               // parked operands [n_k][tiles of 32 tail destinations][TAIL_NFT][2 K-steps][32 rows][2 lane halves][hi, lo] x 16 B (work buffer)
  const float* gx;         // channel factors of the staged rows (MfArgs::gx)
  int sC;
  const float4* wm;        // the weight stream of k_conv_mf (scalar outputs: blocks 20 w + 2 (2 n + s2) + {hi, lo})
  const float4* wmt;       // [k][24 blocks] vector outputs under one column scale: x1 (4 blocks: 2 s2 + {hi, lo}), cross (4), x0 (x) v of the
                           // four scalar channel tiles (8 + 4 w + 2 s2 + {hi, lo}); the same for the three planes
  const float *cf0, *cf1t; // inverse column scales: scalar outputs (as MfArgs::cf0), vector outputs of wmt
  // initial projector (k_tail_form_init, k_tail_contract<true>): the split embedding rows and weight stream of k_conv_mfx (MfxArgs), its
  // static row scale; cf0 / cf1t then hold MfxArgs::cf0 / cf1
  const unsigned* xph;
  const unsigned* xpl;
  const float4* wx;
  int sX;
  float* partial0;
  float* partial1;
  int* err;
};
int launch_conv_tail(const TailArgs& a, hipStream_t st);
int launch_conv_tail_init(const TailArgs& a, hipStream_t st);

// initial projector formed from the feature rows (k_conv_mfx in jamun_conv_mf.hip): batches with many distinct embedding rows
struct MfxArgs {
  const int* deg;
  const int* epair;  // k_geom's pair table (MfArgs::epair)
  const int* esrc;
  const float4* egeo;
  const float* h;  // [hidden unit k (65 rows)][h_kstride]
  size_t h_kstride;
  int n_pad, S;
  const int2* tile_span;
  const int2* tile_atoms;
  const int4* segs;
  int max_segs, nt0;
  // the noise-scaled atom embedding, padded to 64 channels, times the channel factors 2^e_u and 2^sX, split: word [atom pair p][channel]
  // = halves of atoms 2 p (low) and 2 p + 1 (high); hi plane and lo plane; zero rows behind the last atom (a window reads 32 pairs)
  const unsigned* xph;
  const unsigned* xpl;
  // wx [k][48 blocks of 64 lanes x 8 halves], (hi, lo) pairs, K index permuted as MfArgs::wm (half p of lane (column c, hh), K-step s2
  // <-> channel 32 t + 16 s2 + (p & 3) + 8 (p >> 2) + 4 hh):  20 t + 2 (2 n + s2) + {0, 1}: channel tile t -> scalar-output tile n;
  // 40 + 4 t + 2 s2 + {0, 1}: channel tile t -> the vector rows (the same block for the three planes)
  const float4* wx;
  int sX, sC;
  const float *cf0, *cf1;  // inverse column scales of the balanced weights, as MfArgs
  float* partial0;
  float* partial1;
  int* err;
};

// k_conv_init_v (jamun_conv_initv.hip): the initial projector edge by edge on the VALU, on the tiles / segments of k_conv_dg
struct InitVArgs {
  const int* deg;
  const int* esrc;
  const float4* egeo;
  const float* h;  // [hidden unit k (65 rows)][h_kstride]
  size_t h_kstride;
  int n_pad, S, PMAX, RS, nt0;
  int nbuf;  // row buffers in LDS: 2 (one barrier per k) or 1
  const int2* tile_span;
  const int2* tile_atoms;
  const int4* segs;
  int max_segs;
  const int* atom_uid;  // [n_atoms] index of the atom's distinct embedding row
  const float* tt2;     // [k][U][192]: scalar columns 0..127 | per lane u (column 128+u, vector column u)
  size_t tt2_kstride;   // U * 192
  int dbg;              // timing experiment (JAMUN_IV_DBG): 1 no edge loops (outputs zero)
  float* partial0;      // [slab][n_pad][nt0*32]
  float* partial1;      // [slab][n_pad][3][32]
};

// SeparableConv (jamun_sepconv.hip): per-edge depth-wise weights formed and consumed in registers, per-destination sums, point-wise Linear
struct SepArgs {
  const int* deg;
  const int* esrc;
  const float4* egeo;
  const float* h;  // [hidden unit k (65 rows)][h_kstride]
  size_t h_kstride;
  const float* x;  // [n_atoms][XS] input features (scalars first, then vectors [u][3])
  int n_atoms, S, XS;
  int n0, n1;       // scalar / vector input channels
  // W2~ (radial MLP's second layer with every constant folded in) as B fragments of v_mfma_f32_32x32x16_f16: [4 K-steps of 16 hidden
  // units][11 column tiles][hi, lo][64 lanes x 8 halves] (lane (c, hh): hidden units 16 s + 8 hh + j, column 32 ct + c); column tiles
  // A 0..3, B 4..7, C 8, D 9, E 10 (jamun_sepconv.hip); columns balanced by 2^sW_col: cfw [352] = 2^-sW_col, bias [352] = the bias row
  const float4* w2b;
  const float* cfw;
  const float* bias;
  int sH;            // h~ is scaled by 2^sH (static bound of the radial MLP's hidden activations -> [2^13, 2^14))
  float* D;          // [n_atoms][K0 + 3 K1] per-destination sums (work buffer): scalars [D0 | D3], per component m [D1 | D2 | D4]
  const float* wl0;  // [n0 + n1][G0]      point-wise Linear, scalar outputs (1 / sqrt(fan_in) folded)
  const float* wl1;  // [n0 + 2 n1][G1]    vector outputs
  int G0, G1, nt0, nt1;
  float* partial0;  // [n_pad][nt0 * 32]  (one slab)
  float* partial1;  // [n_pad][3][nt1 * 32]
};
const char* sep_conv_unsupported(int n0, int n1, int nt0, int nt1, int S, int edge_attr_dim);
int sep_conv_set_max_lds();
int launch_sep_conv(const SepArgs& a, int cus, hipStream_t st);

struct NodeArgs {
  const float* partial0;  // [n_slices][n_pad][nt0*32]
  const float* partial1;  // [n_slices][n_pad][3][nt1*32]
  const int* deg;
  const float* x_in;  // [n_atoms][XSin]  (x_old for hidden layers, scaled embedding for the initial projector)
  float* x_out;       // [n_atoms][mul0+3*mul1]
  // [W_self ; W_skip] concatenated along K, padded to K0p / K1p rows, as MFMA B fragments: wcat0 [nt][K0p/8][64 lanes]
  // float4 (lane (c, hh): rows 2 (4 sg + st) + hh, column 32 nt + c), wcat1 [K1p/8][64]
  const float4* wcat0;
  const float4* wcat1;
  int K0p, K1p;
  // f16x3 node update (jamun_node.hip): the same matrices scaled by 2^sW0 / 2^sW1, split hi + lo, as B fragments of
  // v_mfma_f32_32x32x16_f16: wh0 [nt][K0h/16][hi, lo][64 lanes] (lane (c, hh): rows 16 s + 8 hh + p, column 32 nt + c), wh1 [K1h/16][hi, lo][64]
  const float4* wh0;
  const float4* wh1;
  int K0h, K1h;
  // ... balanced by exact powers of two on the host: row K of [W_self ; W_skip] times 2^-e_K (its largest magnitude -> [0.5, 1)), then
  // column w times 2^sW_w (largest -> [2^13, 2^14)).  The kernel multiplies the inputs by 2^e_K when it stages them — kga0 [mul0]
  // (activated scalars), kga1 [mul1] (gated vectors), kgx [XSin] (the x_in channels, in x_in's layout) — and the output columns by
  // cg0 [mul0] / cg1 [mul1] = 2^-sW_w.  A per-tensor scale would leave rows / columns 2^-14 below the largest weight with denormal
  // lo halves (trained checkpoints spread their channels over many octaves).
  const float *kga0, *kga1, *kgx, *cg0, *cg1;
  const float* mix;      // [mul0+mul1] or nullptr (initial projector)
  float cL, cS;
  int n_atoms, n_pad, n_slices, nt0, nt1;
  int mul0, mul1, in0, in1, XSin;
  const int* atom_nslab;  // [n_atoms] partial slabs to sum for the atom (fused conv), or nullptr: n_slices everywhere
  int max_slabs;          // upper bound of the above
};

struct HeadArgs {
  const float* x;       // [n_atoms][mul0+3*mul1]
  const float* w_gate;  // [mul0][mul1]   (gate pre-activations: columns mul0.. of the 0e block of Linear(hidden->gate_in))
  const float* w_vec;   // [mul1][mul1]
  const float* w_out;   // [mul1]         (includes 1/sqrt(mul1) and output_gain)
  float* g;             // [n_atoms][3]
  float cS;
  int n_atoms, mul0, mul1;
};

struct LangevinConsts {
  float u_half_delta;  // (float)(u * (delta/2))
  float half_delta;    // (float)(delta/2)
  float exp_mg;        // (float)exp(-friction)
  float zeta_sqrt_u;   // (float)(sqrt(1-exp(-2 friction)) * sqrt(u))
  float beta;          // inverse_temperature
  float clip;
  int has_clip;
};

// BAOAB halves fused into the ends of the forward (k_geom / k_finalize); a null `v` / `psi_out` switches the half off
struct LangevinPre {
  float* v = nullptr;             // in: v, out: vhat
  const float* psi = nullptr;
  const float* noise = nullptr;   // this iteration's [n,3] draws, or null: Philox keyed by (seed, iter, atom)
  uint64_t seed = 0;
  uint32_t iter = 0;
  LangevinConsts k{};
};
struct LangevinPost {
  float* psi_out = nullptr;
  float* v = nullptr;
  int update_v = 0;
  float *y_frame = nullptr, *score_frame = nullptr, *xhat_frame = nullptr;
  LangevinConsts k{};
};

// launchers implemented in jamun_kernels.hip
void launch_mean_center(const float* pos, const int* ptr, int n_graphs, float* out, hipStream_t st);
void launch_radius_graph(const float* pos, const int* ptr, int n_graphs, float r2, int stride, int* nbr, int* deg,
                         hipStream_t st);
void launch_finalize_geom(float* y, float* yc, const float* g, const int* ptr, int n_graphs, float c_skip, float c_out, float sigma2, int mean_center,
                          float* tmp, float* xhat, float* score, const LangevinPost& post, float c_in, float r2, int S, const int* bip, const int* bis,
                          int* deg, int* esrc, float4* egeo, int* epair, const LangevinPre& pre, hipStream_t st);
void launch_geom(float* y, const int* ptr, int n_graphs, float c_in, float r2, int S, const int* bip,
                 const int* bis, int mean_center, float* yc, int* deg, int* esrc, float4* egeo, int* epair, const LangevinPre& pre,
                 hipStream_t st);
// (w1h / isc_all non-null: the f16x3 kernel k_edge_h16 — W1's radial part as scaled hi + lo A fragments [layer][2][2][2][64], 2^-(14 + sW) per layer)
void launch_edge_h(const int* deg, const int* esrc, const float4* egeo, int n_atoms, int S, const float* w1r_all,
                   const float* cmask_all, int n_layers, const float* mu, float step, float* h_all, size_t h_layer_stride,
                   size_t h_kstride, hipStream_t st, const float4* w1h = nullptr, const float* isc_all = nullptr);
int launch_conv(const ConvArgs& a, int rc, int nt, hipStream_t st);
int conv_set_max_lds();
int launch_conv_dg(const DgArgs& a, int grid, hipStream_t st);
int conv_dg_set_max_lds();
void conv_dg_print_stamps();
void node_print_stamps();  // (-DNH_TRACE builds: per-wave timeline of k_node_update_h)
void conv_initv_print_stamps();
void conv_mf_print_stamps();
size_t conv_dg_lds_bytes(int rs, int pmax, int mode, int emu);
void launch_tprod(const float* x, int XS, int n_atoms, int n_k, const float4* wt, const float4* wth, const float* gT, const float* cfT, float* T, int t_stride,
                  hipStream_t st, bool no_tprod_t = false);  // (t_stride > 0 and f16x3 weights: k_tprod_t unless no_tprod_t)
int launch_conv_mf(const MfArgs& a, int grid, hipStream_t st);
int launch_conv_mfi(const MfiArgs& a, int grid, hipStream_t st);
int launch_conv_mfx(const MfxArgs& a, int grid, hipStream_t st);
int conv_mf_set_max_lds();
size_t conv_mf_lds_bytes();
int launch_conv_initv(const InitVArgs& a, int grid, hipStream_t st);
int conv_initv_set_max_lds();
size_t conv_initv_lds_bytes(int rs, int pmax, int nbuf);
void launch_node_update(const NodeArgs& a, hipStream_t st);
void launch_node_update_h(const NodeArgs& a, int cus, hipStream_t st);
bool node_update_h_supported(const NodeArgs& a);
size_t node_update_lds_bytes(const NodeArgs& a);
int node_update_set_max_lds();
void launch_head(const HeadArgs& a, hipStream_t st);
void launch_finalize(const float* y, const float* yc, const float* g, const int* ptr, int n_graphs, float c_skip,
                     float c_out, float sigma2, int mean_center, float* tmp, float* xhat, float* score,
                     const LangevinPost& post, hipStream_t st);
void launch_baoab_pre(float* y, float* v, const float* psi, const float* noise, uint64_t seed, uint32_t iter, int n,
                      const LangevinConsts& k, hipStream_t st);
void launch_baoab_post(float* v, float* psi, const float* score, const float* y, const float* xhat, int n,
                       const LangevinConsts& k, int update_v, float* yf, float* sf, float* xf, hipStream_t st);
void launch_aboba_a(float* y, const float* v, int n, float half_delta, hipStream_t st);
void launch_aboba_b(float* y, float* v, const float* score, const float* noise, uint64_t seed, uint32_t iter, int n,
                    const LangevinConsts& k, float* yf, float* sf, hipStream_t st);
void launch_copy(const float* src, float* dst, int n, hipStream_t st);
void launch_deg_to_float(const int* deg, float* out, int n, hipStream_t st);
void launch_count_edges(const int* deg, int n, unsigned long long* out, hipStream_t st);
void launch_scatter_mean(const float* src, const int* seg_ptr, int n_out, int width, float* out, hipStream_t st);
void launch_edge_geometry(const float* pos, const long long* src, const long long* dst, int n_edges, int n_atoms, float cutoff, int n_basis,
                          float* sh, float* radial, hipStream_t st);
void launch_philox_normal(float* out, int n, uint64_t seed, uint32_t iter, uint32_t atom0, hipStream_t st);
int launch_node_linear(const float* x, int n_atoms, int in0, int in1, int out0, int out1, const float* w, float* out, hipStream_t st);
