/*
 * jamun_hip.h — C ABI of the MI355X-native JAMUN walk-jump sampling path.
 *
 * The reference (prescient-design/jamun) is pure Python and has no FFI of its own; its
 * extension points for this path are duck-typed Python protocols.  This header is the C ABI
 * that sits UNDER those protocols: each entry point names the reference interface it
 * replaces (paths relative to /root/reference).  Plain pointers and sizes only — no torch
 * types.  All device pointers are HIP device memory owned by the caller; all work is
 * enqueued on the `hipStream_t` passed in (as `void*`), with no internal synchronisation
 * unless stated.  Every function returns 0 on success or a negative error code; the message
 * is available from jamun_last_error() (thread-local).
 *
 * Build: hipcc --offload-arch=gfx950 -shared -fPIC (see jamun_amd/csrc/build.py).
 */
#ifndef JAMUN_HIP_H
#define JAMUN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JAMUN_OK 0
#define JAMUN_ERR_INVALID -1   /* bad argument / unsupported configuration */
#define JAMUN_ERR_MISSING -2   /* a required checkpoint tensor is missing  */
#define JAMUN_ERR_HIP -3       /* a HIP runtime call failed                */

/* A named host tensor in the reference's state-dict layout (fp32, contiguous).
 * Names are the reference's parameter names WITHOUT the "g." / "g._orig_mod." prefix,
 * e.g. "layers.3.gated_conv.f.f.radial_nn.3.weight"  (src/jamun/model/arch/e3conv.py:15-85,
 * src/jamun/e3tools/nn/_conv.py:76-92). */
typedef struct jamun_tensor {
  const char* name;
  const float* data; /* host pointer */
  int64_t numel;
} jamun_tensor;

/* Architecture + denoiser hyper-parameters
 * (src/jamun/hydra_config/model/arch/e3conv.yaml:3-14, src/jamun/model/denoiser.py:16-33). */
typedef struct jamun_hparams {
  int32_t n_layers;               /* hidden ConvBlocks (5)                       */
  int32_t mul0, mul1;             /* irreps_hidden = mul0 x0e + mul1 x1e (120,32) */
  int32_t edge_attr_dim;          /* 64: 32 bonded-embedding + 32 radial          */
  int32_t emb_dim[4];             /* atom_type, atom_code, residue_code, residue_index (8,8,32,8) */
  int32_t emb_rows[4];            /* embedding table rows (20,10,25,10)           */
  int32_t use_residue_sequence_index; /* 0: index zeroed (atom_embedding.py:69-71) */
  int32_t mean_center;            /* Denoiser.mean_center                         */
  float max_radius;               /* Denoiser.max_radius                          */
  float average_squared_distance; /* Denoiser.average_squared_distance            */
  float act_scalar_const;         /* e3nn normalize2mom(LeakyReLU(0.01))  = 1.4162684 */
  float act_gate_const;           /* e3nn normalize2mom(sigmoid)          = 1.8467055 */
  float w3j_111_sign;             /* +1: wigner_3j(1,1,1) = +eps/sqrt(6) (e3nn 0.5.4)   */
  int32_t separable;              /* hidden_layer_factory.conv: 0 jamun.e3tools.nn.Conv (e3conv.yaml), 1 SeparableConv
                                     (e3conv_separable.yaml; src/jamun/e3tools/nn/_conv.py:122-135, _tensor_product.py:8-58):
                                     expects <block>.gated_conv.f.f.tp.lin.weight and a radial_nn.3 of 2 n0 + 3 n1 rows */
} jamun_hparams;

/* Static description of the walker batch — what torch_geometric's Batch.from_data_list gives the
 * reference (src/jamun/cmdline/sample.py:27-38, src/jamun/utils/data_with_residue_info.py:17-33).
 * All host pointers.  bonds are directed pairs (src,dst) already offset into the batch. */
typedef struct jamun_topology {
  int32_t n_atoms;
  int32_t n_graphs;
  const int32_t* ptr;                    /* [n_graphs+1] first atom of each walker */
  const int32_t* atom_type_index;        /* [n_atoms] */
  const int32_t* atom_code_index;        /* [n_atoms] */
  const int32_t* residue_code_index;     /* [n_atoms] */
  const int32_t* residue_sequence_index; /* [n_atoms] */
  int32_t n_bonds;
  const int64_t* bond_src;               /* [n_bonds] edge_index[0] of the dataset's bonded edges */
  const int64_t* bond_dst;               /* [n_bonds] edge_index[1]                               */
} jamun_topology;

/* Langevin splitting-integrator parameters — the fields of the BAOAB/ABOBA dataclasses
 * (src/jamun/sampling/mcmc/_splitting.py:11-58) consumed by
 * src/jamun/sampling/mcmc/functional/_splitting.py:44-178. */
typedef struct jamun_mcmc_params {
  int32_t steps;              /* range(1, steps): steps-1 integrator iterations     */
  int32_t save_every_n_steps; /* >= 1                                               */
  int32_t burn_in_steps;
  int32_t has_clip;           /* 0: score_fn_clip = None                            */
  float delta, friction, M, inverse_temperature, score_fn_clip;
} jamun_mcmc_params;

/* Kernel-selection switches of jamun_sampler_create.  NOT needed in production: NULL (or all zero) selects the default kernels, and the
 * library reads no environment variables — these exist so that tests can run every implementation of a block against the others
 * (tests/test_gpu_parity.py) and profiling scripts can time an alternative.  Each field only EXCLUDES a kernel; what then runs is
 * reported by jamun_sampler_stats (conv_path / dg_mode / init_path / dg_emu). */
typedef struct jamun_tuning {
  int32_t no_dg;        /* hidden layers and initial projector on the general kernel k_conv (no destination-grouped tile plan)     */
  int32_t no_mf;        /* hidden layers: not k_conv_mf (A operand formed on the matrix cores); k_conv_dg (vector-ALU forming)      */
  int32_t dg_fp32;      /* k_conv_dg with v_mfma_f32_32x32x2_f32 instead of the f16x3 scheme (implies no_mf)                        */
  int32_t dg_no_alt;    /* k_conv_dg: not mode 1 (two passes over the hidden units for molecules above ~80 atoms)                   */
  int32_t dg_no_sp;     /* k_conv_dg: not mode 2 (single phase, spans up to ~52 atoms)                                              */
  int32_t dg_no_sph;    /* k_conv_dg: not mode 3 (single phase with one Y tile, spans up to ~73 atoms)                              */
  int32_t no_mfi;       /* initial projector: not k_conv_mfi / k_conv_mfx (matrix-core forming)                                      */
  int32_t no_init_v;    /* initial projector: not k_conv_init_v (edge by edge on the vector ALUs)                                    */
  int32_t node_fp32;    /* node update with v_mfma_f32_32x32x2_f32 (k_node_update) instead of f16x3 (k_node_update_h)                */
  int32_t edge_h_fp32;  /* radial MLP first layer with fp32 MFMAs (k_edge_h) instead of f16x3 (k_edge_h16)                           */
  int32_t dg_kgroups;   /* hidden-unit slices over XCD groups for the destination-grouped kernels: 0 (default = 1), 1, 2, 4, 8       */
  int32_t no_tail;      /* k_conv_mf: tiles with few destinations stay whole tiles (no k_tail_form / k_tail_contract)                 */
  int32_t no_short_k;   /* k_conv_mf: always four forming K-steps (64 source rows), also when every tile's sources fit the first 48      */
  int32_t no_ml;        /* hidden layers: not k_conv_ml (matrix-core forming for source spans of 63..167 atoms); k_conv_dg there              */
  int32_t seg_cost_tenths; /* work lists of the destination-grouped kernels: cost of a segment's prologue + epilogue in tenths of a (tile, hidden unit)
                              item when the lists are cut (0: the kernel's measured default — k_conv_mf 3.6, k_conv_ml 5.8 items —, -1: none)        */
  int32_t f16x1;        /* OPT-IN reduced precision of the hidden-layer conv (k_conv_mf / k_conv_ml): each fp32 product as ONE f16 MFMA (operands
                           rounded to 11 bits, fp32 accumulation) instead of the three of the f16x3 scheme; state, integrator, radial MLPs, initial
                           projector, node update and head stay as they are.  Never the default; x-hat then sits 2.5e-5 .. 7.6e-5 nm from the fp32
                           path on the test batches (asserted <= 1e-3 nm; the level of the reference's TF32 GPU path).  jamun_stats.dg_emu reports 2.  Ignored by the other conv kernels.   */
  int32_t no_fuse_geom; /* walks: k_finalize of an iteration and k_geom of the next as two launches instead of one (k_finalize_geom, round 6); A/B aid */
  int32_t selfcheck;    /* create-time self-check of the selected kernels against the general ones (one forward on synthetic positions through both;
                           node features after every block within 2e-5, else jamun_sampler_create returns JAMUN_ERR_INVALID): 0 / 1 on (default),
                           -1 off, 2 on with an injected fault in the selected conv kernel's weight stream (tests: the check must fire)   */
  int32_t no_tprod_t;   /* T pre-pass of k_conv_mf / k_conv_ml with k_tprod_h instead of k_tprod_t (round 6: weights through LDS, rows fetched whole,
                           transposed T stored as whole lines; bit-identical T); A/B aid                                                   */
} jamun_tuning;

typedef struct jamun_model jamun_model;     /* raw checkpoint tensors kept on the host          */
typedef struct jamun_sampler jamun_sampler; /* sigma- and topology-specific device state        */

const char* jamun_last_error(void);
int jamun_version(void);

/* Replaces Denoiser.load_from_checkpoint's module construction + load_state_dict
 * (src/jamun/hydra_config/model/denoiser_pretrained.yaml:1-2, src/jamun/model/denoiser.py:16-42). */
int jamun_model_create(const jamun_hparams* hp, const jamun_tensor* tensors, int32_t n_tensors, jamun_model** out);
void jamun_model_destroy(jamun_model* m);

/* Binds a model to one noise level and one walker batch: folds the noise-conditioning constants
 * (src/jamun/model/noise_conditioning.py:50-73), path normalisations and Clebsch-Gordan factors into
 * MFMA-ordered packed weights, uploads them, and allocates every work buffer.  Replaces
 * ModelSamplingWrapper.__init__ + the per-call graph clone (src/jamun/utils/sampling_wrapper.py:12-47).
 * Synchronous (uploads weights). */
int jamun_sampler_create(const jamun_model* m, float sigma, const jamun_topology* topo, const jamun_tuning* tuning /* NULL: defaults */,
                         jamun_sampler** out);
void jamun_sampler_destroy(jamun_sampler* s);

/* Denoiser.xhat (src/jamun/model/denoiser.py:203-217): y_dev [n_atoms,3] -> xhat_dev [n_atoms,3]. */
int jamun_xhat(jamun_sampler* s, const float* y_dev, float* xhat_dev, void* stream);
/* Denoiser.score (src/jamun/model/denoiser.py:111-114). */
int jamun_score(jamun_sampler* s, const float* y_dev, float* score_dev, void* stream);

/* baoab() (src/jamun/sampling/mcmc/functional/_splitting.py:112-178) with the model's score, fused with the
 * jump of SingleMeasurementSampler.walk_jump (src/jamun/sampling/walkjump/_single_measurement.py:49-79).
 *   y_dev, v_dev  [n_atoms,3]  in: y_init, v_init (already initialised) ; out: final y, v
 *   noise_dev     [steps-1, n_atoms, 3] standard-normal draws, one per iteration (reference RNG call order),
 *                 or NULL to draw them in-kernel from Philox4x32-10 keyed by (seed, iteration, atom)
 *   y_traj_dev, score_traj_dev, xhat_traj_dev  [T, n_atoms, 3] or NULL (save_trajectory=False);
 *                 T = jamun_num_frames(params, 0).  xhat_traj[t] = y_traj[t] + sigma^2 * score_traj[t]
 *                 (identical to the reference's extra forward per frame, in exact arithmetic).
 *                 With y_traj_dev == NULL (save_trajectory=False) a non-NULL score_traj_dev receives exactly ONE
 *                 frame, the initial score — the reference appends later scores only together with y
 *                 (_splitting.py:155,168-170) — so a [1, n_atoms, 3] buffer is enough and nothing past it is written.
 *   xhat_dev     [n_atoms,3] xhat(y_final) or NULL
 * No host synchronisation; all steps are enqueued on `stream`. */
int jamun_walk_baoab(jamun_sampler* s, float* y_dev, float* v_dev, const jamun_mcmc_params* p, const float* noise_dev,
                     uint64_t seed, float* y_traj_dev, float* score_traj_dev, float* xhat_traj_dev, float* xhat_dev,
                     void* stream);
/* aboba() (src/jamun/sampling/mcmc/functional/_splitting.py:44-109).  score_traj has T-1 frames (scores at the
 * half-step positions, :90,99-101); xhat_traj (if requested) costs one extra forward per saved frame, as the
 * reference (the half-step score cannot be reused). */
int jamun_walk_aboba(jamun_sampler* s, float* y_dev, float* v_dev, const jamun_mcmc_params* p, const float* noise_dev,
                     uint64_t seed, float* y_traj_dev, float* score_traj_dev, float* xhat_traj_dev, float* xhat_dev,
                     void* stream);
/* Number of saved y frames for (steps, save_every_n_steps, burn_in_steps); aboba's score_traj has one fewer
 * when burn_in_steps == 0.  */
int jamun_num_frames(const jamun_mcmc_params* p, int32_t* n_y_frames, int32_t* n_score_frames_baoab,
                     int32_t* n_score_frames_aboba);

/* ---- stand-alone operators of the path (also used for unit-level parity tests) -------------------------- */

/* mean_center (src/jamun/utils/mean_center.py:7-12): out = pos - centroid(graph).  ptr_dev [n_graphs+1] int32. */
int jamun_mean_center(const float* pos_dev, const int32_t* ptr_dev, int32_t n_graphs, float* out_dev, void* stream);

/* torch_geometric.nn.radius_graph(pos, r, batch) with torch_cluster CUDA semantics, call site
 * src/jamun/model/denoiser.py:149: per centre, scan its graph in index order, keep while d^2 < r^2 until 33 hits,
 * drop self.  Output is a fixed-stride neighbour table: nbr_dev [n_atoms, stride] (src indices, ascending),
 * deg_dev [n_atoms].  stride >= 33. */
int jamun_radius_graph(const float* pos_dev, const int32_t* ptr_dev, int32_t n_graphs, int32_t n_atoms, float r,
                       int32_t stride, int32_t* nbr_dev, int32_t* deg_dev, void* stream);

/* torch_scatter.scatter(src, index, dim=0, dim_size, reduce="mean"), call site src/jamun/e3tools/nn/_conv.py:117,
 * for a destination-sorted index given as CSR: rows seg_ptr[d]..seg_ptr[d+1] of src are averaged into out[d]
 * (empty segment -> 0).  Fixed summation order (row order), no atomics.  src_dev [E, width], out_dev [n_out, width]. */
int jamun_scatter_mean(const float* src_dev, const int32_t* seg_ptr_dev, int32_t n_out, int32_t width, float* out_dev,
                       void* stream);

/* One BAOAB iteration's state update around the score evaluation
 * (src/jamun/sampling/mcmc/functional/_splitting.py:158-166):
 *   pre : v += u*(delta/2)*psi ; y += (delta/2)*v ; vhat = exp(-gamma)*v + zeta*sqrt(u)*R ; y += (delta/2)*vhat
 *   post: psi = clip(score)*beta ; v = vhat + (delta/2)*psi          (no u, as the reference :166)
 * n = number of atoms; arrays are [n,3]. */
int jamun_baoab_pre(float* y_dev, float* v_dev, const float* psi_dev, const float* noise_dev, int32_t n,
                    const jamun_mcmc_params* p, void* stream);
int jamun_baoab_post(float* v_dev, float* psi_dev, const float* score_dev, int32_t n, const jamun_mcmc_params* p,
                     void* stream);

/* One ABOBA iteration's state update around the score evaluation
 * (src/jamun/sampling/mcmc/functional/_splitting.py:86-97):
 *   a : y += (delta/2)*v                                              (:87, the score is then evaluated at this y)
 *   b : psi = clip(score)*beta ; v += u*(delta/2)*psi ; vhat = exp(-gamma)*v + zeta*sqrt(u)*R ;
 *       v = vhat + (delta/2)*psi (no u, :96) ; y += (delta/2)*v
 * n = number of atoms; arrays are [n,3]. */
int jamun_aboba_a(float* y_dev, const float* v_dev, int32_t n, const jamun_mcmc_params* p, void* stream);
int jamun_aboba_b(float* y_dev, float* v_dev, const float* score_dev, const float* noise_dev, int32_t n,
                  const jamun_mcmc_params* p, void* stream);

/* Edge geometry of E3Conv.forward (src/jamun/model/arch/e3conv.py:114-123) for an explicit edge list: per edge e = (src -> dst),
 * from positions ALREADY scaled by c_in (as E3Conv receives them, src/jamun/model/denoiser.py:198):
 *   sh_dev     [n_edges, 4]        o3.SphericalHarmonics("1x0e+1x1e", normalize=True, normalization="component") = [1, sqrt(3) v/|v|]
 *   radial_dev [n_edges, n_basis]  e3nn soft_one_hot_linspace(|v|, 0, radial_cutoff, n_basis, basis="gaussian", cutoff=True)
 * (the radial half of edge_attr; the bonded half is an embedding lookup).  src_dev / dst_dev: int64 device arrays (edge_index);
 * pos_dev is [n_atoms, 3]: an edge with an index outside [0, n_atoms) reads nothing and gets NaN outputs. */
int jamun_edge_geometry(const float* pos_dev, int32_t n_atoms, const int64_t* src_dev, const int64_t* dst_dev, int32_t n_edges,
                        float radial_cutoff, int32_t n_basis, float* sh_dev, float* radial_dev, void* stream);

/* e3nn o3.Linear on node features between irreps (in0 x0e + in1 x1e) and (out0 x0e + out1 x1e): the skip / self-interaction /
 * head Linears of the path (src/jamun/e3tools/nn/_interaction.py:23-24, _mlp.py:69,109).  w_dev: the flat e3nn weight
 * [in0 x out0 | in1 x out1] on the device (instruction order i_in outer, i_out inner; path normalisation 1/sqrt(fan_in) applied
 * here).  x_dev [n_atoms, in0 + 3 in1] -> out_dev [n_atoms, out0 + 3 out1], e3nn layout (channel-major, m fastest). */
int jamun_node_linear(const float* x_dev, int32_t n_atoms, int32_t in0, int32_t in1, int32_t out0, int32_t out1, const float* w_dev,
                      int64_t w_numel, float* out_dev, void* stream);

/* The noise of the walks when noise_dev == NULL — the in-kernel replacement of the reference's per-step torch.randn_like
 * (src/jamun/sampling/mcmc/functional/_splitting.py:161; iid N(0,1) per atom and component): out_dev [n, 3] = the draws R of
 * integrator iteration `iteration` (1 .. steps-1) for atoms first_atom .. first_atom + n - 1 of a rank's batch under `seed`.
 * Philox4x32-10 (Salmon et al., SC'11) with key = seed (low, high word), counter = (atom, iteration, 0x4a414d55, 0); the four
 * output words w0..w3 give u_i = (float(w_i) + 0.5) * 2^-32, R = (sqrt(-2 ln u0) cos 2 pi u1, sqrt(-2 ln u0) sin 2 pi u1,
 * sqrt(-2 ln u2) cos 2 pi u3) (Box-Muller).  jamun_walk_baoab / jamun_walk_aboba with noise_dev == NULL are bit-identical to the
 * same walk fed these draws as noise_dev. */
int jamun_philox_normal(float* out_dev, int32_t n, uint64_t seed, uint32_t iteration, uint32_t first_atom, void* stream);

/* The graph half of a forward on its own: Denoiser.add_edges + the edge geometry + the radial MLPs' hidden layer
 * (src/jamun/model/denoiser.py:138-166, arch/e3conv.py:110-127, e3tools/nn/_conv.py:112) for positions y_dev [n_atoms,3]; the
 * result stays inside the sampler as the edge table the blocks below run on. */
int jamun_build_edges(jamun_sampler* s, const float* y_dev, void* stream);

/* ONE block of E3Conv on caller-owned node features and the sampler's current edge table (jamun_build_edges, or the last forward):
 *   layer 0      x_out = ConvBlock_initial(noise-scaled atom embedding)                     x_in_dev must be NULL (the input is the
 *                                                                                           sampler's own constant embedding)
 *   layer l >= 1 x_out = w_l * x_in + (1 - w_l) * ConvBlock_l(s_l * x_in)                   x_in_dev [n_atoms, mul0 + 3 mul1]
 * i.e. ConvBlock = LinearSelfInteraction(Gated(Conv)) (src/jamun/e3tools/nn/_conv.py:147-221: gather, radial-MLP weights, tensor
 * product, scatter-mean, gate, self-interaction + skip Linear) with the noise-conditional scaling and skip of
 * src/jamun/model/arch/e3conv.py:129-133 around it.  x_out_dev [n_atoms, mul0 + 3 mul1]; must not alias x_in_dev. */
int jamun_conv_block(jamun_sampler* s, int32_t layer, const float* x_in_dev, float* x_out_dev, void* stream);

/* Introspection for tests / benchmarks. */
typedef struct jamun_stats {
  int64_t n_edges;        /* directed edges (radial + bonded) in the last forward             */
  int64_t flop_ref_assoc; /* FLOPs of one forward in the reference association (SURVEY §8 d)  */
  int64_t flop_executed;  /* FLOPs actually issued on MFMA/VALU by this implementation         */
  int64_t conv_k0, conv_k1; /* padded contraction depth of the scalar / vector conv GEMMs (hidden layer) */
  int64_t conv0_flop_alg;   /* useful FLOPs of ONE hidden-layer scalar-row conv launch: 2*n_atoms*65*(mul0+mul1)*(mul0+mul1) */
  int64_t conv1_flop_alg;   /* useful FLOPs of ONE hidden-layer vector-row conv launch: 2*3*n_atoms*65*(mul0+2*mul1)*mul1   */
  int32_t edge_stride;
  int32_t n_slices;       /* partial slabs per tile summed by the node update (max over tiles)          */
  int32_t conv_path;      /* hidden layers: 2 destination-grouped kernels on host-planned tiles (jamun_conv_mf.hip / jamun_conv_dg.hip),
                             0 general k_conv (any irreps, any topology; also SeparableConv's slot in this field) */
  int32_t dg_mode;        /* destination-grouped kernel: 5 jamun_conv_ml.hip (as 4, for source spans of 63..167 atoms: two passes over the hidden
                             units with the vector / scalar channels resident, block-sparse forming over the occupied 16-row source blocks);
                             4 jamun_conv_mf.hip (A operand formed on the matrix cores and chained into the
                             contraction; source spans up to 62 atoms); jamun_conv_dg.hip (A operand formed edge by edge on the vector ALUs):
                             0 two phases per hidden unit with resident source rows, 1 two passes over the hidden units (molecules above
                             ~80 atoms), 2 single phase (spans up to ~52 atoms), 3 single phase with one Y tile (spans up to ~73 atoms);
                             -1: not in use */
  int32_t init_path;      /* initial projector: 5 k_conv_mlx (jamun_conv_ml.hip: as 4 on the dg_mode 5 tiles — source spans of 63..167 atoms, block-sparse forming),
                             4 k_conv_mfx (jamun_conv_mf.hip: aggregated operand formed on the matrix cores from the embedding
                             rows, as a hidden layer with 64 scalar channels; dg_mode 4 tiles, any number of distinct embedding rows),
                             3 k_conv_mfi (coefficient sums per distinct embedding row formed with a one-hot selector, contracted with the
                             input-times-weight table; dg_mode 4 tiles, <= 32 distinct rows),
                             2 edge-by-edge VALU kernel on the tiles of jamun_conv_dg.hip (jamun_conv_initv.hip),
                             0 the general kernel k_conv  (1 was the table kernel jamun_conv_init.hip, retired in round 4) */
  int32_t dg_row_blocks;  /* jamun_conv_dg.hip: 1 when some molecule exceeds the span budget and its sources are cut into row blocks */
  int32_t dg_emu;         /* hidden-layer conv arithmetic: 2 f16x1 (jamun_tuning.f16x1: one f16 MFMA per product; k_conv_mf / k_conv_ml only), 1 f16x3 (three v_mfma_f32_*_f16 per fp32 product, operands split hi + lo),
                             0 v_mfma_f32_32x32x2_f32 (jamun_tuning.dg_fp32); -1: not in use */
  int64_t conv_flop_exec_launch; /* matrix-core FLOPs EXECUTED by ONE launch of the hidden-layer conv kernel (k_conv_mf / k_conv_dg; padding,
                             structural zeros of the forming GEMMs and the three products of the f16x3 scheme included; the T pre-pass is
                             a separate launch and not counted); 0 when another conv path is in use */
  int64_t conv_flop_useful_launch; /* of those, what the destination-grouped association needs at the edge count of the last forward:
                             3 (f16x3) x [2 x 65 x n_atoms x (G0 K0 + 3 G1 K1) contraction + 2 x 65 x n_edges x (in0 + 15 in1) forming
                             (x0, dot, x1, cross and T term per edge)], no padding, no zero blocks */
  int32_t n_tail_tiles;   /* tiles of the dg_mode-4 plan that go through k_tail_form / k_tail_contract instead of k_conv_mf (0: none)            */
  int32_t n_tail;         /* ... and their destinations                                                                                   */
  int64_t conv_bytes_alg_launch;  /* algorithmic HBM bytes of that launch: h~ of the layer, T, the weight stream once, the feature
                             rows once, the partial slabs written */
  int32_t mf_nks;         /* k_conv_mf: forming K-steps of 16 source rows per product (4; 3 when every tile's sources fit 48 rows; 0: other kernel) */
  int32_t ml_window;      /* k_conv_ml (dg_mode 5): source rows of the instantiation in use (96, 128 or 168); 0: other kernel */
} jamun_stats;
/* Synchronises `stream`. */
int jamun_sampler_stats(jamun_sampler* s, jamun_stats* out, void* stream);
/* Synchronises `stream` and returns JAMUN_ERR_INVALID if a matrix-formed conv kernel flagged, in any forward enqueued so far, an edge table
 * its coefficient tiles cannot represent (more than three edges of one ordered pair; a source outside the tile's window: host plan and kernel
 * disagree).  The flag also surfaces, unsynchronised, at the next entry point after its copy landed; call this where the results are about to
 * be consumed (jamun_amd.sampling.Sampler does, once per batch). */
int jamun_sampler_check(jamun_sampler* s, void* stream);

/* Per-kernel-class timing with HIP events recorded on the launch stream around each launch of the forward
 * (bench.py's roofline leg).  enable(1) starts collecting; read() synchronises `stream`, returns the summed
 * elapsed milliseconds and launch count per class (arrays of JAMUN_PROF_NCLASS) and clears the collection. */
#define JAMUN_PROF_GEOM 0        /* centring + radius graph + edge geometry                         */
#define JAMUN_PROF_EDGE_H 1      /* radial-MLP hidden layer per edge                                */
#define JAMUN_PROF_CONV0_INIT 2  /* conv contraction, initial projector (fused path: all rows)        */
#define JAMUN_PROF_CONV1_INIT 3  /* conv contraction, vector-output rows, initial projector          */
#define JAMUN_PROF_CONV0 4       /* conv contraction, hidden layers (dominant; fused path: all rows)  */
#define JAMUN_PROF_CONV1 5       /* hidden layers: tail tiles (k_tail_form + k_tail_contract); general kernel: vector-output rows */
#define JAMUN_PROF_NODE 6        /* partial-slab reduce + gate + self/skip Linear + noise skip mix   */
#define JAMUN_PROF_HEAD 7        /* output head + xhat/score finalize                                */
#define JAMUN_PROF_TPROD 8       /* T pre-pass of a hidden layer (k_tprod / k_tprod_h), in front of the conv kernel */
#define JAMUN_PROF_NCLASS 9
/* on = 0: off; 1: every class; otherwise a mask with bit (c + 1) set for each class c to time (event records are not free:
 * timing all 16 launches of a forward costs ~4 % of a step, the dominant class alone ~1 %). */
int jamun_profile_enable(jamun_sampler* s, int32_t on);
/* Sample instead of timing every launch: only every `every`-th launch of each enabled class gets its two event records (every >= 1;
 * 1 = all, the default after jamun_sampler_create).  An event record sits between two launches of the stream and costs ~1.3 us of a
 * 0.87 ms step each: the five hidden-layer conv launches of every step timed = 3 % of the step (bench.py: 295 k against 304 k
 * conformations/s); a stride that is coprime to the launches of the class per forward still visits every layer. */
int jamun_profile_sample(jamun_sampler* s, int32_t every);
int jamun_profile_read(jamun_sampler* s, double* ms_total, int64_t* launches, void* stream);

/* Diagnostic builds only (-DJAMUN_STAMP): summed s_memtime cycles per phase of the fused conv kernel
 * {coefficient-tile write, barrier wait, forming MFMAs, main MFMAs}; synchronises the device and clears the counters. */
int jamun_debug_stamps(unsigned long long* out8);

/* Copy internal buffers out for layer-level parity tests (device->device on `stream`):
 *   what = 0: node features after block `layer` (0 = initial projector) [n_atoms, mul0+3*mul1]
 *   what = 1: in-degree (radial + bonded) as float [n_atoms]
 *   what = 2: network output g [n_atoms,3] (before c_skip/c_out)                          */
int jamun_debug_read(jamun_sampler* s, int32_t what, int32_t layer, float* out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JAMUN_HIP_H */
