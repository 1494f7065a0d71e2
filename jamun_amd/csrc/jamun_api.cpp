// jamun_api.cpp — host runtime behind include/jamun_hip.h: checkpoint-tensor lookup, constant folding and
// MFMA-ordered weight packing, work-buffer management, and the per-step launch sequence.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/jamun_hip.h"
#include "jamun_internal.h"

namespace {

thread_local std::string g_err;

struct Err : std::runtime_error {
  int code;
  Err(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HIPCHECK(expr)                                                                                 \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) throw Err(JAMUN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

template <typename F>
int guarded(F&& f) {
  try {
    f();
    return JAMUN_OK;
  } catch (const Err& e) {
    g_err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    g_err = e.what();
    return JAMUN_ERR_INVALID;
  }
}

template <typename T>
T* dev_alloc(size_t n) {
  T* p = nullptr;
  HIPCHECK(hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)));
  return p;
}
template <typename T>
T* dev_upload(const std::vector<T>& v) {
  T* p = dev_alloc<T>(v.size());
  if (!v.empty()) HIPCHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

}  // namespace

struct jamun_model {
  jamun_hparams hp;
  std::map<std::string, std::vector<float>> t;
  const std::vector<float>& get(const std::string& name, int64_t numel = -1) const {
    auto it = t.find(name);
    if (it == t.end()) throw Err(JAMUN_ERR_MISSING, "missing checkpoint tensor: " + name);
    if (numel >= 0 && (int64_t)it->second.size() != numel)
      throw Err(JAMUN_ERR_INVALID, "tensor " + name + " has " + std::to_string(it->second.size()) +
                                       " elements, expected " + std::to_string(numel));
    return it->second;
  }
};

namespace {

// ---- packed conv problem ------------------------------------------------------------------------
struct UEntry {
  int type;      // JAMUN_T_*
  int cross;     // 1 for the cross-product half of an X1C block
  int xoff;      // offset of the channel's first float inside a node feature row
  int64_t wbase; // offset of W row (u, :) inside the flat tensor-product weight vector
  double scale;  // path coefficient * CG factor * input noise scaling
};
struct UBlock {
  int type;
  std::vector<UEntry> e;  // nu entries (even)
};
struct ConvProblemDev {
  float4* wpack = nullptr;
  int4* chunks = nullptr;
  int* slice_ptr = nullptr;
  int4* ublk = nullptr;
  int* lane_xoff = nullptr;
  int planes = 0, nt = 0, xw = 0;
  int64_t K = 0;  // padded contraction depth
};
struct DgDev {
  float4 *wx = nullptr, *wd = nullptr, *wv = nullptr, *wt = nullptr;  // null: the layer cannot use jamun_conv_dg.hip
  float4* wxh = nullptr;  // f16x3 contraction: hi / lo planes of the scaled weights, one stream per (hidden unit, matrix wave)
  float4* wth = nullptr;  // f16x3 T pre-pass: [k][8 groups of 16 inputs][hi, lo][64 lanes], A operand (lane (w', hh): inputs 16 g + 8 hh + j)
  float4* wm = nullptr;   // jamun_conv_mf.hip: [k][4 matrix waves][40 blocks], K index permuted to the forming MFMA's accumulator layout
  int sB = 0, sBt = 0, sTw = 0;
  float hmax2 = 2.f;
  // f16x3 balancing (build_layer): gx [216] 2^e_u per feature element (layout of a feature row), gT [128] the T pre-pass's input factors,
  // cf0 [160] / cf1 [32] / cfT [32] the inverse column scales of the scalar / vector outputs / T
  float *gx = nullptr, *gT = nullptr, *cf0 = nullptr, *cf1 = nullptr, *cfT = nullptr;
  float4* wmt = nullptr;   // tail tiles (k_tail_contract): vector-output weights [k][24 blocks] under one column scale
  float* cf1t = nullptr;   // ... its inverse [32]
};
struct SepDev {
  float4* w2b = nullptr;  // null: not a SeparableConv layer
  float *cfw = nullptr, *bias = nullptr, *wl0 = nullptr, *wl1 = nullptr;
  int n0 = 0, n1 = 0, sH = 0;
};
struct LayerDev {
  ConvProblemDev p0, p1;
  DgDev dg;
  SepDev sep;
  std::vector<float> w1r_h, cmask_h;  // radial MLP first layer (uploaded for all layers together: jamun_sampler::w1r_all)
  int tt_U = 0;  // distinct embedding rows of the tables below
  float* tt2 = nullptr;  // the same table re-laid for k_conv_init_v: [k][U][192]
  float4* tabw = nullptr;  // ... and scaled by 2^tab_sB, split hi + lo, as MFMA B fragments for k_conv_mfi (U <= 32): [k][24 blocks][64 lanes]
  int tab_sB = 0, tab_ut = 0;
  // k_conv_mfx (initial projector formed from the feature rows; batches with more than 32 distinct embedding rows)
  float4* wx = nullptr;                  // [k][48 blocks] balanced, split weights (MfxArgs::wx)
  unsigned *xph = nullptr, *xpl = nullptr;  // the embedding rows times channel factors and 2^x_sX, split, two atoms per word
  int x_sX = 0;
  float *xcf0 = nullptr, *xcf1 = nullptr;
  float4 *wcat0 = nullptr, *wcat1 = nullptr;  // node update: [W_self ; W_skip] as MFMA fragments
  int K0p = 0, K1p = 0;
  float4 *wh0 = nullptr, *wh1 = nullptr;      // ... and balanced per row / column, split hi + lo for the f16x3 kernel (jamun_node.hip)
  int K0h = 0, K1h = 0;
  float *kga0 = nullptr, *kga1 = nullptr, *kgx = nullptr, *cg0 = nullptr, *cg1 = nullptr;  // its row (input) / column (output) powers of two
  float* mix = nullptr;
  int in0 = 0, in1 = 0, XSin = 0;
  int64_t tp_numel = 0;
};

void free_dg(DgDev& d) {
  hipFree(d.wx); hipFree(d.wd); hipFree(d.wv); hipFree(d.wt); hipFree(d.wxh); hipFree(d.wth); hipFree(d.wm);
  hipFree(d.gx); hipFree(d.gT); hipFree(d.cf0); hipFree(d.cf1); hipFree(d.cfT); hipFree(d.wmt); hipFree(d.cf1t);
  d = DgDev{};
}

void free_problem(ConvProblemDev& p) {
  hipFree(p.wpack); hipFree(p.chunks); hipFree(p.slice_ptr); hipFree(p.ublk); hipFree(p.lane_xoff);
}

// fp32 -> IEEE binary16, round to nearest even (the device side uses v_cvt_pk_f16_f32 in the default rounding mode); values
// beyond the f16 range do not occur (the caller scales into [-2^14, 2^14])
uint16_t f32_to_f16_rne(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);  // >= 65536 (or inf / nan): inf
  if (x < 0x38800000u) {                                     // below the smallest normal half (2^-14): subnormal or zero
    if (x < 0x33000000u) return (uint16_t)sign;              // < 2^-25: rounds to zero
    const int e = (int)(x >> 23);                             // biased exponent, 102 .. 112
    const uint32_t mant = (x & 0x7fffffu) | 0x800000u;        // 24-bit significand
    const int shift = 126 - e;                                // result = mant >> shift, in units of 2^-24
    const uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
    return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
  }
  const uint32_t mant = x & 0x7fffffu, e = (x >> 23) - 112u;  // half exponent field 1 .. 30
  uint32_t h = (e << 10) | (mant >> 13);
  const uint32_t rem = mant & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;     // (a carry into the exponent is the correct result)
  return (uint16_t)(sign | h);
}
float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  float out;
  if (e == 0) {
    out = std::ldexp((float)m, -24);
    if (sign) out = -out;
    return out;
  }
  const uint32_t x = sign | ((e == 31 ? 255u : e + 112u) << 23) | (m << 13);
  std::memcpy(&out, &x, 4);
  return out;
}
// value -> (hi, lo) halves with hi = rne16(v), lo = rne16(v - hi)
inline void split_f16(double v, uint16_t& hi, uint16_t& lo) {
  const float f = (float)v;
  hi = f32_to_f16_rne(f);
  lo = f32_to_f16_rne(f - f16_to_f32(hi));
}

std::vector<double> noise_mlp(const jamun_model& m, const std::string& prefix, int k, double c_noise) {
  // Linear(1->k) . SELU . Linear(k->k)   (src/jamun/model/noise_conditioning.py:33-37)
  const auto& w0 = m.get(prefix + ".0.weight", k);
  const auto& b0 = m.get(prefix + ".0.bias", k);
  const auto& w2 = m.get(prefix + ".2.weight", (int64_t)k * k);
  const auto& b2 = m.get(prefix + ".2.bias", k);
  const double alpha = 1.6732632423543772848170429916717, scale = 1.0507009873554804934193349852946;
  std::vector<double> h(k), out(k);
  for (int i = 0; i < k; ++i) {
    double z = (double)w0[i] * c_noise + (double)b0[i];
    h[i] = scale * (z > 0 ? z : alpha * (std::exp(z) - 1.0));
  }
  for (int o = 0; o < k; ++o) {
    double s = b2[o];
    for (int i = 0; i < k; ++i) s += (double)w2[(size_t)o * k + i] * h[i];
    out[o] = s;
  }
  return out;
}

struct InBlock { int mul, l, xoff, ch0; };

// K-slices are ranges of the hidden index k (hidden units + bias row); each is processed in k-subgroups of
// ksub or ksub-1 hidden units (the two sizes the conv kernel is instantiated for).
std::vector<std::pair<int, int>> k_subgroups(int n_k, int n_slices, int ksub, std::vector<int>& slice_first_sub) {
  std::vector<std::pair<int, int>> subs;  // (k0, ks)
  slice_first_sub.assign(n_slices + 1, 0);
  const int base = n_k / n_slices, rem = n_k % n_slices;
  int k = 0;
  for (int s = 0; s < n_slices; ++s) {
    const int size = base + (s >= n_slices - rem ? 1 : 0);
    slice_first_sub[s] = (int)subs.size();
    if (size == 0) continue;
    const int n_sub = (size + ksub - 1) / ksub;
    const int lo = size / n_sub, n_hi = size % n_sub;  // n_hi subgroups of lo+1, the rest of lo
    if (lo + (n_hi ? 1 : 0) > ksub || lo < ksub - 1 || (lo < 1))
      throw Err(JAMUN_ERR_INVALID, "cannot split a K-slice of " + std::to_string(size) + " hidden units into groups of " +
                                       std::to_string(ksub - 1) + "/" + std::to_string(ksub));
    for (int i = 0; i < n_sub; ++i) {
      const int ks = lo + (i < n_hi ? 1 : 0);
      subs.push_back({k, ks});
      k += ks;
    }
  }
  slice_first_sub[n_slices] = (int)subs.size();
  return subs;
}

ConvProblemDev pack_problem(const std::vector<UBlock>& blocks, int planes, int G, int n_slices, int ksub,
                            const std::vector<float>& W3, const std::vector<float>& b3, int hidden) {
  ConvProblemDev P;
  P.planes = planes;
  P.nt = (G + 31) / 32;
  const int NT = P.nt;
  std::vector<int> first_sub;
  const auto subs = k_subgroups(hidden + 1, n_slices, ksub, first_sub);
  std::vector<int4> chunks;
  std::vector<int> sp(n_slices + 1, 0);
  int64_t gofs = 0;  // in weight groups (4 K-steps x NT tiles x 64 lanes x float)
  for (int s = 0; s < n_slices; ++s) {
    sp[s] = (int)chunks.size();
    for (int si = first_sub[s]; si < first_sub[s + 1]; ++si)  // k-subgroup major: the staged h~ records are reused by the u-blocks
      for (size_t b = 0; b < blocks.size(); ++b) {
        const int nu = (int)blocks[b].e.size(), k0 = subs[si].first, ks = subs[si].second;
        const int ng = (ks * nu / 2 + 3) / 4;
        chunks.push_back(make_int4((int)b, k0 | (ks << 16), (int)gofs, ng));
        gofs += ng;
      }
  }
  sp[n_slices] = (int)chunks.size();
  if (gofs * NT * 64 > (int64_t)0x7fffffff) throw Err(JAMUN_ERR_INVALID, "packed conv weights too large");
  std::vector<float4> wp((size_t)gofs * NT * 64, make_float4(0.f, 0.f, 0.f, 0.f));
  int64_t Ktot = 0;
  for (const int4& cd : chunks) {
    const UBlock& B = blocks[cd.x];
    const int nu = (int)B.e.size(), k0 = cd.y & 0xffff, ks = cd.y >> 16;
    Ktot += (int64_t)cd.w * 8;
    for (int g = 0; g < cd.w; ++g)
      for (int nt = 0; nt < NT; ++nt)
        for (int lane = 0; lane < 64; ++lane) {
          const int hh = lane >> 5, c = lane & 31, col = nt * 32 + c;
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          for (int st = 0; st < 4; ++st) {
            const int kidx = 2 * (4 * g + st) + hh;
            if (kidx >= ks * nu) continue;
            const int kk = kidx / nu, ul = kidx % nu, k = k0 + kk;
            const UEntry& ue = B.e[ul];
            if (k <= hidden && col < G && ue.scale != 0.0) {
              const int64_t p = ue.wbase + col;
              const double w = (k < hidden) ? (double)W3[(size_t)p * hidden + k] : (double)b3[p];
              v[st] = (float)(w * ue.scale);
            }
          }
          wp[((size_t)(cd.z + g) * NT + nt) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
  }
  P.K = Ktot;
  std::vector<int4> ub;
  std::vector<int> lx;
  int xw = 1;
  for (const UBlock& B : blocks) {
    int lo = 1 << 30, hi = 0;
    const int per = (B.type == JAMUN_T_X0 || B.type == JAMUN_T_X0V) ? 1 : 3;
    for (const UEntry& e : B.e)
      if (e.scale != 0.0) { lo = std::min(lo, e.xoff); hi = std::max(hi, e.xoff + per); }
    if (hi == 0) { lo = 0; hi = per; }
    ub.push_back(make_int4(B.type, (int)B.e.size(), lo, hi - lo));
    xw = std::max(xw, hi - lo);
    for (int lane = 0; lane < 64; ++lane) {
      int v = 0;
      if (lane < (int)B.e.size() && B.e[lane].scale != 0.0) v = (B.e[lane].xoff - lo) | (B.e[lane].cross ? JAMUN_XOFF_CROSS : 0);
      lx.push_back(v);
    }
  }
  P.xw = xw;
  P.wpack = dev_upload(wp);
  P.chunks = dev_upload(chunks);
  P.slice_ptr = dev_upload(sp);
  P.ublk = dev_upload(ub);
  P.lane_xoff = dev_upload(lx);
  return P;
}

void pad_even(UBlock& b) {
  if (b.e.size() % 2) b.e.push_back(UEntry{b.type, 0, 0, 0, 0.0});
}

}  // namespace

struct jamun_sampler {
  jamun_hparams hp;
  jamun_tuning tune{};  // kernel-selection switches of jamun_sampler_create (all zero: defaults)
  float sigma = 0;
  int n_atoms = 0, n_graphs = 0, n_pad = 0, S = 0, n_slices = 8;
  int XS = 0, n_emb = 0;
  float c_in = 0, c_skip = 0, c_out = 0, r_cut = 0, r2 = 0, rb_step = 0;
  // static device data
  int *ptr = nullptr, *bond_in_ptr = nullptr, *bond_in_src = nullptr;
  int n_tiles = 0;
  bool initv_on = false;              // initial projector on k_conv_init_v (tiles / segments of the dg kernel)
  int initv_nbuf = 2;                 // its row buffers in LDS
  int* atom_uid = nullptr;            // [n_atoms] index of the atom's distinct (scaled) embedding row
  // destination-grouped VALU-forming conv kernel (jamun_conv_dg.hip; hidden layers): own tile plan (larger source spans)
  bool dg_on = false, dg_row_blocks = false;
  int dg_mode = 0;  // 0 two-phase resident, 1 alternating residency, 2 single phase (see jamun_sampler_create)
  int dg_emu = 1;   // 1: f16x3 contraction (three f16 MFMAs per fp32 product); 0 (jamun_tuning.dg_fp32): v_mfma_f32_32x32x2_f32; stats report 2 for the opt-in f16x1 mode (s->x1)
  int dg_RS = 0, dg_grid = 0, dg_max_segs = 0, dg_n_slabs = 0, dg_n_tiles = 0;
  int2 *dg_tile_atoms = nullptr, *dg_tile_span = nullptr;
  int4* dg_segs = nullptr;
  int* dg_atom_nslab = nullptr;
  float* sep_D = nullptr;  // SeparableConv: [n_atoms][K0 + 3 K1] per-destination sums of the layer at hand
  int cus = 1;
  float* dg_T = nullptr;  // [n_k][n_atoms][32] pre-pass product of a hidden layer (k_tprod), reused by every layer
  int dg_tstride = 0;     // mode 4 (jamun_conv_mf.hip): dg_T is [n_k][32][dg_tstride], transposed
  // tail tiles of the mode-4 plan (tiles with few destinations): formed with the hidden unit in the column index and contracted 32 gathered
  // destinations at a time (k_tail_form / k_tail_contract) instead of as whole tiles of k_conv_mf; the initial projector keeps them as tiles
  int n_tail_tiles = 0, n_tail = 0, tail_runs = 0;
  int mf_nks = 4;  // forming K-steps of k_conv_mf (3: every whole tile's sources lie in the first 48 rows of its window)
  int x1 = 0;         // 1: reduced-precision hidden-layer conv (jamun_tuning.f16x1) — honoured by k_conv_mf / k_conv_ml (dg_mode 4 / 5) only
  int ml_window = 0;  // mode 5 (jamun_conv_ml.hip): source rows of the instantiation (96, 128, 168)
  unsigned long long* ml_count = nullptr;  // device: v_mfma_f32_32x32x16_f16 executed by k_conv_ml since create (depends on the occupied source blocks)
  int64_t ml_launches = 0;                 // ... over this many launches
  int4* tail_tiles = nullptr;
  int* tail_atom = nullptr;
  float* tail_scale = nullptr;
  float4* tail_P = nullptr;
  int4* init_segs = nullptr;   // segment lists of the initial projector (ALL tiles) when the hidden layers' lists leave the tail tiles out
  int* init_atom_nslab = nullptr;
  int init_max_segs = 0, init_n_slabs = 0;
  bool init_tail = false;      // the initial projector sends the tail tiles through k_tail_form_init / k_tail_contract as well
  bool mfi_on = false;    // initial projector on k_conv_mfi (mode 4 tiles, at most 32 distinct embedding rows)
  bool mfx_on = false;    // ... or on k_conv_mfx (mode 4 tiles, any number of distinct rows: formed from the feature rows)
  bool mlx_on = false;    // ... or, on the mode 5 tiles (large spans), k_conv_mlx
  int n_uniq = 0;         // distinct (noise-scaled) embedding rows of the batch
  int* mf_err = nullptr;  // device flag of k_conv_mf
  int* mf_err_host = nullptr;  // pinned copy, refreshed behind every entry point that ran a forward (mf_err_fetch / mf_err_check)
  float *x_emb = nullptr, *mu = nullptr;
  std::vector<LayerDev> layers;
  float *w_gate = nullptr, *w_vec = nullptr, *w_out = nullptr;
  // work buffers
  float *w1r_all = nullptr, *cmask_all = nullptr;  // [layers][64][32], [layers][2][64]
  float4* w1h_all = nullptr;                        // k_edge_h16: [layers][2 k-tiles][2 K-steps][hi, lo][64 lanes]
  float* w1isc_all = nullptr;                       // [layers] 2^-(14 + sW)
  size_t h_stride = 0, h_kstride = 0;  // per layer: [65 hidden rows][h_kstride edge slots]
  bool h_batched = false, edges_built = false;
  float *yc = nullptr, *h = nullptr, *partial0 = nullptr, *partial1 = nullptr, *g = nullptr, *tmp = nullptr;
  float *xhat_buf = nullptr, *score_buf = nullptr, *psi = nullptr;
  int *deg = nullptr, *esrc = nullptr;
  int* epair = nullptr;  // k_geom's pair table (jamun_internal.h: JAMUN_EP_*), one word per edge slot
  float4* egeo = nullptr;
  std::vector<float*> x;  // per block output [n_atoms][XS]
  unsigned long long* counter = nullptr;
  int64_t flop_ref_per_edge = 0, flop_exec = 0, conv_flop_exec_launch = 0;
  // optional per-kernel-class timing with HIP events on the launch stream (jamun_profile_*)
  unsigned prof_mask = 0;  // bit c: record HIP events around launches of profile class c
  int prof_every = 1;      // ... around every prof_every-th launch of the class (jamun_profile_sample)
  int prof_seen[JAMUN_PROF_NCLASS] = {0};
  std::vector<hipEvent_t> ev_pool;
  std::vector<std::pair<int, std::pair<int, int>>> ev_used;  // (class, (begin, end))
  size_t ev_next = 0;

  ~jamun_sampler() {
    hipFree(ptr); hipFree(bond_in_ptr); hipFree(bond_in_src); hipFree(x_emb); hipFree(mu);
    hipFree(atom_uid); hipFree(w1r_all); hipFree(cmask_all); hipFree(w1h_all); hipFree(w1isc_all);
    hipFree(epair); hipFree(dg_tile_atoms); hipFree(dg_tile_span); hipFree(dg_segs); hipFree(dg_atom_nslab); hipFree(dg_T); hipFree(mf_err); hipFree(ml_count);
    hipFree(tail_tiles); hipFree(tail_atom); hipFree(tail_scale); hipFree(tail_P); hipFree(init_segs); hipFree(init_atom_nslab);
    if (mf_err_host) hipHostFree(mf_err_host);
    for (auto& L : layers) {
      free_problem(L.p0); free_problem(L.p1); free_dg(L.dg);
      hipFree(L.sep.w2b); hipFree(L.sep.cfw); hipFree(L.sep.bias); hipFree(L.sep.wl0); hipFree(L.sep.wl1);
      hipFree(L.wcat0); hipFree(L.wcat1); hipFree(L.wh0); hipFree(L.wh1); hipFree(L.kga0); hipFree(L.kga1); hipFree(L.kgx); hipFree(L.cg0); hipFree(L.cg1); hipFree(L.mix); hipFree(L.tt2); hipFree(L.tabw);
      hipFree(L.wx); hipFree(L.xph); hipFree(L.xpl); hipFree(L.xcf0); hipFree(L.xcf1);
    }
    hipFree(w_gate); hipFree(w_vec); hipFree(w_out);
    hipFree(yc); hipFree(h); hipFree(partial0); hipFree(partial1); hipFree(g); hipFree(tmp);
    hipFree(xhat_buf); hipFree(score_buf); hipFree(psi); hipFree(deg); hipFree(esrc); hipFree(egeo);
    for (float* p : x) hipFree(p);
    hipFree(counter); hipFree(sep_D);
    for (hipEvent_t e : ev_pool) hipEventDestroy(e);
  }
};

namespace {

void build_layer_common(const jamun_model& m, const std::string& prefix, const std::vector<InBlock>& in_blocks, const std::vector<double>& s_in,
                        LayerDev& L, int in0, int in1) {
  const jamun_hparams& hp = m.hp;
  const int mul0 = hp.mul0, mul1 = hp.mul1, H = hp.edge_attr_dim;
  // ---- radial MLP first layer: split into the constant bonded part and the radial part
  const auto& W1 = m.get(prefix + ".gated_conv.f.f.radial_nn.0.weight", (int64_t)H * H);
  const auto& b1 = m.get(prefix + ".gated_conv.f.f.radial_nn.0.bias", H);
  const int nb = H / 2, nr = (H + 1) / 2;
  const auto& Eb = m.get("embed_bondedness.weight", 2 * nb);
  std::vector<float> w1r((size_t)H * nr), cmask(2 * (size_t)H);
  for (int k = 0; k < H; ++k) {
    for (int r = 0; r < nr; ++r) w1r[(size_t)r * H + k] = W1[(size_t)k * H + nb + r];  // [basis][hidden]: lane = hidden unit
    for (int mk = 0; mk < 2; ++mk) {
      double s = b1[k];
      for (int c = 0; c < nb; ++c) s += (double)W1[(size_t)k * H + c] * Eb[(size_t)mk * nb + c];
      cmask[(size_t)mk * H + k] = (float)s;
    }
  }
  L.w1r_h = w1r;
  L.cmask_h = cmask;
  {
    // static bound of |h~| = |SiLU(c_mask + W1[:, radial part] . radial(d))| over the layer: the Gaussian basis values are
    // positive and sum to at most sqrt(pi) / 1.12 < 1.6 at any distance, |SiLU(z)| <= max(|z|, 0.2785); the bias row is 1
    double hm = 1.0;
    for (int k = 0; k < H; ++k) {
      double wm = 0;
      for (int r = 0; r < nr; ++r) wm = std::max(wm, std::fabs((double)W1[(size_t)k * H + nb + r]));
      const double z = std::max(std::fabs((double)cmask[k]), std::fabs((double)cmask[(size_t)H + k])) + 1.6 * wm;
      hm = std::max(hm, z);
    }
    L.dg.hmax2 = (float)(2.0 * hm * 1.0001);
  }

  // ---- o3.Linear skip (in -> hidden) and self-interaction (hidden -> hidden)  (_interaction.py:23-30)
  int64_t n_skip = 0;
  for (auto& ib : in_blocks) n_skip += (int64_t)ib.mul * (ib.l == 0 ? mul0 : mul1);
  const auto& Wskip = m.get(prefix + ".gated_conv.skip_connection.weight", n_skip);
  const auto& Wself = m.get(prefix + ".gated_conv.self_interaction.weight", (int64_t)mul0 * mul0 + (int64_t)mul1 * mul1);
  std::vector<float> ws0((size_t)std::max(in0, 1) * mul0, 0.f), ws1((size_t)std::max(in1, 1) * std::max(mul1, 1), 0.f);
  {
    int64_t o = 0;
    int u0 = 0, u1 = 0;
    for (auto& ib : in_blocks) {
      if (ib.l == 0) {
        for (int u = 0; u < ib.mul; ++u, ++u0)
          for (int w = 0; w < mul0; ++w)
            ws0[(size_t)u0 * mul0 + w] = (float)((double)Wskip[o + (int64_t)u * mul0 + w] / std::sqrt((double)in0) * s_in[ib.ch0 + u]);
        o += (int64_t)ib.mul * mul0;
      } else {
        for (int u = 0; u < ib.mul; ++u, ++u1)
          for (int w = 0; w < mul1; ++w)
            ws1[(size_t)u1 * mul1 + w] = (float)((double)Wskip[o + (int64_t)u * mul1 + w] / std::sqrt((double)in1) * s_in[ib.ch0 + u]);
        o += (int64_t)ib.mul * mul1;
      }
    }
  }
  std::vector<float> wf0((size_t)mul0 * mul0), wf1((size_t)std::max(mul1 * mul1, 1));
  for (int i = 0; i < mul0 * mul0; ++i) wf0[i] = (float)((double)Wself[i] / std::sqrt((double)mul0));
  for (int i = 0; i < mul1 * mul1; ++i) wf1[i] = (float)((double)Wself[(size_t)mul0 * mul0 + i] / std::sqrt((double)mul1));
  // concatenate along K ([self ; skip]) and pack as MFMA B fragments for k_node_update
  auto pack_cat = [](const std::vector<float>& wself, int ks, const std::vector<float>& wskip, int kk, int ncol, int& Kp) {
    Kp = (ks + kk + 7) & ~7;
    const int nt = (ncol + 31) / 32, nsg = Kp / 8;
    std::vector<float4> out((size_t)nt * nsg * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int t = 0; t < nt; ++t)
      for (int sg = 0; sg < nsg; ++sg)
        for (int lane = 0; lane < 64; ++lane) {
          const int hh = lane >> 5, c = t * 32 + (lane & 31);
          float v[4] = {0.f, 0.f, 0.f, 0.f};
          for (int st = 0; st < 4; ++st) {
            const int row = 2 * (4 * sg + st) + hh;
            if (c >= ncol) continue;
            if (row < ks) v[st] = wself[(size_t)row * ncol + c];
            else if (row < ks + kk) v[st] = wskip[(size_t)(row - ks) * ncol + c];
          }
          out[((size_t)t * nsg + sg) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
    return out;
  };
  L.wcat0 = dev_upload(pack_cat(wf0, mul0, ws0, in0, mul0, L.K0p));
  L.wcat1 = dev_upload(pack_cat(wf1, mul1, ws1, in1, std::max(mul1, 1), L.K1p));
  // f16x3 node update: the same matrices balanced by exact powers of two — row K (an input channel) times 2^-e_K so that its largest
  // magnitude sits in [0.5, 1), then column w times 2^sW_w so that its largest sits in [2^13, 2^14) — and split hi + lo; K padded to 16.
  // rowf[K] = 2^e_K multiplies the input when the kernel stages it, colf[w] = 2^-sW_w the output column.
  auto pow2_above = [](double v) { int ex = 0; if (v > 0 && std::isfinite(v)) std::frexp(v, &ex); return std::max(-40, std::min(40, ex)); };  // v < 2^ex
  auto pack_cat_h = [&](const std::vector<float>& wself, int ks, const std::vector<float>& wskip, int kk, int ncol, int& Kh,
                        std::vector<float>& rowf, std::vector<float>& colf) {
    Kh = (ks + kk + 15) & ~15;
    auto W = [&](int row, int c) -> double {
      if (c >= ncol) return 0.0;
      if (row < ks) return wself[(size_t)row * ncol + c];
      if (row < ks + kk) return wskip[(size_t)(row - ks) * ncol + c];
      return 0.0;
    };
    rowf.assign(Kh, 1.f);
    std::vector<double> rinv(Kh, 1.0);
    for (int r = 0; r < ks + kk; ++r) {
      double m = 0;
      for (int c = 0; c < ncol; ++c) m = std::max(m, std::fabs(W(r, c)));
      const int ex = pow2_above(m);
      rowf[r] = (float)std::ldexp(1.0, ex);
      rinv[r] = std::ldexp(1.0, -ex);
    }
    const int nt = (ncol + 31) / 32, nst = Kh / 16;
    colf.assign((size_t)nt * 32, 0.f);
    std::vector<double> csc((size_t)nt * 32, 1.0);
    for (int c = 0; c < ncol; ++c) {
      double m = 0;
      for (int r = 0; r < ks + kk; ++r) m = std::max(m, std::fabs(W(r, c) * rinv[r]));
      const int sW = 14 - pow2_above(m);
      csc[c] = std::ldexp(1.0, sW);
      colf[c] = (float)std::ldexp(1.0, -sW);
    }
    std::vector<float4> out((size_t)nt * nst * 2 * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int t = 0; t < nt; ++t)
      for (int st = 0; st < nst; ++st)
        for (int lane = 0; lane < 64; ++lane) {
          const int hh = lane >> 5, c = t * 32 + (lane & 31);
          uint32_t h[4], l[4];
          for (int i = 0; i < 4; ++i) {
            uint16_t hp[2], lp[2];
            for (int e = 0; e < 2; ++e) {
              const int row = 16 * st + 8 * hh + 2 * i + e;
              split_f16(W(row, c) * rinv[row] * csc[c], hp[e], lp[e]);
            }
            h[i] = (uint32_t)hp[0] | ((uint32_t)hp[1] << 16);
            l[i] = (uint32_t)lp[0] | ((uint32_t)lp[1] << 16);
          }
          const size_t b = (((size_t)t * nst + st) * 2) * 64 + lane;
          std::memcpy(&out[b], h, 16);
          std::memcpy(&out[b + 64], l, 16);
        }
    return out;
  };
  {
    std::vector<float> r0, c0, r1, c1;
    L.wh0 = dev_upload(pack_cat_h(wf0, mul0, ws0, in0, mul0, L.K0h, r0, c0));
    L.wh1 = dev_upload(pack_cat_h(wf1, mul1, ws1, in1, std::max(mul1, 1), L.K1h, r1, c1));
    // row factors in the layouts the kernel reads them: activated scalars [mul0], gated vectors [mul1], and the channels of x_in in
    // x_in's own layout (in0 scalars, then in1 vectors x 3 components)
    std::vector<float> ka0(r0.begin(), r0.begin() + mul0), ka1(std::max(mul1, 1), 1.f), kx((size_t)((in0 + 3 * in1 + 3) & ~3), 1.f);
    for (int u = 0; u < mul1; ++u) ka1[u] = r1[u];
    for (int u = 0; u < in0; ++u) kx[u] = r0[mul0 + u];
    for (int u = 0; u < in1; ++u)
      for (int mm = 0; mm < 3; ++mm) kx[in0 + 3 * u + mm] = r1[mul1 + u];
    ka0.resize((size_t)((mul0 + 3) & ~3) + 4, 1.f);
    L.kga0 = dev_upload(ka0); L.kga1 = dev_upload(ka1); L.kgx = dev_upload(kx); L.cg0 = dev_upload(c0); L.cg1 = dev_upload(c1);
  }
}

// SeparableConv block (src/jamun/e3tools/nn/_tensor_product.py:27-47): depth-wise "uvu" instructions in e3nn order — for every
// input block, for sh in (0e, 1e), for l_out = |l1 - l2| .. l1 + l2 kept when it occurs in the output irreps or is 0e — each with
// mul_in weights and its own block of irreps_out_dtp; then o3.Linear(irreps_out_dtp -> G0 x0e + G1 x1e).  Packed in the canonical
// order of jamun_sepconv.hip: weights [A | B | C | D | E], Linear rows scalars [D0 | D3], vectors [D1 | D2 | D4].
LayerDev build_layer_separable(const jamun_model& m, const std::string& prefix, const std::vector<InBlock>& in_blocks, const std::vector<double>& s_in,
                               LayerDev& L) {
  const jamun_hparams& hp = m.hp;
  const int mul0 = hp.mul0, mul1 = hp.mul1, G0 = mul0 + mul1, G1 = mul1, H = hp.edge_attr_dim;
  int n0 = 0, n1 = 0;
  for (auto& ib : in_blocks) (ib.l == 0 ? n0 : n1) += ib.mul;
  struct Tri { int kind, u0, mul; int64_t woff, loff; };  // kind: 0 A, 1 B, 2 C, 3 D, 4 E; u0: first canonical channel; offsets: radial_nn.3 row, lin weight
  std::vector<Tri> tri;
  int64_t woff = 0, loff = 0;
  int u0 = 0, u1 = 0;
  for (auto& ib : in_blocks) {
    if (ib.l == 0) {
      tri.push_back({0, u0, ib.mul, woff, loff}); woff += ib.mul; loff += (int64_t)ib.mul * G0;
      tri.push_back({1, u0, ib.mul, woff, loff}); woff += ib.mul; loff += (int64_t)ib.mul * G1;
      u0 += ib.mul;
    } else {
      tri.push_back({2, u1, ib.mul, woff, loff}); woff += ib.mul; loff += (int64_t)ib.mul * G1;
      tri.push_back({3, u1, ib.mul, woff, loff}); woff += ib.mul; loff += (int64_t)ib.mul * G0;
      tri.push_back({4, u1, ib.mul, woff, loff}); woff += ib.mul; loff += (int64_t)ib.mul * G1;
      u1 += ib.mul;
    }
  }
  const auto& W3 = m.get(prefix + ".gated_conv.f.f.radial_nn.3.weight", woff * H);
  const auto& b3 = m.get(prefix + ".gated_conv.f.f.radial_nn.3.bias", woff);
  const auto& WL = m.get(prefix + ".gated_conv.f.f.tp.lin.weight", loff);
  // input noise scaling per canonical channel
  std::vector<double> s0(n0, 1.0), s1(n1, 1.0);
  {
    int a0 = 0, a1 = 0;
    for (auto& ib : in_blocks)
      for (int u = 0; u < ib.mul; ++u) (ib.l == 0 ? s0[a0++] : s1[a1++]) = s_in[ib.ch0 + u];
  }
  const int NW = 2 * n0 + 3 * n1, NWp = (NW + 31) & ~31, n_ct = NWp / 32;
  const int base[5] = {0, n0, 2 * n0, 2 * n0 + n1, 2 * n0 + 2 * n1};
  // path weight sqrt(2 l_out + 1) x Clebsch-Gordan factor x the sqrt(3) of Y_1 = sqrt(3) v:  A 1, B sqrt(3) (delta/sqrt(3) sqrt(3) sqrt(3)),
  // C 1 (sqrt(3) delta/sqrt(3)), D 1 (delta/sqrt(3) sqrt(3)), E sign sqrt(3) (eps/sqrt(6)) sqrt(3) = sign sqrt(3/2)
  const double fac[5] = {1.0, std::sqrt(3.0), 1.0, 1.0, (double)hp.w3j_111_sign * std::sqrt(1.5)};
  std::vector<double> w2c((size_t)(H + 1) * NWp, 0.0);
  for (const Tri& t : tri)
    for (int u = 0; u < t.mul; ++u) {
      const double sc = fac[t.kind] * (t.kind < 2 ? s0[t.u0 + u] : s1[t.u0 + u]);
      const int col = base[t.kind] + t.u0 + u;
      for (int k = 0; k <= H; ++k)
        w2c[(size_t)k * NWp + col] = sc * (k < H ? (double)W3[(size_t)(t.woff + u) * H + k] : (double)b3[t.woff + u]);
    }
  // B fragments of the f16x3 weight GEMM of k_sep_fused: column tiles A 0..3 (x0 -> 0e, channel u at column 32 ct + c), B 4..7, C 8, D 9,
  // E 10; every column balanced by its own power of two (the depth-wise weights inherit the spread of the channels they multiply),
  // split hi + lo; the bias row (hidden unit H: the radial MLP's output bias) is added in fp32 after the product
  // (the envelope of k_sep_fused / k_sep_linear, with the reason: the same texts as sep_conv_unsupported, which sees the edge stride too)
  if (n0 > 128 || n1 > 32) throw Err(JAMUN_ERR_INVALID, "SeparableConv: input irreps wider than 128x0e + 32x1e");
  if (H != 64) throw Err(JAMUN_ERR_INVALID, "SeparableConv: radial MLP with other than 64 hidden units");
  (void)n_ct;
  std::vector<float> cfw(352, 0.f), bias(352, 0.f);
  std::vector<float4> w2b((size_t)4 * 11 * 2 * 64, make_float4(0.f, 0.f, 0.f, 0.f));
  {
    auto old_col = [&](int nc) -> int {  // new column -> canonical column of w2c (-1: padding)
      if (nc < 128) return nc < n0 ? base[0] + nc : -1;
      if (nc < 256) return nc - 128 < n0 ? base[1] + (nc - 128) : -1;
      const int kind = 2 + (nc - 256) / 32, u = (nc - 256) % 32;
      return u < n1 ? base[kind] + u : -1;
    };
    std::vector<double> csc(352, 1.0);
    for (int nc = 0; nc < 352; ++nc) {
      const int oc = old_col(nc);
      if (oc < 0) continue;
      double mx = 0;
      for (int k = 0; k < H; ++k) mx = std::max(mx, std::fabs(w2c[(size_t)k * NWp + oc]));
      int ex = 0;
      if (mx > 0 && std::isfinite(mx)) std::frexp(mx, &ex);
      const int sW = 14 - std::max(-40, std::min(40, ex));
      csc[nc] = std::ldexp(1.0, sW);
      cfw[nc] = (float)std::ldexp(1.0, -sW);
      bias[nc] = (float)w2c[(size_t)H * NWp + oc];
    }
    for (int s4 = 0; s4 < 4; ++s4)
      for (int ct = 0; ct < 11; ++ct)
        for (int lane = 0; lane < 64; ++lane) {
          const int hh = lane >> 5, nc = 32 * ct + (lane & 31), oc = old_col(nc);
          uint32_t hw[4], lw[4];
          for (int i = 0; i < 4; ++i) {
            uint16_t hp[2], lp[2];
            for (int e = 0; e < 2; ++e) {
              const int k = 16 * s4 + 8 * hh + 2 * i + e;
              split_f16(oc >= 0 ? w2c[(size_t)k * NWp + oc] * csc[nc] : 0.0, hp[e], lp[e]);
            }
            hw[i] = (uint32_t)hp[0] | ((uint32_t)hp[1] << 16);
            lw[i] = (uint32_t)lp[0] | ((uint32_t)lp[1] << 16);
          }
          const size_t bidx = (((size_t)s4 * 11 + ct) * 2) * 64 + lane;
          std::memcpy(&w2b[bidx], hw, 16);
          std::memcpy(&w2b[bidx + 64], lw, 16);
        }
  }
  const int K0 = n0 + n1, K1 = n0 + 2 * n1;
  std::vector<float> wl0((size_t)K0 * G0, 0.f), wl1((size_t)std::max(K1 * G1, 1), 0.f);
  for (const Tri& t : tri)
    for (int u = 0; u < t.mul; ++u) {
      const bool scalar_out = t.kind == 0 || t.kind == 3;
      const int G = scalar_out ? G0 : G1;
      const int row = t.kind == 0 ? t.u0 + u : t.kind == 3 ? n0 + t.u0 + u : t.kind == 1 ? t.u0 + u : t.kind == 2 ? n0 + t.u0 + u : n0 + n1 + t.u0 + u;
      const double nrm = 1.0 / std::sqrt((double)(scalar_out ? K0 : K1));
      for (int w = 0; w < G; ++w) (scalar_out ? wl0 : wl1)[(size_t)row * G + w] = (float)((double)WL[t.loff + (int64_t)u * G + w] * nrm);
    }
  L.sep.w2b = dev_upload(w2b);
  L.sep.cfw = dev_upload(cfw);
  L.sep.bias = dev_upload(bias);
  L.sep.wl0 = dev_upload(wl0);
  L.sep.wl1 = dev_upload(wl1);
  L.sep.n0 = n0; L.sep.n1 = n1;
  L.p0.nt = (G0 + 31) / 32;  // (the node update reads the slab widths from here)
  L.p1.nt = (G1 + 31) / 32;
  L.p0.planes = 1; L.p1.planes = 3;
  L.in0 = n0; L.in1 = n1; L.XSin = n0 + 3 * n1;
  L.tp_numel = woff;
  build_layer_common(m, prefix, in_blocks, s_in, L, n0, n1);
  {  // static scale of h~ (bounded by the radial MLP's first layer: build_layer_common)
    int ex = 0;
    std::frexp(0.5 * (double)L.dg.hmax2, &ex);
    L.sep.sH = std::max(-40, std::min(40, 14 - ex));
  }
  return L;
}

LayerDev build_layer(const jamun_model& m, const std::string& prefix, const std::vector<InBlock>& in_blocks,
                     const std::vector<double>& s_in, int n_slices,
                     const std::vector<float>* uniq_rows = nullptr, int row_len = 0, bool pack_dg = false,
                     const std::vector<float>* all_rows = nullptr) {
  const jamun_hparams& hp = m.hp;
  const int mul0 = hp.mul0, mul1 = hp.mul1, G0 = mul0 + mul1, G1 = mul1, H = hp.edge_attr_dim;
  LayerDev L;
  // ---- FullyConnectedTensorProduct instruction table (e3nn order: for i1, for i2 in (0e,1e), for i_out in (0e,1e))
  struct Ins { int b, l2, lo; int64_t off; };
  std::vector<Ins> ins;
  int64_t off = 0;
  double sum0 = 0, sum1 = 0;  // sum over instructions of mul1*mul2 feeding each output irrep
  for (size_t b = 0; b < in_blocks.size(); ++b)
    for (int l2 = 0; l2 <= 1; ++l2)
      for (int lo = 0; lo <= 1; ++lo) {
        const int l1 = in_blocks[b].l;
        if (lo < std::abs(l1 - l2) || lo > l1 + l2) continue;
        const int gout = lo == 0 ? G0 : G1;
        if (gout == 0) continue;
        ins.push_back({(int)b, l2, lo, off});
        off += (int64_t)in_blocks[b].mul * gout;
        (lo == 0 ? sum0 : sum1) += in_blocks[b].mul;
      }
  L.tp_numel = off;
  if (hp.separable) return build_layer_separable(m, prefix, in_blocks, s_in, L);
  const auto& W3 = m.get(prefix + ".gated_conv.f.f.radial_nn.3.weight", off * H);
  const auto& b3 = m.get(prefix + ".gated_conv.f.f.radial_nn.3.bias", off);
  const double c0 = std::sqrt(1.0 / sum0), c1 = sum1 > 0 ? std::sqrt(3.0 / sum1) : 0.0;
  auto find = [&](int b, int l2, int lo) -> int64_t {
    for (auto& i : ins)
      if (i.b == b && i.l2 == l2 && i.lo == lo) return i.off;
    throw Err(JAMUN_ERR_INVALID, "internal: missing tensor-product instruction");
  };
  std::vector<UEntry> x0e, dote, x0ve, x1e, crosse;
  int in0 = 0, in1 = 0;
  for (size_t b = 0; b < in_blocks.size(); ++b) {
    const InBlock& ib = in_blocks[b];
    for (int u = 0; u < ib.mul; ++u) {
      const double s = s_in[ib.ch0 + u];
      if (ib.l == 0) {
        x0e.push_back({JAMUN_T_X0, 0, ib.xoff + u, find(b, 0, 0) + (int64_t)u * G0, c0 * s});
        if (G1) x0ve.push_back({JAMUN_T_X0V, 0, ib.xoff + u, find(b, 1, 1) + (int64_t)u * G1, c1 * s});
      } else {
        dote.push_back({JAMUN_T_DOT, 0, ib.xoff + 3 * u, find(b, 1, 0) + (int64_t)u * G0, c0 * s});
        x1e.push_back({JAMUN_T_X1C, 0, ib.xoff + 3 * u, find(b, 0, 1) + (int64_t)u * G1, c1 / std::sqrt(3.0) * s});
        crosse.push_back({JAMUN_T_X1C, 1, ib.xoff + 3 * u, find(b, 1, 1) + (int64_t)u * G1,
                          c1 * (double)hp.w3j_111_sign / std::sqrt(2.0) * s});
      }
    }
    if (ib.l == 0) in0 += ib.mul; else in1 += ib.mul;
  }
  L.in0 = in0; L.in1 = in1; L.XSin = in0 + 3 * in1;
  auto chunked = [](const std::vector<UEntry>& v, int type, std::vector<UBlock>& out) {
    for (size_t i = 0; i < v.size(); i += 64) {
      UBlock b; b.type = type;
      b.e.assign(v.begin() + i, v.begin() + std::min(v.size(), i + 64));
      pad_even(b);
      out.push_back(b);
    }
  };
  std::vector<UBlock> blocks0, blocks1;
  chunked(x0e, JAMUN_T_X0, blocks0);
  chunked(dote, JAMUN_T_DOT, blocks0);
  chunked(x0ve, JAMUN_T_X0V, blocks1);
  for (size_t i = 0; i < x1e.size(); i += 32) {
    UBlock b; b.type = JAMUN_T_X1C;
    const size_t hi = std::min(x1e.size(), i + 32);
    b.e.assign(x1e.begin() + i, x1e.begin() + hi);
    b.e.insert(b.e.end(), crosse.begin() + i, crosse.begin() + hi);
    pad_even(b);
    blocks1.push_back(b);
  }
  L.p0 = pack_problem(blocks0, 1, G0, n_slices, JAMUN_KSUB0, W3, b3, H);
  L.p1 = pack_problem(blocks1, 3, G1, n_slices, JAMUN_KSUB1, W3, b3, H);

  bool x0_contig = true;
  for (size_t i = 1; i < x0ve.size(); ++i) x0_contig = x0_contig && x0ve[i].xoff == x0ve[0].xoff + (int)i;
  const int NT0 = (G0 + 31) / 32;

  // ---- destination-grouped VALU-forming kernel (jamun_conv_dg.hip): weights as 64-lane float4 blocks in MFMA operand order
  if (pack_dg && mul0 == 120 && mul1 == 32 && x0e.size() == 120 && dote.size() == 32 && x1e.size() == 32 && crosse.size() == 32 &&
      x0ve.size() == 120 && x0_contig && x0e[0].xoff == 0 && dote[0].xoff == 120) {
    const int n_k = H + 1;
    auto Wk = [&](int k, int64_t p) -> double { return (k < H) ? (double)W3[(size_t)p * H + k] : (double)b3[p]; };
    std::vector<float4> wx((size_t)n_k * 5 * 16 * 64), wd((size_t)n_k * 5 * 4 * 64), wv((size_t)n_k * 2 * 4 * 64), wt((size_t)n_k * 15 * 64);
    for (int k = 0; k < n_k; ++k) {
      for (int t = 0; t < 5; ++t)
        for (int g = 0; g < 16; ++g)
          for (int lane = 0; lane < 64; ++lane) {
            const int hh = lane >> 5, c = lane & 31, col = 32 * t + c;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int st = 0; st < 4; ++st) {
              const int u = 8 * g + 4 * hh + st;
              if (u < 120 && col < G0) v[st] = (float)(Wk(k, x0e[u].wbase + col) * x0e[u].scale);
            }
            wx[(((size_t)k * 5 + t) * 16 + g) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
            if (g < 4) {
              float d[4] = {0.f, 0.f, 0.f, 0.f};
              for (int st = 0; st < 4; ++st) {
                const int u = 8 * g + 4 * hh + st;
                if (col < G0) d[st] = (float)(Wk(k, dote[u].wbase + col) * dote[u].scale);
              }
              wd[(((size_t)k * 5 + t) * 4 + g) * 64 + lane] = make_float4(d[0], d[1], d[2], d[3]);
            }
          }
      for (int lane = 0; lane < 64; ++lane) {
        const int kq = lane >> 4, c = lane & 15;
        for (int ch = 0; ch < 2; ++ch) {
          const int col = 16 * ch + c;
          for (int g = 0; g < 4; ++g) {  // vector planes: kappa = 16 g + 4 kq + st over [x1 (32) | cross (32)]
            float v[4];
            for (int st = 0; st < 4; ++st) {
              const int kap = 16 * g + 4 * kq + st;
              const UEntry& e = kap < 32 ? x1e[kap] : crosse[kap - 32];
              v[st] = (float)(Wk(k, e.wbase + col) * e.scale);
            }
            wv[(((size_t)k * 2 + ch) * 4 + g) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
      for (int g = 0; g < 15; ++g)  // T pre-pass (k_tprod, 32x32x2): lane (c = w', hh), u = 8 g + 4 hh + st over the 120 scalar inputs
        for (int lane = 0; lane < 64; ++lane) {
          const int hh = lane >> 5, c = lane & 31;
          float v[4];
          for (int st = 0; st < 4; ++st) {
            const int u = 8 * g + 4 * hh + st;
            v[st] = (float)(Wk(k, x0ve[u].wbase + c) * x0ve[u].scale);
          }
          wt[((size_t)k * 15 + g) * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    L.dg.wx = dev_upload(wx);
    L.dg.wd = dev_upload(wd);
    L.dg.wv = dev_upload(wv);
    L.dg.wt = dev_upload(wt);
    // f16x3 contraction: the same weights BALANCED by exact powers of two and split into hi + lo halves.  An f16 pair carries 22 bits
    // only while its lo half is a normal number, i.e. within 2^-14 .. 2^-17 of the largest value sharing its scale, and trained
    // checkpoints spread their channels over many octaves (a feature channel that is small has large weights, and the other way
    // round), so one scale per tensor is not enough:
    //   * input channel u: its weight rows (k, u) times 2^-e_u (largest magnitude over k and columns -> [0.5, 1)); the kernels multiply
    //     the feature rows by 2^e_u when they stage them (DgDev::gx, in the layout of a feature row), BEFORE they measure the maxima their
    //     dynamic scales come from — every input then enters with the weight of its contribution.  The vector channel u shares one
    //     exponent over its three blocks (dot, x1, cross); the T pre-pass has its own (gT);
    //   * output column w: times 2^sB_w (largest -> [2^13, 2^14)); undone per column in the kernels' epilogues (cf0 / cf1 / cfT).
    // one block = 64 lanes x 8 halves = the B fragment of one v_mfma_f32_32x32x16_f16 (lane (c, hh): inputs 16 g + 8 hh + j,
    // column 32 t + c) or v_mfma_f32_16x16x32_f16 (lane (c16, kq): kappa = 32 G + 8 kq + j, column 16 ch + c16)
    {
      auto pow2_above = [](double v) { int ex = 0; if (v > 0 && std::isfinite(v)) std::frexp(v, &ex); return std::max(-40, std::min(40, ex)); };  // v < 2^ex
      auto row_exp = [&](const UEntry& e, int ncols) {
        double m = 0;
        for (int k = 0; k < n_k; ++k) for (int col = 0; col < ncols; ++col) m = std::max(m, std::fabs(Wk(k, e.wbase + col) * e.scale));
        return m;
      };
      std::vector<int> e0(120), e1(32), eT(120);
      for (int u = 0; u < 120; ++u) { e0[u] = pow2_above(row_exp(x0e[u], G0)); eT[u] = pow2_above(row_exp(x0ve[u], G1)); }
      for (int u = 0; u < 32; ++u) e1[u] = pow2_above(std::max(row_exp(dote[u], G0), std::max(row_exp(x1e[u], G1), row_exp(crosse[u], G1))));
      auto Wg = [&](const UEntry& e, int ex, int k, int col) { return std::ldexp(Wk(k, e.wbase + col) * e.scale, -ex); };
      std::vector<double> sc0(160, 1.0), sc1(32, 1.0), scT(32, 1.0);  // column scales 2^sB_w
      std::vector<float> cf0(160, 0.f), cf1(32, 0.f), cfT(32, 0.f);   // ... and their inverses for the epilogues
      for (int col = 0; col < G0; ++col) {
        double m = 0;
        for (int k = 0; k < n_k; ++k) {
          for (int u = 0; u < 120; ++u) m = std::max(m, std::fabs(Wg(x0e[u], e0[u], k, col)));
          for (int u = 0; u < 32; ++u) m = std::max(m, std::fabs(Wg(dote[u], e1[u], k, col)));
        }
        const int sB = 14 - pow2_above(m);
        sc0[col] = std::ldexp(1.0, sB); cf0[col] = (float)std::ldexp(1.0, -sB);
      }
      for (int col = 0; col < G1; ++col) {
        double m = 0, mt = 0;
        for (int k = 0; k < n_k; ++k) {
          for (int u = 0; u < 32; ++u) m = std::max(m, std::max(std::fabs(Wg(x1e[u], e1[u], k, col)), std::fabs(Wg(crosse[u], e1[u], k, col))));
          for (int u = 0; u < 120; ++u) mt = std::max(mt, std::fabs(Wg(x0ve[u], eT[u], k, col)));
        }
        const int sB = 14 - pow2_above(m), sT = 14 - pow2_above(mt);
        sc1[col] = std::ldexp(1.0, sB); cf1[col] = (float)std::ldexp(1.0, -sB);
        scT[col] = std::ldexp(1.0, sT); cfT[col] = (float)std::ldexp(1.0, -sT);
      }
      {
        std::vector<float> gx(216), gT(128, 1.f);
        for (int u = 0; u < 120; ++u) { gx[u] = (float)std::ldexp(1.0, e0[u]); gT[u] = (float)std::ldexp(1.0, eT[u]); }
        for (int u = 0; u < 32; ++u) for (int mm = 0; mm < 3; ++mm) gx[120 + 3 * u + mm] = (float)std::ldexp(1.0, e1[u]);
        L.dg.gx = dev_upload(gx); L.dg.gT = dev_upload(gT);
        L.dg.cf0 = dev_upload(cf0); L.dg.cf1 = dev_upload(cf1); L.dg.cfT = dev_upload(cfT);
      }
      L.dg.sB = 0;  // (the column factors carry the weight scales)
      auto pack8 = [&](const double (&v)[8], float4& hi, float4& lo) {
        uint32_t h[4], l[4];
        for (int i = 0; i < 4; ++i) {
          uint16_t h0, l0, h1, l1;
          split_f16(v[2 * i], h0, l0);
          split_f16(v[2 * i + 1], h1, l1);
          h[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
          l[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
        }
        std::memcpy(&hi, h, 16);
        std::memcpy(&lo, l, 16);
      };
      std::vector<float4> wxh((size_t)n_k * 5 * 8 * 2 * 64), wdh((size_t)n_k * 5 * 2 * 2 * 64), wvh((size_t)n_k * 2 * 2 * 2 * 64);
      for (int k = 0; k < n_k; ++k) {
        for (int t = 0; t < 5; ++t)
          for (int lane = 0; lane < 64; ++lane) {
            const int hh = lane >> 5, c = lane & 31, col = 32 * t + c;
            for (int g = 0; g < 8; ++g) {
              double v[8];
              for (int j = 0; j < 8; ++j) {
                const int u = 16 * g + 8 * hh + j;
                v[j] = (u < 120 && col < G0) ? Wg(x0e[u], e0[u], k, col) * sc0[col] : 0.0;
              }
              const size_t b = ((((size_t)k * 5 + t) * 8 + g) * 2) * 64 + lane;
              pack8(v, wxh[b], wxh[b + 64]);
            }
            for (int g = 0; g < 2; ++g) {
              double v[8];
              for (int j = 0; j < 8; ++j) {
                const int u = 16 * g + 8 * hh + j;
                v[j] = col < G0 ? Wg(dote[u], e1[u], k, col) * sc0[col] : 0.0;
              }
              const size_t b = ((((size_t)k * 5 + t) * 2 + g) * 2) * 64 + lane;
              pack8(v, wdh[b], wdh[b + 64]);
            }
          }
        for (int ch = 0; ch < 2; ++ch)
          for (int G = 0; G < 2; ++G)
            for (int lane = 0; lane < 64; ++lane) {
              const int kq = lane >> 4, c = lane & 15, col = 16 * ch + c;
              double v[8];
              for (int j = 0; j < 8; ++j) {
                const int kap = 32 * G + 8 * kq + j;  // input order of the vector planes' A tiles: 2 u + {x1, cross}
                const UEntry& e = (kap & 1) ? crosse[kap >> 1] : x1e[kap >> 1];
                v[j] = Wg(e, e1[kap >> 1], k, col) * sc1[col];
              }
              const size_t b = ((((size_t)k * 2 + ch) * 2 + G) * 2) * 64 + lane;
              pack8(v, wvh[b], wvh[b + 64]);
            }
      }
      // one stream per (hidden unit, matrix wave) in the order the wave consumes it — 34 blocks: four chunks of the scalar inputs
      // (own tile: groups 2c, 2c+1 as hi, lo, hi, lo; then the wave's group of scalar tile 4, w + 4 (c >> 1), when it falls into this
      // chunk — (c & 1) == (w >> 1) — else unused), the dot inputs (own tile groups 0, 1; tile 4: group w for w < 2), the vector
      // planes (column half w >> 1: groups 0, 1): every load is (uniform base of (k, w)) + constant + lane
      std::vector<float4> wh((size_t)n_k * 4 * 34 * 64, make_float4(0.f, 0.f, 0.f, 0.f));
      auto copy_blocks = [&](const std::vector<float4>& src, size_t src_block, size_t dst_block) {  // hi and lo block
        std::copy(src.begin() + src_block * 64, src.begin() + (src_block + 2) * 64, wh.begin() + dst_block * 64);
      };
      for (int k = 0; k < n_k; ++k)
        for (int w = 0; w < 4; ++w) {
          const size_t base = ((size_t)k * 4 + w) * 34;
          for (int c = 0; c < 4; ++c) {
            for (int gi = 0; gi < 2; ++gi) copy_blocks(wxh, (((size_t)k * 5 + w) * 8 + 2 * c + gi) * 2, base + 6 * c + 2 * gi);
            if ((w >> 1) == (c & 1)) copy_blocks(wxh, (((size_t)k * 5 + 4) * 8 + w + 4 * (c >> 1)) * 2, base + 6 * c + 4);
          }
          for (int g = 0; g < 2; ++g) copy_blocks(wdh, (((size_t)k * 5 + w) * 2 + g) * 2, base + 24 + 2 * g);
          if (w < 2) copy_blocks(wdh, (((size_t)k * 5 + 4) * 2 + w) * 2, base + 28);
          for (int G = 0; G < 2; ++G) copy_blocks(wvh, (((size_t)k * 2 + (w >> 1)) * 2 + G) * 2, base + 30 + 2 * G);
        }
      L.dg.wxh = dev_upload(wh);
      // T pre-pass (k_tprod_h): scalar inputs -> vector rows, weights as the A operand of v_mfma_f32_32x32x16_f16
      {
        L.dg.sBt = 0;  // (balanced per input channel (gT) and per column (cfT), as the contraction's weights)
        std::vector<float4> wth((size_t)n_k * 16 * 64);
        for (int k = 0; k < n_k; ++k)
          for (int g = 0; g < 8; ++g)
            for (int lane = 0; lane < 64; ++lane) {
              const int hh = lane >> 5, c = lane & 31;
              double v[8];
              for (int j = 0; j < 8; ++j) {
                const int u = 16 * g + 8 * hh + j;
                v[j] = (u < 120 && c < G1) ? Wg(x0ve[u], eT[u], k, c) * scT[c] : 0.0;
              }
              const size_t b = ((size_t)k * 16 + 2 * g) * 64 + lane;
              pack8(v, wth[b], wth[b + 64]);
            }
        L.dg.wth = dev_upload(wth);
      }
      // jamun_conv_mf.hip: the A operand of the contraction is the ACCUMULATOR of the forming MFMA (lane = destination, registers =
      // channels), so half p of lane (column c, hh) in K-step s2 is input u = 16 s2 + (p & 3) + 8 (p >> 2) + 4 hh of the wave's 32
      // channels.  Same scale 2^sB as the stream of k_conv_dg.  sTw: 2^sTw x (largest column sum of the T weights) < 1, so that
      // T_k = x0 W times 2^(sX + sTw) stays below 2^14 with |x| 2^sX < 2^14.
      {
        double wcs = 0;
        for (int k = 0; k < n_k; ++k)
          for (int c = 0; c < G1; ++c) {
            double cs = 0;
            for (int u = 0; u < 120; ++u) cs += std::fabs(Wg(x0ve[u], e0[u], k, c));  // (the gauge of the conv kernel's scalar channels: |T| <= max|x'| x this)
            wcs = std::max(wcs, cs);
          }
        int exs = 0;
        if (wcs > 0 && std::isfinite(wcs)) std::frexp(wcs, &exs);
        L.dg.sTw = std::max(-40, std::min(40, -exs));
        auto pack8m = [&](const double (&v)[8], float4& hi, float4& lo) { pack8(v, hi, lo); };
        auto u_of = [](int s2, int hh, int p) { return 16 * s2 + (p & 3) + 8 * (p >> 2) + 4 * hh; };
        // stream of hidden unit k: 124 blocks = waves 0..3 (scalar channels 32 w ..: 20 blocks, (hi, lo) per (output tile n, K-step s2)),
        // wave 4 (dot inputs: 20), waves 5..7 (vector plane: x1 inputs 4 blocks, cross inputs 4 — the same for every plane)
        std::vector<float4> wm((size_t)n_k * 124 * 64, make_float4(0.f, 0.f, 0.f, 0.f));
        for (int k = 0; k < n_k; ++k) {
          const size_t kb = (size_t)k * 124 * 64;
          for (int lane = 0; lane < 64; ++lane) {
            const int hh = lane >> 5, c = lane & 31;
            for (int w = 0; w < 5; ++w)
              for (int n = 0; n < 5; ++n)
                for (int s2 = 0; s2 < 2; ++s2) {
                  const int col = 32 * n + c;
                  double v[8];
                  for (int p = 0; p < 8; ++p) {
                    const int u = 32 * w + u_of(s2, hh, p);
                    if (w < 4) v[p] = (u < 120 && col < G0) ? Wg(x0e[u], e0[u], k, col) * sc0[col] : 0.0;
                    else v[p] = col < G0 ? Wg(dote[u - 128], e1[u - 128], k, col) * sc0[col] : 0.0;
                  }
                  const size_t b = kb + (size_t)(20 * w + 2 * (2 * n + s2)) * 64 + lane;
                  pack8m(v, wm[b], wm[b + 64]);
                }
            for (int m = 0; m < 3; ++m)
              for (int part = 0; part < 2; ++part)  // x1 inputs, then cross inputs -> vector rows (32 columns)
                for (int s2 = 0; s2 < 2; ++s2) {
                  double v[8];
                  for (int p = 0; p < 8; ++p) {
                    const UEntry& e = part == 0 ? x1e[u_of(s2, hh, p)] : crosse[u_of(s2, hh, p)];
                    v[p] = c < G1 ? Wg(e, e1[u_of(s2, hh, p)], k, c) * sc1[c] : 0.0;
                  }
                  const size_t b = kb + (size_t)(100 + 8 * m + 4 * part + 2 * s2) * 64 + lane;
                  pack8m(v, wm[b], wm[b + 64]);
                }
          }
        }
        L.dg.wm = dev_upload(wm);
        // tail tiles (k_tail_contract): the vector outputs take x1, cross AND the scalar channels times v_m (no T pre-pass there) in one
        // accumulator, so the three weight blocks share one column scale; 24 blocks per hidden unit: x1 (2 K-steps x hi, lo), cross,
        // then the scalar channel tiles w = 0..3 (input gauge e0: the rows are staged once, with the conv kernel's channel factors)
        {
          std::vector<double> sct(32, 1.0);
          std::vector<float> cf1t(32, 0.f);
          for (int col = 0; col < G1; ++col) {
            double mx = 0;
            for (int k = 0; k < n_k; ++k) {
              for (int u = 0; u < 32; ++u) mx = std::max(mx, std::max(std::fabs(Wg(x1e[u], e1[u], k, col)), std::fabs(Wg(crosse[u], e1[u], k, col))));
              for (int u = 0; u < 120; ++u) mx = std::max(mx, std::fabs(Wg(x0ve[u], e0[u], k, col)));
            }
            const int sB = 14 - pow2_above(mx);
            sct[col] = std::ldexp(1.0, sB); cf1t[col] = (float)std::ldexp(1.0, -sB);
          }
          std::vector<float4> wmt((size_t)n_k * 24 * 64, make_float4(0.f, 0.f, 0.f, 0.f));
          for (int k = 0; k < n_k; ++k)
            for (int lane = 0; lane < 64; ++lane) {
              const int hh = lane >> 5, c = lane & 31;
              for (int g = 0; g < 6; ++g)
                for (int s2 = 0; s2 < 2; ++s2) {
                  double v[8];
                  for (int pp = 0; pp < 8; ++pp) {
                    const int ul = u_of(s2, hh, pp);
                    if (c >= G1) v[pp] = 0.0;
                    else if (g == 0) v[pp] = Wg(x1e[ul], e1[ul], k, c) * sct[c];
                    else if (g == 1) v[pp] = Wg(crosse[ul], e1[ul], k, c) * sct[c];
                    else { const int u = 32 * (g - 2) + ul; v[pp] = u < 120 ? Wg(x0ve[u], e0[u], k, c) * sct[c] : 0.0; }
                  }
                  const size_t b = ((size_t)k * 24 + 4 * g + 2 * s2) * 64 + lane;
                  pack8m(v, wmt[b], wmt[b + 64]);
                }
            }
          L.dg.wmt = dev_upload(wmt);
          L.dg.cf1t = dev_upload(cf1t);
        }
      }
    }
  }

  // ---- initial projector: input-times-weight table (inputs are constant per distinct embedding row) for k_conv_init_v and k_conv_mfi
  if (uniq_rows && G0 <= 32 * NT0 && G1 <= 32) {
    bool scalar_only = true;
    for (auto& ib : in_blocks) scalar_only = scalar_only && ib.l == 0;
    const int U = (int)(uniq_rows->size() / (size_t)row_len);
    const int tt_row = 32 * (NT0 + 1);
    if (scalar_only && U > 0 && (size_t)U * tt_row * (H + 1) * sizeof(float) <= ((size_t)256 << 20)) {
      std::vector<float> tt((size_t)(H + 1) * U * tt_row, 0.f);
      for (int k = 0; k <= H; ++k)
        for (int uid = 0; uid < U; ++uid) {
          const float* xr = uniq_rows->data() + (size_t)uid * row_len;
          float* out = tt.data() + ((size_t)k * U + uid) * tt_row;
          for (int w = 0; w < G0; ++w) {
            double acc = 0;
            for (const UEntry& e : x0e) {
              const int64_t p = e.wbase + w;
              acc += (double)xr[e.xoff] * ((k < H) ? (double)W3[(size_t)p * H + k] : (double)b3[p]) * e.scale;
            }
            out[w] = (float)acc;
          }
          for (int w = 0; w < G1; ++w) {
            double acc = 0;
            for (const UEntry& e : x0ve) {
              const int64_t p = e.wbase + w;
              acc += (double)xr[e.xoff] * ((k < H) ? (double)W3[(size_t)p * H + k] : (double)b3[p]) * e.scale;
            }
            out[32 * NT0 + w] = (float)acc;
          }
        }
      L.tt_U = U;
      if (NT0 == 5 && G0 <= 152 && G1 <= 32) {  // scalar columns 0..127 as they are, then per lane u (column 128+u, vector column u)
        std::vector<float> tt2((size_t)(H + 1) * U * 192, 0.f);
        for (int k = 0; k <= H; ++k)
          for (int uid = 0; uid < U; ++uid) {
            const float* in = tt.data() + ((size_t)k * U + uid) * tt_row;
            float* out = tt2.data() + ((size_t)k * U + uid) * 192;
            for (int c = 0; c < 128; ++c) out[c] = in[c];
            for (int u = 0; u < 32; ++u) {
              out[128 + 2 * u] = u < 24 ? in[128 + u] : 0.f;
              out[128 + 2 * u + 1] = in[32 * NT0 + u];
            }
          }
        L.tt2 = dev_upload(tt2);
      }
      if (NT0 == 5 && G0 <= 160 && G1 <= 32 && U <= 128) {
        const int UT = U <= 32 ? 1 : (U <= 64 ? 2 : 4);
        L.tab_ut = UT;
        // k_conv_mfi: blocks 4 r + 2 s2 + {hi, lo}; half p of lane (column c, hh) <-> uid 16 s2 + (p & 3) + 8 (p >> 2) + 4 hh (the
        // accumulator layout of the forming MFMA); r < 5: scalar-output columns 32 r + c, r = 5: the vector columns
        double tmax = 0;
        for (float v : tt) tmax = std::max(tmax, (double)std::fabs(v));
        int ex = 0;
        if (tmax > 0 && std::isfinite(tmax)) std::frexp(tmax, &ex);
        L.tab_sB = std::max(-40, std::min(40, 14 - ex));
        const double sc = std::ldexp(1.0, L.tab_sB);
        std::vector<float4> tw((size_t)(H + 1) * 6 * 4 * UT * 64, make_float4(0.f, 0.f, 0.f, 0.f));
        for (int k = 0; k <= H; ++k)
          for (int r = 0; r < 6; ++r)
            for (int ts = 0; ts < 2 * UT; ++ts)
              for (int lane = 0; lane < 64; ++lane) {
                const int s2 = ts & 1, ut = ts >> 1;
                const int hh = lane >> 5, c = lane & 31;
                const int col = r < 5 ? 32 * r + c : 32 * NT0 + c;
                const bool col_ok = r < 5 ? col < G0 : c < G1;
                uint32_t h[4], l[4];
                for (int i = 0; i < 4; ++i) {
                  uint16_t hp[2], lp[2];
                  for (int e = 0; e < 2; ++e) {
                    const int pp = 2 * i + e, uid = 32 * ut + 16 * s2 + (pp & 3) + 8 * (pp >> 2) + 4 * hh;
                    const double v = (col_ok && uid < U) ? (double)tt[((size_t)k * U + uid) * tt_row + col] : 0.0;
                    split_f16(v * sc, hp[e], lp[e]);
                  }
                  h[i] = (uint32_t)hp[0] | ((uint32_t)hp[1] << 16);
                  l[i] = (uint32_t)lp[0] | ((uint32_t)lp[1] << 16);
                }
                const size_t b = (((size_t)k * 6 + r) * 4 * UT + 4 * ut + 2 * s2) * 64 + lane;
                std::memcpy(&tw[b], h, 16);
                std::memcpy(&tw[b + 64], l, 16);
              }
        L.tabw = dev_upload(tw);
      }
    }
  }

  // ---- k_conv_mfx: the initial projector formed from the feature rows themselves (scalar inputs only, at most 64 channels): weights
  // balanced per input channel (2^-e_u; the factor goes into the stored rows) and per output column (2^sB_w, undone by xcf0 / xcf1), the
  // rows x 2^e_u x 2^x_sX (ONE static scale: the rows are constants of (topology, sigma)) split hi + lo on the host, two atoms per word
  if (all_rows && row_len > 0 && row_len <= 64 && in1 == 0 && NT0 == 5 && G1 <= 32 && G1 > 0 && (int)x0e.size() == row_len && (int)x0ve.size() == row_len) {
    const int n_k = H + 1, C = row_len;
    auto Wk = [&](int k, int64_t p) -> double { return (k < H) ? (double)W3[(size_t)p * H + k] : (double)b3[p]; };
    auto pow2_above = [](double v) { int ex = 0; if (v > 0 && std::isfinite(v)) std::frexp(v, &ex); return std::max(-40, std::min(40, ex)); };
    std::vector<const UEntry*> es(64, nullptr), ev(64, nullptr);  // by feature column (xoff)
    for (const UEntry& e : x0e) es[e.xoff] = &e;
    for (const UEntry& e : x0ve) ev[e.xoff] = &e;
    std::vector<int> eu(64, 0);
    for (int u = 0; u < C; ++u) {
      double mx = 0;
      for (int k = 0; k < n_k; ++k) {
        for (int col = 0; col < G0; ++col) mx = std::max(mx, std::fabs(Wk(k, es[u]->wbase + col) * es[u]->scale));
        for (int col = 0; col < G1; ++col) mx = std::max(mx, std::fabs(Wk(k, ev[u]->wbase + col) * ev[u]->scale));
      }
      eu[u] = pow2_above(mx);
    }
    auto Wg = [&](const UEntry* e, int u, int k, int col) { return e ? std::ldexp(Wk(k, e->wbase + col) * e->scale, -eu[u]) : 0.0; };
    std::vector<double> sc0(160, 1.0), sc1(32, 1.0);
    std::vector<float> cf0(160, 0.f), cf1(32, 0.f);
    for (int col = 0; col < G0; ++col) {
      double mx = 0;
      for (int k = 0; k < n_k; ++k) for (int u = 0; u < C; ++u) mx = std::max(mx, std::fabs(Wg(es[u], u, k, col)));
      const int sB = 14 - pow2_above(mx);
      sc0[col] = std::ldexp(1.0, sB); cf0[col] = (float)std::ldexp(1.0, -sB);
    }
    for (int col = 0; col < G1; ++col) {
      double mx = 0;
      for (int k = 0; k < n_k; ++k) for (int u = 0; u < C; ++u) mx = std::max(mx, std::fabs(Wg(ev[u], u, k, col)));
      const int sB = 14 - pow2_above(mx);
      sc1[col] = std::ldexp(1.0, sB); cf1[col] = (float)std::ldexp(1.0, -sB);
    }
    auto pack8 = [&](const double (&v)[8], float4& hi, float4& lo) {
      uint32_t h[4], l[4];
      for (int i = 0; i < 4; ++i) {
        uint16_t h0, l0, h1, l1;
        split_f16(v[2 * i], h0, l0);
        split_f16(v[2 * i + 1], h1, l1);
        h[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        l[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
      }
      std::memcpy(&hi, h, 16);
      std::memcpy(&lo, l, 16);
    };
    auto u_of = [](int t, int s2, int hh, int p) { return 32 * t + 16 * s2 + (p & 3) + 8 * (p >> 2) + 4 * hh; };
    std::vector<float4> wx((size_t)n_k * 48 * 64, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int k = 0; k < n_k; ++k)
      for (int lane = 0; lane < 64; ++lane) {
        const int hh = lane >> 5, c = lane & 31;
        for (int t = 0; t < 2; ++t)
          for (int s2 = 0; s2 < 2; ++s2) {
            for (int n = 0; n < 5; ++n) {
              const int col = 32 * n + c;
              double v[8];
              for (int pp = 0; pp < 8; ++pp) {
                const int u = u_of(t, s2, hh, pp);
                v[pp] = (u < C && col < G0) ? Wg(es[u], u, k, col) * sc0[col] : 0.0;
              }
              const size_t b = ((size_t)k * 48 + 20 * t + 2 * (2 * n + s2)) * 64 + lane;
              pack8(v, wx[b], wx[b + 64]);
            }
            double v[8];
            for (int pp = 0; pp < 8; ++pp) {
              const int u = u_of(t, s2, hh, pp);
              v[pp] = (u < C && c < G1) ? Wg(ev[u], u, k, c) * sc1[c] : 0.0;
            }
            const size_t b = ((size_t)k * 48 + 40 + 4 * t + 2 * s2) * 64 + lane;
            pack8(v, wx[b], wx[b + 64]);
          }
      }
    const size_t N = all_rows->size() / (size_t)row_len;
    double xm = 0;
    for (size_t i = 0; i < N; ++i)
      for (int u = 0; u < C; ++u) xm = std::max(xm, std::fabs(std::ldexp((double)(*all_rows)[i * row_len + u], eu[u])));
    L.x_sX = 14 - pow2_above(xm);
    const size_t n_pairs = (N + 1) / 2 + 96;  // a window reads up to 88 pairs (k_conv_mlx; k_conv_mfx: 32) from the pair of its first atom: zero rows behind the batch
    std::vector<unsigned> xph(n_pairs * 64, 0u), xpl(n_pairs * 64, 0u);
    for (size_t i = 0; i < N; ++i)
      for (int u = 0; u < C; ++u) {
        uint16_t hi, lo;
        split_f16(std::ldexp((double)(*all_rows)[i * row_len + u], eu[u] + L.x_sX), hi, lo);
        const size_t w = (i >> 1) * 64 + u;
        const int sh = (i & 1) ? 16 : 0;
        xph[w] |= (unsigned)hi << sh;
        xpl[w] |= (unsigned)lo << sh;
      }
    L.wx = dev_upload(wx);
    L.xph = dev_upload(xph); L.xpl = dev_upload(xpl);
    L.xcf0 = dev_upload(cf0); L.xcf1 = dev_upload(cf1);
  }

  build_layer_common(m, prefix, in_blocks, s_in, L, in0, in1);
  return L;
}

struct ProfScope {
  jamun_sampler* s; int cls; hipStream_t st; int b = -1;
  ProfScope(jamun_sampler* s_, int cls_, hipStream_t st_) : s(s_), cls(cls_), st(st_) {
    if (!(s->prof_mask >> cls & 1u)) return;
    if (s->prof_seen[cls]++ % s->prof_every != 0) return;
    while (s->ev_pool.size() < s->ev_next + 2) {
      hipEvent_t e;
      HIPCHECK(hipEventCreate(&e));
      s->ev_pool.push_back(e);
    }
    b = (int)s->ev_next;
    s->ev_next += 2;
    HIPCHECK(hipEventRecord(s->ev_pool[b], st));
  }
  ~ProfScope() {
    if (b < 0) return;
    (void)hipEventRecord(s->ev_pool[b + 1], st);
    s->ev_used.push_back({cls, {b, b + 1}});
  }
};

// One denoiser forward.  `pre` / `post`: the two halves of a BAOAB iteration fused into the first and the last kernel of the
// forward (y is then advanced in place before the geometry is built).
// Geometry of one forward: centring, radius graph + bonded edges, unit vectors and distances (k_geom; `pre`: the first half of a
// BAOAB iteration fused in front of it), and — when the buffer holds all layers — the radial MLPs' hidden activations of every
// layer (k_edge_h; they depend on the geometry only).
void build_edges(jamun_sampler* s, float* y, hipStream_t st, const LangevinPre& pre = LangevinPre(), bool geom_done = false) {
  if (!geom_done) {  // (geom_done: the previous walk iteration's last launch, k_finalize_geom, already advanced y and built the edge table)
    ProfScope ps(s, JAMUN_PROF_GEOM, st);
    launch_geom(y, s->ptr, s->n_graphs, s->c_in, s->r2, s->S, s->bond_in_ptr, s->bond_in_src, s->hp.mean_center, s->yc,
                s->deg, s->esrc, s->egeo, s->epair, pre, st);
  }
  if (s->h_batched) {
    ProfScope ps(s, JAMUN_PROF_EDGE_H, st);
    launch_edge_h(s->deg, s->esrc, s->egeo, s->n_atoms, s->S, s->w1r_all, s->cmask_all, (int)s->layers.size(), s->mu, s->rb_step, s->h,
                  s->h_stride, s->h_kstride, st, s->w1h_all, s->w1isc_all);
  }
  s->edges_built = true;
}

// One block of the network on the current edge table: ConvBlock l (conv contraction + gate + self-interaction + skip Linear)
// and, for the hidden layers, the noise-conditional input scaling and skip mix around it (e3conv.py:129-133).
void run_layer(jamun_sampler* s, size_t l, const float* x_in, int XSin, float* x_out, hipStream_t st) {
  LayerDev& L = s->layers[l];
  const float* h_l = s->h + (s->h_batched ? l * s->h_stride : 0);
  if (!s->h_batched) {  // (batches above 4 GiB of activations: one layer's radial MLP at a time)
    ProfScope ps(s, JAMUN_PROF_EDGE_H, st);
    launch_edge_h(s->deg, s->esrc, s->egeo, s->n_atoms, s->S, s->w1r_all + l * 64 * 32, s->cmask_all + l * 128, 1, s->mu, s->rb_step, s->h,
                  s->h_stride, s->h_kstride, st, s->w1h_all ? s->w1h_all + l * 8 * 64 : nullptr, s->w1isc_all ? s->w1isc_all + l : nullptr);
  }
    if (L.sep.w2b) {
      SepArgs f{};
      f.deg = s->deg; f.esrc = s->esrc; f.egeo = s->egeo; f.h = h_l; f.h_kstride = s->h_kstride; f.x = x_in;
      f.n_atoms = s->n_atoms; f.S = s->S; f.XS = XSin;
      f.n0 = L.sep.n0; f.n1 = L.sep.n1; f.w2b = L.sep.w2b; f.cfw = L.sep.cfw; f.bias = L.sep.bias; f.sH = L.sep.sH; f.D = s->sep_D;
      f.wl0 = L.sep.wl0; f.wl1 = L.sep.wl1;
      f.G0 = s->hp.mul0 + s->hp.mul1; f.G1 = s->hp.mul1; f.nt0 = L.p0.nt; f.nt1 = L.p1.nt;
      f.partial0 = s->partial0; f.partial1 = s->partial1;
      ProfScope ps(s, l == 0 ? JAMUN_PROF_CONV0_INIT : JAMUN_PROF_CONV0, st);
      if (launch_sep_conv(f, s->cus, st) != 0) throw Err(JAMUN_ERR_INVALID, "separable conv launch failed (irreps not supported)");
    } else if (l == 0 && s->mfi_on) {
      MfiArgs f{};
      f.deg = s->deg; f.epair = s->epair; f.esrc = s->esrc; f.egeo = s->egeo; f.h = h_l; f.h_kstride = s->h_kstride;
      f.n_pad = s->n_pad; f.S = s->S; f.nt0 = L.p0.nt;
      f.tile_span = s->dg_tile_span; f.tile_atoms = s->dg_tile_atoms; f.segs = s->init_segs ? s->init_segs : s->dg_segs; f.max_segs = s->init_segs ? s->init_max_segs : s->dg_max_segs;
      f.atom_uid = s->atom_uid; f.tabw = L.tabw; f.sB = L.tab_sB; f.ut = L.tab_ut;
      {
        int e3 = 0;
        std::frexp(1.5 * (double)L.dg.hmax2, &e3);
        f.sC = std::max(-40, std::min(40, 14 - e3));
      }
      f.partial0 = s->partial0; f.partial1 = s->partial1; f.err = s->mf_err;
      {
        ProfScope ps(s, JAMUN_PROF_CONV0_INIT, st);
        if (launch_conv_mfi(f, s->dg_grid, st) != 0) throw Err(JAMUN_ERR_INVALID, "initial-projector conv launch failed (configuration not supported)");
      }
      if (s->init_tail) {
        TailArgs t{};
        t.deg = s->deg; t.epair = s->epair; t.esrc = s->esrc; t.egeo = s->egeo; t.h = h_l; t.h_kstride = s->h_kstride;
        t.n_pad = s->n_pad; t.S = s->S; t.n_k = s->hp.edge_attr_dim + 1; t.nt0 = L.p0.nt;
        t.tile_span = s->dg_tile_span; t.tile_atoms = s->dg_tile_atoms; t.tail_tiles = s->tail_tiles;
        t.n_tail_tiles = s->n_tail_tiles; t.n_tail = s->n_tail; t.n_runs = s->tail_runs; t.tail_atom = s->tail_atom; t.tail_scale = s->tail_scale;
        t.P = s->tail_P; t.sC = f.sC; t.xph = L.xph; t.xpl = L.xpl; t.wx = L.wx; t.sX = L.x_sX; t.cf0 = L.xcf0; t.cf1t = L.xcf1;
        t.partial0 = s->partial0; t.partial1 = s->partial1; t.err = s->mf_err;
        ProfScope pt(s, JAMUN_PROF_CONV1_INIT, st);
        if (launch_conv_tail_init(t, st) != 0) throw Err(JAMUN_ERR_INVALID, "tail-tile conv launch failed (configuration not supported)");
      }
    } else if (l == 0 && s->mfx_on) {
      MfxArgs f{};
      f.deg = s->deg; f.epair = s->epair; f.esrc = s->esrc; f.egeo = s->egeo; f.h = h_l; f.h_kstride = s->h_kstride;
      f.n_pad = s->n_pad; f.S = s->S; f.nt0 = L.p0.nt;
      f.tile_span = s->dg_tile_span; f.tile_atoms = s->dg_tile_atoms; f.segs = s->init_segs ? s->init_segs : s->dg_segs; f.max_segs = s->init_segs ? s->init_max_segs : s->dg_max_segs;
      f.xph = L.xph; f.xpl = L.xpl; f.wx = L.wx; f.sX = L.x_sX; f.cf0 = L.xcf0; f.cf1 = L.xcf1;
      {
        int e3 = 0;
        std::frexp(1.5 * (double)L.dg.hmax2, &e3);
        f.sC = std::max(-40, std::min(40, 14 - e3));
      }
      f.partial0 = s->partial0; f.partial1 = s->partial1; f.err = s->mf_err;
      {
        ProfScope ps(s, JAMUN_PROF_CONV0_INIT, st);
        if (launch_conv_mfx(f, s->dg_grid, st) != 0) throw Err(JAMUN_ERR_INVALID, "initial-projector conv launch failed (configuration not supported)");
      }
      if (s->init_tail) {
        TailArgs t{};
        t.deg = s->deg; t.epair = s->epair; t.esrc = s->esrc; t.egeo = s->egeo; t.h = h_l; t.h_kstride = s->h_kstride;
        t.n_pad = s->n_pad; t.S = s->S; t.n_k = s->hp.edge_attr_dim + 1; t.nt0 = L.p0.nt;
        t.tile_span = s->dg_tile_span; t.tile_atoms = s->dg_tile_atoms; t.tail_tiles = s->tail_tiles;
        t.n_tail_tiles = s->n_tail_tiles; t.n_tail = s->n_tail; t.n_runs = s->tail_runs; t.tail_atom = s->tail_atom; t.tail_scale = s->tail_scale;
        t.P = s->tail_P; t.sC = f.sC; t.xph = L.xph; t.xpl = L.xpl; t.wx = L.wx; t.sX = L.x_sX; t.cf0 = L.xcf0; t.cf1t = L.xcf1;
        t.partial0 = s->partial0; t.partial1 = s->partial1; t.err = s->mf_err;
        ProfScope pt(s, JAMUN_PROF_CONV1_INIT, st);
        if (launch_conv_tail_init(t, st) != 0) throw Err(JAMUN_ERR_INVALID, "tail-tile conv launch failed (configuration not supported)");
      }
    } else if (l == 0 && s->mlx_on) {
      MlxArgs f{};
      f.deg = s->deg; f.epair = s->epair; f.esrc = s->esrc; f.egeo = s->egeo; f.h = h_l; f.h_kstride = s->h_kstride;
      f.n_pad = s->n_pad; f.S = s->S; f.nt0 = L.p0.nt; f.window = s->ml_window;
      f.tile_span = s->dg_tile_span; f.tile_atoms = s->dg_tile_atoms; f.segs = s->dg_segs; f.max_segs = s->dg_max_segs;
      f.xph = L.xph; f.xpl = L.xpl; f.wx = L.wx; f.sX = L.x_sX; f.cf0 = L.xcf0; f.cf1 = L.xcf1;
      {
        int e3 = 0;
        std::frexp(1.5 * (double)L.dg.hmax2, &e3);
        f.sC = std::max(-40, std::min(40, 14 - e3));
      }
      f.partial0 = s->partial0; f.partial1 = s->partial1; f.err = s->mf_err; f.mfma_count = nullptr;
      ProfScope ps(s, JAMUN_PROF_CONV0_INIT, st);
      if (launch_conv_mlx(f, s->dg_grid, st) != 0) throw Err(JAMUN_ERR_INVALID, "initial-projector conv launch failed (configuration not supported)");
    } else if (l == 0 && s->initv_on) {
      InitVArgs f{};
      f.deg = s->deg; f.esrc = s->esrc; f.egeo = s->egeo; f.h = h_l; f.h_kstride = s->h_kstride;
      f.n_pad = s->n_pad; f.S = s->S; f.PMAX = (s->S + 3) & ~3; f.RS = s->dg_RS; f.nt0 = L.p0.nt; f.nbuf = s->initv_nbuf;
      f.tile_span = s->dg_tile_span; f.tile_atoms = s->dg_tile_atoms; f.segs = s->init_segs ? s->init_segs : s->dg_segs; f.max_segs = s->init_segs ? s->init_max_segs : s->dg_max_segs;
      f.atom_uid = s->atom_uid; f.tt2 = L.tt2; f.tt2_kstride = (size_t)L.tt_U * 192;
      f.dbg = 0;
      f.partial0 = s->partial0; f.partial1 = s->partial1;
      ProfScope ps(s, JAMUN_PROF_CONV0_INIT, st);
      if (launch_conv_initv(f, s->dg_grid, st) != 0) throw Err(JAMUN_ERR_INVALID, "initial-projector conv launch failed (configuration not supported)");
    } else if (l > 0 && s->dg_on && s->dg_mode == 4) {
      MfArgs f{};
      f.deg = s->deg; f.esrc = s->esrc; f.egeo = s->egeo; f.h = h_l; f.h_kstride = s->h_kstride; f.x = x_in;
      f.n_pad = s->n_pad; f.S = s->S; f.XS = XSin; f.n_atoms = s->n_atoms;
      f.tile_span = s->dg_tile_span; f.tile_atoms = s->dg_tile_atoms; f.segs = s->dg_segs; f.max_segs = s->dg_max_segs; f.nt0 = L.p0.nt;
      f.wm = L.dg.wm; f.Tt = s->dg_T; f.t_stride = s->dg_tstride; f.sB = L.dg.sB; f.sTw = L.dg.sTw; f.nks = s->mf_nks; f.epair = s->epair;
      f.gx = L.dg.gx; f.cf0 = L.dg.cf0; f.cf1 = L.dg.cf1;
      {
        int e3 = 0;
        std::frexp(1.5 * (double)L.dg.hmax2, &e3);  // 3 max|h~| < 2^e3
        f.sC = std::max(-40, std::min(40, 14 - e3));
      }
      f.partial0 = s->partial0; f.partial1 = s->partial1; f.err = s->mf_err; f.x1 = s->x1;
      {
        ProfScope pt(s, JAMUN_PROF_TPROD, st);
        launch_tprod(x_in, XSin, s->n_atoms, s->hp.edge_attr_dim + 1, L.dg.wt, L.dg.wth, L.dg.gT, L.dg.cfT, s->dg_T, s->dg_tstride, st, s->tune.no_tprod_t != 0);
      }
      {
        ProfScope ps(s, JAMUN_PROF_CONV0, st);
        if (launch_conv_mf(f, s->dg_grid, st) != 0) throw Err(JAMUN_ERR_INVALID, "matrix-formed conv launch failed (configuration not supported)");
      }
      if (s->n_tail_tiles) {
        TailArgs t{};
        t.deg = s->deg; t.epair = s->epair; t.esrc = s->esrc; t.egeo = s->egeo; t.h = h_l; t.h_kstride = s->h_kstride; t.x = x_in;
        t.n_pad = s->n_pad; t.S = s->S; t.XS = XSin; t.n_k = s->hp.edge_attr_dim + 1; t.nt0 = L.p0.nt;
        t.tile_span = s->dg_tile_span; t.tile_atoms = s->dg_tile_atoms; t.tail_tiles = s->tail_tiles;
        t.n_tail_tiles = s->n_tail_tiles; t.n_tail = s->n_tail; t.n_runs = s->tail_runs; t.tail_atom = s->tail_atom; t.tail_scale = s->tail_scale;
        t.P = s->tail_P; t.gx = L.dg.gx; t.sC = f.sC; t.wm = L.dg.wm; t.wmt = L.dg.wmt; t.cf0 = L.dg.cf0; t.cf1t = L.dg.cf1t;
        t.partial0 = s->partial0; t.partial1 = s->partial1; t.err = s->mf_err;
        ProfScope pt(s, JAMUN_PROF_CONV1, st);
        if (launch_conv_tail(t, st) != 0) throw Err(JAMUN_ERR_INVALID, "tail-tile conv launch failed (configuration not supported)");
      }
    } else if (l > 0 && s->dg_on && s->dg_mode == 5) {
      MlArgs f{};
      f.deg = s->deg; f.epair = s->epair; f.esrc = s->esrc; f.egeo = s->egeo; f.h = h_l; f.h_kstride = s->h_kstride; f.x = x_in;
      f.n_pad = s->n_pad; f.S = s->S; f.XS = XSin; f.n_atoms = s->n_atoms;
      f.tile_span = s->dg_tile_span; f.tile_atoms = s->dg_tile_atoms; f.segs = s->dg_segs; f.max_segs = s->dg_max_segs; f.nt0 = L.p0.nt;
      f.window = s->ml_window;
      f.wm = L.dg.wm; f.Tt = s->dg_T; f.t_stride = s->dg_tstride; f.sB = L.dg.sB; f.sTw = L.dg.sTw;
      f.gx = L.dg.gx; f.cf0 = L.dg.cf0; f.cf1 = L.dg.cf1;
      {
        int e3 = 0;
        std::frexp(1.5 * (double)L.dg.hmax2, &e3);  // 3 max|h~| < 2^e3
        f.sC = std::max(-40, std::min(40, 14 - e3));
      }
      f.partial0 = s->partial0; f.partial1 = s->partial1; f.err = s->mf_err; f.mfma_count = s->ml_count; f.x1 = s->x1;
      {
        ProfScope pt(s, JAMUN_PROF_TPROD, st);
        launch_tprod(x_in, XSin, s->n_atoms, s->hp.edge_attr_dim + 1, L.dg.wt, L.dg.wth, L.dg.gT, L.dg.cfT, s->dg_T, s->dg_tstride, st, s->tune.no_tprod_t != 0);
      }
      {
        ProfScope ps(s, JAMUN_PROF_CONV0, st);
        if (launch_conv_ml(f, s->dg_grid, st) != 0) throw Err(JAMUN_ERR_INVALID, "large-span matrix-formed conv launch failed (configuration not supported)");
        ++s->ml_launches;
      }
    } else if (l > 0 && s->dg_on) {
      DgArgs f{};
      f.deg = s->deg; f.esrc = s->esrc; f.egeo = s->egeo; f.h = h_l; f.h_kstride = s->h_kstride; f.x = x_in;
      f.n_pad = s->n_pad; f.S = s->S; f.XS = XSin; f.RS = s->dg_RS; f.PMAX = (s->S + 3) & ~3;  // multiple of the forming batch
      f.tile_span = s->dg_tile_span; f.tile_atoms = s->dg_tile_atoms; f.segs = s->dg_segs; f.max_segs = s->dg_max_segs;
      f.row_blocks = s->dg_row_blocks ? 1 : 0; f.nt0 = L.p0.nt; f.alt = s->dg_mode;
      f.wx = L.dg.wx; f.wd = L.dg.wd; f.wv = L.dg.wv; f.T = s->dg_T; f.n_atoms = s->n_atoms;
      f.emu = s->dg_emu; f.wh = L.dg.wxh; f.sB = L.dg.sB; f.hmax2 = L.dg.hmax2;
      f.gx = L.dg.gx; f.cf0 = L.dg.cf0; f.cf1 = L.dg.cf1;
      f.partial0 = s->partial0; f.partial1 = s->partial1;
      f.dbg = 0;
      f.dump = nullptr;
      {
        ProfScope pt(s, JAMUN_PROF_TPROD, st);
        launch_tprod(x_in, XSin, s->n_atoms, s->hp.edge_attr_dim + 1, L.dg.wt, s->dg_emu ? L.dg.wth : nullptr, L.dg.gT, L.dg.cfT, s->dg_T, 0, st);
      }
      ProfScope ps(s, JAMUN_PROF_CONV0, st);
      const int rcode = launch_conv_dg(f, s->dg_grid, st);
      if (rcode != 0) throw Err(JAMUN_ERR_INVALID, "destination-grouped conv launch failed (configuration not supported)");
    } else {
    ConvArgs a{};
    a.deg = s->deg; a.esrc = s->esrc; a.egeo = s->egeo; a.h = h_l; a.h_kstride = s->h_kstride; a.x = x_in;
    a.n_atoms = s->n_atoms; a.n_pad = s->n_pad; a.n_tiles = s->n_tiles; a.S = s->S; a.S4 = (s->S + 3) & ~3; a.XS = XSin;
    a.n_slices = s->n_slices;
    for (int pi = 0; pi < 2; ++pi) {
      ConvProblemDev& P = pi == 0 ? L.p0 : L.p1;
      if (P.nt == 0) continue;
      a.wpack = P.wpack; a.chunks = P.chunks; a.slice_ptr = P.slice_ptr; a.ublk = P.ublk; a.lane_xoff = P.lane_xoff;
      a.partial = pi == 0 ? s->partial0 : s->partial1;
      ProfScope ps(s, l == 0 ? (pi == 0 ? JAMUN_PROF_CONV0_INIT : JAMUN_PROF_CONV1_INIT) : (pi == 0 ? JAMUN_PROF_CONV0 : JAMUN_PROF_CONV1), st);
      const int rcode = launch_conv(a, P.planes, P.nt, st);
      if (rcode == -2) throw Err(JAMUN_ERR_INVALID, "walker batch needs more than 160 KiB of LDS per conv workgroup (edge stride too large)");
      if (rcode != 0) throw Err(JAMUN_ERR_INVALID, "unsupported conv tile configuration");
    }
    }
    NodeArgs n{};
    n.partial0 = s->partial0; n.partial1 = s->partial1; n.deg = s->deg; n.x_in = x_in; n.x_out = x_out;
    n.wcat0 = L.wcat0; n.wcat1 = L.wcat1; n.K0p = L.K0p; n.K1p = L.K1p; n.mix = L.mix;
    n.cL = s->hp.act_scalar_const; n.cS = s->hp.act_gate_const;
    n.n_atoms = s->n_atoms; n.n_pad = s->n_pad; n.n_slices = s->n_slices; n.nt0 = L.p0.nt; n.nt1 = L.p1.nt;
    n.mul0 = s->hp.mul0; n.mul1 = s->hp.mul1; n.in0 = L.in0; n.in1 = L.in1; n.XSin = XSin;
    const bool dg_layer = (l > 0 && s->dg_on) || (l == 0 && (s->initv_on || s->mfi_on || s->mfx_on || s->mlx_on));  // (slabs of the dg tile plan)
    const bool init_plan = l == 0 && dg_layer && s->init_segs != nullptr;  // (the initial projector keeps the tail tiles on its own segment lists)
    n.atom_nslab = dg_layer ? (init_plan ? s->init_atom_nslab : s->dg_atom_nslab) : nullptr;
    n.max_slabs = dg_layer ? (init_plan ? s->init_n_slabs : s->dg_n_slabs) : s->n_slices;
    if (L.sep.w2b) { n.n_slices = 1; n.atom_nslab = nullptr; n.max_slabs = 1; }  // SeparableConv writes the summed messages as ONE slab
    n.wh0 = L.wh0; n.wh1 = L.wh1; n.K0h = L.K0h; n.K1h = L.K1h;
    n.kga0 = L.kga0; n.kga1 = L.kga1; n.kgx = L.kgx; n.cg0 = L.cg0; n.cg1 = L.cg1;
    {
      ProfScope ps(s, JAMUN_PROF_NODE, st);
      if (!s->tune.node_fp32 && node_update_h_supported(n)) launch_node_update_h(n, s->cus, st);  // (node_fp32: the v_mfma_f32_32x32x2_f32 kernel, A/B aid)
      else launch_node_update(n, st);
    }
}

// k_conv_mf / k_conv_mfi set a device flag when an ordered (source, destination) pair carries more edges than one coefficient
// entry can hold (the host plan excludes such topologies at create time; the kernels still report what they see).  Every entry
// point that ran a forward copies the flag into a pinned word behind its work (no synchronisation); every entry point first looks
// at that word, so a disagreement between plan and kernel surfaces as JAMUN_ERR_INVALID at the next call after the copy landed —
// at the latest in jamun_sampler_stats, which synchronises — instead of as silently wrong coordinates.
void mf_err_check(jamun_sampler* s) {
  if (s->mf_err_host && *(volatile int*)s->mf_err_host != 0)
    throw Err(JAMUN_ERR_INVALID, (*(volatile int*)s->mf_err_host & 2)
                                     ? "k_conv_mf: an edge's source lies outside the rows the selected instantiation multiplies (host plan and kernel disagree) — results of this sampler are invalid"
                                     : "k_conv_mf: more than three edges of one (source, destination) pair — results of this sampler are invalid");
}
void mf_err_fetch(jamun_sampler* s, hipStream_t st) {
  if (s->mf_err && s->mf_err_host) HIPCHECK(hipMemcpyAsync(s->mf_err_host, s->mf_err, sizeof(int), hipMemcpyDeviceToHost, st));
}

// One denoiser forward.  `pre` / `post`: the two halves of a BAOAB iteration fused into the first and the last kernel of the
// forward (y is then advanced in place before the geometry is built).
void forward(jamun_sampler* s, float* y, float* xhat, float* score, hipStream_t st, const LangevinPre& pre = LangevinPre(),
             const LangevinPost& post = LangevinPost(), const LangevinPre* next_pre = nullptr, bool geom_done = false) {
  build_edges(s, y, st, pre, geom_done);
  const float* x_in = s->x_emb;
  int XSin = s->n_emb;
  for (size_t l = 0; l < s->layers.size(); ++l) {
    run_layer(s, l, x_in, XSin, s->x[l], st);
    x_in = s->x[l];
    XSin = s->XS;
  }
  HeadArgs hd{};
  hd.x = x_in; hd.w_gate = s->w_gate; hd.w_vec = s->w_vec; hd.w_out = s->w_out; hd.g = s->g;
  hd.cS = s->hp.act_gate_const; hd.n_atoms = s->n_atoms; hd.mul0 = s->hp.mul0; hd.mul1 = s->hp.mul1;
  {
    ProfScope ps(s, JAMUN_PROF_HEAD, st);
    launch_head(hd, st);
    if (next_pre)  // walk: this iteration's finalize and the next iteration's geometry in one launch
      launch_finalize_geom(y, s->yc, s->g, s->ptr, s->n_graphs, s->c_skip, s->c_out, s->sigma * s->sigma, s->hp.mean_center, s->tmp, xhat, score, post,
                           s->c_in, s->r2, s->S, s->bond_in_ptr, s->bond_in_src, s->deg, s->esrc, s->egeo, s->epair, *next_pre, st);
    else
      launch_finalize(y, s->yc, s->g, s->ptr, s->n_graphs, s->c_skip, s->c_out, s->sigma * s->sigma, s->hp.mean_center,
                      s->tmp, xhat, score, post, st);
  }
  HIPCHECK(hipGetLastError());
}

LangevinConsts make_consts(const jamun_mcmc_params* p) {
  if (p->M <= 0) throw Err(JAMUN_ERR_INVALID, "M must be positive");
  const double u = 1.0 / (double)p->M;  // pow(M, -1)
  const double zeta2 = std::sqrt(1.0 - std::exp(-2.0 * (double)p->friction));
  LangevinConsts k;
  k.u_half_delta = (float)(u * ((double)p->delta / 2));
  k.half_delta = (float)((double)p->delta / 2);
  k.exp_mg = (float)std::exp(-(double)p->friction);
  k.zeta_sqrt_u = (float)(zeta2 * std::sqrt(u));
  k.beta = p->inverse_temperature;
  k.clip = p->score_fn_clip;
  k.has_clip = p->has_clip;
  return k;
}

void check_mcmc(const jamun_mcmc_params* p) {
  if (!p) throw Err(JAMUN_ERR_INVALID, "null mcmc params");
  if (p->steps < 1) throw Err(JAMUN_ERR_INVALID, "steps must be >= 1");
  if (p->save_every_n_steps < 1) throw Err(JAMUN_ERR_INVALID, "save_every_n_steps must be >= 1");
}

bool saves(const jamun_mcmc_params* p, int i) { return (i % p->save_every_n_steps) == 0 && i >= p->burn_in_steps; }

// Tiles of the destination-grouped conv kernels.  A tile = up to 32 consecutive destination atoms whose source span (whole
// molecules) has at most `cap` rows; tiles are cut greedily at molecule granularity.  A molecule larger than `cap`: its sources
// are cut into row blocks of <= cap atoms and every destination chunk (<= 32 atoms of the molecule) gets one tile PER source
// block — the contraction is linear in the edge coefficients, so the blocks' results are just more partial slabs for the node
// update to sum (edges whose source lies outside a tile's block are skipped by that tile).
void plan_tiles(const int32_t* ptr, const std::vector<int>& graph_of, int N, int cap, std::vector<int2>& t_atoms,
                std::vector<int2>& t_span, std::vector<int>& t_chunk, int& n_chunks, int& span_max, bool& row_blocks) {
  int a0 = 0;
  while (a0 < N) {
    const int g0 = graph_of[a0], lo = ptr[g0], mol_hi = ptr[g0 + 1];
    if (mol_hi - lo > cap) {
      const int n_mol = mol_hi - lo, nb = (n_mol + cap - 1) / cap;
      row_blocks = true;
      for (int d0 = lo; d0 < mol_hi; d0 += 32) {
        const int cnt = std::min(32, mol_hi - d0);
        for (int b = 0; b < nb; ++b) {
          const int blo = lo + (int)((int64_t)n_mol * b / nb), bhi = lo + (int)((int64_t)n_mol * (b + 1) / nb);
          t_atoms.push_back(make_int2(d0, cnt));
          t_span.push_back(make_int2(blo, bhi));
          t_chunk.push_back(n_chunks);
          span_max = std::max(span_max, bhi - blo);
        }
        ++n_chunks;
      }
      a0 = mol_hi;
      continue;
    }
    int cnt = 0, hi = lo;
    while (a0 + cnt < N && cnt < 32) {
      const int g2 = graph_of[a0 + cnt], nhi = ptr[g2 + 1];
      if (nhi - lo > cap) break;  // (also stops in front of a molecule that needs row blocks)
      cnt += std::min(nhi - (a0 + cnt), 32 - cnt);
      hi = nhi;
    }
    t_atoms.push_back(make_int2(a0, cnt));
    t_span.push_back(make_int2(lo, hi));
    t_chunk.push_back(n_chunks++);
    span_max = std::max(span_max, hi - lo);
    a0 += cnt;
  }
}

// Work lists of the persistent conv kernels.  Work items are (tile, hidden unit k).  k is sliced over `ng` groups of XCDs
// (workgroup g runs on XCD g % 8, so an XCD's L2 holds only its slice of the weights); the n_k % ng left-over k are dealt
// round-robin over (tile, slice).  Each slice's item list (tile-major) is cut over its workgroups — evenly by item count for
// near-uniform batches, by modelled cost otherwise: a workgroup gets a few runs of k ("segments"), each written to its own
// partial slab of the tile's destination chunk.
struct SegPlan {
  std::vector<int4> segs;  // [cus][max_segs][2]
  int max_segs = 1, n_slabs = 1;
  std::vector<int> atom_nslab;
};
template <typename WeightFn>
SegPlan plan_segments(int cus, int ng, int n_k, int N, const std::vector<int2>& t_atoms, const std::vector<int>& t_chunk, int n_chunks,
                      WeightFn weight, const std::vector<char>* skip = nullptr,  // skip[t]: tile t is not on this plan (its atoms get 0 slabs)
                      double seg_cost = 0.0) {                                   // cost of a segment's prologue + epilogue, in items
  SegPlan P;
  const int ncx_all = cus / ng;
  std::vector<std::vector<int>> wg_of(ng);  // workgroups of k-slice x, in launch order
  for (int g = 0; g < cus; ++g) wg_of[ng == 1 ? 0 : (g % 8) % ng].push_back(g);
  const int base = n_k / ng, rem = n_k % ng;
  std::vector<std::vector<int4>> wg_segs(cus);
  const int n_tiles = (int)t_atoms.size();
  std::vector<int> nslab(n_chunks, 0);  // per destination chunk: its tiles (source row blocks, k runs) number their slabs jointly
  for (int x = 0; x < ng; ++x) {
    auto extra_of = [&](int t) { const int e = ((x - t) % ng + ng) % ng; return e < rem ? ng * base + e : -1; };
    // (near-uniform batches are cut by item count: measured 1 % better on cfg2 than the modelled weights, whose error
    // then exceeds the spread they describe)
    auto skipped = [&](int t) { return skip && (*skip)[t]; };
    int64_t w_min = -1, w_max = -1;
    for (int t = 0; t < n_tiles; ++t) {
      if (skipped(t)) continue;
      w_min = w_min < 0 ? weight(t) : std::min<int64_t>(w_min, weight(t));
      w_max = std::max<int64_t>(w_max, weight(t));
    }
    if (w_max < 0) continue;  // (no tile on this plan)
    const bool uniform = 4 * (w_max - w_min) < w_max;
    auto weight_of = [&](int t) -> int64_t { return uniform ? 1 : weight(t); };
    const double unit = uniform ? 1.0 : 1.0 / (double)std::max<int64_t>(w_min, 1);
    int64_t Lx = 0, Wx = 0;
    for (int t = 0; t < n_tiles; ++t) {
      if (skipped(t)) continue;
      const int cnt = base + (extra_of(t) >= 0 ? 1 : 0);
      Lx += cnt;
      Wx += cnt * weight_of(t);
    }
    // small batches: do not cut the list finer than 8 items per workgroup (a tile's partial slabs are summed by the node
    // update; one slab per hidden unit would make that kernel the bottleneck)
    const int ncx = (int)std::max<int64_t>(1, std::min<int64_t>(ncx_all, Lx / 8));
    // Every segment costs its workgroup a prologue and an epilogue (staging the span's rows, the edge records, the partial slab): `seg_cost`
    // items' worth (measured with the kernels' segment stamps: k_conv_mf 17 k cycles against 4.75 k per item, k_conv_ml 49 k against 8.5 k).  A
    // workgroup whose share of the list crosses a tile boundary runs two segments, one that does not runs one: the list is cut so that
    // items x weight + segments x seg_cost is level — the smallest per-workgroup budget for which a greedy walk over the list fits ncx workgroups.
    auto walk = [&](double budget, std::vector<std::vector<int4>>* out) {
      int c = 0;
      double acc = 0;
      for (int t = 0; t < n_tiles; ++t) {
        if (skipped(t)) continue;
        const int ex = extra_of(t), cnt = base + (ex >= 0 ? 1 : 0);
        const double w = (double)weight_of(t) * unit;
        int i0 = 0;
        while (i0 < cnt) {
          int take = (int)std::floor((budget - acc - seg_cost) / w + 1e-9);
          if (take < 1 && acc > 0) { ++c; acc = 0; continue; }  // (no room for a segment with one item: next workgroup)
          take = std::max(1, std::min(take, cnt - i0));
          if (c >= ncx) return false;
          if (out) {
            const int i1 = i0 + take;
            const int kb = x * base + std::min(i0, base), ke = x * base + std::min(i1, base);
            auto& v = (*out)[c];
            v.push_back(make_int4(t, nslab[t_chunk[t]]++, kb, ke));
            v.push_back(make_int4(i1 > base ? ex : -1, 0, 0, 0));
          }
          acc += seg_cost + take * w;
          i0 += take;
        }
      }
      return c < ncx;
    };
    double lo = 0, hi = 0;
    {
      // unit: items are counted in units of the lightest tile's weight
      for (int t = 0; t < n_tiles; ++t)
        if (!skipped(t)) hi += (base + (extra_of(t) >= 0 ? 1 : 0)) * (double)weight_of(t) * unit + seg_cost;
      lo = hi / ncx * 0.5;
    }
    for (int iter = 0; iter < 60; ++iter) {
      const double mid = 0.5 * (lo + hi);
      if (walk(mid, nullptr)) hi = mid; else lo = mid;
    }
    // Which workgroup runs which share.  Workgroup g runs on XCD g % 8 (round-robin dispatch), every XCD has its own 4 MB L2, and every
    // workgroup streams the layer's weights (127 KB per hidden unit) at the pace of its k loop: a block is served by the L2 a second
    // time only to a workgroup of the SAME XCD that reaches the same hidden unit within a few steps (32 streams x 127 KB = the whole L2
    // per step).  The shares are therefore dealt to the XCDs by the PHASE of their k loop — the hidden unit their first segment starts
    // at; a share continues with k = 0 of the next tile when it crosses a tile boundary —: the 32 workgroups of an XCD then walk a
    // window of ~65 / 8 hidden units together, ~1 MB of weights, and every XCD fetches the stream once (cfg2: FETCH_SIZE of k_conv_mf
    // 291 -> 90 MB per launch, +2.3 % conformations/s; cfg5 +1.5 %; in launch order the phases of an XCD's workgroups were spread over
    // all 65 units — profiles/EXPERIMENTS.md).
    // (the greedy walk is not strictly monotone in the budget: the bisection's `hi` is verified, and widened if need be, BEFORE the walk that
    // numbers the slabs — the first budget tried above, the whole list in one share, always fits)
    for (int grow = 0; grow < 64 && !walk(hi, nullptr); ++grow) hi *= 1.05;
    std::vector<std::vector<int4>> share((size_t)ncx);
    if (!walk(hi, &share)) throw Err(JAMUN_ERR_INVALID, "plan_segments: no feasible cut of the work list");
    {  // every (tile, hidden unit) of this plan exactly once
      std::vector<int> covered((size_t)n_tiles, 0), extra_seen((size_t)n_tiles, 0);
      for (auto& v : share)
        for (size_t q = 0; q + 1 < v.size(); q += 2) {
          covered[v[q].x] += v[q].w - v[q].z;
          if (v[q + 1].x >= 0) ++extra_seen[v[q].x];
        }
      for (int t = 0; t < n_tiles; ++t) {
        if (skipped(t)) continue;
        if (covered[t] != base || extra_seen[t] != (extra_of(t) >= 0 ? 1 : 0)) throw Err(JAMUN_ERR_INVALID, "plan_segments: a tile's hidden units are not covered exactly once");
      }
    }
    std::vector<int> order;
    for (int c = 0; c < ncx; ++c)
      if (!share[c].empty()) order.push_back(c);
    const int kspan = std::max(1, base + (rem ? 1 : 0));
    auto phase = [&](int c) { return ((share[c][0].z - x * base) % kspan + kspan) % kspan; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return phase(a) < phase(b); });
    std::vector<std::vector<int>> by_xcd(8);
    for (int g : wg_of[x]) by_xcd[g % 8].push_back(g);
    std::vector<int> xcds;
    for (int q = 0; q < 8; ++q)
      if (!by_xcd[q].empty()) xcds.push_back(q);
    // (equal counts per XCD, the remainder to the first ones — the shares are level, so are the XCDs)
    size_t pos = 0;
    for (size_t qi = 0; qi < xcds.size(); ++qi) {
      const size_t n_q = order.size() / xcds.size() + (qi < order.size() % xcds.size() ? 1 : 0);
      auto& ids = by_xcd[xcds[qi]];
      for (size_t j = 0; j < n_q && pos < order.size(); ++j, ++pos) {
        if (j >= ids.size()) throw Err(JAMUN_ERR_INVALID, "plan_segments: more shares than workgroups on an XCD");
        wg_segs[ids[j]] = share[order[pos]];
      }
    }
  }
  size_t ms = 1;
  for (auto& v : wg_segs) ms = std::max(ms, v.size() / 2 + 1);
  P.max_segs = (int)ms;
  P.segs.assign((size_t)cus * ms * 2, make_int4(-1, 0, 0, 0));
  for (int g = 0; g < cus; ++g) std::copy(wg_segs[g].begin(), wg_segs[g].end(), P.segs.begin() + (size_t)g * ms * 2);
  for (int v : nslab) P.n_slabs = std::max(P.n_slabs, v);
  P.atom_nslab.assign(N, 1);
  for (int t = 0; t < n_tiles; ++t)
    for (int i = 0; i < t_atoms[t].y; ++i) P.atom_nslab[t_atoms[t].x + i] = nslab[t_chunk[t]];
  return P;
}

}  // namespace

extern "C" {

const char* jamun_last_error(void) { return g_err.c_str(); }
int jamun_version(void) { return 6; }

int jamun_model_create(const jamun_hparams* hp, const jamun_tensor* tensors, int32_t n_tensors, jamun_model** out) {
  return guarded([&] {
    if (!hp || !out || (!tensors && n_tensors > 0)) throw Err(JAMUN_ERR_INVALID, "null argument");
    if (hp->edge_attr_dim != 64) throw Err(JAMUN_ERR_INVALID, "only edge_attr_dim = 64 is supported");
    if (hp->mul0 < 1 || hp->mul1 < 0 || hp->n_layers < 0) throw Err(JAMUN_ERR_INVALID, "bad irreps_hidden / n_layers");
    if ((hp->mul0 + hp->mul1 + 31) / 32 > 5 || (hp->mul1 + 31) / 32 > 2)
      throw Err(JAMUN_ERR_INVALID, "irreps_hidden too wide for the compiled conv tiles (mul0 + mul1 <= 160, mul1 <= 64)");
    if (hp->mul1 == 0) throw Err(JAMUN_ERR_INVALID, "irreps_hidden needs at least one 1e channel (output is 1x1e)");
    if (hp->emb_dim[0] != hp->emb_dim[1])
      throw Err(JAMUN_ERR_INVALID, "atom_type and atom_code embedding dims must match (reference atom_embedding.py:54-56)");
    auto* m = new jamun_model();
    m->hp = *hp;
    for (int i = 0; i < n_tensors; ++i) {
      if (!tensors[i].name || (!tensors[i].data && tensors[i].numel > 0)) {
        delete m;
        throw Err(JAMUN_ERR_INVALID, "tensor table entry with null name/data");
      }
      m->t[tensors[i].name] = std::vector<float>(tensors[i].data, tensors[i].data + tensors[i].numel);
    }
    *out = m;
  });
}
void jamun_model_destroy(jamun_model* m) { delete m; }

static void sampler_create_impl(const jamun_model* m, float sigma, const jamun_topology* topo, const jamun_tuning* tuning, jamun_sampler** out);
static void sampler_self_check(const jamun_model* m, float sigma, const jamun_topology* topo, jamun_sampler* s);

int jamun_sampler_create(const jamun_model* m, float sigma, const jamun_topology* topo, const jamun_tuning* tuning, jamun_sampler** out) {
  return guarded([&] {
    if (!m || !topo || !out) throw Err(JAMUN_ERR_INVALID, "null argument");
    if (tuning && (tuning->selfcheck < -1 || tuning->selfcheck > 2)) throw Err(JAMUN_ERR_INVALID, "jamun_tuning.selfcheck must be -1 (off), 0 (default: on), 1 (on) or 2 (on, with an injected fault)");
    jamun_sampler* s = nullptr;
    sampler_create_impl(m, sigma, topo, tuning, &s);
    std::unique_ptr<jamun_sampler> hold(s);
    if (!tuning || tuning->selfcheck >= 0) sampler_self_check(m, sigma, topo, s);
    *out = hold.release();
  });
}

static void sampler_create_impl(const jamun_model* m, float sigma, const jamun_topology* topo, const jamun_tuning* tuning, jamun_sampler** out) {
  {
    jamun_tuning tn{};
    if (tuning) tn = *tuning;
    if (tn.dg_kgroups != 0 && tn.dg_kgroups != 1 && tn.dg_kgroups != 2 && tn.dg_kgroups != 4 && tn.dg_kgroups != 8)
      throw Err(JAMUN_ERR_INVALID, "jamun_tuning.dg_kgroups must be 0 (default), 1, 2, 4 or 8");
    if (tn.f16x1 != 0 && tn.f16x1 != 1) throw Err(JAMUN_ERR_INVALID, "jamun_tuning.f16x1 must be 0 or 1");
    if (tn.seg_cost_tenths < -1 || tn.seg_cost_tenths > 1000) throw Err(JAMUN_ERR_INVALID, "jamun_tuning.seg_cost_tenths must be -1 (no segment cost), 0 (default) or 1..1000");
    if (!(sigma > 0)) throw Err(JAMUN_ERR_INVALID, "sigma must be positive");
    if (topo->n_atoms < 1 || topo->n_graphs < 1) throw Err(JAMUN_ERR_INVALID, "empty walker batch");
    const jamun_hparams& hp = m->hp;
    std::unique_ptr<jamun_sampler> s(new jamun_sampler());
    s->hp = hp;
    s->tune = tn;
    s->sigma = sigma;
    s->n_atoms = topo->n_atoms;
    s->n_graphs = topo->n_graphs;
    s->n_pad = ((topo->n_atoms + 31) / 32) * 32;
    s->XS = hp.mul0 + 3 * hp.mul1;
    s->n_emb = hp.emb_dim[0] + hp.emb_dim[1] + hp.emb_dim[2] + hp.emb_dim[3];
    if (s->n_emb > 224 || hp.mul0 + 3 * hp.mul1 > 224 || hp.mul0 + hp.mul1 > 160 || hp.mul1 > 32)
      throw Err(JAMUN_ERR_INVALID, "irreps too wide for the node-update tiling (embedding <= 224, hidden <= 160 channels, <= 32 vectors)");
    const int N = topo->n_atoms, W = topo->n_graphs;
    // ---- normalisation factors in fp32, op for op as Denoiser.normalization_factors (denoiser.py:116-136,177-178)
    {
      const float A = hp.average_squared_distance;
      const float B = 6.0f * (sigma * sigma);
      s->c_in = 1.0f / sqrtf(A + B);
      s->c_skip = A / (A + B);
      s->c_out = sqrtf((A * B) / (A + B));
      const float mr2 = (float)((double)hp.max_radius * (double)hp.max_radius);
      s->r_cut = sqrtf(mr2 + 6.0f * (sigma * sigma)) / s->c_in;
      s->r2 = s->r_cut * s->r_cut;
    }
    const double c_noise = std::log((double)sigma) / 4.0;
    // ---- graph structure
    int nmax = 0;
    for (int g = 0; g < W; ++g) {
      if (topo->ptr[g + 1] < topo->ptr[g]) throw Err(JAMUN_ERR_INVALID, "ptr must be non-decreasing");
      nmax = std::max(nmax, topo->ptr[g + 1] - topo->ptr[g]);
    }
    if (topo->ptr[0] != 0 || topo->ptr[W] != N) throw Err(JAMUN_ERR_INVALID, "ptr must span [0, n_atoms]");
    std::vector<int> graph_of(N);
    for (int g = 0; g < W; ++g)
      for (int a = topo->ptr[g]; a < topo->ptr[g + 1]; ++a) graph_of[a] = g;
    std::vector<int> bip(N + 1, 0), bis(topo->n_bonds);
    for (int b = 0; b < topo->n_bonds; ++b) {
      const int64_t sa = topo->bond_src[b], da = topo->bond_dst[b];
      if (sa < 0 || sa >= N || da < 0 || da >= N) throw Err(JAMUN_ERR_INVALID, "bond index out of range");
      if (graph_of[sa] != graph_of[da]) throw Err(JAMUN_ERR_INVALID, "bond connects two different walkers");
      bip[da + 1]++;
    }
    int max_in = 0;
    for (int i = 0; i < N; ++i) { max_in = std::max(max_in, bip[i + 1]); bip[i + 1] += bip[i]; }
    {
      std::vector<int> fill(bip.begin(), bip.end() - 1);
      for (int b = 0; b < topo->n_bonds; ++b) bis[fill[topo->bond_dst[b]]++] = (int)topo->bond_src[b];  // stable: list order
    }
    s->S = std::min(std::max(nmax - 1, 0), JAMUN_MAX_NEIGHBORS + 1) + max_in;
    if (s->S < 1) s->S = 1;
    s->n_tiles = s->n_pad / 32;
    {
      const std::pair<const char*, int (*)()> lds_attr[] = {{"k_conv", conv_set_max_lds}, {"k_node_update", node_update_set_max_lds},
                                                            {"k_conv_init_v", conv_initv_set_max_lds}, {"k_conv_dg", conv_dg_set_max_lds},
                                                            {"jamun_conv_mf.hip", conv_mf_set_max_lds}, {"jamun_conv_ml.hip", conv_ml_set_max_lds},
                                                            {"jamun_sepconv.hip", sep_conv_set_max_lds}};
      for (auto& f : lds_attr)
        if (f.second() != 0)
          throw Err(JAMUN_ERR_HIP, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for ") + f.first + ": " + hipGetErrorString(hipGetLastError()));
    }
    std::vector<int> ptr_h(topo->ptr, topo->ptr + W + 1);
    s->ptr = dev_upload(ptr_h);
    s->bond_in_ptr = dev_upload(bip);
    s->bond_in_src = dev_upload(bis);
    // ---- radial basis centres: torch.linspace(0, r_cut, 34)[1:-1] in fp32 (e3nn soft_one_hot_linspace)
    {
      const int nr = (hp.edge_attr_dim + 1) / 2, steps = nr + 2;
      const float start = 0.f, end = s->r_cut;
      const float step = (end - start) / (float)(steps - 1);
      std::vector<float> vals(steps);
      for (int i = 0; i < steps; ++i) vals[i] = (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - i - 1);
      s->rb_step = vals[1] - vals[0];
      std::vector<float> mu(vals.begin() + 1, vals.end() - 1);
      s->mu = dev_upload(mu);
    }
    // ---- scaled atom embeddings (constant per topology and sigma): atom_embedding.py:58-76, noise_conditioning.py:50-54
    std::vector<float> xe_host;
    {
      const char* names[4] = {"atom_embedder.atom_type_embedding.weight", "atom_embedder.atom_code_embedding.weight",
                              "atom_embedder.residue_code_embedding.weight", "atom_embedder.residue_index_embedding.weight"};
      const int32_t* idx[4] = {topo->atom_type_index, topo->atom_code_index, topo->residue_code_index,
                               topo->residue_sequence_index};
      std::vector<double> s0 = noise_mlp(*m, "initial_noise_scaling.scale_predictor", s->n_emb, c_noise);
      std::vector<float>& xe = xe_host;
      xe.assign((size_t)N * s->n_emb, 0.f);
      int col = 0;
      for (int tb = 0; tb < 4; ++tb) {
        const auto& T = m->get(names[tb], (int64_t)hp.emb_rows[tb] * hp.emb_dim[tb]);
        for (int i = 0; i < N; ++i) {
          int row = idx[tb][i];
          if (tb == 3 && !hp.use_residue_sequence_index) row = 0;
          if (row < 0 || row >= hp.emb_rows[tb]) throw Err(JAMUN_ERR_INVALID, std::string("index out of range for ") + names[tb]);
          for (int c = 0; c < hp.emb_dim[tb]; ++c)
            xe[(size_t)i * s->n_emb + col + c] = (float)((double)T[(size_t)row * hp.emb_dim[tb] + c] * s0[col + c]);
        }
        col += hp.emb_dim[tb];
      }
      s->x_emb = dev_upload(xe);
    }
    // ---- layers
    {
      // initial projector: four scalar blocks (irreps not simplified, atom_embedding.py:54-56); scaling already in x_emb
      std::vector<InBlock> ib;
      int xo = 0;
      const int muls[4] = {hp.emb_dim[0], hp.emb_dim[0], hp.emb_dim[2], hp.emb_dim[3]};
      for (int b = 0; b < 4; ++b) { ib.push_back({muls[b], 0, xo, xo}); xo += muls[b]; }
      std::vector<double> ones(s->n_emb, 1.0);
      // distinct rows of the scaled embedding (atoms with equal embedding indices share one): the initial projector's
      // input-times-weight products are tabulated per distinct row
      std::vector<float> uniq;
      std::vector<int> uid(N);
      {
        std::map<std::vector<float>, int> seen;
        for (int i = 0; i < N; ++i) {
          std::vector<float> row(xe_host.begin() + (size_t)i * s->n_emb, xe_host.begin() + (size_t)(i + 1) * s->n_emb);
          auto it = seen.find(row);
          if (it == seen.end()) {
            it = seen.emplace(row, (int)seen.size()).first;
            uniq.insert(uniq.end(), row.begin(), row.end());
          }
          uid[i] = it->second;
        }
      }
      s->layers.push_back(build_layer(*m, "initial_projector", ib, ones, s->n_slices, &uniq, s->n_emb, false, &xe_host));
      s->n_uniq = (int)(uniq.size() / (size_t)std::max(s->n_emb, 1));
      if (s->layers.back().tt2 || s->layers.back().tabw) s->atom_uid = dev_upload(uid);
    }
    for (int l = 0; l < hp.n_layers; ++l) {
      std::vector<InBlock> ib = {{hp.mul0, 0, 0, 0}, {hp.mul1, 1, hp.mul0, hp.mul0}};
      const std::string li = std::to_string(l);
      std::vector<double> sc = noise_mlp(*m, "noise_scalings." + li + ".scale_predictor", hp.mul0 + hp.mul1, c_noise);
      LayerDev L = build_layer(*m, "layers." + li, ib, sc, s->n_slices, nullptr, 0, /*pack_dg=*/!tn.no_dg);
      std::vector<double> wm = noise_mlp(*m, "skip_connections." + li + ".weights.scale_predictor", hp.mul0 + hp.mul1, c_noise);
      std::vector<float> mix(wm.size());
      for (size_t i = 0; i < wm.size(); ++i) mix[i] = (float)(1.0 / (1.0 + std::exp(-wm[i])));
      L.mix = dev_upload(mix);
      s->layers.push_back(L);
    }
    for (auto& L : s->layers)
      if (L.sep.w2b)
        if (const char* why = sep_conv_unsupported(L.sep.n0, L.sep.n1, L.p0.nt, L.p1.nt, s->S, hp.edge_attr_dim)) throw Err(JAMUN_ERR_INVALID, why);
    // ---- head (EquivariantMLP, _mlp.py:84-114) and output gain (e3conv.py:134-135)
    {
      const int mul0 = hp.mul0, mul1 = hp.mul1, G0 = mul0 + mul1;
      const auto& Wl = m->get("output_head.0.lin.weight", (int64_t)mul0 * G0 + (int64_t)mul1 * mul1);
      const auto& Wo = m->get("output_head.1.weight", mul1);
      const auto& gain = m->get("output_gain", 1);
      std::vector<float> wg((size_t)mul0 * mul1), wv((size_t)mul1 * mul1), wo(mul1);
      for (int u = 0; u < mul0; ++u)
        for (int w = 0; w < mul1; ++w) wg[(size_t)u * mul1 + w] = (float)((double)Wl[(size_t)u * G0 + mul0 + w] / std::sqrt((double)mul0));
      for (int i = 0; i < mul1 * mul1; ++i) wv[i] = (float)((double)Wl[(size_t)mul0 * G0 + i] / std::sqrt((double)mul1));
      for (int w = 0; w < mul1; ++w) wo[w] = (float)((double)Wo[w] / std::sqrt((double)mul1) * (double)gain[0]);
      s->w_gate = dev_upload(wg);
      s->w_vec = dev_upload(wv);
      s->w_out = dev_upload(wo);
    }
    hipDeviceProp_t prop;
    {
      int dev = 0;
      HIPCHECK(hipGetDevice(&dev));
      HIPCHECK(hipGetDeviceProperties(&prop, dev));
    }
    const int cus = std::max(prop.multiProcessorCount, 1);
    s->cus = cus;
    const int n_k = hp.edge_attr_dim + 1;
    // ---- destination-grouped VALU-forming kernel for the hidden layers (jamun_conv_dg.hip): own tile plan, larger spans
    {
      bool ok = !tn.no_dg && hp.n_layers > 0 && s->S <= 64 && (int64_t)N * s->S < (int64_t)0x7fffffff;
      for (size_t l = 1; l < s->layers.size(); ++l) ok = ok && s->layers[l].dg.wx != nullptr;
      // Source rows resident in LDS for the whole segment when the largest molecule fits the resident budget (~80 rows);
      // otherwise the alternating-residency mode of the kernel (rows re-staged per phase: spans up to ~170 rows), and only
      // molecules above THAT are cut into source row blocks.
      const int pmax = (s->S + 3) & ~3;
      s->dg_emu = tn.dg_fp32 ? 0 : 1;
      auto cap_of = [&](int mode) {
        for (int rs = mode == 1 ? 192 : 128; rs >= 16; rs -= 4)
          if (conv_dg_lds_bytes(rs, pmax, mode, s->dg_emu) <= JAMUN_MAX_DYN_LDS) return rs;
        return 0;
      };
      // Modes of k_conv_dg: 0 two phases per k, source rows resident (spans up to ~80 rows); 1 alternating residency (molecules
      // above that, up to ~176 atoms; only larger ones are cut into source row blocks); 2 single phase with double-buffered A
      // tiles (spans up to ~52 rows) — chosen when the smaller span budget does not cost tiles (17-atom molecules: three per
      // tile either way; a 40-atom molecule would fall from straddling tiles to 32 + 8 destinations).
      int cap = ok ? cap_of(0) : 0;
      std::vector<int2> t_atoms, t_span;
      std::vector<int> t_chunk;
      int n_chunks = 0, span_max = 0;
      if (ok && nmax > cap && !tn.dg_no_alt) {
        const int cap_alt = cap_of(1);
        if (cap_alt > cap) { cap = cap_alt; s->dg_mode = 1; }
      }
      if (ok && cap > 0) {
        plan_tiles(topo->ptr, graph_of, N, cap, t_atoms, t_span, t_chunk, n_chunks, span_max, s->dg_row_blocks);
        const int cap_sp = cap_of(2);
        if (s->dg_mode == 0 && cap_sp >= 16 && !tn.dg_no_sp) {
          std::vector<int2> a2, s2;
          std::vector<int> c2;
          int nc2 = 0, sm2 = 0;
          bool rb2 = false;
          plan_tiles(topo->ptr, graph_of, N, cap_sp, a2, s2, c2, nc2, sm2, rb2);
          if (!rb2 && 100 * a2.size() <= 103 * t_atoms.size()) {
            t_atoms.swap(a2); t_span.swap(s2); t_chunk.swap(c2);
            n_chunks = nc2; span_max = sm2; s->dg_row_blocks = false; s->dg_mode = 2;
          }
        }
        // single phase with a double-buffered X tile and ONE Y tile (spans up to ~73 rows: two 33-atom molecules per tile) when the
        // fully double-buffered variant does not fit: a k-step of the two-phase kernel takes 25 k cycles on such tiles, of this
        // one ~17 k, so up to 15 % more tiles are accepted
        const int cap_sph = cap_of(3);
        if (s->dg_mode == 0 && cap_sph >= 16 && !tn.dg_no_sph) {
          std::vector<int2> a2, s2;
          std::vector<int> c2;
          int nc2 = 0, sm2 = 0;
          bool rb2 = false;
          plan_tiles(topo->ptr, graph_of, N, cap_sph, a2, s2, c2, nc2, sm2, rb2);
          if (!rb2 && 100 * a2.size() <= 115 * t_atoms.size()) {
            t_atoms.swap(a2); t_span.swap(s2); t_chunk.swap(c2);
            n_chunks = nc2; span_max = sm2; s->dg_row_blocks = false; s->dg_mode = 3;
          }
        }
        // A operand formed on the matrix cores (jamun_conv_mf.hip): spans that fit one K = 64 window of source rows (from an even
        // atom: 62 rows), when that budget costs no tiles; f16x3 only
        if ((s->dg_mode == 0 || s->dg_mode == 2 || s->dg_mode == 3) && s->dg_emu && !tn.no_mf && s->layers.size() > 1 && s->layers[1].dg.wm) {
          std::vector<int2> a2, s2;
          std::vector<int> c2;
          int nc2 = 0, sm2 = 0;
          bool rb2 = false;
          plan_tiles(topo->ptr, graph_of, N, 62, a2, s2, c2, nc2, sm2, rb2);
          // (the edges of one ordered pair share one coefficient entry, owned by the first with up to two more added: radial edge +
          // at most two listings of the bond)
          int mult = 0;
          {
            std::vector<std::pair<int64_t, int64_t>> bb;
            for (int b = 0; b < topo->n_bonds; ++b) bb.push_back({topo->bond_src[b], topo->bond_dst[b]});
            std::sort(bb.begin(), bb.end());
            for (size_t i = 0, j = 0; i < bb.size(); i = j) {
              while (j < bb.size() && bb[j] == bb[i]) ++j;
              mult = std::max(mult, (int)(j - i));
            }
          }
          // (measured per (tile, k) and workgroup: 3.7 us here, 5.7 us single-phase k_conv_dg, 9.5 us its one-Y-tile variant: the smaller
          // span budget may cost tiles — 33-atom molecules go from two per tile pair to 32 + 1 destinations)
          const size_t allow = s->dg_mode == 2 ? 140 : 230;
          if (!rb2 && sm2 <= 62 && mult <= 2 && 100 * a2.size() <= allow * t_atoms.size()) {
            t_atoms.swap(a2); t_span.swap(s2); t_chunk.swap(c2);
            n_chunks = nc2; span_max = sm2; s->dg_row_blocks = false; s->dg_mode = 4;
          }
        }
        // ... and for larger spans (molecules of 63 .. 167 atoms) the two-pass, block-sparse variant jamun_conv_ml.hip: whole molecules as
        // spans of up to 167 rows; edge strides 33..40 (32 radial slots + bonded in-edges); f16x3 only
        if ((s->dg_mode == 0 || s->dg_mode == 1 || s->dg_mode == 3) && s->dg_emu && !tn.no_mf && !tn.no_ml && s->layers.size() > 1 && s->layers[1].dg.wm &&
            s->S >= 33 && s->S <= 40) {
          std::vector<int2> a2, s2;
          std::vector<int> c2;
          int nc2 = 0, sm2 = 0;
          bool rb2 = false;
          plan_tiles(topo->ptr, graph_of, N, 167, a2, s2, c2, nc2, sm2, rb2);
          int mult = 0, need = 0;
          {
            std::vector<std::pair<int64_t, int64_t>> bb;
            for (int b = 0; b < topo->n_bonds; ++b) bb.push_back({topo->bond_src[b], topo->bond_dst[b]});
            std::sort(bb.begin(), bb.end());
            for (size_t i = 0, j = 0; i < bb.size(); i = j) {
              while (j < bb.size() && bb[j] == bb[i]) ++j;
              mult = std::max(mult, (int)(j - i));
            }
            for (auto& sp : s2) need = std::max(need, sp.y - (sp.x & ~1));
          }
          const int window = conv_ml_window(need);
          if (!rb2 && window > 0 && mult <= 2 && (int64_t)(hp.edge_attr_dim + 1) * 32 * (((int64_t)N + 31 & ~31) + 64) * 4 < ((int64_t)1 << 40)) {
            t_atoms.swap(a2); t_span.swap(s2); t_chunk.swap(c2);
            n_chunks = nc2; span_max = sm2; s->dg_row_blocks = false; s->dg_mode = 5; s->ml_window = window;
          }
        }
        s->dg_RS = std::max((span_max + 3) & ~3, 16);  // (>= 16 rows: the segment-end staging tile of the forming waves aliases the source rows)
        s->dg_n_tiles = (int)t_atoms.size();
        // k-slices over XCD groups (jamun_tuning.dg_kgroups = 1, 2, 4, 8).  Measured on MI355X (cfg2, profiles/r2*): 1 slice 0.317 ms per
        // launch, 2: 0.318, 4: 0.328, 8: 0.343 and the node update slows from 25 to 71 us (more partial slabs per tile): the
        // ~7.7 MB of weight blocks per layer are served from L2 / Infinity Cache fast enough, longer runs of k per segment win.
        const int ng = (tn.dg_kgroups > 1 && cus % 8 == 0 && n_k >= tn.dg_kgroups) ? tn.dg_kgroups : 1;
        auto weight = [&](int t) -> int64_t { return 476 + (s->dg_mode == 1 ? 24 : 2) * ((t_span[t].y - t_span[t].x + 15) / 16); };
        // Tail tiles (mode 4): a tile with at most 8 destinations costs k_conv_mf a whole tile per hidden unit (a 33-atom molecule cuts into
        // 32 + 1: twice the work of a 32-atom one).  They leave the hidden layers' segment lists and go through k_tail_form /
        // k_tail_contract (jamun_conv_mf.hip); worth two more launches per layer when they are at least 4 and 3 % of the tiles.
        std::vector<char> is_tail(t_atoms.size(), 0);
        if (s->dg_mode == 4 && !tn.no_tail && s->layers.size() > 1 && s->layers[1].dg.wmt) {
          std::vector<int4> tt;
          std::vector<int> tatom;
          for (size_t t = 0; t < t_atoms.size(); ++t)
            if (t_atoms[t].y <= 8) {
              tt.push_back(make_int4((int)t, (int)tatom.size(), 0, 0));
              for (int i = 0; i < t_atoms[t].y; ++i) tatom.push_back(t_atoms[t].x + i);
            }
          const size_t p_bytes = (size_t)((tatom.size() + 31) / 32) * 32 * (size_t)n_k * TAIL_NFT * 8 * 16;
          if (tt.size() >= 4 && 100 * tt.size() >= 3 * t_atoms.size() && tt.size() < t_atoms.size() && p_bytes <= ((size_t)2 << 30)) {
            for (auto& e : tt) {
              is_tail[e.x] = 1;
              // (the record carries its tile's descriptor — {first tail destination, first atom, atoms | source rows << 8, first source row} —
              // so that the tail kernels do not start with a second, dependent trip to the tile tables; as the segment records of k_conv_mf)
              const int t = e.x;
              e = make_int4(e.y, t_atoms[t].x, t_atoms[t].y | ((t_span[t].y - t_span[t].x) << 8), t_span[t].x);
            }
            s->n_tail_tiles = (int)tt.size();
            s->n_tail = (int)tatom.size();
            const int n_ct = (s->n_tail + 31) / 32;
            // runs of hidden units of the contraction = partial slabs of the tail atoms: one workgroup per (32 destinations, run, output
            // tile); the node update fetches three slabs at once, so at most three
            s->tail_runs = std::max(1, std::min(3, (32 + n_ct - 1) / n_ct));
            s->tail_tiles = dev_upload(tt);
            s->tail_atom = dev_upload(tatom);
            s->tail_scale = dev_alloc<float>(tatom.size());
            s->tail_P = dev_alloc<float4>(p_bytes / 16);
          }
        }
        // (a segment's prologue + epilogue in items of its k loop, from the kernels' segment stamps; jamun_tuning.seg_cost_tenths overrides)
        const double seg_cost = tn.seg_cost_tenths < 0 ? 0.0 : tn.seg_cost_tenths > 0 ? 0.1 * tn.seg_cost_tenths : s->dg_mode == 4 ? 3.6 : s->dg_mode == 5 ? 5.8 : 0.0;
        // (round 6) every segment record carries its tile's descriptor — {k_extra, first destination, destinations | source rows << 8, first source row} —
        // so that a kernel's first prologue is ONE round trip behind the segment list instead of two (list -> tile tables -> loads)
        auto embed = [&](std::vector<int4>& segs) {
          for (size_t i = 0; i + 1 < segs.size(); i += 2) {
            const int t = segs[i].x;
            if (t < 0) continue;
            segs[i + 1].y = t_atoms[t].x;
            segs[i + 1].z = t_atoms[t].y | ((t_span[t].y - t_span[t].x) << 8);
            segs[i + 1].w = t_span[t].x;
          }
        };
        SegPlan P = plan_segments(cus, ng, n_k, N, t_atoms, t_chunk, n_chunks, weight, s->n_tail_tiles ? &is_tail : nullptr, seg_cost);
        if (s->n_tail_tiles) {
          for (size_t t = 0; t < t_atoms.size(); ++t)
            if (is_tail[t])
              for (int i = 0; i < t_atoms[t].y; ++i) P.atom_nslab[t_atoms[t].x + i] = s->tail_runs;
          P.n_slabs = std::max(P.n_slabs, s->tail_runs);
          s->init_tail = s->layers[0].wx != nullptr && s->layers[0].p0.nt == 5 && !tn.no_mfi;
          if (!s->init_tail) {  // the initial projector keeps every tile on segment lists of its own
            SegPlan PI = plan_segments(cus, ng, n_k, N, t_atoms, t_chunk, n_chunks, weight, nullptr, seg_cost);
            embed(PI.segs);
            s->init_segs = dev_upload(PI.segs);
            s->init_atom_nslab = dev_upload(PI.atom_nslab);
            s->init_max_segs = PI.max_segs;
            s->init_n_slabs = PI.n_slabs;
          }
        }
        s->dg_grid = cus;
        s->dg_max_segs = P.max_segs;
        embed(P.segs);
        s->dg_segs = dev_upload(P.segs);
        s->dg_n_slabs = P.n_slabs;
        s->dg_atom_nslab = dev_upload(P.atom_nslab);
        s->dg_tile_atoms = dev_upload(t_atoms);
        s->dg_tile_span = dev_upload(t_span);
        if (s->dg_mode == 5) {
          s->ml_count = dev_alloc<unsigned long long>(1);
          HIPCHECK(hipMemset(s->ml_count, 0, sizeof(unsigned long long)));
        }
        if (s->dg_mode == 4 || s->dg_mode == 5) {
          int need = 0;  // rows of the window a tile's sources reach (the window starts at an even atom)
          for (size_t t = 0; t < t_span.size(); ++t)
            if (!is_tail[t]) need = std::max(need, t_span[t].y - (t_span[t].x & ~1));
          s->mf_nks = (need <= 48 && !tn.no_short_k) ? 3 : 4;
          s->dg_tstride = ((N + 31) & ~31) + 64;
          s->dg_T = dev_alloc<float>((size_t)n_k * 32 * s->dg_tstride);
          HIPCHECK(hipMemset(s->dg_T, 0, sizeof(float) * (size_t)n_k * 32 * s->dg_tstride));
          s->mf_err = dev_alloc<int>(1);
          HIPCHECK(hipMemset(s->mf_err, 0, sizeof(int)));
          HIPCHECK(hipHostMalloc((void**)&s->mf_err_host, sizeof(int), hipHostMallocDefault));
          *s->mf_err_host = 0;
        } else {
          s->dg_T = dev_alloc<float>((size_t)n_k * N * 32);
        }
        s->dg_on = true;
        s->x1 = (tn.f16x1 && (s->dg_mode == 4 || s->dg_mode == 5) && s->dg_emu) ? 1 : 0;
        // initial projector on the same tiles: two LDS buffers of table rows when they fit (spans up to ~90 rows), else one
        // (up to ~170 rows)
        // (mid-size ragged batches keep the MFMA table kernel: on 17-57 atom molecules, mean in-degree 11, it takes 0.283 ms
        // against 0.312 — the per-k staging of ~76 table rows outweighs the few edges; 33-atom molecules: 0.398 against 0.328)
        if (!tn.no_init_v && !s->dg_row_blocks && s->layers[0].tt2 != nullptr && s->layers[0].p0.nt == 5 && s->dg_RS <= 170) {
          // (one buffer only for the large-molecule plan: measured on the ragged 17-57 atom batch the MFMA table kernel is 10 %
          // faster than the one-buffer variant, on 166-atom molecules — where it falls back to source row blocks — 2.1x slower)
          for (int nbuf = 2; nbuf >= ((s->dg_mode == 1 || s->dg_mode == 5) ? 1 : 2) && !s->initv_on; --nbuf)
            if (conv_initv_lds_bytes(s->dg_RS, pmax, nbuf) <= JAMUN_MAX_DYN_LDS) { s->initv_on = true; s->initv_nbuf = nbuf; }
        }
        // ... or, on the tiles of k_conv_mf (spans within one K = 64 window) and with at most 32 distinct embedding rows, the same
        // scheme with a one-hot selector in place of the feature rows (k_conv_mfi)
        if (s->dg_mode == 4 && s->layers[0].p0.nt == 5 && !tn.no_mfi) {
          // up to 32 distinct rows: one selector tile (112 MFMAs per (tile, k), eight equal waves); more: from the feature rows (192)
          if (s->n_uniq <= 32 && s->layers[0].tabw != nullptr && s->atom_uid != nullptr && s->layers[0].tab_ut == 1) s->mfi_on = true;
          else if (s->layers[0].wx != nullptr) s->mfx_on = true;
        }
        if (s->dg_mode == 5 && s->layers[0].p0.nt == 5 && !tn.no_mfi && s->layers[0].wx != nullptr) s->mlx_on = true;
        if (s->mfi_on || s->mfx_on || s->mlx_on) s->initv_on = false;
      }
      if (!s->dg_on)
        for (auto& L : s->layers) free_dg(L.dg);
    }
    // ---- work buffers
    const size_t NS = (size_t)N * s->S;
    s->yc = dev_alloc<float>((size_t)N * 3);
    s->deg = dev_alloc<int>(N);
    s->esrc = dev_alloc<int>(NS);
    s->epair = dev_alloc<int>(NS);  // (pair table of the matrix-formed kernels, written by k_geom with the edges)
    HIPCHECK(hipMemset(s->epair, 0, sizeof(int) * NS));
    s->egeo = dev_alloc<float4>(NS);
    {
      std::vector<float> w1r_all, cmask_all;
      for (auto& L : s->layers) {
        w1r_all.insert(w1r_all.end(), L.w1r_h.begin(), L.w1r_h.end());
        cmask_all.insert(cmask_all.end(), L.cmask_h.begin(), L.cmask_h.end());
      }
      s->w1r_all = dev_upload(w1r_all);
      if (hp.edge_attr_dim == 64 && !tn.edge_h_fp32) {
        // f16x3 radial MLP (k_edge_h16): W1's radial part per layer scaled to the top of the f16 range and split hi + lo, as A fragments
        std::vector<float4> w1h;
        std::vector<float> isc;
        for (auto& L : s->layers) {
          double wmax = 0;
          for (float v : L.w1r_h) wmax = std::max(wmax, (double)std::fabs(v));
          int ex = 0;
          if (wmax > 0 && std::isfinite(wmax)) std::frexp(wmax, &ex);
          const int sW = std::max(-40, std::min(40, 14 - ex));
          isc.push_back((float)std::ldexp(1.0, -14 - sW));
          const double sc = std::ldexp(1.0, sW);
          for (int mt = 0; mt < 2; ++mt)
            for (int s2 = 0; s2 < 2; ++s2) {
              std::vector<float4> hi(64), lo(64);
              for (int lane = 0; lane < 64; ++lane) {
                const int hh = lane >> 5, k = 32 * mt + (lane & 31);
                uint32_t h[4], l[4];
                for (int i = 0; i < 4; ++i) {
                  uint16_t hp2[2], lp2[2];
                  for (int e = 0; e < 2; ++e) {
                    const int r = 16 * s2 + 8 * hh + 2 * i + e;
                    split_f16((double)L.w1r_h[(size_t)r * 64 + k] * sc, hp2[e], lp2[e]);  // w1r: [basis][hidden]
                  }
                  h[i] = (uint32_t)hp2[0] | ((uint32_t)hp2[1] << 16);
                  l[i] = (uint32_t)lp2[0] | ((uint32_t)lp2[1] << 16);
                }
                std::memcpy(&hi[lane], h, 16);
                std::memcpy(&lo[lane], l, 16);
              }
              w1h.insert(w1h.end(), hi.begin(), hi.end());
              w1h.insert(w1h.end(), lo.begin(), lo.end());
            }
        }
        s->w1h_all = dev_upload(w1h);
        s->w1isc_all = dev_upload(isc);
      }
      s->cmask_all = dev_upload(cmask_all);
      s->h_kstride = (NS + 63) & ~(size_t)63;
      s->h_stride = s->h_kstride * JAMUN_HROWS;
      s->h_batched = s->h_stride * s->layers.size() * sizeof(float) <= ((size_t)4 << 30);  // all layers' h~ at once, up to 4 GiB
      // (+ slack: k_conv_mf reads h~ at slot0 + p * stride without a bounds test; lanes past the last atom's slots read up to
      // 32 * S + 128 floats beyond the table and never use them)
      s->h = dev_alloc<float>(s->h_stride * (s->h_batched ? s->layers.size() : 1) + 32 * (size_t)s->S + 256);
    }
    int nt0 = 0, nt1 = 0;
    for (auto& L : s->layers) { nt0 = std::max(nt0, L.p0.nt); nt1 = std::max(nt1, L.p1.nt); }
    const size_t n_part = (size_t)std::max(std::max(s->n_slices, s->dg_n_slabs), s->init_n_slabs);
    s->partial0 = dev_alloc<float>(n_part * s->n_pad * nt0 * 32);
    s->partial1 = dev_alloc<float>(n_part * s->n_pad * 3 * nt1 * 32);
    {
      int dw = 0;
      for (auto& L : s->layers)
        if (L.sep.w2b) dw = std::max(dw, L.sep.n0 + L.sep.n1 + 3 * (L.sep.n0 + 2 * L.sep.n1));
      if (dw > 0) s->sep_D = dev_alloc<float>((size_t)N * dw);
    }
    s->g = dev_alloc<float>((size_t)N * 3);
    s->tmp = dev_alloc<float>((size_t)N * 3);
    s->xhat_buf = dev_alloc<float>((size_t)N * 3);
    s->score_buf = dev_alloc<float>((size_t)N * 3);
    s->psi = dev_alloc<float>((size_t)N * 3);
    for (size_t l = 0; l < s->layers.size(); ++l) s->x.push_back(dev_alloc<float>((size_t)N * s->XS));
    s->counter = dev_alloc<unsigned long long>(1);
    // ---- FLOP bookkeeping
    s->flop_ref_per_edge = 0;
    s->flop_exec = 0;
    for (auto& L : s->layers) {
      s->flop_ref_per_edge += 2LL * 64 * 64 + 130LL * L.tp_numel;  // SURVEY.md §8 d (SeparableConv: tp_numel = the 336 depth-wise weights)
      if (s->dg_on && &L != &s->layers[0]) {
        // per (tile, k) in k_conv_dg: fp32 path 476 units of v_mfma_f32_32x32x2 (4096 FLOP); f16x3 path 150 v_mfma_f32_32x32x16_f16
        // (32768 FLOP: 50 groups of 16 inputs x 3 products) + 72 v_mfma_f32_16x16x32_f16 (16384 FLOP); + 60 fp32 units per (32 atoms, k)
        // in the T pre-pass
        // (mode 4, jamun_conv_mf.hip: 414 v_mfma_f32_32x32x16_f16 per (tile, k): 228 forming + 186 contraction)
        const int64_t per_tile_k = s->dg_mode == 5 ? (57LL * ((s->ml_window + 15) / 16) + 186) * 32768 : s->dg_mode == 4 ? (s->mf_nks == 3 ? 357LL : 414LL) * 32768 : s->dg_emu ? (150LL * 32768 + 72LL * 16384) : 476LL * 4096;
        s->conv_flop_exec_launch = (int64_t)(s->dg_n_tiles - s->n_tail_tiles) * per_tile_k * (hp.edge_attr_dim + 1);  // (tail tiles run in their own kernels)
        s->flop_exec += s->conv_flop_exec_launch + (int64_t)((s->n_atoms + 31) / 32) * (s->dg_emu ? 24LL * 32768 : 60LL * 4096) * (hp.edge_attr_dim + 1);
      }
      else if (L.sep.w2b) s->flop_exec += 3LL * 2 * (int64_t)N * 32 * ((s->S + 31) / 32) * 64 * 352;  // the per-edge weight GEMM as f16x3 (the rest is VALU work per edge)
      else s->flop_exec += 2LL * s->n_pad * (1LL * L.p0.K * L.p0.nt * 32 + 3LL * L.p1.K * L.p1.nt * 32);
    }
    HIPCHECK(hipDeviceSynchronize());
    *out = s.release();
  }
}

// Create-time self-check: a build whose specialised kernels compute something else than the general ones must not sample.
// The sampler's FIRST forward — on synthetic positions: one random-walk chain per walker, 0.15 nm steps, so that neighbourhoods are as dense
// as a peptide's and the 32-neighbour cap binds on large molecules — is run twice: through the kernels this sampler selected (matrix-formed /
// destination-grouped conv, table or matrix-formed initial projector, f16x3 node update and radial MLP) and through a second, temporary
// sampler restricted to the general kernels (k_conv, k_node_update, k_edge_h: fp32 MFMAs / vector ALUs, no host-planned tiles).  Node
// features after every block and the network output must agree to 2e-5 of the block's largest feature (the parity tests' bound; the f16x3
// kernels sit at 1e-6).  The opt-in reduced-precision mode (f16x1) is checked against its own bound, 2e-2.  SeparableConv has one
// implementation and is not checked.  Cost: one general-kernel forward + the packing of its weights (cfg2: 0.3 s, once per sampler).
static void sampler_self_check(const jamun_model* m, float sigma, const jamun_topology* topo, jamun_sampler* s) {
  const bool special = s->dg_on || s->mfi_on || s->mfx_on || s->mlx_on || s->initv_on;
  if (!special) return;
  for (auto& L : s->layers)
    if (L.sep.w2b) return;
  jamun_tuning rt{};
  rt.no_dg = rt.no_mf = rt.no_mfi = rt.no_init_v = rt.no_ml = rt.no_tail = 1;
  rt.node_fp32 = rt.edge_h_fp32 = 1;
  rt.selfcheck = -1;
  jamun_sampler* rp = nullptr;
  sampler_create_impl(m, sigma, topo, &rt, &rp);
  std::unique_ptr<jamun_sampler> r(rp);
  const int N = s->n_atoms;
  std::vector<float> y((size_t)N * 3);
  {
    uint64_t z = 0x9e3779b97f4a7c15ull;
    auto u = [&]() { z = z * 6364136223846793005ull + 1442695040888963407ull; return (float)((z >> 40) & 0xffffff) / 8388608.0f - 1.0f; };  // (-1, 1)
    for (int g = 0; g < topo->n_graphs; ++g) {
      float px = 0.f, py = 0.f, pz = 0.f;
      for (int a = topo->ptr[g]; a < topo->ptr[g + 1]; ++a) {
        float dx, dy, dz, n2;
        do { dx = u(); dy = u(); dz = u(); n2 = dx * dx + dy * dy + dz * dz; } while (n2 < 0.05f || n2 > 1.f);
        const float inv = 0.15f / sqrtf(n2);
        px += dx * inv; py += dy * inv; pz += dz * inv;
        y[(size_t)a * 3] = px; y[(size_t)a * 3 + 1] = py; y[(size_t)a * 3 + 2] = pz;
      }
    }
  }
  float* y_dev = dev_upload(y);
  struct Free { float* p; ~Free() { hipFree(p); } } free_y{y_dev};
  if (s->tune.selfcheck == 2 && s->layers.size() > 1) {
    // fault injection (tests): 4 KB of the first hidden layer's weight stream of the SELECTED kernel read as zeros from here on
    float4* w = s->layers[1].dg.wm ? s->layers[1].dg.wm : s->layers[1].dg.wxh ? s->layers[1].dg.wxh : s->layers[1].dg.wx;
    if (w) HIPCHECK(hipMemset(reinterpret_cast<char*>(w) + 64 * 1024, 0, 4096));
  }
  hipStream_t st = nullptr;
  forward(s, y_dev, s->xhat_buf, nullptr, st);
  forward(r.get(), y_dev, r->xhat_buf, nullptr, st);
  HIPCHECK(hipStreamSynchronize(st));
  mf_err_fetch(s, st);
  HIPCHECK(hipStreamSynchronize(st));
  mf_err_check(s);
  const int n_cmp = std::min(N, 16384);  // (rows compared per block: the first 512 tiles — every tile runs the same code)
  std::vector<float> a((size_t)n_cmp * s->XS), b((size_t)n_cmp * s->XS);
  const double tol = s->x1 ? 2e-2 : 2e-5;
  auto compare = [&](const float* da, const float* db, size_t n, const char* what, int layer) {
    HIPCHECK(hipMemcpy(a.data(), da, n * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(b.data(), db, n * sizeof(float), hipMemcpyDeviceToHost));
    double mx = 0, dv = 0;
    bool finite = true;
    for (size_t i = 0; i < n; ++i) {
      finite = finite && std::isfinite(a[i]) && std::isfinite(b[i]);
      mx = std::max(mx, (double)std::fabs(b[i]));
      dv = std::max(dv, (double)std::fabs(a[i] - b[i]));
    }
    if (!finite || dv > tol * std::max(mx, 1e-6)) {
      char msg[320];
      snprintf(msg, sizeof msg, "create-time self-check failed: %s%d of the selected kernels (conv path %d, mode %d, initial projector %d) deviates from the general kernels by %.3g of its largest value (bound %.0e)%s — this build must not sample",
               what, layer, s->dg_on ? 2 : 0, s->dg_on ? s->dg_mode : -1, s->mlx_on ? 5 : s->mfx_on ? 4 : s->mfi_on ? 3 : s->initv_on ? 2 : 0, finite ? dv / std::max(mx, 1e-6) : NAN, tol, finite ? "" : " (non-finite values)");
      throw Err(JAMUN_ERR_INVALID, msg);
    }
  };
  for (size_t l = 0; l < s->layers.size(); ++l) compare(s->x[l], r->x[l], (size_t)n_cmp * s->XS, "node features after block ", (int)l);
  compare(s->g, r->g, (size_t)n_cmp * 3, "network output g, block ", (int)s->layers.size());
  if (s->ml_count) { HIPCHECK(hipMemset(s->ml_count, 0, sizeof(unsigned long long))); s->ml_launches = 0; }  // (the check's launches are not the run's)
  s->edges_built = false;
}
void jamun_sampler_destroy(jamun_sampler* s) { delete s; }

int jamun_xhat(jamun_sampler* s, const float* y_dev, float* xhat_dev, void* stream) {
  return guarded([&] {
    if (!s || !y_dev || !xhat_dev) throw Err(JAMUN_ERR_INVALID, "null argument");
    mf_err_check(s);
    forward(s, const_cast<float*>(y_dev), xhat_dev, nullptr, (hipStream_t)stream);  // (y is written only with a fused pre-update)
    mf_err_fetch(s, (hipStream_t)stream);
  });
}
int jamun_score(jamun_sampler* s, const float* y_dev, float* score_dev, void* stream) {
  return guarded([&] {
    if (!s || !y_dev || !score_dev) throw Err(JAMUN_ERR_INVALID, "null argument");
    mf_err_check(s);
    forward(s, const_cast<float*>(y_dev), nullptr, score_dev, (hipStream_t)stream);
    mf_err_fetch(s, (hipStream_t)stream);
  });
}

int jamun_num_frames(const jamun_mcmc_params* p, int32_t* n_y, int32_t* n_s_baoab, int32_t* n_s_aboba) {
  return guarded([&] {
    check_mcmc(p);
    int ny = saves(p, 0) ? 1 : 0, extra = 0;
    for (int i = 1; i < p->steps; ++i) extra += saves(p, i) ? 1 : 0;
    if (n_y) *n_y = ny + extra;
    if (n_s_baoab) *n_s_baoab = 1 + extra;
    if (n_s_aboba) *n_s_aboba = extra;
  });
}

int jamun_walk_baoab(jamun_sampler* s, float* y, float* v, const jamun_mcmc_params* p, const float* noise,
                     uint64_t seed, float* y_traj, float* score_traj, float* xhat_traj, float* xhat_out, void* stream) {
  return guarded([&] {
    if (!s || !y || !v) throw Err(JAMUN_ERR_INVALID, "null argument");
    check_mcmc(p);
    mf_err_check(s);
    hipStream_t st = (hipStream_t)stream;
    const LangevinConsts k = make_consts(p);
    const int n = s->n_atoms;
    const size_t fr = (size_t)n * 3;
    int fy = 0, fs = 0;
    // The two halves of an iteration run inside the first and the last kernel of the forward (k_geom, k_finalize): one step =
    // the forward's launches and nothing else — and (round 6) the last kernel of iteration i IS the first of iteration i + 1
    // (k_finalize_geom: 20 launches per step instead of 21; jamun_tuning.no_fuse_geom: the two separate kernels).
    // i = 0: initial frame + initial score (_splitting.py:136-155)
    const bool fuse = !s->tune.no_fuse_geom;
    auto make_pre = [&](int i) {
      LangevinPre pre;
      pre.v = v; pre.psi = s->psi; pre.noise = noise ? noise + fr * (size_t)(i - 1) : nullptr; pre.seed = seed; pre.iter = (uint32_t)i; pre.k = k;
      return pre;
    };
    {
      const bool sv = saves(p, 0);
      LangevinPost post;
      post.psi_out = s->psi; post.v = v; post.update_v = 0; post.k = k;
      post.y_frame = (y_traj && sv) ? y_traj + fr * fy : nullptr;
      post.xhat_frame = (xhat_traj && sv) ? xhat_traj + fr * fy : nullptr;
      post.score_frame = score_traj ? score_traj + fr * fs : nullptr;
      const LangevinPre pre1 = make_pre(1);
      forward(s, y, s->xhat_buf, s->score_buf, st, LangevinPre(), post, (fuse && p->steps > 1) ? &pre1 : nullptr);
      if (sv) ++fy;
      ++fs;
    }
    for (int i = 1; i < p->steps; ++i) {
      const LangevinPre pre = make_pre(i), pre_next = make_pre(i + 1);
      const bool sv = saves(p, i);
      LangevinPost post;
      post.psi_out = s->psi; post.v = v; post.update_v = 1; post.k = k;
      post.y_frame = (y_traj && sv) ? y_traj + fr * fy : nullptr;
      post.xhat_frame = (xhat_traj && sv) ? xhat_traj + fr * fy : nullptr;
      // scores after the initial one are kept only together with the trajectory (_splitting.py:168-170): without y_traj
      // the caller's score_traj holds ONE frame
      post.score_frame = (score_traj && y_traj && sv) ? score_traj + fr * fs : nullptr;
      forward(s, y, s->xhat_buf, s->score_buf, st, pre, post, (fuse && i + 1 < p->steps) ? &pre_next : nullptr, /*geom_done=*/fuse);
      if (sv) { ++fy; ++fs; }
    }
    if (xhat_out) launch_copy(s->xhat_buf, xhat_out, n * 3, st);  // last forward was evaluated at the final y
    mf_err_fetch(s, st);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_walk_aboba(jamun_sampler* s, float* y, float* v, const jamun_mcmc_params* p, const float* noise,
                     uint64_t seed, float* y_traj, float* score_traj, float* xhat_traj, float* xhat_out, void* stream) {
  return guarded([&] {
    if (!s || !y || !v) throw Err(JAMUN_ERR_INVALID, "null argument");
    check_mcmc(p);
    mf_err_check(s);
    hipStream_t st = (hipStream_t)stream;
    const LangevinConsts k = make_consts(p);
    const int n = s->n_atoms;
    const size_t fr = (size_t)n * 3;
    int fy = 0, fs = 0;
    if (saves(p, 0)) {
      if (y_traj) launch_copy(y, y_traj, n * 3, st);
      if (xhat_traj) forward(s, y, xhat_traj, nullptr, st);
      ++fy;
    }
    for (int i = 1; i < p->steps; ++i) {
      launch_aboba_a(y, v, n, k.half_delta, st);
      forward(s, y, nullptr, s->score_buf, st);
      const bool sv = saves(p, i);
      float* yf = (y_traj && sv) ? y_traj + fr * fy : nullptr;
      float* sf = (score_traj && sv) ? score_traj + fr * fs : nullptr;
      launch_aboba_b(y, v, s->score_buf, noise ? noise + fr * (size_t)(i - 1) : nullptr, seed, (uint32_t)i, n, k, yf, sf, st);
      if (sv) {
        if (xhat_traj) forward(s, y, xhat_traj + fr * fy, nullptr, st);  // one extra forward per saved frame, as the reference
        ++fy; ++fs;
      }
    }
    if (xhat_out) forward(s, y, xhat_out, nullptr, st);
    mf_err_fetch(s, st);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_mean_center(const float* pos, const int32_t* ptr, int32_t n_graphs, float* out, void* stream) {
  return guarded([&] {
    if (!pos || !ptr || !out || n_graphs < 0) throw Err(JAMUN_ERR_INVALID, "bad argument");
    if (n_graphs == 0) return;
    launch_mean_center(pos, ptr, n_graphs, out, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_radius_graph(const float* pos, const int32_t* ptr, int32_t n_graphs, int32_t n_atoms, float r, int32_t stride,
                       int32_t* nbr, int32_t* deg, void* stream) {
  return guarded([&] {
    if (!pos || !ptr || !nbr || !deg || n_graphs < 0 || n_atoms < 0) throw Err(JAMUN_ERR_INVALID, "bad argument");
    if (stride < JAMUN_MAX_NEIGHBORS + 1) throw Err(JAMUN_ERR_INVALID, "stride must be >= 33");
    if (n_graphs == 0) return;
    launch_radius_graph(pos, ptr, n_graphs, r * r, stride, nbr, deg, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_scatter_mean(const float* src, const int32_t* seg_ptr, int32_t n_out, int32_t width, float* out, void* stream) {
  return guarded([&] {
    if (!seg_ptr || !out || n_out < 0 || width < 1) throw Err(JAMUN_ERR_INVALID, "bad argument");
    if (n_out == 0) return;
    launch_scatter_mean(src, seg_ptr, n_out, width, out, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_baoab_pre(float* y, float* v, const float* psi, const float* noise, int32_t n, const jamun_mcmc_params* p,
                    void* stream) {
  return guarded([&] {
    if (!y || !v || !psi || !noise || !p) throw Err(JAMUN_ERR_INVALID, "null argument");
    if (n == 0) return;
    launch_baoab_pre(y, v, psi, noise, 0, 0, n, make_consts(p), (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}
int jamun_baoab_post(float* v, float* psi, const float* score, int32_t n, const jamun_mcmc_params* p, void* stream) {
  return guarded([&] {
    if (!v || !psi || !score || !p) throw Err(JAMUN_ERR_INVALID, "null argument");
    if (n == 0) return;
    launch_baoab_post(v, psi, score, nullptr, nullptr, n, make_consts(p), 1, nullptr, nullptr, nullptr, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_aboba_a(float* y, const float* v, int32_t n, const jamun_mcmc_params* p, void* stream) {
  return guarded([&] {
    if (!y || !v || !p) throw Err(JAMUN_ERR_INVALID, "null argument");
    if (n == 0) return;
    launch_aboba_a(y, v, n, make_consts(p).half_delta, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}
int jamun_aboba_b(float* y, float* v, const float* score, const float* noise, int32_t n, const jamun_mcmc_params* p,
                  void* stream) {
  return guarded([&] {
    if (!y || !v || !score || !noise || !p) throw Err(JAMUN_ERR_INVALID, "null argument");
    if (n == 0) return;
    launch_aboba_b(y, v, score, noise, 0, 0, n, make_consts(p), nullptr, nullptr, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_edge_geometry(const float* pos, int32_t n_atoms, const int64_t* src, const int64_t* dst, int32_t n_edges, float radial_cutoff,
                        int32_t n_basis, float* sh, float* radial, void* stream) {
  return guarded([&] {
    if (!pos || !src || !dst || !sh || !radial || n_edges < 0 || n_atoms < 0) throw Err(JAMUN_ERR_INVALID, "bad argument");
    if (n_basis < 1 || !(radial_cutoff > 0)) throw Err(JAMUN_ERR_INVALID, "n_basis must be >= 1 and radial_cutoff positive");
    if (n_edges == 0) return;
    launch_edge_geometry(pos, (const long long*)src, (const long long*)dst, n_edges, n_atoms, radial_cutoff, n_basis, sh, radial, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_node_linear(const float* x, int32_t n_atoms, int32_t in0, int32_t in1, int32_t out0, int32_t out1, const float* w, int64_t w_numel,
                      float* out, void* stream) {
  return guarded([&] {
    if (!x || !w || !out || n_atoms < 0 || in0 < 0 || in1 < 0 || out0 < 0 || out1 < 0 || in0 + in1 < 1 || out0 + out1 < 1)
      throw Err(JAMUN_ERR_INVALID, "bad argument");
    if (w_numel != (int64_t)in0 * out0 + (int64_t)in1 * out1)
      throw Err(JAMUN_ERR_INVALID, "weight has " + std::to_string(w_numel) + " elements, the irreps need in0*out0 + in1*out1 = " +
                                       std::to_string((int64_t)in0 * out0 + (int64_t)in1 * out1));
    if (n_atoms == 0) return;
    if (launch_node_linear(x, n_atoms, in0, in1, out0, out1, w, out, (hipStream_t)stream) != 0)
      throw Err(JAMUN_ERR_INVALID, "input irreps too wide (8 feature rows must fit 60 KiB of LDS)");
    HIPCHECK(hipGetLastError());
  });
}

int jamun_philox_normal(float* out_dev, int32_t n, uint64_t seed, uint32_t iteration, uint32_t first_atom, void* stream) {
  return guarded([&] {
    if (!out_dev || n < 0) throw Err(JAMUN_ERR_INVALID, "bad argument");
    if (n == 0) return;
    launch_philox_normal(out_dev, n, seed, iteration, first_atom, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_build_edges(jamun_sampler* s, const float* y_dev, void* stream) {
  return guarded([&] {
    if (!s || !y_dev) throw Err(JAMUN_ERR_INVALID, "null argument");
    build_edges(s, const_cast<float*>(y_dev), (hipStream_t)stream);  // (y is written only with a fused pre-update)
    HIPCHECK(hipGetLastError());
  });
}

int jamun_conv_block(jamun_sampler* s, int32_t layer, const float* x_in_dev, float* x_out_dev, void* stream) {
  return guarded([&] {
    if (!s || !x_out_dev) throw Err(JAMUN_ERR_INVALID, "null argument");
    if (layer < 0 || layer >= (int)s->layers.size()) throw Err(JAMUN_ERR_INVALID, "layer out of range");
    if (!s->edges_built) throw Err(JAMUN_ERR_INVALID, "no edge table yet: call jamun_build_edges (or a forward) first");
    mf_err_check(s);
    if (layer == 0) {
      if (x_in_dev) throw Err(JAMUN_ERR_INVALID, "block 0 (initial projector) takes the sampler's own noise-scaled atom embedding: pass x_in = NULL");
      run_layer(s, 0, s->x_emb, s->n_emb, x_out_dev, (hipStream_t)stream);
    } else {
      if (!x_in_dev) throw Err(JAMUN_ERR_INVALID, "null x_in");
      if (x_in_dev == x_out_dev) throw Err(JAMUN_ERR_INVALID, "x_in and x_out must not alias (the skip path reads x_in after the conv)");
      run_layer(s, (size_t)layer, x_in_dev, s->XS, x_out_dev, (hipStream_t)stream);
    }
    mf_err_fetch(s, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
  });
}

int jamun_sampler_stats(jamun_sampler* s, jamun_stats* out, void* stream) {
  return guarded([&] {
    if (!s || !out) throw Err(JAMUN_ERR_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    HIPCHECK(hipMemsetAsync(s->counter, 0, sizeof(unsigned long long), st));
    launch_count_edges(s->deg, s->n_atoms, s->counter, st);
    unsigned long long e = 0;
    HIPCHECK(hipMemcpyAsync(&e, s->counter, sizeof(e), hipMemcpyDeviceToHost, st));
    int mf_flag = 0;
    if (s->mf_err) HIPCHECK(hipMemcpyAsync(&mf_flag, s->mf_err, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    // (k_conv_mf / k_conv_mfi: an ordered pair with more than three edges cannot share one coefficient entry; the host excludes such
    // topologies at create time, the kernels still flag what they see)
    if (mf_flag != 0) {
      if (s->mf_err_host) *s->mf_err_host = mf_flag;
      mf_err_check(s);
    }
    out->n_edges = (int64_t)e;
    out->flop_ref_assoc = (int64_t)e * s->flop_ref_per_edge;
    out->flop_executed = s->flop_exec;
    out->conv_k0 = s->layers.back().p0.K;
    out->conv_k1 = s->layers.back().p1.K;
    {
      const int64_t m0 = s->hp.mul0, m1 = s->hp.mul1, H1 = s->hp.edge_attr_dim + 1;
      out->conv0_flop_alg = 2 * (int64_t)s->n_atoms * H1 * (m0 + m1) * (m0 + m1);
      out->conv1_flop_alg = 2 * 3 * (int64_t)s->n_atoms * H1 * (m0 + 2 * m1) * m1;
    }
    out->edge_stride = s->S;
    out->n_slices = s->dg_on ? s->dg_n_slabs : s->n_slices;
    out->conv_path = s->dg_on ? 2 : 0;
    out->dg_mode = s->dg_on ? s->dg_mode : -1;
    out->init_path = s->mlx_on ? 5 : s->mfx_on ? 4 : s->mfi_on ? 3 : s->initv_on ? 2 : 0;
    out->dg_row_blocks = s->dg_on && s->dg_row_blocks ? 1 : 0;
    out->dg_emu = s->dg_on ? (s->x1 ? 2 : s->dg_emu) : -1;
    out->conv_flop_exec_launch = (s->x1 && s->dg_mode == 4) ? s->conv_flop_exec_launch / 3 : s->conv_flop_exec_launch;
    if (s->dg_on && s->dg_mode == 5 && s->ml_count && s->ml_launches > 0) {  // (block-sparse forming: counted by the kernel; mean over its launches so far)
      unsigned long long cnt = 0;
      HIPCHECK(hipMemcpy(&cnt, s->ml_count, sizeof(cnt), hipMemcpyDeviceToHost));
      out->conv_flop_exec_launch = (int64_t)((double)cnt / (double)s->ml_launches) * 32768;
    }
    if (s->dg_on && s->layers.size() > 1)  // one basis for both figures: the hidden layers' share of flop_executed follows the per-launch figure reported above (f16x1: a third; k_conv_ml: kernel-counted)
      out->flop_executed += (int64_t)(s->layers.size() - 1) * (out->conv_flop_exec_launch - s->conv_flop_exec_launch);
    out->n_tail_tiles = s->n_tail_tiles;
    out->n_tail = s->n_tail;
    out->mf_nks = (s->dg_on && s->dg_mode == 4) ? s->mf_nks : 0;
    out->ml_window = (s->dg_on && s->dg_mode == 5) ? s->ml_window : 0;
    out->conv_flop_useful_launch = 0;
    out->conv_bytes_alg_launch = 0;
    if (s->dg_on && s->layers.size() > 1) {
      const int64_t m0 = s->hp.mul0, m1 = s->hp.mul1, H1 = s->hp.edge_attr_dim + 1, N = s->n_atoms;
      // (the launch these figures describe is the main conv kernel: destinations that go through the tail kernels are not its work — their
      // edges are taken as the batch's mean in-degree, the slab and h~ bytes below stay whole: every slot is read, every slab row written)
      const int64_t Nm = N - ((s->dg_mode == 4 && s->n_tail_tiles > 0) ? s->n_tail : 0), em = N > 0 ? (int64_t)((double)e * (double)Nm / (double)N) : 0;
      const int64_t contraction = 2 * H1 * Nm * ((m0 + m1) * (m0 + m1) + 3 * m1 * (2 * m1));
      // per edge and k: x0 (m0), dot 3 m1, x1 3 m1, cross 6 m1, T term 3 m1 — on the matrix cores only in k_conv_mf (k_conv_dg forms on the vector ALUs)
      const int64_t forming = (s->dg_mode == 4 || s->dg_mode == 5) ? 2 * H1 * em * (m0 + 15 * m1) : 0;
      out->conv_flop_useful_launch = (s->x1 ? 1 : s->dg_emu ? 3 : 1) * (contraction + forming);
      const int64_t slots = (int64_t)s->h_kstride;
      out->conv_bytes_alg_launch = 4 * (H1 * slots          // h~ of the layer
                                        + H1 * 32 * N         // T
                                        + N * s->XS           // feature rows
                                        + (int64_t)s->dg_n_slabs * s->n_pad * 32 * (s->layers[1].p0.nt + 3 * s->layers[1].p1.nt)) +  // slabs
                                   ((s->dg_mode == 4 || s->dg_mode == 5) ? H1 * 124 * 64 * 16 : s->dg_emu ? H1 * 4 * 34 * 64 * 16 : H1 * (5 * 16 + 5 * 4 + 2 * 4) * 64 * 16);  // weights
    } else if (s->layers.size() > 1 && s->layers[1].sep.w2b) {
      // SeparableConv hidden layer (k_sep_fused + k_sep_linear): h~ of the layer, one feature row per edge, the per-destination sums written
      // and read once, the slab, W2~ and the Linear's weights once
      const int64_t n0 = s->layers[1].sep.n0, n1 = s->layers[1].sep.n1, DW = n0 + n1 + 3 * (n0 + 2 * n1), N = s->n_atoms;
      out->conv_bytes_alg_launch = 4 * ((int64_t)(s->hp.edge_attr_dim + 1) * (int64_t)s->h_kstride + (int64_t)e * s->XS + 2 * N * DW + N * (160 + 96) +
                                        (n0 + n1) * (int64_t)(s->hp.mul0 + s->hp.mul1) + (n0 + 2 * n1) * (int64_t)s->hp.mul1) + 4 * 11 * 2 * 1024;
    }
  });
}

int jamun_sampler_check(jamun_sampler* s, void* stream) {
  return guarded([&] {
    if (!s) throw Err(JAMUN_ERR_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    mf_err_fetch(s, st);
    HIPCHECK(hipStreamSynchronize(st));
    mf_err_check(s);
  });
}

int jamun_profile_enable(jamun_sampler* s, int32_t on) {
  return guarded([&] {
    if (!s) throw Err(JAMUN_ERR_INVALID, "null argument");
    s->prof_mask = on == 1 ? 0xffffffffu : (unsigned)on >> 1;  // 1: every class; otherwise bit (c + 1) selects class c
    s->ev_used.clear();
    s->ev_next = 0;
    for (int& n : s->prof_seen) n = 0;
  });
}

int jamun_profile_sample(jamun_sampler* s, int32_t every) {
  return guarded([&] {
    if (!s || every < 1) throw Err(JAMUN_ERR_INVALID, "every must be >= 1");
    s->prof_every = every;
  });
}

int jamun_profile_read(jamun_sampler* s, double* ms_total, int64_t* launches, void* stream) {
  return guarded([&] {
    if (!s || !ms_total || !launches) throw Err(JAMUN_ERR_INVALID, "null argument");
    HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    for (int c = 0; c < JAMUN_PROF_NCLASS; ++c) { ms_total[c] = 0; launches[c] = 0; }
    for (auto& u : s->ev_used) {
      float ms = 0;
      HIPCHECK(hipEventElapsedTime(&ms, s->ev_pool[u.second.first], s->ev_pool[u.second.second]));
      ms_total[u.first] += ms;
      launches[u.first] += 1;
    }
    s->ev_used.clear();
    s->ev_next = 0;
  });
}

int jamun_debug_stamps(unsigned long long* out8) {
  return guarded([&] {
    if (!out8) throw Err(JAMUN_ERR_INVALID, "null argument");
    HIPCHECK(hipDeviceSynchronize());
    conv_dg_print_stamps();
    conv_initv_print_stamps();
    conv_mf_print_stamps();
    conv_ml_print_stamps();
    node_print_stamps();
    for (int i = 0; i < 8; ++i) out8[i] = 0;
  });
}

int jamun_debug_read(jamun_sampler* s, int32_t what, int32_t layer, float* out, void* stream) {
  return guarded([&] {
    if (!s || !out) throw Err(JAMUN_ERR_INVALID, "null argument");
    hipStream_t st = (hipStream_t)stream;
    if (what == 0) {
      if (layer < 0 || layer >= (int)s->x.size()) throw Err(JAMUN_ERR_INVALID, "layer out of range");
      launch_copy(s->x[layer], out, s->n_atoms * s->XS, st);
    } else if (what == 1) {
      launch_deg_to_float(s->deg, out, s->n_atoms, st);
    } else if (what == 2) {
      launch_copy(s->g, out, s->n_atoms * 3, st);
    } else {
      throw Err(JAMUN_ERR_INVALID, "unknown debug buffer");
    }
    HIPCHECK(hipGetLastError());
  });
}

}  // extern "C"
