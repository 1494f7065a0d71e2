// mfma_srcab_war.hip — does a VALU write to the SrcA / SrcB registers of a v_mfma_f32_32x32x16_f16 that was issued a few instructions
// earlier corrupt that MFMA?  (Round 6: the question behind "k_conv_ml<8> with inline-asm splits gives wrong and irreproducible rows".)
//
// LLVM's gfx940/gfx950 hazard table has a software-managed WAR rule for SrcC only (XDL reads SrcC -> VALU writes it); SrcA / SrcB are taken
// as read at issue.  The sequence probed here, with explicit physical registers inside ONE asm block so that nothing is rescheduled:
//     v_mfma acc, A, B, acc          (x CHAIN: back-to-back DEPENDENT on the same accumulator, as the three products of an f16x3 term)
//     v_mfma acc, A2, B, acc         the victim: its SrcC is the result of the MFMA in front of it
//     s_nop (N - 1)                  N = 0 .. 12 wait states
//     v_mov_b32 A2.x, junk           or: v_fma_mix_f32 A2.x, ...   (the instruction the inline-asm split helpers emit)
// and the same with an INDEPENDENT victim (own accumulator, nothing to wait for).  Every lane checks the victim's result against the same
// MFMAs executed with 40 wait states in front of the overwrite.
//   hipcc --offload-arch=gfx950 -O2 mfma_srcab_war.hip -o mfma_srcab_war && ./mfma_srcab_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LOAD_TUPLES                                                                                                   \
  "v_mov_b32 v100, %[a2x]\n v_mov_b32 v101, %[a2y]\n v_mov_b32 v102, %[a2z]\n v_mov_b32 v103, %[a2w]\n"                  \
  "v_mov_b32 v104, %[bx]\n v_mov_b32 v105, %[by]\n v_mov_b32 v106, %[bz]\n v_mov_b32 v107, %[bw]\n"                      \
  "v_mov_b32 v108, %[ax]\n v_mov_b32 v109, %[ay]\n v_mov_b32 v110, %[az]\n v_mov_b32 v111, %[aw]\n s_nop 4\n"
#define OPS [a2x] "v"(a2.x), [a2y] "v"(a2.y), [a2z] "v"(a2.z), [a2w] "v"(a2.w), [bx] "v"(b.x), [by] "v"(b.y), [bz] "v"(b.z), [bw] "v"(b.w), \
            [ax] "v"(a.x), [ay] "v"(a.y), [az] "v"(a.z), [aw] "v"(a.w), [junk] "v"(junk)
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111"
#define DRAIN "s_nop 15\n s_nop 15\n s_nop 15\n"

#define CHAIN "v_mfma_f32_32x32x16_f16 %[acc], v[108:111], v[104:107], %[acc]\n v_mfma_f32_32x32x16_f16 %[acc], v[108:111], v[104:107], %[acc]\n"
#define VICTIM_DEP "v_mfma_f32_32x32x16_f16 %[acc], v[100:103], v[104:107], %[acc]\n"
#define VICTIM_IND "v_mfma_f32_32x32x16_f16 %[acc2], v[100:103], v[104:107], %[acc2]\n"
#define OVW_A "v_mov_b32 v100, %[junk]\n"
#define OVW_B "v_mov_b32 v104, %[junk]\n"
#define OVW_MIX "v_fma_mix_f32 v100, %[junk], -1.0, %[junk] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define N0 ""
#define N1 "s_nop 0\n"
#define N2 "s_nop 1\n"
#define N4 "s_nop 3\n"
#define N8 "s_nop 7\n"
#define N40 "s_nop 15\n s_nop 15\n s_nop 7\n"

// ONE asm block per probe: tuples loaded into fixed registers, the chain, the victim, N wait states, the overwrite, a drain (the compiler
// sees none of the hazards of an asm body, so the block ends only when every MFMA has retired)
#define DEF_KERNEL(NAME, VICTIM, NOPS, OVW)                                                                                             \
  __global__ __launch_bounds__(256) void NAME(const f32x4* __restrict__ in, float* __restrict__ out, int reps) {                        \
    const int t = blockIdx.x * blockDim.x + threadIdx.x;                                                                                \
    const f32x4 a = in[3 * t], a2 = in[3 * t + 1], b = in[3 * t + 2];                                                                   \
    f32x16 acc, acc2;                                                                                                                   \
    for (int q = 0; q < 16; ++q) acc[q] = acc2[q] = 0.f;                                                                                \
    for (int r = 0; r < reps; ++r) {                                                                                                    \
      const float junk = __uint_as_float(0x7bff7bffu - r); /* two large halves */                                                       \
      asm volatile(LOAD_TUPLES CHAIN VICTIM NOPS OVW DRAIN : [acc] "+v"(acc), [acc2] "+v"(acc2) : OPS : CLOB);                          \
    }                                                                                                                                   \
    float s = 0.f;                                                                                                                      \
    for (int q = 0; q < 16; ++q) s += acc[q] + acc2[q];                                                                                 \
    out[t] = s;                                                                                                                         \
  }
#define DEF_MODE(M, VICTIM, OVW)                                                                                                        \
  DEF_KERNEL(k_##M##_0, VICTIM, N0, OVW) DEF_KERNEL(k_##M##_1, VICTIM, N1, OVW) DEF_KERNEL(k_##M##_2, VICTIM, N2, OVW)                     \
  DEF_KERNEL(k_##M##_4, VICTIM, N4, OVW) DEF_KERNEL(k_##M##_8, VICTIM, N8, OVW) DEF_KERNEL(k_##M##_40, VICTIM, N40, OVW)
DEF_MODE(depA, VICTIM_DEP, OVW_A)
DEF_MODE(depB, VICTIM_DEP, OVW_B)
DEF_MODE(depMix, VICTIM_DEP, OVW_MIX)
DEF_MODE(indA, VICTIM_IND, OVW_A)

// The rule that DID bite (round 6): a VGPR written by a vector-ALU instruction and read by the MFMA behind it as SrcB, N wait states later.
// v104 (B.x) holds the OLD value from LOAD_TUPLES; the probe overwrites it with v_cvt_pk_f16_f32 of two new values and multiplies.
#define DEF_V2M(NAME, NOPS)                                                                                                             \
  __global__ __launch_bounds__(256) void NAME(const f32x4* __restrict__ in, float* __restrict__ out, int reps) {                        \
    const int t = blockIdx.x * blockDim.x + threadIdx.x;                                                                                \
    const f32x4 a = in[3 * t], a2 = in[3 * t + 1], b = in[3 * t + 2];                                                                   \
    f32x16 acc, acc2;                                                                                                                   \
    for (int q = 0; q < 16; ++q) acc[q] = acc2[q] = 0.f;                                                                                \
    for (int r = 0; r < reps; ++r) {                                                                                                    \
      const float junk = 0.25f + 0.001f * r; /* the two NEW halves of B.x: cvt_pk(junk, junk) */                                        \
      asm volatile(LOAD_TUPLES "v_cvt_pk_f16_f32 v104, %[junk], %[junk]\n" NOPS VICTIM_IND DRAIN                                         \
                   : [acc] "+v"(acc), [acc2] "+v"(acc2) : OPS : CLOB);                                                                  \
    }                                                                                                                                   \
    float s = 0.f;                                                                                                                      \
    for (int q = 0; q < 16; ++q) s += acc[q] + acc2[q];                                                                                 \
    out[t] = s;                                                                                                                         \
  }
DEF_V2M(k_v2m_0, N0) DEF_V2M(k_v2m_1, N1) DEF_V2M(k_v2m_2, N2) DEF_V2M(k_v2m_4, N4) DEF_V2M(k_v2m_8, N8) DEF_V2M(k_v2m_40, N40)

typedef void (*kern_t)(const f32x4*, float*, int);
static void run(kern_t k, const f32x4* d_in, float* d_out, std::vector<float>& h, int nthreads) {
  hipLaunchKernelGGL(k, dim3(nthreads / 256), dim3(256), 0, 0, d_in, d_out, 50);
  hipMemcpy(h.data(), d_out, nthreads * sizeof(float), hipMemcpyDeviceToHost);
}
static void sweep(const char* what, const kern_t (&ks)[6], const f32x4* d_in, float* d_out, int nthreads) {
  const int ns[6] = {0, 1, 2, 4, 8, 40};
  std::vector<float> ref(nthreads), got(nthreads);
  run(ks[5], d_in, d_out, ref, nthreads);
  printf("%-72s lanes (of %d) with a wrong result, by wait states N:", what, nthreads);
  for (int i = 0; i < 6; ++i) {
    run(ks[i], d_in, d_out, got, nthreads);
    int bad = 0;
    for (int t = 0; t < nthreads; ++t) bad += got[t] != ref[t];
    printf("  N=%d: %d", ns[i], bad);
  }
  printf("\n");
}
#define KS(M) {k_##M##_0, k_##M##_1, k_##M##_2, k_##M##_4, k_##M##_8, k_##M##_40}

int main() {
  const int nthreads = 256 * 1024;  // 4 waves per workgroup, 1024 workgroups: every SIMD holds several waves
  std::vector<f32x4> h(3 * nthreads);
  unsigned s = 12345u;
  auto half2 = [&]() { s = s * 1664525u + 1013904223u; unsigned lo = 0x3800u + ((s >> 8) & 0x3ffu); s = s * 1664525u + 1013904223u; unsigned hi = 0x3800u + ((s >> 8) & 0x3ffu); return __builtin_bit_cast(float, lo | (hi << 16)); };
  for (auto& v : h) v = f32x4{half2(), half2(), half2(), half2()};  // halves in [0.5, 1)
  f32x4* d_in; float* d_out;
  hipMalloc(&d_in, h.size() * sizeof(f32x4)); hipMalloc(&d_out, nthreads * sizeof(float));
  hipMemcpy(d_in, h.data(), h.size() * sizeof(f32x4), hipMemcpyHostToDevice);
  const kern_t depA[6] = KS(depA), depB[6] = KS(depB), depMix[6] = KS(depMix), indA[6] = KS(indA);
  sweep("dependent victim (SrcC = result of the MFMA in front), v_mov to SrcA", depA, d_in, d_out, nthreads);
  sweep("dependent victim, v_mov to SrcB", depB, d_in, d_out, nthreads);
  sweep("dependent victim, v_fma_mix_f32 (op_sel) to SrcA", depMix, d_in, d_out, nthreads);
  sweep("INDEPENDENT victim (own accumulator), v_mov to SrcA", indA, d_in, d_out, nthreads);
  const kern_t v2m[6] = {k_v2m_0, k_v2m_1, k_v2m_2, k_v2m_4, k_v2m_8, k_v2m_40};
  sweep("v_cvt_pk_f16_f32 writes B.x, the MFMA N wait states behind it reads B", v2m, d_in, d_out, nthreads);
  return 0;
}
