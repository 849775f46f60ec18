// Stand-alone experiment (GPU box): WHEN does v_mfma_f32_4x4x1_16B_f32 return wrong values if its destination registers
// overlap a source operand?  (csrc/row_common.h mfa_new works around one such case; tools/check_isa.py wants the exact rule.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_mfma_overlap.hip -o /tmp/exp_mfma_overlap && /tmp/exp_mfma_overlap
// Every variant runs the instruction with hard-coded registers inside one asm block, next to a reference MFMA on registers
// that overlap nothing, and counts the lanes whose four results differ bit-wise, by lane position inside its row of 16.
//   D = v[100:103], C = v[104:107] (or the literal 0), A / B at v110 / v111 (reference) or INSIDE D at v(100 + pos)
//   abid 0 / 5 / 15 (the weights' broadcast block), cbsz 4
//   FOLLOW 0: s_nop padding on both sides; 1: four independent MFMAs right after; 2: four before and four after
//   waves per SIMD: 1, 2, 4 (grid size and a dynamic-LDS occupancy limiter)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

#define PRE_MFMAS \
  "v_mfma_f32_4x4x1_16b_f32 v[112:115], v110, v111, v[112:115] cbsz:4 abid:3\n\t" \
  "v_mfma_f32_4x4x1_16b_f32 v[116:119], v111, v110, v[116:119] cbsz:4 abid:7\n\t" \
  "v_mfma_f32_4x4x1_16b_f32 v[112:115], v110, v111, v[112:115] cbsz:4 abid:9\n\t" \
  "v_mfma_f32_4x4x1_16b_f32 v[116:119], v111, v110, v[116:119] cbsz:4 abid:12\n\t"
#define NOPS "s_nop 15\n\ts_nop 15\n\t"

// AREG / BREG: register names of the operands of the MFMA under test; CARG: its C operand; FOL: 0 / 1 / 2
#define PROBE(NAME, AREG, BREG, CARG, ABID, PRE, POST)                                                              \
  __global__ void __launch_bounds__(256) NAME(const float* __restrict__ in, unsigned* __restrict__ bad, int iters) { \
    const int lane = threadIdx.x & 63;                                                                              \
    unsigned nbad = 0;                                                                                              \
    for (int it = 0; it < iters; ++it) {                                                                            \
      const float a = in[(it * 131 + lane + blockIdx.x * 7) & 4095];                                                \
      const float b = in[4096 + ((it * 17 + lane * 3 + blockIdx.x) & 4095)];                                        \
      const float c0 = in[(it + lane) & 4095], c1 = in[(it * 3 + lane) & 4095], c2 = in[(it * 5 + lane) & 4095],    \
                  c3 = in[(it * 7 + lane) & 4095];                                                                  \
      float r0, r1, r2, r3, d0, d1, d2, d3;                                                                         \
      asm volatile(                                                                                                 \
          "v_mov_b32 v110, %[a]\n\tv_mov_b32 v111, %[b]\n\t"                                                        \
          "v_mov_b32 v104, %[c0]\n\tv_mov_b32 v105, %[c1]\n\tv_mov_b32 v106, %[c2]\n\tv_mov_b32 v107, %[c3]\n\t"    \
          "v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\t"                    \
          "v_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\t" NOPS               \
          "v_mfma_f32_4x4x1_16b_f32 v[120:123], v110, v111, " CARG " cbsz:4 abid:" ABID "\n\t" NOPS                  \
          "v_mov_b32 %[r0], v120\n\tv_mov_b32 %[r1], v121\n\tv_mov_b32 %[r2], v122\n\tv_mov_b32 %[r3], v123\n\t"    \
          "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"                    \
          "v_mov_b32 " AREG ", %[a]\n\tv_mov_b32 " BREG ", %[b]\n\t" NOPS PRE                                        \
          "v_mfma_f32_4x4x1_16b_f32 v[100:103], " AREG ", " BREG ", " CARG " cbsz:4 abid:" ABID "\n\t" POST NOPS     \
          "v_mov_b32 %[d0], v100\n\tv_mov_b32 %[d1], v101\n\tv_mov_b32 %[d2], v102\n\tv_mov_b32 %[d3], v103\n\t"    \
          : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [d0] "=&v"(d0), [d1] "=&v"(d1),         \
            [d2] "=&v"(d2), [d3] "=&v"(d3)                                                                          \
          : [a] "v"(a), [b] "v"(b), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3)                          \
          : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", \
            "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123");                                \
      const bool same = __float_as_uint(r0) == __float_as_uint(d0) && __float_as_uint(r1) == __float_as_uint(d1) && \
                        __float_as_uint(r2) == __float_as_uint(d2) && __float_as_uint(r3) == __float_as_uint(d3);   \
      nbad += same ? 0u : 1u;                                                                                       \
    }                                                                                                               \
    if (nbad) atomicAdd(&bad[lane & 15], nbad);                                                                     \
  }

#define VARIANTS(TAG, AREG, BREG, CARG)                                          \
  PROBE(TAG##_a0_f0, AREG, BREG, CARG, "0", "", "")                              \
  PROBE(TAG##_a5_f0, AREG, BREG, CARG, "5", "", "")                              \
  PROBE(TAG##_a15_f0, AREG, BREG, CARG, "15", "", "")                            \
  PROBE(TAG##_a0_f1, AREG, BREG, CARG, "0", "", PRE_MFMAS)                       \
  PROBE(TAG##_a5_f1, AREG, BREG, CARG, "5", "", PRE_MFMAS)                       \
  PROBE(TAG##_a15_f1, AREG, BREG, CARG, "15", "", PRE_MFMAS)                     \
  PROBE(TAG##_a0_f2, AREG, BREG, CARG, "0", PRE_MFMAS, PRE_MFMAS)                \
  PROBE(TAG##_a5_f2, AREG, BREG, CARG, "5", PRE_MFMAS, PRE_MFMAS)                \
  PROBE(TAG##_a15_f2, AREG, BREG, CARG, "15", PRE_MFMAS, PRE_MFMAS)

VARIANTS(ref, "v108", "v109", "v[104:107]")          // control: the tested MFMA overlaps nothing either
VARIANTS(b0, "v108", "v100", "v[104:107]")
VARIANTS(b1, "v108", "v101", "v[104:107]")
VARIANTS(b2, "v108", "v102", "v[104:107]")
VARIANTS(b3, "v108", "v103", "v[104:107]")
VARIANTS(a0, "v100", "v109", "v[104:107]")
VARIANTS(a1, "v101", "v109", "v[104:107]")
VARIANTS(a2, "v102", "v109", "v[104:107]")
VARIANTS(a3, "v103", "v109", "v[104:107]")
VARIANTS(ab, "v101", "v100", "v[104:107]")            // both inside D (as vrn_row16's down2 kernel has it)
VARIANTS(b0z, "v108", "v100", "0")                    // C = literal zero
VARIANTS(b3z, "v108", "v103", "0")

typedef void (*kern_t)(const float*, unsigned*, int);
struct Var { const char* name; kern_t k; };
#define V9(TAG) {#TAG "_a0_f0", TAG##_a0_f0}, {#TAG "_a5_f0", TAG##_a5_f0}, {#TAG "_a15_f0", TAG##_a15_f0}, \
                {#TAG "_a0_f1", TAG##_a0_f1}, {#TAG "_a5_f1", TAG##_a5_f1}, {#TAG "_a15_f1", TAG##_a15_f1}, \
                {#TAG "_a0_f2", TAG##_a0_f2}, {#TAG "_a5_f2", TAG##_a5_f2}, {#TAG "_a15_f2", TAG##_a15_f2}

int main() {
  std::vector<Var> vars = {V9(ref), V9(b0), V9(b1), V9(b2), V9(b3), V9(a0), V9(a1), V9(a2), V9(a3), V9(ab), V9(b0z), V9(b3z)};
  std::vector<float> h(8192);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f * 4.0f - 2.0f; }
  float* in; unsigned* bad;
  CK(hipMalloc(&in, h.size() * 4)); CK(hipMalloc(&bad, 64));
  CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  printf("# device %s, %d CUs; columns: variant  waves/SIMD  lanes*iters tested  mismatches  [by lane & 15]\n", pr.gcnArchName, cus);
  const int iters = 2000;
  for (int wps : {1, 2, 4}) {
    // one 256-thread workgroup = 4 waves = one per SIMD; wps workgroups resident per CU, capped by dynamic LDS
    const size_t lds = wps == 1 ? 100 * 1024 : (wps == 2 ? 60 * 1024 : 30 * 1024);
    for (auto& v : vars) {
      CK(hipMemset(bad, 0, 64));
      CK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(v.k, dim3(cus * wps * 4), dim3(256), lds, 0, in, bad, iters);
      CK(hipDeviceSynchronize());
      unsigned hb[16];
      CK(hipMemcpy(hb, bad, 64, hipMemcpyDeviceToHost));
      unsigned long long tot = 0;
      for (unsigned x : hb) tot += x;
      printf("%-12s %d %llu %llu  [", v.name, wps, (unsigned long long)cus * wps * 4 * 256 * iters, tot);
      for (int i = 0; i < 16; ++i) printf("%u%s", hb[i], i == 15 ? "]\n" : " ");
    }
  }
  return 0;
}
