// Stand-alone experiment (GPU box): what costs v_mfma_f32_4x4x1_16B_f32 issue slots?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_mfma_issue.hip -o /tmp/exp_mfma_issue && /tmp/exp_mfma_issue
// One iteration = 64 MFMAs.  Variants:
//   DIST  accumulators the 64 MFMAs rotate over (dependent-chain distance): 1, 2, 4, 8, 16
//   NDPP  lane-shift (v_mov_b32 dpp) instructions per iteration whose results feed the MFMAs' B operands: 0 .. 64
//   NSALU scalar adds per iteration
// and waves per SIMD (1, 2, 3, 4).  Prints TFLOP/s (512 flop per MFMA per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

template <int DIST, int NDPP, int NSALU>
__global__ void __launch_bounds__(256) probe(float* out, const float* in, int iters, int sseed) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[DIST];
#pragma unroll
  for (int i = 0; i < DIST; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float w = in[lane], x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = in[64 + i * 64 + lane];
  int sacc = sseed;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      if (NDPP > 0 && (m % (64 / (NDPP > 64 ? 64 : NDPP))) == 0) {
        constexpr int per = NDPP > 64 ? NDPP / 64 : 1;
#pragma unroll
        for (int r = 0; r < per; ++r) x[(m + r) & 7] = shr1(x[(m + r + 1) & 7]);
      }
      if (NSALU > 0 && (m % (64 / NSALU)) == 0) sacc = sacc * 3 + it;
      acc[m % DIST] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x[m & 7], acc[m % DIST], 4, 5, 0);
    }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < DIST; ++i) s += acc[i];
  out[(blockIdx.x * 256 + threadIdx.x)] = s[0] + s[1] + s[2] + s[3] + (float)sacc;
}

__device__ __forceinline__ f32x4 mfa(int abid, float a, float b, f32x4 c) {    // abid folds to an immediate after unrolling
  switch (abid & 3) {
    case 0: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 0, 0);
    case 1: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 1, 0);
    case 2: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 2, 0);
    default: return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 3, 0);
  }
}

// pure MFMA stream whose A / B operands cycle through NA / NB different registers (VGPR bank effects?)
template <int NA, int NB, int STRIDE>
__global__ void __launch_bounds__(256, 2) probe3(float* out, const float* in, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float w[NA], x[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) w[i] = in[i * 64 + lane];
#pragma unroll
  for (int i = 0; i < NB; ++i) x[i] = in[2048 + i * 64 + lane];
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 216; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(w[(m * STRIDE) % NA], x[(m * STRIDE / 3) % NB], acc[m & 7], 4, 3, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int NA, int NB, int STRIDE>
static void run3(const char* name, float* out, const float* in) {
  const int blocks = 512, iters = 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe3<NA, NB, STRIDE>), dim3(blocks), dim3(256), 0, 0, out, in, 100);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe3<NA, NB, STRIDE>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  printf("%-34s waves/SIMD 2: %7.1f TFLOP/s\n", name, 512.0 * 216 * iters * blocks * 4 / (best * 1e-3) / 1e12);
}

// closer to the row kernels: per "channel" 12 lane shifts of freshly selected values, then 108 MFMAs that read them with
// 27 different weight registers into 12 accumulators.
//   ORDER 0: the three kw taps of a (row, kh) back to back on one accumulator (as vrn16a_row_kernel is written)
//   ORDER 1: kw outermost: consecutive MFMAs go to different accumulators
//   AHEAD 1: the lane shifts of channel c + 1 are issued before the MFMAs of channel c (nothing consumes a fresh VGPR)
//   BR    1: uniform branches (runtime flags, all true) around the three groups of 36 MFMAs
template <int ORDER, int AHEAD, int BR, int SCHED = 0>
__global__ void __launch_bounds__(256, 2) probe2(float* out, const float* in, int iters, int f0, int f1, int f2) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[j][r] = f32x4{0.f, 0.f, 0.f, 0.f};
  float W[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) W[t] = in[t * 64 + lane];
  f32x4 X[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) X[r] = f32x4{in[2048 + r * 64 + lane], in[2049 + r * 64 + lane], in[2050 + r * 64 + lane], in[2051 + r * 64 + lane]};
  auto shifts = [&](int c, float (&x0)[6], float (&xm)[6], float (&xp)[6]) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      x0[r] = X[r][c];
      xm[r] = shr1(x0[r]);
      xp[r] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0[r]), 0x130, 0xf, 0xf, true));
    }
  };
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    float x0[2][6], xm[2][6], xp[2][6];
    if (AHEAD) shifts(0, x0[0], xm[0], xp[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int cur = AHEAD ? (c & 1) : 0;
      if (AHEAD) { if (c + 1 < 4) shifts(c + 1, x0[cur ^ 1], xm[cur ^ 1], xp[cur ^ 1]); }
      else shifts(c, x0[0], xm[0], xp[0]);
      const int flags[3] = {f0, f1, f2};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (!BR || flags[j]) {
          if (ORDER == 0) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
              for (int kh = 0; kh < 3; ++kh) {
                const int jr = r - kh;
                if (jr >= 0 && jr < 4) {
                  const int t = ((2 - j) * 3 + kh) * 3;
                  acc[j][jr] = mfa(c, W[t], xm[cur][r], acc[j][jr]);
                  acc[j][jr] = mfa(c, W[t + 1], x0[cur][r], acc[j][jr]);
                  acc[j][jr] = mfa(c, W[t + 2], xp[cur][r], acc[j][jr]);
                }
              }
          } else {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
              for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int jr = 0; jr < 4; ++jr) {
                  const int r = jr + kh, t = ((2 - j) * 3 + kh) * 3 + kw;
                  const float xv = kw == 0 ? xm[cur][r] : (kw == 1 ? x0[cur][r] : xp[cur][r]);
                  acc[j][jr] = mfa(c, W[t], xv, acc[j][jr]);
                }
          }
        }
      }
      if (SCHED) {                                          // spread the channel's other VALU work between its MFMAs
#pragma unroll
        for (int g = 0; g < 108 / SCHED; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, SCHED, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
      }
    }
    // keep every accumulator live and the inputs changing
#pragma unroll
    for (int r = 0; r < 6; ++r) X[r] += acc[0][r & 3] + acc[1][r & 3] + acc[2][(r + 1) & 3];
  }
  f32x4 s = acc[0][0];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + X[0][0];
}

template <int ORDER, int AHEAD, int BR, int SCHED = 0>
static void run2(const char* name, int waves_per_simd, float* out, const float* in) {
  const int blocks = 256 * waves_per_simd, iters = 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe2<ORDER, AHEAD, BR, SCHED>), dim3(blocks), dim3(256), 0, 0, out, in, 100, 1, 1, 1);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe2<ORDER, AHEAD, BR, SCHED>), dim3(blocks), dim3(256), 0, 0, out, in, iters, 1, 1, 1);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double flop = 512.0 * (4 * 3 * 36) * iters * blocks * 4;      // 4 channels x 3 sets x 12 (r, kh) x 3 kw
  printf("%-34s waves/SIMD %d: %7.1f TFLOP/s\n", name, waves_per_simd, flop / (best * 1e-3) / 1e12);
}

template <int DIST, int NDPP, int NSALU>
static void run(const char* name, int waves_per_simd, float* out, const float* in) {
  const int blocks = 256 * waves_per_simd;       // 4 waves per block, 4 SIMDs per CU
  const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<DIST, NDPP, NSALU>), dim3(blocks), dim3(256), 0, 0, out, in, 200, 1);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<DIST, NDPP, NSALU>), dim3(blocks), dim3(256), 0, 0, out, in, iters, 1);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double flop = 512.0 * 64 * iters * blocks * 4;
  printf("%-34s waves/SIMD %d: %7.1f TFLOP/s\n", name, waves_per_simd, flop / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  float *in, *out;
  CK(hipMalloc(&in, 4096 * sizeof(float)));
  CK(hipMalloc(&out, 256 * 4 * 256 * sizeof(float) * 4));
  CK(hipMemset(in, 0, 4096 * sizeof(float)));
  run3<1, 1, 1>("operands: 1 A reg, 1 B reg", out, in);
  run3<27, 1, 1>("operands: 27 A regs, 1 B reg", out, in);
  run3<1, 18, 3>("operands: 1 A reg, 18 B regs", out, in);
  run3<27, 18, 1>("operands: 27 A regs, 18 B regs", out, in);
  run3<27, 18, 5>("operands: 27 A, 18 B, stride 5", out, in);
  for (int w = 1; w <= 3; ++w) run2<0, 1, 0, 4>("A-like: ahead, 1 VALU per 4 MFMA", w, out, in);
  for (int w = 1; w <= 3; ++w) run2<0, 1, 0, 6>("A-like: ahead, 1 VALU per 6 MFMA", w, out, in);
  for (int w = 1; w <= 3; ++w) run2<0, 0, 0, 4>("A-like: not ahead, 1 VALU per 4", w, out, in);
  for (int w = 1; w <= 3; ++w) run2<0, 0, 0>("A-like: kw chains", w, out, in);
  for (int w = 1; w <= 3; ++w) run2<0, 0, 1>("A-like: kw chains + branches", w, out, in);
  for (int w = 1; w <= 3; ++w) run2<1, 0, 0>("A-like: kw outermost", w, out, in);
  for (int w = 1; w <= 3; ++w) run2<0, 1, 0>("A-like: kw chains, shifts ahead", w, out, in);
  for (int w = 1; w <= 3; ++w) run2<1, 1, 0>("A-like: kw outermost, shifts ahead", w, out, in);
  for (int w = 1; w <= 3; ++w) run2<1, 1, 1>("A-like: outermost, ahead, branches", w, out, in);
  if (argc > 1) {                                      // sustained clock: the same pure-MFMA launch back to back for a while
    for (int rep = 0; rep < atoi(argv[1]); ++rep) run<8, 0, 0>("dist 8, pure (sustained)", 2, out, in);
    return 0;
  }
  for (int w = 1; w <= 4; ++w) run<16, 0, 0>("dist 16, pure", w, out, in);
  for (int w = 1; w <= 3; ++w) run<8, 0, 0>("dist 8, pure", w, out, in);
  for (int w = 1; w <= 3; ++w) run<4, 0, 0>("dist 4, pure", w, out, in);
  for (int w = 1; w <= 3; ++w) run<2, 0, 0>("dist 2, pure", w, out, in);
  for (int w = 1; w <= 3; ++w) run<1, 0, 0>("dist 1, pure", w, out, in);
  for (int w = 1; w <= 3; ++w) run<8, 8, 0>("dist 8, 8 dpp / 64 mfma", w, out, in);
  for (int w = 1; w <= 3; ++w) run<8, 16, 0>("dist 8, 16 dpp / 64 mfma", w, out, in);
  for (int w = 1; w <= 3; ++w) run<8, 32, 0>("dist 8, 32 dpp / 64 mfma", w, out, in);
  for (int w = 1; w <= 3; ++w) run<8, 64, 0>("dist 8, 64 dpp / 64 mfma", w, out, in);
  for (int w = 1; w <= 3; ++w) run<8, 128, 0>("dist 8, 128 dpp / 64 mfma", w, out, in);
  for (int w = 1; w <= 3; ++w) run<8, 0, 16>("dist 8, 16 salu / 64 mfma", w, out, in);
  for (int w = 1; w <= 3; ++w) run<8, 0, 64>("dist 8, 64 salu / 64 mfma", w, out, in);
  for (int w = 1; w <= 3; ++w) run<8, 32, 32>("dist 8, 32 dpp + 32 salu", w, out, in);
  return 0;
}
