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

template <int DIST, int NDPP, int NSALU>
static void run(const char* name, int waves_per_simd, float* out, const float* in) {
  const int blocks = 256 * waves_per_simd;       // 4 waves per block, 4 SIMDs per CU
  const int iters = 2000;
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

int main() {
  float *in, *out;
  CK(hipMalloc(&in, 4096 * sizeof(float)));
  CK(hipMalloc(&out, 256 * 4 * 256 * sizeof(float) * 4));
  CK(hipMemset(in, 0, 4096 * sizeof(float)));
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
