// Bounded probe (GPU box): could a split-bf16 form (an fp32 operand = three bf16 terms, six products per MAC) of the 64^3 row kernels on the
// bf16 matrix pipe beat v_mfma_f32_4x4x1_16B_f32?  Two measurements decide it without writing the kernel:
//   1. the issue rate of v_mfma_f32_16x16x32_bf16 against v_mfma_f32_4x4x1_16B_f32, same occupancy (2 waves per SIMD), independent accumulators;
//   2. the error of a 6-product split dot product (K = 432 = conv1_1's contraction) against fp64, next to the fp32 FMA chain's own error.
// With (1), the cost per voxel of kernel A / BC on 16 x 16 x 32 tiles follows from their shapes: M = 4 .. 8 output channels of 16 tile rows.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_bf16_split.hip -o /tmp/exp_bf16_split && /tmp/exp_bf16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <int DIST>
__global__ void __launch_bounds__(256, 2) rate_bf16(float* out, const float* in, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[DIST];
#pragma unroll
  for (int i = 0; i < DIST; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[lane + i]; b[i] = (__bf16)in[64 + lane + i]; }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m) acc[m % DIST] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % DIST], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < DIST; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
template <int DIST>
__global__ void __launch_bounds__(256, 2) rate_f32(float* out, const float* in, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[DIST];
#pragma unroll
  for (int i = 0; i < DIST; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = in[lane], b = in[64 + lane];
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 64; ++m) acc[m % DIST] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m % DIST], 4, 5, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < DIST; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <class K>
static double run(K kern, float* out, const float* in, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, out, in, iters);           // 2 048 waves: two on every SIMD
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(512), dim3(256), 0, 0, out, in, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3 * ms;     // us
}

static float bf16_round(float v) {      // round to nearest even to bfloat16, returned as float
  unsigned u; memcpy(&u, &v, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  u &= 0xffff0000u;
  float r; memcpy(&r, &u, 4);
  return r;
}

int main(int argc, char** argv) {
  float *in, *out;
  CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 512 * 256 * 4));
  std::vector<float> h(4096);
  srand(3);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  if (argc > 1) {          // `exp_bf16_split long [iters]`: the fp32 stream alone for seconds (tools/exp/clock_watch.py samples the clock meanwhile)
    const int it = argc > 2 ? atoi(argv[2]) : 4000000;
    const double t = run(rate_f32<8>, out, in, it);
    printf("v_mfma_f32_4x4x1_16B_f32, %d iterations of 64 on 2 048 waves: %.1f ms per launch -> %.1f TFLOP/s sustained\n", it, t / 1e3,
           2048.0 * it * 64 * 512 / (t * 1e-6) / 1e12);
    return 0;
  }
  // the clocks of an idle part take ~0.5 s of load to ramp (tools/exp/clock_watch.py): 2 000 iterations (1 ms) right after start-up
  // measured 123.8 TFLOP/s for the fp32 stream where the warm part sustains 148-154 — every figure below is from a warm part
  const int iters = 200000;
  (void)run(rate_f32<8>, out, in, 1000000);
  const double n_mfma = 2048.0 * iters * 64;
  const double t_bf = run(rate_bf16<8>, out, in, iters), t_f32 = run(rate_f32<8>, out, in, iters);
  const double clk = 2.4e3;     // MHz nominal (the part holds 2.34-2.39 GHz under these streams); the ratio is what matters
  printf("v_mfma_f32_16x16x32_bf16 : %.1f us for %.0f MFMAs -> %.2f PFLOP/s (16 384 flop each), %.1f cycles per MFMA per SIMD at %.1f GHz\n", t_bf, n_mfma,
         n_mfma * 16384 / (t_bf * 1e-6) / 1e15, t_bf * clk / (iters * 64 * 2), clk / 1e3);
  printf("v_mfma_f32_4x4x1_16B_f32 : %.1f us for %.0f MFMAs -> %.1f TFLOP/s (512 flop each), %.1f cycles per MFMA per SIMD\n", t_f32, n_mfma,
         n_mfma * 512 / (t_f32 * 1e-6) / 1e12, t_f32 * clk / (iters * 64 * 2));
  const double cb = t_bf / (iters * 64 * 2), cf = t_f32 / (iters * 64 * 2);      // us per MFMA per SIMD
  // kernel A (conv1_1 3^3 16 -> 4 + conv2_1 1^3 16 -> 4): fp32 row kernel = 448 MFMAs per 64 voxels.  bf16 16x16x32: a tile is 16 output channels
  // (4 used by conv1_1; conv2_1's 4 ride in other rows of the same tile at no extra cost) x 16 voxels x 32 of the K = 432 (tap, ci) pairs:
  // 14 K-steps x 6 split products per 16 voxels
  const double a_f32 = 448.0 * cf / 64, a_bf = 14.0 * 6 * cb / 16;
  // kernel BC: conv1_2 (K = 108, 8 channels) + conv2_2 (K = 108, 4 channels) as one block-diagonal tile (K = 216, 12 of 16 rows, half of each row
  // zeros): 7 K-steps x 6 per 16 voxels (+ conv2_3 on the fp32 pipe either way); fp32 row kernel = 648 MFMAs per 128 voxels
  const double bc_f32 = 648.0 * cf / 128, bc_bf = 7.0 * 6 * cb / 16;
  printf("matrix-pipe time per voxel, kernel A : fp32 4x4x1 %.4f ns   6-product split bf16 %.4f ns   (ratio %.2f)\n", 1e3 * a_f32, 1e3 * a_bf, a_bf / a_f32);
  printf("matrix-pipe time per voxel, kernel BC: fp32 4x4x1 %.4f ns   6-product split bf16 %.4f ns   (ratio %.2f)\n", 1e3 * bc_f32, 1e3 * bc_bf, bc_bf / bc_f32);
  // ---- error of the split against fp64 (host): K = 432 products, operands like the activations / weights of the 64^3 stage
  double worst_split = 0, worst_f32 = 0, worst_split3 = 0;
  for (int trial = 0; trial < 2000; ++trial) {
    double ref = 0; float f = 0.f, s6 = 0.f, s3 = 0.f;
    for (int k = 0; k < 432; ++k) {
      const float x = std::fmax((float)rand() / RAND_MAX * 2.f - 0.6f, 0.f), w = ((float)rand() / RAND_MAX - 0.5f) * 0.3f;
      ref += (double)x * w;
      f = fmaf(x, w, f);
      const float x1 = bf16_round(x), x2 = bf16_round(x - x1), x3 = bf16_round(x - x1 - x2);
      const float w1 = bf16_round(w), w2 = bf16_round(w - w1), w3 = bf16_round(w - w1 - w2);
      // products of bf16 terms are exact in fp32; the pipe accumulates in fp32
      s6 += x1 * w1; s6 += x1 * w2; s6 += x2 * w1; s6 += x2 * w2; s6 += x1 * w3; s6 += x3 * w1;
      s3 += x1 * w1; s3 += x1 * w2; s3 += x2 * w1;
    }
    worst_f32 = std::fmax(worst_f32, std::fabs(f - ref));
    worst_split = std::fmax(worst_split, std::fabs(s6 - ref));
    worst_split3 = std::fmax(worst_split3, std::fabs(s3 - ref));
  }
  printf("max |sum - fp64| over 2000 dot products of K = 432: fp32 FMA chain %.3g, 6-product split %.3g, 3-product split %.3g\n", worst_f32, worst_split, worst_split3);
  return 0;
}
