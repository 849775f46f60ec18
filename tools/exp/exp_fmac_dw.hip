// Stand-alone probe (GPU box): can a weight gradient of the 64^3 blocks run on the VECTOR pipe at the rate the matrix pipe has?
//   dW[kd,kh,kw][ci][co] = sum_v x[v + off][ci] * g[v][co]
// is a reduction over voxels with 4-16 output columns: on v_mfma_f32_16x16x4 a quarter of the columns is padding and every
// instruction needs fresh operands (conv_dw_mfma_16xn_kernel<4, true>: 117 us per 8 cubes = 0.41 of the peak).  Row scheme
// instead: lane = voxel w of a row, one accumulator REGISTER per (tap, ci, co) of the wave's share, v_fmac_f32 with the kw
// shift folded in as a DPP operand (wave_shr / wave_shl, zero fill = the cube face), a butterfly over the 64 lanes at the
// end.  A wave of the role (ci quad q, co pair p) keeps 27 x 4 x 2 = 216 accumulators; per g row it reads 9 x row quads and
// one g row and issues 216 FMACs = 432 cycles.  This probe runs exactly that instruction stream (operands from global /
// L2, 8 role waves per workgroup, 2 per SIMD) and prints the rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_fmac_dw.hip -o tools/exp/_build/exp_fmac_dw && tools/exp/_build/exp_fmac_dw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

__device__ __forceinline__ void fmac(float& a, float x, float g) { asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(g)); }
__device__ __forceinline__ void fmac_shr(float& a, float x, float g) {   // a += x[lane - 1] * g, lane 0 reads 0
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(x), "v"(g));
}
__device__ __forceinline__ void fmac_shl(float& a, float x, float g) {   // a += x[lane + 1] * g, lane 63 reads 0
  asm("v_fmac_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(x), "v"(g));
}

// x: Q4 [d][h][4 quads][w 64][4], g: [d][h][w 64][4]; one workgroup = 8 role waves on `rows` consecutive rows of plane pd
__device__ __forceinline__ void fmac_rshr(float& a, float x, float g) {   // row_shr:1 (inside each 16-lane row)
  asm("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(x), "v"(g));
}
__device__ __forceinline__ void fmac_rshl(float& a, float x, float g) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(x), "v"(g));
}
template <int MODE>
__global__ void __launch_bounds__(512, 2) fmac_dw_probe(const float* x, const float* g, float* out, int rows, int planes) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = wave >> 1, cp = wave & 1;
  const int tile = blockIdx.x;
  const int pd = 1 + tile % planes;
  float acc[27][4][2];
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) { acc[t][c][0] = 0.f; acc[t][c][1] = 0.f; }
  const f32x4* xq = reinterpret_cast<const f32x4*>(x) + lane;
  const f32x4* gq = reinterpret_cast<const f32x4*>(g) + lane;
#pragma unroll 1
  for (int h = 1; h <= rows; ++h) {
    const f32x4 gv = gq[(size_t)(pd * 66 + h) * 64];
    const float g0 = gv[cp * 2], g1 = gv[cp * 2 + 1];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const f32x4 xv = xq[(size_t)(((pd + kd - 1) * 66 + h + kh - 1) * 4 + q) * 64];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float x0 = xv[c];
          const int t = (kd * 3 + kh) * 3;
          // the kw = 0 / 2 taps: the lane shift rides in the FMAC as a DPP operand (hipcc would emit v_mov_b32_dpp + v_pk_fma_f32)
          if constexpr (MODE == 0) {
            fmac_shr(acc[t + 0][c][0], x0, g0); fmac_shr(acc[t + 0][c][1], x0, g1);
            fmac(acc[t + 1][c][0], x0, g0); fmac(acc[t + 1][c][1], x0, g1);
            fmac_shl(acc[t + 2][c][0], x0, g0); fmac_shl(acc[t + 2][c][1], x0, g1);
          } else if constexpr (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { acc[t + k][c][0] = fmaf(x0, g0, acc[t + k][c][0]); acc[t + k][c][1] = fmaf(x0, g1, acc[t + k][c][1]); }
          } else if constexpr (MODE == 2) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { fmac(acc[t + k][c][0], x0, g0); fmac(acc[t + k][c][1], x0, g1); }
          } else {
            fmac_rshr(acc[t + 0][c][0], x0, g0); fmac_rshr(acc[t + 0][c][1], x0, g1);
            fmac(acc[t + 1][c][0], x0, g0); fmac(acc[t + 1][c][1], x0, g1);
            fmac_rshl(acc[t + 2][c][0], x0, g0); fmac_rshl(acc[t + 2][c][1], x0, g1);
          }
        }
      }
  }
  float s = 0.f;                                            // keep every accumulator alive
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) s += acc[t][c][0] + acc[t][c][1];
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  const int planes = 8, R = 66;
  std::vector<float> hx((size_t)(planes + 2) * R * 4 * 64 * 4, 0.5f), hg((size_t)(planes + 2) * R * 64 * 4, 0.25f);
  float *x, *g, *out;
  hipMalloc(&x, hx.size() * 4); hipMalloc(&g, hg.size() * 4); hipMalloc(&out, 4096 * 512 * 4);
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"v_fmac_f32 + wave_shr / wave_shl DPP (2 of 3)", "fmaf, no shifts (hipcc: v_pk_fma_f32)", "v_fmac_f32_e32, no shifts",
                          "v_fmac_f32 + row_shr / row_shl DPP (2 of 3)"};
  for (int mode = 0; mode < 4; ++mode)
    for (int blocks : {256, 512}) {
      const int rows = 64;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) {
          if (mode == 0) hipLaunchKernelGGL(fmac_dw_probe<0>, dim3(blocks), dim3(512), 0, 0, x, g, out, rows, planes);
          else if (mode == 1) hipLaunchKernelGGL(fmac_dw_probe<1>, dim3(blocks), dim3(512), 0, 0, x, g, out, rows, planes);
          else if (mode == 2) hipLaunchKernelGGL(fmac_dw_probe<2>, dim3(blocks), dim3(512), 0, 0, x, g, out, rows, planes);
          else hipLaunchKernelGGL(fmac_dw_probe<3>, dim3(blocks), dim3(512), 0, 0, x, g, out, rows, planes);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 64 * 216.0 * 8 * rows * blocks * 20;
        if (rep == 2) printf("%-50s blocks %4d: %.1f us per launch, %.1f TFLOP/s (%.2f of 157.3)\n", names[mode], blocks, ms * 1e3 / 20, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
      }
    }
  return 0;
}
