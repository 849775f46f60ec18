// Voxception-ResNet block (_VoxceptionResNet.call, models/model_voxception.py:56-68) for C = 16 at full
// resolution, as two LDS-tiled VALU kernels.
//
// At C = 16 the block's layers have 4 or 8 output channels; the fp32 matrix cores would run them with
// 44-75 % of their rows multiplying zeros (see conv_mfma.hip, row packing), while fp32 VALU FMAs with
// wave-uniform (scalar-register) weights do only useful work at the same peak rate.  Measured on MI355X
// (tools/bench_conv.py) the VALU form is 1.4-1.8x faster for these shapes, and it leaves the MFMA pipe
// free for other waves.
//
//   kernel A   x[16]  -> t12[8] = [ relu(conv1_1(x)) (3x3x3, 16->4) | relu(conv2_1(x)) (1x1x1, 16->4) ]
//   kernel BC  t12[8] -> out[16] = relu( x + [ relu(conv1_2(t12[0:4])) (3x3x3, 4->8)
//                                            | relu(conv2_3(relu(conv2_2(t12[4:8])))) (3x3x3 4->4, 1x1x1 4->8) ] )
// One thread per output voxel (4 x 4 x 16 voxels per 256-thread workgroup), accumulators in registers,
// input tile + halo staged in LDS, fixed (tap, channel) fp32 FMA order: deterministic, batch-invariant.
#include "mfma_common.h"

namespace pcgc {

struct VrnArgs {
  const float* x;      // block input  [B, D^3, 16]
  float* t12;          // scratch      [B, D^3, 8]
  float* out;          // block output [B, D^3, 16]
  const float *w11, *b11, *w21, *b21;   // conv1_1 [27][16][4], conv2_1 [16][4]
  const float *w12, *b12, *w22, *b22, *w23, *b23;   // conv1_2 [27][4][8], conv2_2 [27][4][4], conv2_3 [4][8]
  int B, D;
};

constexpr int kTD = 4, kTH = 4, kTW = 16;
constexpr int kID = kTD + 2, kIH = kTH + 2, kIW = kTW + 2;

__device__ __forceinline__ void tile_coords(int D, int& b, int& od0, int& oh0, int& ow0) {
  const int tw = D / kTW, th = D / kTH, td = D / kTD;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  b = bid; od0 = tx * kTD; oh0 = ty * kTH; ow0 = tz * kTW;
}

__global__ void __launch_bounds__(256) vrn16_a_kernel(VrnArgs a) {
  // The 16 input channels go through LDS in two passes of 8 (31 KB per workgroup instead of 52 KB): five
  // workgroups per CU instead of three hide the LDS / scalar-load latency of the FMA loop (measured 117 -> 99 us
  // per 8 cubes, tools/exp/exp_valu.hip).  Summation order: (channel half, tap, channel).
  constexpr int VS = 12, CK = 8;
  __shared__ __attribute__((aligned(16))) float tile[kID * kIH * kIW * VS];
  int b, od0, oh0, ow0;
  tile_coords(a.D, b, od0, oh0, ow0);
  const int w = threadIdx.x & 15, h = (threadIdx.x >> 4) & 3, d = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float acc2[4] = {0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ w11 = a.w11;
  const float* __restrict__ w21 = a.w21;
  for (int cb = 0; cb < 16 / CK; ++cb) {
    if (cb) __syncthreads();
    stage_tile<kID, kIH, kIW, CK / 4, VS>(tile, a.x + (int64_t)b * a.D * a.D * a.D * 16 + cb * CK, a.D, 16, od0 - 1, oh0 - 1,
                                          ow0 - 1);
    __syncthreads();
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
#pragma unroll 1
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = (kd * 3 + kh) * 3 + kw;
          const float* xp = &tile[(((d + kd) * kIH + (h + kh)) * kIW + (w + kw)) * VS];
#pragma unroll
          for (int q = 0; q < CK / 4; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + 4 * q);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[c] = fmaf(xs[r], w11[(tap * 16 + cb * CK + 4 * q + r) * 4 + c], acc[c]);
          }
        }
      }
    }
    // conv2_1 on the centre voxel
    const float* xp = &tile[(((d + 1) * kIH + (h + 1)) * kIW + (w + 1)) * VS];
#pragma unroll
    for (int q = 0; q < CK / 4; ++q) {
      const float4 xv = *reinterpret_cast<const float4*>(xp + 4 * q);
      const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc2[c] = fmaf(xs[r], w21[(cb * CK + 4 * q + r) * 4 + c], acc2[c]);
    }
  }
  const int64_t vox = (((int64_t)b * a.D + od0 + d) * a.D + oh0 + h) * a.D + ow0 + w;
  float* tp = a.t12 + vox * 8;
  *reinterpret_cast<float4*>(tp) = make_float4(fmaxf(acc[0] + a.b11[0], 0.f), fmaxf(acc[1] + a.b11[1], 0.f),
                                               fmaxf(acc[2] + a.b11[2], 0.f), fmaxf(acc[3] + a.b11[3], 0.f));
  *reinterpret_cast<float4*>(tp + 4) = make_float4(fmaxf(acc2[0] + a.b21[0], 0.f), fmaxf(acc2[1] + a.b21[1], 0.f),
                                                   fmaxf(acc2[2] + a.b21[2], 0.f), fmaxf(acc2[3] + a.b21[3], 0.f));
}

__global__ void __launch_bounds__(256) vrn16_bc_kernel(VrnArgs a) {
  constexpr int VS = 12;
  __shared__ __attribute__((aligned(16))) float tile[kID * kIH * kIW * VS];
  int b, od0, oh0, ow0;
  tile_coords(a.D, b, od0, oh0, ow0);
  stage_tile<kID, kIH, kIW, 2, VS>(tile, a.t12 + (int64_t)b * a.D * a.D * a.D * 8, a.D, 8, od0 - 1, oh0 - 1, ow0 - 1);
  __syncthreads();
  const int w = threadIdx.x & 15, h = (threadIdx.x >> 4) & 3, d = threadIdx.x >> 6;
  float p1[8], p2[4];
#pragma unroll
  for (int c = 0; c < 8; ++c) p1[c] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) p2[c] = 0.f;
  const float* __restrict__ w12 = a.w12;
  const float* __restrict__ w22 = a.w22;
#pragma unroll 1
  for (int kd = 0; kd < 3; ++kd) {
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tap = (kd * 3 + kh) * 3 + kw;
        const float* xp = &tile[(((d + kd) * kIH + (h + kh)) * kIW + (w + kw)) * VS];
        const float4 u = *reinterpret_cast<const float4*>(xp);          // tensor1_1
        const float4 v = *reinterpret_cast<const float4*>(xp + 4);      // tensor2_1
        const float us[4] = {u.x, u.y, u.z, u.w}, vs[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int c = 0; c < 8; ++c) p1[c] = fmaf(us[r], w12[(tap * 4 + r) * 8 + c], p1[c]);
#pragma unroll
          for (int c = 0; c < 4; ++c) p2[c] = fmaf(vs[r], w22[(tap * 4 + r) * 4 + c], p2[c]);
        }
      }
    }
  }
  // tensor2_2 -> conv2_3
  float t[4], q3[8];
#pragma unroll
  for (int c = 0; c < 4; ++c) t[c] = fmaxf(p2[c] + a.b22[c], 0.f);
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) s = fmaf(t[c], a.w23[c * 8 + o], s);
    q3[o] = fmaxf(s + a.b23[o], 0.f);
  }
  const int64_t vox = (((int64_t)b * a.D + od0 + d) * a.D + oh0 + h) * a.D + ow0 + w;
  const float* xr = a.x + vox * 16;
  float* op = a.out + vox * 16;
#pragma unroll
  for (int c = 0; c < 8; c += 4) {
    const float4 r = *reinterpret_cast<const float4*>(xr + c);
    *reinterpret_cast<float4*>(op + c) =
        make_float4(fmaxf(r.x + fmaxf(p1[c] + a.b12[c], 0.f), 0.f), fmaxf(r.y + fmaxf(p1[c + 1] + a.b12[c + 1], 0.f), 0.f),
                    fmaxf(r.z + fmaxf(p1[c + 2] + a.b12[c + 2], 0.f), 0.f), fmaxf(r.w + fmaxf(p1[c + 3] + a.b12[c + 3], 0.f), 0.f));
  }
#pragma unroll
  for (int c = 0; c < 8; c += 4) {
    const float4 r = *reinterpret_cast<const float4*>(xr + 8 + c);
    *reinterpret_cast<float4*>(op + 8 + c) = make_float4(fmaxf(r.x + q3[c], 0.f), fmaxf(r.y + q3[c + 1], 0.f),
                                                         fmaxf(r.z + q3[c + 2], 0.f), fmaxf(r.w + q3[c + 3], 0.f));
  }
}

// which: 0 = kernel A, 1 = kernel BC.  D must be a multiple of 16.
int launch_vrn16_valu(const float* x, float* t12, float* out, const float* const* w, int B, int D, int which, hipStream_t s) {
  if (D % 16) return 0;
  VrnArgs a;
  a.x = x; a.t12 = t12; a.out = out;
  a.w11 = w[0]; a.b11 = w[1]; a.w12 = w[2]; a.b12 = w[3]; a.w21 = w[4]; a.b21 = w[5];
  a.w22 = w[6]; a.b22 = w[7]; a.w23 = w[8]; a.b23 = w[9];
  a.B = B; a.D = D;
  const int blocks = B * (D / kTD) * (D / kTH) * (D / kTW);
  if (which == 0) hipLaunchKernelGGL(vrn16_a_kernel, dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(vrn16_bc_kernel, dim3(blocks), dim3(256), 0, s, a);
  int rc = launch_ok("vrn16 valu kernel");
  return rc ? rc : 1;
}

}  // namespace pcgc
