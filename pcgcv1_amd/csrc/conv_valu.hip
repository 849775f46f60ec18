// LDS-tiled direct (VALU) 3x3x3 convolution for layers whose channel counts are too small for the
// matrix cores: conv_in (1 -> 16, models/model_voxception.py:83-88), deconv_out (16 -> 1, :188-192) and —
// as an alternative to the row-packed MFMA form — the 4/8-channel VRN layers.
//
// One thread per output voxel, all COUT accumulators in registers; the NDHWC input tile with its halo
// is staged in LDS; weights are wave-uniform, so they are fetched with scalar loads (SGPR operands of
// v_fmac) and cost no vector or LDS bandwidth.  fp32 FMA chain in fixed (tap, channel) order.
#include "mfma_common.h"

namespace pcgc {

template <int CIN, int COUT>
__global__ void __launch_bounds__(256) conv_valu_kernel(ConvArgs a) {
  constexpr int TD = 4, TH = 4, TW = 16;
  constexpr int ID = TD + 2, IH = TH + 2, IW = TW + 2;
  constexpr int CQ = (CIN + 3) / 4;                     // float4 per voxel (CIN = 1 handled separately)
  constexpr int VS = CIN >= 4 ? (CIN == 16 ? 20 : (CIN == 8 ? 12 : CIN)) : 1;
  __shared__ __attribute__((aligned(16))) float tile[ID * IH * IW * VS];

  const int tw = a.Dout / TW, th = a.Dout / TH, td = a.Dout / TD;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  const int b = bid;
  const int od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
  const float* xb = a.x + (int64_t)b * a.Din * a.Din * a.Din * a.x_cs + a.x_co;

  if constexpr (CIN >= 4) {
    stage_tile<ID, IH, IW, CQ, VS>(tile, xb, a.Din, a.x_cs, od0 - 1, oh0 - 1, ow0 - 1);
  } else {
    for (int v = threadIdx.x; v < ID * IH * IW; v += 256) {
      const int zw = v % IW, zh = (v / IW) % IH, zd = v / (IW * IH);
      const int gd = od0 - 1 + zd, gh = oh0 - 1 + zh, gw = ow0 - 1 + zw;
      float val = 0.f;
      if ((unsigned)gd < (unsigned)a.Din && (unsigned)gh < (unsigned)a.Din && (unsigned)gw < (unsigned)a.Din)
        val = xb[(((int64_t)gd * a.Din + gh) * a.Din + gw) * a.x_cs];
      tile[v] = val;
    }
  }
  __syncthreads();

  const int w = threadIdx.x & 15, h = (threadIdx.x >> 4) & 3, d = threadIdx.x >> 6;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
  const float* __restrict__ wt = a.w;                  // TF layout [tap][ci][co], wave-uniform addresses

#pragma unroll 1
  for (int kd = 0; kd < 3; ++kd) {
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tap = (kd * 3 + kh) * 3 + kw;
        const float* xp = &tile[(((d + kd) * IH + (h + kh)) * IW + (w + kw)) * VS];
        if constexpr (CIN >= 4) {
#pragma unroll
          for (int q = 0; q < CQ; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + 4 * q);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int c = 0; c < COUT; ++c) acc[c] = fmaf(xs[r], wt[(tap * CIN + 4 * q + r) * COUT + c], acc[c]);
          }
        } else {
          const float xs = xp[0];
#pragma unroll
          for (int c = 0; c < COUT; ++c) acc[c] = fmaf(xs, wt[tap * COUT + c], acc[c]);
        }
      }
    }
  }
  const int64_t vox = (((int64_t)b * a.Dout + od0 + d) * a.Dout + oh0 + h) * a.Dout + ow0 + w;
  const int64_t eo = vox * a.y_cs + a.y_co;
  float* yp = a.y + eo;
  // residual / accumulate / mask operands: one 16-byte load per 4 channels when the layout allows it
  float rv[COUT], av[COUT], mv[COUT];
  if constexpr (COUT % 4 == 0) {
#pragma unroll
    for (int c = 0; c < COUT; c += 4) {
      if (a.res) { const float4 t = *reinterpret_cast<const float4*>(a.res + eo + c); rv[c] = t.x; rv[c + 1] = t.y; rv[c + 2] = t.z; rv[c + 3] = t.w; }
      if (a.add_to) { const float4 t = *reinterpret_cast<const float4*>(a.add_to + eo + c); av[c] = t.x; av[c + 1] = t.y; av[c + 2] = t.z; av[c + 3] = t.w; }
      if (a.mask) { const float4 t = *reinterpret_cast<const float4*>(a.mask + eo + c); mv[c] = t.x; mv[c + 1] = t.y; mv[c + 2] = t.z; mv[c + 3] = t.w; }
    }
  } else {
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
      if (a.res) rv[c] = a.res[eo + c];
      if (a.add_to) av[c] = a.add_to[eo + c];
      if (a.mask) mv[c] = a.mask[eo + c];
    }
  }
#pragma unroll
  for (int c = 0; c < COUT; ++c) {
    float v = acc[c];
    if (a.bias) v += a.bias[c];
    if (a.relu) v = fmaxf(v, 0.f);
    if (a.absval) v = fmaxf(fabsf(v), a.lower_bound);
    if (a.res) v = fmaxf(rv[c] + v, 0.f);
    if (a.add_to) v += av[c];
    if (a.mask) v = mv[c] > 0.f ? v : 0.f;
    acc[c] = v;
  }
  if constexpr (COUT % 4 == 0) {
#pragma unroll
    for (int c = 0; c < COUT; c += 4) *reinterpret_cast<float4*>(yp + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
  } else {
#pragma unroll
    for (int c = 0; c < COUT; ++c) yp[c] = acc[c];
  }
}

template <int CIN, int COUT>
static int run_valu(const ConvArgs& a, hipStream_t s) {
  const int blocks = a.B * (a.Dout / 4) * (a.Dout / 4) * (a.Dout / 16);
  hipLaunchKernelGGL((conv_valu_kernel<CIN, COUT>), dim3(blocks), dim3(256), 0, s, a);
  int rc = launch_ok("conv_valu_kernel");
  return rc ? rc : 1;
}

// a.w must be the TF-layout weights.  Returns 1 launched / would launch, 0 unsupported, <0 error.
int launch_conv_valu(const ConvArgs& a, hipStream_t s, bool run) {
  if (a.mode != 0 || a.ksize != 3 || a.Dout % 16) return 0;
  if (a.x_q4 || a.y_q4) return 0;
  if (a.Cin >= 4 && (a.x_cs % 4 || a.x_co % 4)) return 0;
  if (a.Cout % 4 == 0 && (a.y_cs % 4 || a.y_co % 4)) return 0;
#define TRY(ci, co)                             \
  if (a.Cin == ci && a.Cout == co) {            \
    if (!run) return 1;                         \
    return run_valu<ci, co>(a, s);              \
  }
  TRY(1, 16) TRY(16, 1) TRY(16, 4) TRY(4, 8) TRY(4, 4)
  TRY(4, 16) TRY(8, 4)        // adjoints of the C = 16 VRN layers (training, pcgc_conv3d_bwd_data)
#undef TRY
  return 0;
}

}  // namespace pcgc
