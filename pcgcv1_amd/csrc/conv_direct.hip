// Direct (VALU) 3-D convolution for gfx950: the shape-generic kernel.
//
// One thread per output voxel and per group of CO_T output channels; weights are
// wave-uniform (scalar loads), activations come straight from global memory
// (L1/L2 absorb the 27-fold tap reuse).  It covers every layer shape of the
// reference's layer tables (models/model_voxception.py:21-54, 83-122, 153-192,
// 224-244, 263-297) including the ones the MFMA kernels do not take (8^3 hyper
// layers, 1-channel input/output convs, cube sizes other than 64), and it is the
// on-device cross-check for the MFMA kernels.  Fixed summation order
// (tap-major, then input channel, fmaf chain) => bit-reproducible.
//
// Semantics (SURVEY.md §8a row a7 = Keras padding='same'):
//   mode 0: y[o] = b + sum_k x[o + k - (K-1)/2] W[k]              zero outside
//   mode 1: y[o] = b + sum_k x[2o + k - pb] W[k]                  pb = (K-2)/2: TF pads K-2 in total, the smaller
//                                                                  half in front (K = 3: 0/1, 5: 1/2, 9: 3/4)
//   mode 2: y[o] = b + sum_{2i+k-pb=o} x[i] W[k]  (W is [k,Cout,Cin]) o in [0, 2*Din): adjoint of mode 1
// K is any odd size (3 and 1 in model_voxception.py; 5 and 9 in model_simple.py:20-41, 56-86).
#include "common.h"

namespace pcgc {

template <int CO_T>
__global__ void __launch_bounds__(256) conv_direct_kernel(ConvArgs a) {
  const int64_t total = (int64_t)a.B * a.Dout * a.Dout * a.Dout;
  const int64_t vox = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (vox >= total) return;
  const int co0 = blockIdx.y * CO_T;
  const int ow = (int)(vox % a.Dout);
  const int oh = (int)((vox / a.Dout) % a.Dout);
  const int od = (int)((vox / ((int64_t)a.Dout * a.Dout)) % a.Dout);
  const int b = (int)(vox / ((int64_t)a.Dout * a.Dout * a.Dout));
  const int K = a.ksize;
  const int pad = (K - 1) / 2;
  const int pb = (K > 2 ? K - 2 : 0) / 2;

  float acc[CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; ++j) acc[j] = 0.f;

  for (int kd = 0; kd < K; ++kd) {
    int id;
    bool vd;
    if (a.mode == 0) { id = od + kd - pad; vd = (id >= 0 && id < a.Din); }
    else if (a.mode == 1) { id = 2 * od + kd - pb; vd = (id >= 0 && id < a.Din); }
    else { int t = od - kd + pb; id = t >> 1; vd = (t >= 0) && !(t & 1) && (id < a.Din); }
    for (int kh = 0; kh < K; ++kh) {
      int ih;
      bool vh;
      if (a.mode == 0) { ih = oh + kh - pad; vh = (ih >= 0 && ih < a.Din); }
      else if (a.mode == 1) { ih = 2 * oh + kh - pb; vh = (ih >= 0 && ih < a.Din); }
      else { int t = oh - kh + pb; ih = t >> 1; vh = (t >= 0) && !(t & 1) && (ih < a.Din); }
      for (int kw = 0; kw < K; ++kw) {
        int iw;
        bool vw;
        if (a.mode == 0) { iw = ow + kw - pad; vw = (iw >= 0 && iw < a.Din); }
        else if (a.mode == 1) { iw = 2 * ow + kw - pb; vw = (iw >= 0 && iw < a.Din); }
        else { int t = ow - kw + pb; iw = t >> 1; vw = (t >= 0) && !(t & 1) && (iw < a.Din); }
        if (!(vd && vh && vw)) continue;
        const float* xp = a.x + ((((int64_t)b * a.Din + id) * a.Din + ih) * a.Din + iw) * a.x_cs + a.x_co;
        const int tap = (kd * K + kh) * K + kw;
        if (a.mode != 2) {
          const float* wp = a.w + (int64_t)tap * a.Cin * a.Cout + co0;
          for (int ci = 0; ci < a.Cin; ++ci) {
            const float xv = xp[ci];
#pragma unroll
            for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(xv, wp[(int64_t)ci * a.Cout + j], acc[j]);
          }
        } else {
          const float* wp = a.w + ((int64_t)tap * a.Cout + co0) * a.Cin;
          for (int ci = 0; ci < a.Cin; ++ci) {
            const float xv = xp[ci];
#pragma unroll
            for (int j = 0; j < CO_T; ++j) acc[j] = fmaf(xv, wp[(int64_t)j * a.Cin + ci], acc[j]);
          }
        }
      }
    }
  }
  float* yp = a.y + vox * a.y_cs + a.y_co + co0;
  const float* rp = a.res ? a.res + vox * a.y_cs + a.y_co + co0 : nullptr;
#pragma unroll
  for (int j = 0; j < CO_T; ++j) {
    float v = acc[j];
    if (a.bias) v += a.bias[co0 + j];
    if (a.relu) v = fmaxf(v, 0.f);
    if (a.absval) v = fmaxf(fabsf(v), a.lower_bound);
    if (rp) v = fmaxf(rp[j] + v, 0.f);
    if (a.add_to) v += a.add_to[vox * a.y_cs + a.y_co + co0 + j];
    if (a.mask) v = a.mask[vox * a.y_cs + a.y_co + co0 + j] > 0.f ? v : 0.f;
    yp[j] = v;
  }
}

int launch_conv_direct(const ConvArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.B * a.Dout * a.Dout * a.Dout;
  if (total == 0) return 0;
  PCGC_REQUIRE(!a.x_q4 && !a.y_q4, "conv_direct: Q4 tensors are not supported by the direct kernel");
  PCGC_REQUIRE(total < ((int64_t)1 << 31) * 256, "conv_direct: too many voxels");
  int cot = 1;
  if (a.Cout % 16 == 0) cot = 16;
  else if (a.Cout % 8 == 0) cot = 8;
  else if (a.Cout % 4 == 0) cot = 4;
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)(a.Cout / cot));
  switch (cot) {
    case 16: hipLaunchKernelGGL(conv_direct_kernel<16>, grid, dim3(256), 0, s, a); break;
    case 8: hipLaunchKernelGGL(conv_direct_kernel<8>, grid, dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(conv_direct_kernel<4>, grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(conv_direct_kernel<1>, grid, dim3(256), 0, s, a); break;
  }
  return launch_ok("conv_direct_kernel");
}

}  // namespace pcgc
