// Row kernels (v_mfma_f32_4x4x1_16B_f32, scheme of vrn_row.hip) for the 8^3 layers of the hyperprior networks
// (models/model_voxception.py:217-308), which the tile kernels cannot take (their tiles are 16 voxels wide) and which
// ran on the one-thread-per-voxel direct kernel before.  An 8 x 8 plane of a cube is ONE 64-lane vector
// ("plane vector": lane = (row r = lane >> 3, voxel i = lane & 7)); tensors are NDHWC.
//   up8_row_kernel  : HyperDecoder.conv2, stride-2 transposed conv 3^3 16 -> 16, 8^3 -> 16^3
//   conv8_row_kernel: HyperDecoder.conv1 (8 -> 16) and HyperEncoder.conv3 (16 -> 8), 3^3 stride 1 at 8^3
//   down8_row_kernel: HyperEncoder.conv2, stride-2 conv 3^3 16 -> 16, 16^3 -> 8^3
// Summation order per output: bias, then (input plane, channel quad, channel, taps in program order): fixed, independent of the
// batch size and of the cube's position in the batch (encoder and decoder run the same launch geometry per cube).
#include <type_traits>
#include "row_common.h"

namespace pcgc {

constexpr int kH = 8;                     // cube edge of the hyper latents

// lane i <- lane i-1 / i+1 inside each 8-lane row, zero at the row ends ('same' padding)
__device__ __forceinline__ float shr8(float v, bool first) { const float s = shr1(v); return first ? 0.f : s; }
__device__ __forceinline__ float shl8(float v, bool last) { const float s = shl1(v); return last ? 0.f : s; }

struct HyperRowArgs {
  const float* x;
  float* y;
  const float* w;
  const float* bias;
  int B, relu;
  int ps = kH;       // conv8 / down8: output planes per wave (8 = the whole cube; small batches — the training step's 8
                     // cubes — take 2, four times the waves; the sums per output do not depend on it)
};

// plane vector of plane p, rows r + dr (dr = -1, 0, +1), channel quad q of an NDHWC tensor [8][8][8][C], C = 4 * NQ
template <int NQ>
__device__ __forceinline__ f32x4 load_plane(i32x4 rs, int lane, int p, int q, int dr) {
  const int r = (lane >> 3) + dr;
  const bool ok = (unsigned)p < (unsigned)kH && (unsigned)r < (unsigned)kH;
  return raw_load4(rs, ok ? (((p * kH + r) * kH + (lane & 7)) * NQ + q) * 16 : kOOB, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Transposed conv, stride 2: y[o] = bias + sum_{o = 2i + k} x[i] W[k] per axis (alignment of tconv_mfma_kernel): an even
// output o = 2i takes k = 0 from input i and k = 2 from input i - 1, an odd output o = 2i + 1 takes k = 1 from input i.
// Lane = input voxel; it owns the 2 x 2 outputs (oh, ow) of its voxel in each output plane; along d the wave slides over
// the input planes with three accumulator sets (plane 2p: kd = 0 now + kd = 2 carried, plane 2p+1: kd = 1, plane 2p+2:
// kd = 2 carried on) — exactly up2_row_kernel (vrn_row32.hip) with a plane vector instead of a row pair.
// One wave = one cube, one output-channel quad, LD input planes.  w: TF Conv3DTranspose layout [27][Cout=16][Cin=16].
// ---------------------------------------------------------------------------------------------------------------
template <int LD>
__global__ void __launch_bounds__(256) up8_row_kernel(HyperRowArgs a) {
  constexpr int CH = 27 * 16;                               // floats of a (channel quad, cout quad) chunk: [tap][ci4][4 couts]
  constexpr int NW = (CH + 63) / 64;                        // 7 registers, 4 taps each
  __shared__ float wl[16 * CH + 64];
  for (int i = threadIdx.x; i < 16 * CH; i += 256) {
    const int qg = i / CH, f = i - qg * CH, q = qg >> 2, g = qg & 3;
    const int tap = f >> 4, c = (f >> 2) & 3, co = f & 3;
    wl[i] = a.w[(tap * 16 + g * 4 + co) * 16 + 4 * q + c];
  }
  if (threadIdx.x < 64) wl[16 * CH + threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const bool first = (lane & 7) == 0;
  int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int g = wv & 3; wv >>= 2;
  const int d0 = (wv % (kH / LD)) * LD; wv /= (kH / LD);
  const int b = wv;
  if (b >= a.B) return;
  f32x4 bi = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bi = f32x4{a.bias[g * 4], a.bias[g * 4 + 1], a.bias[g * 4 + 2], a.bias[g * 4 + 3]};
  f32x4 acc[3][2][2];                                        // [set][oh parity][ow parity]
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) acc[s_][ph][pw] = bi;
  const i32x4 rs = make_rsrc(a.x + (size_t)b * kH * kH * kH * 16, kH * kH * kH * 16 * 4);
  const i32x4 ro = make_rsrc(a.y + (size_t)b * 4096 * 16, 4096 * 16 * 4);
  // output voxel (od, 2r + ph, 2i + pw), 16 channels of 4 B: this lane's cout quad g
  const int out_lane = ((2 * (lane >> 3)) * 16 + 2 * (lane & 7)) * 64 + g * 16;
  auto quad = [&](const f32x4& P, const f32x4& O, int q, bool v0, bool v2) {
    float W[NW];
#pragma unroll
    for (int v = 0; v < NW; ++v) W[v] = wl[(q * 4 + g) * CH + v * 64 + lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x0 = comp(P, c), x1 = comp(O, c);
      const float r0 = shr8(x0, first), r1 = shr8(x1, first);
      const bool vj[3] = {v0, v0, v2};
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        if (vj[s_]) {
          auto mf_ = [&](int kh, int kw, float xv, f32x4& d) {
            const int t = (s_ * 3 + kh) * 3 + kw;
            d = mfa((t & 3) * 4 + c, W[t >> 2], xv, d);
          };
          mf_(0, 0, x0, acc[s_][0][0]); mf_(0, 2, r0, acc[s_][0][0]); mf_(2, 0, x1, acc[s_][0][0]); mf_(2, 2, r1, acc[s_][0][0]);
          mf_(0, 1, x0, acc[s_][0][1]); mf_(2, 1, x1, acc[s_][0][1]);
          mf_(1, 0, x0, acc[s_][1][0]); mf_(1, 2, r0, acc[s_][1][0]);
          mf_(1, 1, x0, acc[s_][1][1]);
        }
      }
    }
  };
  auto store_plane = [&](int set, int od, bool ok) {
    const int base = ok ? od * (16 * 16 * 64) + out_lane : kOOB;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) {
        f32x4 v = acc[set][ph][pw];
        if (a.relu) v = relu4(v);
        raw_store4(v, ro, base + (ph * 16 + pw) * 64, 0, 0);
      }
  };
#pragma unroll 1
  for (int p = d0 - 1; p < d0 + LD; ++p) {
    const bool v0 = p >= d0, v2 = p >= 0 && p + 1 < d0 + LD;
    f32x4 P[4], O[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { P[q] = load_plane<4>(rs, lane, p, q, 0); O[q] = load_plane<4>(rs, lane, p, q, -1); }
#pragma unroll
    for (int q = 0; q < 4; ++q) quad(P[q], O[q], q, v0, v2);
    store_plane(0, 2 * p, v0);
    store_plane(1, 2 * p + 1, v0);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw) { acc[0][ph][pw] = acc[2][ph][pw]; acc[1][ph][pw] = bi; acc[2][ph][pw] = bi; }
  }
}

int launch_up8_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s) {
  HyperRowArgs a{x, y, w, bias, B, relu};
  constexpr int LD = 4;
  const int waves = B * (kH / LD) * 4;
  hipLaunchKernelGGL((up8_row_kernel<LD>), dim3((waves + 3) / 4), dim3(256), 0, s, a);
  return launch_ok("up8_row_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-1 conv 3^3 at 8^3, CIN -> COUT (8 -> 16, 16 -> 8): lane = output voxel of a plane; kw taps are lane shifts inside
// the 8-lane rows, kh taps the plane vectors of rows r - 1 / r + 1 (loaded), kd three sliding plane accumulators.
// One wave = one cube, one output-channel quad, all 8 planes.  w: TF layout [27][CIN][COUT].
// ---------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ void __launch_bounds__(256) conv8_row_kernel(HyperRowArgs a) {
  constexpr int NQI = CIN / 4, NQO = COUT / 4;
  constexpr int CH = 27 * 16;                               // chunk (channel quad, cout quad): [tap][ci4][4 couts]
  constexpr int NW = (CH + 63) / 64;
  __shared__ float wl[NQI * NQO * CH + 64];
  for (int i = threadIdx.x; i < NQI * NQO * CH; i += 256) {
    const int qg = i / CH, f = i - qg * CH, q = qg / NQO, g = qg % NQO;
    const int tap = f >> 4, c = (f >> 2) & 3, co = f & 3;
    wl[i] = a.w[(tap * CIN + 4 * q + c) * COUT + g * 4 + co];
  }
  if (threadIdx.x < 64) wl[NQI * NQO * CH + threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const bool first = (lane & 7) == 0, last = (lane & 7) == 7;
  int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int g = wv % NQO; wv /= NQO;
  const int nseg = kH / a.ps;
  const int p0 = (wv % nseg) * a.ps; wv /= nseg;            // this wave's output planes [p0, p0 + ps)
  const int b = wv;
  if (b >= a.B) return;
  f32x4 bi = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bi = f32x4{a.bias[g * 4], a.bias[g * 4 + 1], a.bias[g * 4 + 2], a.bias[g * 4 + 3]};
  f32x4 acc[3] = {bi, bi, bi};                              // output planes p - 1, p, p + 1 of input plane p
  const i32x4 rs = make_rsrc(a.x + (size_t)b * kH * kH * kH * CIN, kH * kH * kH * CIN * 4);
  const i32x4 ro = make_rsrc(a.y + (size_t)b * kH * kH * kH * COUT, kH * kH * kH * COUT * 4);
  const int out_lane = lane * (COUT * 4) + g * 16;
#pragma unroll 1
  for (int p = p0 - 1; p <= p0 + a.ps; ++p) {
    const bool pin = (unsigned)p < (unsigned)kH;
    // which of the output planes p - 1 / p / p + 1 are this wave's (and exist)
    const bool v0 = pin && p - 1 >= p0, v1 = pin && p >= p0 && p < p0 + a.ps, v2 = pin && p + 1 < p0 + a.ps;
#pragma unroll
    for (int q = 0; q < NQI && pin; ++q) {
      f32x4 X[3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) X[kh] = load_plane<NQI>(rs, lane, p, q, kh - 1);
      float W[NW];
#pragma unroll
      for (int v = 0; v < NW; ++v) W[v] = wl[(q * NQO + g) * CH + v * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x0[3], xm[3], xp[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) { x0[kh] = comp(X[kh], c); xm[kh] = shr8(x0[kh], first); xp[kh] = shl8(x0[kh], last); }
        const bool vj[3] = {v0, v1, v2};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int kd = 2 - j;                              // input plane p feeds output plane p - 1 + j with tap kd = 2 - j
          if (vj[j]) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) {
                const int t = (kd * 3 + kh) * 3 + kw;
                const float xv = kw == 0 ? xm[kh] : (kw == 1 ? x0[kh] : xp[kh]);
                acc[j] = mfa((t & 3) * 4 + c, W[t >> 2], xv, acc[j]);
              }
          }
        }
      }
    }
    if (p - 1 >= p0) {                                       // output plane p - 1 has seen its three input planes
      f32x4 v = acc[0];
      if (a.relu) v = relu4(v);
      raw_store4(v, ro, (p - 1) * (kH * kH * COUT * 4) + out_lane, 0, 0);
    }
    acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = bi;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// HyperEncoder.conv2: stride-2 conv 3^3, 16 -> 16, 16^3 -> 8^3: y[o] = bias + sum_k x[2o + k] W[k] per axis (one zero
// behind).  Lane = output voxel of a plane vector; the even / odd input voxels of a row are two strided loads (kw = 0 / 1),
// kw = 2 is the even vector shifted by a lane; output plane j reads input planes 2j, 2j+1, 2j+2 and 2j+2 is also plane
// j+1's kd = 0 (two accumulators) — down1_row_kernel (vrn_row32.hip) on a plane vector.  One wave = (cube, cout quad).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) down8_row_kernel(HyperRowArgs a) {
  constexpr int CH = 27 * 16, NW = (9 * 16 + 63) / 64;      // 3 registers per kd slice (the third is partly the next slice)
  __shared__ float wl[16 * CH + 64];
  for (int i = threadIdx.x; i < 16 * CH; i += 256) {
    const int qg = i / CH, f = i - qg * CH, q = qg >> 2, g = qg & 3;
    const int tap = f >> 4, c = (f >> 2) & 3, co = f & 3;
    wl[i] = a.w[(tap * 16 + 4 * q + c) * 16 + g * 4 + co];
  }
  if (threadIdx.x < 64) wl[16 * CH + threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const bool last = (lane & 7) == 7;
  int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int g = wv & 3; wv >>= 2;
  const int nseg = kH / a.ps;
  const int j0 = (wv % nseg) * a.ps;                        // this wave's output planes [j0, j0 + ps)
  const int b = wv / nseg;
  if (b >= a.B) return;
  f32x4 bi = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) bi = f32x4{a.bias[g * 4], a.bias[g * 4 + 1], a.bias[g * 4 + 2], a.bias[g * 4 + 3]};
  f32x4 cur = bi, nxt = bi;
  const i32x4 rs = make_rsrc(a.x + (size_t)b * 4096 * 16, 4096 * 16 * 4);
  const i32x4 ro = make_rsrc(a.y + (size_t)b * 512 * 16, 512 * 16 * 4);
  const int r2 = 2 * (lane >> 3), i2 = 2 * (lane & 7);
  // one input plane: tap slice kd = KA into accA, and kd = KB into accB when KB >= 0
  auto plane = [&](int p, auto KA_, f32x4& accA, auto KB_, f32x4& accB) {
    constexpr int KA = decltype(KA_)::value, KB = decltype(KB_)::value;   // compile-time: they select MFMA block immediates
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      f32x4 E[3], O[3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const bool ok = r2 + kh < 16;
        const int off = ok ? (((p * 16 + r2 + kh) * 16 + i2) * 4 + q) * 16 : kOOB;
        E[kh] = raw_load4(rs, off, 0, 0);
        O[kh] = raw_load4(rs, off + 64, 0, 0);
      }
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int kd = pass == 0 ? KA : KB;
        if (kd < 0) continue;
        f32x4& acc = pass == 0 ? accA : accB;
        float W[NW];
#pragma unroll
        for (int v = 0; v < NW; ++v) W[v] = wl[(q * 4 + g) * CH + ((kd * 9 * 16) & ~63) + v * 64 + lane];
        const int sh = (kd * 9 * 16) & 63;                    // float offset of the slice inside its first register (folds: kd is constant)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const float xe = comp(E[kh], c), xo = comp(O[kh], c), x2 = shl8(xe, last);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const float xv = kw == 0 ? xe : (kw == 1 ? xo : x2);
              const int fo = sh + (kh * 3 + kw) * 16 + c * 4;
              acc = mfa((fo & 63) >> 2, W[fo >> 6], xv, acc);
            }
          }
      }
    }
  };
  f32x4 dummy = bi;
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using KN = std::integral_constant<int, -1>;
  plane(2 * j0, K0{}, cur, KN{}, dummy);
#pragma unroll 1
  for (int j = j0; j < j0 + a.ps; ++j) {
    plane(2 * j + 1, K1{}, cur, KN{}, dummy);
    if (j + 1 < kH) plane(2 * j + 2, K2{}, cur, K0{}, nxt);   // input plane 16 does not exist (the zero behind)
    f32x4 v = cur;
    if (a.relu) v = relu4(v);
    raw_store4(v, ro, (j * 64 + lane) * 64 + g * 16, 0, 0);
    cur = nxt;
    nxt = bi;
  }
}

// output planes per wave: the whole cube when that already gives every SIMD a wave, else halved until it does
static int planes_per_wave(int waves_per_cube_at_8, int B) {
  int ps = kH;
  while (ps > 1 && (int64_t)B * waves_per_cube_at_8 * (kH / ps) < 1024) ps /= 2;
  return ps;
}

int launch_down8_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s) {
  HyperRowArgs a{x, y, w, bias, B, relu};
  a.ps = planes_per_wave(4, B);
  const int waves = B * 4 * (kH / a.ps);
  hipLaunchKernelGGL(down8_row_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, a);
  return launch_ok("down8_row_kernel");
}

int launch_conv8_row(const float* x, float* y, const float* w, const float* bias, int B, int Cin, int Cout, int relu, hipStream_t s) {
  HyperRowArgs a{x, y, w, bias, B, relu};
  if (Cin == 8 && Cout == 16) {
    a.ps = planes_per_wave(4, B);
    hipLaunchKernelGGL((conv8_row_kernel<8, 16>), dim3((B * 4 * (kH / a.ps) + 3) / 4), dim3(256), 0, s, a);
  } else if (Cin == 16 && Cout == 8) {
    a.ps = planes_per_wave(2, B);
    hipLaunchKernelGGL((conv8_row_kernel<16, 8>), dim3((B * 2 * (kH / a.ps) + 3) / 4), dim3(256), 0, s, a);
  } else {
    return 0;                                                // unsupported shape: the caller keeps its generic path
  }
  const int rc = launch_ok("conv8_row_kernel");
  return rc ? rc : 1;
}

// out = (mask > 0) ? out + add_to : 0  — the bwd-data epilogue of the training step (ConvArgs::mask / add_to) as a pass of
// its own behind the row kernels above (the tensors of these layers are 4096 voxels per cube)
__global__ void mask_add_kernel(float* y, const float* mask, const float* add_to, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = y[i];
  if (add_to) v += add_to[i];
  if (mask && !(mask[i] > 0.f)) v = 0.f;
  y[i] = v;
}

// The 8^3 layers of the hyperprior networks for the training step's generic conv arguments (forward, and bwd-data written
// as the adjoint convolution): 1 launched, 0 not one of these layers (the caller keeps its generic path), < 0 error.
int launch_hyper_row_conv(const ConvArgs& a, hipStream_t s) {
  if (a.ksize != 3 || a.w2 || a.y2 || a.res || a.absval || a.x_co || a.y_co || a.x_cs != a.Cin || a.y_cs != a.Cout) return 0;
  // the epilogue below runs AFTER the row kernel has written y: with add_to aliasing y (ConvArgs allows it, and the
  // trainer's bwd-data passes dx == add_to) the accumulated gradient would be overwritten before it is added — such a
  // call keeps the generic kernels, which apply add_to / mask in their own store
  if (a.add_to && a.add_to == a.y) return 0;
  int rc = 0;
  if (a.mode == 0 && a.Din == kH) rc = launch_conv8_row(a.x, a.y, a.w, a.bias, a.B, a.Cin, a.Cout, a.relu, s);
  else if (a.mode == 1 && a.Din == 2 * kH && a.Cin == 16 && a.Cout == 16) rc = launch_down8_row(a.x, a.y, a.w, a.bias, a.B, a.relu, s) ? -1 : 1;
  else if (a.mode == 2 && a.Din == kH && a.Cin == 16 && a.Cout == 16) rc = launch_up8_row(a.x, a.y, a.w, a.bias, a.B, a.relu, s) ? -1 : 1;
  if (rc != 1) return rc;
  if (a.mask || a.add_to) {
    const int64_t n = (int64_t)a.B * a.Dout * a.Dout * a.Dout * a.Cout;
    hipLaunchKernelGGL(mask_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.y, a.mask, a.add_to, n);
    if (launch_ok("mask_add_kernel")) return -1;
  }
  return 1;
}

}  // namespace pcgc
