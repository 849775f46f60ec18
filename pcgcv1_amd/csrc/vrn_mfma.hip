// Tap-split row-packed MFMA convolutions for the Voxception-ResNet blocks
// (_VoxceptionResNet.call, models/model_voxception.py:56-68) at small channel counts.
//
// The row-packed 3x3x3 layers (conv_mfma.hip header) have an effective filter of
// (QD+2) x (QH+2) x 3 taps with QD+2 == 4 or QH+2 == 4.  Here each of the 4 waves
// of a workgroup takes ONE slice of that 4-deep axis and runs it over ALL patches
// of the workgroup's tile, so a wave fetches a quarter of the packed weights and
// reuses every weight fragment across NP patches (independent accumulators keep the
// MFMA pipe issuing back to back).  The four partial sums per patch are then added
// through LDS in the fixed order wave 0+1+2+3 — deterministic, no atomics.
//
// Optional fusions, selected per launch:
//   fuse 1  the block's conv2_1 (1x1x1, C -> C/4, ReLU) is evaluated on the SAME
//           staged input tile (centre voxels) and written to a second tensor;
//   fuse 2  the result is conv2_2's; bias + ReLU, then conv2_3 (1x1x1, C/4 -> C/2,
//           ReLU) on the lane-resident channels with VALU FMAs, then the residual
//           add + ReLU of model_voxception.py:65-67, written into channels
//           [y_co, y_co + C/2) of the block output.
#include "mfma_common.h"

namespace pcgc {

template <int CIN, int COUTP, int QD, int QH, int GD, int GH, int FUSE>
__global__ void __launch_bounds__(256) conv_ks_kernel(ConvArgs a) {
  using C = Chunk<CIN>;
  constexpr int CK = C::CK, NCH = C::NCH, VEC = C::VEC, VS = C::VS;
  constexpr int KD = QD + 2, KH = QH + 2, KW = 3;
  constexpr bool SPLIT_D = (KD == 4);
  static_assert(KD == 4 || KH == 4, "one filter axis must be 4 deep (one slice per wave)");
  static_assert(QD * QH * COUTP == 16, "patch rows must fill exactly one M tile");
  constexpr int NP = GD * GH;
  static_assert(NP % 4 == 0, "patches per workgroup must be a multiple of 4");
  constexpr int ID = (GD - 1) * QD + KD, IH = (GH - 1) * QH + KH, IW = 18;
  constexpr int NVOX = ID * IH * IW;
  constexpr int TAPS = KD * KH * KW;
  constexpr int TD = GD * QD, TH = GH * QH;
  constexpr int NR = TD * TH / 4;                              // output rows per wave (fuse 1)
  constexpr int TILE_FLOATS = NVOX * VS;
  constexpr int RED_FLOATS = 4 * NP * 256;
  constexpr int LDS_FLOATS = TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int tw = a.Dout / 16, th = a.Dout / TH, td = a.Dout / TD;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  const int b = bid;
  const int od0 = tx * TD, oh0 = ty * TH, ow0 = tz * 16;
  const int id0 = od0 - 1, ih0 = oh0 - 1, iw0 = ow0 - 1;

  f32x4 acc[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc2[FUSE == 1 ? NR : 1];
#pragma unroll
  for (int i = 0; i < (FUSE == 1 ? NR : 1); ++i) acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* wl = a.w + (size_t)lane * VEC;

  for (int cb = 0; cb < NCH; ++cb) {
    if (cb) __syncthreads();
    stage_tile<ID, IH, IW, CK / 4, VS>(lds, a.x + (int64_t)b * a.Din * a.Din * a.Din * a.x_cs + a.x_co + cb * CK, a.Din,
                                       a.x_cs, id0, ih0, iw0);
    __syncthreads();
    const float* wc = wl + (size_t)cb * TAPS * 64 * VEC;
    constexpr int K2 = SPLIT_D ? KH : KD;
#pragma unroll
    for (int k2 = 0; k2 < K2; ++k2) {
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const int kd = SPLIT_D ? wv : k2, kh = SPLIT_D ? k2 : wv;
        const int tap = (kd * KH + kh) * KW + kw;
        float av[4];
        read_vec<VEC>(wc + (size_t)tap * 64 * VEC, av);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int pd = p / GH, ph = p % GH;
          const int pos = ((pd * QD + kd) * IH + (ph * QH + kh)) * IW + (j + kw);
          float bv[4];
          read_vec<VEC>(&lds[pos * VS + VEC * g], bv);
#pragma unroll
          for (int r = 0; r < VEC; ++r) acc[p] = mfma4(av[r], bv[r], acc[p]);
        }
      }
    }
    if constexpr (FUSE == 1) {
      // conv2_1: 1x1x1 on the centre voxels of this wave's share of output rows
      float a2[4];
      read_vec<VEC>(a.w2 + (size_t)cb * 64 * VEC + (size_t)lane * VEC, a2);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int row = wv * NR + i;
        const int rd = row / TH, rh = row % TH;
        const int pos = ((rd + 1) * IH + (rh + 1)) * IW + (j + 1);
        float bv[4];
        read_vec<VEC>(&lds[pos * VS + VEC * g], bv);
#pragma unroll
        for (int r = 0; r < VEC; ++r) acc2[i] = mfma4(a2[r], bv[r], acc2[i]);
      }
    }
  }

  if constexpr (FUSE == 1) {
    // tensor2_1 = relu(conv2_1 + bias): rows = channels; lane group g holds channels 4g..4g+3
    if (4 * g < a.cout2) {
      const float4 bv = *reinterpret_cast<const float4*>(a.bias2 + 4 * g);
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int row = wv * NR + i;
        const int rd = row / TH, rh = row % TH;
        const int64_t vox = (((int64_t)b * a.Dout + od0 + rd) * a.Dout + oh0 + rh) * a.Dout + ow0 + j;
        *reinterpret_cast<float4*>(a.y2 + vox * a.y2_cs + 4 * g) =
            make_float4(fmaxf(acc2[i][0] + bv.x, 0.f), fmaxf(acc2[i][1] + bv.y, 0.f), fmaxf(acc2[i][2] + bv.z, 0.f),
                        fmaxf(acc2[i][3] + bv.w, 0.f));
      }
    }
  }

  // ---- cross-wave reduction of the four filter slices (fixed order) ----
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NP; ++p)
    *reinterpret_cast<float4*>(&lds[((wv * NP + p) * 64 + lane) * 4]) = make_float4(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
  __syncthreads();
#pragma unroll
  for (int pi = 0; pi < NP / 4; ++pi) {
    const int p = pi * 4 + wv;
    float4 s0 = *reinterpret_cast<const float4*>(&lds[((0 * NP + p) * 64 + lane) * 4]);
    const float4 s1 = *reinterpret_cast<const float4*>(&lds[((1 * NP + p) * 64 + lane) * 4]);
    const float4 s2 = *reinterpret_cast<const float4*>(&lds[((2 * NP + p) * 64 + lane) * 4]);
    const float4 s3 = *reinterpret_cast<const float4*>(&lds[((3 * NP + p) * 64 + lane) * 4]);
    s0.x = ((s0.x + s1.x) + s2.x) + s3.x; s0.y = ((s0.y + s1.y) + s2.y) + s3.y;
    s0.z = ((s0.z + s1.z) + s2.z) + s3.z; s0.w = ((s0.w + s1.w) + s2.w) + s3.w;
    const int pd = p / GH, ph = p % GH;
    const int rho = 4 * g;
    const int q = rho / COUTP, c0 = rho % COUTP;
    const int od = od0 + pd * QD + q / QH, oh = oh0 + ph * QH + q % QH;
    const int64_t vox = (((int64_t)b * a.Dout + od) * a.Dout + oh) * a.Dout + ow0 + j;
    if constexpr (FUSE != 2) {
      store_acc(a, vox, c0, f32x4{s0.x, s0.y, s0.z, s0.w});
    } else {
      // t = relu(conv2_2 + bias)  (this lane: channels c0..c0+3 of one voxel)
      const float4 b22 = *reinterpret_cast<const float4*>(a.bias + c0);
      float t[COUTP];
      float own[4] = {fmaxf(s0.x + b22.x, 0.f), fmaxf(s0.y + b22.y, 0.f), fmaxf(s0.z + b22.z, 0.f), fmaxf(s0.w + b22.w, 0.f)};
      int o0 = 0;                                  // first output channel this lane produces
      if constexpr (COUTP == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = own[r];
      } else {
        // COUTP == 8: the voxel's other 4 channels live in the neighbouring lane group (lane ^ 16)
        float oth[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) oth[r] = __shfl_xor(own[r], 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) { t[r] = c0 ? oth[r] : own[r]; t[4 + r] = c0 ? own[r] : oth[r]; }
        o0 = c0 * 2;                               // lanes with c0 = 0 -> outputs 0..7, c0 = 4 -> 8..15
      }
      // conv2_3 (C/4 -> C/2) restricted to 8 outputs per lane, bias, ReLU, residual add, ReLU
      constexpr int CO2 = COUTP * 2;
      float u[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < COUTP; ++c) s = fmaf(t[c], a.w2[c * CO2 + o0 + o], s);
        u[o] = fmaxf(s + a.bias2[o0 + o], 0.f);
      }
      float* yp = a.y + vox * a.y_cs + a.y_co + o0;
      const float* rp = a.res + vox * a.y_cs + a.y_co + o0;
      const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
      *reinterpret_cast<float4*>(yp) = make_float4(fmaxf(r0.x + u[0], 0.f), fmaxf(r0.y + u[1], 0.f), fmaxf(r0.z + u[2], 0.f),
                                                   fmaxf(r0.w + u[3], 0.f));
      *reinterpret_cast<float4*>(yp + 4) = make_float4(fmaxf(r1.x + u[4], 0.f), fmaxf(r1.y + u[5], 0.f),
                                                       fmaxf(r1.z + u[6], 0.f), fmaxf(r1.w + u[7], 0.f));
    }
  }
}

template <int CIN, int COUTP, int QD, int QH, int GD, int GH, int FUSE>
static int run_ks(const ConvArgs& a, hipStream_t s) {
  constexpr int TD = GD * QD, TH = GH * QH;
  if (a.Dout % TD || a.Dout % TH || a.Dout % 16) return 0;
  const int blocks = a.B * (a.Dout / TD) * (a.Dout / TH) * (a.Dout / 16);
  hipLaunchKernelGGL((conv_ks_kernel<CIN, COUTP, QD, QH, GD, GH, FUSE>), dim3(blocks), dim3(256), 0, s, a);
  int rc = launch_ok("conv_ks_kernel");
  return rc ? rc : 1;
}

int launch_conv_ks(const ConvArgs& a, const float* packed_w, int fuse, hipStream_t s, bool run) {
  if (a.mode != 0 || a.ksize != 3 || a.Dout % 16) return 0;
  if (a.x_cs % 4 || a.x_co % 4 || a.y_cs % 4 || a.y_co % 4 || a.Cout % 4) return 0;
  if (a.x_q4 || a.y_q4 || a.mask || a.add_to) return 0;
  ConvArgs b = a;
  b.w = packed_w;
#define TRY(cond, call)                        \
  if (cond) {                                  \
    if (!run) return 1;                        \
    return call;                               \
  }
  if (fuse == 1) {
    if (a.absval || a.res || !a.relu || !a.bias || !a.bias2) return 0;
    TRY(a.Cin == 16 && a.Cout == 4 && a.cout2 == 4, (run_ks<16, 4, 2, 2, 2, 2, 1>(b, s)))
    TRY(a.Cin == 32 && a.Cout == 8 && a.cout2 == 8, (run_ks<32, 8, 1, 2, 4, 2, 1>(b, s)))
    return 0;
  }
  if (fuse == 2) {
    if (a.absval || !a.res || !a.relu || !a.bias || !a.bias2) return 0;
    TRY(a.Cin == 4 && a.Cout == 4 && a.cout2 == 8, (run_ks<4, 4, 2, 2, 4, 4, 2>(b, s)))
    TRY(a.Cin == 8 && a.Cout == 8 && a.cout2 == 16, (run_ks<8, 8, 1, 2, 4, 4, 2>(b, s)))
    return 0;
  }
  TRY(a.Cin == 16 && a.Cout == 4, (run_ks<16, 4, 2, 2, 2, 2, 0>(b, s)))
  TRY(a.Cin == 4 && a.Cout == 4, (run_ks<4, 4, 2, 2, 4, 4, 0>(b, s)))
  TRY(a.Cin == 4 && a.Cout == 8, (run_ks<4, 8, 1, 2, 4, 4, 0>(b, s)))
  TRY(a.Cin == 8 && a.Cout == 8, (run_ks<8, 8, 1, 2, 4, 4, 0>(b, s)))
  TRY(a.Cin == 32 && a.Cout == 8, (run_ks<32, 8, 1, 2, 4, 2, 0>(b, s)))
#undef TRY
  return 0;
}

}  // namespace pcgc
