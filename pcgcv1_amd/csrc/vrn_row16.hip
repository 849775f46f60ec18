// Voxception-ResNet block for C = 64 at 16^3 (the low-resolution stage of both transforms, models/model_voxception.py:
// 56-68) on the v_mfma_f32_4x4x1_16B_f32 row scheme of vrn_row.hip with FOUR cube rows per 64-lane vector:
//   lane = (row & 3, w):  lanes 16r..16r+15 = row a + r  ("quad vector" starting at row a).
// Tensors are Q4 [b][d][h][C/4][w][4]: a quad vector of one channel quad is one dwordx4 buffer load per lane (four 256-B
// segments).  The three kh taps of an output vector (rows 4k..4k+3) are the input vectors starting at rows 4k-1, 4k,
// 4k+1 — three loads (rows come from L1 / L2 the second and third time); kw = 0 / 2 are DPP row_shr:1 / row_shl:1, which
// shift inside each 16-lane row and fill with zero: exactly the 'same' padding, no lane needs fixing.
// Weights are packed one chunk per input-channel quad, 64 consecutive floats = one A-operand VGPR = 16 (tap, ci,
// cout-quad) blocks, staged once per workgroup in LDS: 112 KB for kernel A (512-thread workgroups, one per CU), 54 / 27 KB
// for kernels B / C.
//   kernel A: t12 = [ relu(conv1_1(x)) 3^3 64->16 | relu(conv2_1(x)) 1^3 64->16 ]
//   kernel B: out[0:32]  = relu(x[0:32]  + relu(conv1_2(t11)))                          3^3 16->32
//   kernel C: out[32:64] = relu(x[32:64] + relu(conv2_3(relu(conv2_2(t21)))))           3^3 16->16, 1^3 16->32
// (B and C are separate launches: together their accumulators, 12 registers x 4 per output vector and plane slot, and
// the 16 residual quads do not fit one wave.)  Summation order: bias, then (plane, channel, kh, kw): fixed.
#include "row_common.h"

namespace pcgc {

constexpr int kW16 = 16;                  // cube edge of this stage
constexpr int kRowQ16 = kW16 * 16;        // bytes of one (row, channel quad)

__device__ __forceinline__ float rshr1(float v) {   // lane i <- lane i-1 inside each 16-lane row, first lane <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float rshl1(float v) {   // lane i <- lane i+1 inside each 16-lane row, last lane <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

// quad vector (rows a .. a+3) of plane p, channel quad q of a Q4 tensor with NQ quads; rows / planes outside read 0
template <int NQ>
__device__ __forceinline__ f32x4 load_vec(i32x4 rs, int lane_off, int lane_row, int p, int q, int a) {
  const bool ok = (unsigned)p < (unsigned)kW16 && (unsigned)(a + lane_row) < (unsigned)kW16;
  const int base = ((p * kW16 + a) * NQ + q) * kRowQ16;
  return raw_load4(rs, ok ? base + lane_off : kOOB, 0, 0);
}

struct Tile16 {
  int b, k, d0;
};
template <int LD>
__device__ __forceinline__ Tile16 wave_tile16() {
  int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 8 + (threadIdx.x >> 6));
  Tile16 t;
  t.k = wv % 4; wv /= 4;
  t.d0 = (wv % (kW16 / LD)) * LD; wv /= (kW16 / LD);
  t.b = wv;
  return t;
}

// LDS images of a block's filters, built once per net (vrn64_image_kernel): [A: 16 chunks of 1792][B: 8 chunks of 1728]
// [C: 4 chunks of 1728] floats, exactly what the kernels' staging loops gather from the TensorFlow layouts — copied with
// 16-byte loads that are all in flight at once instead of ~56 dependent gathers per thread with their index arithmetic
// (a workgroup's staging prologue was ~10 % of vrn64a's time: 1 workgroup per CU, two rounds per 103-cube launch)
constexpr int kVrn64ImgA = 16 * 28 * 64, kVrn64ImgB = 8 * 27 * 64, kVrn64ImgC = 4 * 27 * 64;
constexpr int kVrn64ImageFloats = kVrn64ImgA + kVrn64ImgB + kVrn64ImgC;

struct Vrn64Args {
  const float* img = nullptr;    // the block's image (nullptr: gather from the TensorFlow layouts in the kernel)
  const float* x;      // block input,  Q4 [B][16][16][16][16][4]
  float* t12;          // scratch,      Q4 [B][16][16][8][16][4]: quads 0-3 = tensor1_1, quads 4-7 = tensor2_1
  float* out;          // block output, Q4 like x (may alias x)
  const float *w11, *b11, *w21, *b21, *w12, *b12, *w22, *b22, *w23, *b23;   // TensorFlow layouts
  int B;
};

// one input channel: the three kh-aligned vectors X[kh] -> 27 taps into NCO output-channel quads.
// chunk layout [tap][ci4][16 couts]: weight register = tap, abid = c*4 + coq
template <int NCO>
__device__ __forceinline__ void vec_channel(f32x4 (&acc)[3][NCO], const float (&W)[28], int c, const f32x4 (&X)[3], bool v0, bool v1,
                                            bool v2) {
  float x0[3], xm[3], xp[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) { x0[kh] = comp(X[kh], c); xm[kh] = rshr1(x0[kh]); xp[kh] = rshl1(x0[kh]); }
  const bool vj[3] = {v0, v1, v2};
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const int kd = 2 - jj;
    if (vj[jj]) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int t = (kd * 3 + kh) * 3 + kw;
          const float xv = kw == 0 ? xm[kh] : (kw == 1 ? x0[kh] : xp[kh]);
#pragma unroll
          for (int coq = 0; coq < NCO; ++coq) acc[jj][coq] = mfa(c * 4 + coq, W[t], xv, acc[jj][coq]);
        }
    }
  }
}

// vec_channel for the four channels of a quad with the validity tests hoisted: one wave-uniform branch per (quad, output
// plane) instead of one per (channel, output plane) (vrn_row.hip: a_quad).  W2 >= 0: kernel A's conv2_1 (1^3, weight
// register W2) on the centre vector rides in the jj = 1 block.  Same order of contributions per accumulator.
template <int NCO, int W2 = -1>
__device__ __forceinline__ void vec_quad(f32x4 (&acc)[3][NCO], f32x4* acc2, const float (&W)[28], const f32x4 (&X)[3], bool v0, bool v1,
                                         bool v2) {
  float x0[4][3], xm[4][3], xp[4][3];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) { x0[c][kh] = comp(X[kh], c); xm[c][kh] = rshr1(x0[c][kh]); xp[c][kh] = rshl1(x0[c][kh]); }
  const bool vj[3] = {v0, v1, v2};
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const int kd = 2 - jj;
    if (vj[jj]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int t = (kd * 3 + kh) * 3 + kw;
            const float xv = kw == 0 ? xm[c][kh] : (kw == 1 ? x0[c][kh] : xp[c][kh]);
#pragma unroll
            for (int coq = 0; coq < NCO; ++coq) acc[jj][coq] = mfa(c * 4 + coq, W[t], xv, acc[jj][coq]);
          }
        if constexpr (W2 >= 0) {
          if (jj == 1) {
#pragma unroll
            for (int coq = 0; coq < NCO; ++coq) acc2[coq] = mfa(c * 4 + coq, W[W2], x0[c][1], acc2[coq]);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// kernel A
// ---------------------------------------------------------------------------------------------------------------
constexpr int kA64Chunk = 28 * 64;                          // floats per quad chunk: 27*4*16 conv1_1 + 4*16 conv2_1

template <int LD, bool QJ = true>
__global__ void __launch_bounds__(512, 2) vrn64a_row_kernel(Vrn64Args a) {
  constexpr int CH = kA64Chunk;
  __shared__ float wl[16 * CH];                             // 112 KB: every input quad's chunk, staged once per workgroup
  if (a.img) stage_image_t<16 * CH, 512>(wl, a.img);
  else
    for (int i = threadIdx.x; i < 16 * CH; i += 512) {
      const int q = i / CH, f = i - q * CH;
      wl[i] = f < 1728 ? a.w11[((f >> 6) * 64 + 4 * q + ((f >> 4) & 3)) * 16 + (f & 15)] : a.w21[(4 * q + ((f - 1728) >> 4)) * 16 + (f & 15)];
    }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int lane_row = lane >> 4;
  const Tile16 tl = wave_tile16<LD>();
  if (tl.b >= a.B) return;
  const int r0 = 4 * tl.k, d0 = tl.d0;
  f32x4 bi[4], bi2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bi[q] = f32x4{a.b11[4 * q], a.b11[4 * q + 1], a.b11[4 * q + 2], a.b11[4 * q + 3]};
    bi2[q] = f32x4{a.b21[4 * q], a.b21[4 * q + 1], a.b21[4 * q + 2], a.b21[4 * q + 3]};
  }
  f32x4 acc[3][4], acc2[4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[j][q] = bi[q];
  const i32x4 rs = make_rsrc(a.x + (size_t)tl.b * kW16 * kW16 * kW16 * 64, kW16 * kW16 * kW16 * 64 * 4);
  const int lane_off = lane_row * (16 * kRowQ16) + (lane & 15) * 16;                 // x has 16 quads per row
  f32x4* tb = reinterpret_cast<f32x4*>(a.t12) + (size_t)tl.b * kW16 * kW16 * 8 * kW16 + lane_row * (8 * kW16) + (lane & 15);
  f32x4 XA[3], XB[3];
  auto load = [&](f32x4 (&X)[3], int p, int q) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) X[kh] = load_vec<16>(rs, lane_off, lane_row, p, q, r0 + kh - 1);
  };
  // a quad's 28 weight registers are fetched from LDS one quad AHEAD of their MFMAs (WA / WB alternate like XA / XB): fetched
  // right before them, every quad step began with an exposed LDS round trip (SQ_WAIT_ANY 16 % of the wave cycles)
  auto fetch = [&](float (&W)[28], int q) {
#pragma unroll
    for (int v = 0; v < 28; ++v) W[v] = wl[q * CH + v * 64 + lane];
  };
  auto quad = [&](const f32x4 (&X)[3], const float (&W)[28], bool v0, bool v1, bool v2) {
    if constexpr (QJ) { vec_quad<4, 27>(acc, acc2, W, X, v0, v1, v2); return; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      vec_channel<4>(acc, W, c, X, v0, v1, v2);
      if (v1) {                                             // conv2_1 on the centre voxel: register 27
#pragma unroll
        for (int coq = 0; coq < 4; ++coq) acc2[coq] = mfa(c * 4 + coq, W[27], comp(X[1], c), acc2[coq]);
      }
    }
  };
  float WA[28], WB[28];
  load(XA, d0 - 1, 0);
  fetch(WA, 0);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kW16;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc2[q] = bi2[q];
#pragma unroll 1
    for (int q = 0; q < 16; q += 2) {
      load(XB, p, q + 1);
      fetch(WB, q + 1);
      quad(XA, WA, v0, v1, v2);
      if (q + 2 < 16) load(XA, p, q + 2); else load(XA, p + 1, 0);
      fetch(WA, (q + 2) & 15);
      quad(XB, WB, v0, v1, v2);
    }
    if (v1) {
#pragma unroll
      for (int coq = 0; coq < 4; ++coq) tb[((size_t)(p * kW16 + r0) * 8 + 4 + coq) * kW16] = relu4(acc2[coq]);
    }
    if (p - 1 >= d0) {
#pragma unroll
      for (int coq = 0; coq < 4; ++coq) tb[((size_t)((p - 1) * kW16 + r0) * 8 + coq) * kW16] = relu4(acc[0][coq]);
    }
#pragma unroll
    for (int coq = 0; coq < 4; ++coq) { acc[0][coq] = acc[1][coq]; acc[1][coq] = acc[2][coq]; acc[2][coq] = bi[coq]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// kernel B: conv1_2 (16 -> 32 as two halves of 16 output channels) + residual on channels 0..31
// ---------------------------------------------------------------------------------------------------------------
template <int LD, bool QJ = true>
__global__ void __launch_bounds__(512, 2) vrn64b_row_kernel(Vrn64Args a) {
  constexpr int CH = 27 * 64;                               // floats per (quad, half) chunk [tap][ci4][16]
  __shared__ float wl[8 * CH];                              // 54 KB
  if (a.img) stage_image_t<8 * CH, 512>(wl, a.img + kVrn64ImgA);
  else
    for (int i = threadIdx.x; i < 8 * CH; i += 512) {
      const int qh = i / CH, f = i - qh * CH, q = qh >> 1, half = qh & 1;
      wl[i] = a.w12[((f >> 6) * 16 + 4 * q + ((f >> 4) & 3)) * 32 + 16 * half + (f & 15)];
    }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int lane_row = lane >> 4;
  const Tile16 tl = wave_tile16<LD>();
  if (tl.b >= a.B) return;
  const int r0 = 4 * tl.k, d0 = tl.d0;
  f32x4 bi[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bi[q] = f32x4{a.b12[4 * q], a.b12[4 * q + 1], a.b12[4 * q + 2], a.b12[4 * q + 3]};
  f32x4 accL[3][4], accH[3][4];                             // output channels 0..15 / 16..31
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) { accL[j][q] = bi[q]; accH[j][q] = bi[4 + q]; }
  const i32x4 rs = make_rsrc(a.t12 + (size_t)tl.b * kW16 * kW16 * kW16 * 32, kW16 * kW16 * kW16 * 32 * 4);
  const i32x4 rx = make_rsrc(a.x + (size_t)tl.b * kW16 * kW16 * kW16 * 64, kW16 * kW16 * kW16 * 64 * 4);
  const i32x4 ro = make_rsrc(a.out + (size_t)tl.b * kW16 * kW16 * kW16 * 64, kW16 * kW16 * kW16 * 64 * 4);
  const int lane_off = lane_row * (8 * kRowQ16) + (lane & 15) * 16;                  // t12: 8 quads per row
  const int lane_off_x = lane_row * (16 * kRowQ16) + (lane & 15) * 16;               // x / out: 16 quads per row
  f32x4 XA[3], XB[3];
  auto load = [&](f32x4 (&X)[3], int p, int q) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) X[kh] = load_vec<8>(rs, lane_off, lane_row, p, q, r0 + kh - 1);
  };
  auto quad = [&](const f32x4 (&X)[3], int q, bool v0, bool v1, bool v2) {
    float W[28];
#pragma unroll
    for (int v = 0; v < 27; ++v) W[v] = wl[(2 * q) * CH + v * 64 + lane];
    if constexpr (QJ) vec_quad<4>(accL, nullptr, W, X, v0, v1, v2);
    else {
#pragma unroll
      for (int c = 0; c < 4; ++c) vec_channel<4>(accL, W, c, X, v0, v1, v2);
    }
#pragma unroll
    for (int v = 0; v < 27; ++v) W[v] = wl[(2 * q + 1) * CH + v * 64 + lane];
    if constexpr (QJ) vec_quad<4>(accH, nullptr, W, X, v0, v1, v2);
    else {
#pragma unroll
      for (int c = 0; c < 4; ++c) vec_channel<4>(accH, W, c, X, v0, v1, v2);
    }
  };
  load(XA, d0 - 1, 0);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kW16;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
    load(XB, p, 1);
    quad(XA, 0, v0, v1, v2);
    load(XA, p, 2);
    const int obase = p - 1 >= d0 ? ((p - 1) * kW16 + r0) * (16 * kRowQ16) + lane_off_x : kOOB;
    f32x4 res[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) res[q] = raw_load4(rx, obase + q * kRowQ16, 0, 0);
    quad(XB, 1, v0, v1, v2);
    load(XB, p, 3);
    quad(XA, 2, v0, v1, v2);
    load(XA, p + 1, 0);
    quad(XB, 3, v0, v1, v2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      raw_store4(relu4(res[q] + relu4(accL[0][q])), ro, obase + q * kRowQ16, 0, 0);
      raw_store4(relu4(res[4 + q] + relu4(accH[0][q])), ro, obase + (4 + q) * kRowQ16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      accL[0][q] = accL[1][q]; accL[1][q] = accL[2][q]; accL[2][q] = bi[q];
      accH[0][q] = accH[1][q]; accH[1][q] = accH[2][q]; accH[2][q] = bi[4 + q];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// kernel C: conv2_2 (16 -> 16) -> ReLU -> conv2_3 (1^3, 16 -> 32) + residual on channels 32..63
// ---------------------------------------------------------------------------------------------------------------
template <int LD, bool QJ = false>     // per-channel tests here: the hoisted form measured 3 % slower (88 against 85 us per 103 cubes)
__global__ void __launch_bounds__(512, 2) vrn64c_row_kernel(Vrn64Args a) {
  constexpr int CH = 27 * 64;
  __shared__ float wl[4 * CH];                              // 27 KB
  if (a.img) stage_image_t<4 * CH, 512>(wl, a.img + kVrn64ImgA + kVrn64ImgB);
  else
    for (int i = threadIdx.x; i < 4 * CH; i += 512) {
      const int q = i / CH, f = i - q * CH;
      wl[i] = a.w22[((f >> 6) * 16 + 4 * q + ((f >> 4) & 3)) * 16 + (f & 15)];
    }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int lane_row = lane >> 4;
  const Tile16 tl = wave_tile16<LD>();
  if (tl.b >= a.B) return;
  const int r0 = 4 * tl.k, d0 = tl.d0;
  float W23[8];                                             // [16][32]: register ci>>1, abid (ci&1)*8 + coq
#pragma unroll
  for (int v = 0; v < 8; ++v) W23[v] = a.w23[v * 64 + lane];
  f32x4 bi22[4], bi23[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) bi22[q] = f32x4{a.b22[4 * q], a.b22[4 * q + 1], a.b22[4 * q + 2], a.b22[4 * q + 3]};
#pragma unroll
  for (int q = 0; q < 8; ++q) bi23[q] = f32x4{a.b23[4 * q], a.b23[4 * q + 1], a.b23[4 * q + 2], a.b23[4 * q + 3]};
  f32x4 acc[3][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[j][q] = bi22[q];
  const i32x4 rs = make_rsrc(a.t12 + (size_t)tl.b * kW16 * kW16 * kW16 * 32, kW16 * kW16 * kW16 * 32 * 4);
  const i32x4 rx = make_rsrc(a.x + (size_t)tl.b * kW16 * kW16 * kW16 * 64, kW16 * kW16 * kW16 * 64 * 4);
  const i32x4 ro = make_rsrc(a.out + (size_t)tl.b * kW16 * kW16 * kW16 * 64, kW16 * kW16 * kW16 * 64 * 4);
  const int lane_off = lane_row * (8 * kRowQ16) + (lane & 15) * 16;
  const int lane_off_x = lane_row * (16 * kRowQ16) + (lane & 15) * 16;
  f32x4 XA[3], XB[3];
  auto load = [&](f32x4 (&X)[3], int p, int q) {            // tensor2_1 = quads 4..7 of t12
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) X[kh] = load_vec<8>(rs, lane_off, lane_row, p, 4 + q, r0 + kh - 1);
  };
  auto quad = [&](const f32x4 (&X)[3], int q, bool v0, bool v1, bool v2) {
    float W[28];
#pragma unroll
    for (int v = 0; v < 27; ++v) W[v] = wl[q * CH + v * 64 + lane];
    if constexpr (QJ) vec_quad<4>(acc, nullptr, W, X, v0, v1, v2);
    else {
#pragma unroll
      for (int c = 0; c < 4; ++c) vec_channel<4>(acc, W, c, X, v0, v1, v2);
    }
  };
  load(XA, d0 - 1, 0);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kW16;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
    load(XB, p, 1);
    quad(XA, 0, v0, v1, v2);
    load(XA, p, 2);
    const int obase = p - 1 >= d0 ? ((p - 1) * kW16 + r0) * (16 * kRowQ16) + 8 * kRowQ16 + lane_off_x : kOOB;   // quads 8..15
    f32x4 res[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) res[q] = raw_load4(rx, obase + q * kRowQ16, 0, 0);
    quad(XB, 1, v0, v1, v2);
    load(XB, p, 3);
    quad(XA, 2, v0, v1, v2);
    load(XA, p + 1, 0);
    quad(XB, 3, v0, v1, v2);
    // output plane p-1: conv2_3 on relu(conv2_2), residual, ReLU, store
    f32x4 t22[4], q3[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) t22[q] = relu4(acc[0][q]);
#pragma unroll
    for (int q = 0; q < 8; ++q) q3[q] = bi23[q];
#pragma unroll
    for (int ci = 0; ci < 16; ++ci)
#pragma unroll
      for (int coq = 0; coq < 8; ++coq) q3[coq] = mfa((ci & 1) * 8 + coq, W23[ci >> 1], comp(t22[ci >> 2], ci & 3), q3[coq]);
#pragma unroll
    for (int q = 0; q < 8; ++q) raw_store4(relu4(res[q] + relu4(q3[q])), ro, obase + q * kRowQ16, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[0][q] = acc[1][q]; acc[1][q] = acc[2][q]; acc[2][q] = bi22[q]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// up_1: stride-2 transposed conv 3^3, 64 -> 32 channels, 16^3 -> 32^3 (models/model_voxception.py:160-165), + ReLU:
// up2_row_kernel (vrn_row32.hip) on quad vectors.  y[o] = bias + sum_{o = 2i + k} x[i] W[k] per axis: lane = input voxel
// (row 4k + r, voxel i), owning outputs (2ih + ph, 2i + pw); x[i-1] is row_shr:1 (zero at the row start by itself),
// x[ih-1] the quad vector starting one row higher (loaded); input plane p completes output plane 2p (kd 0 + carried
// kd 2), makes 2p+1 (kd 1) and opens 2p+2 (kd 2).  A workgroup works on one group of 8 output channels (2 quads): the
// group's filter image, [channel quad][tap][ci4][8 couts] = 55 KB, is copied into LDS (a.w = image of all four groups,
// up1_image_kernel).  x Q4 [B][16][16][16][16][4], y Q4 [B][32][32][8][32][4].
// ---------------------------------------------------------------------------------------------------------------
struct Up1Args {
  const float* x;
  float* y;
  const float* w;
  const float* bias;
  int B, relu;
};
constexpr int kUp1Group = 16 * 27 * 32;                     // floats of one cout group's image

template <int LD>
__global__ void __launch_bounds__(256, 2) up1_row_kernel(Up1Args a) {
  constexpr int NCO = 2, CHT = 16 * NCO, CH = 27 * CHT, NW = (CH + 63) / 64;
  __shared__ __attribute__((aligned(16))) float wl[kUp1Group + 64];
  int wg = blockIdx.x;
  const int g = wg & 3; wg >>= 2;                            // cout group of this workgroup
  stage_image<kUp1Group>(wl, a.w + (size_t)g * kUp1Group);
  if (threadIdx.x < 64) wl[kUp1Group + threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, lane_row = lane >> 4;
  const int k = threadIdx.x >> 6;                             // the workgroup's four waves = the four row quads of a plane
  const int d0 = (wg % (kW16 / LD)) * LD;
  const int b = wg / (kW16 / LD);
  if (b >= a.B) return;
  f32x4 bi[NCO];
#pragma unroll
  for (int c = 0; c < NCO; ++c) {
    bi[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bi[c] = f32x4{a.bias[(g * NCO + c) * 4], a.bias[(g * NCO + c) * 4 + 1], a.bias[(g * NCO + c) * 4 + 2], a.bias[(g * NCO + c) * 4 + 3]};
  }
  f32x4 acc[3][2][2][NCO];                                   // [set][oh parity][ow parity][cout quad], sets as in up2_row_kernel
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw)
#pragma unroll
        for (int c = 0; c < NCO; ++c) acc[s_][ph][pw][c] = bi[c];
  const i32x4 rs = make_rsrc(a.x + (size_t)b * kW16 * kW16 * kW16 * 64, kW16 * kW16 * kW16 * 64 * 4);
  const i32x4 ro = make_rsrc(a.y + (size_t)b * 32 * 32 * 32 * 32, 32 * 32 * 32 * 32 * 4);
  const int lane_off = lane_row * (16 * kRowQ16) + (lane & 15) * 16;                 // x: 16 quads per row
  // output row 8k + 2r (+ parity), 8 quads of 32 x 16 B per row; this lane's voxel pair starts at ow = 2i
  const int out_lane = ((8 * k + 2 * lane_row) * 8 + g * NCO) * 512 + (lane & 15) * 32;
  f32x4 PA, OA, PB, OB;
  auto load = [&](f32x4& P, f32x4& O, int p, int q) {
    P = load_vec<16>(rs, lane_off, lane_row, p, q, 4 * k);       // rows ih
    O = load_vec<16>(rs, lane_off, lane_row, p, q, 4 * k - 1);   // rows ih - 1
  };
  auto quad = [&](const f32x4& P, const f32x4& O, int q, bool v0, bool v2) {
    float W[NW];
#pragma unroll
    for (int v = 0; v < NW; ++v) W[v] = wl[q * CH + v * 64 + lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float x0 = comp(P, c), x1 = comp(O, c);
      const float r0 = rshr1(x0), r1 = rshr1(x1);
      const bool vj[3] = {v0, v0, v2};
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        if (vj[s_]) {
#pragma unroll
          for (int co = 0; co < NCO; ++co) {
            auto mf_ = [&](int kh, int kw, float xv, f32x4& d) {
              const int t = (s_ * 3 + kh) * 3 + kw, fo = t * CHT + c * 4 * NCO + co * 4;
              d = mfa((fo & 63) >> 2, W[fo >> 6], xv, d);
            };
            mf_(0, 0, x0, acc[s_][0][0][co]); mf_(0, 2, r0, acc[s_][0][0][co]); mf_(2, 0, x1, acc[s_][0][0][co]); mf_(2, 2, r1, acc[s_][0][0][co]);
            mf_(0, 1, x0, acc[s_][0][1][co]); mf_(2, 1, x1, acc[s_][0][1][co]);
            mf_(1, 0, x0, acc[s_][1][0][co]); mf_(1, 2, r0, acc[s_][1][0][co]);
            mf_(1, 1, x0, acc[s_][1][1][co]);
          }
        }
      }
    }
  };
  auto store_plane = [&](int set, int od, bool ok) {
    const int base = ok ? od * (32 * 8 * 512) + out_lane : kOOB;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int co = 0; co < NCO; ++co)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
          f32x4 v = acc[set][ph][pw][co];
          if (a.relu) v = relu4(v);
          raw_store4(v, ro, base + (ph * 8 + co) * 512 + pw * 16, 0, 0);
        }
  };
  load(PA, OA, d0 - 1, 0);
#pragma unroll 1
  for (int p = d0 - 1; p < d0 + LD; ++p) {
    const bool v0 = p >= d0, v2 = p >= 0 && p + 1 < d0 + LD;
#pragma unroll 1
    for (int q = 0; q < 16; q += 2) {
      load(PB, OB, p, q + 1);
      quad(PA, OA, q, v0, v2);
      if (q + 2 < 16) load(PA, OA, p, q + 2); else load(PA, OA, p + 1, 0);
      quad(PB, OB, q + 1, v0, v2);
    }
    store_plane(0, 2 * p, v0);
    store_plane(1, 2 * p + 1, v0);
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw)
#pragma unroll
        for (int c = 0; c < NCO; ++c) { acc[0][ph][pw][c] = acc[2][ph][pw][c]; acc[1][ph][pw][c] = bi[c]; acc[2][ph][pw][c] = bi[c]; }
  }
}

// image of up_1's filter for up1_row_kernel: [cout group 4][channel quad 16][tap 27][ci4][8 couts] from the
// Conv3DTranspose layout [27][Cout = 32][Cin = 64]
__global__ void __launch_bounds__(256) up1_image_kernel(const float* w, float* dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 4 * kUp1Group) return;
  const int g = i / kUp1Group, f0 = i - g * kUp1Group, q = f0 / 864, f = f0 - q * 864;
  const int tap = f >> 5, c = (f >> 3) & 3, co = f & 7;
  dst[i] = w[(tap * 32 + g * 8 + co) * 64 + 4 * q + c];
}
size_t up1_image_floats() { return 4 * (size_t)kUp1Group; }
int launch_up1_image(const float* w_tf, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(up1_image_kernel, dim3((4 * kUp1Group + 255) / 256), dim3(256), 0, s, w_tf, dst);
  return launch_ok("up1_image_kernel");
}
int launch_up1_row(const float* x, float* y, const float* w_image, const float* bias, int B, int relu, hipStream_t s) {
  Up1Args a{x, y, w_image, bias, B, relu};
  constexpr int LD = 4;                                      // 4 input planes per workgroup: 16 workgroups (64 waves) per cube
  // (2 planes: 427.7 against 433.5 us per 103 cubes, 8 planes: 474 — profiles/r05_vH_tile_by_launch_size.txt)
  hipLaunchKernelGGL((up1_row_kernel<LD>), dim3(B * (kW16 / LD) * 4), dim3(256), 0, s, a);
  return launch_ok("up1_row_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// down_2: stride-2 conv 3^3, 32 -> 64 channels, 32^3 -> 16^3 (models/model_voxception.py:109-114), + ReLU:
// down1_row_kernel (vrn_row32.hip) on quad vectors.  y[o] = bias + sum_k x[2o + k] W[k] per axis (one zero behind).
// Lane = output voxel (row 4k + r, voxel o); the even / odd input voxels of input row 2(4k + r) + kh are two strided
// loads (kw = 0 / 1), kw = 2 is the even vector shifted by a lane (row_shl:1: zero at the row end by itself); output
// plane j reads input planes 2j, 2j+1, 2j+2, and 2j+2 is also plane j+1's kd = 0 (two accumulator sets).  A workgroup
// (512 threads: 4 row quads x 2 plane segments) works on one group of 32 output channels, whose filter image
// [channel quad][tap][ci4][32 couts] = 110 KB sits in LDS (a.w = image of both groups, down2_image_kernel).
// x Q4 [B][32][32][8][32][4], y Q4 [B][16][16][16][16][4].
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDown2Group = 8 * 27 * 128;                   // floats of one cout group's image

template <int LD>
__global__ void __launch_bounds__(512, 2) down2_row_kernel(Up1Args a) {
  constexpr int NCO = 8, CHT = 16 * NCO, CH = 27 * CHT, NWK = 9 * CHT / 64;      // 18 weight registers per kd slice
  __shared__ __attribute__((aligned(16))) float wl[kDown2Group];
  int wg = blockIdx.x;
  const int g = wg & 1; wg >>= 1;
  for (int i = threadIdx.x * 4; i < kDown2Group; i += 2048)
    *reinterpret_cast<float4*>(&wl[i]) = *reinterpret_cast<const float4*>(&a.w[(size_t)g * kDown2Group + i]);
  __syncthreads();
  const int lane = threadIdx.x & 63, lane_row = lane >> 4, wave = threadIdx.x >> 6;
  const int k = wave & 3;                                    // row quad: output rows 4k .. 4k+3
  constexpr int SEGS = kW16 / LD;                            // plane segments per cube, two per workgroup
  const int seg = (wg % (SEGS / 2)) * 2 + (wave >> 2);
  const int b = wg / (SEGS / 2);
  const int d0 = seg * LD;
  if (b >= a.B) return;
  f32x4 bi[NCO];
#pragma unroll
  for (int c = 0; c < NCO; ++c) {
    bi[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bi[c] = f32x4{a.bias[(g * NCO + c) * 4], a.bias[(g * NCO + c) * 4 + 1], a.bias[(g * NCO + c) * 4 + 2], a.bias[(g * NCO + c) * 4 + 3]};
  }
  f32x4 cur[NCO], nxt[NCO];
#pragma unroll
  for (int c = 0; c < NCO; ++c) { cur[c] = bi[c]; nxt[c] = bi[c]; }
  const i32x4 rs = make_rsrc(a.x + (size_t)b * 32 * 32 * 32 * 32, 32 * 32 * 32 * 32 * 4);
  const i32x4 ro = make_rsrc(a.y + (size_t)b * kW16 * kW16 * kW16 * 64, kW16 * kW16 * kW16 * 64 * 4);
  // input: voxel 2o (+1) of row 8k + 2r + kh, 8 quads of 32 x 16 B per row
  const int in_lane = lane_row * (2 * 8 * 512) + (lane & 15) * 32;
  const int out_lane = ((4 * k + lane_row) * 16 + g * NCO) * kRowQ16 + (lane & 15) * 16;
  struct Rows { f32x4 e[3], o[3]; };
  auto load = [&](Rows& R, int p, int q) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = 8 * k + kh;                             // + 2 * lane_row: only that can leave the cube (row 32)
      const bool ok = (unsigned)p < 32u && ih + 2 * lane_row < 32;
      const int off = ok ? ((p * 32 + ih) * 8 + q) * 512 + in_lane : kOOB;
      R.e[kh] = raw_load4(rs, off, 0, 0);
      R.o[kh] = raw_load4(rs, off + 16, 0, 0);
    }
  };
  // one channel quad of one input plane: kd = KA into accA (and kd = KB into accB when KB >= 0)
  auto quad = [&](const Rows& R, int q, int KA, f32x4 (&accA)[NCO], int KB, f32x4 (&accB)[NCO], bool vB) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int kd = pass == 0 ? KA : KB;
      if (kd < 0 || (pass == 1 && !vB)) continue;
      float W[NWK];
#pragma unroll
      for (int vv = 0; vv < NWK; ++vv) W[vv] = wl[q * CH + kd * 9 * CHT + vv * 64 + lane];
      f32x4 (&acc)[NCO] = pass == 0 ? accA : accB;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const float xe = comp(R.e[kh], c), xo = comp(R.o[kh], c), x2 = rshl1(xe);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float xv = kw == 0 ? xe : (kw == 1 ? xo : x2);
#pragma unroll
            for (int co = 0; co < NCO; ++co) {
              const int fo = (kh * 3 + kw) * CHT + c * 4 * NCO + co * 4;
              acc[co] = mfa((fo & 63) >> 2, W[fo >> 6], xv, acc[co]);
            }
          }
        }
    }
  };
  Rows RA, RB;
  load(RA, 2 * d0, 0);
#pragma unroll 1
  for (int q = 0; q < 8; ++q) {                              // prologue: input plane 2 d0 is kd = 0 of the first output plane
    if (q + 1 < 8) load(RB, 2 * d0, q + 1); else load(RB, 2 * d0 + 1, 0);
    quad(RA, q, 0, cur, -1, nxt, false);
    RA = RB;
  }
#pragma unroll 1
  for (int j = d0; j < d0 + LD; ++j) {
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {                            // plane 2j + 1: kd = 1 of output plane j
      if (q + 1 < 8) load(RB, 2 * j + 1, q + 1); else load(RB, 2 * j + 2, 0);
      quad(RA, q, 1, cur, -1, nxt, false);
      RA = RB;
    }
    const bool more = j + 1 < d0 + LD;
#pragma unroll 1
    for (int q = 0; q < 8; ++q) {                            // plane 2j + 2: kd = 2 of plane j, kd = 0 of plane j + 1
      if (q + 1 < 8) load(RB, 2 * j + 2, q + 1); else load(RB, 2 * j + 3, 0);
      quad(RA, q, 2, cur, 0, nxt, more);
      RA = RB;
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      f32x4 v = cur[co];
      if (a.relu) v = relu4(v);
      raw_store4(v, ro, j * (kW16 * 16 * kRowQ16) + out_lane + co * kRowQ16, 0, 0);
      cur[co] = nxt[co];
      nxt[co] = bi[co];
    }
  }
}

// image of down_2's filter for down2_row_kernel: [cout group 2][channel quad 8][tap 27][ci4][32 couts] from the Conv3D
// layout [27][Cin = 32][Cout = 64]
__global__ void __launch_bounds__(256) down2_image_kernel(const float* w, float* dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * kDown2Group) return;
  const int g = i / kDown2Group, f0 = i - g * kDown2Group, q = f0 / 3456, f = f0 - q * 3456;
  const int tap = f >> 7, c = (f >> 5) & 3, co = f & 31;
  dst[i] = w[(tap * 32 + 4 * q + c) * 64 + g * 32 + co];
}
size_t down2_image_floats() { return 2 * (size_t)kDown2Group; }
int launch_down2_image(const float* w_tf, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(down2_image_kernel, dim3((2 * kDown2Group + 255) / 256), dim3(256), 0, s, w_tf, dst);
  return launch_ok("down2_image_kernel");
}
int launch_down2_row(const float* x, float* y, const float* w_image, const float* bias, int B, int relu, hipStream_t s) {
  Up1Args a{x, y, w_image, bias, B, relu};
  // Output planes per wave (a workgroup = 4 row quads x 2 plane segments, one per CU: its 110 KB filter image) by launch
  // size: B * 16 / LD workgroups for 256 CUs run in whole rounds, and a 39-cube launch at LD = 2 is 312 workgroups = a full
  // round and a fifth of one — 244.7 us, as long as 64 cubes (253.2).  Pick the LD whose rounds are fullest, larger LD first
  // among equals (fewer image copies): 64 cubes LD 4 244.9 us (LD 2 253.2, LD 1 267.0), 39 cubes LD 1 198.7 (LD 4 232.4);
  // same sums per output (profiles/r05_vH_tile_by_launch_size.txt).  PCGC_DOWN2_LD forces one.
  const char* e = getenv("PCGC_DOWN2_LD");
  int ld = e ? atoi(e) : 0;
  if (ld != 1 && ld != 2 && ld != 4) {
    double best = 1e30;
    for (int c : {4, 2, 1}) {
      const int wgs = B * 16 / c, rounds = (wgs + 255) / 256;
      const double cost = (1.0 + 0.125 / c) * rounds * 256.0 / wgs;      // image copy per LD planes / fill of the rounds
      if (cost < best - 1e-9) { best = cost; ld = c; }
    }
  }
  if (ld == 1) hipLaunchKernelGGL((down2_row_kernel<1>), dim3(B * (kW16 / 1 / 2) * 2), dim3(512), 0, s, a);
  else if (ld == 4) hipLaunchKernelGGL((down2_row_kernel<4>), dim3(B * (kW16 / 4 / 2) * 2), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((down2_row_kernel<2>), dim3(B * (kW16 / 2 / 2) * 2), dim3(512), 0, s, a);
  return launch_ok("down2_row_kernel");
}

// the block's LDS images from the TensorFlow layouts (the kernels' own gather formulas): w = {w11,b11,w12,b12,w21,b21,w22,...}
__global__ void __launch_bounds__(256) vrn64_image_kernel(const float* w11, const float* w21, const float* w12, const float* w22, float* dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kVrn64ImageFloats) return;
  if (i < kVrn64ImgA) {
    constexpr int CH = 28 * 64;
    const int q = i / CH, f = i - q * CH;
    dst[i] = f < 1728 ? w11[((f >> 6) * 64 + 4 * q + ((f >> 4) & 3)) * 16 + (f & 15)] : w21[(4 * q + ((f - 1728) >> 4)) * 16 + (f & 15)];
  } else if (i < kVrn64ImgA + kVrn64ImgB) {
    constexpr int CH = 27 * 64;
    const int j = i - kVrn64ImgA, qh = j / CH, f = j - qh * CH, q = qh >> 1, half = qh & 1;
    dst[i] = w12[((f >> 6) * 16 + 4 * q + ((f >> 4) & 3)) * 32 + 16 * half + (f & 15)];
  } else {
    constexpr int CH = 27 * 64;
    const int j = i - kVrn64ImgA - kVrn64ImgB, q = j / CH, f = j - q * CH;
    dst[i] = w22[((f >> 6) * 16 + 4 * q + ((f >> 4) & 3)) * 16 + (f & 15)];
  }
}
size_t vrn64_image_floats() { return kVrn64ImageFloats; }
int launch_vrn64_image(const float* const* w, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(vrn64_image_kernel, dim3((kVrn64ImageFloats + 255) / 256), dim3(256), 0, s, w[0], w[4], w[2], w[6], dst);
  return launch_ok("vrn64_image_kernel");
}

// which: 0 = kernel A, 1 = kernel B, 2 = kernel C.  All tensors Q4, D = 16, C = 64.  img: the block's image (launch_vrn64_image) or nullptr
int launch_vrn64_row(const float* x, float* t12, float* out, const float* const* w, int B, int which, hipStream_t s, const float* img) {
  Vrn64Args a;
  a.img = img;
  a.x = x; a.t12 = t12; a.out = out;
  a.w11 = w[0]; a.b11 = w[1]; a.w12 = w[2]; a.b12 = w[3]; a.w21 = w[4]; a.b21 = w[5];
  a.w22 = w[6]; a.b22 = w[7]; a.w23 = w[8]; a.b23 = w[9];
  a.B = B;
  // Planes per wave (LD = 1 / 2 / 4: 64 / 32 / 16 waves per cube) by launch size.  A launch runs in rounds of 2 048 waves (kernel A:
  // one 512-thread workgroup per CU) and lasts as long as its last round, however empty: 79 cubes (a decoder pipeline's second
  // slice) at LD = 2 are 2 528 waves = 238 us, the same as 103 cubes, at LD = 1 194 us; 103 cubes are 0.8 / 1.6 / 3.2 rounds
  // for LD = 4 / 2 / 1 — equally full, and the larger tile wins (220 / 237 / 254 us: fewer halo planes).  Cost model: (1 +
  // 0.25 / LD) for the halo x the fill of the rounds (profiles/r05_vH_tile_by_launch_size.txt; B and C follow A's choice in
  // every measured case).  The sums per output do not depend on it.  PCGC_V64_LD = 1 / 2 / 4 forces one, 0 = the fixed
  // thresholds used before (LD 1 below 32 cubes, 2 below 128, else 4).
  const char* e_ld = getenv("PCGC_V64_LD");
  int ld = e_ld ? atoi(e_ld) : -1;
  if (ld == 0) ld = B * 16 < 512 ? 1 : (B * 16 < 2048 ? 2 : 4);
  if (ld != 1 && ld != 2 && ld != 4) {
    double best = 1e30;
    for (int c : {4, 2, 1}) {
      const int waves = B * 4 * (kW16 / c), rounds = (waves + 2047) / 2048;
      const double cost = (1.0 + 0.25 / c) * rounds * 2048.0 / waves;
      if (cost < best - 1e-9) { best = cost; ld = c; }
    }
  }
  if (ld == 1) {
    constexpr int LD = 1;
    const int blocks = (B * 4 * (kW16 / LD) + 7) / 8;
    if (which == 0) hipLaunchKernelGGL((vrn64a_row_kernel<LD>), dim3(blocks), dim3(512), 0, s, a);
    else if (which == 1) hipLaunchKernelGGL((vrn64b_row_kernel<LD>), dim3(blocks), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((vrn64c_row_kernel<LD>), dim3(blocks), dim3(512), 0, s, a);
    return launch_ok("vrn64 row kernel");
  }
  if (ld == 2) {
    constexpr int LD = 2;
    const int blocks = (B * 4 * (kW16 / LD) + 7) / 8;
    if (which == 0) hipLaunchKernelGGL((vrn64a_row_kernel<LD>), dim3(blocks), dim3(512), 0, s, a);
    else if (which == 1) hipLaunchKernelGGL((vrn64b_row_kernel<LD>), dim3(blocks), dim3(512), 0, s, a);
    else hipLaunchKernelGGL((vrn64c_row_kernel<LD>), dim3(blocks), dim3(512), 0, s, a);
    return launch_ok("vrn64 row kernel");
  }
  constexpr int LD = 4;
  const int waves = B * 4 * (kW16 / LD);
  const int blocks = (waves + 7) / 8;
  if (which == 0) hipLaunchKernelGGL((vrn64a_row_kernel<LD>), dim3(blocks), dim3(512), 0, s, a);
  else if (which == 1) hipLaunchKernelGGL((vrn64b_row_kernel<LD>), dim3(blocks), dim3(512), 0, s, a);
  else hipLaunchKernelGGL((vrn64c_row_kernel<LD>), dim3(blocks), dim3(512), 0, s, a);
  return launch_ok("vrn64 row kernel");
}

}  // namespace pcgc
