// Voxception-ResNet block for C = 32 at 32^3 (the middle stage of both transforms, models/model_voxception.py:56-68)
// on the same v_mfma_f32_4x4x1_16B_f32 row scheme as vrn_row.hip, with TWO rows of the cube per 64-lane vector:
//   lane = (row parity, w):  lanes 0..31 = row a, lanes 32..63 = row a + 1  ("pair vector" starting at row a).
// Tensors are Q4 [b][d][h][C/4][w][4]; a pair vector of one channel quad is one dwordx4 buffer load per lane (two 512-B
// segments).  An output row pair (2k, 2k+1) takes its three kh taps from three pair vectors of the input:
//   kh = 1: rows (2k, 2k+1)  = the ALIGNED pair P[k];   kh = 0: rows (2k-1, 2k) = O[k];   kh = 2: rows (2k+1, 2k+2) = O[k+1]
// so the odd-aligned pairs O are simply loaded as well (each input row is read twice; loads are ~2 % of the instruction
// stream) instead of being permuted across lanes.  kw = 0 / 2 are DPP wave shifts by one lane, with the lane that
// crosses from one row into the other (32 for the right shift, 31 for the left shift) forced to zero = 'same' padding.
// The wave walks LD planes with three rotating plane accumulators exactly like vrn_row.hip.
// Weights: a layer of this stage has up to 6912 values — too many for registers — so each workgroup packs them once into
// LDS as one chunk per input-channel quad, laid out so that 64 consecutive floats = one A-operand VGPR whose 16 blocks
// are 16 (tap, ci, cout-quad) combinations; a quad step fetches its 14 or 27 VGPRs with lane-linear ds_read_b32.
//   kernel A : t12 = [ relu(conv1_1(x)) 3^3 32->8 | relu(conv2_1(x)) 1^3 32->8 ]
//   kernel BC: out = relu(x + [ relu(conv1_2(t11)) 3^3 8->16 | relu(conv2_3(relu(conv2_2(t21)))) 3^3 8->8, 1^3 8->16 ])
// Summation order per output: bias, then (plane, channel, kh, kw): fixed, batch- and placement-independent.
#include <type_traits>
#include "row_common.h"

namespace pcgc {

constexpr int kW = 32;                    // cube edge of this stage
constexpr int kRowQ = kW * 16;            // bytes of one (row, channel quad)

// The lane that would take its value from the other row of the pair gets zero through a multiplication by a per-lane 0 / 1
// constant (lane_masks32: opaque to the optimiser, which would otherwise turn the product back into shift + select): the lane
// shift folds into the multiply — ONE v_mul_f32_dpp where shift + v_cndmask were two.  x * 1 is x; x * 0 is +-0, and every sum a
// shifted value enters ends in a ReLU (or started from a bias), so a zero's sign never reaches an output.
__device__ __forceinline__ float shr1p(float v, float not_first_of_row2) {   // lane i <- lane i-1 inside each 32-lane row
  return shr1(v) * not_first_of_row2;
}
__device__ __forceinline__ float shl1p(float v, float not_last_of_row1) {    // lane i <- lane i+1 inside each 32-lane row
  return shl1(v) * not_last_of_row1;
}
struct LaneMasks32 {
  float m32, m31;      // 0 on lane 32 / lane 31, 1 elsewhere
};
__device__ __forceinline__ LaneMasks32 lane_masks32(int lane) {
  LaneMasks32 m{lane == 32 ? 0.f : 1.f, lane == 31 ? 0.f : 1.f};
  asm volatile("" : "+v"(m.m32), "+v"(m.m31));
  return m;
}

// Byte offset of (plane p, row a, quad q) and of a lane inside its row for a tensor with NQ quads per voxel: Q4
// [d][h][NQ][w][4], or NDHWC [d][h][w][4 NQ] when NHWC (the training step's tensors).
template <int NQ, bool NHWC>
__device__ __forceinline__ int row_base32(int p, int a, int q) {
  return NHWC ? (p * kW + a) * (kW * NQ * 16) + q * 16 : ((p * kW + a) * NQ + q) * kRowQ;
}
template <int NQ, bool NHWC>
__device__ __forceinline__ int lane_off32(int lane) {      // upper half = the next row
  return NHWC ? (lane >> 5) * (kW * NQ * 16) + (lane & 31) * (NQ * 16) : (lane >> 5) * (NQ * kRowQ) + (lane & 31) * 16;
}

// pair vector (rows a, a+1) of plane p, channel quad q of a tensor with NQ quads; rows / planes outside the cube read 0
template <int NQ, bool NHWC = false>
__device__ __forceinline__ f32x4 load_pair(i32x4 rs, int lane_off, bool hi, int p, int q, int a) {
  const bool pin = (unsigned)p < (unsigned)kW;
  const bool lo_ok = pin && (unsigned)a < (unsigned)kW, hi_ok = pin && (unsigned)(a + 1) < (unsigned)kW;
  const int base = row_base32<NQ, NHWC>(p, a, q);
  const bool ok = hi ? hi_ok : lo_ok;
  return raw_load4(rs, ok ? base + lane_off : kOOB, 0, 0);
}

struct Tile32 {
  int b, k0, d0;
};
template <int TP, int LD>
__device__ __forceinline__ Tile32 wave_tile32() {
  int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  Tile32 t;
  t.k0 = (wv % (kW / 2 / TP)) * TP; wv /= (kW / 2 / TP);
  t.d0 = (wv % (kW / LD)) * LD; wv /= (kW / LD);
  t.b = wv;
  return t;
}

// --- exact skipping of empty space in the analysis (RowSkip, common.h; vrn_row.hip has the 64^3 stage) ---------------
// The wave's tile through the launch's tile order; *heavy = false: the tile is copied from the empty-cube response.
template <int TP, int LD>
__device__ __forceinline__ Tile32 wave_tile32_ordered(const RowSkip& k, bool* heavy) {
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  int wv = wid;
  *heavy = true;
  if (k.order) {
    wv = __builtin_amdgcn_readfirstlane((int)(k.order[wid] & ~kTileUnread));
    *heavy = wid < (int)*k.n_heavy;
    if (!*heavy && k.counter && (threadIdx.x & 63) == 0) atomicAdd(k.counter, 1u);
  }
  Tile32 t;
  t.k0 = (wv % (kW / 2 / TP)) * TP; wv /= (kW / 2 / TP);
  t.d0 = (wv % (kW / LD)) * LD; wv /= (kW / LD);
  t.b = wv;
  return t;
}
// every wave of the workgroup copies: no weights to stage (workgroup-uniform; heavy tiles come first in the order)
__device__ __forceinline__ bool workgroup_all_empty(const RowSkip& k) {
  return k.order && (int)(blockIdx.x * 4) >= (int)*k.n_heavy;
}
// TP row pairs x LD planes of an NQ-quad Q4 tensor at 32^3, copied from the empty-cube response (one cube, same layout)
template <int TP, int LD, int NQ>
__device__ __forceinline__ void copy_empty_tile32(const float* empty, float* dst_cube, const Tile32& tl, int lane) {
  const char* src = reinterpret_cast<const char*>(empty) + lane_off32<NQ, false>(lane);
  char* dst = reinterpret_cast<char*>(dst_cube) + lane_off32<NQ, false>(lane);
#pragma unroll
  for (int p = 0; p < LD; ++p)
#pragma unroll
    for (int j = 0; j < TP; ++j) {
      f32x4 v[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[q] = *reinterpret_cast<const f32x4*>(src + row_base32<NQ, false>(tl.d0 + p, 2 * (tl.k0 + j), q));
#pragma unroll
      for (int q = 0; q < NQ; ++q) *reinterpret_cast<f32x4*>(dst + row_base32<NQ, false>(tl.d0 + p, 2 * (tl.k0 + j), q)) = v[q];
    }
}

// LDS images of a block's filters, built once per net (vrn32_image_kernel): [A: 8 chunks of 896][BC: 2 chunks of 1728 conv1_2,
// 2 chunks of 896 conv2_2] floats = what the staging gathers from the TensorFlow layouts, copied with 16-byte loads instead
// of one indexed gather per float (the gathers' index arithmetic was 10-15 % of these kernels' vector instructions)
constexpr int kVrn32ImgA = 8 * 896, kVrn32ImgBC = 2 * 27 * 64 + 2 * 896;
constexpr int kVrn32ImageFloats = kVrn32ImgA + kVrn32ImgBC;

struct Vrn32Args {
  const float* img = nullptr;    // the block's image (nullptr: gather from the TensorFlow layouts in the kernel)
  const float* x;      // block input,  Q4 [B][32][32][8][32][4]
  float* t12;          // scratch,      Q4 [B][32][32][4][32][4]: quads 0,1 = tensor1_1, quads 2,3 = tensor2_1
  float* out;          // block output, Q4 like x (may alias x)
  const float *w11, *b11, *w21, *b21, *w12, *b12, *w22, *b22, *w23, *b23;   // TensorFlow layouts
  int B;
  // training variant (TRAIN = true): x / out / pre NDHWC [B][32][32][32][32]; t12 = tensor1_1, t21 = tensor2_1,
  // t22 = relu(conv2_2), NDHWC [..][8] each; pre = [relu(conv1_2) | relu(conv2_3)] (the reverse pass needs its sign)
  float* t21 = nullptr;
  float* t22 = nullptr;
  float* pre = nullptr;
  int* pre_signs = nullptr;   // instead of pre: one word per voxel, bit c = (pre[c] > 0) — all the reverse pass reads of it
  RowSkip skip;        // inference, analysis only: tile order + empty-cube response (skip.order != nullptr)
};

// One input channel of a quad step: TP aligned pairs P, TP+1 odd pairs O -> 3^3 taps into NCO output-channel quads.
// WMAP(tap, coq, c) gives (weight register index, abid) of the layer's packed chunk.
template <int TP, int NCO, int NW, class WMAP>
__device__ __forceinline__ void pair_channel(f32x4 (&acc)[3][TP][NCO], const float (&W)[NW], int c, const f32x4 (&P)[TP],
                                             const f32x4 (&O)[TP + 1], bool v0, bool v1, bool v2, float l32, float l31, WMAP wmap) {
  float p0[TP], pm[TP], pp[TP], o0[TP + 1], om[TP + 1], op[TP + 1];
#pragma unroll
  for (int j = 0; j < TP; ++j) { p0[j] = comp(P[j], c); pm[j] = shr1p(p0[j], l32); pp[j] = shl1p(p0[j], l31); }
#pragma unroll
  for (int j = 0; j <= TP; ++j) { o0[j] = comp(O[j], c); om[j] = shr1p(o0[j], l32); op[j] = shl1p(o0[j], l31); }
  const bool vj[3] = {v0, v1, v2};
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const int kd = 2 - jj;                     // input plane p feeds output plane p - 1 + jj
    if (vj[jj]) {
#pragma unroll
      for (int j = 0; j < TP; ++j)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int t = (kd * 3 + kh) * 3 + kw;
            const float xv = kh == 1 ? (kw == 0 ? pm[j] : (kw == 1 ? p0[j] : pp[j]))
                                     : (kh == 0 ? (kw == 0 ? om[j] : (kw == 1 ? o0[j] : op[j]))
                                                : (kw == 0 ? om[j + 1] : (kw == 1 ? o0[j + 1] : op[j + 1])));
#pragma unroll
            for (int coq = 0; coq < NCO; ++coq) acc[jj][j][coq] = mfa(wmap.abid(t, coq, c), W[wmap.reg(t)], xv, acc[jj][j][coq]);
          }
    }
  }
}

// pair_channel for the four channels of a quad with the validity tests hoisted: one wave-uniform branch per (quad, output
// plane) instead of one per (channel, output plane).  Same order of contributions per accumulator (channel, then kh, kw).
template <int TP, int NCO, int NW, class WMAP>
__device__ __forceinline__ void pair_quad(f32x4 (&acc)[3][TP][NCO], const float (&W)[NW], const f32x4 (&P)[TP], const f32x4 (&O)[TP + 1],
                                          bool v0, bool v1, bool v2, float l32, float l31, WMAP wmap) {
  float p0[4][TP], pm[4][TP], pp[4][TP], o0[4][TP + 1], om[4][TP + 1], op[4][TP + 1];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int j = 0; j < TP; ++j) { p0[c][j] = comp(P[j], c); pm[c][j] = shr1p(p0[c][j], l32); pp[c][j] = shl1p(p0[c][j], l31); }
#pragma unroll
    for (int j = 0; j <= TP; ++j) { o0[c][j] = comp(O[j], c); om[c][j] = shr1p(o0[c][j], l32); op[c][j] = shl1p(o0[c][j], l31); }
  }
  const bool vj[3] = {v0, v1, v2};
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const int kd = 2 - jj;
    if (vj[jj]) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int t = (kd * 3 + kh) * 3 + kw;
              const float xv = kh == 1 ? (kw == 0 ? pm[c][j] : (kw == 1 ? p0[c][j] : pp[c][j]))
                                       : (kh == 0 ? (kw == 0 ? om[c][j] : (kw == 1 ? o0[c][j] : op[c][j]))
                                                  : (kw == 0 ? om[c][j + 1] : (kw == 1 ? o0[c][j + 1] : op[c][j + 1])));
#pragma unroll
              for (int coq = 0; coq < NCO; ++coq) acc[jj][j][coq] = mfa(wmap.abid(t, coq, c), W[wmap.reg(t)], xv, acc[jj][j][coq]);
            }
    }
  }
}

struct Map8 {    // chunk [tap][ci4][8 couts] (+ 32 floats of a 1^3 layer behind tap 26): 32 floats per tap
  __device__ static constexpr int reg(int t) { return t >> 1; }
  __device__ static constexpr int abid(int t, int coq, int c) { return (t & 1) * 8 + c * 2 + coq; }
};
struct Map16 {   // chunk [tap][ci4][16 couts]: 64 floats per tap
  __device__ static constexpr int reg(int t) { return t; }
  __device__ static constexpr int abid(int, int coq, int c) { return c * 4 + coq; }
};

// Kernel A's form of pair_channel: the lane shifts act on the 8 output channels instead of the 32 input channels (a
// lane shift commutes with the convolution, vrn_row.hip kernel A).  S[set][kw] collects tap column kw applied to the
// UNSHIFTED pair vectors; a finished plane is S[1] + shr(S[0]) + shl(S[2]) with the row-crossing lane zeroed.  The
// input-shift form spent 2 x (shift + select) per channel and pair vector — 160 VALU instructions per 896 MFMAs, in the
// partial plane steps too (VALU : MFMA 0.40 in profiles/r02_vB_pmc_per_kernel.csv); this form spends 48 per finished
// pair.  Summation order per output: per kw column bias / 0, then (plane, channel, kh); then S1 + shr(S0) + shl(S2).
template <int TP, int NW, class WMAP>
__device__ __forceinline__ void pair_channel_os(f32x4 (&S)[3][3][TP][2], const float (&W)[NW], int c, const f32x4 (&P)[TP],
                                                const f32x4 (&O)[TP + 1], bool v0, bool v1, bool v2, WMAP wmap) {
  const bool vj[3] = {v0, v1, v2};
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const int kd = 2 - jj;                     // input plane p feeds output plane p - 1 + jj
    if (vj[jj]) {
#pragma unroll
      for (int j = 0; j < TP; ++j)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const float xv = kh == 1 ? comp(P[j], c) : (kh == 0 ? comp(O[j], c) : comp(O[j + 1], c));
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int t = (kd * 3 + kh) * 3 + kw;
#pragma unroll
            for (int coq = 0; coq < 2; ++coq) S[jj][kw][j][coq] = mfa(wmap.abid(t, coq, c), W[wmap.reg(t)], xv, S[jj][kw][j][coq]);
          }
        }
    }
  }
}
// pair_channel_os for the four channels of a quad, validity tests hoisted (see pair_quad); conv2_1 (1^3, register 13's upper
// half) of the centre plane rides in the jj = 1 block.  Same order of contributions per accumulator.
template <int TP, int NW, class WMAP>
__device__ __forceinline__ void pair_quad_os(f32x4 (&S)[3][3][TP][2], f32x4 (&acc2)[1][TP][2], const float (&W)[NW], const f32x4 (&P)[TP],
                                             const f32x4 (&O)[TP + 1], bool v0, bool v1, bool v2, WMAP wmap) {
  const bool vj[3] = {v0, v1, v2};
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) {
    const int kd = 2 - jj;
    if (vj[jj]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const float xv = kh == 1 ? comp(P[j], c) : (kh == 0 ? comp(O[j], c) : comp(O[j + 1], c));
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int t = (kd * 3 + kh) * 3 + kw;
#pragma unroll
              for (int coq = 0; coq < 2; ++coq) S[jj][kw][j][coq] = mfa(wmap.abid(t, coq, c), W[wmap.reg(t)], xv, S[jj][kw][j][coq]);
            }
          }
        if (jj == 1) {
#pragma unroll
          for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int coq = 0; coq < 2; ++coq) acc2[0][j][coq] = mfa(8 + c * 2 + coq, W[13], comp(P[j], c), acc2[0][j][coq]);
        }
      }
    }
  }
}
__device__ __forceinline__ f32x4 shr4p(f32x4 v, float l32) { return f32x4{shr1p(v[0], l32), shr1p(v[1], l32), shr1p(v[2], l32), shr1p(v[3], l32)}; }
__device__ __forceinline__ f32x4 shl4p(f32x4 v, float l31) { return f32x4{shl1p(v[0], l31), shl1p(v[1], l31), shl1p(v[2], l31), shl1p(v[3], l31)}; }

// ---------------------------------------------------------------------------------------------------------------
// kernel A
// ---------------------------------------------------------------------------------------------------------------
template <int TP, int LD, bool TRAIN = false, bool QJ = true>
__global__ void __launch_bounds__(256, 2) vrn32a_row_kernel(Vrn32Args a) {
  constexpr int CH = 896;                                   // floats per quad chunk: 27*4*8 conv1_1 + 4*8 conv2_1
  __shared__ float wl[8 * CH];
  const bool no_work = !TRAIN && workgroup_all_empty(a.skip);
  if (!no_work) {
    if (a.img) stage_image<8 * CH>(wl, a.img);
    else
      stage_indexed<8 * CH>(wl, [&](int i) {
        const int q = i / CH, f = i - q * CH;
        return f < 864 ? a.w11[((f >> 5) * 32 + 4 * q + ((f >> 3) & 3)) * 8 + (f & 7)] : a.w21[(4 * q + ((f - 864) >> 3)) * 8 + (f & 7)];
      });
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const bool hi = lane >= 32;
  const LaneMasks32 lm = lane_masks32(lane);
  const float l32 = lm.m32, l31 = lm.m31;
  bool heavy = true;
  const Tile32 tl = TRAIN ? wave_tile32<TP, LD>() : wave_tile32_ordered<TP, LD>(a.skip, &heavy);
  const int k0 = tl.k0, d0 = tl.d0;
  if (!heavy) {
    copy_empty_tile32<TP, LD, 4>(a.skip.empty, a.t12 + (size_t)tl.b * kW * kW * kW * 16, tl, lane);
    return;
  }
  const f32x4 bi[2] = {{a.b11[0], a.b11[1], a.b11[2], a.b11[3]}, {a.b11[4], a.b11[5], a.b11[6], a.b11[7]}};
  const f32x4 bi2[2] = {{a.b21[0], a.b21[1], a.b21[2], a.b21[3]}, {a.b21[4], a.b21[5], a.b21[6], a.b21[7]}};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 S[3][3][TP][2], acc2[1][TP][2];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int r = 0; r < TP; ++r) { S[j][kw][r][0] = kw == 1 ? bi[0] : zero4; S[j][kw][r][1] = kw == 1 ? bi[1] : zero4; }
  const i32x4 rs = make_rsrc(a.x + (size_t)tl.b * kW * kW * kW * 32, kW * kW * kW * 32 * 4);
  const int lane_off = lane_off32<8, TRAIN>(lane);                                   // x has 8 quads per voxel
  // inference: t12 = one Q4 tensor of 4 quads (0,1 = tensor1_1, 2,3 = tensor2_1); training: two NDHWC tensors of 2 quads
  const i32x4 rt1 = make_rsrc(a.t12 + (size_t)tl.b * kW * kW * kW * (TRAIN ? 8 : 16), kW * kW * kW * (TRAIN ? 8 : 16) * 4);
  const i32x4 rt2 = TRAIN ? make_rsrc(a.t21 + (size_t)tl.b * kW * kW * kW * 8, kW * kW * kW * 8 * 4) : rt1;
  constexpr int TQ = TRAIN ? 2 : 4;
  const int lane_off_t = lane_off32<TQ, TRAIN>(lane);
  f32x4 PA[TP], OA[TP + 1], PB[TP], OB[TP + 1];
  auto load = [&](f32x4 (&P)[TP], f32x4 (&O)[TP + 1], int p, int q) {
#pragma unroll
    for (int j = 0; j < TP; ++j) P[j] = load_pair<8, TRAIN>(rs, lane_off, hi, p, q, 2 * (k0 + j));
#pragma unroll
    for (int j = 0; j <= TP; ++j) O[j] = load_pair<8, TRAIN>(rs, lane_off, hi, p, q, 2 * (k0 + j) - 1);
  };
  auto quad = [&](const f32x4 (&P)[TP], const f32x4 (&O)[TP + 1], int q, bool v0, bool v1, bool v2) {
    float W[14];
#pragma unroll
    for (int v = 0; v < 14; ++v) W[v] = wl[q * CH + v * 64 + lane];
    if constexpr (QJ) { pair_quad_os<TP, 14>(S, acc2, W, P, O, v0, v1, v2, Map8()); return; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      pair_channel_os<TP, 14>(S, W, c, P, O, v0, v1, v2, Map8());
      if (v1) {                                             // conv2_1 on the centre voxel: lanes 32..63 of register 13
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
          for (int coq = 0; coq < 2; ++coq) acc2[0][j][coq] = mfa(8 + c * 2 + coq, W[13], comp(P[j], c), acc2[0][j][coq]);
      }
    }
  };
  load(PA, OA, d0 - 1, 0);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kW;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
#pragma unroll
    for (int r = 0; r < TP; ++r) { acc2[0][r][0] = bi2[0]; acc2[0][r][1] = bi2[1]; }
#pragma unroll 1
    for (int q = 0; q < 8; q += 2) {
      load(PB, OB, p, q + 1);
      quad(PA, OA, q, v0, v1, v2);
      if (q + 2 < 8) load(PA, OA, p, q + 2); else load(PA, OA, p + 1, 0);
      quad(PB, OB, q + 1, v0, v1, v2);
    }
    if (v1) {
#pragma unroll
      for (int r = 0; r < TP; ++r)
#pragma unroll
        for (int coq = 0; coq < 2; ++coq)
          raw_store4(relu4(acc2[0][r][coq]), rt2, row_base32<TQ, TRAIN>(p, 2 * (k0 + r), (TRAIN ? 0 : 2) + coq) + lane_off_t, 0, 0);
    }
    if (p - 1 >= d0) {
#pragma unroll
      for (int r = 0; r < TP; ++r)
#pragma unroll
        for (int coq = 0; coq < 2; ++coq) {
          const f32x4 y = S[0][1][r][coq] + shr4p(S[0][0][r][coq], l32) + shl4p(S[0][2][r][coq], l31);
          raw_store4(relu4(y), rt1, row_base32<TQ, TRAIN>(p - 1, 2 * (k0 + r), coq) + lane_off_t, 0, 0);
        }
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int r = 0; r < TP; ++r)
#pragma unroll
        for (int coq = 0; coq < 2; ++coq) {
          S[0][kw][r][coq] = S[1][kw][r][coq]; S[1][kw][r][coq] = S[2][kw][r][coq]; S[2][kw][r][coq] = kw == 1 ? bi[coq] : zero4;
        }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// kernel BC (one row pair per wave)
// ---------------------------------------------------------------------------------------------------------------
// NONNEG: the caller vouches that x >= 0 (the block follows a ReLU layer); the sum with the ReLU'd branches needs no
// second ReLU then (bit-identical)
template <int LD, bool TRAIN = false, bool NONNEG = false, bool QJ = true>
__global__ void __launch_bounds__(256, 2) vrn32bc_row_kernel(Vrn32Args a) {
  constexpr int C12 = 27 * 64, C22 = 896;                   // floats per quad chunk of conv1_2 / conv2_2
  __shared__ float wl[2 * C12 + 2 * C22];
  const bool no_work = !TRAIN && workgroup_all_empty(a.skip);
  if (!no_work) {
    if (a.img) stage_image<2 * C12 + 2 * C22>(wl, a.img + kVrn32ImgA);
    else {
      stage_indexed<2 * C12>(wl, [&](int i) {                 // [q][tap][ci4][16]
        const int q = i / C12, f = i - q * C12;
        return a.w12[((f >> 6) * 8 + 4 * q + ((f >> 4) & 3)) * 16 + (f & 15)];
      });
      stage_indexed<2 * C22>(wl + 2 * C12, [&](int i) {       // [q][tap][ci4][8], 864 used
        const int q = i / C22, f = i - q * C22;
        return f < 864 ? a.w22[((f >> 5) * 8 + 4 * q + ((f >> 3) & 3)) * 8 + (f & 7)] : 0.f;
      });
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const bool hi = lane >= 32;
  const LaneMasks32 lm = lane_masks32(lane);
  const float l32 = lm.m32, l31 = lm.m31;
  bool heavy = true;
  const Tile32 tl = TRAIN ? wave_tile32<1, LD>() : wave_tile32_ordered<1, LD>(a.skip, &heavy);
  const int k0 = tl.k0, d0 = tl.d0;
  if (!heavy) {
    copy_empty_tile32<1, LD, 8>(a.skip.empty, a.out + (size_t)tl.b * kW * kW * kW * 32, tl, lane);
    return;
  }
  const float W23[2] = {a.w23[lane], a.w23[64 + lane]};     // [8][16]: register ci>>2, abid (ci&3)*4 + coq
  f32x4 bi12[4], bi23[4], bi22[2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bi12[q] = f32x4{a.b12[4 * q], a.b12[4 * q + 1], a.b12[4 * q + 2], a.b12[4 * q + 3]};
    bi23[q] = f32x4{a.b23[4 * q], a.b23[4 * q + 1], a.b23[4 * q + 2], a.b23[4 * q + 3]};
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) bi22[q] = f32x4{a.b22[4 * q], a.b22[4 * q + 1], a.b22[4 * q + 2], a.b22[4 * q + 3]};
  f32x4 acc12[3][1][4], acc22[3][1][2];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc12[j][0][q] = bi12[q];
    acc22[j][0][0] = bi22[0]; acc22[j][0][1] = bi22[1];
  }
  constexpr int TQ = TRAIN ? 2 : 4;                         // quads per voxel of the tensor(s) holding tensor1_1 / tensor2_1
  const i32x4 rs = make_rsrc(a.t12 + (size_t)tl.b * kW * kW * kW * 4 * TQ, kW * kW * kW * 4 * TQ * 4);
  const i32x4 rs2 = TRAIN ? make_rsrc(a.t21 + (size_t)tl.b * kW * kW * kW * 8, kW * kW * kW * 8 * 4) : rs;
  const i32x4 rx = make_rsrc(a.x + (size_t)tl.b * kW * kW * kW * 32, kW * kW * kW * 32 * 4);
  const i32x4 ro = make_rsrc(a.out + (size_t)tl.b * kW * kW * kW * 32, kW * kW * kW * 32 * 4);
  const int lane_off = lane_off32<TQ, TRAIN>(lane);
  const int lane_off_x = lane_off32<8, TRAIN>(lane);                                 // x / out / pre: 8 quads per voxel
  f32x4 PA[1], OA[2], PB[1], OB[2];
  auto load = [&](f32x4 (&P)[1], f32x4 (&O)[2], int p, int q) {   // q 0,1 = tensor1_1, q 2,3 = tensor2_1
    const i32x4 r = (TRAIN && q >= 2) ? rs2 : rs;
    const int qq = TRAIN ? (q & 1) : q;
    P[0] = load_pair<TQ, TRAIN>(r, lane_off, hi, p, qq, 2 * k0);
    O[0] = load_pair<TQ, TRAIN>(r, lane_off, hi, p, qq, 2 * k0 - 1);
    O[1] = load_pair<TQ, TRAIN>(r, lane_off, hi, p, qq, 2 * k0 + 1);
  };
  load(PA, OA, d0 - 1, 0);
  load(PB, OB, d0 - 1, 1);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kW;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
    {   // conv1_2, input quads 0 and 1 (tensor1_1)
      float W[27];
#pragma unroll
      for (int v = 0; v < 27; ++v) W[v] = wl[v * 64 + lane];
      if constexpr (QJ) pair_quad<1, 4, 27>(acc12, W, PA, OA, v0, v1, v2, l32, l31, Map16());
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) pair_channel<1, 4, 27>(acc12, W, c, PA, OA, v0, v1, v2, l32, l31, Map16());
      }
      load(PA, OA, p, 2);
#pragma unroll
      for (int v = 0; v < 27; ++v) W[v] = wl[C12 + v * 64 + lane];
      if constexpr (QJ) pair_quad<1, 4, 27>(acc12, W, PB, OB, v0, v1, v2, l32, l31, Map16());
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) pair_channel<1, 4, 27>(acc12, W, c, PB, OB, v0, v1, v2, l32, l31, Map16());
      }
      load(PB, OB, p, 3);
    }
    // residual row pair of output plane p-1 (out of range before the first finished plane: zeros, and the stores drop)
    const int obase = p - 1 >= d0 ? row_base32<8, TRAIN>(p - 1, 2 * k0, 0) + lane_off_x : kOOB;
    constexpr int QS = TRAIN ? 16 : kRowQ;                  // byte step from quad to quad of a voxel
    f32x4 res[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) res[q] = raw_load4(rx, obase + q * QS, 0, 0);
    {   // conv2_2, input quads 2 and 3 (tensor2_1)
      float W[14];
#pragma unroll
      for (int v = 0; v < 14; ++v) W[v] = wl[2 * C12 + v * 64 + lane];
      if constexpr (QJ) pair_quad<1, 2, 14>(acc22, W, PA, OA, v0, v1, v2, l32, l31, Map8());
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) pair_channel<1, 2, 14>(acc22, W, c, PA, OA, v0, v1, v2, l32, l31, Map8());
      }
      load(PA, OA, p + 1, 0);
#pragma unroll
      for (int v = 0; v < 14; ++v) W[v] = wl[2 * C12 + C22 + v * 64 + lane];
      if constexpr (QJ) pair_quad<1, 2, 14>(acc22, W, PB, OB, v0, v1, v2, l32, l31, Map8());
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) pair_channel<1, 2, 14>(acc22, W, c, PB, OB, v0, v1, v2, l32, l31, Map8());
      }
      load(PB, OB, p + 1, 1);
    }
    // output plane p-1: conv2_3 on relu(conv2_2), residual, ReLU, store
    const f32x4 t22[2] = {relu4(acc22[0][0][0]), relu4(acc22[0][0][1])};
    f32x4 q3[4] = {bi23[0], bi23[1], bi23[2], bi23[3]};
#pragma unroll
    for (int ci = 0; ci < 8; ++ci)
#pragma unroll
      for (int coq = 0; coq < 4; ++coq) q3[coq] = mfa((ci & 3) * 4 + coq, W23[ci >> 2], comp(t22[ci >> 2], ci & 3), q3[coq]);
    unsigned sbits = 0;                                     // TRAIN with pre_signs: bit c = (pre[c] > 0), one word per voxel
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 p12 = relu4(acc12[0][0][q]), p23 = relu4(q3[q]);
      raw_store4(NONNEG ? res[q] + p12 : relu4(res[q] + p12), ro, obase + q * QS, 0, 0);
      raw_store4(NONNEG ? res[4 + q] + p23 : relu4(res[4 + q] + p23), ro, obase + (4 + q) * QS, 0, 0);
      if constexpr (TRAIN) {                                // what the reverse pass reads: the pre-residual output (or its signs) ...
        if (a.pre_signs) {
#pragma unroll
          for (int i = 0; i < 4; ++i) sbits |= (p12[i] > 0.f ? 1u << (4 * q + i) : 0u) | (p23[i] > 0.f ? 1u << (16 + 4 * q + i) : 0u);
        } else {
          const i32x4 rp = make_rsrc(a.pre + (size_t)tl.b * kW * kW * kW * 32, kW * kW * kW * 32 * 4);
          raw_store4(p12, rp, obase + q * QS, 0, 0);
          raw_store4(p23, rp, obase + (4 + q) * QS, 0, 0);
        }
      }
    }
    if constexpr (TRAIN) {
      if (a.pre_signs) {                                    // rows 2 k0, 2 k0 + 1 of a [32][32][32] word tensor are 64 consecutive words
        const i32x4 rg = make_rsrc(a.pre_signs + (size_t)tl.b * kW * kW * kW, kW * kW * kW * 4);
        raw_store1i((int)sbits, rg, p - 1 >= d0 ? (((p - 1) * kW + 2 * k0) * kW + lane) * 4 : kOOB, 0, 0);
      }
    }
    if constexpr (TRAIN) {                                  // ... and tensor2_2
      const i32x4 r22 = make_rsrc(a.t22 + (size_t)tl.b * kW * kW * kW * 8, kW * kW * kW * 8 * 4);
      const int tbase = p - 1 >= d0 ? row_base32<2, true>(p - 1, 2 * k0, 0) + lane_off : kOOB;
      raw_store4(t22[0], r22, tbase, 0, 0);
      raw_store4(t22[1], r22, tbase + 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc12[0][0][q] = acc12[1][0][q]; acc12[1][0][q] = acc12[2][0][q]; acc12[2][0][q] = bi12[q]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) { acc22[0][0][q] = acc22[1][0][q]; acc22[1][0][q] = acc22[2][0][q]; acc22[2][0][q] = bi22[q]; }
  }
}

// which: 0 = kernel A, 1 = kernel BC.  All tensors Q4, D = 32, C = 32.  w = {w11,b11,w12,b12,w21,b21,w22,b22,w23,b23}
// ---------------------------------------------------------------------------------------------------------------
// up_2: stride-2 transposed conv 3^3, 32 -> 16 channels, 32^3 -> 64^3 (models/model_voxception.py:167-172), + ReLU.
//   y[o] = bias + sum_{o = 2i + k} x[i] W[k]  per axis (the alignment of tconv_mfma_kernel, checked against the oracle):
//   an even output o = 2i takes tap k = 0 from input i and k = 2 from input i - 1, an odd output o = 2i + 1 takes k = 1
//   from input i.
// Lane = INPUT voxel of a pair vector (rows 2k, 2k+1 of plane p), so no lane multiplies a zero: every lane owns the two
// outputs ow = 2i (acc "e") and ow = 2i + 1 (acc "o") of its voxel; x[i-1] is one DPP shift.  Input row ih feeds output
// rows 2ih (kh = 0) and 2ih + 1 (kh = 1); the odd-aligned pair (rows 2k-1, 2k) is x[ih - 1] for both halves and gives
// kh = 2 of the even output rows.  Along d the wave slides: input plane p completes output plane 2p (kd = 0; its kd = 2
// part came from plane p - 1), produces 2p + 1 (kd = 1) and opens 2p + 2 (kd = 2).  Each wave computes NCO of the four
// output-channel quads (the grid's fastest index), weights per (channel quad, cout group) chunk in LDS.
// x Q4 [B][32][32][8][32][4], y Q4 [B][64][64][4][64][4], w = the filter's LDS image (row_image_kernel, kind 0).
// ---------------------------------------------------------------------------------------------------------------
struct UpRowArgs {
  const float* x;
  float* y;
  const float* w;
  const float* bias;
  int B, relu;
  RowSkip skip;        // down_1 of the analysis only (see Vrn32Args)
  // training step (reverse of the OTHER resampler through this kernel): y = mask > 0 ? result : 0, mask laid out like y
  const float* mask = nullptr;
  SegRead seg;         // down_1 behind the segment form of the 64^3 blocks (SEGV): x has slots nobody wrote
};

// XNHWC: the 32^3 input is NDHWC [b][d][h][w][32] (the training step's 32^3 tensors) instead of Q4; the 64^3 output (and the
// optional mask) is Q4 either way
template <int LD, int NCO, bool XNHWC = false>
__global__ void __launch_bounds__(256, 2) up2_row_kernel(UpRowArgs a) {
  constexpr int NG = 4 / NCO;                               // cout groups
  constexpr int CHT = 16 * NCO;                             // floats per tap of a (group, quad) chunk: [ci4][4 * NCO couts]
  constexpr int CH = 27 * CHT;
  constexpr int NW = (CH + 63) / 64;
  __shared__ __attribute__((aligned(16))) float wl[8 * CH + 64];
  // a workgroup works on ONE cout group: its four waves are four row pairs, and only that group's part of the filter's
  // LDS image (row_image_kernel, made once per net: [group][channel quad][tap][ci4][couts]) is copied in
  int wg = blockIdx.x;
  const int g = wg % NG; wg /= NG;
  stage_image<8 * CH>(wl, a.w + (size_t)g * 8 * CH);
  if (threadIdx.x < 64) wl[8 * CH + threadIdx.x] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const bool hi = lane >= 32;
  const float l32 = lane_masks32(lane).m32;
  int wv = __builtin_amdgcn_readfirstlane(wg * 4 + (threadIdx.x >> 6));
  const int k = wv % (kW / 2); wv /= (kW / 2);
  const int d0 = (wv % (kW / LD)) * LD; wv /= (kW / LD);
  const int b = wv;
  if (b >= a.B) return;
  f32x4 bi[NCO];
#pragma unroll
  for (int c = 0; c < NCO; ++c) {
    bi[c] = f32x4{0.f, 0.f, 0.f, 0.f};                        // a layer without bias (down_1) passes nullptr
    if (a.bias) bi[c] = f32x4{a.bias[(g * NCO + c) * 4], a.bias[(g * NCO + c) * 4 + 1], a.bias[(g * NCO + c) * 4 + 2], a.bias[(g * NCO + c) * 4 + 3]};
  }
  // acc[set][oh parity][ow parity][cout quad]; set 0 = output plane 2p (kd = 0 here, kd = 2 carried in), 1 = plane
  // 2p + 1 (kd = 1), 2 = plane 2p + 2 (kd = 2 here, carried to the next input plane)
  f32x4 acc[3][2][2][NCO];
#pragma unroll
  for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pw = 0; pw < 2; ++pw)
#pragma unroll
        for (int c = 0; c < NCO; ++c) acc[s_][ph][pw][c] = bi[c];
  const i32x4 rs = make_rsrc(a.x + (size_t)b * kW * kW * kW * 32, kW * kW * kW * 32 * 4);
  const i32x4 ro = make_rsrc(a.y + (size_t)b * 64 * 64 * 64 * 16, 64 * 64 * 64 * 16 * 4);
  const i32x4 rm = a.mask ? make_rsrc(a.mask + (size_t)b * 64 * 64 * 64 * 16, 64 * 64 * 64 * 16 * 4) : ro;
  const int lane_off = lane_off32<8, XNHWC>(lane);
  // output row of this lane's half: oh = 4k + 2 * hi (+ parity); its even / odd voxel pair starts at ow = 2i
  const int out_lane = ((4 * k + 2 * (lane >> 5)) * 4 + g * NCO) * (64 * 16) + (lane & 31) * 32;
  f32x4 PA, OA, PB, OB;
  auto load = [&](f32x4& P, f32x4& O, int p, int q) {
    P = load_pair<8, XNHWC>(rs, lane_off, hi, p, q, 2 * k);     // rows ih     = (2k, 2k + 1)
    O = load_pair<8, XNHWC>(rs, lane_off, hi, p, q, 2 * k - 1); // rows ih - 1 = (2k - 1, 2k)
  };
  // Accumulator sets by ROLE, not by register: set R0 = output plane 2p (kd = 0 here, kd = 2 carried in), R1 = plane 2p + 1
  // (kd = 1), R2 = plane 2p + 2 (kd = 2 here, carried to the next input plane).  The plane loop is unrolled three times with
  // the roles rotating over the three register sets and a fresh set's first MFMA takes the bias as its C operand (before:
  // acc[0] = acc[2]; acc[1] = acc[2] = bias behind the stores of every step).  81.6 -> 79.7 us per 8 cubes; the ablations in
  // profiles/r05_vE_up2_ablation.txt (no stores 71.5, no moves / resets 70.8, neither 55.8) say the step's epilogue — 16
  // stores behind 128 ReLU operations on MFMA results — is where this kernel's other 25 % are.
  auto quad = [&](const f32x4& P, const f32x4& O, int q, bool v0, bool v1, bool v2, auto R0_, auto R1_, auto R2_, auto FRESH_) {
    constexpr int R[3] = {decltype(R0_)::value, decltype(R1_)::value, decltype(R2_)::value};
    constexpr bool FRESH = decltype(FRESH_)::value;          // the step's first quad: sets R1 / R2 are born here
    float W[NW];
#pragma unroll
    for (int v = 0; v < NW; ++v) W[v] = wl[q * CH + v * 64 + lane];
    float x0[4], x1[4], r0[4], r1[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { x0[c] = comp(P, c); x1[c] = comp(O, c); r0[c] = shr1p(x0[c], l32); r1[c] = shr1p(x1[c], l32); }
    const bool vj[3] = {v0, v1, v2};
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {                        // validity tests hoisted: one branch per (quad, output plane), see pair_quad
      const int kd = s_;
      if (vj[s_]) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int co = 0; co < NCO; ++co) {
            auto mf_ = [&](int kh, int kw, float xv, f32x4& d, bool first) {
              const int t = (kd * 3 + kh) * 3 + kw, fo = t * CHT + c * 4 * NCO + co * 4;
              if (FRESH && s_ > 0 && c == 0 && first) d = mfa_new((fo & 63) >> 2, W[fo >> 6], xv, bi[co]);
              else d = mfa((fo & 63) >> 2, W[fo >> 6], xv, d);
            };
            f32x4 (&A)[2][2][NCO] = acc[R[s_]];
            mf_(0, 0, x0[c], A[0][0][co], true);             // even row (kh = 0 from ih, kh = 2 from ih - 1), even voxel
            mf_(0, 2, r0[c], A[0][0][co], false);
            mf_(2, 0, x1[c], A[0][0][co], false);
            mf_(2, 2, r1[c], A[0][0][co], false);
            mf_(0, 1, x0[c], A[0][1][co], true);             // even row, odd voxel (kw = 1)
            mf_(2, 1, x1[c], A[0][1][co], false);
            mf_(1, 0, x0[c], A[1][0][co], true);             // odd row (kh = 1), even voxel
            mf_(1, 2, r0[c], A[1][0][co], false);
            mf_(1, 1, x0[c], A[1][1][co], true);             // odd row, odd voxel
          }
      }
    }
  };
  auto store_plane = [&](const f32x4 (&A)[2][2][NCO], int od, bool ok) {
    const int base = ok ? od * (64 * 4 * 64 * 16) + out_lane : kOOB;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int co = 0; co < NCO; ++co)
#pragma unroll
        for (int pw = 0; pw < 2; ++pw) {
          f32x4 v = A[ph][pw][co];
          if (a.relu) v = relu4(v);
          if (a.mask) {                                       // wave-uniform
            const f32x4 m = raw_load4(rm, base + (ph * 4 + co) * (64 * 16) + pw * 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = m[i] > 0.f ? v[i] : 0.f;
          }
          raw_store4(v, ro, base + (ph * 4 + co) * (64 * 16) + pw * 16, 0, 0);
        }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  // one input plane: every set holds the bias at the first step of the tile (nothing was born before it), later steps' fresh
  // sets are born by their first MFMA; a set whose step is skipped (v false) is either never stored or still holds the bias
  auto step = [&](int p, auto R0_, auto R1_, auto R2_) {
    const bool pin = p >= 0;
    const bool v0 = p >= d0, v1 = v0, v2 = pin && p + 1 < d0 + LD;
    quad(PA, OA, 0, v0, v1, v2, R0_, R1_, R2_, std::true_type{});
    load(PA, OA, p, 2);
    quad(PB, OB, 1, v0, v1, v2, R0_, R1_, R2_, std::false_type{});
#pragma unroll 1
    for (int q = 2; q < 8; q += 2) {
      load(PB, OB, p, q + 1);
      quad(PA, OA, q, v0, v1, v2, R0_, R1_, R2_, std::false_type{});
      if (q + 2 < 8) load(PA, OA, p, q + 2); else load(PA, OA, p + 1, 0);
      quad(PB, OB, q + 1, v0, v1, v2, R0_, R1_, R2_, std::false_type{});
    }
    // BOTH buffers of the next plane are requested before this plane's stores: vmcnt counts in order, so a wait for a load
    // issued behind the 16 stores would be a wait for the stores' acknowledgement too
    load(PB, OB, p + 1, 1);
    store_plane(acc[decltype(R0_)::value], 2 * p, v0);
    store_plane(acc[decltype(R1_)::value], 2 * p + 1, v0);
  };
  load(PA, OA, d0 - 1, 0);
  load(PB, OB, d0 - 1, 1);
#pragma unroll 1
  for (int p = d0 - 1; p < d0 + LD; p += 3) {               // roles rotate instead of registers: (R0, R1, R2) -> (R2, R0, R1)
    step(p, I0{}, I1{}, I2{});
    if (p + 1 >= d0 + LD) break;
    step(p + 1, I2{}, I0{}, I1{});
    if (p + 2 >= d0 + LD) break;
    step(p + 2, I1{}, I2{}, I0{});
  }
}

// LDS images of the two layers' filters (kind 0 = up_2 for up2_row_kernel<., 2>, kind 1 = down_1 for
// down1_row_kernel<., 8>), built once per net from the TF layouts: [channel quad][cout group][tap][ci4][4 * NCO couts]
constexpr int kRowImageFloats = 27 * 16 * 32;
__global__ void __launch_bounds__(256) row_image_kernel(const float* w, float* dst, int kind) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kRowImageFloats) return;
  const int NCO = kind == 0 ? 2 : 8, NG = kind == 0 ? 2 : 1;
  const int CHT = 16 * NCO, CH = 27 * CHT;
  const int qg = i / CH, f = i - qg * CH;
  const int q = kind == 0 ? qg % 8 : qg / NG, g = kind == 0 ? qg / 8 : qg % NG;     // up_2: group-major (one group per workgroup)
  const int tap = f / CHT, r = f - tap * CHT, c = r / (4 * NCO), co = r % (4 * NCO);
  dst[i] = kind == 0 ? w[(tap * 16 + g * 4 * NCO + co) * 32 + 4 * q + c]      // Conv3DTranspose [27][Cout = 16][Cin = 32]
                     : w[(tap * 16 + 4 * q + c) * 32 + g * 4 * NCO + co];     // Conv3D          [27][Cin = 16][Cout = 32]
}
size_t row_image_floats(int cin, int cout, int k, int mode) {
  return (k == 3 && ((mode == 2 && cin == 32 && cout == 16) || (mode == 1 && cin == 16 && cout == 32))) ? kRowImageFloats : 0;
}
int launch_row_image(const float* w_tf, float* dst, int mode, hipStream_t s) {
  hipLaunchKernelGGL(row_image_kernel, dim3((kRowImageFloats + 255) / 256), dim3(256), 0, s, w_tf, dst, mode == 2 ? 0 : 1);
  return launch_ok("row_image_kernel");
}
// several images in one launch (the training plan rebuilds four per step: up_2 and down_1, each with its adjoint)
__global__ void __launch_bounds__(256) row_images_kernel(RowImageJobs jobs) {
  const int j = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kRowImageFloats) return;
  const int kind = jobs.kind[j];
  const float* w = jobs.w[j];
  const int NCO = kind == 0 ? 2 : 8, NG = kind == 0 ? 2 : 1;
  const int CHT = 16 * NCO, CH = 27 * CHT;
  const int qg = i / CH, f = i - qg * CH;
  const int q = kind == 0 ? qg % 8 : qg / NG, g = kind == 0 ? qg / 8 : qg % NG;
  const int tap = f / CHT, r = f - tap * CHT, c = r / (4 * NCO), co = r % (4 * NCO);
  jobs.dst[j][i] = kind == 0 ? w[(tap * 16 + g * 4 * NCO + co) * 32 + 4 * q + c] : w[(tap * 16 + 4 * q + c) * 32 + g * 4 * NCO + co];
}
int launch_row_images(const RowImageJobs& jobs, hipStream_t s) {
  if (jobs.n <= 0) return 0;
  hipLaunchKernelGGL(row_images_kernel, dim3((kRowImageFloats + 255) / 256, jobs.n), dim3(256), 0, s, jobs);
  return launch_ok("row_images_kernel");
}

int launch_up2_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s, bool x_nhwc, const float* mask) {
  UpRowArgs a{x, y, w, bias, B, relu};
  a.mask = mask;
  // 4 input planes x 2 cout quads per wave: 2048 waves per 8 cubes (measured per 8 cubes: <8,2> 109 us, <4,2> 94 us,
  // <8,1> 104 us, <4,1> 115 us, <16,1> 132 us; tconv_mfma_kernel 123 us)
  constexpr int LD = 4, NCO = 2;
  const int blocks = B * (kW / LD) * (kW / 2) / 4 * (4 / NCO);          // 4 row pairs per workgroup, one cout group each
  if (x_nhwc) hipLaunchKernelGGL((up2_row_kernel<LD, NCO, true>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((up2_row_kernel<LD, NCO>), dim3(blocks), dim3(256), 0, s, a);
  return launch_ok("up2_row_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// down_1: stride-2 conv 3^3, 16 -> 32 channels, 64^3 -> 32^3 (models/model_voxception.py:96-101), + ReLU.
//   y[o] = bias + sum_k x[2o + k] W[k], k = 0..2 per axis ('same' padding of an even input: nothing in front, one zero
//   behind).
// Lane = OUTPUT voxel of a pair vector (output rows 2k, 2k+1): the even and the odd voxels of an input row are two
// strided loads (E = x[2o], O = x[2o+1]: kw = 0 / 1), kw = 2 is E shifted by one lane.  Output row oh reads input rows
// 2oh + kh; output plane j reads input planes 2j, 2j+1, 2j+2, and 2j+2 is also plane j+1's kd = 0: the wave slides
// along d with two accumulator sets.  NCO output-channel quads per wave, weights per (channel quad, cout group) in LDS.
// x Q4 [B][64][64][4][64][4], y Q4 [B][32][32][8][32][4], w = the filter's LDS image (row_image_kernel, kind 1).
// ---------------------------------------------------------------------------------------------------------------
// YNHWC: the 32^3 output (and the optional mask) is NDHWC [b][d][h][w][32] (the training step's 32^3 tensors) instead of Q4
// SEGV: x was written slot by slot (vrn_seg.hip: 8 planes x 2 rows x 16 voxels); a lane whose voxel lies in a slot nobody wrote
// (a.seg.virt) reads the producer's empty-cube response instead — same window, other offset
template <int LD, int NCO, bool YNHWC = false, bool SEGV = false>
__global__ void __launch_bounds__(256, 2) down1_row_kernel(UpRowArgs a) {
  constexpr int NG = 8 / NCO;
  constexpr int CHT = 16 * NCO;                             // floats per tap of a (quad, group) chunk: [ci4][4 * NCO couts]
  constexpr int CH = 27 * CHT;
  constexpr int NWK = (9 * CHT + 63) / 64;                  // weight registers of one kd slice
  static_assert((9 * CHT) % 64 == 0, "a kd slice must start on a register boundary");
  __shared__ __attribute__((aligned(16))) float wl[4 * NG * CH];
  const bool no_work = workgroup_all_empty(a.skip);
  if (!no_work) {
    stage_image<4 * NG * CH>(wl, a.w);                      // a.w = the LDS image
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const bool hi = lane >= 32;
  const float l31 = lane_masks32(lane).m31;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  int wv = wid;
  bool heavy = true;
  if (a.skip.order) {                                       // NG == 1 there: one wave per (row pair, LD planes) tile
    if (wid >= a.B * (kW / LD) * (kW / 2) * NG) return;
    wv = __builtin_amdgcn_readfirstlane((int)(a.skip.order[wid] & ~kTileUnread));
    heavy = wid < (int)*a.skip.n_heavy;
    if (!heavy && a.skip.counter && lane == 0) atomicAdd(a.skip.counter, 1u);
  }
  const int g = wv % NG; wv /= NG;
  const int k = wv % (kW / 2); wv /= (kW / 2);
  const int d0 = (wv % (kW / LD)) * LD; wv /= (kW / LD);
  const int b = wv;
  if (b >= a.B) return;
  if (!heavy) {
    Tile32 tl;
    tl.b = b; tl.k0 = k; tl.d0 = d0;
    copy_empty_tile32<1, LD, 8>(a.skip.empty, a.y + (size_t)b * kW * kW * kW * 32, tl, lane);
    return;
  }
  f32x4 bi[NCO];
#pragma unroll
  for (int c = 0; c < NCO; ++c) {
    bi[c] = f32x4{0.f, 0.f, 0.f, 0.f};                        // a layer without bias (down_1) passes nullptr
    if (a.bias) bi[c] = f32x4{a.bias[(g * NCO + c) * 4], a.bias[(g * NCO + c) * 4 + 1], a.bias[(g * NCO + c) * 4 + 2], a.bias[(g * NCO + c) * 4 + 3]};
  }
  f32x4 cur[NCO], nxt[NCO];
#pragma unroll
  for (int c = 0; c < NCO; ++c) { cur[c] = bi[c]; nxt[c] = bi[c]; }
  const i32x4 rs = SEGV ? make_rsrc(a.seg.win, (unsigned)kOOB) : make_rsrc(a.x + (size_t)b * 64 * 64 * 64 * 16, 64 * 64 * 64 * 16 * 4);
  const i32x4 ro = make_rsrc(a.y + (size_t)b * kW * kW * kW * 32, kW * kW * kW * 32 * 4);
  // input: voxel 2o (+1) of row 4k + 2 * hi + kh, 4 quads of 64 x 16 B per row
  const int in_lane = (lane >> 5) * (2 * 4 * 1024) + (lane & 31) * 32;
  // SEGV: the lane's offsets in the tensor / in the empty-cube response, and which of its (plane tile, row tile) pairs are not
  // written at its segment: the wave's planes 2 d0 .. 2 d0 + 2 LD lie in plane tile dtA = d0 / 4 or the next, the lane's rows
  // 4k + 2 hi + kh in row tile 2k + hi (kh < 2) or the next (kh = 2); bit (plane tile != dtA) * 2 + (kh == 2)
  unsigned offT = 0, offE = 0, vb = 0;
  const int dtA = (2 * d0) >> 3;
  if constexpr (SEGV) {
    offT = a.seg.x_off + (unsigned)b * (64u * 64 * 64 * 16 * 4) + (unsigned)in_lane;
    offE = a.seg.e_off + (unsigned)in_lane;
    const int sg = (lane & 31) >> 3, htA = 2 * k + (lane >> 5);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int dt = dtA + (i >> 1), ht = htA + (i & 1);
      dt = dt > 7 ? 7 : dt; ht = ht > 31 ? 31 : ht;            // a clamped entry belongs to planes / rows outside the cube: never used
      vb |= ((a.seg.virt[(size_t)b * 256 + dt * 32 + ht] >> sg) & 1u) << i;
    }
  }
  // output voxel (plane j, row 2k + hi, w = lane & 31), channel quad g * NCO + co
  const int out_lane = YNHWC ? ((2 * k + (lane >> 5)) * kW + (lane & 31)) * (8 * 16) + g * NCO * 16
                             : ((2 * k + (lane >> 5)) * 8 + g * NCO) * kRowQ + (lane & 31) * 16;
  constexpr int kOutPlane = kW * 8 * kRowQ, kOutQuad = YNHWC ? 16 : kRowQ;     // (a plane has the same bytes in both layouts)
  const i32x4 rm = a.mask ? make_rsrc(a.mask + (size_t)b * kW * kW * kW * 32, kW * kW * kW * 32 * 4) : ro;
  struct Rows { f32x4 e[3], o[3]; };
  auto load = [&](Rows& R, int p, int q) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = 4 * k + kh;                             // + 2 for the upper half: only that row can leave the cube (ih = 64)
      const bool ok = (unsigned)p < 64u && (ih + (hi ? 2 : 0)) < 64;
      int off;
      if constexpr (SEGV) {
        const unsigned bit = (vb >> (((p >> 3) != dtA ? 2 : 0) + (kh == 2 ? 1 : 0))) & 1u;
        off = ok ? ((p * 64 + ih) * 4 + q) * 1024 + (int)(bit ? offE : offT) : kOOB;
      } else off = ok ? ((p * 64 + ih) * 4 + q) * 1024 + in_lane : kOOB;
      R.e[kh] = raw_load4(rs, off, 0, 0);
      R.o[kh] = raw_load4(rs, off + 16, 0, 0);
    }
  };
  // one channel quad of one input plane: kd = KA into accA (and kd = KB into accB when KB >= 0)
  auto quad = [&](const Rows& R, int q, int KA, f32x4 (&accA)[NCO], int KB, f32x4 (&accB)[NCO], bool vA, bool vB) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int kd = pass == 0 ? KA : KB;
      const bool v = pass == 0 ? vA : vB;
      if (kd < 0 || !v) continue;
      float W[NWK];
#pragma unroll
      for (int vv = 0; vv < NWK; ++vv) W[vv] = wl[(q * NG + g) * CH + kd * 9 * CHT + vv * 64 + lane];
      f32x4 (&acc)[NCO] = pass == 0 ? accA : accB;
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const float xe = comp(R.e[kh], c), xo = comp(R.o[kh], c), x2 = shl1p(xe, l31);
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
  // prologue: input plane 2 d0 is kd = 0 of the first output plane
  load(RA, 2 * d0, 0);
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    if (q + 1 < 4) load(RB, 2 * d0, q + 1); else load(RB, 2 * d0 + 1, 0);
    quad(RA, q, 0, cur, -1, nxt, true, false);
    RA = RB;
  }
#pragma unroll 1
  for (int j = d0; j < d0 + LD; ++j) {
    // plane 2j + 1: kd = 1 of output plane j
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      if (q + 1 < 4) load(RB, 2 * j + 1, q + 1); else load(RB, 2 * j + 2, 0);
      quad(RA, q, 1, cur, -1, nxt, true, false);
      RA = RB;
    }
    // plane 2j + 2: kd = 2 of output plane j and kd = 0 of plane j + 1 (of this wave's segment)
    const bool more = j + 1 < d0 + LD;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      if (q + 1 < 4) load(RB, 2 * j + 2, q + 1); else load(RB, 2 * j + 3, 0);
      quad(RA, q, 2, cur, 0, nxt, true, more);
      RA = RB;
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      f32x4 v = cur[co];
      if (a.relu) v = relu4(v);
      if (a.mask) {                                           // wave-uniform
        const f32x4 m = raw_load4(rm, j * kOutPlane + out_lane + co * kOutQuad, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = m[i] > 0.f ? v[i] : 0.f;
      }
      raw_store4(v, ro, j * kOutPlane + out_lane + co * kOutQuad, 0, 0);
      cur[co] = nxt[co];
      nxt[co] = bi[co];
    }
  }
}

int launch_down1_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s, const RowSkip* skip,
                     bool y_nhwc, const float* mask, const SegRead* seg) {
  UpRowArgs a{x, y, w, bias, B, relu};
  if (skip) a.skip = *skip;
  a.mask = mask;
  if (seg) a.seg = *seg;
  static_assert(kDown1TileRows == 2 && kDown1TilePlanes == 2, "tile orders for down_1 are built for 1 row pair x 2 planes");
  // 2 output planes x all 8 cout quads per wave: 2048 waves per 8 cubes (measured per 8 cubes: <2,8> 75 us, <4,8> 81 us,
  // <4,4> 83 us, <8,4> 90 us, <2,4> 93 us; conv_mfma_kernel 106 us)
  constexpr int LD = 2, NCO = 8;
  const int waves = B * (kW / LD) * (kW / 2) * (8 / NCO);
  if (seg && !y_nhwc) hipLaunchKernelGGL((down1_row_kernel<LD, NCO, false, true>), dim3((waves + 3) / 4), dim3(256), 0, s, a);
  else if (seg) { set_error("launch_down1_row: a slot-wise input with an NDHWC output is not built"); return -1; }
  else if (y_nhwc) hipLaunchKernelGGL((down1_row_kernel<LD, NCO, true>), dim3((waves + 3) / 4), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((down1_row_kernel<LD, NCO>), dim3((waves + 3) / 4), dim3(256), 0, s, a);
  return launch_ok("down1_row_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// Reverse pass of the C = 32 block at 32^3 (training step; NDHWC tensors; train_hyper.py:_vrn_bwd) on pair vectors — the
// 64^3 stage's vrn16a_bwd / vrn16bc_bwd_row_kernel (vrn_row.hip) in this stage's geometry, instead of five bwd-data
// launches of the generic implicit-GEMM kernel per block (0.9 ms of a 10 ms step: profiles/r04_vD_train_kernel_stats.csv).
// A transposed stride-1 convolution is a convolution with mirrored taps and swapped channel roles:
//   dx[ci](v) = sum_t K[t][co][ci] g[co](v + off(t)),  K[t] = W[26 - t]^T
// so the forward kernels' pair_channel machinery applies unchanged to LDS images of K.
//   vrn32a_bwd : dx = [x > 0] * ( dpre + conv1_1^T(dt11) (3^3, 8 -> 32) + conv2_1^T(dt21) (1^3, 8 -> 32) ); dx may alias dpre
//   vrn32bc_bwd: dt11 = [t11 > 0] * conv1_2^T(dz12) (3^3, 16 -> 8);  dt22 = [t22 > 0] * conv2_3^T(dz23) (1^3, 16 -> 8), made
//                on the fly for the three pair vectors conv2_2^T reads and written for the wave's own rows;
//                dt21 = [t21 > 0] * conv2_2^T(dt22) (3^3, 8 -> 8)
// Summation order per output: (plane, channel, kh, kw), then the 1^3 layer's channels, then + dpre: fixed.
// ---------------------------------------------------------------------------------------------------------------
struct Vrn32BwdInArgs {
  const float *dt11, *dt21, *dpre, *x;   // x = nullptr: no mask
  const float *w11, *w21;                // TensorFlow layouts [27][32][8], [1][32][8]
  float* dx;
  int B;
};

template <int LD, bool MASK>
__global__ void __launch_bounds__(256, 2) vrn32a_bwd_row_kernel(Vrn32BwdInArgs a) {
  constexpr int C11 = 27 * 64;                              // floats per (g quad, dx half) chunk [tap][g4][16 dx channels]
  __shared__ float wl[4 * C11 + 4 * 64];
  stage_indexed<4 * C11>(wl, [&](int i) {
    const int qh = i / C11, f = i - qh * C11, q = qh >> 1, h = qh & 1;
    return a.w11[((26 - (f >> 6)) * 32 + 16 * h + (f & 15)) * 8 + 4 * q + ((f >> 4) & 3)];
  });
  stage_indexed<4 * 64>(wl + 4 * C11, [&](int i) {          // conv2_1^T: [g quad][dx half][g4][16]
    const int qh = i >> 6, f = i & 63, q = qh >> 1, h = qh & 1;
    return a.w21[(16 * h + (f & 15)) * 8 + 4 * q + (f >> 4)];
  });
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const bool hi = lane >= 32;
  const LaneMasks32 lm = lane_masks32(lane);
  const float l32 = lm.m32, l31 = lm.m31;
  const Tile32 tl = wave_tile32<1, LD>();
  if (tl.b >= a.B) return;
  const int k0 = tl.k0, d0 = tl.d0;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[2][3][1][4];                                    // [dx half][plane set][pair][dx quad of the half]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[h][j][0][q] = zero;
  constexpr int V8 = kW * kW * kW * 8, V32 = kW * kW * kW * 32;
  const i32x4 rg = make_rsrc(a.dt11 + (size_t)tl.b * V8, V8 * 4);
  const i32x4 rg2 = make_rsrc(a.dt21 + (size_t)tl.b * V8, V8 * 4);
  const i32x4 rp = make_rsrc(a.dpre + (size_t)tl.b * V32, V32 * 4);
  const i32x4 rx = MASK ? make_rsrc(a.x + (size_t)tl.b * V32, V32 * 4) : rp;
  const i32x4 ro = make_rsrc(a.dx + (size_t)tl.b * V32, V32 * 4);
  const int lane_g = lane_off32<2, true>(lane), lane_x = lane_off32<8, true>(lane);
  f32x4 PA[1], OA[2], PB[1], OB[2];
  auto load = [&](f32x4 (&P)[1], f32x4 (&O)[2], int p, int q) {
    P[0] = load_pair<2, true>(rg, lane_g, hi, p, q, 2 * k0);
    O[0] = load_pair<2, true>(rg, lane_g, hi, p, q, 2 * k0 - 1);
    O[1] = load_pair<2, true>(rg, lane_g, hi, p, q, 2 * k0 + 1);
  };
  auto quad = [&](const f32x4 (&P)[1], const f32x4 (&O)[2], int q, bool v0, bool v1, bool v2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float W[27];
#pragma unroll
      for (int v = 0; v < 27; ++v) W[v] = wl[(q * 2 + h) * C11 + v * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) pair_channel<1, 4, 27>(acc[h], W, c, P, O, v0, v1, v2, l32, l31, Map16());
    }
  };
  load(PA, OA, d0 - 1, 0);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kW;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
    const bool done = p - 1 >= d0;
    // what the finished plane p - 1 needs besides its sums, requested before the MFMAs of this step
    const int obase = done ? row_base32<8, true>(p - 1, 2 * k0, 0) + lane_x : kOOB;
    const int gbase = done ? row_base32<2, true>(p - 1, 2 * k0, 0) + lane_g : kOOB;
    f32x4 g2[2], res[8], xs[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) g2[q] = raw_load4(rg2, gbase + q * 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      res[q] = raw_load4(rp, obase + q * 16, 0, 0);
      if constexpr (MASK) xs[q] = raw_load4(rx, obase + q * 16, 0, 0);
    }
    load(PB, OB, p, 1);
    quad(PA, OA, 0, v0, v1, v2);
    load(PA, OA, p + 1, 0);
    quad(PB, OB, 1, v0, v1, v2);
    // output plane p - 1: the 1^3 layer's part, the skip gradient, the mask, store
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float W1 = wl[4 * C11 + (q * 2 + h) * 64 + lane];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int oq = 0; oq < 4; ++oq) acc[h][0][0][oq] = mfa(c * 4 + oq, W1, comp(g2[q], c), acc[h][0][0][oq]);
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int oq = 0; oq < 4; ++oq) {
        f32x4 y = acc[h][0][0][oq] + res[4 * h + oq];
        if constexpr (MASK) {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = xs[4 * h + oq][i] > 0.f ? y[i] : 0.f;
        }
        raw_store4(y, ro, obase + (4 * h + oq) * 16, 0, 0);
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int oq = 0; oq < 4; ++oq) { acc[h][0][0][oq] = acc[h][1][0][oq]; acc[h][1][0][oq] = acc[h][2][0][oq]; acc[h][2][0][oq] = zero; }
  }
}

struct Vrn32BwdTailArgs {
  const float *dz12, *dz23, *t11, *t21, *t22;   // dz [..][16], t [..][8], NDHWC
  const float *w12, *w22, *w23;                 // TensorFlow layouts [27][8][16], [27][8][8], [1][8][16]
  float *dt11, *dt21, *dt22;
  int B;
};

template <int LD>
__global__ void __launch_bounds__(256, 2) vrn32bc_bwd_row_kernel(Vrn32BwdTailArgs a) {
  constexpr int CK = 896;                                   // floats per in-quad chunk [tap][c4][8 outs] (864 used)
  __shared__ float wl[4 * CK + 2 * CK];
  stage_indexed<4 * CK>(wl, [&](int i) {                    // K12[t][dz12 channel][t11 channel] = w12[26 - t][t11 ch][dz12 ch]
    const int q = i / CK, f = i - q * CK;
    return f < 864 ? a.w12[((26 - (f >> 5)) * 8 + (f & 7)) * 16 + 4 * q + ((f >> 3) & 3)] : 0.f;
  });
  stage_indexed<2 * CK>(wl + 4 * CK, [&](int i) {           // K22[t][dt22 channel][dt21 channel] = w22[26 - t][dt21 ch][dt22 ch]
    const int q = i / CK, f = i - q * CK;
    return f < 864 ? a.w22[((26 - (f >> 5)) * 8 + (f & 7)) * 8 + 4 * q + ((f >> 3) & 3)] : 0.f;
  });
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const bool hi = lane >= 32;
  const LaneMasks32 lm = lane_masks32(lane);
  const float l32 = lm.m32, l31 = lm.m31;
  const Tile32 tl = wave_tile32<1, LD>();
  if (tl.b >= a.B) return;
  const int k0 = tl.k0, d0 = tl.d0;
  // K23[dz23 channel 16][dt22 channel 8] = w23[dt22 ch][dz23 ch]: register c >> 3 holds [c & 7][8], abid = (c & 7) * 2 + out quad
  const float W23[2] = {a.w23[(lane & 7) * 16 + (lane >> 3)], a.w23[(lane & 7) * 16 + 8 + (lane >> 3)]};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc12[3][1][2], acc22[3][1][2];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 2; ++q) { acc12[j][0][q] = zero; acc22[j][0][q] = zero; }
  constexpr int V8 = kW * kW * kW * 8, V16 = kW * kW * kW * 16;
  const i32x4 r12 = make_rsrc(a.dz12 + (size_t)tl.b * V16, V16 * 4);
  const i32x4 r23 = make_rsrc(a.dz23 + (size_t)tl.b * V16, V16 * 4);
  const i32x4 rt11 = make_rsrc(a.t11 + (size_t)tl.b * V8, V8 * 4);
  const i32x4 rt21 = make_rsrc(a.t21 + (size_t)tl.b * V8, V8 * 4);
  const i32x4 rt22 = make_rsrc(a.t22 + (size_t)tl.b * V8, V8 * 4);
  const i32x4 ro11 = make_rsrc(a.dt11 + (size_t)tl.b * V8, V8 * 4);
  const i32x4 ro21 = make_rsrc(a.dt21 + (size_t)tl.b * V8, V8 * 4);
  const i32x4 ro22 = make_rsrc(a.dt22 + (size_t)tl.b * V8, V8 * 4);
  const int lane_z = lane_off32<4, true>(lane), lane_t = lane_off32<2, true>(lane);
  // the three pair vectors of a plane: 0 = aligned (rows 2k, 2k+1), 1 = rows (2k-1, 2k), 2 = rows (2k+1, 2k+2)
  auto vrow = [&](int v) { return v == 0 ? 2 * k0 : (v == 1 ? 2 * k0 - 1 : 2 * k0 + 1); };
  f32x4 PA[1], OA[2], PB[1], OB[2];
  auto load12 = [&](f32x4 (&P)[1], f32x4 (&O)[2], int p, int q) {
    P[0] = load_pair<4, true>(r12, lane_z, hi, p, q, vrow(0));
    O[0] = load_pair<4, true>(r12, lane_z, hi, p, q, vrow(1));
    O[1] = load_pair<4, true>(r12, lane_z, hi, p, q, vrow(2));
  };
  load12(PA, OA, d0 - 1, 0);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kW;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
    const bool done = p - 1 >= d0;
    // the finished plane's masks and this plane's dz23 / t22 vectors, requested before the MFMAs of this step
    const int tbase = done ? row_base32<2, true>(p - 1, 2 * k0, 0) + lane_t : kOOB;
    f32x4 k11[2], k21[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { k11[q] = raw_load4(rt11, tbase + q * 16, 0, 0); k21[q] = raw_load4(rt21, tbase + q * 16, 0, 0); }
    f32x4 in23[3][4], m22[3][2];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
#pragma unroll
      for (int q = 0; q < 4; ++q) in23[v][q] = load_pair<4, true>(r23, lane_z, hi, p, q, vrow(v));
#pragma unroll
      for (int q = 0; q < 2; ++q) m22[v][q] = load_pair<2, true>(rt22, lane_t, hi, p, q, vrow(v));
    }
    // conv1_2^T: the four channel quads of dz12
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
      float W[14];
      load12(PB, OB, p, q + 1);
#pragma unroll
      for (int v = 0; v < 14; ++v) W[v] = wl[q * CK + v * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) pair_channel<1, 2, 14>(acc12, W, c, PA, OA, v0, v1, v2, l32, l31, Map8());
      if (q + 2 < 4) load12(PA, OA, p, q + 2); else load12(PA, OA, p + 1, 0);
#pragma unroll
      for (int v = 0; v < 14; ++v) W[v] = wl[(q + 1) * CK + v * 64 + lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) pair_channel<1, 2, 14>(acc12, W, c, PB, OB, v0, v1, v2, l32, l31, Map8());
    }
    // dt22 of this plane on the three pair vectors: the 1^3 layer's reverse, masked by t22 > 0
    f32x4 d22[3][2];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      f32x4 g[2] = {zero, zero};
#pragma unroll
      for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int oq = 0; oq < 2; ++oq) g[oq] = mfa((c & 7) * 2 + oq, W23[c >> 3], comp(in23[v][c >> 2], c & 3), g[oq]);
#pragma unroll
      for (int oq = 0; oq < 2; ++oq)
#pragma unroll
        for (int i = 0; i < 4; ++i) d22[v][oq][i] = m22[v][oq][i] > 0.f ? g[oq][i] : 0.f;
    }
    if (v1) {                                               // the wave's own rows of dt22 (conv2_2's weight gradient reads them)
      const int b22 = row_base32<2, true>(p, 2 * k0, 0) + lane_t;
      raw_store4(d22[0][0], ro22, b22, 0, 0);
      raw_store4(d22[0][1], ro22, b22 + 16, 0, 0);
    }
    // conv2_2^T on dt22
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float W[14];
#pragma unroll
      for (int v = 0; v < 14; ++v) W[v] = wl[(4 + q) * CK + v * 64 + lane];
      const f32x4 Pq[1] = {d22[0][q]};
      const f32x4 Oq[2] = {d22[1][q], d22[2][q]};
#pragma unroll
      for (int c = 0; c < 4; ++c) pair_channel<1, 2, 14>(acc22, W, c, Pq, Oq, v0, v1, v2, l32, l31, Map8());
    }
    // output plane p - 1: masks, stores
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 y11 = acc12[0][0][q], y21 = acc22[0][0][q];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y11[i] = k11[q][i] > 0.f ? y11[i] : 0.f;
        y21[i] = k21[q][i] > 0.f ? y21[i] : 0.f;
      }
      raw_store4(y11, ro11, tbase + q * 16, 0, 0);
      raw_store4(y21, ro21, tbase + q * 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      acc12[0][0][q] = acc12[1][0][q]; acc12[1][0][q] = acc12[2][0][q]; acc12[2][0][q] = zero;
      acc22[0][0][q] = acc22[1][0][q]; acc22[1][0][q] = acc22[2][0][q]; acc22[2][0][q] = zero;
    }
  }
}

// dx = [x > 0] * (dpre + conv1_1^T(dt11) + conv2_1^T(dt21)) of a C = 32 block at D = 32 (x = nullptr: no mask); NDHWC
int launch_vrn32_bwd_input(const float* dt11, const float* dt21, const float* dpre, const float* x, const float* w11, const float* w21,
                           float* dx, int B, hipStream_t s) {
  Vrn32BwdInArgs a{dt11, dt21, dpre, x, w11, w21, dx, B};
  constexpr int LD = 2;                                      // one row pair x 2 planes per wave: 256 waves per cube (the training batch is 8 cubes)
  const dim3 grid(B * (kW / 2) * (kW / LD) / 4);
  if (x) hipLaunchKernelGGL((vrn32a_bwd_row_kernel<LD, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((vrn32a_bwd_row_kernel<LD, false>), grid, dim3(256), 0, s, a);
  return launch_ok("vrn32a_bwd_row_kernel");
}
// dt11 / dt22 / dt21 of a C = 32 block at D = 32 from the block tail's reverse (vrn32bc_bwd_row_kernel); NDHWC
int launch_vrn32_bwd_tail(const float* dz12, const float* dz23, const float* t11, const float* t21, const float* t22, const float* w12,
                          const float* w22, const float* w23, float* dt11, float* dt21, float* dt22, int B, hipStream_t s) {
  Vrn32BwdTailArgs a{dz12, dz23, t11, t21, t22, w12, w22, w23, dt11, dt21, dt22, B};
  constexpr int LD = 2;
  hipLaunchKernelGGL((vrn32bc_bwd_row_kernel<LD>), dim3(B * (kW / 2) * (kW / LD) / 4), dim3(256), 0, s, a);
  return launch_ok("vrn32bc_bwd_row_kernel");
}

// The same block for the training step: NDHWC tensors, every intermediate the reverse pass needs is kept (Vrn32Args).
int launch_vrn32_row_train(const float* x, float* t11, float* t21, float* t22, float* pre, float* out, const float* const* w, int B,
                           hipStream_t s, int* pre_signs) {
  Vrn32Args a;
  a.x = x; a.t12 = t11; a.out = out; a.t21 = t21; a.t22 = t22; a.pre = pre; a.pre_signs = pre_signs;
  a.w11 = w[0]; a.b11 = w[1]; a.w12 = w[2]; a.b12 = w[3]; a.w21 = w[4]; a.b21 = w[5];
  a.w22 = w[6]; a.b22 = w[7]; a.w23 = w[8]; a.b23 = w[9];
  a.B = B;
  if (B <= 16) {          // the training batch (8 cubes): one row pair x 2 planes per wave = 256 waves per cube, enough to fill the chip
    hipLaunchKernelGGL((vrn32a_row_kernel<1, 2, true>), dim3(B * (kW / 2) * (kW / 2) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL((vrn32bc_row_kernel<2, true>), dim3(B * (kW / 2) * (kW / 2) / 4), dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((vrn32a_row_kernel<2, 4, true>), dim3(B * (kW / 4) * (kW / 4) / 4), dim3(256), 0, s, a);
    hipLaunchKernelGGL((vrn32bc_row_kernel<8, true>), dim3(B * (kW / 2) * (kW / 8) / 4), dim3(256), 0, s, a);
  }
  return launch_ok("vrn32 row kernels (training)");
}

// rows x planes of a wave tile, by launch size: what launch_vrn32_row uses (and what a tile order must be built for)
void vrn32_tile_geometry(int B, int which, int* th, int* ld) {
  if (B <= 16) { *th = 2; *ld = 2; }
  else if (which == 0) { *th = 4; *ld = 4; }
  else { *th = 2; *ld = 8; }
}

// the block's LDS images from the TensorFlow layouts (the kernels' own gather formulas): w = {w11,b11,w12,b12,w21,b21,w22,...}
__global__ void __launch_bounds__(256) vrn32_image_kernel(const float* w11, const float* w21, const float* w12, const float* w22, float* dst) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= kVrn32ImageFloats) return;
  if (i < kVrn32ImgA) {
    const int q = i / 896, f = i - q * 896;
    dst[i] = f < 864 ? w11[((f >> 5) * 32 + 4 * q + ((f >> 3) & 3)) * 8 + (f & 7)] : w21[(4 * q + ((f - 864) >> 3)) * 8 + (f & 7)];
  } else if (i < kVrn32ImgA + 2 * 27 * 64) {
    const int j = i - kVrn32ImgA, q = j / (27 * 64), f = j - q * (27 * 64);
    dst[i] = w12[((f >> 6) * 8 + 4 * q + ((f >> 4) & 3)) * 16 + (f & 15)];
  } else {
    const int j = i - kVrn32ImgA - 2 * 27 * 64, q = j / 896, f = j - q * 896;
    dst[i] = f < 864 ? w22[((f >> 5) * 8 + 4 * q + ((f >> 3) & 3)) * 8 + (f & 7)] : 0.f;
  }
}
size_t vrn32_image_floats() { return kVrn32ImageFloats; }
int launch_vrn32_image(const float* const* w, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(vrn32_image_kernel, dim3((kVrn32ImageFloats + 255) / 256), dim3(256), 0, s, w[0], w[4], w[2], w[6], dst);
  return launch_ok("vrn32_image_kernel");
}

int launch_vrn32_row(const float* x, float* t12, float* out, const float* const* w, int B, int which, hipStream_t s, bool x_nonneg,
                     const RowSkip* skip, const float* img) {
  Vrn32Args a;
  a.img = img;
  if (skip) a.skip = *skip;
  a.x = x; a.t12 = t12; a.out = out;
  a.w11 = w[0]; a.b11 = w[1]; a.w12 = w[2]; a.b12 = w[3]; a.w21 = w[4]; a.b21 = w[5];
  a.w22 = w[6]; a.b22 = w[7]; a.w23 = w[8]; a.b23 = w[9];
  a.B = B;
  // A: 2 row pairs x 4 planes per wave; BC: 1 row pair x 8 planes: 64 waves per cube each; small batches (the training
  // step's tiles): one row pair x 2 planes, 256 waves per cube — the same sums
  if (B <= 16) {
    if (which == 0) hipLaunchKernelGGL((vrn32a_row_kernel<1, 2>), dim3(B * (kW / 2) * (kW / 2) / 4), dim3(256), 0, s, a);
    else if (x_nonneg) hipLaunchKernelGGL((vrn32bc_row_kernel<2, false, true>), dim3(B * (kW / 2) * (kW / 2) / 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((vrn32bc_row_kernel<2>), dim3(B * (kW / 2) * (kW / 2) / 4), dim3(256), 0, s, a);
    return launch_ok("vrn32 row kernel");
  }
  // kernel A with the validity tests hoisted per quad (QJ) is 3-4 % faster on dense launches; on the analysis' skipping
  // launches of 64 cubes (2 660 heavy waves for 2 048 slots) the per-channel form ends its ragged second round earlier
  // (200 against 241 us, profiles/r05_vA_row_variants.txt) — same sums either way
  if (which == 0 && a.skip.order) hipLaunchKernelGGL((vrn32a_row_kernel<2, 4, false, false>), dim3(B * (kW / 4) * (kW / 4) / 4), dim3(256), 0, s, a);
  else if (which == 0) {
    // dense launches: (row pairs x planes) per wave by launch size, as for BC below — 64 cubes: 2 x 8 243.6 us, 2 x 4 248.6,
    // 1 x 4 254; a 39-cube remainder (2 496 waves of 2 x 4 for 2 048 slots: a ragged second round): 1 x 4 165.8 us, 2 x 2 174,
    // 2 x 4 196, 2 x 8 238 (profiles/r05_vH_tile_by_launch_size.txt); PCGC_A32_TILE=24 / 28 / 14 forces one
    const char* e = getenv("PCGC_A32_TILE");
    const int t = e ? atoi(e) : (B >= 56 ? 28 : 14);
    if (t == 28) hipLaunchKernelGGL((vrn32a_row_kernel<2, 8>), dim3(B * (kW / 4) * (kW / 8) / 4), dim3(256), 0, s, a);
    else if (t == 14) hipLaunchKernelGGL((vrn32a_row_kernel<1, 4>), dim3(B * (kW / 2) * (kW / 4) / 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((vrn32a_row_kernel<2, 4>), dim3(B * (kW / 4) * (kW / 4) / 4), dim3(256), 0, s, a);
  }
  else if (x_nonneg && !a.skip.order) {
    // dense launches (the synthesis): planes per wave by launch size — 64 cubes run 4 % faster as 2 048 waves of 16 planes than
    // as 4 096 of 8, a 39-cube remainder 12 % faster as 4 992 waves of 4 planes than as 2 496 of 8 (0.6 of the wave slots,
    // unevenly spread); same sums per output (PCGC_BC32_LD=4 / 8 / 16 forces one; read per launch: tools/exp/t_rows.py;
    // profiles/r05_vH_tile_by_launch_size.txt)
    const char* e = getenv("PCGC_BC32_LD");
    const int ld = e ? atoi(e) : (B >= 56 ? 16 : 4);
    if (ld == 16) hipLaunchKernelGGL((vrn32bc_row_kernel<16, false, true>), dim3(B * (kW / 2) * (kW / 16) / 4), dim3(256), 0, s, a);
    else if (ld == 4) hipLaunchKernelGGL((vrn32bc_row_kernel<4, false, true>), dim3(B * (kW / 2) * (kW / 4) / 4), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((vrn32bc_row_kernel<8, false, true>), dim3(B * (kW / 2) * (kW / 8) / 4), dim3(256), 0, s, a);
  }
  else if (x_nonneg) hipLaunchKernelGGL((vrn32bc_row_kernel<8, false, true>), dim3(B * (kW / 2) * (kW / 8) / 4), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((vrn32bc_row_kernel<8>), dim3(B * (kW / 2) * (kW / 8) / 4), dim3(256), 0, s, a);
  return launch_ok("vrn32 row kernel");
}

}  // namespace pcgc
