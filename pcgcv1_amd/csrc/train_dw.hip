// Weight gradient of the stride-1 3x3x3 convolutions (train_hyper.py:200-207, tf.GradientTape over the
// model_voxception.py layers) as an LDS-tiled VALU kernel.
//
//   dW[tap][ci][co] = sum over (cube, voxel v)  x[v + tap - 1][ci] * dz[v][co]
//
// The reduction runs over millions of voxels and the result is tiny (27*Cin*Cout floats), so the tiling is the
// transpose of the forward kernels': a workgroup walks 4 x 4 x 16-voxel tiles (x tile with halo and dz tile in
// LDS, each global byte read once per channel chunk instead of once per tap) and every thread keeps a TI x TJ
// register block of dW for one tap.  With T = 27 * (CIN/TI) * (COUT/TJ) such blocks per channel chunk,
//   T <  256: S = 256 / T threads share a block and split the voxels of each tile (fixed-order LDS reduction),
//   T >= 256: a thread owns ceil(T / 256) blocks.
// Workgroups are persistent (grid.x of them, each summing a fixed, strided set of tiles in a fixed order) and
// write one partial dW each; conv_dw_final_kernel (train.hip) adds the partials in index order.  No float
// atomics anywhere: the result is bit-reproducible for a given (shape, grid).
#include <cstdlib>
#include <vector>
#include "mfma_common.h"

namespace pcgc {

constexpr int kDwGroups = 512;     // persistent workgroups (x2 per CU) = partial sums per weight

// Several weight gradients of the same shape in one launch (grid.z = job): the 16^3 stage of the training step has 30 of
// them per step with 128-512 workgroups each, which leave most wave slots idle one at a time.  n = 0: the plain arguments.
struct DwBatch {
  int n = 0;
  const float* x[kDwBatchMax];
  const float* dz[kDwBatchMax];
  float* partial[kDwBatchMax];
};
// A thread that set a capture list (dw_capture_begin; train_plan.hip's deferred mode) has the batchable shapes recorded
// instead of launched; launch_dw_calls runs the list, equal shapes together.
static thread_local std::vector<DwCall>* t_dw_capture = nullptr;
void dw_capture_begin(std::vector<DwCall>* sink) { t_dw_capture = sink; }
void dw_capture_end() { t_dw_capture = nullptr; }

// STRIDE 2 (k = 3): the first operand lives on the twice finer grid, x[2o + k] pairs with dz[o] — the stride-2 conv
// (x = layer input, dz = output gradient, result [tap][ci][co]) and, with the operands swapped, the transposed conv
// (first operand = dz on the fine grid, second = x, result [tap][co][ci] = the TF layout of its kernel).
template <int CIN, int COUT, int KS, int STRIDE = 1>
__global__ void __launch_bounds__(256) conv_dw_tile_kernel(const float* x, const float* dz, float* partial, int B, int D,
                                                           int cin_total, int with_bias, DwBatch batch = DwBatch{}) {
  static_assert(STRIDE == 1 || (STRIDE == 2 && KS == 3), "stride 2 is implemented for 3x3x3 (front padding 0)");
  if (batch.n) { x = batch.x[blockIdx.z]; dz = batch.dz[blockIdx.z]; partial = batch.partial[blockIdx.z]; }
  constexpr int TD = STRIDE == 1 ? 4 : 2, TH = TD, TW = 16, PAD = STRIDE == 1 ? (KS - 1) / 2 : 0;
  constexpr int ID = STRIDE * (TD - 1) + KS, IH = STRIDE * (TH - 1) + KS, IW = STRIDE * (TW - 1) + KS;
  constexpr int TVOX = TD * TH * TW;
  constexpr int TAPS = KS * KS * KS;
  constexpr int TI = CIN < 4 ? CIN : 4, TJ = COUT < 4 ? COUT : 4;
  constexpr int NIB = CIN / TI, NJB = COUT / TJ;
  constexpr int T = TAPS * NIB * NJB;
  constexpr int PASSES = (T + 255) / 256;
  constexpr int S = T >= 256 ? 1 : 256 / T;
  constexpr int XVS = CIN == 16 ? 20 : (CIN == 8 ? 12 : CIN);
  constexpr int ZVS = COUT;
  __shared__ __attribute__((aligned(16))) float xt[ID * IH * IW * XVS];
  __shared__ __attribute__((aligned(16))) float zt[TVOX * ZVS];
  __shared__ float red[256];

  const int chunk = blockIdx.y;                        // channel chunk of CIN input channels
  const int tw = D / TW, th = D / TH, td = D / TD;
  const int ntiles = B * td * th * tw;

  // this thread's dW blocks
  int xbase[PASSES], zbase[PASSES];
  bool live[PASSES];
  const int split = S > 1 ? threadIdx.x / T : 0;
  const int t0 = S > 1 ? threadIdx.x - split * T : threadIdx.x;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int tt = t0 + 256 * p;
    live[p] = tt < T && split < S;
    const int jb = tt % NJB, ib = (tt / NJB) % NIB, tap = tt / (NJB * NIB);
    const int kw = tap % KS, kh = (tap / KS) % KS, kd = tap / (KS * KS);
    xbase[p] = live[p] ? ((kd * IH + kh) * IW + kw) * XVS + ib * TI : 0;
    zbase[p] = live[p] ? jb * TJ : 0;
  }
  // bias gradient db[co] = sum of dz over voxels, from the dz tile that is in LDS anyway (channel chunk 0 only):
  // thread t owns channel t % COUT and every (256 / COUT)-th voxel of each tile
  constexpr int BL = 256 / COUT;
  const bool do_bias = with_bias && blockIdx.y == 0;
  const int bc = threadIdx.x % COUT, bl = threadIdx.x / COUT;
  float bsum = 0.f;
  float acc[PASSES][TI][TJ];
#pragma unroll
  for (int p = 0; p < PASSES; ++p)
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) acc[p][i][j] = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int tz = bid % tw; bid /= tw;
    const int ty = bid % th; bid /= th;
    const int tx = bid % td; bid /= td;
    const int b = bid, od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
    const int DX = STRIDE * D;                           // grid of the first operand
    const float* xb = x + (int64_t)b * DX * DX * DX * cin_total + chunk * CIN;
    const float* zb = dz + (int64_t)b * D * D * D * COUT;
    __syncthreads();
    if constexpr (CIN >= 4) {
      stage_tile<ID, IH, IW, CIN / 4, XVS>(xt, xb, DX, cin_total, STRIDE * od0 - PAD, STRIDE * oh0 - PAD, STRIDE * ow0 - PAD);
    } else {
      for (int v = threadIdx.x; v < ID * IH * IW; v += 256) {
        const int zw = v % IW, zh = (v / IW) % IH, zd = v / (IW * IH);
        const int gd = STRIDE * od0 - PAD + zd, gh = STRIDE * oh0 - PAD + zh, gw = STRIDE * ow0 - PAD + zw;
        float val = 0.f;
        if ((unsigned)gd < (unsigned)DX && (unsigned)gh < (unsigned)DX && (unsigned)gw < (unsigned)DX)
          val = xb[(((int64_t)gd * DX + gh) * DX + gw) * cin_total];
        xt[v] = val;
      }
    }
    if constexpr (COUT >= 4) {
      stage_tile<TD, TH, TW, COUT / 4, ZVS>(zt, zb, D, COUT, od0, oh0, ow0);
    } else {
      const int v = threadIdx.x, w = v & 15, h = (v >> 4) % TH, d = v / (16 * TH);
      if (v < TVOX) zt[v] = zb[((int64_t)(od0 + d) * D + oh0 + h) * D + ow0 + w];
    }
    __syncthreads();
    if (do_bias) {
#pragma unroll 4
      for (int v = bl; v < TVOX; v += BL) bsum += zt[v * ZVS + bc];
    }

#pragma unroll 2
    for (int v = split; v < TVOX; v += S) {
      const int w = v & 15, h = (v >> 4) % TH, d = v / (16 * TH);
      const int xo = ((STRIDE * d * IH + STRIDE * h) * IW + STRIDE * w) * XVS, zo = v * ZVS;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        float xv[TI], zv[TJ];
        if constexpr (TI == 4) {
          const float4 q = *reinterpret_cast<const float4*>(&xt[xo + xbase[p]]);
          xv[0] = q.x; xv[1] = q.y; xv[2] = q.z; xv[3] = q.w;
        } else {
#pragma unroll
          for (int i = 0; i < TI; ++i) xv[i] = xt[xo + xbase[p] + i];
        }
        if constexpr (TJ == 4) {
          const float4 q = *reinterpret_cast<const float4*>(&zt[zo + zbase[p]]);
          zv[0] = q.x; zv[1] = q.y; zv[2] = q.z; zv[3] = q.w;
        } else {
#pragma unroll
          for (int j = 0; j < TJ; ++j) zv[j] = zt[zo + zbase[p] + j];
        }
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) acc[p][i][j] = fmaf(xv[i], zv[j], acc[p][i][j]);
      }
    }
  }

  // partial[group][tap][ci][co] (+ [co] bias sums behind the weights when with_bias)
  const size_t wn = (size_t)TAPS * cin_total * COUT;
  float* out = partial + (size_t)blockIdx.x * (wn + (with_bias ? COUT : 0));
  if (do_bias) {                          // fixed-order sum over the BL voxel lanes of each channel
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < COUT) {
      float s = 0.f;
      for (int l = 0; l < BL; ++l) s += red[l * COUT + threadIdx.x];
      out[wn + threadIdx.x] = s;
    }
  }
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int tt = t0 + 256 * p;
    const int jb = tt % NJB, ib = (tt / NJB) % NIB, tap = tt / (NJB * NIB);
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        float s = acc[p][i][j];
        if constexpr (S > 1) {            // fixed-order sum over the S voxel splits
          __syncthreads();
          red[threadIdx.x] = s;
          __syncthreads();
          s = 0.f;
          if (split == 0)
            for (int l = 0; l < S; ++l) s += red[l * T + t0];
        }
        if (tt < T && split == 0) out[((size_t)tap * cin_total + chunk * CIN + ib * TI + i) * COUT + jb * TJ + j] = s;
      }
  }
}

// Stride-1 3x3x3, small channel counts: "sliding kw" variant.  A thread owns the three kw taps of one (kd, kh) for a
// 4 x 4 (ci x co) block and walks WSEG consecutive voxels of a row, keeping the three x quads of the kw window in
// registers: per voxel ONE new x quad and one dz quad are read from LDS for 48 FMAs (the kernel above reads two quads
// per 16 FMAs and is LDS-bandwidth bound at about a quarter of the FMA rate).  T = 9 * (CIN/4) * (COUT/4) blocks,
// S = 256 / T threads per block splitting the tile's 256 / WSEG row segments; partial sums are combined in a fixed order.
template <int CIN, int COUT, int WSEG>
__global__ void __launch_bounds__(256) conv_dw_slide_kernel(const float* x, const float* dz, float* partial, int B, int D,
                                                            int cin_total, int with_bias) {
  constexpr int TD = 4, TH = 4, TW = 16, PAD = 1;
  constexpr int ID = TD + 2, IH = TH + 2, IW = TW + 2;
  constexpr int TVOX = TD * TH * TW;
  constexpr int NIB = CIN / 4, NJB = COUT / 4;
  constexpr int T = 9 * NIB * NJB;
  constexpr int S = 256 / T;
  constexpr int SEGS = TW / WSEG, NU = TD * TH * SEGS;
  static_assert(CIN % 4 == 0 && COUT % 4 == 0 && T <= 256 && S >= 1, "shape");
  constexpr int XVS = CIN == 16 ? 20 : (CIN == 8 ? 12 : CIN);
  constexpr int ZVS = COUT;
  __shared__ __attribute__((aligned(16))) float xt[ID * IH * IW * XVS];
  __shared__ __attribute__((aligned(16))) float zt[TVOX * ZVS];
  __shared__ float red[256];

  const int chunk = blockIdx.y;
  const int tw = D / TW, th = D / TH, td = D / TD;
  const int ntiles = B * td * th * tw;
  const int split = threadIdx.x / T, t0 = threadIdx.x - split * T;
  const bool live = split < S;
  const int jb = t0 % NJB, ib = (t0 / NJB) % NIB, khd = t0 / (NJB * NIB);
  const int kh = khd % 3, kd = khd / 3;
  constexpr int BL = 256 / COUT;
  const bool do_bias = with_bias && blockIdx.y == 0;
  const int bc = threadIdx.x % COUT, bl = threadIdx.x / COUT;
  float bsum = 0.f;
  float acc[3][4][4];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[k][i][j] = 0.f;

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int tz = bid % tw; bid /= tw;
    const int ty = bid % th; bid /= th;
    const int tx = bid % td; bid /= td;
    const int b = bid, od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
    const float* xb = x + (int64_t)b * D * D * D * cin_total + chunk * CIN;
    const float* zb = dz + (int64_t)b * D * D * D * COUT;
    __syncthreads();
    stage_tile<ID, IH, IW, CIN / 4, XVS>(xt, xb, D, cin_total, od0 - PAD, oh0 - PAD, ow0 - PAD);
    stage_tile<TD, TH, TW, COUT / 4, ZVS>(zt, zb, D, COUT, od0, oh0, ow0);
    __syncthreads();
    if (do_bias) {
#pragma unroll 4
      for (int v = bl; v < TVOX; v += BL) bsum += zt[v * ZVS + bc];
    }
    if (live) {
      for (int u = split; u < NU; u += S) {
        const int seg = u % SEGS, row = u / SEGS, h = row % TH, d = row / TH, w0 = seg * WSEG;
        const float* xr = &xt[(((d + kd) * IH + (h + kh)) * IW + w0) * XVS + ib * 4];
        const float* zr = &zt[((d * TH + h) * TW + w0) * ZVS + jb * 4];
        float4 xa = *reinterpret_cast<const float4*>(xr);
        float4 xb4 = *reinterpret_cast<const float4*>(xr + XVS);
#pragma unroll
        for (int w = 0; w < WSEG; ++w) {
          const float4 xc = *reinterpret_cast<const float4*>(xr + (w + 2) * XVS);
          const float4 z = *reinterpret_cast<const float4*>(zr + w * ZVS);
          const float zv[4] = {z.x, z.y, z.z, z.w};
          const float x0[4] = {xa.x, xa.y, xa.z, xa.w}, x1[4] = {xb4.x, xb4.y, xb4.z, xb4.w}, x2[4] = {xc.x, xc.y, xc.z, xc.w};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[0][i][j] = fmaf(x0[i], zv[j], acc[0][i][j]);
              acc[1][i][j] = fmaf(x1[i], zv[j], acc[1][i][j]);
              acc[2][i][j] = fmaf(x2[i], zv[j], acc[2][i][j]);
            }
          xa = xb4;
          xb4 = xc;
        }
      }
    }
  }

  const size_t wn = (size_t)27 * cin_total * COUT;
  float* out = partial + (size_t)blockIdx.x * (wn + (with_bias ? COUT : 0));
  if (do_bias) {
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < COUT) {
      float s = 0.f;
      for (int l = 0; l < BL; ++l) s += red[l * COUT + threadIdx.x];
      out[wn + threadIdx.x] = s;
    }
  }
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int tap = (kd * 3 + kh) * 3 + kw;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = acc[kw][i][j];
        if constexpr (S > 1) {            // fixed-order sum over the S row-segment splits
          __syncthreads();
          red[threadIdx.x] = s;
          __syncthreads();
          s = 0.f;
          if (split == 0)
            for (int l = 0; l < S; ++l) s += red[l * T + t0];
        }
        if (split == 0) out[((size_t)tap * cin_total + chunk * CIN + ib * 4 + i) * COUT + jb * 4 + j] = s;
      }
  }
}

// Layers with at least 16 x 16 channels per chunk (the 32^3 / 16^3 stages, the resamplers, the hyperprior nets): the same
// LDS tiles, the products on the matrix cores.  dW[tap] is a [16 ci x COUT] matrix and the sum over voxels is the K
// dimension of a GEMM:  v_mfma_f32_16x16x4_f32 with A = x (lane: ci = l % 16, voxel k = l / 16 of four consecutive voxels
// along w, shifted by the tap) and B = dz (lane: co = l % 16, the same four voxels) adds four voxels' outer products to a
// 16 x 16 block of one tap per instruction.  Wave w owns taps w, w + 4, ... (7, 7, 7, 6 of 27) x COUT / 16 column blocks
// = up to 28 accumulator quads; per four voxels it reads one B word per column block and one A word per tap from LDS.
// Summation order per weight: tiles in the workgroup's fixed strided order, voxel groups ascending, the four voxels of
// a group inside the instruction — bit-reproducible for a given (shape, grid), as above.  Same partial layout and bias sums
// as conv_dw_tile_kernel<16, COUT, 3, STRIDE> (VALU: 16 x 16 34 us, 16 x 32 41 us, 16 x 64 64 us per launch at 16^3 / 32^3).
// CIN = 8 (conv1_2 / conv2_2 of the C = 32 blocks at 32^3): the 16 rows carry TWO taps — row i reads channel i % 8 of tap
// 2 p + i / 8 — so 14 instructions cover the 27 taps (the last one half idle); COUT = 8 leaves half of the columns idle.
// Float offset of channel quad q of voxel (d, h, w) in one cube of a [D^3] tensor with C channels: NDHWC, or — q4 — the Q4
// layout [d][h][C/4][w][4] the training step keeps its 64^3 stage in (Trainer(q4=True): the row kernels' native layout).
__device__ __forceinline__ int64_t vox_off(int q4, int d, int h, int w, int q, int D, int C) {
  return q4 ? ((((int64_t)d * D + h) * (C >> 2) + q) * D + w) * 4 : (((int64_t)d * D + h) * D + w) * C + q * 4;
}

template <int COUT, int STRIDE, int CIN = 16>
__global__ void __launch_bounds__(256) conv_dw_mfma_kernel(const float* x, const float* dz, float* partial, int B, int D,
                                                           int cin_total, int with_bias, int x_q4 = 0, DwBatch batch = DwBatch{}) {
  constexpr int KS = 3;
  if (batch.n) { x = batch.x[blockIdx.z]; dz = batch.dz[blockIdx.z]; partial = batch.partial[blockIdx.z]; }
  static_assert((CIN == 16 || (CIN == 8 && STRIDE == 1)) && COUT % 16 == 0, "16 (or 8) x 16 / 32 / 64");
  constexpr int TD = STRIDE == 1 ? 4 : 2, TH = TD, TW = 16, PAD = STRIDE == 1 ? 1 : 0;
  constexpr int ID = STRIDE * (TD - 1) + KS, IH = STRIDE * (TH - 1) + KS, IW = STRIDE * (TW - 1) + KS;
  constexpr int TPR = 16 / CIN;                           // taps per row block
  constexpr int NP = (27 + TPR - 1) / TPR;                // row blocks (27 taps or 14 pairs), split over the four waves
  constexpr int TVOX = TD * TH * TW, TAPS = 27, NT = (COUT + 15) / 16, MAXT = (NP + 3) / 4;
  constexpr int XVS = CIN;                                // k-th voxel of a group lands CIN banks further
  constexpr int ZVS = COUT <= 16 ? COUT : COUT + 16;      // same for B when a voxel holds more than 16 channels
  __shared__ __attribute__((aligned(16))) float xt[ID * IH * IW * XVS];
  __shared__ __attribute__((aligned(16))) float zt[TVOX * ZVS];
  __shared__ float red[256];
  const int chunk = blockIdx.y;
  const int tw = D / TW, th = D / TH, td = D / TD;
  const int ntiles = B * td * th * tw;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  int xlane[MAXT];                                         // per-lane LDS index of voxel k = lane / 16, channel lane % 16, per tap
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int tap = (wv + 4 * t) * TPR + li / CIN;
    const int kw = tap % 3, kh = (tap / 3) % 3, kd = tap / 9;
    xlane[t] = (tap < TAPS ? ((kd * IH + kh) * IW + kw) * XVS : 0) + STRIDE * lk * XVS + li % CIN;
  }
  const int zlane = lk * ZVS + (COUT >= 16 ? li : li % COUT);
  constexpr int BL = 256 / COUT;
  const bool do_bias = with_bias && blockIdx.y == 0;
  const int bc = threadIdx.x % COUT, bl = threadIdx.x / COUT;
  float bsum = 0.f;
  f32x4 acc[MAXT][NT];                                     // (a wave's row block beyond NP multiplies tap 0 again and is never stored)
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int tz = bid % tw; bid /= tw;
    const int ty = bid % th; bid /= th;
    const int tx = bid % td; bid /= td;
    const int b = bid, od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
    const int DX = STRIDE * D;
    // x_q4: x is a Q4 tensor [d][h][cin_total / 4][w][4]; the chunk's first channel quad starts chunk * CIN / 4 rows of DX float4 further
    const float* xb = x + (int64_t)b * DX * DX * DX * cin_total + (x_q4 ? chunk * (CIN / 4) * DX * 4 : chunk * CIN);
    const float* zb = dz + (int64_t)b * D * D * D * COUT;
    __syncthreads();
    stage_tile<ID, IH, IW, CIN / 4, XVS>(xt, xb, DX, cin_total, STRIDE * od0 - PAD, STRIDE * oh0 - PAD, STRIDE * ow0 - PAD, x_q4 != 0);
    stage_tile<TD, TH, TW, COUT / 4, ZVS>(zt, zb, D, COUT, od0, oh0, ow0);
    __syncthreads();
    if (do_bias) {
#pragma unroll 4
      for (int v = bl; v < TVOX; v += BL) bsum += zt[v * ZVS + bc];
    }
    // 64 (stride 1) or 16 (stride 2) groups of four voxels, fully unrolled: every LDS address is a per-lane base (one per
    // tap) plus a compile-time offset, nothing between the reads and the MFMAs is conditional (a wave whose last tap does
    // not exist multiplies tap 0 again into an accumulator that is never stored), so the compiler issues the reads of a
    // group ahead of the previous group's MFMAs and waits with counted lgkmcnt
#pragma unroll
    for (int d = 0; d < TD; ++d)
#pragma unroll
      for (int h = 0; h < TH; ++h)
#pragma unroll
        for (int gw = 0; gw < 4; ++gw) {
          const int g = (d * TH + h) * 4 + gw;
          const int xc = ((STRIDE * d * IH + STRIDE * h) * IW + STRIDE * 4 * gw) * XVS;      // compile-time
          float bz[NT], ax[MAXT];
#pragma unroll
          for (int n = 0; n < NT; ++n) bz[n] = zt[zlane + 4 * g * ZVS + n * 16];
#pragma unroll
          for (int t = 0; t < MAXT; ++t) ax[t] = xt[xlane[t] + xc];
#pragma unroll
          for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[t][n] = mfma4(ax[t], bz[n], acc[t][n]);
        }
  }
  const size_t wn = (size_t)TAPS * cin_total * COUT;
  float* out = partial + (size_t)blockIdx.x * (wn + (with_bias ? COUT : 0));
  if (do_bias) {
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < COUT) {
      float s_ = 0.f;
      for (int l = 0; l < BL; ++l) s_ += red[l * COUT + threadIdx.x];
      out[wn + threadIdx.x] = s_;
    }
  }
  // D quad r of lane l = row 4 * (l / 16) + r (tap row / CIN of the row block, channel row % CIN), column co = l % 16
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lk + r, tap = (wv + 4 * t) * TPR + row / CIN;
      if (tap < TAPS && (COUT >= 16 || li < COUT)) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
          out[((size_t)tap * cin_total + chunk * CIN + row % CIN) * COUT + n * 16 + li] = acc[t][n][r];
      }
    }
  }
}

// 16 -> 4 (conv1_1 of the C = 16 blocks at 64^3: the largest weight gradient of the step): four output channels fill a
// quarter of the 16 MFMA columns, so the three kw taps go into the columns as well — column j = kw * 4 + co reads dz SHIFTED
// by -(kw - 1) along w (the sum over voxels runs over the x position v' = v + kw - 1 instead of the output voxel v, which
// is the same set of products: dz outside the cube contributes nothing and is staged as zeros):
//   D(kd, kh)[ci][kw * 4 + co] += sum over the 4 voxels v' of a group  x[v' + (kd - 1, kh - 1, 0)][ci] * dz[v' - (0, 0, kw - 1)][co]
// 12 of 16 columns carry weights (75 % of the MFMA), one B read serves all nine (kd, kh) accumulators, x needs no w halo.
// The four waves split the tile by depth slice (balanced: 9 accumulators do not divide by 4) and add their sums in wave
// order through LDS after the last tile.  Same partial layout / bias sums as conv_dw_slide_kernel<16, 4, 8> (138 us per
// 8 cubes of 64^3).
// PAIR: the 1x1x1 layer that reads the same input (conv2_1 next to conv1_1: x = the block input, dz1 = its own output
// gradient) rides along — its weight gradient is one more accumulator fed by the centre row of the x tile already in LDS
// and a dz1 tile without halo (D1[ci][co] += x[v][ci] * dz1[v][co]: COUT of the 16 columns, one instruction in ten), instead
// of a kernel of its own that reads x from memory again (conv_dw_1x1_kernel<16, 4>: 38 us per 8 cubes of 64^3, at the HBM rate).
template <int COUT, bool PAIR = false>
__global__ void __launch_bounds__(256) conv_dw_mfma_16xn_kernel(const float* x, const float* dz, float* partial, int B, int D,
                                                                int cin_total, int with_bias, const float* dz1 = nullptr,
                                                                float* partial1 = nullptr, int x_q4 = 0) {
  // COUT = 4: the form described above.  COUT = 8 (conv1_1 of the C = 32 blocks at 32^3, 16 input channels per chunk): the
  // 3 x 8 = 24 columns take two column blocks (kw 0, 1 | kw 2 and 8 idle columns), 18 accumulators, two B reads per group.
  static_assert(COUT == 4 || COUT == 8, "16 -> 4 and 16 -> 8");
  constexpr int CIN = 16, TD = 4, TH = 4, TW = 16, ID = TD + 2, IH = TH + 2, ZW = TW + 2;
  constexpr int NT = (3 * COUT + 15) / 16, ZQ = COUT / 4;
  constexpr int TVOX = TD * TH * TW, XVS = 16;
  __shared__ __attribute__((aligned(16))) float xt[ID * IH * TW * XVS];        // 36 KB; reused for the final wave sum
  __shared__ __attribute__((aligned(16))) float zt[TD * TH * ZW * COUT];
  __shared__ __attribute__((aligned(16))) float z1t[PAIR ? TD * TH * TW * COUT : 4];
  __shared__ float red[256];
  const int chunk = blockIdx.y;
  const int tw = D / TW, th = D / TH, td = D / TD;
  const int ntiles = B * td * th * tw;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int xlane = (wv * IH * TW + lk) * XVS + li;                             // depth slice d = wv, voxel k, channel li
  int zlane[NT];                                                                // dz tile has one halo voxel on each side in w
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int j = n * 16 + li;                                                  // column of the [3 kw x COUT] matrix
    const int jkw = j < 3 * COUT ? j / COUT : 2, jco = j % COUT;                // idle columns: a valid address, never stored
    zlane[n] = (wv * TH * ZW + lk + 2 - jkw) * COUT + jco;
  }
  const int z1lane = (wv * TH * TW + lk) * COUT + li % COUT;                    // columns >= COUT repeat, never stored
  constexpr int BL = 256 / COUT;
  const bool do_bias = with_bias && blockIdx.y == 0;
  const int bc = threadIdx.x % COUT, bl = threadIdx.x / COUT;
  float bsum = 0.f, bsum1 = 0.f;
  f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[9][NT];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Tiles are double-buffered through REGISTERS: the global loads of tile n + 1 are issued before the MFMAs of tile n and
  // land in LDS after them (one memory round trip per tile hidden behind ~2 us of matrix work; staged-then-computed the
  // kernel spent 5 of its 7 us per tile waiting for two serial round trips: 115 us per launch against 138 us for the VALU
  // kernel it replaces).  x tile: 36 rows of 16 voxels x 4 float4, one float4 per lane per row, wave w takes rows
  // w, w + 4, ...; dz tile: 16 rows of 18 voxels (w halo, zero outside the cube) x COUT / 4 float4 over 256 threads.
  constexpr int XR = (ID * IH) / 4;                       // x rows per wave (9)
  constexpr int ZN = TD * TH * ZW * ZQ, ZPT = (ZN + 255) / 256;
  constexpr int Z1N = TD * TH * TW * ZQ, Z1PT = PAIR ? Z1N / 256 : 1;          // 256 or 512 float4 per tile
  float4 xr[XR], zr[ZPT], z1r[Z1PT];
  const int xvox = lane >> 2, xq = lane & 3;
  auto load_tile = [&](int tile) {
    int bid = tile;
    const int tz = bid % tw; bid /= tw;
    const int ty = bid % th; bid /= th;
    const int tx = bid % td; bid /= td;
    const int b = bid, od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
    const float* xb = x + (int64_t)b * D * D * D * cin_total + (x_q4 ? 0 : chunk * CIN);
    const float* zb = dz + (int64_t)b * D * D * D * COUT;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int r = wv + 4 * i, zd = r / IH, zh = r - zd * IH;
      const int gd = od0 - 1 + zd, gh = oh0 - 1 + zh;                         // wave-uniform
      xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)D)
        xr[i] = *reinterpret_cast<const float4*>(xb + (x_q4 ? vox_off(1, gd, gh, ow0 + xvox, chunk * (CIN / 4) + xq, D, cin_total)
                                                            : ((int64_t)(gd * D + gh) * D + ow0 + xvox) * cin_total + xq * 4));
    }
#pragma unroll
    for (int j = 0; j < ZPT; ++j) {
      const int i = threadIdx.x + 256 * j, q = i % ZQ, v = i / ZQ, row = v / ZW, vox = v - row * ZW;
      const int gw = ow0 - 1 + vox;
      zr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < ZN && (unsigned)gw < (unsigned)D)
        zr[j] = *reinterpret_cast<const float4*>(zb + ((int64_t)((od0 + row / TH) * D + oh0 + row % TH) * D + gw) * COUT + q * 4);
    }
    if constexpr (PAIR) {
      const float* z1b = dz1 + (int64_t)b * D * D * D * COUT;
#pragma unroll
      for (int j = 0; j < Z1PT; ++j) {
        const int i = threadIdx.x + 256 * j, q = i % ZQ, v = i / ZQ, row = v / TW, vox = v - row * TW;
        z1r[j] = *reinterpret_cast<const float4*>(z1b + ((int64_t)((od0 + row / TH) * D + oh0 + row % TH) * D + ow0 + vox) * COUT + q * 4);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < XR; ++i) *reinterpret_cast<float4*>(&xt[((wv + 4 * i) * TW + xvox) * XVS + xq * 4]) = xr[i];
#pragma unroll
    for (int j = 0; j < ZPT; ++j) {
      const int i = threadIdx.x + 256 * j;
      if (i < ZN) *reinterpret_cast<float4*>(&zt[i * 4]) = zr[j];
    }
    if constexpr (PAIR) {
#pragma unroll
      for (int j = 0; j < Z1PT; ++j) *reinterpret_cast<float4*>(&z1t[(threadIdx.x + 256 * j) * 4]) = z1r[j];
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                                       // the previous tile's MFMAs have read their operands
    store_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
    if (do_bias) {
#pragma unroll 4
      for (int v = bl; v < TVOX; v += BL) {
        const int w = v & 15, hd = v >> 4;
        bsum += zt[(hd * ZW + w + 1) * COUT + bc];
        if constexpr (PAIR) bsum1 += z1t[v * COUT + bc];
      }
    }
#pragma unroll
    for (int h = 0; h < TH; ++h)
#pragma unroll
      for (int gw = 0; gw < 4; ++gw) {
        float bz[NT], ax[9];
#pragma unroll
        for (int n = 0; n < NT; ++n) bz[n] = zt[zlane[n] + (h * ZW + 4 * gw) * COUT];
#pragma unroll
        for (int t = 0; t < 9; ++t) ax[t] = xt[xlane + (((t / 3) * IH + h + (t % 3)) * TW + 4 * gw) * XVS];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[t][n] = mfma4(ax[t], bz[n], acc[t][n]);
        if constexpr (PAIR) acc1 = mfma4(ax[4], z1t[z1lane + (h * TW + 4 * gw) * COUT], acc1);     // ax[4]: kd = kh = 1
      }
  }
  const size_t wn = (size_t)27 * cin_total * COUT;
  float* out = partial + (size_t)blockIdx.x * (wn + (with_bias ? COUT : 0));
  if (do_bias) {
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < COUT) {
      float s_ = 0.f;
      for (int l = 0; l < BL; ++l) s_ += red[l * COUT + threadIdx.x];
      out[wn + threadIdx.x] = s_;
    }
  }
  // the four depth slices' sums, added in wave order, one column block at a time: xt[wave][t][r][lane]
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) xt[((wv * 9 + t) * 4 + r) * 64 + lane] = acc[t][n][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * 4 * 64; e += 256) {
      const float v = ((xt[e] + xt[9 * 256 + e]) + xt[2 * 9 * 256 + e]) + xt[3 * 9 * 256 + e];
      const int l = e & 63, r = (e >> 6) & 3, t = e >> 8;
      const int j = n * 16 + (l & 15), ci = 4 * (l >> 4) + r;                     // D quad r of lane l = [ci = 4 (l / 16) + r][column l % 16]
      if (j < 3 * COUT) out[((size_t)(t * 3 + j / COUT) * cin_total + chunk * CIN + ci) * COUT + (j % COUT)] = v;
    }
  }
  if constexpr (PAIR) {
    const size_t wn1 = (size_t)cin_total * COUT;
    float* out1 = partial1 + (size_t)blockIdx.x * (wn1 + (with_bias ? COUT : 0));
    if (do_bias) {
      __syncthreads();
      red[threadIdx.x] = bsum1;
      __syncthreads();
      if (threadIdx.x < COUT) {
        float s_ = 0.f;
        for (int l = 0; l < BL; ++l) s_ += red[l * COUT + threadIdx.x];
        out1[wn1 + threadIdx.x] = s_;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) xt[(wv * 4 + r) * 64 + lane] = acc1[r];
    __syncthreads();
    if (threadIdx.x < 256) {
      const int e = threadIdx.x;
      const float v = ((xt[e] + xt[256 + e]) + xt[2 * 256 + e]) + xt[3 * 256 + e];
      const int l = e & 63, r = e >> 6, co = l & 15, ci = 4 * (l >> 4) + r;
      if (co < COUT) out1[(size_t)(chunk * CIN + ci) * COUT + co] = v;
    }
  }
}

// 4 -> 4 and 4 -> 8 (conv2_2 / conv1_2 of the C = 16 blocks at 64^3): both operands are narrow, so BOTH MFMA dimensions
// carry taps — row i = kh * 4 + ci reads x of the row h + kh - 1, column j = kw * COUT + co reads dz shifted by -(kw - 1)
// along w (as above), one accumulator per kd:
//   D(kd)[kh * 4 + ci][kw * COUT + co] += sum over the 4 voxels v' of a group  x[v' + (kd - 1, kh - 1, 0)][ci] * dz[v' - (0, 0, kw - 1)][co]
// 12 of 16 rows and 12 of 16 (COUT = 4) or 24 of 32 (COUT = 8, two column blocks) columns are weights: 56 % of the MFMA, but
// 3 or 6 instructions per four voxels where the VALU kernels (conv_dw_slide_kernel<4, 4, 4> 79 us, <4, 8, 4> 90 us per 8 cubes,
// 23 and 40 TFLOP/s: bound by LDS reads) issue 108 / 216 FMAs per voxel.  Tiles are whole rows (4 x 4 x 64 voxels) so that a
// tile carries enough MFMAs (192 / 384 per wave) to cover the register-double-buffered staging of the next one; a wave owns
// one depth slice, the four waves' sums are added in wave order at the end.  x rows are 264 floats apart in LDS (the three kh
// rows of a lane group then fall into different banks).
template <int COUT>
__global__ void __launch_bounds__(256) conv_dw_mfma_4xn_kernel(const float* x, const float* dz, float* partial, int B, int D,
                                                               int with_bias, int dz_q4 = 0) {
  constexpr int CIN = 4, TD = 4, TH = 4, TW = 64, ID = TD + 2, IH = TH + 2, ZW = TW + 2, RS = TW * CIN + 8;
  constexpr int NT = COUT / 4, ZQ = COUT / 4;                                  // column blocks; float4 per dz voxel
  constexpr int TVOX = TD * TH * TW;
  constexpr int ZN = TD * TH * ZW * ZQ, ZPT = (ZN + 255) / 256;                // dz float4 per tile / per thread
  __shared__ __attribute__((aligned(16))) float xt[ID * IH * RS];              // 38 KB; reused for the final wave sum
  __shared__ __attribute__((aligned(16))) float zt[TD * TH * ZW * COUT];
  __shared__ float red[256];
  const int th = D / TH, td = D / TD;
  const int ntiles = B * td * th;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int ikh = li < 12 ? li >> 2 : 2, ici = li & 3;                          // rows 12..15: a valid address, never stored
  const int xlane = (wv * IH + ikh) * RS + lk * CIN + ici;
  int zlane[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int j = n * 16 + li;                                                  // column of the [3 kw x COUT] matrix (24 or 12 valid)
    const int kw = j < 3 * COUT ? j / COUT : 2, co = j % COUT;
    zlane[n] = (wv * TH * ZW + lk + 2 - kw) * COUT + co;
  }
  constexpr int BL = 256 / COUT;
  const bool do_bias = with_bias != 0;
  const int bc = threadIdx.x % COUT, bl = threadIdx.x / COUT;
  float bsum = 0.f;
  f32x4 acc[3][NT];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int XR = (ID * IH) / 4;                                             // x rows per wave (9), one float4 per lane per row
  float4 xr[XR], zr[ZPT];
  auto load_tile = [&](int tile) {
    int bid = tile;
    const int ty = bid % th; bid /= th;
    const int tx = bid % td; bid /= td;
    const int b = bid, od0 = tx * TD, oh0 = ty * TH;
    const float* xb = x + (int64_t)b * D * D * D * CIN;
    const float* zb = dz + (int64_t)b * D * D * D * COUT;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      const int r = wv + 4 * i, zd = r / IH, zh = r - zd * IH;
      const int gd = od0 - 1 + zd, gh = oh0 - 1 + zh;                           // wave-uniform
      xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)D)
        xr[i] = *reinterpret_cast<const float4*>(xb + ((int64_t)(gd * D + gh) * D + lane) * CIN);
    }
#pragma unroll
    for (int j = 0; j < ZPT; ++j) {
      const int i = threadIdx.x + 256 * j, q = i % ZQ, v = i / ZQ, row = v / ZW, vox = v - row * ZW;
      const int gw = vox - 1;
      zr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < ZN && (unsigned)gw < (unsigned)D)
        zr[j] = *reinterpret_cast<const float4*>(zb + vox_off(dz_q4, od0 + row / TH, oh0 + row % TH, gw, q, D, COUT));
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < XR; ++i) *reinterpret_cast<float4*>(&xt[(wv + 4 * i) * RS + lane * CIN]) = xr[i];
#pragma unroll
    for (int j = 0; j < ZPT; ++j) {
      const int i = threadIdx.x + 256 * j;
      if (i < ZN) *reinterpret_cast<float4*>(&zt[i * 4]) = zr[j];
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
    if (do_bias) {
#pragma unroll 4
      for (int v = bl; v < TVOX; v += BL) {
        const int w = v & 63, hd = v >> 6;
        bsum += zt[(hd * ZW + w + 1) * COUT + bc];
      }
    }
#pragma unroll
    for (int h = 0; h < TH; ++h)
#pragma unroll 4
      for (int g = 0; g < TW / 4; ++g) {
        float bz[NT], ax[3];
#pragma unroll
        for (int n = 0; n < NT; ++n) bz[n] = zt[zlane[n] + (h * ZW + 4 * g) * COUT];
#pragma unroll
        for (int t = 0; t < 3; ++t) ax[t] = xt[xlane + (t * IH + h) * RS + 4 * g * CIN];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[t][n] = mfma4(ax[t], bz[n], acc[t][n]);
      }
  }
  const size_t wn = (size_t)27 * CIN * COUT;
  float* out = partial + (size_t)blockIdx.x * (wn + (with_bias ? COUT : 0));
  if (do_bias) {
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < COUT) {
      float s_ = 0.f;
      for (int l = 0; l < BL; ++l) s_ += red[l * COUT + threadIdx.x];
      out[wn + threadIdx.x] = s_;
    }
  }
  // the four depth slices' sums, added in wave order: xt[wave][kd][n][r][lane]
  constexpr int NA = 3 * NT * 4 * 64;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) xt[wv * NA + ((t * NT + n) * 4 + r) * 64 + lane] = acc[t][n][r];
  __syncthreads();
  for (int e = threadIdx.x; e < NA; e += 256) {
    const float v = ((xt[e] + xt[NA + e]) + xt[2 * NA + e]) + xt[3 * NA + e];
    const int l = e & 63, r = (e >> 6) & 3, n = (e >> 8) % NT, t = e / (256 * NT);
    const int kh = l >> 4, j = n * 16 + (l & 15);                               // D quad r of lane l = [row 4 (l / 16) + r][column l % 16]
    if (kh < 3 && j < 3 * COUT) out[((size_t)((t * 3 + kh) * 3 + j / COUT) * CIN + r) * COUT + j % COUT] = v;
  }
}

// conv_in (1 -> 16) and deconv_out (16 -> 1) at 64^3, the two ends of the transforms: one operand has a single channel, so
// the 27 taps fill that side of the matrix instead — MODE 0 (16 -> 1): rows = ci, columns = taps reading dz shifted by
// -(tap - 1) in all three dimensions (dz tile with a halo, zero outside the cube); MODE 1 (1 -> 16): rows = taps reading x
// shifted by tap - 1 (x tile with a halo), columns = co.  27 of 32 rows / columns carry weights, two instructions per four
// voxels; a wave owns one depth slice of the 4 x 4 x 16 tile, the four waves' sums are added in wave order at the end.
// Both kernels read their 16-channel operand once (134 MB per 8 cubes of 64^3) and run at the memory rate; the VALU tile
// kernels they replace (conv_dw_tile_kernel<16, 1, 3> / <1, 16, 3>) took 162 / 154 us per 8 cubes.
template <int MODE>
__global__ void __launch_bounds__(256) conv_dw_mfma_edge_kernel(const float* x, const float* dz, float* partial, int B, int D,
                                                                int with_bias, int wide_q4 = 0) {
  constexpr int TD = 4, TH = 4, TW = 16, HD_ = TD + 2, HH = TH + 2, HW = TW + 2, C = 16;
  constexpr int CIN = MODE == 0 ? 16 : 1, COUT = MODE == 0 ? 1 : 16;
  constexpr int TVOX = TD * TH * TW;
  __shared__ __attribute__((aligned(16))) float wide[TVOX * C];              // the 16-channel operand (x in MODE 0, dz in MODE 1)
  __shared__ float halo[HD_ * HH * HW];                                      // the 1-channel operand with its halo
  __shared__ float red[256];
  const int tw = D / TW, th = D / TH, td = D / TD;
  const int ntiles = B * td * th * tw;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int wlane = (wv * TH * TW + lk) * C + li;                            // depth slice wv, voxel k, channel li
  int hlane[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int j = n * 16 + li;                                               // tap index of this row / column (27 valid)
    const int tap = j < 27 ? j : 26;
    const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
    // MODE 0: dz[v' - (tap - 1)] -> halo index (d' + 2 - kd, h + 2 - kh, w + 2 - kw); MODE 1: x[v + tap - 1] -> (d' + kd, ...)
    hlane[n] = MODE == 0 ? ((wv + 2 - kd) * HH + (2 - kh)) * HW + (2 - kw) + lk : ((wv + kd) * HH + kh) * HW + kw + lk;
  }
  const bool do_bias = with_bias != 0;
  constexpr int BL = 256 / COUT;
  const int bc = threadIdx.x % COUT, bl = threadIdx.x / COUT;
  float bsum = 0.f;
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int bid = tile;
    const int tz = bid % tw; bid /= tw;
    const int ty = bid % th; bid /= th;
    const int tx = bid % td; bid /= td;
    const int b = bid, od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
    const float* wb = (MODE == 0 ? x : dz) + (int64_t)b * D * D * D * C;
    const float* hb = (MODE == 0 ? dz : x) + (int64_t)b * D * D * D;
    __syncthreads();
    stage_tile<TD, TH, TW, C / 4, C>(wide, wb, D, C, od0, oh0, ow0, wide_q4 != 0);     // wide_q4: the 16-channel operand is a Q4 tensor
    for (int v = threadIdx.x; v < HD_ * HH * HW; v += 256) {
      const int zw = v % HW, zh = (v / HW) % HH, zd = v / (HW * HH);
      const int gd = od0 - 1 + zd, gh = oh0 - 1 + zh, gw = ow0 - 1 + zw;
      float val = 0.f;
      if ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)D && (unsigned)gw < (unsigned)D)
        val = hb[((int64_t)gd * D + gh) * D + gw];
      halo[v] = val;
    }
    __syncthreads();
    if (do_bias) {                                          // sums of dz over the tile's own voxels
      if constexpr (MODE == 0) {
        const int v = threadIdx.x, w = v & 15, h = (v >> 4) % TH, d = v / (16 * TH);
        bsum += halo[((d + 1) * HH + h + 1) * HW + w + 1];
      } else {
#pragma unroll 4
        for (int v = bl; v < TVOX; v += BL) bsum += wide[v * C + bc];
      }
    }
#pragma unroll
    for (int h = 0; h < TH; ++h)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float wv_ = wide[wlane + (h * TW + 4 * g) * C];
        const float h0 = halo[hlane[0] + h * HW + 4 * g], h1 = halo[hlane[1] + h * HW + 4 * g];
        if constexpr (MODE == 0) {                          // rows ci, columns taps
          acc[0] = mfma4(wv_, h0, acc[0]);
          acc[1] = mfma4(wv_, h1, acc[1]);
        } else {                                            // rows taps, columns co
          acc[0] = mfma4(h0, wv_, acc[0]);
          acc[1] = mfma4(h1, wv_, acc[1]);
        }
      }
  }
  const size_t wn = (size_t)27 * CIN * COUT;
  float* out = partial + (size_t)blockIdx.x * (wn + (with_bias ? COUT : 0));
  if (do_bias) {
    __syncthreads();
    red[threadIdx.x] = bsum;
    __syncthreads();
    if (threadIdx.x < COUT) {
      float s_ = 0.f;
      for (int l = 0; l < BL; ++l) s_ += red[l * COUT + threadIdx.x];
      out[wn + threadIdx.x] = s_;
    }
  }
  // the four depth slices' sums, added in wave order: wide[wave][n][r][lane]
  __syncthreads();
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) wide[((wv * 2 + n) * 4 + r) * 64 + lane] = acc[n][r];
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * 4 * 64; e += 256) {
    const float v = ((wide[e] + wide[512 + e]) + wide[2 * 512 + e]) + wide[3 * 512 + e];
    const int l = e & 63, r = (e >> 6) & 3, n = e >> 8;
    const int row = 4 * (l >> 4) + r, col = l & 15;        // D quad r of lane l = [row 4 (l / 16) + r][column l % 16]
    if constexpr (MODE == 0) {
      const int tap = n * 16 + col;                         // columns are taps, rows ci: out[tap][ci][0]
      if (tap < 27) out[tap * 16 + row] = v;
    } else {
      const int tap = n * 16 + row;                         // rows are taps, columns co: out[tap][0][co]
      if (tap < 27) out[tap * 16 + col] = v;
    }
  }
}

template <int COUT, int STRIDE, int CIN = 16>
static int run_dw_mfma(const float* x, const float* dz, float* partial, int B, int D, int Cin, int groups, int with_bias,
                       hipStream_t s, int x_q4 = 0) {
  if (t_dw_capture && STRIDE == 1 && CIN == 16 && !x_q4) {
    t_dw_capture->push_back(DwCall{COUT == 16 ? 0 : (COUT == 32 ? 1 : 2), x, dz, partial, B, D, Cin, groups, with_bias});
    return 1;
  }
  hipLaunchKernelGGL((conv_dw_mfma_kernel<COUT, STRIDE, CIN>), dim3(groups, Cin / CIN), dim3(256), 0, s, x, dz, partial, B, D, Cin,
                     with_bias, x_q4);
  int rc = launch_ok("conv_dw_mfma_kernel");
  return rc ? rc : 1;
}
static bool dw_mfma_enabled() {
  static const bool on = !(getenv("PCGC_DW_MFMA") && atoi(getenv("PCGC_DW_MFMA")) == 0);          // experiment knob
  return on;
}

static bool dw_edge_enabled() {
  static const bool on = !(getenv("PCGC_DW_EDGE") && atoi(getenv("PCGC_DW_EDGE")) == 0);          // experiment knob: 1 <-> 16 channels
  return on;
}
static bool dw_mfma32_enabled() {
  static const bool on = !(getenv("PCGC_DW_MFMA32") && atoi(getenv("PCGC_DW_MFMA32")) == 0);      // experiment knob: 16 -> 8, 8 -> 16 / 8
  return on;
}

template <int CIN, int COUT, int WSEG>
static int run_dw_slide(const float* x, const float* dz, float* partial, int B, int D, int Cin, int groups, int with_bias,
                        hipStream_t s) {
  hipLaunchKernelGGL((conv_dw_slide_kernel<CIN, COUT, WSEG>), dim3(groups, Cin / CIN), dim3(256), 0, s, x, dz, partial, B, D, Cin,
                     with_bias);
  int rc = launch_ok("conv_dw_slide_kernel");
  return rc ? rc : 1;
}

template <int CIN, int COUT, int KS>
static int run_dw(const float* x, const float* dz, float* partial, int B, int D, int Cin, int groups, int with_bias,
                  hipStream_t s) {
  if (t_dw_capture && KS == 1 && CIN == 16 && (COUT == 16 || COUT == 32)) {
    t_dw_capture->push_back(DwCall{COUT == 16 ? 3 : 4, x, dz, partial, B, D, Cin, groups, with_bias});
    return 1;
  }
  hipLaunchKernelGGL((conv_dw_tile_kernel<CIN, COUT, KS>), dim3(groups, Cin / CIN), dim3(256), 0, s, x, dz, partial, B, D, Cin,
                     with_bias);
  int rc = launch_ok("conv_dw_tile_kernel");
  return rc ? rc : 1;
}

// Tiles per workgroup of a launch with fewer tiles than 2 x kDwGroups (PCGC_DW_TPG / PCGC_DW_TPG_S2: experiment knobs).  Every
// workgroup writes a partial sum of the WHOLE filter gradient, which the final reduction reads again: for down_2 / up_1 (27 x 32 x 64
// weights) on a batch of 8 cubes that is 512 tiles of 2 x 2 x 16 coarse voxels -> 113 MB written and read per layer with one tile
// per workgroup.  Two tiles each: the step 8.54 -> 8.47 ms (four: the same; the stride-1 layers of the 16^3 stage, 128 tiles:
// 8.53 with two, 8.63 with four — left at one; 256 instead of 512 workgroups for the 1 024 tiles of the 32^3 stage: 8.36 against 8.34).
static int dw_tiles_per_group(const char* name, int dflt) {
  const char* e = getenv(name);
  const int d = e ? atoi(e) : dflt;
  return d < 1 ? 1 : d;
}
int conv_dw_tile_groups(int B, int D) {
  static const int div = dw_tiles_per_group("PCGC_DW_TPG", 1);
  const int ntiles = (B * (D / 4) * (D / 4) * (D / 16) + div - 1) / div;
  return ntiles < kDwGroups ? ntiles : kDwGroups;
}
int conv_dw_tile_groups_s2(int B, int D) {          // D = coarse grid; tiles of 2 x 2 x 16
  static const int div = dw_tiles_per_group("PCGC_DW_TPG_S2", 2);
  const int ntiles = (B * (D / 2) * (D / 2) * (D / 16) + div - 1) / div;
  return ntiles < kDwGroups ? ntiles : kDwGroups;
}

// Stride-2 pair, 3x3x3.  fine = operand on the 2D grid with Ca channels, coarse = operand on the D grid with Cb
// channels; partial = [groups][27][Ca][Cb] (+ Cb sums of `coarse` when with_bias).  Returns 1 / 0 / <0 as below.
int launch_conv_dw_tile_s2(const float* fine, const float* coarse, float* partial, int B, int D, int Ca, int Cb, int with_bias,
                           hipStream_t s, int fine_q4) {
  if (D % 16) return 0;
  const int g = conv_dw_tile_groups_s2(B, D);
  if (dw_mfma_enabled() && Ca % 16 == 0) {
    if (Cb == 16) return run_dw_mfma<16, 2>(fine, coarse, partial, B, D, Ca, g, with_bias, s, fine_q4);
    if (Cb == 32) return run_dw_mfma<32, 2>(fine, coarse, partial, B, D, Ca, g, with_bias, s, fine_q4);
    if (Cb == 64) return run_dw_mfma<64, 2>(fine, coarse, partial, B, D, Ca, g, with_bias, s, fine_q4);
  }
  if (fine_q4) { set_error("weight gradient of a stride-2 layer: no kernel reads a Q4 operand for Ca=%d Cb=%d", Ca, Cb); return -1; }
#define TRY2(cb)                                                                                                     \
  if (Ca % 16 == 0 && Cb == cb) {                                                                                    \
    hipLaunchKernelGGL((conv_dw_tile_kernel<16, cb, 3, 2>), dim3(g, Ca / 16), dim3(256), 0, s, fine, coarse, partial, B, D, Ca, \
                       with_bias);                                                                                   \
    int rc = launch_ok("conv_dw_tile_kernel (stride 2)");                                                           \
    return rc ? rc : 1;                                                                                              \
  }
  TRY2(16) TRY2(32) TRY2(64)
#undef TRY2
  return 0;
}

// 1^3 layers with at most 64 (ci, co) pairs (conv2_1 16 -> 4 and conv2_3 4 -> 8 of the C = 16 blocks): the weight gradient
// is one outer product per voxel summed over the voxels, i.e. a reduction that reads x and dz exactly once.  A thread walks
// voxels tid, tid + stride, ... with the CIN x COUT sums in registers; then the 64 lanes of a wave are added by a
// butterfly, the 4 waves through LDS in wave order: fixed order, partial = [groups][CIN*COUT (+ COUT bias sums)] like the
// tile kernel (which staged 4 x 4 x 16-voxel tiles through LDS for these layers: 55 / 49 us per 8 cubes of 64^3).
template <int CIN, int COUT>
__global__ void __launch_bounds__(256) conv_dw_1x1_kernel(const float* x, const float* dz, float* partial, int64_t nvox, int with_bias,
                                                          int dz_q4_w = 0) {      // dz_q4_w = W of a Q4 dz tensor ([row][COUT / 4][w][4]), 0: NDHWC
  constexpr int QI = CIN / 4, QO = COUT / 4, NACC = CIN * COUT;
  static_assert(NACC <= 64, "register-resident sums");
  __shared__ float sh[4][NACC + COUT];
  float acc[CIN][COUT], bs[COUT];
#pragma unroll
  for (int i = 0; i < CIN; ++i)
#pragma unroll
    for (int j = 0; j < COUT; ++j) acc[i][j] = 0.f;
#pragma unroll
  for (int j = 0; j < COUT; ++j) bs[j] = 0.f;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* g4 = reinterpret_cast<const float4*>(dz);
  // a thread's voxels in ascending order, U of them requested before the first is used (one voxel per round trip left the
  // kernel at 3.5 TB/s: 16 dependent iterations of ~2 us)
  constexpr int U = 4;                       // (8 for the 4 x 8 layer: 34 against 33 us per 8 cubes — the launch is at its floor of fixed cost + bytes)
  const int64_t step = (int64_t)gridDim.x * 256;
  // Q4 dz: W divides 256 (launch_conv_dw_tile checks), so a thread keeps its w and its row advances by step / W per voxel —
  // no 64-bit division per voxel
  const int64_t v_first = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int qw = dz_q4_w ? (int)(v_first % dz_q4_w) : 0;
  const int64_t row_first = dz_q4_w ? v_first / dz_q4_w : 0, row_step = dz_q4_w ? step / dz_q4_w : 0;
  int64_t it = 0;
  for (int64_t v = v_first; v < nvox; v += U * step, it += U) {
    float4 xq[U][QI], gq[U][QO];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t vu = v + u * step;
      if (vu < nvox) {
#pragma unroll
        for (int q = 0; q < QI; ++q) xq[u][q] = x4[vu * QI + q];
#pragma unroll
        for (int q = 0; q < QO; ++q) gq[u][q] = dz_q4_w ? g4[((row_first + (it + u) * row_step) * QO + q) * dz_q4_w + qw] : g4[vu * QO + q];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (v + u * step >= nvox) break;
      float xv[CIN], gv[COUT];
#pragma unroll
      for (int q = 0; q < QI; ++q) { const float4 t = xq[u][q]; xv[4 * q] = t.x; xv[4 * q + 1] = t.y; xv[4 * q + 2] = t.z; xv[4 * q + 3] = t.w; }
#pragma unroll
      for (int q = 0; q < QO; ++q) { const float4 t = gq[u][q]; gv[4 * q] = t.x; gv[4 * q + 1] = t.y; gv[4 * q + 2] = t.z; gv[4 * q + 3] = t.w; }
#pragma unroll
      for (int i = 0; i < CIN; ++i)
#pragma unroll
        for (int j = 0; j < COUT; ++j) acc[i][j] += xv[i] * gv[j];
#pragma unroll
      for (int j = 0; j < COUT; ++j) bs[j] += gv[j];
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < CIN; ++i)
#pragma unroll
    for (int j = 0; j < COUT; ++j) {
      float t = acc[i][j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
      if (lane == 0) sh[wv][i * COUT + j] = t;
    }
#pragma unroll
  for (int j = 0; j < COUT; ++j) {
    float t = bs[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (lane == 0) sh[wv][NACC + j] = t;
  }
  __syncthreads();
  const int total = NACC + (with_bias ? COUT : 0);
  float* out = partial + (size_t)blockIdx.x * total;
  if ((int)threadIdx.x < total) out[threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// stride-1 convs (3x3x3 and 1x1x1).  Returns 1 launched (partial = [groups][taps*Cin*Cout (+ Cout bias sums)]),
// 0 unsupported shape, <0 error.
int launch_conv_dw_tile(const float* x, const float* dz, float* partial, int B, int D, int Cin, int Cout, int ksize,
                        int with_bias, hipStream_t s, int x_q4, int dz_q4) {
  if (D % 16) return 0;
  const int g = conv_dw_tile_groups(B, D);
  if (x_q4 || dz_q4) {
    // Q4 operands (the training step's 64^3 stage): exactly the kernels of that stage read them — 1 <-> 16 channels (edge),
    // 4 -> 8 (3^3 and 1^3: dz has 8 channels; x, 4 channels, is the same in both layouts); 16 -> 4 goes through
    // launch_conv_dw_pair.  Anything else would silently read the wrong voxels: refused.
    if (dw_mfma_enabled() && dw_edge_enabled() && ksize == 3 && Cin == 16 && Cout == 1 && x_q4) {
      hipLaunchKernelGGL(conv_dw_mfma_edge_kernel<0>, dim3(g), dim3(256), 0, s, x, dz, partial, B, D, with_bias, 1);
    } else if (dw_mfma_enabled() && dw_edge_enabled() && ksize == 3 && Cin == 1 && Cout == 16 && dz_q4) {
      hipLaunchKernelGGL(conv_dw_mfma_edge_kernel<1>, dim3(g), dim3(256), 0, s, x, dz, partial, B, D, with_bias, 1);
    } else if (dw_mfma_enabled() && ksize == 3 && Cin == 4 && Cout == 8 && D == 64 && dz_q4 && !x_q4) {
      hipLaunchKernelGGL((conv_dw_mfma_4xn_kernel<8>), dim3(g), dim3(256), 0, s, x, dz, partial, B, D, with_bias, 1);
    } else if (ksize == 1 && Cin == 4 && Cout == 8 && dz_q4 && !x_q4) {
      if (256 % D) { set_error("weight gradient of a 1x1x1 layer on a Q4 operand: D=%d must divide 256", D); return -1; }
      hipLaunchKernelGGL((conv_dw_1x1_kernel<4, 8>), dim3(g), dim3(256), 0, s, x, dz, partial, (int64_t)B * D * D * D, with_bias, D);
    } else {
      set_error("weight gradient: no kernel reads Q4 operands for Cin=%d Cout=%d k=%d D=%d (x_q4=%d dz_q4=%d)", Cin, Cout, ksize, D, x_q4, dz_q4);
      return -1;
    }
    const int rc = launch_ok("conv_dw kernel (Q4 operand)");
    return rc ? rc : 1;
  }
#define TRY(ck, co, ks)                                                                           \
  if (ksize == ks && ((Cin >= 16 && ck == 16) || Cin == ck) && Cin % ck == 0 && Cout == co)      \
    return run_dw<ck, co, ks>(x, dz, partial, B, D, Cin, g, with_bias, s);
  static const bool dw1 = !(getenv("PCGC_DW_1X1") && atoi(getenv("PCGC_DW_1X1")) == 0);           // experiment knob
  if (dw1 && ksize == 1 && ((Cin == 16 && Cout == 4) || (Cin == 4 && Cout == 8))) {
    const int64_t nvox = (int64_t)B * D * D * D;
    if (Cin == 16) hipLaunchKernelGGL((conv_dw_1x1_kernel<16, 4>), dim3(g), dim3(256), 0, s, x, dz, partial, nvox, with_bias);
    else hipLaunchKernelGGL((conv_dw_1x1_kernel<4, 8>), dim3(g), dim3(256), 0, s, x, dz, partial, nvox, with_bias);
    const int rc = launch_ok("conv_dw_1x1_kernel");
    return rc ? rc : 1;
  }
  static const bool slide = !(getenv("PCGC_DW_SLIDE") && atoi(getenv("PCGC_DW_SLIDE")) == 0);     // experiment knob
#define SLIDE(ck, co, wseg)                                                                       \
  if (slide && ksize == 3 && ((Cin >= 16 && ck == 16) || Cin == ck) && Cin % ck == 0 && Cout == co) \
    return run_dw_slide<ck, co, wseg>(x, dz, partial, B, D, Cin, g, with_bias, s);
  if (dw_mfma_enabled() && ksize == 3 && Cin == 4 && (Cout == 4 || Cout == 8) && D == 64) {
    // g workgroups (what the caller sized `partial` for) over B * 16 * 16 tiles of 4 x 4 x 64 voxels; a workgroup without a
    // tile writes a partial of zeros
    if (Cout == 4) hipLaunchKernelGGL((conv_dw_mfma_4xn_kernel<4>), dim3(g), dim3(256), 0, s, x, dz, partial, B, D, with_bias);
    else hipLaunchKernelGGL((conv_dw_mfma_4xn_kernel<8>), dim3(g), dim3(256), 0, s, x, dz, partial, B, D, with_bias);
    const int rc = launch_ok("conv_dw_mfma_4xn_kernel");
    return rc ? rc : 1;
  }
  if (dw_mfma_enabled() && ksize == 3 && Cin % 16 == 0 && (Cout == 4 || (Cout == 8 && dw_mfma32_enabled()))) {
    if (Cout == 4) hipLaunchKernelGGL(conv_dw_mfma_16xn_kernel<4>, dim3(g, Cin / 16), dim3(256), 0, s, x, dz, partial, B, D, Cin, with_bias);
    else hipLaunchKernelGGL(conv_dw_mfma_16xn_kernel<8>, dim3(g, Cin / 16), dim3(256), 0, s, x, dz, partial, B, D, Cin, with_bias);
    const int rc = launch_ok("conv_dw_mfma_16xn_kernel");
    return rc ? rc : 1;
  }
  if (dw_mfma_enabled() && dw_mfma32_enabled() && ksize == 3 && Cin == 8) {
    if (Cout == 16) return run_dw_mfma<16, 1, 8>(x, dz, partial, B, D, Cin, g, with_bias, s);
    // (8 -> 8 with half of the columns idle: 25 us per 8 cubes of 32^3 against 23 us for the sliding VALU kernel — not used)
  }
  SLIDE(4, 4, 4) SLIDE(4, 8, 4) SLIDE(8, 4, 4) SLIDE(4, 16, 8) SLIDE(8, 8, 8) SLIDE(16, 4, 8) SLIDE(8, 16, 16) SLIDE(16, 8, 16)
#undef SLIDE
  if (dw_mfma_enabled() && ksize == 3 && Cin % 16 == 0) {
    if (Cout == 16) return run_dw_mfma<16, 1>(x, dz, partial, B, D, Cin, g, with_bias, s);
    if (Cout == 32) return run_dw_mfma<32, 1>(x, dz, partial, B, D, Cin, g, with_bias, s);
    if (Cout == 64) return run_dw_mfma<64, 1>(x, dz, partial, B, D, Cin, g, with_bias, s);
  }
  if (dw_mfma_enabled() && dw_edge_enabled() && ksize == 3 && ((Cin == 16 && Cout == 1) || (Cin == 1 && Cout == 16))) {
    if (Cout == 1) hipLaunchKernelGGL(conv_dw_mfma_edge_kernel<0>, dim3(g), dim3(256), 0, s, x, dz, partial, B, D, with_bias);
    else hipLaunchKernelGGL(conv_dw_mfma_edge_kernel<1>, dim3(g), dim3(256), 0, s, x, dz, partial, B, D, with_bias);
    const int rc = launch_ok("conv_dw_mfma_edge_kernel");
    return rc ? rc : 1;
  }
  TRY(1, 16, 3) TRY(16, 1, 3)
  TRY(4, 4, 3) TRY(4, 8, 3) TRY(4, 16, 3)
  TRY(8, 4, 3) TRY(8, 8, 3) TRY(8, 16, 3) TRY(8, 32, 3)
  TRY(16, 4, 3) TRY(16, 8, 3) TRY(16, 16, 3) TRY(16, 32, 3) TRY(16, 64, 3)
  TRY(16, 4, 1) TRY(4, 8, 1) TRY(16, 8, 1) TRY(8, 16, 1) TRY(16, 16, 1) TRY(16, 32, 1)
#undef TRY
  return 0;
}

// conv1_1 (3x3x3, Cin -> Cout) and conv2_1 (1x1x1, Cin -> Cout) of a VRN block read the same input: both weight
// gradients in one pass over it (conv_dw_mfma_16xn_kernel<COUT, true>).  partial / partial1 as launch_conv_dw_tile would
// lay them out for the two layers.  Returns 1 launched, 0 unsupported shape (the caller takes the layers one by one).
bool conv_dw_pair_supported(int D, int Cin, int Cout) {
  static const bool on = !(getenv("PCGC_DW_PAIR") && atoi(getenv("PCGC_DW_PAIR")) == 0);          // experiment knob
  return on && dw_mfma_enabled() && D % 16 == 0 && Cin % 16 == 0 && (Cout == 4 || (Cout == 8 && dw_mfma32_enabled()));
}
int launch_conv_dw_pair(const float* x, const float* dz3, const float* dz1, float* partial3, float* partial1, int B, int D,
                        int Cin, int Cout, int with_bias, hipStream_t s, int x_q4) {
  if (!conv_dw_pair_supported(D, Cin, Cout)) return 0;
  const int g = conv_dw_tile_groups(B, D);
  if (Cout == 4)
    hipLaunchKernelGGL((conv_dw_mfma_16xn_kernel<4, true>), dim3(g, Cin / 16), dim3(256), 0, s, x, dz3, partial3, B, D, Cin, with_bias, dz1, partial1, x_q4);
  else
    hipLaunchKernelGGL((conv_dw_mfma_16xn_kernel<8, true>), dim3(g, Cin / 16), dim3(256), 0, s, x, dz3, partial3, B, D, Cin, with_bias, dz1, partial1, x_q4);
  const int rc = launch_ok("conv_dw_mfma_16xn_kernel<pair>");
  return rc ? rc : 1;
}

// The recorded calls, equal (kernel, shape) together as the jobs of one launch (at most kDwBatchMax per launch); a job's
// workgroups do exactly what its own launch would have done (same tiles, same order, same partial buffer).
int launch_dw_calls(const std::vector<DwCall>& calls, hipStream_t s) {
  std::vector<char> done(calls.size(), 0);
  for (size_t i = 0; i < calls.size(); ++i) {
    if (done[i]) continue;
    const DwCall& c = calls[i];
    DwBatch b;
    for (size_t k = i; k < calls.size() && b.n < kDwBatchMax; ++k) {
      const DwCall& o = calls[k];
      if (done[k] || o.kind != c.kind || o.B != c.B || o.D != c.D || o.Cin != c.Cin || o.groups != c.groups || o.with_bias != c.with_bias) continue;
      b.x[b.n] = o.x; b.dz[b.n] = o.dz; b.partial[b.n] = o.partial; ++b.n;
      done[k] = 1;
    }
    const dim3 grid(c.groups, c.Cin / 16, b.n);
    switch (c.kind) {
      case 0: hipLaunchKernelGGL((conv_dw_mfma_kernel<16, 1, 16>), grid, dim3(256), 0, s, c.x, c.dz, c.partial, c.B, c.D, c.Cin, c.with_bias, 0, b); break;
      case 1: hipLaunchKernelGGL((conv_dw_mfma_kernel<32, 1, 16>), grid, dim3(256), 0, s, c.x, c.dz, c.partial, c.B, c.D, c.Cin, c.with_bias, 0, b); break;
      case 2: hipLaunchKernelGGL((conv_dw_mfma_kernel<64, 1, 16>), grid, dim3(256), 0, s, c.x, c.dz, c.partial, c.B, c.D, c.Cin, c.with_bias, 0, b); break;
      case 3: hipLaunchKernelGGL((conv_dw_tile_kernel<16, 16, 1>), grid, dim3(256), 0, s, c.x, c.dz, c.partial, c.B, c.D, c.Cin, c.with_bias, b); break;
      case 4: hipLaunchKernelGGL((conv_dw_tile_kernel<16, 32, 1>), grid, dim3(256), 0, s, c.x, c.dz, c.partial, c.B, c.D, c.Cin, c.with_bias, b); break;
      default: set_error("launch_dw_calls: unknown kind %d", c.kind); return -1;
    }
  }
  return launch_ok("batched weight-gradient kernels");
}

}  // namespace pcgc
