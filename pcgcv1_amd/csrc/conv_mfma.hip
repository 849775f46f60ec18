// Implicit-GEMM 3-D convolutions on the gfx950 fp32 matrix cores.
//
// Replaces tf.keras.layers.Conv3D / Conv3DTranspose (models/model_voxception.py:
// 21-54, 83-122, 153-192, 224-244, 263-297) for NDHWC fp32 tensors.
//
// Mapping onto v_mfma_f32_16x16x4_f32 (exact fp32, D = A*B + C, one wave):
//   M (rows of D)  = 16 output "rows"     -> A operand = weights
//   N (cols of D)  = 16 output voxels consecutive along W -> B operand = activations
//   K              = 4 input channels of one filter tap
// With this orientation lane (j = lane&15, g = lane>>4) ends up holding rows
// 4g..4g+3 of voxel j — for Cout >= 4 that is one float4 of the NDHWC output, so
// the epilogue is a 16-byte store per lane — and the activation operand of lane
// (j,g) is VEC = min(Cin,16)/4 consecutive input channels of voxel j (+tap): one
// ds_read_b128/b64/b32 from an NDHWC tile staged in LDS with its halo.
//
// Row packing.  Most of the network's MACs sit in layers with 4 or 8 output
// channels (VRN conv1_1 C->C/4, conv2_2 C/4->C/4, ... at 64^3), which would fill
// only 4 or 8 of the 16 MFMA rows.  Those layers are run as the mathematically
// identical convolution that produces a QD x QH patch of output rows at once:
//   rows   = (qd, qh, co)                       QD*QH*CoutP of them  (16 = one M tile)
//   taps   = (rd, rh, kw) in (QD+2) x (QH+2) x 3, stride (QD, QH, 1)
//   W'[rd,rh,kw][ci][(qd,qh,co)] = W[rd-qd, rh-qh, kw][ci][co]   (0 when rd-qd or rh-qh is outside 0..2)
// i.e. a depth-to-space arrangement of the outputs.  A 16->4 layer then needs 48
// MFMAs per 16 outputs instead of 108; every output is still the same fixed-order
// fp32 sum (zeros added exactly), so results stay batch- and placement-invariant.
//
// Workgroup = 256 threads = 4 waves; it owns GD x GH patches x 16 voxels along W
// and all output channels; wave w owns GD*GH/4 patches.  Input channels go through
// LDS in chunks of 16.  No atomics, no split-K.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "mfma_common.h"

namespace pcgc {

// ---------------------------------------------------------------------------
// stride-1 (KS = 1 or 3, optionally row-packed) and stride-2 (KS = 3) convolution
//   CIN    input channels read (multiple of 4; chunks of 16)
//   COUTP  output channels padded to a multiple of 4 (rows per patch position)
//   QD,QH  patch of output rows produced together (1,1 = plain)
//   KS,S   true kernel size / stride
//   GD,GH  patches per workgroup along D and H (GD*GH multiple of 4)
// ---------------------------------------------------------------------------
// floats of LDS the body needs for its input tile
template <int CIN, int QD, int QH, int KS, int S, int GD, int GH>
constexpr int conv_mfma_tile_floats() {
  constexpr bool PACKED = (S == 1 && KS == 3);
  constexpr int KD = PACKED ? QD + 2 : KS, KH = PACKED ? QH + 2 : KS, KW = KS;
  constexpr int SD = S == 2 ? 2 : QD, SH = S == 2 ? 2 : QH, SW = S;
  return ((GD - 1) * SD + KD) * ((GH - 1) * SH + KH) * (15 * SW + KW) * Chunk<CIN>::VS;
}
// blk / nblk: the workgroup's index among those of THIS layer (a launch may carry two layers: conv_mfma_pair_kernel)
template <int CIN, int COUTP, int QD, int QH, int KS, int S, int GD, int GH>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs& a, float* tile, int blk, int nblk) {
  using C = Chunk<CIN>;
  constexpr int CK = C::CK, NCH = C::NCH, VEC = C::VEC, VS = C::VS;
  constexpr bool PACKED = (S == 1 && KS == 3);
  constexpr int KD = PACKED ? QD + 2 : KS, KH = PACKED ? QH + 2 : KS, KW = KS;   // effective taps
  constexpr int SD = S == 2 ? 2 : QD, SH = S == 2 ? 2 : QH, SW = S;               // effective strides
  constexpr int PAD = PACKED ? 1 : 0;
  constexpr int MROWS = QD * QH * COUTP;
  constexpr int MT = (MROWS + 15) / 16;
  constexpr int NT = GD * GH / 4;
  constexpr int ID = (GD - 1) * SD + KD, IH = (GH - 1) * SH + KH, IW = 15 * SW + KW;
  constexpr int NVOX = ID * IH * IW;
  constexpr int TAPS = KD * KH * KW;
  constexpr int TD = GD * QD, TH = GH * QH;                                       // output rows per workgroup
  static_assert(S == 1 || (QD == 1 && QH == 1), "row packing is for stride-1 layers");
  static_assert(NT >= 1 && (GD * GH) % 4 == 0, "need a multiple of 4 patches per workgroup");
  static_assert(NVOX * VS == conv_mfma_tile_floats<CIN, QD, QH, KS, S, GD, GH>(), "tile size");

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tw = a.Dout / 16, th = a.Dout / TH, td = a.Dout / TD;
  int bid = xcd_remap(blk, nblk);
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  const int b = bid;
  const int od0 = tx * TD, oh0 = ty * TH, ow0 = tz * 16;
  const int id0 = od0 * S - PAD, ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* wl = a.w + (size_t)lane * VEC;                   // this lane's slot inside every packed block

  for (int cb = 0; cb < NCH; ++cb) {
    if (cb) __syncthreads();
    // ---- stage the input chunk (zero outside the volume = 'same' padding) ----
    const int ch0 = a.x_co + cb * CK;                                                   // first channel of the chunk
    stage_tile<ID, IH, IW, CK / 4, VS>(tile, a.x + (int64_t)b * a.Din * a.Din * a.Din * a.x_cs + (a.x_q4 ? (ch0 >> 2) * a.Din * 4 : ch0),
                                       a.Din, a.x_cs, id0, ih0, iw0, a.x_q4 != 0);
    __syncthreads();
    // ---- taps ----
    const float* wc = wl + (size_t)cb * TAPS * MT * 64 * VEC;
#pragma unroll 1
    for (int kd = 0; kd < KD; ++kd) {
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const int tap = (kd * KH + kh) * KW + kw;
          float av[MT][4];
#pragma unroll
          for (int m = 0; m < MT; ++m) read_vec<VEC>(wc + (size_t)(tap * MT + m) * 64 * VEC, av[m]);
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            const int nt = wv * NT + i;
            const int pd = nt / GH, ph = nt % GH;
            const int pos = ((pd * SD + kd) * IH + (ph * SH + kh)) * IW + (j * SW + kw);
            float bv[4];
            read_vec<VEC>(&tile[pos * VS + VEC * g], bv);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int r = 0; r < VEC; ++r) acc[m][i] = mfma4(av[m][r], bv[r], acc[m][i]);
          }
        }
      }
    }
  }
  // ---- epilogue: row rho = 16m + 4g + r  ->  (patch row q = rho / COUTP, channel rho % COUTP) ----
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int nt = wv * NT + i;
    const int pd = nt / GH, ph = nt % GH;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int rho = m * 16 + 4 * g;
      if (rho < MROWS) {
        const int q = rho / COUTP, c0 = rho % COUTP;
        const int od = od0 + pd * QD + q / QH, oh = oh0 + ph * QH + q % QH;
        const int64_t vox = (((int64_t)b * a.Dout + od) * a.Dout + oh) * a.Dout + ow0 + j;
        store_acc(a, vox, c0, acc[m][i]);
      }
    }
  }
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// The 2 x 2-row tiles of a SMALL launch (a training batch at 16^3: 512 workgroups, two per CU) with every global load taken off
// the critical path.  conv_mfma_body's wave waits for a memory round trip once per input chunk (the tile) and once per kd (nine
// taps of the filter) and does 0.5 us of MFMAs (COUT = 16) in between: at two to four waves per SIMD the kernel ran at 0.25-0.39 of
// the matrix peak, the memory pipe idle.  Here (everything unrolled, indices compile-time): the filter arrives in units of one
// (kd, kh) row of taps through a register ring PD units ahead of its use — across chunk boundaries, it does not depend on the
// tile — and the next chunk's tile is fetched into registers before the current chunk's MFMAs and written to LDS after them.
// The same MFMAs in the same order on every accumulator: bit-identical to conv_mfma_body<CIN, COUT, 1, 1, KS, 1, 2, 2>.
// S = 2: the stride-2 convolution (down_2 / the reverse of up_1), tile rows 2 apart, no padding on the low side.
// G = 4: the 4 x 4-row tiles of the large launches, four patch rows per wave.  Tried for the inference path's 16^3 layers (three
// workgroups per CU there): round trip 39.93 / 39.68 / 39.66 ms against 39.16 / 39.90 / 39.64 with conv_mfma_body — the
// occupancy already hides those round trips.  Not instantiated; the small launches are where the form pays.
template <int CIN, int COUT, int KS, int S = 1, int G = 2>
__device__ __forceinline__ void conv_mfma_small_body(const ConvArgs& a, float* tile, int blk, int nblk) {
  using C = Chunk<CIN>;
  constexpr int CK = C::CK, NCH = C::NCH, VEC = C::VEC, VS = C::VS;
  static_assert(VEC == 4 && COUT % 16 == 0 && (KS == 1 || KS == 3) && (S == 1 || (S == 2 && KS == 3)), "16 | CIN, 16 | COUT");
  constexpr int PAD = (S == 1 && KS == 3) ? 1 : 0;
  constexpr int MT = COUT / 16;
  constexpr int NT = G * G / 4;                                                  // patch rows per wave
  constexpr int ID = (G - 1) * S + KS, IH = (G - 1) * S + KS, IW = 15 * S + KS;
  constexpr int TAPS = KS * KS * KS, NU = KS * KS, NTOT = NCH * NU;            // units: one (kd, kh) row of KS taps
  constexpr int PD0 = G == 4 ? (MT == 1 ? 2 : 1) : (MT == 1 ? 6 : (MT == 2 ? 3 : (S == 2 ? 1 : 2)));
  constexpr int PD = PD0 < NTOT ? PD0 : NTOT;                                  // units in flight ahead of the MFMAs
  constexpr int RING = PD + 1;
  constexpr bool TPF = S == 1 && G == 2;    // the next chunk's tile through registers (stride 2: 84 registers, 4 x 4 rows: 72 — occupancy is worth more)
  constexpr int Q = CK / 4, E = IW * Q, KP = (E + 63) / 64, NROW = ID * IH, RPW = (NROW + 3) / 4;
  static_assert(IW * ID * IH * VS == conv_mfma_tile_floats<CIN, 1, 1, KS, S, G, G>(), "tile geometry");

  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int tw = a.Dout / 16, th = a.Dout / G, td = a.Dout / G;
  int bid = xcd_remap(blk, nblk);
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  const int b = bid;
  const int od0 = tx * G, oh0 = ty * G, ow0 = tz * 16;
  const int id0 = od0 * S - PAD, ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
  f32x4 acc[MT][NT];                                                             // the wave's patch rows nt = wv * NT + i: (nt / G, nt % G)
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wl = a.w + (size_t)lane * VEC;
  float av[RING][KS][MT][4];
  auto load_unit = [&](auto T_) {
    constexpr int t = decltype(T_)::value, cb = t / NU, u = t % NU;
#pragma unroll
    for (int kw = 0; kw < KS; ++kw)
#pragma unroll
      for (int m = 0; m < MT; ++m) read_vec<VEC>(wl + (size_t)((cb * TAPS + u * KS + kw) * MT + m) * 64 * VEC, av[t % RING][kw][m]);
  };
  // the tile's rows as stage_tile splits them: wave w takes rows w, w + 4, ..., a lane the columns lane, lane + 64
  const float* xb = a.x + (int64_t)b * a.Din * a.Din * a.Din * a.x_cs + a.x_co;
  float4 xr[RPW][KP];
  int gofs[KP], lofs[KP];
  bool ok[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int c = lane + 64 * k, vox = c / Q, q = c & (Q - 1);
    gofs[k] = vox * a.x_cs + q * 4;
    lofs[k] = vox * VS + q * 4;
    ok[k] = (c < E) && ((unsigned)(iw0 + vox) < (unsigned)a.Din);
  }
  auto tile_load = [&](int cb) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int r = wv + 4 * i, zd = r / IH, zh = r - zd * IH;
      const int gd = id0 + zd, gh = ih0 + zh;
      const bool row_ok = (r < NROW) && ((unsigned)gd < (unsigned)a.Din) && ((unsigned)gh < (unsigned)a.Din);
      const int row_off = ((gd * a.Din + gh) * a.Din + iw0) * a.x_cs + cb * CK;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        xr[i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row_ok && ok[k]) xr[i][k] = *reinterpret_cast<const float4*>(xb + (row_off + gofs[k]));
      }
    }
  };
  auto tile_store = [&]() {
#pragma unroll
    for (int i = 0; i < RPW; ++i)
#pragma unroll
      for (int k = 0; k < KP; ++k)
        if (wv + 4 * i < NROW && lane + 64 * k < E) *reinterpret_cast<float4*>(&tile[(wv + 4 * i) * (IW * VS) + lofs[k]]) = xr[i][k];
  };

  const float* tb[NT];                                                           // the lane's voxel of patch row i at tap (0, 0, 0)
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int nt = wv * NT + i, pd = nt / G, ph = nt % G;
    tb[i] = tile + ((pd * S * IH + ph * S) * IW + j * S) * VS + VEC * g;
  }
  tile_load(0);
  static_for<PD>([&](auto T_) { load_unit(T_); });
  tile_store();
  __syncthreads();
  static_for<NCH>([&](auto CB_) {
    constexpr int cb = decltype(CB_)::value;
    if constexpr (TPF && cb + 1 < NCH) {
      tile_load(cb + 1);
      __builtin_amdgcn_sched_barrier(0);            // the scheduler would sink these loads to the LDS stores behind the MFMAs
    }
    static_for<NU>([&](auto U_) {
      constexpr int u = decltype(U_)::value, t = cb * NU + u, kd = u / KS, kh = u % KS;
      if constexpr (t + PD < NTOT) {
        load_unit(std::integral_constant<int, t + PD>{});
        __builtin_amdgcn_sched_barrier(0);          // ... and these to their first use, PD units later
      }
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          float bv[4];
          read_vec<VEC>(tb[i] + ((kd * IH + kh) * IW + kw) * VS, bv);       // a compile-time offset from the patch row's base
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < VEC; ++r) acc[m][i] = mfma4(av[t % RING][kw][m][r], bv[r], acc[m][i]);
        }
      }
      if constexpr (G > 2) __builtin_amdgcn_sched_barrier(0);   // keeps the scheduler from hoisting later units' LDS reads (296 registers)
    });
    if constexpr (cb + 1 < NCH) {
      __syncthreads();
      if constexpr (!TPF) tile_load(cb + 1);
      tile_store();
      __syncthreads();
    }
  });
  // store_acc's arithmetic in store_acc's order, with every load of a patch row issued before the first use (store_acc one
  // accumulator at a time waits for bias / residual / add_to / mask one after the other: up to 3 MT round trips)
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int nt = wv * NT + i, pd = nt / G, ph = nt % G;
    const int64_t vox = (((int64_t)b * a.Dout + od0 + pd) * a.Dout + oh0 + ph) * a.Dout + ow0 + j;
    const int64_t eo = vox * a.y_cs + a.y_co + 4 * g;
    float4 bq[MT], rq[MT], aq[MT], mq[MT];
    if (a.bias) {
#pragma unroll
      for (int m = 0; m < MT; ++m) bq[m] = *reinterpret_cast<const float4*>(a.bias + m * 16 + 4 * g);
    }
    if (a.res) {
#pragma unroll
      for (int m = 0; m < MT; ++m) rq[m] = *reinterpret_cast<const float4*>(a.res + eo + m * 16);
    }
    if (a.add_to) {
#pragma unroll
      for (int m = 0; m < MT; ++m) aq[m] = *reinterpret_cast<const float4*>(a.add_to + eo + m * 16);
    }
    if (a.mask) {
#pragma unroll
      for (int m = 0; m < MT; ++m) mq[m] = *reinterpret_cast<const float4*>(a.mask + eo + m * 16);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v[4] = {acc[m][i][0], acc[m][i][1], acc[m][i][2], acc[m][i][3]};
      if (a.bias) { v[0] += bq[m].x; v[1] += bq[m].y; v[2] += bq[m].z; v[3] += bq[m].w; }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (a.relu) v[r] = fmaxf(v[r], 0.f);
        if (a.absval) v[r] = fmaxf(fabsf(v[r]), a.lower_bound);
      }
      if (a.res) {
        v[0] = fmaxf(rq[m].x + v[0], 0.f); v[1] = fmaxf(rq[m].y + v[1], 0.f);
        v[2] = fmaxf(rq[m].z + v[2], 0.f); v[3] = fmaxf(rq[m].w + v[3], 0.f);
      }
      if (a.add_to) { v[0] += aq[m].x; v[1] += aq[m].y; v[2] += aq[m].z; v[3] += aq[m].w; }
      if (a.mask) {
        v[0] = mq[m].x > 0.f ? v[0] : 0.f; v[1] = mq[m].y > 0.f ? v[1] : 0.f;
        v[2] = mq[m].z > 0.f ? v[2] : 0.f; v[3] = mq[m].w > 0.f ? v[3] : 0.f;
      }
      *reinterpret_cast<float4*>(a.y + eo + m * 16) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// one layer's 2 x 2-row tile: PIPE = the software-pipelined form (PCGC_CONV_PIPE=0 at run time: conv_mfma_body, for comparisons)
template <int CIN, int COUT, int KS, bool PIPE>
__device__ __forceinline__ void small_tile(const ConvArgs& a, float* tile, int blk, int nblk) {
  if constexpr (PIPE) conv_mfma_small_body<CIN, COUT, KS>(a, tile, blk, nblk);
  else conv_mfma_body<CIN, COUT, 1, 1, KS, 1, 2, 2>(a, tile, blk, nblk);
}

template <int CIN, int COUTP, int QD, int QH, int KS, int S, int GD, int GH>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[conv_mfma_tile_floats<CIN, QD, QH, KS, S, GD, GH>()];
  conv_mfma_body<CIN, COUTP, QD, QH, KS, S, GD, GH>(a, tile, blockIdx.x, gridDim.x);
}
template <int CIN, int COUT, int KS, int S = 1, int G = 2>
__global__ void __launch_bounds__(256) conv_mfma_small_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[conv_mfma_tile_floats<CIN, 1, 1, KS, S, G, G>()];
  conv_mfma_small_body<CIN, COUT, KS, S, G>(a, tile, blockIdx.x, gridDim.x);
}

// TWO independent stride-1 layers in one launch (the training step's 16^3 blocks at a batch of 8 cubes: each of their layers
// alone is 128-512 workgroups of 11-37 us; conv1_1 | conv2_1 read the same input, conv1_2 | conv2_2 and their adjoints
// neither read nor write each other's tensors): workgroups [0, na) run layer a, [na, na + nb) layer b, each exactly as its own
// launch would (same tiles, same sums).  KSA / KSB = the two kernel sizes, everything else of the tile geometry is shared.
template <int CINA, int COUTA, int KSA, int CINB, int COUTB, int KSB, bool PIPE>
__global__ void __launch_bounds__(256) conv_mfma_pair_kernel(ConvArgs a, ConvArgs b, int na, int nb) {
  constexpr int FA = conv_mfma_tile_floats<CINA, 1, 1, KSA, 1, 2, 2>(), FB = conv_mfma_tile_floats<CINB, 1, 1, KSB, 1, 2, 2>();
  __shared__ __attribute__((aligned(16))) float tile[FA > FB ? FA : FB];
  if ((int)blockIdx.x < na) small_tile<CINA, COUTA, KSA, PIPE>(a, tile, blockIdx.x, na);
  else small_tile<CINB, COUTB, KSB, PIPE>(b, tile, blockIdx.x - na, nb);
}

// Two layers in one launch where the second needs from the first only what the SAME workgroup wrote: layer b is 1x1x1 (no halo)
// on its own input and reads layer a's output at its own output voxels (b.add_to == a.y: the reverse of a block's two input layers,
// dx = m * (m * (dx + conv1_1^T(dt11)) + conv2_1^T(dt21)) in place — the sums and their order are those of the two launches).
template <int CINA, int COUTA, int KSA, int CINB, int COUTB, int KSB, bool PIPE>
__global__ void __launch_bounds__(256) conv_mfma_chain_kernel(ConvArgs a, ConvArgs b) {
  static_assert(KSB == 1, "the second layer may not read its neighbours' tiles");
  constexpr int FA = conv_mfma_tile_floats<CINA, 1, 1, KSA, 1, 2, 2>(), FB = conv_mfma_tile_floats<CINB, 1, 1, KSB, 1, 2, 2>();
  __shared__ __attribute__((aligned(16))) float tile[FA > FB ? FA : FB];
  small_tile<CINA, COUTA, KSA, PIPE>(a, tile, blockIdx.x, gridDim.x);
  __syncthreads();                                  // layer a's stores are visible to the workgroup; the LDS tile is free again
  small_tile<CINB, COUTB, KSB, PIPE>(b, tile, blockIdx.x, gridDim.x);
}

// The last layer of a block's second path (conv2_3, 1x1x1) and the block's merge out = relu(x + [t12 | t23]) (train.hip
// vrn_merge_kernel: the same one add and one maximum per value) on the workgroup's own 2 x 2 rows of 16 voxels.
template <int CIN, int COUT, int KS, bool PIPE>
__global__ void __launch_bounds__(256) conv_mfma_merge_kernel(ConvArgs a, MergeArgs m) {
  __shared__ __attribute__((aligned(16))) float tile[conv_mfma_tile_floats<CIN, 1, 1, KS, 1, 2, 2>()];
  small_tile<CIN, COUT, KS, PIPE>(a, tile, blockIdx.x, gridDim.x);
  __syncthreads();                                  // t23 of this tile is visible to the workgroup
  const int tw = a.Dout / 16, th = a.Dout / 2, td = a.Dout / 2;
  int bid = xcd_remap(blockIdx.x, gridDim.x);      // the tile conv_mfma_body just took
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  const int Q = m.C / 4, hq = Q / 2;
  const float4* x4 = reinterpret_cast<const float4*>(m.x);
  const float4* a4 = reinterpret_cast<const float4*>(m.t12);
  const float4* b4 = reinterpret_cast<const float4*>(a.y);
  float4* o4 = reinterpret_cast<float4*>(m.out);
  for (int i = threadIdx.x; i < 64 * Q; i += 256) {
    const int v = i / Q, q = i - v * Q;
    const int64_t vox = (((int64_t)bid * a.Dout + tx * 2 + (v >> 5)) * a.Dout + ty * 2 + ((v >> 4) & 1)) * a.Dout + tz * 16 + (v & 15);
    const float4 r = q < hq ? a4[vox * hq + q] : b4[vox * hq + q - hq];
    const float4 xv = x4[vox * Q + q];
    o4[vox * Q + q] = float4{fmaxf(xv.x + r.x, 0.f), fmaxf(xv.y + r.y, 0.f), fmaxf(xv.z + r.z, 0.f), fmaxf(xv.w + r.w, 0.f)};
  }
}

// ---------------------------------------------------------------------------
// stride-2 transposed convolution: y[o] = b + sum_{2i+k=o} x[i] W[k]
// The workgroup owns TD x TH x 16 INPUT voxels (+1 low-side halo) and the
// (2TD) x (2TH) x 32 outputs they map to, one output-parity class at a time:
// per axis an even output o=2i has taps k=0 (input i) and k=2 (input i-1), an
// odd output o=2i+1 has tap k=1 (input i).  All CIN channels are staged at once.
// ---------------------------------------------------------------------------
template <int CIN, int COUT, int TD, int TH>
__global__ void __launch_bounds__(256) tconv_mfma_kernel(ConvArgs a) {
  constexpr int NCH = CIN / 16;
  constexpr int VS = CIN + 4;
  constexpr int MT = (COUT + 15) / 16;
  constexpr int NT = TD * TH / 4;
  constexpr int ID = TD + 1, IH = TH + 1, IW = 17;
  constexpr int NVOX = ID * IH * IW;
  __shared__ __attribute__((aligned(16))) float tile[NVOX * VS];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tw = a.Din / 16, th = a.Din / TH, td = a.Din / TD;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  const int b = bid;
  const int id0 = tx * TD, ih0 = ty * TH, iw0 = tz * 16;

  stage_tile<ID, IH, IW, CIN / 4, VS>(tile, a.x + (int64_t)b * a.Din * a.Din * a.Din * a.x_cs + (a.x_q4 ? (a.x_co >> 2) * a.Din * 4 : a.x_co),
                                      a.Din, a.x_cs, id0 - 1, ih0 - 1, iw0 - 1, a.x_q4 != 0);
  __syncthreads();
  const float* wl = a.w + (size_t)lane * 4;

#pragma unroll 1
  for (int cls = 0; cls < 8; ++cls) {
    const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nd = pd ? 1 : 2, nh = ph ? 1 : 2, nw = pw ? 1 : 2;
#pragma unroll 1
    for (int td_ = 0; td_ < nd; ++td_) {
      const int kd = pd ? 1 : 2 * td_, offd = (kd == 2) ? 0 : 1;
#pragma unroll 1
      for (int th_ = 0; th_ < nh; ++th_) {
        const int kh = ph ? 1 : 2 * th_, offh = (kh == 2) ? 0 : 1;
#pragma unroll 1
        for (int tw_ = 0; tw_ < nw; ++tw_) {
          const int kw = pw ? 1 : 2 * tw_, offw = (kw == 2) ? 0 : 1;
          const int tap = (kd * 3 + kh) * 3 + kw;
#pragma unroll
          for (int cb = 0; cb < NCH; ++cb) {
            float av[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
              read_vec<4>(wl + (size_t)((cb * 27 + tap) * MT + m) * 256, av[m]);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
              const int nt = wv * NT + i;
              const int dd = nt / TH, hh = nt % TH;
              const int pos = ((dd + offd) * IH + (hh + offh)) * IW + (j + offw);
              float bv[4];
              read_vec<4>(&tile[pos * VS + cb * 16 + 4 * g], bv);
#pragma unroll
              for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][i] = mfma4(av[m][r], bv[r], acc[m][i]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int nt = wv * NT + i;
      const int dd = nt / TH, hh = nt % TH;
      const int od = 2 * (id0 + dd) + pd, oh = 2 * (ih0 + hh) + ph, ow = 2 * (iw0 + j) + pw;
      const int64_t vox = (((int64_t)b * a.Dout + od) * a.Dout + oh) * a.Dout + ow;
#pragma unroll
      for (int m = 0; m < MT; ++m) store_acc(a, vox, m * 16 + 4 * g, acc[m][i]);
    }
  }
}

// The same transposed convolution for a SMALL launch (up_1 / the reverse of down_2 on a training batch: 256 workgroups of
// tconv_mfma_kernel<64, 32, 2, 4>, one wave per SIMD, every tap's filter block fetched right before its MFMAs — 90 us for what
// the matrix pipe does in 23): 2 x 2-row tiles (twice the workgroups), the 27 (class, tap) steps unrolled with the filter
// arriving through a register ring PD taps ahead, and an epilogue that issues all its loads first.  Taps, channel chunks
// and k-steps in tconv_mfma_kernel's order on every accumulator: bit-identical.  (On the inference path's large launches of the
// same layer it changes nothing: round trip 39.42 / 39.52 / 39.24 ms against 38.92 / 39.19 / 39.28 — small launches only.)
struct TconvTap { int pd, ph, pw, tap, offd, offh, offw; bool first, last; };
constexpr TconvTap tconv_tap(int t) {
  int n = 0;
  for (int cls = 0; cls < 8; ++cls) {
    const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
    const int nd = pd ? 1 : 2, nh = ph ? 1 : 2, nw = pw ? 1 : 2;
    for (int td_ = 0; td_ < nd; ++td_)
      for (int th_ = 0; th_ < nh; ++th_)
        for (int tw_ = 0; tw_ < nw; ++tw_) {
          const int kd = pd ? 1 : 2 * td_, kh = ph ? 1 : 2 * th_, kw = pw ? 1 : 2 * tw_;
          if (n == t)
            return TconvTap{pd, ph, pw, (kd * 3 + kh) * 3 + kw, kd == 2 ? 0 : 1, kh == 2 ? 0 : 1, kw == 2 ? 0 : 1,
                            td_ == 0 && th_ == 0 && tw_ == 0, td_ == nd - 1 && th_ == nh - 1 && tw_ == nw - 1};
          ++n;
        }
  }
  return TconvTap{0, 0, 0, 0, 0, 0, 0, false, false};
}

template <int CIN, int COUT>
__global__ void __launch_bounds__(256) tconv_mfma_small_kernel(ConvArgs a) {
  constexpr int TD = 2, TH = 2, NCH = CIN / 16, VS = CIN + 4, MT = COUT / 16;
  static_assert(CIN % 16 == 0 && COUT % 16 == 0, "16 | CIN, 16 | COUT");
  constexpr int ID = TD + 1, IH = TH + 1, IW = 17, NVOX = ID * IH * IW;
  constexpr int PD = 2, RING = PD + 1;
  __shared__ __attribute__((aligned(16))) float tile[NVOX * VS];

  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int tw = a.Din / 16, th = a.Din / TH, td = a.Din / TD;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  const int b = bid;
  const int id0 = tx * TD, ih0 = ty * TH, iw0 = tz * 16;
  const int dd = wv / TH, hh = wv % TH;                                          // the wave's input row

  const float* wl = a.w + (size_t)lane * 4;
  float av[RING][NCH][MT][4];
  auto load_tap = [&](auto T_) {
    constexpr int t = decltype(T_)::value;
    constexpr TconvTap T = tconv_tap(t);
#pragma unroll
    for (int cb = 0; cb < NCH; ++cb)
#pragma unroll
      for (int m = 0; m < MT; ++m) read_vec<4>(wl + (size_t)((cb * 27 + T.tap) * MT + m) * 256, av[t % RING][cb][m]);
  };
  static_for<PD>([&](auto T_) { load_tap(T_); });
  stage_tile<ID, IH, IW, CIN / 4, VS>(tile, a.x + (int64_t)b * a.Din * a.Din * a.Din * a.x_cs + a.x_co, a.Din, a.x_cs, id0 - 1, ih0 - 1, iw0 - 1, false);
  __syncthreads();

  f32x4 acc[MT];
  static_for<27>([&](auto T_) {
    constexpr int t = decltype(T_)::value;
    constexpr TconvTap T = tconv_tap(t);
    if constexpr (T.first) {
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (t + PD < 27) {
      load_tap(std::integral_constant<int, t + PD>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    const int pos = ((dd + T.offd) * IH + (hh + T.offh)) * IW + (j + T.offw);
#pragma unroll
    for (int cb = 0; cb < NCH; ++cb) {
      float bv[4];
      read_vec<4>(&tile[pos * VS + cb * 16 + 4 * g], bv);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m] = mfma4(av[t % RING][cb][m][r], bv[r], acc[m]);
    }
    if constexpr (T.last) {                         // store_acc's arithmetic in its order, the loads issued together
      const int od = 2 * (id0 + dd) + T.pd, oh = 2 * (ih0 + hh) + T.ph, ow = 2 * (iw0 + j) + T.pw;
      const int64_t vox = (((int64_t)b * a.Dout + od) * a.Dout + oh) * a.Dout + ow;
      const int64_t eo = vox * a.y_cs + a.y_co + 4 * g;
      float4 bq[MT], rq[MT], aq[MT], mq[MT];
      if (a.bias) {
#pragma unroll
        for (int m = 0; m < MT; ++m) bq[m] = *reinterpret_cast<const float4*>(a.bias + m * 16 + 4 * g);
      }
      if (a.res) {
#pragma unroll
        for (int m = 0; m < MT; ++m) rq[m] = *reinterpret_cast<const float4*>(a.res + eo + m * 16);
      }
      if (a.add_to) {
#pragma unroll
        for (int m = 0; m < MT; ++m) aq[m] = *reinterpret_cast<const float4*>(a.add_to + eo + m * 16);
      }
      if (a.mask) {
#pragma unroll
        for (int m = 0; m < MT; ++m) mq[m] = *reinterpret_cast<const float4*>(a.mask + eo + m * 16);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        float v[4] = {acc[m][0], acc[m][1], acc[m][2], acc[m][3]};
        if (a.bias) { v[0] += bq[m].x; v[1] += bq[m].y; v[2] += bq[m].z; v[3] += bq[m].w; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (a.relu) v[r] = fmaxf(v[r], 0.f);
          if (a.absval) v[r] = fmaxf(fabsf(v[r]), a.lower_bound);
        }
        if (a.res) {
          v[0] = fmaxf(rq[m].x + v[0], 0.f); v[1] = fmaxf(rq[m].y + v[1], 0.f);
          v[2] = fmaxf(rq[m].z + v[2], 0.f); v[3] = fmaxf(rq[m].w + v[3], 0.f);
        }
        if (a.add_to) { v[0] += aq[m].x; v[1] += aq[m].y; v[2] += aq[m].z; v[3] += aq[m].w; }
        if (a.mask) {
          v[0] = mq[m].x > 0.f ? v[0] : 0.f; v[1] = mq[m].y > 0.f ? v[1] : 0.f;
          v[2] = mq[m].z > 0.f ? v[2] : 0.f; v[3] = mq[m].w > 0.f ? v[3] : 0.f;
        }
        *reinterpret_cast<float4*>(a.y + eo + m * 16) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  });
}

// ---------------------------------------------------------------------------
// plan: which instantiation (if any) takes a layer shape.  Shared by the weight
// packer and the launcher so that both agree on the operand layout.
// ---------------------------------------------------------------------------
struct Plan {
  bool ok;
  int coutp, qd, qh;     // padded channels per patch row, patch shape
};

static Plan plan_for(int Cin, int Cout, int ksize, int mode) {
  Plan p{false, 0, 1, 1};
  if (ksize != 1 && ksize != 3) return p;
  if (Cin % 4 || Cin > 64 || (Cin > 16 && Cin % 16)) return p;
  if (mode == 2) {
    if (ksize == 3 && Cout % 4 == 0 && Cin % 16 == 0) p = Plan{true, Cout, 1, 1};
    return p;
  }
  if (mode == 1) {
    if (ksize == 3 && Cout % 16 == 0) p = Plan{true, Cout, 1, 1};
    return p;
  }
  if (ksize == 1) {
    if (Cout % 4 == 0) p = Plan{true, Cout, 1, 1};
    return p;
  }
  if (Cout <= 4) return Plan{true, 4, 2, 2};
  if (Cout == 8) return Plan{true, 8, 1, 2};
  if (Cout % 16 == 0) return Plan{true, Cout, 1, 1};
  return p;
}

// weight packing: TF layout -> [chunk][tap'][mtile][lane][VEC] with
//   value = W'[tap'][ci = chunk*CK + VEC*(lane>>4) + r][row = mtile*16 + (lane&15)]
__device__ __forceinline__ float packed_weight(const float* w, int idx, int Cin, int Cout, int ksize, int mode, int coutp, int qd,
                                               int qh) {
  const bool packed = (mode == 0 && ksize == 3);
  const int KD = packed ? qd + 2 : ksize, KH = packed ? qh + 2 : ksize, KW = ksize;
  const int taps = KD * KH * KW;
  const int CK = Cin < 16 ? Cin : 16, VEC = CK / 4;
  const int mrows = qd * qh * coutp, MT = (mrows + 15) / 16;
  int t = idx;
  const int r = t % VEC; t /= VEC;
  const int lane = t % 64; t /= 64;
  const int m = t % MT; t /= MT;
  const int tap = t % taps; t /= taps;
  const int cb = t;
  const int ci = cb * CK + VEC * (lane >> 4) + r;
  const int row = m * 16 + (lane & 15);
  float v = 0.f;
  if (row < mrows) {
    const int q = row / coutp, co = row % coutp;
    const int rd = tap / (KH * KW), rh = (tap / KW) % KH, kw = tap % KW;
    const int kd = packed ? rd - q / qh : rd, kh = packed ? rh - q % qh : rh;
    if (co < Cout && kd >= 0 && kd < ksize && kh >= 0 && kh < ksize) {
      const int tt = (kd * ksize + kh) * ksize + kw;
      v = (mode == 2) ? w[((size_t)tt * Cout + co) * Cin + ci] : w[((size_t)tt * Cin + ci) * Cout + co];
    }
  }
  return v;
}

__global__ void pack_weights_kernel(const float* w, float* p, int Cin, int Cout, int ksize, int mode, int coutp, int qd,
                                    int qh, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  p[idx] = packed_weight(w, idx, Cin, Cout, ksize, mode, coutp, qd, qh);
}

// every weight preparation of a training step in one launch (train_plan.hip): block -> job by its first block
__global__ void __launch_bounds__(256) weight_jobs_kernel(const WeightJob* jobs, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {                                         // last job whose block0 <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const WeightJob j = jobs[lo];
  const int idx = ((int)blockIdx.x - j.block0) * 256 + threadIdx.x;
  if (idx >= j.total) return;
  if (j.kind == 0) {
    j.dst[idx] = packed_weight(j.src, idx, j.Cin, j.Cout, j.ksize, j.mode, j.coutp, j.qd, j.qh);
  } else {                                                  // w [K^3][Cin][Cout] -> [K^3][Cout][Cin], taps flipped (stride-1 adjoint)
    const int K = j.ksize, Cin = j.Cin, Cout = j.Cout;
    const int ci = idx % Cin, co = (idx / Cin) % Cout, tap = idx / (Cin * Cout);
    const int kw = tap % K, kh = (tap / K) % K, kd = tap / (K * K);
    const int ftap = ((K - 1 - kd) * K + (K - 1 - kh)) * K + (K - 1 - kw);
    j.dst[idx] = j.src[((size_t)ftap * Cin + ci) * Cout + co];
  }
}

size_t mfma_packed_floats(int Cin, int Cout, int ksize, int mode) {
  const Plan pl = plan_for(Cin, Cout, ksize, mode);
  if (!pl.ok) return 0;
  const bool packed = (mode == 0 && ksize == 3);
  const int taps = (packed ? pl.qd + 2 : ksize) * (packed ? pl.qh + 2 : ksize) * ksize;
  const int MT = (pl.qd * pl.qh * pl.coutp + 15) / 16;
  return (size_t)taps * Cin * MT * 16;      // = chunks * taps * MT * 64 * VEC
}

size_t make_pack_job(const float* src, float* dst, int Cin, int Cout, int ksize, int mode, WeightJob* job) {
  const Plan pl = plan_for(Cin, Cout, ksize, mode);
  const size_t total = mfma_packed_floats(Cin, Cout, ksize, mode);
  if (!pl.ok || total == 0) return 0;
  *job = WeightJob{src, dst, 0, Cin, Cout, ksize, mode, pl.coutp, pl.qd, pl.qh, (int)total, 0};
  return total;
}

void make_flip_job(const float* src, float* dst, int ksize, int Cin, int Cout, WeightJob* job) {
  *job = WeightJob{src, dst, 1, Cin, Cout, ksize, 0, 0, 0, 0, ksize * ksize * ksize * Cin * Cout, 0};
}

int launch_weight_jobs(const WeightJob* jobs_dev, int n_jobs, int total_blocks, hipStream_t s) {
  if (n_jobs <= 0 || total_blocks <= 0) return 0;
  hipLaunchKernelGGL(weight_jobs_kernel, dim3(total_blocks), dim3(256), 0, s, jobs_dev, n_jobs);
  return launch_ok("weight_jobs_kernel");
}

int pack_weights_mfma(const float* w_tf, float* packed, int Cin, int Cout, int ksize, int mode, hipStream_t s) {
  const Plan pl = plan_for(Cin, Cout, ksize, mode);
  const int total = (int)mfma_packed_floats(Cin, Cout, ksize, mode);
  if (!pl.ok || total == 0) return 0;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_tf, packed, Cin, Cout, ksize, mode,
                     pl.coutp, pl.qd, pl.qh, total);
  return launch_ok("pack_weights_kernel");
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
template <int CIN, int COUTP, int QD, int QH, int KS, int S, int GD, int GH>
static int run_conv(const ConvArgs& a, hipStream_t s) {
  constexpr int TD = GD * QD, TH = GH * QH;
  if (a.Dout % TD || a.Dout % TH || a.Dout % 16) return 0;
  const int blocks = a.B * (a.Dout / TD) * (a.Dout / TH) * (a.Dout / 16);
  hipLaunchKernelGGL((conv_mfma_kernel<CIN, COUTP, QD, QH, KS, S, GD, GH>), dim3(blocks), dim3(256), 0, s, a);
  int rc = launch_ok("conv_mfma_kernel");
  return rc ? rc : 1;
}
// PCGC_CONV_PIPE=0: the small launches on conv_mfma_body as before (read per call: tests and tools compare the two forms)
static bool pipe_on() {
  const char* e = getenv("PCGC_CONV_PIPE");
  return !(e && atoi(e) == 0);
}
template <int CIN, int COUT, int KS, int S = 1>
static int run_small(const ConvArgs& a, hipStream_t s) {
  if (a.Dout % 16) return 0;
  const int blocks = a.B * (a.Dout / 2) * (a.Dout / 2) * (a.Dout / 16);
  if (pipe_on() && !a.x_q4 && !a.y_q4) hipLaunchKernelGGL((conv_mfma_small_kernel<CIN, COUT, KS, S>), dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_mfma_kernel<CIN, COUT, 1, 1, KS, S, 2, 2>), dim3(blocks), dim3(256), 0, s, a);
  int rc = launch_ok("conv_mfma_kernel (small launch)");
  return rc ? rc : 1;
}
template <int CIN, int COUT, int TD, int TH>
static int run_tconv(const ConvArgs& a, hipStream_t s) {
  if (a.Din % TD || a.Din % TH || a.Din % 16) return 0;
  const int blocks = a.B * (a.Din / TD) * (a.Din / TH) * (a.Din / 16);
  hipLaunchKernelGGL((tconv_mfma_kernel<CIN, COUT, TD, TH>), dim3(blocks), dim3(256), 0, s, a);
  int rc = launch_ok("tconv_mfma_kernel");
  return rc ? rc : 1;
}

template <int CIN, int COUT>
static int run_tconv_small(const ConvArgs& a, hipStream_t s) {
  if (a.Din % 16) return 0;
  const int blocks = a.B * (a.Din / 2) * (a.Din / 2) * (a.Din / 16);
  hipLaunchKernelGGL((tconv_mfma_small_kernel<CIN, COUT>), dim3(blocks), dim3(256), 0, s, a);
  int rc = launch_ok("tconv_mfma_small_kernel");
  return rc ? rc : 1;
}

// returns 1 launched / would launch, 0 no kernel for this shape, <0 error
int launch_conv_mfma(const ConvArgs& a, const float* packed_w, hipStream_t s, bool run) {
  const Plan pl = plan_for(a.Cin, a.Cout, a.ksize, a.mode);
  if (!pl.ok) return 0;
  if (a.x_cs % 4 || a.x_co % 4) return 0;
  if (a.Cout % 4 == 0 && (a.y_cs % 4 || a.y_co % 4)) return 0;
  if (a.y_q4 && (a.Cout % 4 || a.y_cs % 4 || a.y_co % 4)) return 0;
  ConvArgs b = a;
  b.w = packed_w;
  const int D = a.mode == 2 ? a.Din : a.Dout;
  if (D % 16) return 0;
  const bool small_tiles = !(getenv("PCGC_SMALL_TILES") && atoi(getenv("PCGC_SMALL_TILES")) == 0);   // read per call: tests compare the forms
  const bool small = small_tiles && (int64_t)a.B * (D / 4) * (D / 4) * (D / 16) < 320;     // fewer than 1.25 workgroups per CU
#define TRY(cond, call)                        \
  if (cond) {                                  \
    if (!run) return 1;                        \
    return call;                               \
  }
  if (a.mode == 0 && a.ksize == 3) {
    // row-packed small-Cout layers
    TRY(a.Cin == 16 && pl.coutp == 4, (run_conv<16, 4, 2, 2, 3, 1, 2, 2>(b, s)))
    TRY(a.Cin == 4 && pl.coutp == 4, (run_conv<4, 4, 2, 2, 3, 1, 2, 2>(b, s)))
    TRY(a.Cin == 4 && pl.coutp == 8, (run_conv<4, 8, 1, 2, 3, 1, 4, 2>(b, s)))
    TRY(a.Cin == 8 && pl.coutp == 8, (run_conv<8, 8, 1, 2, 3, 1, 4, 2>(b, s)))
    TRY(a.Cin == 8 && pl.coutp == 4, (run_conv<8, 4, 2, 2, 3, 1, 2, 2>(b, s)))
    TRY(a.Cin == 32 && pl.coutp == 8, (run_conv<32, 8, 1, 2, 3, 1, 4, 2>(b, s)))
    TRY(a.Cin == 16 && pl.coutp == 8, (run_conv<16, 8, 1, 2, 3, 1, 4, 2>(b, s)))
    // plain; small launches (a training batch at 16^3: 128 workgroups with 4 x 4-row tiles) take 2 x 2-row tiles —
    // four times the workgroups, the same packed filter and the same sum per output
    TRY(small && a.Cin == 16 && a.Cout == 16, (run_small<16, 16, 3>(b, s)))
    TRY(small && a.Cin == 16 && a.Cout == 32, (run_small<16, 32, 3>(b, s)))
    TRY(small && a.Cin == 16 && a.Cout == 64, (run_small<16, 64, 3>(b, s)))
    TRY(small && a.Cin == 32 && a.Cout == 16, (run_small<32, 16, 3>(b, s)))
    TRY(small && a.Cin == 64 && a.Cout == 16, (run_small<64, 16, 3>(b, s)))
    TRY(a.Cin == 8 && a.Cout == 16, (run_conv<8, 16, 1, 1, 3, 1, 4, 4>(b, s)))
    TRY(a.Cin == 16 && a.Cout == 16, (run_conv<16, 16, 1, 1, 3, 1, 4, 4>(b, s)))
    TRY(a.Cin == 16 && a.Cout == 32, (run_conv<16, 32, 1, 1, 3, 1, 4, 4>(b, s)))
    TRY(a.Cin == 16 && a.Cout == 64, (run_conv<16, 64, 1, 1, 3, 1, 4, 4>(b, s)))
    TRY(a.Cin == 32 && a.Cout == 16, (run_conv<32, 16, 1, 1, 3, 1, 4, 4>(b, s)))
    TRY(a.Cin == 64 && a.Cout == 16, (run_conv<64, 16, 1, 1, 3, 1, 4, 4>(b, s)))
    // adjoint shapes (training)
    TRY(a.Cin == 4 && a.Cout == 16, (run_conv<4, 16, 1, 1, 3, 1, 4, 4>(b, s)))
    TRY(a.Cin == 8 && a.Cout == 32, (run_conv<8, 32, 1, 1, 3, 1, 4, 4>(b, s)))
    return 0;
  }
  if (a.mode == 0 && a.ksize == 1) {
    TRY(small && a.Cin == 64 && a.Cout == 16, (run_small<64, 16, 1>(b, s)))
    TRY(small && a.Cin == 16 && a.Cout == 32, (run_small<16, 32, 1>(b, s)))
    TRY(small && a.Cin == 16 && a.Cout == 64, (run_small<16, 64, 1>(b, s)))
    TRY(small && a.Cin == 32 && a.Cout == 16, (run_small<32, 16, 1>(b, s)))
    TRY(a.Cin == 16 && a.Cout == 4, (run_conv<16, 4, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 4 && a.Cout == 8, (run_conv<4, 8, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 32 && a.Cout == 8, (run_conv<32, 8, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 8 && a.Cout == 16, (run_conv<8, 16, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 64 && a.Cout == 16, (run_conv<64, 16, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 16 && a.Cout == 32, (run_conv<16, 32, 1, 1, 1, 1, 4, 4>(b, s)))
    // adjoint shapes (training)
    TRY(a.Cin == 4 && a.Cout == 16, (run_conv<4, 16, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 8 && a.Cout == 4, (run_conv<8, 4, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 8 && a.Cout == 32, (run_conv<8, 32, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 16 && a.Cout == 8, (run_conv<16, 8, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 16 && a.Cout == 64, (run_conv<16, 64, 1, 1, 1, 1, 4, 4>(b, s)))
    TRY(a.Cin == 32 && a.Cout == 16, (run_conv<32, 16, 1, 1, 1, 1, 4, 4>(b, s)))
    return 0;
  }
  if (a.mode == 1) {
    TRY(a.Cin == 16 && a.Cout == 32, (run_conv<16, 32, 1, 1, 3, 2, 2, 2>(b, s)))
    TRY(small && a.Cin == 32 && a.Cout == 64, (run_small<32, 64, 3, 2>(b, s)))
    TRY(a.Cin == 32 && a.Cout == 64, (run_conv<32, 64, 1, 1, 3, 2, 2, 2>(b, s)))
    TRY(a.Cin == 16 && a.Cout == 16, (run_conv<16, 16, 1, 1, 3, 2, 2, 2>(b, s)))
    return 0;
  }
  if (a.mode == 2) {
    TRY(small && pipe_on() && a.Cin == 64 && a.Cout == 32 && !a.x_q4 && !a.y_q4 && (a.Cout & 3) == 0, (run_tconv_small<64, 32>(b, s)))
    TRY(a.Cin == 64 && a.Cout == 32, (run_tconv<64, 32, 2, 4>(b, s)))
    TRY(a.Cin == 32 && a.Cout == 16, (run_tconv<32, 16, 4, 4>(b, s)))
    TRY(a.Cin == 16 && a.Cout == 16, (run_tconv<16, 16, 4, 4>(b, s)))
    return 0;
  }
#undef TRY
  return 0;
}

// PCGC_CONV_PAIRS=0: every layer its own launch (read per call: tests compare both)
static bool pairs_on() {
  const char* e = getenv("PCGC_CONV_PAIRS");
  return !(e && atoi(e) == 0) && !(getenv("PCGC_SMALL_TILES") && atoi(getenv("PCGC_SMALL_TILES")) == 0);
}
// what every layer of a two-layer launch must be: stride 1, NDHWC, an MFMA plan, a `small` launch of launch_conv_mfma
static bool small_plain_layer(const ConvArgs& c) {
  const int D = c.Dout;
  return c.mode == 0 && c.Din == D && D % 16 == 0 && plan_for(c.Cin, c.Cout, c.ksize, c.mode).ok && !(c.x_cs % 4 || c.x_co % 4 || c.y_cs % 4 || c.y_co % 4) &&
         !c.x_q4 && !c.y_q4 && (int64_t)c.B * (D / 4) * (D / 4) * (D / 16) < 320;
}

// Two stride-1 layers of the small-launch (2 x 2-row tile) family in ONE launch; a / b as launch_conv_mfma takes them, pa / pb
// their packed filters.  1 = launched, 0 = this pair of shapes has no pair kernel (the caller launches them one by one), < 0 error.
int launch_conv_mfma_pair(const ConvArgs& a0, const float* pa, const ConvArgs& b0, const float* pb, hipStream_t s) {
  if (!pairs_on() || !small_plain_layer(a0) || !small_plain_layer(b0) || a0.B != b0.B || a0.Dout != b0.Dout) return 0;
  const int D = a0.Dout;
  ConvArgs a = a0, b = b0;
  a.w = pa; b.w = pb;
  const int n = a.B * (D / 2) * (D / 2) * (D / 16);                                       // workgroups of each layer (2 x 2-row tiles)
#define PAIR(CA, OA, KA, CB, OB, KB)                                                                                              \
  if (a.Cin == CA && a.Cout == OA && a.ksize == KA && b.Cin == CB && b.Cout == OB && b.ksize == KB) {                            \
    if (pipe_on()) hipLaunchKernelGGL((conv_mfma_pair_kernel<CA, OA, KA, CB, OB, KB, true>), dim3(2 * n), dim3(256), 0, s, a, b, n, n); \
    else hipLaunchKernelGGL((conv_mfma_pair_kernel<CA, OA, KA, CB, OB, KB, false>), dim3(2 * n), dim3(256), 0, s, a, b, n, n);     \
    const int rc = launch_ok("conv_mfma_pair_kernel");                                                                            \
    return rc ? rc : 1;                                                                                                           \
  }
  PAIR(64, 16, 3, 64, 16, 1)       // conv1_1 | conv2_1 of a C = 64 block
  PAIR(16, 32, 3, 16, 16, 3)       // conv1_2 | conv2_2
  PAIR(32, 16, 3, 32, 16, 1)       // their reverse: conv1_2^T | conv2_3^T
#undef PAIR
  return 0;
}

// Layer a, then the 1x1x1 layer b accumulating into a's output (b.add_to == b.y == a.y), one launch: 1 / 0 / < 0 as above.
int launch_conv_mfma_chain(const ConvArgs& a0, const float* pa, const ConvArgs& b0, const float* pb, hipStream_t s) {
  if (!pairs_on() || !small_plain_layer(a0) || !small_plain_layer(b0) || a0.B != b0.B || a0.Dout != b0.Dout) return 0;
  if (b0.y != a0.y || b0.add_to != a0.y || b0.y_cs != a0.y_cs || b0.y_co != a0.y_co || b0.Cout != a0.Cout || b0.res || a0.res) return 0;
  ConvArgs a = a0, b = b0;
  a.w = pa; b.w = pb;
  const int D = a.Dout, n = a.B * (D / 2) * (D / 2) * (D / 16);
  if (a.Cin == 16 && a.Cout == 64 && a.ksize == 3 && b.Cin == 16 && b.ksize == 1) {       // conv1_1^T, conv2_1^T of a C = 64 block
    if (pipe_on()) hipLaunchKernelGGL((conv_mfma_chain_kernel<16, 64, 3, 16, 64, 1, true>), dim3(n), dim3(256), 0, s, a, b);
    else hipLaunchKernelGGL((conv_mfma_chain_kernel<16, 64, 3, 16, 64, 1, false>), dim3(n), dim3(256), 0, s, a, b);
    const int rc = launch_ok("conv_mfma_chain_kernel");
    return rc ? rc : 1;
  }
  return 0;
}

// Layer a (1x1x1, writes t23 = a.y, C / 2 channels) and the block's merge out = relu(x + [t12 | t23]): 1 / 0 / < 0 as above.
int launch_conv_mfma_merge(const ConvArgs& a0, const float* pa, const MergeArgs& m, hipStream_t s) {
  if (!pairs_on() || !small_plain_layer(a0) || a0.ksize != 1 || a0.res || a0.mask || a0.add_to || a0.y_cs != a0.Cout || a0.y_co || 2 * a0.Cout != m.C) return 0;
  ConvArgs a = a0;
  a.w = pa;
  const int D = a.Dout, n = a.B * (D / 2) * (D / 2) * (D / 16);
  if (a.Cin == 16 && a.Cout == 32) {                                                     // conv2_3 of a C = 64 block
    if (pipe_on()) hipLaunchKernelGGL((conv_mfma_merge_kernel<16, 32, 1, true>), dim3(n), dim3(256), 0, s, a, m);
    else hipLaunchKernelGGL((conv_mfma_merge_kernel<16, 32, 1, false>), dim3(n), dim3(256), 0, s, a, m);
    const int rc = launch_ok("conv_mfma_merge_kernel");
    return rc ? rc : 1;
  }
  return 0;
}

}  // namespace pcgc
