// Full-resolution (64^3, C = 16) stage of the auto-encoder transforms on v_mfma_f32_4x4x1_16B_f32 "row" kernels:
//   conv_in (1 -> 16), the C = 16 Voxception-ResNet blocks, deconv_out (16 -> 1)
//   (models/model_voxception.py:83-88, 56-68, 188-192).
//
// Why this instruction.  These layers have 1, 4 or 8 output channels; on the 16x16x4 matrix tile 44-75 % of the
// rows multiply zeros, and the VALU form (vrn_valu.hip) is bound by LDS operand reads (round 1: 0.42-0.48 of the
// fp32 peak).  v_mfma_f32_4x4x1_16B_f32 is 16 independent 4x4x1 outer products per instruction:
//   block b (lanes 4b..4b+3):  D_b[i][j] += A_b[i] * B_b[j]
// with M = 4 = the layers' output-channel quantum, so no row is padding, at the same 64 FLOP/clk/SIMD.
// Mapping: a wave owns whole W rows of the cube — D = 64 voxels = 64 lanes:
//   B operand  = one input channel of the row: lane = voxel w, one VGPR per (row, channel);
//   D operand  = VGPR i of lane w = output channel i of voxel w  (so a layer's output IS the next layer's B operand);
//   A operand  = weights, broadcast from ONE block with cbsz = 4 / abid = k: a single VGPR holds 16 (tap, ci) weight
//                vectors x 4 output channels — 64 consecutive floats of the TensorFlow kernel layout
//                [kd][kh][kw][Cin][Cout], loaded once; all of a layer's weights stay in <= 28 VGPRs;
//   kw = 0 / 2 taps = the same row shifted by one lane: DPP wave_shr / wave_shl with zero fill, which is exactly the
//                'same' zero padding at w = -1 / 64;  kd, kh taps = the same B register feeding other rows' accumulators.
// No LDS, no barriers.  Activations of this stage live in the "Q4" layout [b][d][h][C/4][w][4] (a lane's dwordx4 =
// 4 channels of its voxel, 1 KiB coalesced per wave instruction); rows outside the cube are read through a raw
// buffer descriptor with an out-of-range offset (returns zeros, no branch, vmcnt stays countable).
// A wave walks LD planes of TH rows along d with three rotating plane accumulators (sliding window over kd).
// Summation order per output: bias, then (plane, channel, kh, kw) — fixed, independent of batch and placement.
#include <cstdlib>
#include <type_traits>
#include "row_common.h"

namespace pcgc {

constexpr int kD = 64;              // cube edge these kernels are built for (= wavefront width)

// Byte offsets of (plane p, row h, quad q) and of a lane's voxel in a tensor with NQ quads per voxel: Q4
// [d][h][NQ][w][4] (inference) or NDHWC [d][h][w][NQ*4] (NHWC = true: the training step's tensors; the two coincide
// for NQ = 1).
template <bool NHWC, int NQ>
__device__ __forceinline__ int row_off(int p, int h, int q) {
  return NHWC ? (p * kD + h) * (kD * NQ * 16) + q * 16 : ((p * kD + h) * NQ + q) * (kD * 16);
}
template <bool NHWC, int NQ>
__device__ __forceinline__ int lane_off(int lane) { return NHWC ? lane * NQ * 16 : lane * 16; }

// the descriptor itself when ok, else one of zero records (loads return zeros, stores are dropped); ok is wave-uniform
__device__ __forceinline__ i32x4 rsrc_if(i32x4 rs, bool ok) {
  rs[2] = ok ? rs[2] : 0;
  return rs;
}

// A descriptor whose base is p + byte_off (all wave-uniform: scalar adds), of zero records unless ok.  For STORES: with
// the row offset in the base, the store keeps soffset = 0 and only a constant in its offset field.  A 128-bit buffer
// store with an SGPR soffset whose data register is written by the very next VALU instruction loses that write race
// on gfx950 (measured: one channel of lanes 12..15 of each 16 wrong, run to run different, when two waves share the
// SIMD), and the compiler inserts the wait state only for stores WITHOUT a register soffset.
__device__ __forceinline__ i32x4 rsrc_at(const void* p, int byte_off, bool ok) {
  const unsigned long long a = (unsigned long long)p + (unsigned)byte_off;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = ok ? 0x7ffff000 : 0;
  r[3] = 0x00020000;
  return r;
}

// rows h0-1 .. h0+TH of plane p, channel quad q of a tensor with NQ quads per voxel; lane_b = lane_off<NHWC, NQ>(lane)
template <int TH, int NQ, bool NHWC = false>
__device__ __forceinline__ void load_rows(f32x4 (&buf)[TH + 2], i32x4 rs, int lane_b, int p, int q, int h0) {
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) {
    const int h = h0 - 1 + r;
    const bool ok = (unsigned)h < (unsigned)kD && (unsigned)p < (unsigned)kD;
    // the row offset travels in the scalar offset operand and a row outside the cube reads through a descriptor of
    // zero records (every access out of range -> zeros): no vector instruction goes into the address
    buf[r] = raw_load4(rsrc_if(rs, ok), lane_b, ok ? row_off<NHWC, NQ>(p, h, q) : 0, 0);
  }
}

// The same rows where some may lie in VIRTUAL tiles (RowSkip: tiles their producer skipped without writing): bit h of
// `vrow` (the plane's word of RowSkip::in_virtual, 0 when the tensor is complete) sends row h to rsE, the producer's
// empty-cube response (one cube, same layout: same offsets).  All selects are scalar.
template <int TH, int NQ>
__device__ __forceinline__ void load_rows_v(f32x4 (&buf)[TH + 2], i32x4 rs, i32x4 rsE, unsigned long long vrow, int lane_b, int p, int q,
                                            int h0) {
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) {
    const int h = h0 - 1 + r;
    const bool ok = (unsigned)h < (unsigned)kD && (unsigned)p < (unsigned)kD;
    const bool e = ok && ((vrow >> (h & 63)) & 1ull);
    i32x4 d = rs;
    d[0] = e ? rsE[0] : rs[0];
    d[1] = e ? rsE[1] : rs[1];
    buf[r] = raw_load4(rsrc_if(d, ok), lane_b, ok ? row_off<false, NQ>(p, h, q) : 0, 0);
  }
}
// Kernel A's form of the same: the tile's virtual-row bits are fetched ONCE — lane i (< LD + 3) keeps, for plane d0 - 1 + i,
// bit r = row h0 - 1 + r of the tile's row window (virtual_row_masks) — and a plane's mask comes out of that register with
// one v_readlane.  (The per-load table lookups of load_rows_v cost kernel A, which is bound by instruction issue, 15 %:
// 80 us against 68.6 us for the same tiles, profiles/r04_vC_skip_launches.txt; BC is bound by bytes and keeps them.)
template <int TH, int LD>
__device__ __forceinline__ unsigned virtual_row_masks(const unsigned long long* table, int b, int d0, int h0, int lane) {
  unsigned m = 0;
  const int p = d0 - 1 + lane;
  if (table && lane < LD + 3 && (unsigned)p < (unsigned)kD) {
    const unsigned long long w = table[(size_t)b * kD + p];
    m = (unsigned)(h0 > 0 ? w >> (h0 - 1) : w << 1) & ((1u << (TH + 2)) - 1u);
  }
  return m;
}
template <int TH, int NQ>
__device__ __forceinline__ void load_rows_m(f32x4 (&buf)[TH + 2], i32x4 rs, i32x4 rsE, unsigned mrow, int lane_b, int p, int q, int h0) {
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) {
    const int h = h0 - 1 + r;
    // wave-uniform by construction, and SAID so: for the row behind the tile the compiler had this test in a vector register,
    // took the descriptor built from it for divergent and wrapped the load in a waterfall loop (readfirstlane x 5, two 64-bit
    // compares, saveexec, branch) — one per row fetch, 0.20 instead of 0.07 vector instructions per MFMA in kernel A
    const bool ok = __builtin_amdgcn_readfirstlane((int)((unsigned)h < (unsigned)kD && (unsigned)p < (unsigned)kD)) != 0;
    const bool e = ok && ((mrow >> r) & 1u);
    i32x4 d = rs;
    d[0] = e ? rsE[0] : rs[0];
    d[1] = e ? rsE[1] : rs[1];
    buf[r] = raw_load4(rsrc_if(d, ok), lane_b, __builtin_amdgcn_readfirstlane(ok ? row_off<false, NQ>(p, h, q) : 0), 0);
  }
}
// the plane's word of a virtual-row table (0: no table, or the plane is outside the cube)
__device__ __forceinline__ unsigned long long virtual_rows(const unsigned long long* table, int b, int p) {
  if (!table || (unsigned)p >= (unsigned)kD) return 0ull;
  const unsigned long long v = table[(size_t)b * kD + p];
  return ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

struct Tile {
  int b, h0, d0;
};
// block = index of the wave's workgroup among those of its kernel role (blockIdx.x unless the launch mixes roles)
template <int TH, int LD>
__device__ __forceinline__ Tile wave_tile(int block = blockIdx.x) {
  int wv = __builtin_amdgcn_readfirstlane(block * 4 + (threadIdx.x >> 6));
  Tile t;
  t.h0 = (wv % (kD / TH)) * TH; wv /= (kD / TH);
  t.d0 = (wv % (kD / LD)) * LD; wv /= (kD / LD);
  t.b = wv;
  return t;
}

// --- exact skipping of empty space (RowSkip, common.h) -----------------------------------------------------------
// The wave's tile through the launch's permutation; *heavy = false: copy the tile from the empty-cube response.
template <int TH, int LD>
__device__ __forceinline__ Tile wave_tile_ordered(const RowSkip& k, bool* heavy, int block = blockIdx.x, bool* unread = nullptr) {
  const int wid = __builtin_amdgcn_readfirstlane(block * 4 + (threadIdx.x >> 6));
  int wv = wid;
  *heavy = true;
  if (k.order) {
    const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)k.order[wid]);
    if (unread) *unread = (e & kTileUnread) != 0u;
    wv = (int)(e & ~kTileUnread);
    *heavy = wid < (int)*k.n_heavy;
    if (!*heavy && k.counter && (threadIdx.x & 63) == 0) atomicAdd(k.counter, 1u);
  }
  Tile t;
  t.h0 = (wv % (kD / TH)) * TH; wv /= (kD / TH);
  t.d0 = (wv % (kD / LD)) * LD; wv /= (kD / LD);
  t.b = wv;
  return t;
}

// Tile orders of every launch of every chunk of a stage (one workgroup per (launch configuration, chunk)): a tile is EMPTY
// when the fine (64^3) window its outputs depend on holds no occupied row; the tiles to compute come first, in natural
// order, then the empty ones.  Phase 1: one 64-bit word per (cube, plane tile) = the OR of the row-occupancy words over the
// tile's plane window (<= 22 loads, once per plane tile instead of once per tile: the first form took 47 us per chunk).
// Phase 2: thread t owns a contiguous range of tiles and tests each against its row window; one scan over the per-thread
// counts keeps the order.  Chunk k holds cubes [k * chunk, ...); its orders start at order + first cube * n_cfg * tiles_cap
// (stride n * tiles_cap per configuration), its counts at n_heavy + k * n_cfg, its virtual-row tables (64^3 stage only) at
// virt + first cube * n_cfg * 64 (stride n * 64).
__global__ void __launch_bounds__(1024) tile_order_kernel(const unsigned long long* rowocc, int total, int chunk, const TileCfg* cfgs,
                                                           const TileCfg* cfgs_small, int n_cfg, unsigned* order, unsigned* n_heavy,
                                                           int tiles_cap, unsigned long long* virt) {
  __shared__ unsigned cnt[1024];
  __shared__ unsigned long long win_or[2048];                          // [cube][plane tile]: <= 128 cubes x 16 plane tiles
  __shared__ unsigned long long win_need[2048];                        // the same for the wider window of TileCfg::need
  __shared__ unsigned long long erows[2048];                           // [cube][plane tile]: rows of the plane tile's EMPTY tiles
  const int c0 = blockIdx.y * chunk;
  const int B = total - c0 < chunk ? total - c0 : chunk;
  const TileCfg c = (cfgs_small && B <= 16 ? cfgs_small : cfgs)[blockIdx.x];
  const unsigned long long* ro = rowocc + (size_t)c0 * kD;
  const int G = kD / c.step;                                           // the launch's grid: 64^3 or 32^3
  unsigned long long* vm = (virt && c.step == 1) ? virt + ((size_t)c0 * n_cfg + (size_t)blockIdx.x * B) * kD : nullptr;
  const int nh = G / c.th, nd = G / c.ld, n = B * nh * nd;
  for (int i = threadIdx.x; i < B * nd; i += 1024) {
    erows[i] = 0ull;
    const int b = i / nd, d0 = (i - b * nd) * c.ld;
    int p0 = c.step * d0 - c.lo, p1 = c.step * (d0 + c.ld - 1) + c.hi;             // fine planes the tile's outputs depend on
    p0 = p0 < 0 ? 0 : p0; p1 = p1 > kD - 1 ? kD - 1 : p1;
    unsigned long long any = 0;
    for (int p = p0; p <= p1; ++p) any |= ro[(size_t)b * kD + p];
    win_or[i] = any;
    if (c.need > 0) {
      int q0 = c.step * d0 - c.need, q1 = c.step * (d0 + c.ld - 1) + c.need;
      q0 = q0 < 0 ? 0 : q0; q1 = q1 > kD - 1 ? kD - 1 : q1;
      for (int p = q0; p < p0; ++p) any |= ro[(size_t)b * kD + p];
      for (int p = p1 + 1; p <= q1; ++p) any |= ro[(size_t)b * kD + p];
      win_need[i] = any;
    }
  }
  __syncthreads();
  const int per = (n + 1023) / 1024, t0 = threadIdx.x * per, t1 = t0 + per < n ? t0 + per : n;
  unsigned long long flags = 0, unread = 0;     // per <= 64 tiles per thread (B <= 128 cubes per chunk)
  unsigned heavy = 0;
  for (int t = t0; t < t1; ++t) {
    int wv = t;
    const int h0 = (wv % nh) * c.th; wv /= nh;
    const int dt = wv % nd; wv /= nd;
    int lo = c.step * h0 - c.lo, hi = c.step * (h0 + c.th - 1) + c.hi;             // ... and fine rows
    lo = lo < 0 ? 0 : lo; hi = hi > kD - 1 ? kD - 1 : hi;
    const unsigned long long win = (hi - lo == 63) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << lo);
    if (win_or[wv * nd + dt] & win) { flags |= 1ull << (t - t0); ++heavy; }
    else if (c.need > 0) {
      int nlo = c.step * h0 - c.need, nhi = c.step * (h0 + c.th - 1) + c.need;
      nlo = nlo < 0 ? 0 : nlo; nhi = nhi > kD - 1 ? kD - 1 : nhi;
      const unsigned long long wn = (nhi - nlo == 63) ? ~0ull : (((1ull << (nhi - nlo + 1)) - 1ull) << nlo);
      if (!(win_need[wv * nd + dt] & wn)) unread |= 1ull << (t - t0);
    }
    // the virtual-row table is assembled in LDS (one atomic per empty tile) and written out once below: the per-plane
    // global atomics it used to take (8 per empty tile) were most of this kernel's 69 us
    if (!((flags >> (t - t0)) & 1ull) && vm) atomicOr(&erows[wv * nd + dt], ((1ull << c.th) - 1ull) << h0);
  }
  cnt[threadIdx.x] = heavy;
  __syncthreads();
  if (vm) {
    for (int i = threadIdx.x; i < B * kD; i += 1024) vm[i] = erows[(i / kD) * nd + (i % kD) / c.ld];
  }
  for (int off = 1; off < 1024; off <<= 1) {    // inclusive scan
    const unsigned v = threadIdx.x >= (unsigned)off ? cnt[threadIdx.x - off] : 0u;
    __syncthreads();
    cnt[threadIdx.x] += v;
    __syncthreads();
  }
  const unsigned total_heavy = cnt[1023];
  unsigned hpos = cnt[threadIdx.x] - heavy;                    // heavy tiles before this thread's range
  unsigned epos = total_heavy + (unsigned)t0 - hpos;           // empty tiles before it, behind all heavy ones
  unsigned* o = order + ((size_t)c0 * n_cfg + (size_t)blockIdx.x * B) * tiles_cap;
  for (int t = t0; t < t1; ++t) {
    if ((flags >> (t - t0)) & 1ull) o[hpos++] = (unsigned)t;
    else o[epos++] = (unsigned)t | (((unread >> (t - t0)) & 1ull) ? kTileUnread : 0u);
  }
  if (threadIdx.x == 0) n_heavy[blockIdx.y * n_cfg + blockIdx.x] = total_heavy;
}

int launch_tile_order(const unsigned long long* rowocc, int total, int chunk, const TileCfg* cfg, const TileCfg* cfg_small, int n_cfg,
                      unsigned* order, unsigned* n_heavy, int tiles_cap, unsigned long long* virt, hipStream_t s) {
  if (chunk > 128 || chunk < 1) { set_error("launch_tile_order: 1 .. 128 cubes per chunk (got %d)", chunk); return -1; }
  hipLaunchKernelGGL(tile_order_kernel, dim3(n_cfg, (total + chunk - 1) / chunk), dim3(1024), 0, s, rowocc, total, chunk, cfg, cfg_small, n_cfg,
                     order, n_heavy, tiles_cap, virt);
  return launch_ok("tile_order_kernel");
}

// the tile's rows of an NQ-quad Q4 tensor copied from the empty-cube response (one cube, same layout)
template <int TH, int LD, int NQ>
__device__ __forceinline__ void copy_empty_tile(const float* empty, float* dst_cube, const Tile& tl, int lane) {
  const f32x4* src = reinterpret_cast<const f32x4*>(empty) + lane;
  f32x4* dst = reinterpret_cast<f32x4*>(dst_cube) + lane;
#pragma unroll
  for (int p = 0; p < LD; ++p) {
    f32x4 v[TH * NQ];                        // a plane's rows in flight before their stores
#pragma unroll
    for (int i = 0; i < TH * NQ; ++i) v[i] = src[(((size_t)(tl.d0 + p) * kD + tl.h0) * NQ + i) * 64];
#pragma unroll
    for (int i = 0; i < TH * NQ; ++i) dst[(((size_t)(tl.d0 + p) * kD + tl.h0) * NQ + i) * 64] = v[i];
  }
}

// rowocc[b][d] bit h = row (d, h) of cube b holds a voxel whose bits are not +0.0 (so -0.0 counts as occupied: the
// empty-cube response was made from +0.0 inputs).  One wave per plane: lane = (row in a group of 4, w quad).
__global__ void __launch_bounds__(256) rowocc_kernel(const float* x, unsigned long long* rowocc, int planes) {
  const int lane = threadIdx.x & 63;
  const int pl = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const uint4* px = reinterpret_cast<const uint4*>(x + (size_t)pl * kD * kD) + lane;
  unsigned long long m = 0;
#pragma unroll
  for (int g = 0; g < 16; ++g) {             // rows 4g .. 4g+3: 1 KiB per wave instruction
    const uint4 v = px[g * 64];
    const unsigned long long b = __builtin_amdgcn_ballot_w64((v.x | v.y | v.z | v.w) != 0u);
#pragma unroll
    for (int j = 0; j < 4; ++j) m |= ((b >> (16 * j)) & 0xffffull) ? (1ull << (4 * g + j)) : 0ull;
  }
  if (lane == 0) rowocc[pl] = m;
}

int launch_rowocc(const float* x, unsigned long long* rowocc, int B, hipStream_t s) {
  const int planes = B * kD;
  hipLaunchKernelGGL(rowocc_kernel, dim3((planes + 3) / 4), dim3(256), 0, s, x, rowocc, planes);
  return launch_ok("rowocc_kernel");
}

struct VrnRowArgs {
  const float* x;      // block input, Q4 [B][64][64][4][64][4]
  float* t12;          // scratch,     Q4 [B][64][64][2][64][4]: quad 0 = tensor1_1, quad 1 = tensor2_1
  float* out;          // block output, Q4 like x (may alias x: every element is read once, by the wave that writes it)
  // training variant (TRAIN = true): x / out / pre NDHWC [B][64][64][64][16]; t12 = tensor1_1, t21 = tensor2_1,
  // t22 = relu(conv2_2) as separate [B][64][64][64][4] tensors; pre = [relu(conv1_2) | relu(conv2_3)], the block's
  // pre-residual output (the reverse pass needs its sign)
  float* t21 = nullptr;
  float* t22 = nullptr;
  float* pre = nullptr;
  // instead of pre: one int32 per voxel, bit c = (pre[c] > 0) — all the reverse pass reads of pre (2 B of information
  // instead of 64 B per voxel: the pre stores were 46 us of the pair's 208 us, tools/exp/t_ablate_train.py)
  int* pre_signs = nullptr;
  const float *w11, *b11, *w21, *b21, *w12, *b12, *w22, *b22, *w23, *b23;   // TensorFlow layouts
  int B;
  RowSkip skip;        // inference only: empty tiles are copied from the empty-cube response (skip.order != nullptr)
  int abl = 0;         // tools/exp/t_ablate.py (builds with -DPCGC_EXPERIMENTS only): 1 = stores dropped, 2 = residual loads
                       // read nothing, 4 = input loads read nothing — same instruction stream, no memory traffic
};
#ifdef PCGC_EXPERIMENTS
#define PCGC_ABL(a, bit) ((a).abl & (bit))
// fusion probes (abl & 64): 4 planes x 4 waves x (TH + 2 = 4 rows) x 2 quads x 1 KiB = 128 KiB would not leave room for two
// workgroups per CU; the probes keep 4 planes x 4 waves x 2 rows x 2 quads = 64 KiB (the halo rows alias their neighbours')
__device__ __forceinline__ float* exp_lds_ptr() {
  __shared__ __attribute__((aligned(16))) float buf[4 * 4 * 2 * 2 * 256];
  return buf;
}
#define exp_lds exp_lds_ptr()
#else
#define PCGC_ABL(a, bit) 0
#endif

// ---------------------------------------------------------------------------------------------------------------
// kernel A:  t12 = [ relu(conv1_1(x)) (3^3, 16 -> 4) | relu(conv2_1(x)) (1^3, 16 -> 4) ]
// weights: VGPR t (t = tap) = w11[t*64 + lane] = W[tap][ci = lane>>2][co = lane&3]  => abid = ci
// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// The lane shifts act on the 4 output channels instead of the 16 input channels.  A convolution is linear and a lane
// shift commutes with it: with S_kw[w] = sum over (kd, kh, ci) of W[kd][kh][kw][ci] x[ci][w] (the tap
// column kw applied to the UNSHIFTED rows), the output is  y[w] = S_1[w] + S_0[w-1] + S_2[w+1].  The MFMAs read the
// loaded rows directly (no v_mov_dpp before them: shifting the inputs took 192 per plane step, and every non-MFMA VALU
// instruction costs about one MFMA issue slot, tools/exp/exp_mfma_issue.hip: 101 -> 103 TFLOP/s); the finished plane's three partial sums are
// combined with 8 lane shifts per output row.  Three partial sums per (plane set, row) triple the accumulators, so a
// wave takes TH = 2 rows; the plane loop is unrolled three times with rotating set roles and a fresh accumulator's
// first MFMA takes bias / zero as its C operand (no register moves).  Summation order per output: per kw column
// (plane, channel, kh) in program order, then S_1 + shr(S_0) + shl(S_2).
// ---------------------------------------------------------------------------------------------------------------
template <int TH, int P0, int P1, int P2, bool FRESH>
__device__ __forceinline__ void a_channel(f32x4 (&S)[3][3][TH], f32x4 (&acc2)[TH], const f32x4& bias, const f32x4& bias2,
                                           const float (&W)[27], float W2, int ci, const f32x4 (&buf)[TH + 2], int c, bool v0, bool v1,
                                           bool v2) {
  float x0[TH + 2];
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) x0[r] = comp(buf[r], c);
  const bool vj[3] = {v0, v1, v2};
  constexpr int P[3] = {P0, P1, P2};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int kd = 2 - j;                      // input plane p feeds output plane p + 1 - kd = p - 1 + j
    if (vj[j]) {
#pragma unroll
      for (int r = 0; r < TH + 2; ++r)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int jr = r - kh;
          if (jr >= 0 && jr < TH) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const bool first = FRESH && j == 2 && kh == 0;       // first tap that reaches this accumulator of the new plane
              if (first) S[P[j]][kw][jr] = mfa_new(ci, W[(kd * 3 + kh) * 3 + kw], x0[r], kw == 1 ? bias : zero);
              else S[P[j]][kw][jr] = mfa(ci, W[(kd * 3 + kh) * 3 + kw], x0[r], S[P[j]][kw][jr]);
            }
          }
        }
    }
  }
  if (v1) {
#pragma unroll
    for (int jr = 0; jr < TH; ++jr) acc2[jr] = FRESH ? mfa_new(ci, W2, x0[jr + 1], bias2) : mfa(ci, W2, x0[jr + 1], acc2[jr]);
  }
}

// The same MFMAs for the four channels of one loaded quad with the validity tests hoisted: ONE wave-uniform branch per
// (quad, output plane) instead of one per (channel, output plane) — 12 instead of 48 per plane step, and with them three
// quarters of the s_waitcnt the compiler puts at the head of every conditional block (kernel A: 391 branches + 635 waits per
// 2 688 MFMAs; scalar instructions are free only up to about one per two MFMAs, tools/exp/exp_mfma_issue.hip).  Every
// accumulator still receives its contributions in the order (plane, channel, kh, kw): bit-identical to a_channel.
template <int TH, int P0, int P1, int P2, bool FRESH>
__device__ __forceinline__ void a_quad(f32x4 (&S)[3][3][TH], f32x4 (&acc2)[TH], const f32x4& bias, const f32x4& bias2,
                                        const float (&W)[27], float W2, int ci0, const f32x4 (&buf)[TH + 2], bool v0, bool v1, bool v2) {
  const bool vj[3] = {v0, v1, v2};
  constexpr int P[3] = {P0, P1, P2};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int kd = 2 - j;                      // input plane p feeds output plane p + 1 - kd = p - 1 + j
    if (vj[j]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int r = 0; r < TH + 2; ++r)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int jr = r - kh;
            if (jr >= 0 && jr < TH) {
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) {
                const bool first = FRESH && c == 0 && j == 2 && kh == 0;   // first tap that reaches this accumulator of the new plane
                if (first) S[P[j]][kw][jr] = mfa_new(ci0 + c, W[(kd * 3 + kh) * 3 + kw], comp(buf[r], c), kw == 1 ? bias : zero);
                else S[P[j]][kw][jr] = mfa(ci0 + c, W[(kd * 3 + kh) * 3 + kw], comp(buf[r], c), S[P[j]][kw][jr]);
              }
            }
          }
        if (j == 1) {                            // conv2_1 (1^3) of the wave's own rows rides in the centre plane's block
#pragma unroll
          for (int jr = 0; jr < TH; ++jr)
            acc2[jr] = (FRESH && c == 0) ? mfa_new(ci0 + c, W2, comp(buf[jr + 1], c), bias2) : mfa(ci0 + c, W2, comp(buf[jr + 1], c), acc2[jr]);
        }
      }
    }
  }
}

__device__ __forceinline__ f32x4 shr4(f32x4 v) { return f32x4{shr1(v[0]), shr1(v[1]), shr1(v[2]), shr1(v[3])}; }
__device__ __forceinline__ f32x4 shl4(f32x4 v) { return f32x4{shl1(v[0]), shl1(v[1]), shl1(v[2]), shl1(v[3])}; }

// SKIP: the launch belongs to the analysis' 64^3 stage with empty-space skipping on (a.skip.order != nullptr): the wave's
// tile comes from the launch's tile order, an empty tile is copied from the empty-cube response or not written at all,
// and input rows that lie in tiles THEIR producer did not write are read from that producer's empty-cube response.
// NHWC: the 16-channel tensors (x / out / pre) are NDHWC — the training step's original layout; TRAIN with NHWC = false is
// the training step with its 64^3 stage in the Q4 layout of the inference path (1 KiB per wave instruction instead of
// 16 B per lane at a 64 B stride)
template <int TH, int LD, bool TRAIN = false, bool SKIP = false, bool NHWC = TRAIN, bool QUADJ = true>
__device__ __forceinline__ void vrn16a_row_body(const VrnRowArgs& a, int block) {
  static_assert(!(TRAIN && SKIP), "the training step computes every tile");
  static_assert(TRAIN || !NHWC, "the inference tensors are Q4");
  const int lane = threadIdx.x & 63;
  bool heavy = true;
  const Tile tl = SKIP ? wave_tile_ordered<TH, LD>(a.skip, &heavy, block) : wave_tile<TH, LD>(block);
  const int h0 = tl.h0, d0 = tl.d0;
  if (SKIP && !heavy) {
    if (a.skip.materialize) copy_empty_tile<TH, LD, 2>(a.skip.empty, a.t12 + (size_t)tl.b * kD * kD * kD * 8, tl, lane);
    return;
  }
  float W[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) W[t] = a.w11[t * 64 + lane];
  const float W2 = a.w21[lane];
  f32x4 bi = {a.b11[0], a.b11[1], a.b11[2], a.b11[3]};
  f32x4 bi2 = {a.b21[0], a.b21[1], a.b21[2], a.b21[3]};
  asm volatile("" : "+v"(bi), "+v"(bi2));                 // biases live in VGPRs: they are MFMA C operands
  f32x4 S[3][3][TH], acc2[TH];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int r = 0; r < TH; ++r) S[j][k][r] = bi;
#pragma unroll
  for (int r = 0; r < TH; ++r) acc2[r] = bi2;
  const i32x4 rs = rsrc_if(make_rsrc(a.x + (size_t)tl.b * kD * kD * kD * 16, kD * kD * kD * 16 * 4), !PCGC_ABL(a, 4));
  const int lane16 = lane_off<NHWC, 4>(lane);
  f32x4* tb = reinterpret_cast<f32x4*>(a.t12) + (size_t)tl.b * kD * kD * (TRAIN ? 1 : 2) * 64 + lane;
  f32x4* tb2 = TRAIN ? reinterpret_cast<f32x4*>(a.t21) + (size_t)tl.b * kD * kD * 64 + lane : nullptr;
  // three row buffers, requested TWO channel quads ahead of their use (a quad step is 216 MFMAs, about 0.8 us: one step
  // of lead left part of the memory latency exposed); they rotate with the same permutation as the plane sets
  f32x4 buf[3][TH + 2];
  const i32x4 rsE = SKIP ? make_rsrc(a.skip.in_empty ? a.skip.in_empty : a.x, kD * kD * kD * 16 * 4) : rs;
  // The tile's virtual-row bits, TH + 2 per plane for the LD + 3 planes the loop touches (the look-ahead reaches plane d0 + LD + 1),
  // packed into ONE scalar word pair up front: readlane with constant lanes.  (A readlane with the plane as a run-time lane
  // index in every row fetch was wrapped in a waterfall loop by the compiler each time — 11 of them in the unrolled loop body,
  // v_readfirstlane + two 64-bit compares + saveexec: 0.20 instead of 0.07 vector instructions per MFMA in the skipping launches.)
  static_assert((LD + 3) * (TH + 2) <= 64, "the tile's virtual-row bits fit one 64-bit word");
  unsigned long long vbits = 0;
  if constexpr (SKIP) {
    const unsigned vmask = virtual_row_masks<TH, LD>(a.skip.in_virtual, tl.b, d0, h0, lane);
#pragma unroll
    for (int i = 0; i < LD + 3; ++i) vbits |= (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)vmask, i) << (i * (TH + 2));
  }
  auto rows = [&](f32x4 (&b)[TH + 2], int p, int q) {
    if constexpr (SKIP) load_rows_m<TH, 4>(b, rs, rsE, (unsigned)(vbits >> ((p - (d0 - 1)) * (TH + 2))) & ((1u << (TH + 2)) - 1u), lane16, p, q, h0);
    else load_rows<TH, 4, NHWC>(b, rs, lane16, p, q, h0);
  };
  rows(buf[0], d0 - 1, 0);
  rows(buf[1], d0 - 1, 1);
  auto step = [&](int p, auto P0_, auto P1_, auto P2_) {
    constexpr int P0 = decltype(P0_)::value, P1 = decltype(P1_)::value, P2 = decltype(P2_)::value;
    const bool pin = (unsigned)p < (unsigned)kD;
    // v2 does not ask for the plane to exist: plane p + 1's partial sums are BORN in this step (bias / zero as the C
    // operand of their first MFMA), and an input plane outside the cube reads zeros
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = p + 1 < d0 + LD;
    rows(buf[P2], p, 2);
    if constexpr (QUADJ) {
      a_quad<TH, P0, P1, P2, true>(S, acc2, bi, bi2, W, W2, 0, buf[P0], v0, v1, v2);
      rows(buf[P0], p, 3);
      a_quad<TH, P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, 4, buf[P1], v0, v1, v2);
      rows(buf[P1], p + 1, 0);
      a_quad<TH, P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, 8, buf[P2], v0, v1, v2);
      rows(buf[P2], p + 1, 1);
      a_quad<TH, P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, 12, buf[P0], v0, v1, v2);
    } else {
    a_channel<TH, P0, P1, P2, true>(S, acc2, bi, bi2, W, W2, 0, buf[P0], 0, v0, v1, v2);
#pragma unroll
    for (int c = 1; c < 4; ++c) a_channel<TH, P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, c, buf[P0], c, v0, v1, v2);
    rows(buf[P0], p, 3);
#pragma unroll
    for (int c = 0; c < 4; ++c) a_channel<TH, P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, 4 + c, buf[P1], c, v0, v1, v2);
    rows(buf[P1], p + 1, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) a_channel<TH, P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, 8 + c, buf[P2], c, v0, v1, v2);
    rows(buf[P2], p + 1, 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) a_channel<TH, P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, 12 + c, buf[P0], c, v0, v1, v2);
    }
#ifdef PCGC_EXPERIMENTS
    if (PCGC_ABL(a, 64)) {
      // fusion probe (tools/exp/t_fuse_probe.py): what the A phase of a fused A + BC kernel would add to this instruction
      // stream — the finished rows go into an LDS ring of four planes (2 x ds_write_b128 per row) instead of HBM, and the
      // workgroup meets at a barrier once per plane step (where the BC phase would pick the plane up)
      f32x4* ring = reinterpret_cast<f32x4*>(exp_lds) + (threadIdx.x >> 6) * (TH * 2 * 64) + lane;
      if (v1) {
#pragma unroll
        for (int r = 0; r < TH; ++r) ring[((p & 3) * 4 * TH * 2 + r * 2 + 1) * 64] = relu4(acc2[r]);
      }
      if (p - 1 >= d0) {
#pragma unroll
        for (int r = 0; r < TH; ++r) ring[(((p - 1) & 3) * 4 * TH * 2 + r * 2) * 64] = relu4(S[P0][1][r] + shr4(S[P0][0][r]) + shl4(S[P0][2][r]));
      }
      __syncthreads();
      return;
    }
#endif
    if (v1 && !PCGC_ABL(a, 1)) {
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        if constexpr (TRAIN) tb2[(size_t)(p * kD + h0 + r) * 64] = relu4(acc2[r]);
        else tb[((size_t)(p * kD + h0 + r) * 2 + 1) * 64] = relu4(acc2[r]);
      }
    }
    if (p - 1 >= d0 && !PCGC_ABL(a, 1)) {
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const f32x4 y = relu4(S[P0][1][r] + shr4(S[P0][0][r]) + shl4(S[P0][2][r]));
        if constexpr (TRAIN) tb[(size_t)((p - 1) * kD + h0 + r) * 64] = y;
        else tb[((size_t)((p - 1) * kD + h0 + r) * 2 + 0) * 64] = y;
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; p += 3) {
    step(p, I0{}, I1{}, I2{});
    if (p + 1 > d0 + LD) break;
    step(p + 1, I1{}, I2{}, I0{});
    if (p + 2 > d0 + LD) break;
    step(p + 2, I2{}, I0{}, I1{});
  }
}

// ---------------------------------------------------------------------------------------------------------------
// kernel BC:  out = relu( x + [ relu(conv1_2(t11)) (3^3, 4 -> 8) | relu(conv2_3(relu(conv2_2(t21)))) (3^3 4->4, 1^3 4->8) ] )
// weights: conv1_2 [27][4][8]: VGPR tap>>1, abid = (tap&1)*8 + ci*2 + half;  conv2_2 [27][4][4]: VGPR tap>>2,
//          abid = (tap&3)*4 + ci;  conv2_3 [4][8]: one VGPR (lanes 0..31), abid = ci*2 + half
// ---------------------------------------------------------------------------------------------------------------
// P0, P1, P2: which of the three accumulator sets holds output plane p-1, p, p+1 in this step (the plane loop is unrolled
// three times with the roles rotating, so the sets never move between registers).  FRESH: this call holds the first tap
// that reaches each accumulator of set P2 (plane p+1 gets its first contribution, kd = 0, from input plane p): that MFMA
// takes the bias as its C operand instead of the stale accumulator — no initialisation moves.
template <int TH, int P0, int P1, int P2, bool FRESH>
__device__ __forceinline__ void bc_channel12(f32x4 (&acc)[3][TH][2], const f32x4 (&bias)[2], const float (&W)[14], int ci,
                                             const f32x4 (&buf)[TH + 2], bool v0, bool v1, bool v2) {
  float x0[TH + 2], xm[TH + 2], xp[TH + 2];
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) { x0[r] = comp(buf[r], ci); xm[r] = shr1(x0[r]); xp[r] = shl1(x0[r]); }
  const bool vj[3] = {v0, v1, v2};
  constexpr int P[3] = {P0, P1, P2};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int kd = 2 - j;
    if (vj[j]) {
#pragma unroll
      for (int r = 0; r < TH + 2; ++r)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int jr = r - kh;
          if (jr >= 0 && jr < TH) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int t = (kd * 3 + kh) * 3 + kw;
              const float xv = kw == 0 ? xm[r] : (kw == 1 ? x0[r] : xp[r]);
#pragma unroll
              for (int hf = 0; hf < 2; ++hf) {
                const bool first = FRESH && j == 2 && kh == 0 && kw == 0;
                acc[P[j]][jr][hf] = first ? mfa_new((t & 1) * 8 + ci * 2 + hf, W[t >> 1], xv, bias[hf])
                                          : mfa((t & 1) * 8 + ci * 2 + hf, W[t >> 1], xv, acc[P[j]][jr][hf]);
              }
            }
          }
        }
    }
  }
}

template <int TH, int P0, int P1, int P2, bool FRESH>
__device__ __forceinline__ void bc_channel22(f32x4 (&acc)[3][TH], const f32x4& bias, const float (&W)[7], int ci,
                                             const f32x4 (&buf)[TH + 2], bool v0, bool v1, bool v2) {
  float x0[TH + 2], xm[TH + 2], xp[TH + 2];
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) { x0[r] = comp(buf[r], ci); xm[r] = shr1(x0[r]); xp[r] = shl1(x0[r]); }
  const bool vj[3] = {v0, v1, v2};
  constexpr int P[3] = {P0, P1, P2};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int kd = 2 - j;
    if (vj[j]) {
#pragma unroll
      for (int r = 0; r < TH + 2; ++r)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int jr = r - kh;
          if (jr >= 0 && jr < TH) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int t = (kd * 3 + kh) * 3 + kw;
              const float xv = kw == 0 ? xm[r] : (kw == 1 ? x0[r] : xp[r]);
              const bool first = FRESH && j == 2 && kh == 0 && kw == 0;
              acc[P[j]][jr] = first ? mfa_new((t & 3) * 4 + ci, W[t >> 2], xv, bias) : mfa((t & 3) * 4 + ci, W[t >> 2], xv, acc[P[j]][jr]);
            }
          }
        }
    }
  }
}

// The residual rows of the finished plane are requested BEFORE the conv2_2 phase and every load / store of the loop
// is unconditional (an out-of-range buffer offset reads zeros / drops the store), so the epilogue never waits on
// memory and the compiler keeps counted vmcnt waits across the whole loop body.  TH = 2 rows per wave: with 12
// accumulator registers per output row (8 + 4 channels) TH = 4 leaves no room for the residual prefetch
// (measured: 74 us per 8 cubes with TH = 4 and the residual loaded in the epilogue, 64 us in this form).
template <int TH, int LD, bool TRAIN = false, bool NONNEG = false, bool SKIP = false, bool NHWC = TRAIN>
__device__ __forceinline__ void vrn16bc_row_body(const VrnRowArgs& a, int block) {
  static_assert(!(TRAIN && SKIP), "the training step computes every tile");
  static_assert(TRAIN || !NHWC, "the inference tensors are Q4");
  const int lane = threadIdx.x & 63;
  bool heavy = true, unread = false;
  const Tile tl = SKIP ? wave_tile_ordered<TH, LD>(a.skip, &heavy, block, &unread) : wave_tile<TH, LD>(block);
  const int h0 = tl.h0, d0 = tl.d0;
  if (SKIP && !heavy) {
    if (a.skip.materialize == 1 || (a.skip.materialize == 2 && !unread))
      copy_empty_tile<TH, LD, 4>(a.skip.empty, a.out + (size_t)tl.b * kD * kD * kD * 16, tl, lane);
    return;
  }
  float W12[14], W22[7];
#pragma unroll
  for (int v = 0; v < 14; ++v) W12[v] = (v * 64 + lane < 27 * 32) ? a.w12[v * 64 + lane] : 0.f;
#pragma unroll
  for (int v = 0; v < 7; ++v) W22[v] = (v * 64 + lane < 27 * 16) ? a.w22[v * 64 + lane] : 0.f;
  const float W23 = lane < 32 ? a.w23[lane] : 0.f;
  // the biases stay in VGPRs (the asm keeps the compiler from holding these wave-uniform values in SGPRs, from where
  // every use as an MFMA C operand would cost a copy): a fresh accumulator starts as `bias` in its first MFMA
  f32x4 bi12[2] = {{a.b12[0], a.b12[1], a.b12[2], a.b12[3]}, {a.b12[4], a.b12[5], a.b12[6], a.b12[7]}};
  f32x4 bi22 = {a.b22[0], a.b22[1], a.b22[2], a.b22[3]};
  f32x4 bi23[2] = {{a.b23[0], a.b23[1], a.b23[2], a.b23[3]}, {a.b23[4], a.b23[5], a.b23[6], a.b23[7]}};
  asm volatile("" : "+v"(bi12[0]), "+v"(bi12[1]), "+v"(bi22), "+v"(bi23[0]), "+v"(bi23[1]));
  f32x4 acc12[3][TH][2], acc22[3][TH];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < TH; ++r) { acc12[j][r][0] = bi12[0]; acc12[j][r][1] = bi12[1]; acc22[j][r] = bi22; }
  constexpr int TQ = TRAIN ? 1 : 2;                         // quads per voxel of the tensor(s) holding tensor1_1 / tensor2_1
  const i32x4 rs = rsrc_if(make_rsrc(a.t12 + (size_t)tl.b * kD * kD * kD * 4 * TQ, kD * kD * kD * 4 * TQ * 4), !PCGC_ABL(a, 4));
  const i32x4 rs2 = TRAIN ? make_rsrc(a.t21 + (size_t)tl.b * kD * kD * kD * 4, kD * kD * kD * 4 * 4) : rs;
  const i32x4 rx = rsrc_if(make_rsrc(a.x + (size_t)tl.b * kD * kD * kD * 16, kD * kD * kD * 16 * 4), !PCGC_ABL(a, 2));
  const int lane16 = lane * 16;                             // t tensors: one quad per lane in both layouts
  const int lane_x = lane_off<NHWC, 4>(lane);               // x / out / pre
  constexpr int q21 = TRAIN ? 0 : 1;
  f32x4 bufA[TH + 2], bufB[TH + 2];
  // SKIP: tensor1_1 | tensor2_1 rows of tiles kernel A skipped come from ITS empty-cube response, residual rows of tiles
  // the previous block (or conv_in) skipped from that one's
  const i32x4 rsE = SKIP ? make_rsrc(a.skip.in_empty ? a.skip.in_empty : a.t12, kD * kD * kD * 8 * 4) : rs;
  const i32x4 rxE = SKIP ? make_rsrc(a.skip.res_empty ? a.skip.res_empty : a.x, kD * kD * kD * 16 * 4) : rx;
#ifdef PCGC_EXPERIMENTS
  if (PCGC_ABL(a, 64)) {                                    // finite values in the probe's LDS region before anybody reads it
    for (int i = threadIdx.x; i < 4 * 4 * 2 * 2 * 64; i += 256) reinterpret_cast<f32x4*>(exp_lds)[i] = f32x4{0.25f, 0.5f, 0.125f, 1.f};
    __syncthreads();
  }
#endif
  auto rows = [&](f32x4 (&b)[TH + 2], i32x4 r_, int p, int q) {
#ifdef PCGC_EXPERIMENTS
    if (PCGC_ABL(a, 64)) {
      // fusion probe: what the BC phase of a fused kernel would do instead of these buffer loads — the TH + 2 rows of the
      // plane's quad come out of the LDS ring (ds_read_b128, lane = voxel); the barrier sits at the end of the plane step
      const f32x4* ring = reinterpret_cast<const f32x4*>(exp_lds) + lane;
#pragma unroll
      for (int r = 0; r < TH + 2; ++r) b[r] = ring[(((p & 3) * 4 + ((threadIdx.x >> 6) + (r >> 1)) % 4) * 4 + (r & 1) * 2 + (q & 1)) * 64];
      return;
    }
#endif
    if constexpr (SKIP) load_rows_v<TH, TQ>(b, r_, rsE, virtual_rows(a.skip.in_virtual, tl.b, p), lane16, p, q, h0);
    else load_rows<TH, TQ>(b, r_, lane16, p, q, h0);
  };
  rows(bufA, rs, d0 - 1, 0);
  rows(bufB, rs2, d0 - 1, q21);
  // TRAIN with sign words: the signs of the wave's own rows of tensor1_1 / tensor2_1 (bits 0-3 / 4-7), taken while plane p
  // is in the buffers and stored one step later with the finished plane's word
  unsigned tb_prev[TH];
#pragma unroll
  for (int r = 0; r < TH; ++r) tb_prev[r] = 0;
  auto nibble = [](const f32x4& v) { return (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u); };
  // one input plane: sets P0 / P1 / P2 = output planes p-1 / p / p+1
  auto step = [&](int p, auto P0_, auto P1_, auto P2_) {
    constexpr int P0 = decltype(P0_)::value, P1 = decltype(P1_)::value, P2 = decltype(P2_)::value;
    const bool pin = (unsigned)p < (unsigned)kD;
    // v2 does not ask for the plane to exist: plane p + 1's accumulators are BORN in this step (bias as the C operand of
    // their first MFMA), and an input plane outside the cube reads zeros
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = p + 1 < d0 + LD;
    bc_channel12<TH, P0, P1, P2, true>(acc12, bi12, W12, 0, bufA, v0, v1, v2);
#pragma unroll
    for (int c = 1; c < 4; ++c) bc_channel12<TH, P0, P1, P2, false>(acc12, bi12, W12, c, bufA, v0, v1, v2);
    unsigned tb_cur[TH];
    if constexpr (TRAIN) {
#pragma unroll
      for (int r = 0; r < TH; ++r) tb_cur[r] = a.pre_signs ? nibble(bufA[r + 1]) : 0u;
    }
    rows(bufA, rs, p + 1, 0);
    // residual rows of output plane p-1 (out of range before the first finished plane: zeros, and the stores drop)
    const bool done = p - 1 >= d0;
    const int obase = done ? row_off<NHWC, 4>(p - 1, h0, 0) : 0;      // scalar: the lane's part is the vector offset
    i32x4 rxo = rsrc_if(rx, done);
    if constexpr (SKIP) {                                  // the wave's own rows (both in one tile of the producer: TH = 2 everywhere)
      const bool e = done && ((virtual_rows(a.skip.res_virtual, tl.b, p - 1) >> h0) & 1ull);
      rxo[0] = e ? rxE[0] : rxo[0];
      rxo[1] = e ? rxE[1] : rxo[1];
    }
    f32x4 res[TH][4];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        res[r][q] = PCGC_ABL(a, 8) ? raw_load4(rxo, lane_x, obase + row_off<NHWC, 4>(0, r, q), 2)
                                   : raw_load4(rxo, lane_x, obase + row_off<NHWC, 4>(0, r, q), 0);
    bc_channel22<TH, P0, P1, P2, true>(acc22, bi22, W22, 0, bufB, v0, v1, v2);
#pragma unroll
    for (int c = 1; c < 4; ++c) bc_channel22<TH, P0, P1, P2, false>(acc22, bi22, W22, c, bufB, v0, v1, v2);
    if constexpr (TRAIN) {
      if (a.pre_signs) {
#pragma unroll
        for (int r = 0; r < TH; ++r) tb_cur[r] |= nibble(bufB[r + 1]) << 4;
      }
    }
    rows(bufB, rs2, p + 1, q21);
    // output plane p-1: conv2_3 on relu(conv2_2) (rows interleaved: independent MFMA chains), residual, ReLU, store
    f32x4 t22[TH], q3[TH][2];
#pragma unroll
    for (int r = 0; r < TH; ++r) t22[r] = relu4(acc22[P0][r]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
          q3[r][hf] = c == 0 ? mfa_new(c * 2 + hf, W23, comp(t22[r], c), bi23[hf]) : mfa(c * 2 + hf, W23, comp(t22[r], c), q3[r][hf]);
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      const f32x4 pr[4] = {relu4(acc12[P0][r][0]), relu4(acc12[P0][r][1]), relu4(q3[r][0]), relu4(q3[r][1])};
      // stores: the row in the descriptor base, the quad as a constant offset (rsrc_at)
      const int orow = obase + row_off<NHWC, 4>(0, r, 0);
      const i32x4 roo = rsrc_at(a.out + (size_t)tl.b * kD * kD * kD * 16, orow, done && !PCGC_ABL(a, 1));
#pragma unroll
      // NONNEG: the block input is a ReLU output and pr >= 0, so the sum needs no second ReLU (bit-identical)
      for (int q = 0; q < 4; ++q) {
        const f32x4 y = NONNEG ? res[r][q] + pr[q] : relu4(res[r][q] + pr[q]);
        if (PCGC_ABL(a, 16)) raw_store4(y, roo, lane_x + row_off<NHWC, 4>(0, 0, q), 0, 2);
        else raw_store4(y, roo, lane_x + row_off<NHWC, 4>(0, 0, q), 0, 0);
      }
      if constexpr (TRAIN) {                                // what the reverse pass reads: tensor2_2 and the pre-residual output
        const i32x4 r22 = rsrc_at(a.t22 + (size_t)tl.b * kD * kD * kD * 4, done ? row_off<false, 1>(p - 1, h0 + r, 0) : 0, done);
        if (a.pre_signs) {
          // bit c = pre[c] > 0 (c < 16); bits 16-19 = tensor2_2 > 0, 20-23 = tensor1_1 > 0, 24-27 = tensor2_1 > 0 — every mask
          // the block's reverse needs, 4 B per voxel (the reverse of the tail reads no activation tensor)
          unsigned m = (nibble(t22[r]) << 16) | (tb_prev[r] << 20);
#pragma unroll
          for (int c = 15; c >= 0; --c) m |= (pr[c >> 2][c & 3] > 0.f ? 1u : 0u) << c;
          const i32x4 rm = rsrc_at(a.pre_signs + (size_t)tl.b * kD * kD * kD, done ? ((p - 1) * kD + h0 + r) * kD * 4 : 0, done);
          raw_store1i((int)m, rm, lane * 4, 0, 0);
        } else {
          const i32x4 rp = rsrc_at(a.pre + (size_t)tl.b * kD * kD * kD * 16, orow, done && !PCGC_ABL(a, 32));
#pragma unroll
          for (int q = 0; q < 4; ++q) raw_store4(pr[q], rp, lane_x + row_off<NHWC, 4>(0, 0, q), 0, 0);
        }
        raw_store4(t22[r], r22, lane16, 0, 0);
      }
    }
    if constexpr (TRAIN) {
#pragma unroll
      for (int r = 0; r < TH; ++r) tb_prev[r] = tb_cur[r];
    }
#ifdef PCGC_EXPERIMENTS
    if (PCGC_ABL(a, 64)) __syncthreads();
#endif
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; p += 3) {            // roles rotate instead of registers
    step(p, I0{}, I1{}, I2{});
    if (p + 1 > d0 + LD) break;
    step(p + 1, I1{}, I2{}, I0{});
    if (p + 2 > d0 + LD) break;
    step(p + 2, I2{}, I0{}, I1{});
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Reverse of kernel A (training step):  dx = [x > 0] * ( dpre + conv1_1^T(dt11) (3^3, 4 -> 16) + conv2_1^T(dt21) (1^3, 4 -> 16) )
// — the three gradient contributions of the block input (train_hyper.py:_vrn_bwd) in one pass instead of two bwd-data
// launches that each re-read and re-write the 16-channel gradient.  Tensors NDHWC: dt11 / dt21 [B][64][64][64][4],
// dpre / x / dx [..][16]; dx may alias dpre (every element is read by the lane that writes it).
// A transposed stride-1 convolution is a convolution with the taps mirrored and the channel roles swapped:
//   dx[ci](v) = sum_t sum_co W[t][ci][co] g[co](v - off(t))  =  sum_t K[t][co][ci] g[co](v + off(t)),  K[t] = W[26 - t]^T
// weights: VGPR t = K[t] as [g channel 4][dx channel 16] gathered from the TensorFlow layout (lane = gch * 16 + dxch,
// abid = gch * 4 + dx quad); the 1^3 layer is one more register.  Input-shift form (4 input channels, 16 outputs).
// Summation order per output: (plane, g channel, kh, kw) of conv1_1^T, then the 4 channels of conv2_1^T, then + dpre.
// ---------------------------------------------------------------------------------------------------------------
struct VrnBwdInArgs {
  const float *dt11, *dt21, *dpre, *x;   // x = nullptr: no mask (the block input is not a ReLU output)
  const float *w11, *w21;                // TensorFlow layouts [3][3][3][16][4], [1][1][1][16][4]
  float* dx;
  int B;
};

template <int TH, int P0, int P1, int P2, bool FRESH>
__device__ __forceinline__ void bwd_in_channel(f32x4 (&acc)[3][TH][4], const float (&W)[27], int ci, const f32x4 (&buf)[TH + 2], bool v0,
                                               bool v1, bool v2) {
  float x0[TH + 2], xm[TH + 2], xp[TH + 2];
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) { x0[r] = comp(buf[r], ci); xm[r] = shr1(x0[r]); xp[r] = shl1(x0[r]); }
  const bool vj[3] = {v0, v1, v2};
  constexpr int P[3] = {P0, P1, P2};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int kd = 2 - j;
    if (vj[j]) {
#pragma unroll
      for (int r = 0; r < TH + 2; ++r)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int jr = r - kh;
          if (jr >= 0 && jr < TH) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int t = (kd * 3 + kh) * 3 + kw;
              const float xv = kw == 0 ? xm[r] : (kw == 1 ? x0[r] : xp[r]);
              const bool first = FRESH && j == 2 && kh == 0 && kw == 0;
#pragma unroll
              for (int q = 0; q < 4; ++q)
                acc[P[j]][jr][q] = first ? mfa_new(ci * 4 + q, W[t], xv, zero) : mfa(ci * 4 + q, W[t], xv, acc[P[j]][jr][q]);
            }
          }
        }
    }
  }
}

template <int TH, int LD, bool MASK, bool NHWC = true>
__global__ void __launch_bounds__(256, 2) vrn16a_bwd_row_kernel(VrnBwdInArgs a) {
  const int lane = threadIdx.x & 63;
  const Tile tl = wave_tile<TH, LD>();
  const int h0 = tl.h0, d0 = tl.d0;
  float W[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) W[t] = a.w11[((26 - t) * 16 + (lane & 15)) * 4 + (lane >> 4)];
  const float W1 = a.w21[(lane & 15) * 4 + (lane >> 4)];
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[3][TH][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[j][r][q] = zero;
  constexpr int kCube4 = kD * kD * kD * 4, kCube16 = kD * kD * kD * 16;
  const i32x4 rg = make_rsrc(a.dt11 + (size_t)tl.b * kCube4, kCube4 * 4);
  const i32x4 rg2 = make_rsrc(a.dt21 + (size_t)tl.b * kCube4, kCube4 * 4);
  const i32x4 rp = make_rsrc(a.dpre + (size_t)tl.b * kCube16, kCube16 * 4);
  const i32x4 rx = MASK ? make_rsrc(a.x + (size_t)tl.b * kCube16, kCube16 * 4) : rp;
  const int lane16 = lane * 16;                             // 4-channel tensors: one quad per voxel
  const int lane_x = lane_off<NHWC, 4>(lane);               // 16-channel NDHWC tensors
  f32x4 buf[TH + 2];
  load_rows<TH, 1>(buf, rg, lane16, d0 - 1, 0, h0);
  auto step = [&](int p, auto P0_, auto P1_, auto P2_) {
    constexpr int P0 = decltype(P0_)::value, P1 = decltype(P1_)::value, P2 = decltype(P2_)::value;
    const bool pin = (unsigned)p < (unsigned)kD;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = p + 1 < d0 + LD;
    // what the finished plane p - 1 needs besides its sums: its dt21 rows, the gradient arriving over the skip
    // connection and the sign of the block input — requested before the MFMAs of this step
    const bool done = p - 1 >= d0;
    const int obase = done ? row_off<NHWC, 4>(p - 1, h0, 0) : 0;
    const i32x4 rpo = rsrc_if(rp, done), rxo = rsrc_if(rx, done);
    f32x4 g2[TH], res[TH][4], xs[TH][4];
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      g2[r] = raw_load4(rsrc_if(rg2, done), lane16, done ? row_off<false, 1>(p - 1, h0 + r, 0) : 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        res[r][q] = raw_load4(rpo, lane_x, obase + row_off<NHWC, 4>(0, r, q), 0);
        if constexpr (MASK) xs[r][q] = raw_load4(rxo, lane_x, obase + row_off<NHWC, 4>(0, r, q), 0);
      }
    }
    bwd_in_channel<TH, P0, P1, P2, true>(acc, W, 0, buf, v0, v1, v2);
#pragma unroll
    for (int c = 1; c < 4; ++c) bwd_in_channel<TH, P0, P1, P2, false>(acc, W, c, buf, v0, v1, v2);
    load_rows<TH, 1>(buf, rg, lane16, p + 1, 0, h0);
    // output plane p - 1: the 1^3 layer's part, the skip gradient, the mask, store
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[P0][r][q] = mfa(c * 4 + q, W1, comp(g2[r], c), acc[P0][r][q]);
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      const i32x4 roo = rsrc_at(a.dx + (size_t)tl.b * kCube16, obase + row_off<NHWC, 4>(0, r, 0), done);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 y = acc[P0][r][q] + res[r][q];
        if constexpr (MASK) {
#pragma unroll
          for (int i = 0; i < 4; ++i) y[i] = xs[r][q][i] > 0.f ? y[i] : 0.f;
        }
        raw_store4(y, roo, lane_x + row_off<NHWC, 4>(0, 0, q), 0, 0);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; p += 3) {
    step(p, I0{}, I1{}, I2{});
    if (p + 1 > d0 + LD) break;
    step(p + 1, I1{}, I2{}, I0{});
    if (p + 2 > d0 + LD) break;
    step(p + 2, I2{}, I0{}, I1{});
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Reverse of kernel BC's convolutions (training step), one pass instead of three bwd-data launches:
//   dt11 = [t11 > 0] * conv1_2^T(dz12)                      3^3, 8 -> 4
//   dt22 = [t22 > 0] * conv2_3^T(dz23)                      1^3, 8 -> 4   (also the dz of conv2_2's weight gradient)
//   dt21 = [t21 > 0] * conv2_2^T(dt22)                      3^3, 4 -> 4
// dz12 / dz23 [B][64][64][64][8] are the block tail's reverse (pcgc_vrn_bwd_split), t11 / t21 / t22 and the results
// [..][4], all NDHWC.  dt22 of a plane is made on the fly from dz23 for the TH + 2 rows conv2_2^T needs (the 1^3 layer
// is 8 MFMAs per row) and written for the wave's own rows; planes slide as in the forward kernels.  Adjoint filters
// gathered from the TensorFlow layouts: K12[t][co 8][ci 4] = w12[26 - t][ci][co] (two taps per register, abid =
// (t & 1) * 8 + co), K23[co 8][ci 4] = w23[ci][co] (abid = co), K22[t][co 4][ci 4] = w22[26 - t][ci][co] (four taps per
// register, abid = (t & 3) * 4 + co).  Summation order per output: (plane, channel, kh, kw), channels of the 1^3 layer 0..7.
// ---------------------------------------------------------------------------------------------------------------
struct VrnBwdTailArgs {
  const float *dz12, *dz23, *t11, *t21, *t22;
  const float *w12, *w22, *w23;
  float *dt11, *dt21, *dt22;
  int B;
  // SPLIT = true: the block tail's reverse happens here too (pcgc_vrn_bwd_split_signs folded in): dz12 / dz23 are made
  // from the incoming gradient dout [..][16] (already masked by out > 0) and the sign bits of `pre` for every row the
  // wave touches, and WRITTEN to dz12w / dz23w for its own rows (the weight gradients of conv1_2 / conv2_3 read them); the
  // masks t22 > 0, t11 > 0, t21 > 0 come from bits 16-19 / 20-23 / 24-27 of the same words (vrn16bc_row_body writes them), so
  // t11 / t21 / t22 are NOT read: 184 instead of 228 bytes per voxel
  const float* dout = nullptr;
  const int* signs = nullptr;
  float *dz12w = nullptr, *dz23w = nullptr;
};

template <int TH, int LD, bool SPLIT = false, bool NHWC = true>
__global__ void __launch_bounds__(256, 2) vrn16bc_bwd_row_kernel(VrnBwdTailArgs a) {
  const int lane = threadIdx.x & 63;
  const Tile tl = wave_tile<TH, LD>();
  const int h0 = tl.h0, d0 = tl.d0;
  float W12[14], W22[7];
#pragma unroll
  for (int v = 0; v < 14; ++v) {
    const int t = 2 * v + (lane >> 5);
    W12[v] = t < 27 ? a.w12[((26 - t) * 4 + (lane & 3)) * 8 + ((lane >> 2) & 7)] : 0.f;
  }
#pragma unroll
  for (int v = 0; v < 7; ++v) {
    const int t = 4 * v + (lane >> 4);
    W22[v] = t < 27 ? a.w22[((26 - t) * 4 + (lane & 3)) * 4 + ((lane >> 2) & 3)] : 0.f;
  }
  const float W23 = lane < 32 ? a.w23[(lane & 3) * 8 + (lane >> 2)] : 0.f;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc12[3][TH], acc22[3][TH];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < TH; ++r) { acc12[j][r] = zero; acc22[j][r] = zero; }
  constexpr int kCube4 = kD * kD * kD * 4, kCube8 = kD * kD * kD * 8;
  const i32x4 r12 = make_rsrc(a.dz12 + (size_t)tl.b * kCube8, kCube8 * 4);
  const i32x4 r23 = make_rsrc(a.dz23 + (size_t)tl.b * kCube8, kCube8 * 4);
  const i32x4 rt11 = make_rsrc(a.t11 + (size_t)tl.b * kCube4, kCube4 * 4);
  const i32x4 rt21 = make_rsrc(a.t21 + (size_t)tl.b * kCube4, kCube4 * 4);
  const i32x4 rt22 = make_rsrc(a.t22 + (size_t)tl.b * kCube4, kCube4 * 4);
  const int lane8 = lane_off<NHWC, 2>(lane), lane16 = lane * 16;
  constexpr int kCube16 = kD * kD * kD * 16;
  const i32x4 rdo = SPLIT ? make_rsrc(a.dout + (size_t)tl.b * kCube16, kCube16 * 4) : r12;
  const i32x4 rsg = SPLIT ? make_rsrc(a.signs + (size_t)tl.b * kD * kD * kD, kD * kD * kD * 4) : r12;
  const int lane_x = lane_off<NHWC, 4>(lane);
  f32x4 in12[2][TH + 2], in23[2][TH + 2], m22[TH + 2];
  float sg[TH + 2];                                         // SPLIT: the rows' sign words, applied at the start of the step
  auto load_plane = [&](int p) {
    if constexpr (SPLIT) {                                  // raw gradient rows now, masked when the step that uses them begins
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        load_rows<TH, 4, NHWC>(in12[q], rdo, lane_x, p, q, h0);
        load_rows<TH, 4, NHWC>(in23[q], rdo, lane_x, p, 2 + q, h0);
      }
#pragma unroll
      for (int r = 0; r < TH + 2; ++r) {
        const int h = h0 - 1 + r;
        const bool ok = (unsigned)h < (unsigned)kD && (unsigned)p < (unsigned)kD;
        sg[r] = raw_load1(rsrc_if(rsg, ok), lane * 4, ok ? (p * kD + h) * kD * 4 : 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        load_rows<TH, 2, NHWC>(in12[q], r12, lane8, p, q, h0);
        load_rows<TH, 2, NHWC>(in23[q], r23, lane8, p, q, h0);
      }
    }
    if constexpr (!SPLIT) load_rows<TH, 1>(m22, rt22, lane16, p, 0, h0);     // SPLIT: tensor2_2's signs are bits 16-19 of the word
  };
  load_plane(d0 - 1);
  float sgp[TH];                                            // SPLIT: the sign words of the finished plane's rows (plane p - 1)
#pragma unroll
  for (int r = 0; r < TH; ++r) sgp[r] = 0.f;
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kD;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
    const bool vj[3] = {v0, v1, v2};
    const bool done = p - 1 >= d0;
    // the finished plane's masks, requested before the MFMAs of this step
    f32x4 k11[TH], k21[TH];
    if constexpr (!SPLIT) {                                 // SPLIT: bits 20-23 / 24-27 of the rows' sign words instead
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const int row = done ? row_off<false, 1>(p - 1, h0 + r, 0) : 0;
        k11[r] = raw_load4(rsrc_if(rt11, done), lane16, row, 0);
        k21[r] = raw_load4(rsrc_if(rt21, done), lane16, row, 0);
      }
    }
    if constexpr (SPLIT) {                                  // dz12 / dz23 of this plane's rows: the gradient where pre > 0
#pragma unroll
      for (int r = 0; r < TH + 2; ++r) {
        const unsigned m = __builtin_bit_cast(unsigned, sg[r]);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            in12[q][r][i] = (m >> (4 * q + i)) & 1u ? in12[q][r][i] : 0.f;
            in23[q][r][i] = (m >> (8 + 4 * q + i)) & 1u ? in23[q][r][i] : 0.f;
          }
      }
    }
    // conv1_2^T: 8 input channels of dz12
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
      float x0[TH + 2], xm[TH + 2], xp[TH + 2];
#pragma unroll
      for (int r = 0; r < TH + 2; ++r) { x0[r] = comp(in12[c8 >> 2][r], c8 & 3); xm[r] = shr1(x0[r]); xp[r] = shl1(x0[r]); }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int kd = 2 - j;
        if (vj[j]) {
#pragma unroll
          for (int r = 0; r < TH + 2; ++r)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              const int jr = r - kh;
              if (jr >= 0 && jr < TH) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                  const int t = (kd * 3 + kh) * 3 + kw;
                  const float xv = kw == 0 ? xm[r] : (kw == 1 ? x0[r] : xp[r]);
                  acc12[j][jr] = mfa((t & 1) * 8 + c8, W12[t >> 1], xv, acc12[j][jr]);
                }
              }
            }
        }
      }
    }
    // dt22 of this plane for the TH + 2 rows: the 1^3 layer's reverse, masked by t22 > 0
    f32x4 d22[TH + 2];
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) {
      f32x4 g = zero;
#pragma unroll
      for (int co = 0; co < 8; ++co) g = mfa(co, W23, comp(in23[co >> 2][r], co & 3), g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (SPLIT) g[i] = (__builtin_bit_cast(unsigned, sg[r]) >> (16 + i)) & 1u ? g[i] : 0.f;
        else g[i] = m22[r][i] > 0.f ? g[i] : 0.f;
      }
      d22[r] = g;
    }
    if (v1) {                                               // the wave's own rows of dt22 (conv2_2's weight gradient reads them)
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        raw_store4(d22[r + 1], rsrc_at(a.dt22 + (size_t)tl.b * kCube4, row_off<false, 1>(p, h0 + r, 0), true), lane16, 0, 0);
        if constexpr (SPLIT) {                              // ... and of dz12 / dz23
          const i32x4 w12r = rsrc_at(a.dz12w + (size_t)tl.b * kCube8, row_off<NHWC, 2>(p, h0 + r, 0), true);
          const i32x4 w23r = rsrc_at(a.dz23w + (size_t)tl.b * kCube8, row_off<NHWC, 2>(p, h0 + r, 0), true);
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            raw_store4(in12[q][r + 1], w12r, lane8 + row_off<NHWC, 2>(0, 0, q), 0, 0);
            raw_store4(in23[q][r + 1], w23r, lane8 + row_off<NHWC, 2>(0, 0, q), 0, 0);
          }
        }
      }
    }
    // conv2_2^T on dt22
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x0[TH + 2], xm[TH + 2], xp[TH + 2];
#pragma unroll
      for (int r = 0; r < TH + 2; ++r) { x0[r] = comp(d22[r], c); xm[r] = shr1(x0[r]); xp[r] = shl1(x0[r]); }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int kd = 2 - j;
        if (vj[j]) {
#pragma unroll
          for (int r = 0; r < TH + 2; ++r)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              const int jr = r - kh;
              if (jr >= 0 && jr < TH) {
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                  const int t = (kd * 3 + kh) * 3 + kw;
                  const float xv = kw == 0 ? xm[r] : (kw == 1 ? x0[r] : xp[r]);
                  acc22[j][jr] = mfa((t & 3) * 4 + c, W22[t >> 2], xv, acc22[j][jr]);
                }
              }
            }
        }
      }
    }
    float sg_own[TH];
#pragma unroll
    for (int r = 0; r < TH; ++r) sg_own[r] = SPLIT ? sg[r + 1] : 0.f;
    load_plane(p + 1);
    // output plane p - 1: masks, stores
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      f32x4 y11 = acc12[0][r], y21 = acc22[0][r];
      const unsigned mp = __builtin_bit_cast(unsigned, sgp[r]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (SPLIT) {
          y11[i] = (mp >> (20 + i)) & 1u ? y11[i] : 0.f;
          y21[i] = (mp >> (24 + i)) & 1u ? y21[i] : 0.f;
        } else {
          y11[i] = k11[r][i] > 0.f ? y11[i] : 0.f;
          y21[i] = k21[r][i] > 0.f ? y21[i] : 0.f;
        }
      }
      const int row = done ? row_off<false, 1>(p - 1, h0 + r, 0) : 0;
      raw_store4(y11, rsrc_at(a.dt11 + (size_t)tl.b * kCube4, row, done), lane16, 0, 0);
      raw_store4(y21, rsrc_at(a.dt21 + (size_t)tl.b * kCube4, row, done), lane16, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < TH; ++r) {
      acc12[0][r] = acc12[1][r]; acc12[1][r] = acc12[2][r]; acc12[2][r] = zero;
      acc22[0][r] = acc22[1][r]; acc22[1][r] = acc22[2][r]; acc22[2][r] = zero;
      sgp[r] = sg_own[r];
    }
  }
}

template <int TH, int LD, bool TRAIN = false, bool SKIP = false, bool NHWC = TRAIN, bool QUADJ = true>
__global__ void __launch_bounds__(256, 2) vrn16a_row_kernel(VrnRowArgs a) { vrn16a_row_body<TH, LD, TRAIN, SKIP, NHWC, QUADJ>(a, blockIdx.x); }
template <int TH, int LD, bool TRAIN = false, bool NONNEG = false, bool SKIP = false, bool NHWC = TRAIN>
__global__ void __launch_bounds__(256, 2) vrn16bc_row_kernel(VrnRowArgs a) { vrn16bc_row_body<TH, LD, TRAIN, NONNEG, SKIP, NHWC>(a, blockIdx.x); }

// ---------------------------------------------------------------------------------------------------------------
// conv_in: x [B][64][64][64] (one channel) -> y Q4 [B][64][64][4][64][4], relu(conv 3^3, 1 -> 16 + bias)
// weights [27][1][16]: VGPR tap>>2, abid = (tap&3)*4 + cout quad
// ---------------------------------------------------------------------------------------------------------------
struct ConvRowArgs {
  const float* x;
  float* y;
  const float* w;
  const float* bias;
  int B, relu;
  const float* mask = nullptr;   // conv_in only: y = mask > 0 ? conv : 0 (mask Q4 like y: the training step's reverse of deconv_out)
  RowSkip skip;        // conv_in only (see VrnRowArgs)
  int remap = 0;       // 1: every XCD walks a contiguous range of tiles (xcd_remap): halo rows shared by neighbouring tiles hit one L2
};

template <int TH, int LD>
__global__ void __launch_bounds__(256, 2) conv_in_row_kernel(ConvRowArgs a) {
  const int lane = threadIdx.x & 63;
  bool heavy = true;
  const Tile tl = wave_tile_ordered<TH, LD>(a.skip, &heavy);
  const int h0 = tl.h0, d0 = tl.d0;
  if (!heavy) {
    if (a.skip.materialize) copy_empty_tile<TH, LD, 4>(a.skip.empty, a.y + (size_t)tl.b * kD * kD * kD * 16, tl, lane);
    return;
  }
  float W[7];
#pragma unroll
  for (int v = 0; v < 7; ++v) W[v] = (v * 64 + lane < 27 * 16) ? a.w[v * 64 + lane] : 0.f;
  f32x4 bi[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bi[q] = a.bias ? f32x4{a.bias[4 * q], a.bias[4 * q + 1], a.bias[4 * q + 2], a.bias[4 * q + 3]} : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc[3][TH][4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[j][r][q] = bi[q];
  const i32x4 rs = make_rsrc(a.x + (size_t)tl.b * kD * kD * kD, kD * kD * kD * 4);
  const int lane4 = lane * 4;
  f32x4* yb = reinterpret_cast<f32x4*>(a.y) + (size_t)tl.b * kD * kD * 4 * 64 + lane;
  auto load_plane = [&](float (&buf)[TH + 2], int p) {
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) {
      const int h = h0 - 1 + r;
      const bool ok = (unsigned)h < (unsigned)kD && (unsigned)p < (unsigned)kD;
      buf[r] = raw_load1(rs, (ok ? (p * kD + h) * (kD * 4) : kOOB) + lane4, 0, 0);
    }
  };
  float cur[TH + 2], nxt[TH + 2];
  load_plane(cur, d0 - 1);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kD;
    const bool vj[3] = {pin && p - 1 >= d0, pin && p >= d0 && p < d0 + LD, pin && p + 1 < d0 + LD};
    load_plane(nxt, p + 1);
    float xm[TH + 2], xp[TH + 2];
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) { xm[r] = shr1(cur[r]); xp[r] = shl1(cur[r]); }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int kd = 2 - j;
      if (vj[j]) {
#pragma unroll
        for (int r = 0; r < TH + 2; ++r)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int jr = r - kh;
            if (jr >= 0 && jr < TH) {
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) {
                const int t = (kd * 3 + kh) * 3 + kw;
                const float xv = kw == 0 ? xm[r] : (kw == 1 ? cur[r] : xp[r]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][jr][q] = mfa((t & 3) * 4 + q, W[t >> 2], xv, acc[j][jr][q]);
              }
            }
          }
      }
    }
    if (p - 1 >= d0) {
#pragma unroll
      for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
          const size_t at = ((size_t)((p - 1) * kD + h0 + r) * 4 + q) * 64;
          f32x4 v = a.relu ? relu4(acc[0][r][q]) : acc[0][r][q];
          if (a.mask) {                                       // wave-uniform
            const f32x4 m = (reinterpret_cast<const f32x4*>(a.mask) + (size_t)tl.b * kD * kD * 4 * 64 + lane)[at];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = m[i] > 0.f ? v[i] : 0.f;
          }
          yb[at] = v;
        }
    }
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc[0][r][q] = acc[1][r][q]; acc[1][r][q] = acc[2][r][q]; acc[2][r][q] = bi[q]; }
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) cur[r] = nxt[r];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// deconv_out: x Q4 [B][64][64][4][64][4] -> y [B][64][64][64] (one channel), conv 3^3, 16 -> 1 + bias (no activation)
// One output channel would use one MFMA row in four.  Instead the three kw taps are the rows: with
//   A[i] = W[kd][kh][kw = i][ci] (i = 3: zero),  B = the UNSHIFTED input row,
// the accumulator row i holds P_i[w'] = sum_{kd,kh,ci} W[kd,kh,i,ci] x[.., w', ci], and the output is
//   y[w] = P_0[w-1] + P_1[w] + P_2[w+1] + bias        (two lane shifts per output row, none in the loop)
// 144 MFMAs per output row instead of 432.  Weight VGPR v = kd*3+kh holds lane (ci*4 + i) = W[(v*3+i)][ci]; abid = ci.
// ---------------------------------------------------------------------------------------------------------------
template <int TH, int LD>
__global__ void __launch_bounds__(256, 2) deconv_out_row_kernel(ConvRowArgs a) {
  const int lane = threadIdx.x & 63;
  const Tile tl = wave_tile<TH, LD>(a.remap ? xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x);
  const int h0 = tl.h0, d0 = tl.d0;
  float W[9];
#pragma unroll
  for (int v = 0; v < 9; ++v) W[v] = (lane & 3) < 3 ? a.w[(v * 3 + (lane & 3)) * 16 + (lane >> 2)] : 0.f;
  const float bias = a.bias ? a.bias[0] : 0.f;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[3][TH];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < TH; ++r) acc[j][r] = zero;
  const i32x4 rs = make_rsrc(a.x + (size_t)tl.b * kD * kD * kD * 16, kD * kD * kD * 16 * 4);
  const int lane16 = lane * 16;
  float* yb = a.y + (size_t)tl.b * kD * kD * kD + lane;
  f32x4 bufA[TH + 2], bufB[TH + 2];
  load_rows<TH, 4>(bufA, rs, lane16, d0 - 1, 0, h0);
  auto quad = [&](const f32x4 (&buf)[TH + 2], int q, const bool (&vj)[3]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int kd = 2 - j;
        if (vj[j]) {
#pragma unroll
          for (int r = 0; r < TH + 2; ++r)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              const int jr = r - kh;
              if (jr >= 0 && jr < TH) acc[j][jr] = mfa(q * 4 + c, W[kd * 3 + kh], comp(buf[r], c), acc[j][jr]);
            }
        }
      }
  };
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kD;
    const bool vj[3] = {pin && p - 1 >= d0, pin && p >= d0 && p < d0 + LD, pin && p + 1 < d0 + LD};
    load_rows<TH, 4>(bufB, rs, lane16, p, 1, h0);
    quad(bufA, 0, vj);
    load_rows<TH, 4>(bufA, rs, lane16, p, 2, h0);
    quad(bufB, 1, vj);
    load_rows<TH, 4>(bufB, rs, lane16, p, 3, h0);
    quad(bufA, 2, vj);
    load_rows<TH, 4>(bufA, rs, lane16, p + 1, 0, h0);
    quad(bufB, 3, vj);
    if (p - 1 >= d0) {
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        float v = bias + shr1(acc[0][r][0]);
        v += acc[0][r][1];
        v += shl1(acc[0][r][2]);
        yb[(size_t)((p - 1) * kD + h0 + r) * kD] = a.relu ? fmaxf(v, 0.f) : v;
      }
    }
#pragma unroll
    for (int r = 0; r < TH; ++r) { acc[0][r] = acc[1][r]; acc[1][r] = acc[2][r]; acc[2][r] = zero; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// layout conversion [B][V][C] (NDHWC) <-> Q4 [B][D][H][C/4][W][4].  For the stand-alone block entry
// (pcgc_vrn_fwd) and tests; the transforms keep the full-resolution stage in Q4 end to end.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) q4_convert_kernel(const float* src, float* dst, int64_t rows, int NQ, int W, int to_q4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // float4 index in the Q4 tensor
  if (i >= rows * NQ * W) return;
  const int w = (int)(i % W);
  const int q = (int)((i / W) % NQ);
  const int64_t row = i / ((int64_t)W * NQ);
  const int64_t nd = (row * W + w) * NQ + q;                    // float4 index in the NDHWC tensor
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
  if (to_q4) d4[i] = s4[nd]; else d4[nd] = s4[i];
}

int launch_q4_convert(const float* src, float* dst, int B, int D, int C, int to_q4, hipStream_t s) {
  const int64_t rows = (int64_t)B * D * D;
  const int64_t n = rows * (C / 4) * D;
  hipLaunchKernelGGL(q4_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, rows, C / 4, D, to_q4);
  return launch_ok("q4_convert_kernel");
}

static VrnRowArgs vrn_args(const float* x, float* t12, float* out, const float* const* w, int B) {
  VrnRowArgs a;
  a.x = x; a.t12 = t12; a.out = out;
  a.w11 = w[0]; a.b11 = w[1]; a.w12 = w[2]; a.b12 = w[3]; a.w21 = w[4]; a.b21 = w[5];
  a.w22 = w[6]; a.b22 = w[7]; a.w23 = w[8]; a.b23 = w[9];
  a.B = B;
  return a;
}

// which: 0 = kernel A, 1 = kernel BC.  All tensors Q4, D = 64.  w = {w11,b11,w12,b12,w21,b21,w22,b22,w23,b23}
#ifdef PCGC_EXPERIMENTS
int g_vrn16_abl = 0;   // set by pcgc_exp_vrn16_row: exists in experiment builds only (the product library has no global state)
#define PCGC_ABL_VALUE g_vrn16_abl
#else
#define PCGC_ABL_VALUE 0
#endif
int launch_vrn16_row(const float* x, float* t12, float* out, const float* const* w, int B, int which, hipStream_t s, bool x_nonneg,
                     const RowSkip* skip) {
  VrnRowArgs a = vrn_args(x, t12, out, w, B);
  if (skip) a.skip = *skip;
  a.abl = PCGC_ABL_VALUE;
  // A: 2 rows x 8 planes per wave, BC: 2 rows x 8 planes: 2048 waves per 8 cubes = two per SIMD, all resident
  const dim3 grid(B * (kD / 2) * (kD / 8) / 4);
  if (a.skip.order) {                                       // analysis with empty-space skipping (x_nonneg holds there: every block follows a ReLU)
    if (which == 0) hipLaunchKernelGGL((vrn16a_row_kernel<2, 8, false, true>), grid, dim3(256), 0, s, a);
    else if (x_nonneg) hipLaunchKernelGGL((vrn16bc_row_kernel<2, 8, false, true, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((vrn16bc_row_kernel<2, 8, false, false, true>), grid, dim3(256), 0, s, a);
  } else if (which == 0) hipLaunchKernelGGL((vrn16a_row_kernel<2, 8>), grid, dim3(256), 0, s, a);
  else if (x_nonneg) hipLaunchKernelGGL((vrn16bc_row_kernel<2, 8, false, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((vrn16bc_row_kernel<2, 8>), grid, dim3(256), 0, s, a);
  return launch_ok("vrn16 row kernel");
}

// The same block for the training step: NDHWC tensors, every intermediate the reverse pass needs is kept (VrnRowArgs).
int launch_vrn16_row_train(const float* x, float* t11, float* t21, float* t22, float* pre, float* out, const float* const* w, int B,
                           hipStream_t s, int* pre_signs, bool q4) {
  VrnRowArgs a = vrn_args(x, t11, out, w, B);
  a.t21 = t21; a.t22 = t22; a.pre = pre; a.pre_signs = pre_signs;
  a.abl = PCGC_ABL_VALUE;
  const dim3 grid(B * (kD / 2) * (kD / 8) / 4);
  if (q4) {                                                 // x / out / pre in the Q4 layout
    hipLaunchKernelGGL((vrn16a_row_kernel<2, 8, true, false, false>), grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL((vrn16bc_row_kernel<2, 8, true, false, false, false>), grid, dim3(256), 0, s, a);
  } else {
    hipLaunchKernelGGL((vrn16a_row_kernel<2, 8, true>), grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL((vrn16bc_row_kernel<2, 8, true>), grid, dim3(256), 0, s, a);
  }
  return launch_ok("vrn16 row kernels (training)");
}

// dx = [x > 0] * (dpre + conv1_1^T(dt11) + conv2_1^T(dt21)) of a C = 16 block at D = 64 (x = nullptr: no mask)
int launch_vrn16_bwd_input(const float* dt11, const float* dt21, const float* dpre, const float* x, const float* w11, const float* w21,
                           float* dx, int B, hipStream_t s, bool q4) {
  VrnBwdInArgs a{dt11, dt21, dpre, x, w11, w21, dx, B};
  const dim3 grid(B * (kD / 2) * (kD / 8) / 4);
  if (q4 && x) hipLaunchKernelGGL((vrn16a_bwd_row_kernel<2, 8, true, false>), grid, dim3(256), 0, s, a);
  else if (q4) hipLaunchKernelGGL((vrn16a_bwd_row_kernel<2, 8, false, false>), grid, dim3(256), 0, s, a);
  else if (x) hipLaunchKernelGGL((vrn16a_bwd_row_kernel<2, 8, true>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((vrn16a_bwd_row_kernel<2, 8, false>), grid, dim3(256), 0, s, a);
  return launch_ok("vrn16a_bwd_row_kernel");
}

// dt11 / dt22 / dt21 of a C = 16 block at D = 64 from the block tail's reverse (vrn16bc_bwd_row_kernel)
int launch_vrn16_bwd_tail(const float* dz12, const float* dz23, const float* t11, const float* t21, const float* t22, const float* w12,
                          const float* w22, const float* w23, float* dt11, float* dt21, float* dt22, int B, hipStream_t s) {
  VrnBwdTailArgs a{dz12, dz23, t11, t21, t22, w12, w22, w23, dt11, dt21, dt22, B};
  hipLaunchKernelGGL((vrn16bc_bwd_row_kernel<2, 8>), dim3(B * (kD / 2) * (kD / 8) / 4), dim3(256), 0, s, a);
  return launch_ok("vrn16bc_bwd_row_kernel");
}
// the same with the block tail's reverse folded in: dout (masked by out > 0 already) + sign bits of pre -> dz12 / dz23 too
int launch_vrn16_bwd_tail_split(const float* dout, const int* signs, const float* t11, const float* t21, const float* t22,
                                const float* w12, const float* w22, const float* w23, float* dz12, float* dz23, float* dt11, float* dt21,
                                float* dt22, int B, hipStream_t s, bool q4) {
  VrnBwdTailArgs a{dz12, dz23, t11, t21, t22, w12, w22, w23, dt11, dt21, dt22, B};
  a.dout = dout; a.signs = signs; a.dz12w = dz12; a.dz23w = dz23;
  if (q4) hipLaunchKernelGGL((vrn16bc_bwd_row_kernel<2, 8, true, false>), dim3(B * (kD / 2) * (kD / 8) / 4), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((vrn16bc_bwd_row_kernel<2, 8, true>), dim3(B * (kD / 2) * (kD / 8) / 4), dim3(256), 0, s, a);
  return launch_ok("vrn16bc_bwd_row_kernel (with the tail's split)");
}

// conv_in (x one channel NDHWC -> y Q4 16 channels) / deconv_out (x Q4 16 channels -> y one channel); D = 64
int launch_conv_in_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s, const RowSkip* skip,
                       const float* mask) {
  ConvRowArgs a{x, y, w, bias, B, relu};
  a.mask = mask;
  if (skip) a.skip = *skip;
  constexpr int TH = 2, LD = 4;
  const int waves = B * (kD / TH) * (kD / LD);
  hipLaunchKernelGGL((conv_in_row_kernel<TH, LD>), dim3(waves / 4), dim3(256), 0, s, a);
  return launch_ok("conv_in_row_kernel");
}
int launch_deconv_out_row(const float* x, float* y, const float* w, const float* bias, int B, int relu, hipStream_t s) {
  ConvRowArgs a{x, y, w, bias, B, relu};
  static const int remap = getenv("PCGC_XCD_REMAP_OUT") ? atoi(getenv("PCGC_XCD_REMAP_OUT")) : 0;     // experiment knob
  a.remap = remap;
  // 4 rows x 8 planes per wave: the tile reads 6 x 10 rows for 4 x 8 (1.88 x; 4 x 4 tiles read 2.25 x and were HBM-bound at
  // 1.61 x the input).  Measured per 8 cubes (profiles/r05_vC_deconv_out_tiles.txt): <4,4> 42.4 us, <4,8> 36.6, <2,16> 38.3,
  // <8,4> 41.8, <8,8> 49.8 — 1 024 waves with twice the loads in flight each beat 2 048
  constexpr int TH = 4, LD = 8;
  const int waves = B * (kD / TH) * (kD / LD);
  hipLaunchKernelGGL((deconv_out_row_kernel<TH, LD>), dim3(waves / 4), dim3(256), 0, s, a);
  return launch_ok("deconv_out_row_kernel");
}

}  // namespace pcgc
