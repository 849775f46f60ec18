// Segment ("gather") form of the 64^3 / C = 16 Voxception-ResNet row kernels (vrn_row.hip) for the analysis' exact
// skipping of empty space at a finer grain (models/model_voxception.py:56-68, 125-144).
//
// The row kernels put a whole 64-voxel row on the 64 lanes, so the unit that is computed or skipped is 8 planes x 2 rows x
// 64 voxels, and a surface that crosses a row anywhere makes it heavy: 0.36-0.47 of the tiles of the analysis' 64^3 stage.
// Here the unit is a SLOT of 8 planes x 2 rows x 16 voxels (0.17-0.31 heavy, tools/exp/count_tiles.py) and a wave computes
// FOUR slots at once, one per group of 16 lanes, taken from a list of the launch's heavy slots: lanes 16g .. 16g+15 hold the
// 16 voxels of slot g.  Everything a lane touches is addressed through a per-lane byte offset (the slot's cube, plane, row and
// segment) plus the scalar walk over planes / rows / channel quads that all four slots share, so the instruction stream is
// the row kernels': the same MFMAs in the same order per output, on the same weights.
//   * kw = 0 / 2 taps: the row kernels shift lanes (DPP) with zero fill at the ends of the row; a shift ends at the slot
//     here (row_shr:1 / row_shl:1 inside a group of 16 lanes) and the lane without a source keeps an EDGE value instead: the
//     neighbouring segment's voxel, fetched with the row by one more load in which only lanes 0 / 15 of a group have an
//     offset inside the window (left / right neighbour; a voxel outside the cube: none = 0, the 'same' padding).
//   * kernel A keeps the row kernel's three partial sums per output, one per kw column, each over (plane, channel, kh) in
//     program order and combined as (S_1 + S_0) + S_2 — formed on shifted INPUTS instead of shifted sums: the same bits
//     (tools/exp/t_seg_a_bits.py established it for the quad-vector probe; tests/test_gpu_parity.py for these kernels).
//   * planes / rows outside the cube differ per slot: those lanes read zeros and run the same MFMAs (the row kernels skip
//     them for the whole wave; adding w * 0 to an accumulator that was born from +0 or a bias changes no bit).
//   * tensors and the empty-cube responses they may be replaced by (rows of slots the producer did not write) live in ONE
//     window of < 2 GiB behind one buffer descriptor; which of the two a lane reads is a per-lane select of its offset.
#include "row_common.h"

namespace pcgc {
namespace seg {

constexpr int kD = 64, TH = 2, LD = 8;
constexpr unsigned kBad = 0x7ffff000u;            // a vector offset past the window: loads return 0, stores are dropped
constexpr int kRow4 = 4 * 1024, kRow2 = 2 * 1024; // bytes of one row of a 16- / 8-channel Q4 tensor
constexpr int kCube4 = kD * kD * kRow4, kCube2 = kD * kD * kRow2;

__device__ __forceinline__ i32x4 window(const void* p, int byte_off = 0) {
  const unsigned long long a = (unsigned long long)p + (unsigned)byte_off;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = (int)kBad;
  r[3] = 0x00020000;
  return r;
}

// One wave's four slots, per lane.
struct Slot {
  int b, d0, h0, s, w;       // cube, first plane, first row, segment, lane within the segment
  bool live;                 // the lane's group has a slot (the last wave of a launch may not be full)
};
__device__ __forceinline__ bool wave_slots(const unsigned* slots, const unsigned* n_slots, Slot* sl) {
  const int lane = threadIdx.x & 63;
  const unsigned wid = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const unsigned n = *n_slots;
  if (wid * 4 >= n) return false;
  const unsigned idx = wid * 4 + (lane >> 4);
  sl->live = idx < n;
  const unsigned code = sl->live ? slots[idx] : 0u;
  sl->s = code & 3;
  sl->h0 = 2 * ((code >> 2) & 31);
  sl->d0 = 8 * ((code >> 7) & 7);
  sl->b = code >> 10;
  sl->w = lane & 15;
  return true;
}

// Which of the 3 x 3 (plane tile, row tile) neighbourhoods of the slot lie, at the lane's own / left / right segment, in slots
// the producer did NOT write: bit pt * 3 + rt of m0 (voxel w), mm (voxel w - 1), mp (voxel w + 1).  table[b * 256 + dt * 32 + ht]
// bit s = slot (b, dt, ht, s) was not written.
struct Virt { unsigned m0, mm, mp; };
__device__ __forceinline__ Virt virtual_bits(const unsigned char* table, const Slot& sl) {
  Virt v{0u, 0u, 0u};
  if (!table) return v;
  const int dt = sl.d0 >> 3, ht = sl.h0 >> 1;
#pragma unroll
  for (int pt = 0; pt < 3; ++pt)
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
      int dx = dt - 1 + pt, hx = ht - 1 + rt;                   // a clamped entry belongs to rows / planes outside the cube: never used
      dx = dx < 0 ? 0 : (dx > 7 ? 7 : dx);
      hx = hx < 0 ? 0 : (hx > 31 ? 31 : hx);
      const unsigned nib = table[(size_t)sl.b * 256 + dx * 32 + hx];
      const unsigned own = (nib >> sl.s) & 1u;
      const unsigned left = sl.s > 0 ? (nib >> (sl.s - 1)) & 1u : 0u;
      const unsigned right = sl.s < 3 ? (nib >> (sl.s + 1)) & 1u : 0u;
      v.m0 |= own << (pt * 3 + rt);
      v.mm |= (sl.w == 0 ? left : own) << (pt * 3 + rt);
      v.mp |= (sl.w == 15 ? right : own) << (pt * 3 + rt);
    }
  return v;
}

// Vector offsets of the TH + 2 input rows of one plane step: o0 = the lane's own voxel, oe = the edge load (lane 0 of a group:
// the voxel left of the segment, lane 15: the voxel right of it, other lanes: none).  step = 0 .. LD + 1 (plane d0 - 1 + step);
// baseT / baseE = the lane's offset of (plane d0 - 1, row h0 - 1, voxel w) in the tensor / in the producer's empty-cube response.
struct RowOffs { unsigned o0[TH + 2], oe[TH + 2]; };
__device__ __forceinline__ void row_offsets(RowOffs& o, const Slot& sl, const Virt& v, unsigned baseT, unsigned baseE, int step) {
  const int pt = step == 0 ? 0 : (step == LD + 1 ? 2 : 1);
  const bool pv = sl.live && (step == 0 ? sl.d0 > 0 : (step == LD + 1 ? sl.d0 + LD < kD : true));
  const bool has_m = sl.w == 0 && sl.s > 0, has_p = sl.w == 15 && sl.s < 3;
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) {
    const int rt = r == 0 ? 0 : (r == TH + 1 ? 2 : 1);
    const bool rv = pv && (r == 0 ? sl.h0 > 0 : (r == TH + 1 ? sl.h0 + TH < kD : true));
    const unsigned bit = 1u << (pt * 3 + rt);
    o.o0[r] = rv ? ((v.m0 & bit) ? baseE : baseT) : kBad;
    const unsigned em = ((v.mm & bit) ? baseE : baseT) - 16u, ep = ((v.mp & bit) ? baseE : baseT) + 16u;
    o.oe[r] = rv && (has_m || has_p) ? (has_m ? em : ep) : kBad;
  }
}

// the TH + 2 rows of one (plane step, quad) and their edge values: soff = scalar byte offset of (step, row 0, quad)
template <int ROW>
__device__ __forceinline__ void load2(f32x4 (&b0)[TH + 2], f32x4 (&be)[TH + 2], i32x4 rs, const RowOffs& o, int soff) {
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) {
    b0[r] = raw_load4(rs, (int)o.o0[r], soff + r * ROW, 0);
    be[r] = raw_load4(rs, (int)o.oe[r], soff + r * ROW, 0);
  }
}
// lane i <- lane i - 1 / i + 1 inside each group of 16; the group's first / last lane keeps `edge`
__device__ __forceinline__ float shr_edge(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float shl_edge(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}

// ---------------------------------------------------------------------------------------------------------------
// kernel A:  t12 = [ relu(conv1_1(x)) (3^3, 16 -> 4) | relu(conv2_1(x)) (1^3, 16 -> 4) ]     (vrn_row.hip: a_quad)
// ---------------------------------------------------------------------------------------------------------------
// two channels (c0, c0 + 1 of the loaded quad): their shifted rows are made once, then one wave-uniform branch per output plane
template <int P0, int P1, int P2, bool FRESH>
__device__ __forceinline__ void a_pair(f32x4 (&S)[3][3][TH], f32x4 (&acc2)[TH], const f32x4& bias, const f32x4& bias2, const float (&W)[27],
                                       float W2, int ci0, int c0, const f32x4 (&b0)[TH + 2], const f32x4 (&be)[TH + 2], bool v0, bool v1,
                                       bool v2) {
  const bool vj[3] = {v0, v1, v2};
  constexpr int P[3] = {P0, P1, P2};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  float x0[2][TH + 2], xm[2][TH + 2], xp[2][TH + 2];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) {
      x0[cc][r] = comp(b0[r], c0 + cc);
      xm[cc][r] = shr_edge(x0[cc][r], comp(be[r], c0 + cc));
      xp[cc][r] = shl_edge(x0[cc][r], comp(be[r], c0 + cc));
    }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int kd = 2 - j;
    if (vj[j]) {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c = c0 + cc;
#pragma unroll
        for (int r = 0; r < TH + 2; ++r)
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int jr = r - kh;
            if (jr >= 0 && jr < TH) {
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) {
                const float xv = kw == 0 ? xm[cc][r] : (kw == 1 ? x0[cc][r] : xp[cc][r]);
                const bool first = FRESH && c == 0 && j == 2 && kh == 0;
                if (first) S[P[j]][kw][jr] = mfa_new(ci0 + c, W[(kd * 3 + kh) * 3 + kw], xv, kw == 1 ? bias : zero);
                else S[P[j]][kw][jr] = mfa(ci0 + c, W[(kd * 3 + kh) * 3 + kw], xv, S[P[j]][kw][jr]);
              }
            }
          }
        if (j == 1) {
#pragma unroll
          for (int jr = 0; jr < TH; ++jr)
            acc2[jr] = (FRESH && c == 0) ? mfa_new(ci0 + c, W2, x0[cc][jr + 1], bias2) : mfa(ci0 + c, W2, x0[cc][jr + 1], acc2[jr]);
        }
      }
    }
  }
}
template <int P0, int P1, int P2, bool FRESH>
__device__ __forceinline__ void a_quad(f32x4 (&S)[3][3][TH], f32x4 (&acc2)[TH], const f32x4& bias, const f32x4& bias2, const float (&W)[27],
                                       float W2, int ci0, const f32x4 (&b0)[TH + 2], const f32x4 (&be)[TH + 2], bool v0, bool v1, bool v2) {
  a_pair<P0, P1, P2, FRESH>(S, acc2, bias, bias2, W, W2, ci0, 0, b0, be, v0, v1, v2);
  a_pair<P0, P1, P2, false>(S, acc2, bias, bias2, W, W2, ci0, 2, b0, be, v0, v1, v2);
}

__global__ void __launch_bounds__(256, 2) vrn16a_seg_kernel(SegArgs a) {
  Slot sl;
  if (!wave_slots(a.slots, a.n_slots, &sl)) return;
  const int lane = threadIdx.x & 63;
  float W[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) W[t] = a.w11[t * 64 + lane];
  const float W2 = a.w21[lane];
  f32x4 bi = {a.b11[0], a.b11[1], a.b11[2], a.b11[3]};
  f32x4 bi2 = {a.b21[0], a.b21[1], a.b21[2], a.b21[3]};
  asm volatile("" : "+v"(bi), "+v"(bi2));
  f32x4 S[3][3][TH], acc2[TH];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int r = 0; r < TH; ++r) S[j][k][r] = bi;
#pragma unroll
  for (int r = 0; r < TH; ++r) acc2[r] = bi2;
  const i32x4 rs = window(a.win);
  const unsigned vox = (unsigned)(sl.s * 16 + sl.w) * 16u;
  const unsigned rel = (unsigned)((sl.d0 - 1) * kD + (sl.h0 - 1)) * (unsigned)kRow4 + vox;      // window offsets are > one plane: never negative
  const unsigned baseT = a.x_off + (unsigned)sl.b * (unsigned)kCube4 + rel, baseE = a.ein_off + rel;
  const Virt vb = virtual_bits(a.in_virt, sl);
  // the lane's offset of (plane d0, row h0, voxel w) in the output; dead lanes store nowhere
  const unsigned outb = sl.live ? a.t_off + (unsigned)sl.b * (unsigned)kCube2 + (unsigned)(sl.d0 * kD + sl.h0) * (unsigned)kRow2 + vox : kBad;
  RowOffs ro;
  row_offsets(ro, sl, vb, baseT, baseE, 0);
  // two sets of row buffers (rows + edge values), requested one channel quad ahead of their use
  f32x4 A0[TH + 2], Ae[TH + 2], B0[TH + 2], Be[TH + 2];
  load2<kRow4>(A0, Ae, rs, ro, 0);
  auto step = [&](int i, auto P0_, auto P1_, auto P2_) {
    constexpr int P0 = decltype(P0_)::value, P1 = decltype(P1_)::value, P2 = decltype(P2_)::value;
    // plane d0 - 1 + i feeds output planes d0 + i - 2 (set P0), d0 + i - 1 (P1), d0 + i (P2)
    const bool v0 = i >= 2, v1 = i >= 1 && i <= LD, v2 = i < LD;
    const int soff = i * (kD * kRow4);
    load2<kRow4>(B0, Be, rs, ro, soff + 1 * 1024);
    a_quad<P0, P1, P2, true>(S, acc2, bi, bi2, W, W2, 0, A0, Ae, v0, v1, v2);
    load2<kRow4>(A0, Ae, rs, ro, soff + 2 * 1024);
    a_quad<P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, 4, B0, Be, v0, v1, v2);
    load2<kRow4>(B0, Be, rs, ro, soff + 3 * 1024);
    a_quad<P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, 8, A0, Ae, v0, v1, v2);
    if (i == 0 || i == LD) row_offsets(ro, sl, vb, baseT, baseE, i + 1);     // the next plane lies in another plane tile
    load2<kRow4>(A0, Ae, rs, ro, soff + kD * kRow4);
    a_quad<P0, P1, P2, false>(S, acc2, bi, bi2, W, W2, 12, B0, Be, v0, v1, v2);
    // stores: the plane in the descriptor base (scalar), row and quad in the instruction's offset field, the slot in the
    // vector offset; no register soffset on a 128-bit store (rsrc_at in vrn_row.hip)
    if (v1) {
      const i32x4 ws = window(a.win, (i - 1) * (kD * kRow2));
#pragma unroll
      for (int r = 0; r < TH; ++r) raw_store4(relu4(acc2[r]), ws, (int)outb + r * kRow2 + 1024, 0, 0);
    }
    if (v0) {
      const i32x4 ws = window(a.win, (i - 2) * (kD * kRow2));
#pragma unroll
      for (int r = 0; r < TH; ++r) raw_store4(relu4((S[P0][1][r] + S[P0][0][r]) + S[P0][2][r]), ws, (int)outb + r * kRow2, 0, 0);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
  for (int i = 0; i <= LD + 1; i += 3) {
    step(i, I0{}, I1{}, I2{});
    if (i + 1 > LD + 1) break;
    step(i + 1, I1{}, I2{}, I0{});
    if (i + 2 > LD + 1) break;
    step(i + 2, I2{}, I0{}, I1{});
  }
}

// ---------------------------------------------------------------------------------------------------------------
// kernel BC:  out = relu( x + [ relu(conv1_2(t11)) | relu(conv2_3(relu(conv2_2(t21)))) ] )      (vrn_row.hip: bc_channel12 / 22)
// ---------------------------------------------------------------------------------------------------------------
template <int P0, int P1, int P2, bool FRESH>
__device__ __forceinline__ void bc_channel12(f32x4 (&acc)[3][TH][2], const f32x4 (&bias)[2], const float (&W)[14], int ci,
                                             const f32x4 (&b0)[TH + 2], const f32x4 (&be)[TH + 2], bool v0, bool v1, bool v2) {
  float x0[TH + 2], xm[TH + 2], xp[TH + 2];
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) { x0[r] = comp(b0[r], ci); xm[r] = shr_edge(x0[r], comp(be[r], ci)); xp[r] = shl_edge(x0[r], comp(be[r], ci)); }
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

template <int P0, int P1, int P2, bool FRESH>
__device__ __forceinline__ void bc_channel22(f32x4 (&acc)[3][TH], const f32x4& bias, const float (&W)[7], int ci, const f32x4 (&b0)[TH + 2],
                                             const f32x4 (&be)[TH + 2], bool v0, bool v1, bool v2) {
  float x0[TH + 2], xm[TH + 2], xp[TH + 2];
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) { x0[r] = comp(b0[r], ci); xm[r] = shr_edge(x0[r], comp(be[r], ci)); xp[r] = shl_edge(x0[r], comp(be[r], ci)); }
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

template <bool NONNEG>
__global__ void __launch_bounds__(256, 2) vrn16bc_seg_kernel(SegArgs a) {
  Slot sl;
  if (!wave_slots(a.slots, a.n_slots, &sl)) return;
  const int lane = threadIdx.x & 63;
  float W12[14], W22[7];
#pragma unroll
  for (int v = 0; v < 14; ++v) W12[v] = (v * 64 + lane < 27 * 32) ? a.w12[v * 64 + lane] : 0.f;
#pragma unroll
  for (int v = 0; v < 7; ++v) W22[v] = (v * 64 + lane < 27 * 16) ? a.w22[v * 64 + lane] : 0.f;
  const float W23 = lane < 32 ? a.w23[lane] : 0.f;
  f32x4 bi12[2] = {{a.b12[0], a.b12[1], a.b12[2], a.b12[3]}, {a.b12[4], a.b12[5], a.b12[6], a.b12[7]}};
  f32x4 bi22 = {a.b22[0], a.b22[1], a.b22[2], a.b22[3]};
  f32x4 bi23[2] = {{a.b23[0], a.b23[1], a.b23[2], a.b23[3]}, {a.b23[4], a.b23[5], a.b23[6], a.b23[7]}};
  asm volatile("" : "+v"(bi12[0]), "+v"(bi12[1]), "+v"(bi22), "+v"(bi23[0]), "+v"(bi23[1]));
  f32x4 acc12[3][TH][2], acc22[3][TH];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < TH; ++r) { acc12[j][r][0] = bi12[0]; acc12[j][r][1] = bi12[1]; acc22[j][r] = bi22; }
  const i32x4 rs = window(a.win);
  const unsigned vox = (unsigned)(sl.s * 16 + sl.w) * 16u;
  const unsigned rel = (unsigned)((sl.d0 - 1) * kD + (sl.h0 - 1)) * (unsigned)kRow2 + vox;
  const unsigned baseT = a.t_off + (unsigned)sl.b * (unsigned)kCube2 + rel, baseE = a.ein_off + rel;
  const Virt vb = virtual_bits(a.in_virt, sl);
  // residual / output: the lane's offset of (plane d0, row h0, voxel w) in the block input (or its producer's empty-cube
  // response when that slot was not written) and in the block output
  const unsigned rel4 = (unsigned)(sl.d0 * kD + sl.h0) * (unsigned)kRow4 + vox;
  bool res_e = false;
  if (a.res_virt) res_e = (a.res_virt[(size_t)sl.b * 256 + (sl.d0 >> 3) * 32 + (sl.h0 >> 1)] >> sl.s) & 1u;
  const unsigned resb = sl.live ? (res_e ? a.eres_off + rel4 : a.x_off + (unsigned)sl.b * (unsigned)kCube4 + rel4) : kBad;
  const unsigned outb = sl.live ? a.out_off + (unsigned)sl.b * (unsigned)kCube4 + rel4 : kBad;
  RowOffs ro;
  row_offsets(ro, sl, vb, baseT, baseE, 0);
  f32x4 A0[TH + 2], Ae[TH + 2], B0[TH + 2], Be[TH + 2];     // tensor1_1 / tensor2_1 rows and their edge values
  load2<kRow2>(A0, Ae, rs, ro, 0);
  load2<kRow2>(B0, Be, rs, ro, 1024);
  auto step = [&](int i, auto P0_, auto P1_, auto P2_) {
    constexpr int P0 = decltype(P0_)::value, P1 = decltype(P1_)::value, P2 = decltype(P2_)::value;
    const bool v0 = i >= 2, v1 = i >= 1 && i <= LD, v2 = i < LD;
    bc_channel12<P0, P1, P2, true>(acc12, bi12, W12, 0, A0, Ae, v0, v1, v2);
#pragma unroll
    for (int c = 1; c < 4; ++c) bc_channel12<P0, P1, P2, false>(acc12, bi12, W12, c, A0, Ae, v0, v1, v2);
    if (i == 0 || i == LD) row_offsets(ro, sl, vb, baseT, baseE, i + 1);
    const int snext = (i + 1) * (kD * kRow2);
    load2<kRow2>(A0, Ae, rs, ro, snext);
    // residual rows of output plane d0 + i - 2 (none before the first finished plane: the loads read zeros, the stores drop)
    const bool done = v0;
    const int obase = done ? (i - 2) * (kD * kRow4) : 0;
    f32x4 res[TH][4];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) res[r][q] = raw_load4(rs, (int)(done ? resb : kBad), obase + r * kRow4 + q * 1024, 0);
    bc_channel22<P0, P1, P2, true>(acc22, bi22, W22, 0, B0, Be, v0, v1, v2);
#pragma unroll
    for (int c = 1; c < 4; ++c) bc_channel22<P0, P1, P2, false>(acc22, bi22, W22, c, B0, Be, v0, v1, v2);
    load2<kRow2>(B0, Be, rs, ro, snext + 1024);
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
      const i32x4 ws = window(a.win, obase + r * kRow4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 y = NONNEG ? res[r][q] + pr[q] : relu4(res[r][q] + pr[q]);
        raw_store4(y, ws, (int)(done ? outb : kBad) + q * 1024, 0, 0);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
  for (int i = 0; i <= LD + 1; i += 3) {
    step(i, I0{}, I1{}, I2{});
    if (i + 1 > LD + 1) break;
    step(i + 1, I1{}, I2{}, I0{});
    if (i + 2 > LD + 1) break;
    step(i + 2, I2{}, I0{}, I1{});
  }
}

// ---------------------------------------------------------------------------------------------------------------
// conv_in on slots:  y = relu(conv 3^3, 1 -> 16 + bias) of the occupancy cube x [B][64][64][64] (vrn_row.hip: conv_in_row_kernel;
// models/model_voxception.py:83-88) — the same MFMAs per output in the same order, the row and its edge values as above.
// ---------------------------------------------------------------------------------------------------------------
constexpr unsigned kXPad = 128u << 10;            // the input's descriptor starts this far below its first cube (offsets stay positive)
__global__ void __launch_bounds__(256, 2) conv_in_seg_kernel(ConvInSegArgs a) {
  Slot sl;
  if (!wave_slots(a.slots, a.n_slots, &sl)) return;
  const int lane = threadIdx.x & 63;
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
  const i32x4 rx = window(reinterpret_cast<const char*>(a.x) - kXPad);
  const unsigned vx = (unsigned)(sl.s * 16 + sl.w);
  const unsigned xb = kXPad + (unsigned)sl.b * (unsigned)(kD * kD * kD * 4) + (unsigned)((sl.d0 - 1) * kD + (sl.h0 - 1)) * (unsigned)(kD * 4) + vx * 4u;
  const unsigned outb = sl.live ? a.out_off + (unsigned)sl.b * (unsigned)kCube4 + (unsigned)(sl.d0 * kD + sl.h0) * (unsigned)kRow4 + vx * 16u : kBad;
  const bool has_m = sl.w == 0 && sl.s > 0, has_p = sl.w == 15 && sl.s < 3;
  auto load_plane = [&](float (&x0)[TH + 2], float (&xe)[TH + 2], int i) {
    // (the look-ahead asks for step LD + 2 as well: a plane that no slot has — never an address past the last cube)
    const bool pv = sl.live && i <= LD + 1 && (unsigned)(sl.d0 - 1 + i) < (unsigned)kD;
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) {
      const bool rv = pv && (r == 0 ? sl.h0 > 0 : (r == TH + 1 ? sl.h0 + TH < kD : true));
      const int soff = (i * kD + r) * (kD * 4);
      x0[r] = raw_load1(rx, (int)(rv ? xb : kBad), soff, 0);
      xe[r] = raw_load1(rx, (int)(rv && (has_m || has_p) ? (has_m ? xb - 4u : xb + 4u) : kBad), soff, 0);
    }
  };
  float cur[TH + 2], cue[TH + 2], nxt[TH + 2], nxe[TH + 2];
  load_plane(cur, cue, 0);
#pragma unroll 1
  for (int i = 0; i <= LD + 1; ++i) {
    const bool vj[3] = {i >= 2, i >= 1 && i <= LD, i < LD};
    load_plane(nxt, nxe, i + 1);
    float xm[TH + 2], xp[TH + 2];
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) { xm[r] = shr_edge(cur[r], cue[r]); xp[r] = shl_edge(cur[r], cue[r]); }
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
    if (i >= 2) {
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const i32x4 ws = window(a.win, (i - 2) * (kD * kRow4) + r * kRow4);
#pragma unroll
        for (int q = 0; q < 4; ++q) raw_store4(a.relu ? relu4(acc[0][r][q]) : acc[0][r][q], ws, (int)outb + q * 1024, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc[0][r][q] = acc[1][r][q]; acc[1][r][q] = acc[2][r][q]; acc[2][r][q] = bi[q]; }
#pragma unroll
    for (int r = 0; r < TH + 2; ++r) { cur[r] = nxt[r]; cue[r] = nxe[r]; }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Which slots a launch computes.
// ---------------------------------------------------------------------------------------------------------------
// occ[(b * 64 + d) * 64 + h] bit w = voxel (d, h, w) of cube b is not +0.0 (so -0.0 counts as occupied: the empty-cube
// responses were made from +0.0 inputs); rowocc[b * 64 + d] bit h = the row holds such a voxel (what launch_rowocc writes).
// One wave per plane; a load covers four rows (lane = row & 3, four voxels: 1 KiB per wave instruction), the four ballots of a
// load hold voxel 4q + k of row r at bit 16r + q; lane = row then spreads its 16-bit fields to every fourth bit.
__device__ __forceinline__ unsigned long long spread4(unsigned long long x) {
  x &= 0xffffull;
  x = (x | (x << 24)) & 0x000000ff000000ffull;
  x = (x | (x << 12)) & 0x000f000f000f000full;
  x = (x | (x << 6)) & 0x0303030303030303ull;
  x = (x | (x << 3)) & 0x1111111111111111ull;
  return x;
}
__global__ void __launch_bounds__(256) voxocc_kernel(const float* x, unsigned long long* occ, unsigned long long* rowocc, int planes) {
  const int lane = threadIdx.x & 63;
  const int pl = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pl >= planes) return;
  const uint4* px = reinterpret_cast<const uint4*>(x + (size_t)pl * kD * kD) + lane;
  uint4 v[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) v[g] = px[g * 64];
  unsigned long long b0 = 0, b1 = 0, b2 = 0, b3 = 0;           // the ballots of the lane's row group (lane >> 2)
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const unsigned long long c0 = __builtin_amdgcn_ballot_w64(v[g].x != 0u), c1 = __builtin_amdgcn_ballot_w64(v[g].y != 0u),
                             c2 = __builtin_amdgcn_ballot_w64(v[g].z != 0u), c3 = __builtin_amdgcn_ballot_w64(v[g].w != 0u);
    const bool mine = (lane >> 2) == g;
    b0 = mine ? c0 : b0; b1 = mine ? c1 : b1; b2 = mine ? c2 : b2; b3 = mine ? c3 : b3;
  }
  const int sh = 16 * (lane & 3);
  const unsigned long long w = spread4(b0 >> sh) | (spread4(b1 >> sh) << 1) | (spread4(b2 >> sh) << 2) | (spread4(b3 >> sh) << 3);
  occ[(size_t)pl * kD + lane] = w;
  const unsigned long long rb = __builtin_amdgcn_ballot_w64(w != 0ull);
  if (lane == 0) rowocc[pl] = rb;
}

// Slot lists of the stage's launches for every chunk of `chunk` (<= kSegMaxChunk) cubes among `total`: one workgroup per (launch, chunk).
// Launch c = blockIdx.x: 0 = conv_in (radius 1), 1 .. 6 = kernel A / BC of the three blocks (radius 1 + c).  A slot is HEAVY when the window its outputs depend
// on — its planes, rows and voxels dilated by the radius, clipped to the cube — holds an occupied voxel.  Output per (chunk k
// with first cube c0 and n cubes, launch c):
//   slots + (c0 * kSegLaunches + c * n) * 1024     the heavy slots' codes in natural order
//   counts + k * kSegLaunches + c                   their number
//   virt + (c0 * kSegLaunches + c * n) * 256       byte (cube, plane tile, row tile): bit s = the slot is NOT written by launch c
// (the stage's last table is read by down_1: vrn_row32.hip).  counter (tests): += slots the launches do not compute.
__global__ void __launch_bounds__(1024) seg_order_kernel(const unsigned long long* occ, const unsigned long long* rowocc, int total, int chunk,
                                                          unsigned* slots, unsigned* counts, unsigned char* virt, unsigned* counter) {
  __shared__ unsigned long long colmask[8 * 8 * kD];                   // [cube of the half][plane tile][row]: OR over the tile's plane window
  __shared__ unsigned char nibh[kSegMaxChunk * 256];                   // [cube][plane tile][row tile]: heavy segments
  __shared__ unsigned cnt[1024];
  const int cfg = blockIdx.x, R = 1 + cfg;
  const int c0 = blockIdx.y * chunk;
  const int n = total - c0 < chunk ? total - c0 : chunk;
  const int tid = threadIdx.x;
  for (int hb = 0; hb < n; hb += 8) {
    const int nh = n - hb < 8 ? n - hb : 8;
    __syncthreads();                                                   // the previous half's colmask has been read
    for (int i = tid; i < nh * 512; i += 1024) {
      const int b = i >> 9, dt = (i >> 6) & 7, h = i & 63;
      const unsigned long long* o = occ + ((size_t)(c0 + hb + b) * kD) * kD + h;
      unsigned long long m = 0;
#pragma unroll
      for (int k = 0; k < 8 + 2 * 7; ++k) {                            // the widest window (radius 7), narrower ones masked: loads in flight together
        const int p = 8 * dt - R + k;
        const bool in = k < 8 + 2 * R && (unsigned)p < (unsigned)kD;
        const unsigned long long w = o[(size_t)(in ? p : 0) * kD];
        m |= in ? w : 0ull;
      }
      colmask[i] = m;
    }
    __syncthreads();
    for (int i = tid; i < nh * 256; i += 1024) {
      const int b = i >> 8, dt = (i >> 5) & 7, ht = i & 31;
      int h0 = 2 * ht - R, h1 = 2 * ht + 1 + R;
      h0 = h0 < 0 ? 0 : h0; h1 = h1 > kD - 1 ? kD - 1 : h1;
      unsigned long long rw = 0;
      for (int h = h0; h <= h1; ++h) rw |= colmask[(b * 8 + dt) * kD + h];
      unsigned nib = 0;
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        int lo = 16 * sg - R, hi = 16 * sg + 15 + R;
        lo = lo < 0 ? 0 : lo; hi = hi > kD - 1 ? kD - 1 : hi;
        const unsigned long long m = (hi - lo == 63) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << lo);
        nib |= (rw & m) ? (1u << sg) : 0u;
      }
      nibh[(hb + b) * 256 + (i & 255)] = (unsigned char)nib;
    }
  }
  __syncthreads();
  const int items = n * 256, per = (items + 1023) / 1024;              // (plane tile, row tile) pairs per thread, in natural order
  const int t0 = tid * per, t1 = t0 + per < items ? t0 + per : items;
  unsigned nhv = 0;
  for (int t = t0; t < t1; ++t) nhv += __builtin_popcount(nibh[t]);
  cnt[tid] = nhv;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {                           // inclusive scan
    const unsigned a = tid >= off ? cnt[tid - off] : 0u;
    __syncthreads();
    cnt[tid] += a;
    __syncthreads();
  }
  const unsigned total_heavy = cnt[1023];
  unsigned hpos = cnt[tid] - nhv;
  unsigned* o = slots + ((size_t)c0 * kSegLaunches + (size_t)cfg * n) * 1024;
  unsigned char* vt = virt + ((size_t)c0 * kSegLaunches + (size_t)cfg * n) * 256;
  for (int t = t0; t < t1; ++t) {
    const unsigned h = nibh[t];
#pragma unroll
    for (int sg = 0; sg < 4; ++sg)
      if ((h >> sg) & 1u) o[hpos++] = (unsigned)t * 4u + sg;
    vt[t] = (unsigned char)(~h & 0xfu);
  }
  if (tid == 0) {
    counts[(size_t)blockIdx.y * kSegLaunches + cfg] = total_heavy;
    if (counter) atomicAdd(counter, (unsigned)n * 1024u - total_heavy);
  }
}

}  // namespace seg

int launch_voxocc(const float* x, unsigned long long* occ, unsigned long long* rowocc, int B, hipStream_t s) {
  const int planes = B * seg::kD;
  hipLaunchKernelGGL(seg::voxocc_kernel, dim3((planes + 3) / 4), dim3(256), 0, s, x, occ, rowocc, planes);
  return launch_ok("voxocc_kernel");
}
int launch_seg_order(const unsigned long long* occ, const unsigned long long* rowocc, int total, int chunk, unsigned* slots, unsigned* counts,
                     unsigned char* virt, unsigned* counter, hipStream_t s) {
  if (chunk > kSegMaxChunk || chunk < 1) { set_error("launch_seg_order: 1 .. %d cubes per chunk (got %d)", kSegMaxChunk, chunk); return -1; }
  hipLaunchKernelGGL(seg::seg_order_kernel, dim3(kSegLaunches, (total + chunk - 1) / chunk), dim3(1024), 0, s, occ, rowocc, total, chunk, slots,
                     counts, virt, counter);
  return launch_ok("seg_order_kernel");
}

int launch_conv_in_seg(const ConvInSegArgs& a, int max_slots, hipStream_t s) {
  hipLaunchKernelGGL(seg::conv_in_seg_kernel, dim3((max_slots + 15) / 16), dim3(256), 0, s, a);
  return launch_ok("conv_in_seg_kernel");
}

// which: 0 = kernel A, 1 = kernel BC; max_slots sizes the launch (waves past *a.n_slots leave at once)
int launch_vrn16_seg(const SegArgs& a, int which, bool x_nonneg, int max_slots, hipStream_t s) {
  const dim3 grid((max_slots + 15) / 16);
  if (which == 0) hipLaunchKernelGGL(seg::vrn16a_seg_kernel, grid, dim3(256), 0, s, a);
  else if (x_nonneg) hipLaunchKernelGGL(seg::vrn16bc_seg_kernel<true>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(seg::vrn16bc_seg_kernel<false>, grid, dim3(256), 0, s, a);
  return launch_ok("vrn16 segment kernel");
}

}  // namespace pcgc

#ifdef PCGC_SEG_PROBE
// tools/exp/t_seg_gather.py: the kernels on a caller-made window and slot list (stand-alone build of this file)
namespace pcgc {
static thread_local char g_err[256];
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
}
extern "C" int seg_probe_launch(int which, int nonneg, const void* win, unsigned x_off, unsigned t_off, unsigned out_off, unsigned ein_off,
                                unsigned eres_off, const unsigned* slots, const unsigned* n_slots, int max_slots, const unsigned char* in_virt,
                                const unsigned char* res_virt, const float* const* w, void* stream) {
  pcgc::SegArgs a;
  a.win = (const char*)win; a.x_off = x_off; a.t_off = t_off; a.out_off = out_off; a.ein_off = ein_off; a.eres_off = eres_off;
  a.slots = slots; a.n_slots = n_slots; a.in_virt = in_virt; a.res_virt = res_virt;
  a.w11 = w[0]; a.b11 = w[1]; a.w12 = w[2]; a.b12 = w[3]; a.w21 = w[4]; a.b21 = w[5]; a.w22 = w[6]; a.b22 = w[7]; a.w23 = w[8]; a.b23 = w[9];
  return pcgc::launch_vrn16_seg(a, which, nonneg != 0, max_slots, (hipStream_t)stream);
}
#endif
