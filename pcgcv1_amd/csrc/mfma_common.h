// Device helpers shared by the MFMA convolution kernels (conv_mfma.hip, vrn_mfma.hip).
#pragma once
#include "common.h"

namespace pcgc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int CIN>
struct Chunk {
  static constexpr int CK = CIN < 16 ? CIN : 16;           // channels per LDS chunk
  static constexpr int NCH = CIN / CK;                     // chunks
  static constexpr int VEC = CK / 4;                       // floats per lane per tap (K-steps)
  static constexpr int VS = CK == 16 ? 20 : (CK == 8 ? 12 : 4);  // LDS voxel stride (floats), 16-B multiple
};

template <int VEC>
__device__ __forceinline__ void read_vec(const float* p, float (&v)[4]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}

// epilogue for one accumulator: the lane holds channels c0..c0+3 of output voxel `vox`
__device__ __forceinline__ void store_acc(const ConvArgs& a, int64_t vox, int c0, f32x4 acc) {
  if (c0 >= a.Cout) return;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  // element offset of (vox, channel y_co + c0): NDHWC, or Q4 [row][C/4][w][4] (row = vox / Dout, c0 % 4 == 0)
  const int64_t eo = a.y_q4 ? (((vox / a.Dout) * (a.y_cs >> 2) + ((a.y_co + c0) >> 2)) * a.Dout + vox % a.Dout) * 4
                            : vox * a.y_cs + a.y_co + c0;
  float* yp = a.y + eo;
  if ((a.Cout & 3) == 0) {
    if (a.bias) {
      const float4 bv = *reinterpret_cast<const float4*>(a.bias + c0);
      v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (a.relu) v[r] = fmaxf(v[r], 0.f);
      if (a.absval) v[r] = fmaxf(fabsf(v[r]), a.lower_bound);
    }
    if (a.res) {
      const float4 rv = *reinterpret_cast<const float4*>(a.res + eo);
      v[0] = fmaxf(rv.x + v[0], 0.f); v[1] = fmaxf(rv.y + v[1], 0.f);
      v[2] = fmaxf(rv.z + v[2], 0.f); v[3] = fmaxf(rv.w + v[3], 0.f);
    }
    if (a.add_to) {
      const float4 av = *reinterpret_cast<const float4*>(a.add_to + eo);
      v[0] += av.x; v[1] += av.y; v[2] += av.z; v[3] += av.w;
    }
    if (a.mask) {
      const float4 mv = *reinterpret_cast<const float4*>(a.mask + eo);
      v[0] = mv.x > 0.f ? v[0] : 0.f; v[1] = mv.y > 0.f ? v[1] : 0.f;
      v[2] = mv.z > 0.f ? v[2] : 0.f; v[3] = mv.w > 0.f ? v[3] : 0.f;
    }
    *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    // Cout not a multiple of 4 (deconv_out, 16->1): rows past Cout are zero padding
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (c0 + r < a.Cout) {
        float t = v[r];
        if (a.bias) t += a.bias[c0 + r];
        if (a.relu) t = fmaxf(t, 0.f);
        if (a.absval) t = fmaxf(fabsf(t), a.lower_bound);
        if (a.res) t = fmaxf(a.res[vox * a.y_cs + a.y_co + c0 + r] + t, 0.f);
        if (a.add_to) t += a.add_to[eo + r];
        if (a.mask) t = a.mask[eo + r] > 0.f ? t : 0.f;
        yp[r] = t;
      }
    }
  }
}

}  // namespace pcgc

namespace pcgc {

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with a private L2.
// Remap so that every XCD walks a CONTIGUOUS range of tiles: neighbouring tiles (which share halo voxels)
// then hit the same L2.  Speed only — any placement gives the same results.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  return (nblk & 7) ? bid : (bid & 7) * (nblk >> 3) + (bid >> 3);
}

// Stage an ID x IH x IW voxel tile (Q float4 of channels per voxel, LDS voxel stride VS floats) whose origin is
// (id0, ih0, iw0) in a [Din^3, x_cs] cube starting at `xb`; voxels outside the cube are zero ('same' padding).
// Work split: wave w takes tile rows w, w+4, ... — the row's (d, h) decomposition, bounds test and base offset are
// wave-uniform (scalar ALU); a lane owns fixed columns of the row, so its per-column offsets and W bounds test are
// computed once.  Loads of a batch of rows are all issued before the first LDS store (latencies overlap).
// q4: the source is a Q4 tensor [d][h][x_cs/4][w][4] and xb points at the first channel quad to read.
template <int ID, int IH, int IW, int Q, int VS>
__device__ __forceinline__ void stage_tile(float* lds, const float* xb, int Din, int x_cs, int id0, int ih0, int iw0,
                                           bool q4 = false) {
  static_assert((Q & (Q - 1)) == 0, "float4-per-voxel count must be a power of two");
  constexpr int E = IW * Q;                          // float4 per tile row
  constexpr int KP = (E + 63) / 64;                  // column passes per row
  constexpr int NROW = ID * IH;
  constexpr int RPW = (NROW + 3) / 4;                // rows per wave
  constexpr int RB = (12 / KP) > 0 ? (12 / KP) : 1;  // rows per batch (<= 12 float4 in flight per lane)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int gofs[KP], lofs[KP];
  bool ok[KP];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int c = lane + 64 * k;
    const int vox = c / Q, q = c & (Q - 1);
    gofs[k] = q4 ? vox * 4 + q * Din * 4 : vox * x_cs + q * 4;
    lofs[k] = vox * VS + q * 4;
    ok[k] = (c < E) && ((unsigned)(iw0 + vox) < (unsigned)Din);
  }
#pragma unroll
  for (int r0 = 0; r0 < RPW; r0 += RB) {
    float4 vals[RB][KP];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int r = wv + 4 * (r0 + i);               // wave-uniform
      const int zd = r / IH, zh = r - zd * IH;
      const int gd = id0 + zd, gh = ih0 + zh;
      const bool row_ok = (r0 + i < RPW) && (r < NROW) && ((unsigned)gd < (unsigned)Din) && ((unsigned)gh < (unsigned)Din);
      const int row_off = q4 ? ((gd * Din + gh) * (x_cs >> 2) * Din + iw0) * 4 : ((gd * Din + gh) * Din + iw0) * x_cs;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        vals[i][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row_ok && ok[k]) vals[i][k] = *reinterpret_cast<const float4*>(xb + (row_off + gofs[k]));
      }
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int r = wv + 4 * (r0 + i);
      if ((r0 + i < RPW) && (r < NROW)) {
#pragma unroll
        for (int k = 0; k < KP; ++k)
          if (lane + 64 * k < E) *reinterpret_cast<float4*>(&lds[r * (IW * VS) + lofs[k]]) = vals[i][k];
      }
    }
  }
}

}  // namespace pcgc
