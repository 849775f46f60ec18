// Stand-alone probe (GPU box): kernel A of the 64^3 blocks (conv1_1 3^3 16 -> 4 + conv2_1 1^3 16 -> 4, ReLU) on COMPACT wave tiles.
// The product's row kernels put a whole 64-voxel row on the 64 lanes, so a wave tile of the analysis' empty-space skipping is 8 planes x
// 2 rows x 64 voxels and 0.36-0.47 of the tiles are heavy; tiles of 8 planes x 4 rows x 16 voxels would be heavy in 0.18-0.32 of the cases
// (tools/exp/count_tiles.py).  Such a tile is the "quad vector" of the 16^3 kernels — lane = (row & 3, 16 voxels) — but inside a 64-wide
// row: the kw = 0 / 2 taps of a segment's first / last voxel need the NEIGHBOURING segment's edge voxel.  This probe runs that kernel
// densely: per (plane, channel quad) three quad vectors (rows h0-1.., h0.., h0+1..) plus three "edge vectors" whose lanes (r, 0) /
// (r, 15) hold the voxels left / right of the segment; x[w-1] is ONE v_mov_dpp row_shr:1 with bound_ctrl off and the edge vector as
// the old value (the lane without a source keeps it), x[w+1] likewise.  Question: what does the dense kernel cost per voxel against
// vrn16a_row_kernel (72 us per 8 cubes)?  Checked against a direct convolution on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_seg_a.hip -o tools/exp/_build/exp_seg_a && tools/exp/_build/exp_seg_a
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 raw_load4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void raw_store4(f32x4 v, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
constexpr int kOOB = 0x7ffff000;
constexpr int kD = 64;

__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <int ABID>
__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0); }
__device__ __forceinline__ f32x4 mfa(int abid, float a, float b, f32x4 c) {
  switch (abid) {
    case 0: return mf<0>(a, b, c); case 1: return mf<1>(a, b, c); case 2: return mf<2>(a, b, c); case 3: return mf<3>(a, b, c);
    case 4: return mf<4>(a, b, c); case 5: return mf<5>(a, b, c); case 6: return mf<6>(a, b, c); case 7: return mf<7>(a, b, c);
    case 8: return mf<8>(a, b, c); case 9: return mf<9>(a, b, c); case 10: return mf<10>(a, b, c); case 11: return mf<11>(a, b, c);
    case 12: return mf<12>(a, b, c); case 13: return mf<13>(a, b, c); case 14: return mf<14>(a, b, c); default: return mf<15>(a, b, c);
  }
}
// lane i <- lane i-1 inside each 16-lane row; the row's first lane keeps `edge` (the neighbouring segment's voxel)
__device__ __forceinline__ float shr_edge(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float shl_edge(float v, float edge) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}
// the same IN PLACE on the edge register (it dies here): one instruction, no copy of `old` (variant 2: separate left / right edge vectors)
__device__ __forceinline__ float shr_into(float edge, float v) {
  asm("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(edge) : "v"(v));
  return edge;
}
__device__ __forceinline__ float shl_into(float edge, float v) {
  asm("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(edge) : "v"(v));
  return edge;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)}; }

struct Args {
  const float* x;      // Q4 [B][64][64][4 quads][64][4]
  float* t12;          // Q4 [B][64][64][2 quads][64][4]: quad 0 = relu(conv1_1), quad 1 = relu(conv2_1)
  const float* w11;    // [27][16 ci][4 co] = 64 floats per tap: lane l of the A operand holds (ci = l / 4, co = l % 4)
  const float* b11;
  const float* w21;    // [16][4]
  const float* b21;
  int B;
};

// wave tile: LD planes x 4 rows x 16 voxels.  waves per cube = (64 / LD) * 16 * 4
template <int LD, int VAR = 1>
__global__ void __launch_bounds__(256, 2) seg_a_kernel(Args a) {
  const int lane = threadIdx.x & 63, r = lane >> 4, w = lane & 15;
  int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int s = wv & 3; wv >>= 2;                            // segment: the four waves of a workgroup share their rows
  const int hq = wv & 15; wv >>= 4;
  const int d0 = (wv % (kD / LD)) * LD; wv /= (kD / LD);
  const int b = wv;
  if (b >= a.B) return;
  const int h0 = 4 * hq, w0 = 16 * s;
  float W[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) W[t] = a.w11[t * 64 + lane];
  const float W2 = a.w21[lane];
  const f32x4 bi = {a.b11[0], a.b11[1], a.b11[2], a.b11[3]};
  const f32x4 bi2 = {a.b21[0], a.b21[1], a.b21[2], a.b21[3]};
  // VAR 3: the row kernel's summation — one partial sum per kw column (bias in the kw = 1 column), each over (plane, channel, kh)
  // in program order, combined as (S_1 + S_0) + S_2 — on shifted INPUTS instead of shifted sums: the same bits as vrn16a_row_kernel
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[3], acc2, S[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    acc[j] = bi;
    S[j][0] = zero; S[j][1] = bi; S[j][2] = zero;
  }
  const i32x4 rs = make_rsrc(a.x + (size_t)b * kD * kD * kD * 16, kD * kD * kD * 16 * 4);
  const i32x4 ro = make_rsrc(a.t12 + (size_t)b * kD * kD * kD * 8, kD * kD * kD * 8 * 4);
  // byte offset of (plane p, row h, quad q, voxel v): ((p * 64 + h) * 4 + q) * 1024 + v * 16
  auto off = [&](int p, int h, int q, int v) { return ((p * kD + h) * 4 + q) * 1024 + v * 16; };
  // quad vector starting at row hs, plane p, quad q; rows / planes outside the cube read zeros
  auto load_vec = [&](int p, int hs, int q) {
    const int h = hs + r;
    const bool ok = (unsigned)p < (unsigned)kD && (unsigned)h < (unsigned)kD;
    return raw_load4(rs, ok ? off(p, h, q, w0 + w) : kOOB, 0, 0);
  };
  // edge vector: lane (r, 0) <- voxel w0 - 1, lane (r, 15) <- voxel w0 + 16 (zeros outside the cube), other lanes unused
  auto load_edge = [&](int p, int hs, int q, int side) {       // side 0: both edges in one vector; 1: left only; 2: right only
    const int h = hs + r;
    const int v = (side == 1 || (side == 0 && w == 0)) ? w0 - 1 : w0 + 16;
    const bool mine = side == 0 ? (w == 0 || w == 15) : (side == 1 ? w == 0 : w == 15);
    const bool ok = mine && (unsigned)p < (unsigned)kD && (unsigned)h < (unsigned)kD && (unsigned)v < (unsigned)kD;
    return raw_load4(rs, ok ? off(p, h, q, v) : kOOB, 0, 0);
  };
  f32x4 XA[3], EA[6], XB[3], EB[6];
  auto load = [&](f32x4 (&X)[3], f32x4 (&E)[6], int p, int q) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      X[kh] = load_vec(p, h0 + kh - 1, q);
      if constexpr (VAR >= 2) { E[kh] = load_edge(p, h0 + kh - 1, q, 1); E[3 + kh] = load_edge(p, h0 + kh - 1, q, 2); }
      else E[kh] = load_edge(p, h0 + kh - 1, q, 0);
    }
  };
  auto quad = [&](const f32x4 (&X)[3], const f32x4 (&E)[6], int q, bool v0, bool v1, bool v2) {
    const bool vj[3] = {v0, v1, v2};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x0[3], xm[3], xp[3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        x0[kh] = X[kh][c];
        if constexpr (VAR >= 2) { xm[kh] = shr_into(E[kh][c], x0[kh]); xp[kh] = shl_into(E[3 + kh][c], x0[kh]); }
        else { xm[kh] = shr_edge(x0[kh], E[kh][c]); xp[kh] = shl_edge(x0[kh], E[kh][c]); }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int kd = 2 - j;
        if (vj[j]) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const float xv = kw == 0 ? xm[kh] : (kw == 1 ? x0[kh] : xp[kh]);
              if constexpr (VAR == 3) S[j][kw] = mfa(4 * q + c, W[(kd * 3 + kh) * 3 + kw], xv, S[j][kw]);
              else acc[j] = mfa(4 * q + c, W[(kd * 3 + kh) * 3 + kw], xv, acc[j]);
            }
        }
      }
      if (v1) acc2 = mfa(4 * q + c, W2, x0[1], acc2);
    }
  };
  load(XA, EA, d0 - 1, 0);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kD;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
    acc2 = bi2;
    load(XB, EB, p, 1);
    quad(XA, EA, 0, v0, v1, v2);
    load(XA, EA, p, 2);
    quad(XB, EB, 1, v0, v1, v2);
    load(XB, EB, p, 3);
    quad(XA, EA, 2, v0, v1, v2);
    load(XA, EA, p + 1, 0);
    quad(XB, EB, 3, v0, v1, v2);
    const int ob = ((p * kD + h0 + r) * 2) * 1024 + (w0 + w) * 16;
    if (v1) raw_store4(relu4(acc2), ro, ob + 1024, 0, 0);
    if (p - 1 >= d0) raw_store4(relu4(VAR == 3 ? (S[0][1] + S[0][0]) + S[0][2] : acc[0]), ro, (((p - 1) * kD + h0 + r) * 2) * 1024 + (w0 + w) * 16, 0, 0);
    acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = bi;
#pragma unroll
    for (int k = 0; k < 3; ++k) { S[0][k] = S[1][k]; S[1][k] = S[2][k]; S[2][k] = k == 1 ? bi : zero; }
  }
}

extern "C" int seg_a_launch(const float* x, float* t12, const float* w11, const float* b11, const float* w21, const float* b21, int B, int var,
                            void* stream) {
  Args a{x, t12, w11, b11, w21, b21, B};
  const int waves = B * (kD / 8) * 16 * 4;
  hipStream_t s = (hipStream_t)stream;
  if (var == 3) hipLaunchKernelGGL((seg_a_kernel<8, 3>), dim3(waves / 4), dim3(256), 0, s, a);
  else if (var == 2) hipLaunchKernelGGL((seg_a_kernel<8, 2>), dim3(waves / 4), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((seg_a_kernel<8, 1>), dim3(waves / 4), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}

#ifndef SEG_A_SHARED
static void reference(const std::vector<float>& x, const std::vector<float>& w11, const std::vector<float>& b11, const std::vector<float>& w21,
                      const std::vector<float>& b21, int d, int h, int w, float* o1, float* o2) {   // cube 0, one voxel
  auto X = [&](int p, int y, int v, int ci) -> double {
    if ((unsigned)p >= 64u || (unsigned)y >= 64u || (unsigned)v >= 64u) return 0.0;
    return x[(((size_t)(p * 64 + y) * 4 + ci / 4) * 64 + v) * 4 + ci % 4];
  };
  for (int co = 0; co < 4; ++co) {
    double s = b11[co], s2 = b21[co];
    for (int kd = 0; kd < 3; ++kd)
      for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw)
          for (int ci = 0; ci < 16; ++ci) s += X(d + kd - 1, h + kh - 1, w + kw - 1, ci) * w11[(((kd * 3 + kh) * 3 + kw) * 16 + ci) * 4 + co];
    for (int ci = 0; ci < 16; ++ci) s2 += X(d, h, w, ci) * w21[ci * 4 + co];
    o1[co] = (float)std::fmax(s, 0.0);
    o2[co] = (float)std::fmax(s2, 0.0);
  }
}

template <int LD, int VAR = 1>
static float run(const Args& a, int reps) {
  const int waves = a.B * (kD / LD) * 16 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((seg_a_kernel<LD, VAR>), dim3(waves / 4), dim3(256), 0, 0, a);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((seg_a_kernel<LD, VAR>), dim3(waves / 4), dim3(256), 0, 0, a);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / reps;
}

int main() {
  const int Bmax = 16;
  const size_t nx = (size_t)Bmax * 64 * 64 * 64 * 16, nt = (size_t)Bmax * 64 * 64 * 64 * 8;
  std::vector<float> x(nx), w11(27 * 64), b11(4), w21(64), b21(4);
  srand(7);
  auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
  for (auto& v : x) v = std::fmax(rnd() * 2.f, 0.f);
  for (auto& v : w11) v = rnd() * 0.2f;
  for (auto& v : w21) v = rnd() * 0.5f;
  for (auto& v : b11) v = rnd() * 0.1f;
  for (auto& v : b21) v = rnd() * 0.1f;
  float *dx, *dt, *dw11, *db11, *dw21, *db21;
  hipMalloc(&dx, nx * 4); hipMalloc(&dt, nt * 4); hipMalloc(&dw11, w11.size() * 4); hipMalloc(&db11, 16); hipMalloc(&dw21, 256); hipMalloc(&db21, 16);
  hipMemcpy(dx, x.data(), nx * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw11, w11.data(), w11.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db11, b11.data(), 16, hipMemcpyHostToDevice);
  hipMemcpy(dw21, w21.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db21, b21.data(), 16, hipMemcpyHostToDevice);
  Args a{dx, dt, dw11, db11, dw21, db21, 1};
  hipMemset(dt, 0xff, nt * 4);
  std::vector<float> t((size_t)64 * 64 * 64 * 8);
  double worst = 0;
  for (int var = 1; var <= 3; ++var) {
  seg_a_launch(a.x, a.t12, a.w11, a.b11, a.w21, a.b21, 1, var, nullptr);
  hipMemcpy(t.data(), dt, t.size() * 4, hipMemcpyDeviceToHost);
  static const int probe[][3] = {{0, 0, 0}, {0, 0, 15}, {0, 0, 16}, {63, 63, 63}, {5, 17, 31}, {5, 17, 32}, {8, 3, 47}, {8, 4, 48}, {7, 60, 0}, {31, 32, 16}, {40, 1, 63}, {16, 16, 15}};
  for (auto& pr : probe) {
    float o1[4], o2[4];
    reference(x, w11, b11, w21, b21, pr[0], pr[1], pr[2], o1, o2);
    for (int co = 0; co < 4; ++co) {
      const float g1 = t[(((size_t)(pr[0] * 64 + pr[1]) * 2 + 0) * 64 + pr[2]) * 4 + co], g2 = t[(((size_t)(pr[0] * 64 + pr[1]) * 2 + 1) * 64 + pr[2]) * 4 + co];
      worst = std::fmax(worst, std::fabs(g1 - o1[co]));
      worst = std::fmax(worst, std::fabs(g2 - o2[co]));
    }
  }
  printf("variant %d: max |kernel - direct convolution| over %zu probe voxels (segment edges, cube faces): %.3g\n", var, sizeof(probe) / sizeof(probe[0]), worst);
  }
  for (int B : {8, 16}) {
    a.B = B;
    printf("B = %2d cubes, one edge vector + copy:   LD 8 %.1f us, LD 4 %.1f us, LD 16 %.1f us   (vrn16a_row_kernel: 72 us per 8 cubes dense)\n", B, run<8, 1>(a, 20), run<4, 1>(a, 20), run<16, 1>(a, 20));
    printf("B = %2d cubes, two edge vectors, in place: LD 8 %.1f us, LD 4 %.1f us, LD 16 %.1f us\n", B, run<8, 2>(a, 20), run<4, 2>(a, 20), run<16, 2>(a, 20));
    printf("B = %2d cubes, the row kernel's three kw sums: LD 8 %.1f us, LD 4 %.1f us, LD 16 %.1f us\n", B, run<8, 3>(a, 20), run<4, 3>(a, 20), run<16, 3>(a, 20));
  }
  return worst < 1e-4 ? 0 : 1;
}
#endif
