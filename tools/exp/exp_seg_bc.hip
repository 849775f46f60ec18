// Stand-alone probe (GPU box): kernel BC of the 64^3 blocks on COMPACT wave tiles (LD planes x 4 rows x 16 voxels), the companion of
// exp_seg_a.hip:   out = relu( x + [ relu(conv1_2(t11)) (3^3, 4 -> 8) | relu(conv2_3(relu(conv2_2(t21)))) (3^3 4 -> 4, 1^3 4 -> 8) ] )
// Same mapping (lane = (row & 3, voxel of a 16-voxel segment), three quad vectors per (plane, tensor) plus left / right edge vectors,
// x[w-1] / x[w+1] = one in-place v_mov_dpp row_shr / row_shl whose uncovered lane keeps the neighbouring segment's voxel) and the row
// kernel's summation order per output: bias, then (plane, channel, kh, kw) — so the result is meant to be vrn16bc_row_kernel's, bit
// for bit (tools/exp/t_seg_a_bits.py checks it against pcgc_vrn_fwd).  Question: what does BC cost per voxel on such tiles?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/exp/exp_seg_bc.hip -o tools/exp/_build/libexp_seg_bc.so
#include <hip/hip_runtime.h>

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
  const float* t12;    // Q4 [B][64][64][2 quads][64][4]: quad 0 = tensor1_1, quad 1 = tensor2_1
  const float* x;      // Q4 [B][64][64][4 quads][64][4]: the block input (residual)
  float* out;          // Q4 like x
  const float *w12, *b12, *w22, *b22, *w23, *b23;    // TensorFlow layouts [27][4][8], [8], [27][4][4], [4], [4][8], [8]
  int B;
};

template <int LD>
__global__ void __launch_bounds__(256, 2) seg_bc_kernel(Args a) {
  const int lane = threadIdx.x & 63, r = lane >> 4, w = lane & 15;
  int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int s = wv & 3; wv >>= 2;
  const int hq = wv & 15; wv >>= 4;
  const int d0 = (wv % (kD / LD)) * LD; wv /= (kD / LD);
  const int b = wv;
  if (b >= a.B) return;
  const int h0 = 4 * hq, w0 = 16 * s;
  // weights as in vrn16bc_row_body: conv1_2 register tap >> 1, abid (tap & 1) * 8 + ci * 2 + half; conv2_2 register tap >> 2, abid
  // (tap & 3) * 4 + ci; conv2_3 one register, abid ci * 2 + half
  float W12[14], W22[7];
#pragma unroll
  for (int v = 0; v < 14; ++v) W12[v] = (v * 64 + lane < 27 * 32) ? a.w12[v * 64 + lane] : 0.f;
#pragma unroll
  for (int v = 0; v < 7; ++v) W22[v] = (v * 64 + lane < 27 * 16) ? a.w22[v * 64 + lane] : 0.f;
  const float W23 = lane < 32 ? a.w23[lane] : 0.f;
  const f32x4 bi12[2] = {{a.b12[0], a.b12[1], a.b12[2], a.b12[3]}, {a.b12[4], a.b12[5], a.b12[6], a.b12[7]}};
  const f32x4 bi22 = {a.b22[0], a.b22[1], a.b22[2], a.b22[3]};
  const f32x4 bi23[2] = {{a.b23[0], a.b23[1], a.b23[2], a.b23[3]}, {a.b23[4], a.b23[5], a.b23[6], a.b23[7]}};
  f32x4 acc12[3][2], acc22[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { acc12[j][0] = bi12[0]; acc12[j][1] = bi12[1]; acc22[j] = bi22; }
  const i32x4 rt = make_rsrc(a.t12 + (size_t)b * kD * kD * kD * 8, kD * kD * kD * 8 * 4);
  const i32x4 rx = make_rsrc(a.x + (size_t)b * kD * kD * kD * 16, kD * kD * kD * 16 * 4);
  const i32x4 ro = make_rsrc(a.out + (size_t)b * kD * kD * kD * 16, kD * kD * kD * 16 * 4);
  auto offt = [&](int p, int h, int q, int v) { return ((p * kD + h) * 2 + q) * 1024 + v * 16; };
  // one tensor quad of plane p: three quad vectors (rows h0-1.., h0.., h0+1..) and their left / right edge vectors
  auto load = [&](f32x4 (&X)[3], f32x4 (&EL)[3], f32x4 (&ER)[3], int p, int q) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = h0 + kh - 1 + r;
      const bool rowok = (unsigned)p < (unsigned)kD && (unsigned)h < (unsigned)kD;
      X[kh] = raw_load4(rt, rowok ? offt(p, h, q, w0 + w) : kOOB, 0, 0);
      EL[kh] = raw_load4(rt, (rowok && w == 0 && w0 > 0) ? offt(p, h, q, w0 - 1) : kOOB, 0, 0);
      ER[kh] = raw_load4(rt, (rowok && w == 15 && w0 + 16 < kD) ? offt(p, h, q, w0 + 16) : kOOB, 0, 0);
    }
  };
  f32x4 XA[3], LA[3], RA[3], XB[3], LB[3], RB[3];
  load(XA, LA, RA, d0 - 1, 0);
  load(XB, LB, RB, d0 - 1, 1);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)kD;
    const bool vj[3] = {pin && p - 1 >= d0, pin && p >= d0 && p < d0 + LD, pin && p + 1 < d0 + LD};
    // conv1_2 on tensor1_1: per output (plane, channel, kh, kw), both cout halves
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x0[3], xm[3], xp[3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) { x0[kh] = XA[kh][c]; xm[kh] = shr_into(LA[kh][c], x0[kh]); xp[kh] = shl_into(RA[kh][c], x0[kh]); }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int kd = 2 - j;
        if (vj[j]) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int t = (kd * 3 + kh) * 3 + kw;
              const float xv = kw == 0 ? xm[kh] : (kw == 1 ? x0[kh] : xp[kh]);
#pragma unroll
              for (int hf = 0; hf < 2; ++hf) acc12[j][hf] = mfa((t & 1) * 8 + c * 2 + hf, W12[t >> 1], xv, acc12[j][hf]);
            }
        }
      }
    }
    load(XA, LA, RA, p + 1, 0);
    // the residual rows of the finished plane p - 1, requested before the conv2_2 phase
    const bool done = p - 1 >= d0;
    const int obase = done ? (((p - 1) * kD + h0 + r) * 4) * 1024 + (w0 + w) * 16 : kOOB;
    f32x4 res[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) res[q] = raw_load4(rx, obase == kOOB ? kOOB : obase + q * 1024, 0, 0);
    // conv2_2 on tensor2_1
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x0[3], xm[3], xp[3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) { x0[kh] = XB[kh][c]; xm[kh] = shr_into(LB[kh][c], x0[kh]); xp[kh] = shl_into(RB[kh][c], x0[kh]); }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int kd = 2 - j;
        if (vj[j]) {
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const int t = (kd * 3 + kh) * 3 + kw;
              const float xv = kw == 0 ? xm[kh] : (kw == 1 ? x0[kh] : xp[kh]);
              acc22[j] = mfa((t & 3) * 4 + c, W22[t >> 2], xv, acc22[j]);
            }
        }
      }
    }
    load(XB, LB, RB, p + 1, 1);
    // output plane p - 1: conv2_3 on relu(conv2_2), residual, ReLU, store
    const f32x4 t22 = relu4(acc22[0]);
    f32x4 q3[2] = {bi23[0], bi23[1]};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) q3[hf] = mfa(c * 2 + hf, W23, t22[c], q3[hf]);
    const f32x4 pr[4] = {relu4(acc12[0][0]), relu4(acc12[0][1]), relu4(q3[0]), relu4(q3[1])};
#pragma unroll
    for (int q = 0; q < 4; ++q) raw_store4(relu4(res[q] + pr[q]), ro, obase == kOOB ? kOOB : obase + q * 1024, 0, 0);
    acc12[0][0] = acc12[1][0]; acc12[0][1] = acc12[1][1]; acc12[1][0] = acc12[2][0]; acc12[1][1] = acc12[2][1];
    acc12[2][0] = bi12[0]; acc12[2][1] = bi12[1];
    acc22[0] = acc22[1]; acc22[1] = acc22[2]; acc22[2] = bi22;
  }
}

extern "C" int seg_bc_launch(const float* t12, const float* x, float* out, const float* w12, const float* b12, const float* w22, const float* b22,
                             const float* w23, const float* b23, int B, int ld, void* stream) {
  Args a{t12, x, out, w12, b12, w22, b22, w23, b23, B};
  hipStream_t s = (hipStream_t)stream;
  if (ld == 16) hipLaunchKernelGGL(seg_bc_kernel<16>, dim3(B * (kD / 16) * 16 * 4 / 4), dim3(256), 0, s, a);
  else if (ld == 4) hipLaunchKernelGGL(seg_bc_kernel<4>, dim3(B * (kD / 4) * 16 * 4 / 4), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(seg_bc_kernel<8>, dim3(B * (kD / 8) * 16 * 4 / 4), dim3(256), 0, s, a);
  return (int)hipGetLastError();
}
