// Stand-alone experiment (GPU box): the C=16 Voxception block at 64^3 on v_mfma_f32_4x4x1_16B_f32.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_rowmfma.hip -o tools/exp/_build/exp_rowmfma
//   tools/exp/_build/exp_rowmfma
// A wave owns full W rows (64 voxels = 64 lanes): lane = voxel, VGPR = channel (activations kept in the
// "Q4" layout [b][d][h][C/4][w][4]: one dwordx4 per lane = 4 channels, 1 KiB coalesced per wave).
//   part 1  lane layout of the 16-block MFMA with A broadcast (cbsz=4, abid=k) and of the DPP wave shifts
//   part 2  kernel A = relu(conv1_1(x)) | relu(conv2_1(x)) checked against the CPU and timed at 8 cubes
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <int ABID>
__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
}
__device__ __forceinline__ float shr1(float v) {   // lane i <- lane i-1, lane 0 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float v) {   // lane i <- lane i+1, lane 63 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

__global__ void layout_probe(float* out) {
  const int lane = threadIdx.x;
  // A: lane -> 100*lane ; B: lane -> lane+1 ; abid 5 => A rows come from lanes 20..23
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = mf<5>(100.f * lane, (float)(lane + 1), c);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
  out[256 + lane] = shr1((float)(lane + 1));
  out[320 + lane] = shl1((float)(lane + 1));
}

// ---------------------------------------------------------------------------------------------------
template <int ABID>
__device__ __forceinline__ void tap3(f32x4& acc, float w0, float w1, float w2, float xm, float x0, float xp) {
  acc = mf<ABID>(w0, xm, acc);
  acc = mf<ABID>(w1, x0, acc);
  acc = mf<ABID>(w2, xp, acc);
}

template <int TH, int CI>
__device__ __forceinline__ void channel_step(f32x4 (&acc)[3][TH], f32x4 (&acc2)[TH], const float (&W)[27], float W2,
                                             const float (&xc)[TH + 2], bool v0, bool v1, bool v2) {
  float xm[TH + 2], xp[TH + 2];
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) { xm[r] = shr1(xc[r]); xp[r] = shl1(xc[r]); }
  const bool vj[3] = {v0, v1, v2};
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
            const int t = (kd * 3 + kh) * 3;
            tap3<CI>(acc[j][jr], W[t], W[t + 1], W[t + 2], xm[r], xc[r], xp[r]);
          }
        }
    }
  }
  if (v1) {
#pragma unroll
    for (int jr = 0; jr < TH; ++jr) acc2[jr] = mf<CI>(W2, xc[jr + 1], acc2[jr]);
  }
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}

// x: Q4 [B][64][64][4][64][4]; t12: Q4 [B][64][64][2][64][4]; w11 TF [27][16][4]; w21 TF [16][4]
typedef int i32x4 __attribute__((ext_vector_type(4)));
// hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b128 to a ONE-dword load (splat); bind the intrinsic directly
__device__ f32x4 raw_load4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// rows h0-1 .. h0+TH of plane p, channel quad q: one raw buffer load each; a row outside the cube gets an
// out-of-range offset and reads zeros ('same' padding) — no branches, so the compiler can count vmcnt.
template <int TH>
__device__ __forceinline__ void load_rows(float4 (&buf)[TH + 2], i32x4 rs, int lane16, int p, int q, int h0) {
  constexpr int D = 64;
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) {
    const int h = h0 - 1 + r;
    const bool ok = (unsigned)h < (unsigned)D && (unsigned)p < (unsigned)D;
    const int row = ok ? ((p * D + h) * 4 + q) * 1024 : 0x7ffff000;
    const f32x4 v = raw_load4(rs, row + lane16, 0, 0);
    buf[r] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <int TH, int Q>
__device__ __forceinline__ void quad_step(f32x4 (&acc)[3][TH], f32x4 (&acc2)[TH], const float (&W)[27], float W2,
                                          const float4 (&buf)[TH + 2], bool v0, bool v1, bool v2) {
  float xc[TH + 2];
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) xc[r] = buf[r].x;
  channel_step<TH, 4 * Q + 0>(acc, acc2, W, W2, xc, v0, v1, v2);
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) xc[r] = buf[r].y;
  channel_step<TH, 4 * Q + 1>(acc, acc2, W, W2, xc, v0, v1, v2);
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) xc[r] = buf[r].z;
  channel_step<TH, 4 * Q + 2>(acc, acc2, W, W2, xc, v0, v1, v2);
#pragma unroll
  for (int r = 0; r < TH + 2; ++r) xc[r] = buf[r].w;
  channel_step<TH, 4 * Q + 3>(acc, acc2, W, W2, xc, v0, v1, v2);
}

template <int TH, int LD, int REMAP>
__global__ void __launch_bounds__(256, 2) vrn16a_row(const float* __restrict__ x, float* __restrict__ t12, const float* __restrict__ w11,
                                                     const float* __restrict__ b11, const float* __restrict__ w21,
                                                     const float* __restrict__ b21, int B) {
  constexpr int D = 64;
  if (REMAP >= 2) {   // cap residency at 2 workgroups per CU (LDS is otherwise unused)
    __shared__ float pad[72 * 256];
    if (B < 0) { pad[threadIdx.x] = 1.f; t12[0] = pad[threadIdx.x ^ 1]; }
  }
  const int lane = threadIdx.x & 63;
  int bid = blockIdx.x;
  if ((REMAP & 1) && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  int wv = __builtin_amdgcn_readfirstlane(bid * 4 + (threadIdx.x >> 6));
  const int hg = wv % (D / TH); wv /= (D / TH);
  const int ds = wv % (D / LD); wv /= (D / LD);
  const int b = wv;
  const int h0 = hg * TH, d0 = ds * LD;
  float W[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) W[t] = w11[t * 64 + lane];
  const float W2 = w21[lane];
  const f32x4 bi = {b11[0], b11[1], b11[2], b11[3]};
  const f32x4 bi2 = {b21[0], b21[1], b21[2], b21[3]};
  f32x4 acc[3][TH], acc2[TH];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < TH; ++r) acc[j][r] = bi;
  const i32x4 rs = make_rsrc(x + (size_t)b * D * D * D * 16, D * D * D * 16 * 4);
  const int lane16 = lane * 16;
  float4* tb = reinterpret_cast<float4*>(t12) + (size_t)b * D * D * 2 * 64 + lane;
  float4 bufA[TH + 2], bufB[TH + 2];
  load_rows<TH>(bufA, rs, lane16, d0 - 1, 0, h0);
#pragma unroll 1
  for (int p = d0 - 1; p <= d0 + LD; ++p) {
    const bool pin = (unsigned)p < (unsigned)D;
    const bool v0 = pin && p - 1 >= d0, v1 = pin && p >= d0 && p < d0 + LD, v2 = pin && p + 1 < d0 + LD;
#pragma unroll
    for (int r = 0; r < TH; ++r) acc2[r] = bi2;
    load_rows<TH>(bufB, rs, lane16, p, 1, h0);
    quad_step<TH, 0>(acc, acc2, W, W2, bufA, v0, v1, v2);
    load_rows<TH>(bufA, rs, lane16, p, 2, h0);
    quad_step<TH, 1>(acc, acc2, W, W2, bufB, v0, v1, v2);
    load_rows<TH>(bufB, rs, lane16, p, 3, h0);
    quad_step<TH, 2>(acc, acc2, W, W2, bufA, v0, v1, v2);
    load_rows<TH>(bufA, rs, lane16, p + 1, 0, h0);
    quad_step<TH, 3>(acc, acc2, W, W2, bufB, v0, v1, v2);
    if (v1) {
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const f32x4 o = relu4(acc2[r]);
        tb[((size_t)(p * D + h0 + r) * 2 + 1) * 64] = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
    if (p - 1 >= d0) {
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const f32x4 o = relu4(acc[0][r]);
        tb[((size_t)((p - 1) * D + h0 + r) * 2 + 0) * 64] = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
#pragma unroll
    for (int r = 0; r < TH; ++r) { acc[0][r] = acc[1][r]; acc[1][r] = acc[2][r]; acc[2][r] = bi; }
  }
}

// pure issue-rate probe: NACC independent accumulators, optional DPP movs interleaved (1 per DPPEVERY MFMAs)
template <int NACC, int DPPEVERY>
__global__ void __launch_bounds__(256) rate_probe(float* out, int iters) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f + 1.f, sh = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = mf<3>(a, sh, acc[i]);
        if (DPPEVERY > 0 && ((u * NACC + i) % DPPEVERY) == 0) sh = shr1(sh) + b;
      }
    }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC, int DPPEVERY>
static void run_rate(float* d, int blocks) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  hipLaunchKernelGGL((rate_probe<NACC, DPPEVERY>), dim3(blocks), dim3(256), 0, 0, d, 10);
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((rate_probe<NACC, DPPEVERY>), dim3(blocks), dim3(256), 0, 0, d, iters);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)blocks * 4 * iters * 8 * NACC;   // wave-level MFMAs
  printf("rate NACC=%d dpp-every=%d blocks=%d: %.1f TFLOP/s (%.2f cycles@2.4GHz per MFMA per SIMD)\n", NACC, DPPEVERY, blocks,
         n * 512 / (ms * 1e-3) * 1e-12, (ms * 1e-3) * 2.4e9 / (n / 1024.0));
}

// ---------------------------------------------------------------------------------------------------
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 8;
  const int iters = argc > 2 ? atoi(argv[2]) : 20;
  constexpr int D = 64;
  // ---- part 1
  {
    float* d; CK(hipMalloc(&d, 384 * 4));
    hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(384); CK(hipMemcpy(h.data(), d, 384 * 4, hipMemcpyDeviceToHost));
    // expectation: D[i][j] of block b in vgpr i of lane 4b+j = A_{abid}[i] * B_b[j] = 100*(20+i) * (4b+j+1)
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < 4; ++i) if (h[lane * 4 + i] != 100.f * (20 + i) * (lane + 1)) ++bad;
    printf("mfma 4x4x1 cbsz=4 abid=5 layout: %s (lane1: %g %g %g %g)\n", bad ? "UNEXPECTED" : "as expected", h[4], h[5], h[6], h[7]);
    int bs = 0;
    for (int lane = 0; lane < 64; ++lane) { if (h[256 + lane] != (float)lane) ++bs; if (h[320 + lane] != (lane == 63 ? 0.f : (float)(lane + 2))) ++bs; }
    printf("dpp wave_shr/wave_shl: %s (shr lanes0..2: %g %g %g ; shl lanes 61..63: %g %g %g)\n", bs ? "UNEXPECTED" : "as expected", h[256], h[257], h[258],
           h[320 + 61], h[320 + 62], h[320 + 63]);
    CK(hipFree(d));
  }
  {
    float* d; CK(hipMalloc(&d, 2048 * 256 * 4));
    run_rate<1, 0>(d, 256); run_rate<2, 0>(d, 256); run_rate<3, 0>(d, 256); run_rate<4, 0>(d, 256); run_rate<6, 0>(d, 256); run_rate<8, 0>(d, 256);
    run_rate<1, 0>(d, 512); run_rate<2, 0>(d, 512); run_rate<3, 0>(d, 512); run_rate<4, 0>(d, 512); run_rate<8, 0>(d, 512);
    run_rate<2, 0>(d, 768); run_rate<4, 0>(d, 768);
    CK(hipFree(d));
    if (argc > 3) return 0;
  }
  // ---- part 2
  const size_t nx = (size_t)B * D * D * D * 16, nt = (size_t)B * D * D * D * 8;
  std::vector<float> hx(nx), w11(27 * 64), w21(64), b11(4), b21(4);
  unsigned s = 12345;
  for (auto& v : hx) v = frand(s);
  for (auto& v : w11) v = frand(s) * 0.1f;
  for (auto& v : w21) v = frand(s) * 0.3f;
  for (auto& v : b11) v = frand(s);
  for (auto& v : b21) v = frand(s);
  float *dx, *dt, *dw11, *dw21, *db11, *db21;
  CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dt, nt * 4)); CK(hipMalloc(&dw11, w11.size() * 4)); CK(hipMalloc(&dw21, 256));
  CK(hipMalloc(&db11, 16)); CK(hipMalloc(&db21, 16));
  CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw11, w11.data(), w11.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw21, w21.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(db11, b11.data(), 16, hipMemcpyHostToDevice)); CK(hipMemcpy(db21, b21.data(), 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int NV = 4;
  const char* names[NV] = {"<4,4,0>", "<4,4,1>", "<4,4,2:ldscap>", "<4,4,3:ldscap+remap>"};
  const int wavesv[NV] = {B * 16 * 16, B * 16 * 16, B * 16 * 16, B * 16 * 16};
  std::vector<std::vector<float>> tms(NV);
  for (int round = 0; round < 7; ++round)
    for (int v = 0; v < NV; ++v) {
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) {
        const dim3 g(wavesv[v] / 4), bl(256);
        if (v == 0) hipLaunchKernelGGL((vrn16a_row<4, 4, 0>), g, bl, 0, 0, dx, dt, dw11, db11, dw21, db21, B);
        if (v == 1) hipLaunchKernelGGL((vrn16a_row<4, 4, 1>), g, bl, 0, 0, dx, dt, dw11, db11, dw21, db21, B);
        if (v == 2) hipLaunchKernelGGL((vrn16a_row<4, 4, 2>), g, bl, 0, 0, dx, dt, dw11, db11, dw21, db21, B);
        if (v == 3) hipLaunchKernelGGL((vrn16a_row<4, 4, 3>), g, bl, 0, 0, dx, dt, dw11, db11, dw21, db21, B);
      }
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (round) tms[v].push_back(ms * 1e3f / iters);
    }
  for (int v = 0; v < NV; ++v) {
    std::sort(tms[v].begin(), tms[v].end());
    const double med = tms[v][tms[v].size() / 2];
    printf("vrn16a_row%s: %d cubes, min %.1f med %.1f max %.1f us per launch, %.1f TFLOP/s (median)\n", names[v], B, tms[v][0], med,
           tms[v].back(), (double)B * D * D * D * 3584 / (med * 1e-6) * 1e-12);
  }
  hipLaunchKernelGGL((vrn16a_row<4, 4, 1>), dim3(wavesv[1] / 4), dim3(256), 0, 0, dx, dt, dw11, db11, dw21, db21, B);
  CK(hipDeviceSynchronize());
  // check cube 0 against the CPU (Q4 layouts)
  std::vector<float> ht((size_t)D * D * D * 8);
  CK(hipMemcpy(ht.data(), dt, ht.size() * 4, hipMemcpyDeviceToHost));
  auto X = [&](int d, int h, int w, int c) -> float {
    if ((unsigned)d >= (unsigned)D || (unsigned)h >= (unsigned)D || (unsigned)w >= (unsigned)D) return 0.f;
    return hx[(((size_t)(d * D + h) * 4 + c / 4) * 64 + w) * 4 + c % 4];
  };
  double maxerr = 0, maxref = 0;
  unsigned rs = 777;
  for (int n = 0; n < 4000; ++n) {
    rs = rs * 1664525u + 1013904223u;
    int d = (rs >> 8) % D, h = (rs >> 14) % D, w = (rs >> 20) % D;
    if (n < 64) { d = (n & 1) ? D - 1 : 0; h = (n & 2) ? D - 1 : 0; w = (n & 4) ? D - 1 : (n & 8 ? 0 : w); }
    for (int co = 0; co < 4; ++co) {
      double a = b11[co], a2 = b21[co];
      for (int kd = 0; kd < 3; ++kd) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) for (int ci = 0; ci < 16; ++ci)
        a += (double)X(d + kd - 1, h + kh - 1, w + kw - 1, ci) * w11[(((kd * 3 + kh) * 3 + kw) * 16 + ci) * 4 + co];
      for (int ci = 0; ci < 16; ++ci) a2 += (double)X(d, h, w, ci) * w21[ci * 4 + co];
      a = a > 0 ? a : 0; a2 = a2 > 0 ? a2 : 0;
      const double g = ht[(((size_t)(d * D + h) * 2 + 0) * 64 + w) * 4 + co], g2 = ht[(((size_t)(d * D + h) * 2 + 1) * 64 + w) * 4 + co];
      maxerr = fmax(maxerr, fmax(fabs(g - a), fabs(g2 - a2))); maxref = fmax(maxref, fmax(a, a2));
    }
  }
  printf("check: max|err| %.3g (max ref %.3g) -> %s\n", maxerr, maxref, maxerr < 1e-4 * fmax(1.0, maxref) ? "OK" : "MISMATCH");
  return 0;
}
