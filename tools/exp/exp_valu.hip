// Stand-alone experiment (GPU box): where does the time of the C=16 VALU conv kernel go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/exp/exp_valu.hip -o gpurun_out/exp_valu && gpurun_out/exp_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../pcgcv1_amd/csrc/mfma_common.h"
namespace pcgc { void set_error(const char*, ...) {} }
using namespace pcgc;

// MODE 0 full, 1 stage only, 2 compute only;  VS = LDS voxel stride; V2 = voxels per thread along h (1 or 2)
template <int MODE, int VS, int V2>
__global__ void __launch_bounds__(256 / V2) k16_4(const float* x, const float* w, float* y, int D) {
  constexpr int TD = 4, TH = 4, TW = 16, ID = 6, IH = 6, IW = 18;
  __shared__ __attribute__((aligned(16))) float tile[ID * IH * IW * VS];
  const int tw = D / TW, th = D / TH, td = D / TD;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tz = bid % tw; bid /= tw; const int ty = bid % th; bid /= th; const int tx = bid % td; bid /= td;
  const int b = bid, od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
  if (MODE != 2) {
    if (V2 == 1) stage_tile<ID, IH, IW, 4, VS>(tile, x + (int64_t)b * D * D * D * 16, D, 16, od0 - 1, oh0 - 1, ow0 - 1);
    else {  // 128 threads: generic flat staging
      for (int idx = threadIdx.x; idx < ID * IH * IW * 4; idx += 128) {
        const int v = idx >> 2, q = idx & 3; const int zw = v % IW, zh = (v / IW) % IH, zd = v / (IW * IH);
        const int gd = od0 - 1 + zd, gh = oh0 - 1 + zh, gw = ow0 - 1 + zw; float4 val = make_float4(0, 0, 0, 0);
        if ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)D && (unsigned)gw < (unsigned)D)
          val = *reinterpret_cast<const float4*>(x + ((((int64_t)b * D + gd) * D + gh) * D + gw) * 16 + q * 4);
        *reinterpret_cast<float4*>(&tile[v * VS + q * 4]) = val;
      }
    }
  }
  __syncthreads();
  const int wq = threadIdx.x & 15, hq = (threadIdx.x >> 4) & (V2 == 1 ? 3 : 1), dq = threadIdx.x >> (V2 == 1 ? 6 : 5);
  float acc[V2][4];
  for (int v = 0; v < V2; ++v) for (int c = 0; c < 4; ++c) acc[v][c] = 0.f;
  if (MODE != 1) {
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
#pragma unroll 1
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = (kd * 3 + kh) * 3 + kw;
#pragma unroll
          for (int v = 0; v < V2; ++v) {
            const float* xp = &tile[(((dq + kd) * IH + (hq * V2 + v + kh)) * IW + (wq + kw)) * VS];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 xv = *reinterpret_cast<const float4*>(xp + 4 * q);
              const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[v][c] = fmaf(xs[r], w[(tap * 16 + 4 * q + r) * 4 + c], acc[v][c]);
            }
          }
        }
      }
    }
  } else {
    for (int v = 0; v < V2; ++v) acc[v][0] = tile[(threadIdx.x * 7 + v) % (ID * IH * IW * VS)];
  }
#pragma unroll
  for (int v = 0; v < V2; ++v) {
    const int64_t vox = (((int64_t)b * D + od0 + dq) * D + oh0 + hq * V2 + v) * D + ow0 + wq;
    *reinterpret_cast<float4*>(y + vox * 4) = make_float4(acc[v][0], acc[v][1], acc[v][2], acc[v][3]);
  }
}

// variant: two passes of 8 channels (half the LDS per workgroup -> twice the resident waves)
template <int CK, int VS>
__global__ void __launch_bounds__(256) k16_4_split(const float* x, const float* w, float* y, int D) {
  constexpr int TD = 4, TH = 4, TW = 16, ID = 6, IH = 6, IW = 18;
  __shared__ __attribute__((aligned(16))) float tile[ID * IH * IW * VS];
  const int tw = D / TW, th = D / TH, td = D / TD;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tz = bid % tw; bid /= tw; const int ty = bid % th; bid /= th; const int tx = bid % td; bid /= td;
  const int b = bid, od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
  const int wq = threadIdx.x & 15, hq = (threadIdx.x >> 4) & 3, dq = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int cb = 0; cb < 16 / CK; ++cb) {
    if (cb) __syncthreads();
    stage_tile<ID, IH, IW, CK / 4, VS>(tile, x + (int64_t)b * D * D * D * 16 + cb * CK, D, 16, od0 - 1, oh0 - 1, ow0 - 1);
    __syncthreads();
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
#pragma unroll 1
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = (kd * 3 + kh) * 3 + kw;
          const float* xp = &tile[(((dq + kd) * IH + (hq + kh)) * IW + (wq + kw)) * VS];
#pragma unroll
          for (int q = 0; q < CK / 4; ++q) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + 4 * q);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[c] = fmaf(xs[r], w[(tap * 16 + cb * CK + 4 * q + r) * 4 + c], acc[c]);
          }
        }
      }
    }
  }
  const int64_t vox = (((int64_t)b * D + od0 + dq) * D + oh0 + hq) * D + ow0 + wq;
  *reinterpret_cast<float4*>(y + vox * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// variant: 2 output voxels per thread along H with the input rows held in registers across kh
// (LDS bytes per FMA 1.0 -> 0.67, each scalar weight used twice), 2 passes of 8 channels.
// TH2 = tile height (4: 128 threads per workgroup, 8: 256 threads)
template <int TH2, int MODE>
__global__ void __launch_bounds__(TH2 * 32) k16_4_h2(const float* x, const float* w, float* y, int D) {
  constexpr int TD = 4, TH = TH2, TW = 16, ID = 6, IH = TH + 2, IW = 18, VS = 12, CK = 8, NT = TH2 * 32;
  __shared__ __attribute__((aligned(16))) float tile[ID * IH * IW * VS];
  const int tw = D / TW, th = D / TH, td = D / TD;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tz = bid % tw; bid /= tw; const int ty = bid % th; bid /= th; const int tx = bid % td; bid /= td;
  const int b = bid, od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
  const int wq = threadIdx.x & 15, hq = (threadIdx.x >> 4) % (TH / 2), dq = threadIdx.x / (16 * (TH / 2));
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  for (int cb = 0; cb < 16 / CK; ++cb) {
    if (cb) __syncthreads();
    // generic flat staging (any thread count)
    const float* xb = x + (int64_t)b * D * D * D * 16 + cb * CK;
    if (MODE != 2 && TH2 == 8) {
      stage_tile<ID, IH, IW, 2, VS>(tile, xb, D, 16, od0 - 1, oh0 - 1, ow0 - 1);
    } else if (MODE != 2) for (int idx = threadIdx.x; idx < ID * IH * IW * 2; idx += NT) {
      const int v = idx >> 1, q = idx & 1; const int zw = v % IW, zh = (v / IW) % IH, zd = v / (IW * IH);
      const int gd = od0 - 1 + zd, gh = oh0 - 1 + zh, gw = ow0 - 1 + zw; float4 val = make_float4(0, 0, 0, 0);
      if ((unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)D && (unsigned)gw < (unsigned)D)
        val = *reinterpret_cast<const float4*>(xb + (((int64_t)gd * D + gh) * D + gw) * 16 + q * 4);
      *reinterpret_cast<float4*>(&tile[v * VS + q * 4]) = val;
    }
    __syncthreads();
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        float xr[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* xp = &tile[(((dq + kd) * IH + (hq * 2 + r)) * IW + (wq + kw)) * VS];
          const float4 a = *reinterpret_cast<const float4*>(xp), c = *reinterpret_cast<const float4*>(xp + 4);
          xr[r][0] = a.x; xr[r][1] = a.y; xr[r][2] = a.z; xr[r][3] = a.w; xr[r][4] = c.x; xr[r][5] = c.y; xr[r][6] = c.z; xr[r][7] = c.w;
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int tap = (kd * 3 + kh) * 3 + kw;
#pragma unroll
          for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float wv = w[(tap * 16 + cb * CK + ci) * 4 + c];
              acc[0][c] = fmaf(xr[kh][ci], wv, acc[0][c]);
              acc[1][c] = fmaf(xr[kh + 1][ci], wv, acc[1][c]);
            }
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int64_t vox = (((int64_t)b * D + od0 + dq) * D + oh0 + hq * 2 + v) * D + ow0 + wq;
    *reinterpret_cast<float4*>(y + vox * 4) = make_float4(acc[v][0], acc[v][1], acc[v][2], acc[v][3]);
  }
}

// variant: 2 x 8 channels, software-pipelined over the 27 taps: the LDS operand of tap t+1 is fetched into a second
// register set before the FMAs of tap t are issued
template <int VS>
__global__ void __launch_bounds__(256) k16_4_pipe(const float* x, const float* w, float* y, int D) {
  constexpr int TD = 4, TH = 4, TW = 16, ID = 6, IH = 6, IW = 18, CK = 8;
  __shared__ __attribute__((aligned(16))) float tile[ID * IH * IW * VS];
  const int tw = D / TW, th = D / TH, td = D / TD;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tz = bid % tw; bid /= tw; const int ty = bid % th; bid /= th; const int tx = bid % td; bid /= td;
  const int b = bid, od0 = tx * TD, oh0 = ty * TH, ow0 = tz * TW;
  const int wq = threadIdx.x & 15, hq = (threadIdx.x >> 4) & 3, dq = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int cb = 0; cb < 2; ++cb) {
    if (cb) __syncthreads();
    stage_tile<ID, IH, IW, CK / 4, VS>(tile, x + (int64_t)b * D * D * D * 16 + cb * CK, D, 16, od0 - 1, oh0 - 1, ow0 - 1);
    __syncthreads();
    const float* base = &tile[((dq * IH + hq) * IW + wq) * VS];
    float4 c0 = *reinterpret_cast<const float4*>(base), c1 = *reinterpret_cast<const float4*>(base + 4);
#pragma unroll 1
    for (int kd = 0; kd < 3; ++kd) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = (kd * 3 + kh) * 3 + kw;
          // next tap in (kd, kh, kw) order; the last one re-reads a valid address
          int nkw = kw + 1, nkh = kh, nkd = kd;
          if (nkw == 3) { nkw = 0; ++nkh; }
          if (nkh == 3) { nkh = 0; ++nkd; }
          if (nkd == 3) { nkd = 2; nkh = 2; nkw = 2; }
          const float* np_ = base + ((nkd * IH + nkh) * IW + nkw) * VS;
          const float4 n0 = *reinterpret_cast<const float4*>(np_), n1 = *reinterpret_cast<const float4*>(np_ + 4);
          const float xs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
          for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(xs[r], w[(tap * 16 + cb * CK + r) * 4 + c], acc[c]);
          c0 = n0; c1 = n1;
        }
      }
    }
  }
  const int64_t vox = (((int64_t)b * D + od0 + dq) * D + oh0 + hq) * D + ow0 + wq;
  *reinterpret_cast<float4*>(y + vox * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// ceiling probes for the FMA loop of the 16 -> 4 kernel (no staging):
//   SRC 0: x from LDS (ds_read_b128), weights from scalar loads      (= the kernel's loop)
//   SRC 1: x from registers (no LDS reads), weights from scalar loads
//   SRC 2: x from LDS, weights from 8 SGPR-resident values reused    (no scalar loads in the loop)
//   SRC 3: x from registers, weights SGPR-resident                   (pure pk_fma rate)
template <int SRC>
__global__ void __launch_bounds__(256) k16_4_probe(const float* x, const float* w, float* y, int D) {
  constexpr int IH = 6, IW = 18, VS = 20;
  __shared__ __attribute__((aligned(16))) float tile[6 * IH * IW * VS];
  const int wq = threadIdx.x & 15, hq = (threadIdx.x >> 4) & 3, dq = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  float4 r0 = make_float4(wq, hq, dq, 1.f), r1 = make_float4(hq, wq, 2.f, dq), r2 = r0, r3 = r1;
  float wr[8];
  for (int i = 0; i < 8; ++i) wr[i] = w[i];
#pragma unroll 1
  for (int kd = 0; kd < 3; ++kd) {
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tap = (kd * 3 + kh) * 3 + kw;
        float xs[16];
        if (SRC == 0 || SRC == 2) {
          const float* xp = &tile[(((dq + kd) * IH + (hq + kh)) * IW + (wq + kw)) * VS];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float4 v = *reinterpret_cast<const float4*>(xp + 4 * q); xs[4*q] = v.x; xs[4*q+1] = v.y; xs[4*q+2] = v.z; xs[4*q+3] = v.w; }
        } else {
          xs[0]=r0.x; xs[1]=r0.y; xs[2]=r0.z; xs[3]=r0.w; xs[4]=r1.x; xs[5]=r1.y; xs[6]=r1.z; xs[7]=r1.w;
          xs[8]=r2.x; xs[9]=r2.y; xs[10]=r2.z; xs[11]=r2.w; xs[12]=r3.x; xs[13]=r3.y; xs[14]=r3.z; xs[15]=r3.w;
          r0.x += acc[0]; r2.y += acc[1];              // keep the values live and changing
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float wv = (SRC >= 2) ? wr[(r + c) & 7] : w[(tap * 16 + r) * 4 + c];
            acc[c] = fmaf(xs[r], wv, acc[c]);
          }
      }
    }
  }
  const int64_t vox = (int64_t)blockIdx.x * 256 + threadIdx.x;
  *reinterpret_cast<float4*>(y + vox * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

template <int MODE, int VS, int V2>
float run(const float* x, const float* w, float* y, int B, int D) {
  const int blocks = B * (D / 4) * (D / 4) * (D / 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k16_4<MODE, VS, V2>), dim3(blocks), dim3(256 / V2), 0, 0, x, w, y, D);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k16_4<MODE, VS, V2>), dim3(blocks), dim3(256 / V2), 0, 0, x, w, y, D);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1000;
}

int main() {
  const int B = 8, D = 64; const size_t n = (size_t)B * D * D * D;
  float *x, *w, *y; hipMalloc(&x, n * 16 * 4); hipMalloc(&w, 27 * 64 * 4); hipMalloc(&y, n * 4 * 4);
  std::vector<float> h(n * 16); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(x, h.data(), n * 16 * 4, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), 27 * 64 * 4, hipMemcpyHostToDevice);
  printf("full VS20 %.1f us | stage-only %.1f | compute-only %.1f\n", run<0, 20, 1>(x, w, y, B, D), run<1, 20, 1>(x, w, y, B, D), run<2, 20, 1>(x, w, y, B, D));
  printf("full VS16 %.1f us | compute-only VS16 %.1f\n", run<0, 16, 1>(x, w, y, B, D), run<2, 16, 1>(x, w, y, B, D));
  auto run_split = [&](auto kern, const char* name) {
    const int blocks = B * (D / 4) * (D / 4) * (D / 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, w, y, D);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, x, w, y, D);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("channel-split %s: full %.1f us\n", name, ms / 20 * 1000);
  };
  run_split(k16_4_split<8, 12>, "2 x 8 ch VS12");
  run_split(k16_4_split<8, 8>, "2 x 8 ch VS8");
  run_split(k16_4_split<4, 4>, "4 x 4 ch VS4");
  run_split(k16_4_probe<0>, "PROBE lds x + sload w");
  run_split(k16_4_probe<1>, "PROBE reg x + sload w");
  run_split(k16_4_probe<2>, "PROBE lds x + sgpr w");
  run_split(k16_4_probe<3>, "PROBE reg x + sgpr w (pure pk_fma)");
  run_split(k16_4_pipe<12>, "2 x 8 ch VS12 software-pipelined taps");
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    const int b4 = B * (D / 4) * (D / 4) * (D / 16), b8 = B * (D / 4) * (D / 8) * (D / 16);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k16_4_h2<4, 0>), dim3(b4), dim3(128), 0, 0, x, w, y, D);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k16_4_h2<4, 0>), dim3(b4), dim3(128), 0, 0, x, w, y, D);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("H2 register-blocked, tile 4x4x16 / 128 thr: %.1f us\n", ms / 20 * 1000);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k16_4_h2<8, 0>), dim3(b8), dim3(256), 0, 0, x, w, y, D);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k16_4_h2<8, 0>), dim3(b8), dim3(256), 0, 0, x, w, y, D);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("H2 register-blocked, tile 4x8x16 / 256 thr: %.1f us\n", ms / 20 * 1000);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k16_4_h2<8, 2>), dim3(b8), dim3(256), 0, 0, x, w, y, D);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k16_4_h2<8, 2>), dim3(b8), dim3(256), 0, 0, x, w, y, D);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("H2 compute-only, tile 4x8x16 / 256 thr: %.1f us\n", ms / 20 * 1000);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k16_4_h2<4, 2>), dim3(b4), dim3(128), 0, 0, x, w, y, D);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k16_4_h2<4, 2>), dim3(b4), dim3(128), 0, 0, x, w, y, D);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("H2 compute-only, tile 4x4x16 / 128 thr: %.1f us\n", ms / 20 * 1000);
  }
  printf("2 voxels/thread (128 thr) VS20: full %.1f | compute-only %.1f\n", run<0, 20, 2>(x, w, y, B, D), run<2, 20, 2>(x, w, y, B, D));
  return 0;
}
