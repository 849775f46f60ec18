// Backward-pass and optimiser kernels for the train_hyper.py step (train_hyper.py:174-214; loss.py:8-33).
//
// Correctness-first round-1 implementations: convolution backward-data reuses the forward kernels (the
// adjoint of a stride-1 conv is a stride-1 conv with the flipped / transposed filter, the adjoint of the
// stride-2 conv is the stride-2 transposed conv with the SAME filter tensor and vice versa); backward-weights
// is a two-stage deterministic reduction over voxels; the likelihood / loss gradients are the analytic
// derivatives of the forward formulas in entropy.hip with TensorFlow's gradient conventions (sign() has zero
// gradient, maximum() routes the gradient to the larger argument, clip passes it inside the interval).
// No float atomics: every reduction has a fixed order, so a data-parallel replica computes the same bits.
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "loss_sums.h"

namespace pcgc {

// ---------------------------------------------------------------- filter adjoint for stride-1 bwd-data
// w [K^3][Cin][Cout] -> wt [K^3][Cout][Cin] with all three taps flipped
__global__ void flip_transpose_kernel(const float* w, float* wt, int K, int Cin, int Cout) {
  const int total = K * K * K * Cin * Cout;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ci = idx % Cin, co = (idx / Cin) % Cout, tap = idx / (Cin * Cout);
  const int kw = tap % K, kh = (tap / K) % K, kd = tap / (K * K);
  const int ftap = ((K - 1 - kd) * K + (K - 1 - kh)) * K + (K - 1 - kw);
  wt[idx] = w[((size_t)ftap * Cin + ci) * Cout + co];
}

// ---------------------------------------------------------------- dz = dy[slice] * (y > 0)
__global__ void relu_bwd_kernel(const float* dy, int dy_cs, int dy_co, const float* y, float* dz, int64_t nvox, int C) {
  const int64_t total = nvox * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / C;
    const int c = (int)(i - v * C);
    const float g = dy[v * dy_cs + dy_co + c];
    dz[i] = (y == nullptr || y[i] > 0.f) ? g : 0.f;
  }
}

// out = relu(x + concat(t12, t23))   (model_voxception.py:65-67), C = channels of x (a multiple of 8: float4 quads
// never straddle the two halves)
__global__ void vrn_merge_kernel(const float* x, const float* t12, const float* t23, float* out, int64_t nvox, int C) {
  const int Q = C / 4, hq = Q / 2;
  const int64_t total = nvox * Q;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* a4 = reinterpret_cast<const float4*>(t12);
  const float4* b4 = reinterpret_cast<const float4*>(t23);
  float4* o4 = reinterpret_cast<float4*>(out);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / Q;
    const int q = (int)(i - v * Q);
    const float4 r = q < hq ? a4[v * hq + q] : b4[v * hq + q - hq];
    const float4 xv = x4[i];
    o4[i] = float4{fmaxf(xv.x + r.x, 0.f), fmaxf(xv.y + r.y, 0.f), fmaxf(xv.z + r.z, 0.f), fmaxf(xv.w + r.w, 0.f)};
  }
}

// reverse of vrn_merge in one pass: dpre = dout * (out > 0) (unless premasked), then the two path ends' slices masked
// by their own ReLU outputs
__global__ void vrn_bwd_split_kernel(const float* dout, const float* out, const float* t12, const float* t23, float* dpre,
                                     float* dz12, float* dz23, int64_t nvox, int C, int premasked) {
  const int Q = C / 4, hq = Q / 2;
  const int64_t total = nvox * Q;
  const float4* g4 = reinterpret_cast<const float4*>(dout);
  const float4* o4 = reinterpret_cast<const float4*>(out);
  float4* p4 = reinterpret_cast<float4*>(dpre);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / Q;
    const int q = (int)(i - v * Q);
    float4 g = g4[i];
    if (!premasked) {
      const float4 o = o4[i];
      g = float4{o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f};
      p4[i] = g;
    }
    const bool first = q < hq;
    const int64_t k = first ? v * hq + q : v * hq + q - hq;
    const float4 t = t23 ? reinterpret_cast<const float4*>(first ? t12 : t23)[k] : reinterpret_cast<const float4*>(t12)[i];
    reinterpret_cast<float4*>(first ? dz12 : dz23)[k] =
        float4{t.x > 0.f ? g.x : 0.f, t.y > 0.f ? g.y : 0.f, t.z > 0.f ? g.z : 0.f, t.w > 0.f ? g.w : 0.f};
  }
}

// the same with the signs of the pre-residual output as one int32 per voxel (bit c = pre[c] > 0), C <= 32
__global__ void vrn_bwd_split_signs_kernel(const float* dout, const float* out, const int32_t* signs, float* dpre, float* dz12, float* dz23,
                                           int64_t nvox, int C, int premasked) {
  const int Q = C / 4, hq = Q / 2;
  const int64_t total = nvox * Q;
  const float4* g4 = reinterpret_cast<const float4*>(dout);
  const float4* o4 = reinterpret_cast<const float4*>(out);
  float4* p4 = reinterpret_cast<float4*>(dpre);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / Q;
    const int q = (int)(i - v * Q);
    float4 g = g4[i];
    if (!premasked) {
      const float4 o = o4[i];
      g = float4{o.x > 0.f ? g.x : 0.f, o.y > 0.f ? g.y : 0.f, o.z > 0.f ? g.z : 0.f, o.w > 0.f ? g.w : 0.f};
      p4[i] = g;
    }
    const unsigned m = (unsigned)signs[v] >> (4 * q);
    const bool first = q < hq;
    const int64_t k = first ? v * hq + q : v * hq + q - hq;
    reinterpret_cast<float4*>(first ? dz12 : dz23)[k] =
        float4{(m & 1u) ? g.x : 0.f, (m & 2u) ? g.y : 0.f, (m & 4u) ? g.z : 0.f, (m & 8u) ? g.w : 0.f};
  }
}

__global__ void add_inplace_kernel(float* a, const float* b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) a[i] += b[i];
}

// scale = max(|s|, lb)  /  ds = dscale * sign(s) * (|s| >= lb)
__global__ void abs_max_fwd_kernel(const float* s, float lb, float* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = fmaxf(fabsf(s[i]), lb);
}
__global__ void abs_max_bwd_kernel(const float* dscale, const float* s, float lb, float* ds, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = fabsf(s[i]);
    const float sg = s[i] > 0.f ? 1.f : (s[i] < 0.f ? -1.f : 0.f);
    ds[i] = a >= lb ? dscale[i] * sg : 0.f;
  }
}

// ---------------------------------------------------------------- weight gradient
// dW[tap][ci][co] (conv) or dW[tap][co][ci] (tconv) = sum over (b, voxel) x[in(voxel, tap)][ci] * dz[voxel][co].
// grid = (taps, NCHUNK); a block owns one tap and a contiguous range of output voxels (conv) / input voxels
// (tconv), stages 64 voxels of x and dz in LDS at a time and keeps its Cin*Cout partial sums in registers.
constexpr int kDwChunks = 128;
constexpr int kDwPartials = 512;   // workspace slots per weight: max(kDwChunks, train_dw.hip's persistent groups)
constexpr int kDwVox = 64;

template <int MAXPAIRS>   // pairs per thread = ceil(Cin*Cout / 256) <= MAXPAIRS
__global__ void __launch_bounds__(256) conv_dw_partial_kernel(const float* x, const float* dz, float* partial, int B, int Din,
                                                              int Dout, int Cin, int Cout, int K, int mode) {
  __shared__ float xs[kDwVox * 64];
  __shared__ float ds[kDwVox * 64];
  __shared__ float red[256];
  __shared__ int64_t xoff[kDwVox], zoff[kDwVox];
  const int tap = blockIdx.x, chunk = blockIdx.y;
  const int kw = tap % K, kh = (tap / K) % K, kd = tap / (K * K);
  const int pad = (K - 1) / 2;
  const int pb = (K > 2 ? K - 2 : 0) / 2;          // front padding of the stride-2 pair (conv_direct.hip)
  // iteration space: conv -> output voxels (dz grid, Dout); tconv -> input voxels (x grid, Din)
  const int Dit = mode == 2 ? Din : Dout;
  const int64_t nvox = (int64_t)B * Dit * Dit * Dit;
  const int64_t per = (nvox + kDwChunks - 1) / kDwChunks;
  const int64_t v0 = chunk * per, v1 = min(nvox, v0 + per);
  const int npairs = Cin * Cout;
  // few (ci, co) pairs (the 4/8-channel layers): L threads share a pair and split the voxels of every tile
  const int L = (MAXPAIRS == 1 && npairs < 256) ? 256 / npairs : 1;
  const int lane = MAXPAIRS == 1 ? threadIdx.x / npairs : 0;
  float acc[MAXPAIRS];
#pragma unroll
  for (int p = 0; p < MAXPAIRS; ++p) acc[p] = 0.f;

  for (int64_t vb = v0; vb < v1; vb += kDwVox) {
    const int nv = (int)min((int64_t)kDwVox, v1 - vb);
    __syncthreads();
    // per-voxel source offsets once per tile (the decomposition of the voxel index is the expensive part)
    if (threadIdx.x < kDwVox) {
      const int vv = threadIdx.x;
      int64_t xo = -1, zo = -1;
      if (vv < nv) {
        const int64_t v = vb + vv;
        const int w_ = (int)(v % Dit), h_ = (int)((v / Dit) % Dit), d_ = (int)((v / ((int64_t)Dit * Dit)) % Dit);
        const int b = (int)(v / ((int64_t)Dit * Dit * Dit));
        if (mode == 2) {
          xo = v * Cin;
          const int od = 2 * d_ + kd - pb, oh = 2 * h_ + kh - pb, ow = 2 * w_ + kw - pb;   // output voxel o = 2i + k - pb
          if ((unsigned)od < (unsigned)Dout && (unsigned)oh < (unsigned)Dout && (unsigned)ow < (unsigned)Dout) zo = ((((int64_t)b * Dout + od) * Dout + oh) * Dout + ow) * Cout;
        } else {
          int id, ih, iw;
          if (mode == 0) { id = d_ + kd - pad; ih = h_ + kh - pad; iw = w_ + kw - pad; }
          else { id = 2 * d_ + kd - pb; ih = 2 * h_ + kh - pb; iw = 2 * w_ + kw - pb; }
          if ((unsigned)id < (unsigned)Din && (unsigned)ih < (unsigned)Din && (unsigned)iw < (unsigned)Din)
            xo = ((((int64_t)b * Din + id) * Din + ih) * Din + iw) * Cin;
          zo = v * Cout;
        }
      }
      xoff[vv] = xo;
      zoff[vv] = zo;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < nv * Cin; idx += 256) {
      const int vv = idx / Cin, ci = idx - vv * Cin;
      const int64_t o = xoff[vv];
      xs[idx] = o >= 0 ? x[o + ci] : 0.f;
    }
    for (int idx = threadIdx.x; idx < nv * Cout; idx += 256) {
      const int vv = idx / Cout, co = idx - vv * Cout;
      const int64_t o = zoff[vv];
      ds[idx] = o >= 0 ? dz[o + co] : 0.f;
    }
    __syncthreads();
    if (MAXPAIRS == 1) {
      const int pair = threadIdx.x - lane * npairs;
      if (lane < L) {
        const int ci = pair / Cout, co = pair - ci * Cout;
        float a = acc[0];
        for (int vv = lane; vv < nv; vv += L) a = fmaf(xs[vv * Cin + ci], ds[vv * Cout + co], a);
        acc[0] = a;
      }
    } else {
#pragma unroll
      for (int p = 0; p < MAXPAIRS; ++p) {
        const int pair = threadIdx.x + p * 256;
        if (pair < npairs) {
          const int ci = pair / Cout, co = pair - ci * Cout;
          float a = acc[p];
          for (int vv = 0; vv < nv; ++vv) a = fmaf(xs[vv * Cin + ci], ds[vv * Cout + co], a);
          acc[p] = a;
        }
      }
    }
  }
  if (MAXPAIRS == 1) {
    // fixed-order reduction over the L voxel lanes of each pair
    __syncthreads();
    red[threadIdx.x] = acc[0];
    __syncthreads();
    if (threadIdx.x < npairs) {
      float s = 0.f;
      for (int l = 0; l < L; ++l) s += red[l * npairs + threadIdx.x];
      partial[((size_t)chunk * gridDim.x + tap) * npairs + threadIdx.x] = s;
    }
  } else {
#pragma unroll
    for (int p = 0; p < MAXPAIRS; ++p) {
      const int pair = threadIdx.x + p * 256;
      if (pair < npairs) partial[((size_t)chunk * gridDim.x + tap) * npairs + pair] = acc[p];
    }
  }
}

// dW[tap][..] = sum over chunks; conv layout [tap][ci][co], tconv layout [tap][co][ci].  One 16-lane group per
// weight: lane l adds chunks l, l+16, ... in order, then a fixed xor-butterfly joins the 16 lanes — the same
// association for every run.
__device__ __forceinline__ float group16_sum(float s) {
  s += __shfl_xor(s, 8, 64);
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  return s;
}
__global__ void __launch_bounds__(256) conv_dw_final_kernel(const float* partial, float* dw, float* db, int taps, int Cin, int Cout,
                                                            int transposed, int nchunks, int chunk_stride) {
  // chunk_stride = floats per chunk: taps*Cin*Cout, plus Cout bias sums behind them when db != NULL
  const int wn = taps * Cin * Cout;
  const int total = wn + (db ? Cout : 0);
  const int idx = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  float s = 0.f;
  if (idx < total)
    for (int c = l; c < nchunks; c += 16) s += partial[(size_t)c * chunk_stride + idx];
  s = group16_sum(s);
  if (idx >= total || l) return;
  if (idx >= wn) { db[idx - wn] = s; return; }
  const int tap = idx / (Cin * Cout), pair = idx - tap * Cin * Cout;
  const int ci = pair / Cout, co = pair - ci * Cout;
  dw[transposed ? ((size_t)tap * Cout + co) * Cin + ci : (size_t)idx] = s;
}

// db[c] = sum over voxels dz[v][c]: two-stage, fixed order.  Thread t owns channel t % C (256 % C == 0 for every
// layer width here; C = 1 also works) and strides over the block's voxels, so dz is read once, coalesced.
// q4_w != 0: dz is a Q4 tensor [row][C/4][w = q4_w][4] — same voxels per block, same summation order, other addresses
__global__ void __launch_bounds__(256) bias_partial_kernel(const float* dz, float* partial, int64_t nvox, int C, int q4_w = 0) {
  __shared__ __attribute__((aligned(16))) float sh[256 * 4];
  const int64_t per = (nvox + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = blockIdx.x * per, v1 = min(nvox, v0 + per);
  if ((C & 3) == 0) {
    // thread = (voxel lane, channel quad): float4 loads, four voxels per thread in flight (up_2's 134 MB of output gradient
    // took 65 us with one float per thread and load); per channel: lanes in order, each lane its voxels in four interleaved sums
    const int Q = C >> 2, lanes = 256 / Q, q = threadIdx.x % Q, lane = threadIdx.x / Q;
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* d4 = reinterpret_cast<const float4*>(dz);
    auto at = [&](int64_t v) { return q4_w ? d4[((v / q4_w) * Q + q) * q4_w + v % q4_w] : d4[v * Q + q]; };
    if (lane < lanes) {
      int64_t v = v0 + lane;
      for (; v + 3 * (int64_t)lanes < v1; v += 4 * (int64_t)lanes) {
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = at(v + u * (int64_t)lanes);
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u].x += t[u].x; a[u].y += t[u].y; a[u].z += t[u].z; a[u].w += t[u].w; }
      }
      for (int u = 0; v < v1; v += lanes, ++u) { const float4 t = at(v); a[u].x += t.x; a[u].y += t.y; a[u].z += t.z; a[u].w += t.w; }
    }
    float4 r;
    r.x = (a[0].x + a[1].x) + (a[2].x + a[3].x); r.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
    r.z = (a[0].z + a[1].z) + (a[2].z + a[3].z); r.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
    reinterpret_cast<float4*>(sh)[threadIdx.x] = lane < lanes ? r : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if ((int)threadIdx.x < C) {
      float s = 0.f;
      for (int l = 0; l < lanes; ++l) s += sh[(l * Q + ((int)threadIdx.x >> 2)) * 4 + (threadIdx.x & 3)];
      partial[blockIdx.x * C + threadIdx.x] = s;
    }
    return;
  }
  const int lanes = 256 / C, c = threadIdx.x % C, lane = threadIdx.x / C;
  float a = 0.f;
  if (lane < lanes)
    for (int64_t v = v0 + lane; v < v1; v += lanes)
      a += q4_w ? dz[(((v / q4_w) * (C >> 2) + (c >> 2)) * q4_w + v % q4_w) * 4 + (c & 3)] : dz[v * C + c];
  sh[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < C) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += sh[l * C + threadIdx.x];
    partial[blockIdx.x * C + threadIdx.x] = s;
  }
}
__global__ void bias_final_kernel(const float* partial, float* db, int nblocks, int C) {   // 16 lanes per channel
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  float s = 0.f;
  if (c < C)
    for (int b = l; b < nblocks; b += 16) s += partial[b * C + c];
  s = group16_sum(s);
  if (c < C && l == 0) db[c] = s;
}

// Every final reduction of a backward pass in one launch (train_plan.hip): the jobs arrive by value, a block finds
// its job by its first block and runs the same fixed-order sums as conv_dw_final_kernel / bias_final_kernel.
// The same sums with the lanes along the OUTPUT index (a wave reads 64 consecutive floats of a chunk; the per-layer kernels
// above put 16 chunks side by side and used 16 B of every line they touched): a block takes 64 outputs, wave w keeps the
// partial sums l = w, w + 4, w + 8, w + 12 of the 16 (chunks c = l mod 16, ascending), and the 16 are combined through
// LDS in exactly the order of group16_sum's butterfly — bit-identical to conv_dw_final_kernel / bias_final_kernel.
__device__ __forceinline__ float tree16(const float (&t)[16]) {
  return (((t[0] + t[8]) + (t[4] + t[12])) + ((t[2] + t[10]) + (t[6] + t[14]))) +
         (((t[1] + t[9]) + (t[5] + t[13])) + ((t[3] + t[11]) + (t[7] + t[15])));
}
// Jobs whose chunk stride is a multiple of 4 floats (every layer but the 1-channel ones) are read as float4: a block takes
// 256 outputs, each lane four consecutive ones — the same chunks in the same order per output, a quarter of the load
// instructions (the launch moves 400 MB of partial sums; with one float per lane it ran at 3.4 TB/s).
__device__ __forceinline__ void final_store(const FinalJob& j, int idx, int wn, float s) {
  if (j.kind == 1) { j.db[idx] = s; return; }
  if (idx >= wn) { j.db[idx - wn] = s; return; }
  const int tap = idx / (j.Cin * j.Cout), pair = idx - tap * j.Cin * j.Cout;
  const int ci = pair / j.Cout, co = pair - ci * j.Cout;
  j.dw[j.transposed ? ((size_t)tap * j.Cout + co) * j.Cin + ci : (size_t)idx] = s;
}
__host__ __device__ inline bool final_job_v4(const FinalJob& j) { return ((j.kind == 1 ? j.Cout : j.cstride) & 3) == 0; }
__global__ void __launch_bounds__(256) dw_final_jobs_kernel(FinalJobs jobs) {
  __shared__ __attribute__((aligned(16))) float sh[16][256];
  int ji = 0;
  while (ji + 1 < jobs.n && jobs.j[ji + 1].block0 <= (int)blockIdx.x) ++ji;
  const FinalJob& j = jobs.j[ji];
  const int blk = (int)blockIdx.x - j.block0;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wn = j.kind == 1 ? 0 : j.taps * j.Cin * j.Cout;
  const int total = j.kind == 1 ? j.Cout : wn + (j.db ? j.Cout : 0);
  const size_t stride = j.kind == 1 ? (size_t)j.Cout : (size_t)j.cstride;       // bias jobs: partial [nchunks][Cout]
  if (final_job_v4(j)) {
    const int idx = blk * 256 + lane * 4;
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < total) {
      // the four running sums of the wave take chunks w + 4 k, + 16, ... in ascending order each; eight rounds of loads (32
      // float4 per lane) are in flight before their additions — one load per sum at a time left the launch latency-bound
      const float* base = j.partial + idx;
      int c0 = 0;
      for (; c0 + 16 * 8 <= j.nchunks; c0 += 16 * 8) {
        float4 t[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) t[u][k] = *reinterpret_cast<const float4*>(base + (size_t)(c0 + 16 * u + w + 4 * k) * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < 4; ++k) { acc[k].x += t[u][k].x; acc[k].y += t[u][k].y; acc[k].z += t[u][k].z; acc[k].w += t[u][k].w; }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        for (int c = c0 + w + 4 * k; c < j.nchunks; c += 16) {
          const float4 t = *reinterpret_cast<const float4*>(base + (size_t)c * stride);
          acc[k].x += t.x; acc[k].y += t.y; acc[k].z += t.z; acc[k].w += t.w;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(&sh[w + 4 * k][lane * 4]) = acc[k];
    __syncthreads();
    // the 256 outputs of the block, one per thread: coalesced stores
    const int o = blk * 256 + (int)threadIdx.x;
    if (o >= total) return;
    float t[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) t[l] = sh[l][threadIdx.x];
    final_store(j, o, wn, tree16(t));
    return;
  }
  const int idx = blk * 64 + lane;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (idx < total) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int c = w + 4 * k; c < j.nchunks; c += 16) acc[k] += j.partial[(size_t)c * stride + idx];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sh[w + 4 * k][lane] = acc[k];
  __syncthreads();
  if (w || idx >= total) return;
  float t[16];
#pragma unroll
  for (int l = 0; l < 16; ++l) t[l] = sh[l][lane];
  final_store(j, idx, wn, tree16(t));
}

// ---------------------------------------------------------------- Laplace likelihood backward
// loss term = coef * sum log(max(p, bound));  p as in entropy.hip laplace_likelihood
// count != nullptr (the _dev entry points): coef = num / (mul * *count), the expression the host would have formed from the
// count it had to wait for — the step's loss sums then stay on the device until the end of the reverse pass
__global__ void laplace_bwd_kernel(const float* yt, const float* loc, const float* scale, float coef, float bound,
                                   float* dy, float* dloc, float* dscale, int64_t n, const double* count = nullptr, double num = 0.0,
                                   double mul = 0.0) {
  if (count) coef = (float)(num / (mul * *count));
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = yt[i], l = loc[i], b = scale[i];
    float up = v + 0.5f, lo = v - 0.5f;
    const float t = (up + lo) - l;
    const float s = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);
    up = -s * (up - l) + l;
    lo = -s * (lo - l) + l;
    const float au = fabsf(up - l), al = fabsf(lo - l);
    const float eu = expf(-au / b), el = expf(-al / b);
    const float cu = up <= l ? 0.5f * eu : 1.0f - 0.5f * eu;
    const float cl = lo <= l ? 0.5f * el : 1.0f - 0.5f * el;
    const float delta = cu - cl;
    const float p = fabsf(delta);
    float gy = 0.f, gl = 0.f, gb = 0.f;
    if (p >= bound && delta != 0.f) {
      const float sg = delta > 0.f ? 1.f : -1.f;
      const float fu = eu / (2.f * b), fl = el / (2.f * b);          // Laplace density at the two edges
      const float gu = (up <= l ? 1.f : -1.f) * fu * au / b;          // d c / d scale
      const float gq = (lo <= l ? 1.f : -1.f) * fl * al / b;
      const float k = coef / p * sg;
      gy = k * (-s) * (fu - fl);
      gl = k * s * (fu - fl);
      gb = k * (gu - gq);
    }
    dy[i] = gy; dloc[i] = gl; dscale[i] = gb;
  }
}

// ---------------------------------------------------------------- factorized prior backward
// per channel: 44 parameters [m0 3][b0 3][f0 3][m1 9][b1 3][f1 3][m2 9][b2 3][f2 3][m3 3][b3 1][f3 1] in the order
// of the packed tensor list (entropy_model.py:50-66).  Thread t always sees channel t % C (256 % C == 0).
struct FzP {
  float M0[3], b0[3], t0[3], M1[9], b1[3], t1[3], M2[9], b2[3], t2[3], M3[3], b3, t3;
};

__device__ __forceinline__ float softplus_d(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

__device__ void fz_load(const float* p, int C, int c, FzP& P, float* raw /*44 raw values for the chain rule*/) {
  const float* m0 = p;          const float* b0 = m0 + C * 3; const float* f0 = b0 + C * 3;
  const float* m1 = f0 + C * 3; const float* b1 = m1 + C * 9; const float* f1 = b1 + C * 3;
  const float* m2 = f1 + C * 3; const float* b2 = m2 + C * 9; const float* f2 = b2 + C * 3;
  const float* m3 = f2 + C * 3; const float* b3 = m3 + C * 3; const float* f3 = b3 + C;
  int k = 0;
  for (int i = 0; i < 3; ++i) { raw[k++] = m0[c * 3 + i]; P.M0[i] = softplus_d(m0[c * 3 + i]); }
  for (int i = 0; i < 3; ++i) { raw[k++] = b0[c * 3 + i]; P.b0[i] = b0[c * 3 + i]; }
  for (int i = 0; i < 3; ++i) { raw[k++] = f0[c * 3 + i]; P.t0[i] = tanhf(f0[c * 3 + i]); }
  for (int i = 0; i < 9; ++i) { raw[k++] = m1[c * 9 + i]; P.M1[i] = softplus_d(m1[c * 9 + i]); }
  for (int i = 0; i < 3; ++i) { raw[k++] = b1[c * 3 + i]; P.b1[i] = b1[c * 3 + i]; }
  for (int i = 0; i < 3; ++i) { raw[k++] = f1[c * 3 + i]; P.t1[i] = tanhf(f1[c * 3 + i]); }
  for (int i = 0; i < 9; ++i) { raw[k++] = m2[c * 9 + i]; P.M2[i] = softplus_d(m2[c * 9 + i]); }
  for (int i = 0; i < 3; ++i) { raw[k++] = b2[c * 3 + i]; P.b2[i] = b2[c * 3 + i]; }
  for (int i = 0; i < 3; ++i) { raw[k++] = f2[c * 3 + i]; P.t2[i] = tanhf(f2[c * 3 + i]); }
  for (int i = 0; i < 3; ++i) { raw[k++] = m3[c * 3 + i]; P.M3[i] = softplus_d(m3[c * 3 + i]); }
  raw[k++] = b3[c]; P.b3 = b3[c];
  raw[k++] = f3[c]; P.t3 = tanhf(f3[c]);
}

// forward of the chain at x keeping what backward needs; returns the logit
struct FzAct { float x, a0[3], h0[3], a1[3], h1[3], a2[3], h2[3], a3; };
__device__ __forceinline__ float fz_forward(const FzP& P, float x, FzAct& A) {
  A.x = x;
#pragma unroll
  for (int i = 0; i < 3; ++i) { A.a0[i] = P.M0[i] * x + P.b0[i]; A.h0[i] = A.a0[i] + P.t0[i] * tanhf(A.a0[i]); }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    A.a1[i] = ((P.M1[i * 3] * A.h0[0] + P.M1[i * 3 + 1] * A.h0[1]) + P.M1[i * 3 + 2] * A.h0[2]) + P.b1[i];
    A.h1[i] = A.a1[i] + P.t1[i] * tanhf(A.a1[i]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    A.a2[i] = ((P.M2[i * 3] * A.h1[0] + P.M2[i * 3 + 1] * A.h1[1]) + P.M2[i * 3 + 2] * A.h1[2]) + P.b2[i];
    A.h2[i] = A.a2[i] + P.t2[i] * tanhf(A.a2[i]);
  }
  A.a3 = ((P.M3[0] * A.h2[0] + P.M3[1] * A.h2[1]) + P.M3[2] * A.h2[2]) + P.b3;
  return A.a3 + P.t3 * tanhf(A.a3);
}

// accumulate d(params) (w.r.t. the TRANSFORMED params: M, b, t) for upstream gradient g on the logit; returns d/dx
__device__ __forceinline__ float fz_backward(const FzP& P, const FzAct& A, float g, float* G /*44*/) {
  // layer 3
  float th = tanhf(A.a3);
  G[43] += g * th;                                   // t3
  float da3 = g * (1.f + P.t3 * (1.f - th * th));
  G[42] += da3;                                      // b3
  float dh2[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { G[39 + j] += da3 * A.h2[j]; dh2[j] = P.M3[j] * da3; }
  // layer 2
  float da2[3], dh1[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    th = tanhf(A.a2[i]);
    G[36 + i] += dh2[i] * th;                        // t2
    da2[i] = dh2[i] * (1.f + P.t2[i] * (1.f - th * th));
    G[33 + i] += da2[i];                             // b2
#pragma unroll
    for (int j = 0; j < 3; ++j) { G[24 + i * 3 + j] += da2[i] * A.h1[j]; dh1[j] += P.M2[i * 3 + j] * da2[i]; }
  }
  // layer 1
  float da1[3], dh0[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    th = tanhf(A.a1[i]);
    G[21 + i] += dh1[i] * th;                        // t1
    da1[i] = dh1[i] * (1.f + P.t1[i] * (1.f - th * th));
    G[18 + i] += da1[i];                             // b1
#pragma unroll
    for (int j = 0; j < 3; ++j) { G[9 + i * 3 + j] += da1[i] * A.h0[j]; dh0[j] += P.M1[i * 3 + j] * da1[i]; }
  }
  // layer 0
  float dx = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    th = tanhf(A.a0[i]);
    G[6 + i] += dh0[i] * th;                         // t0
    const float da0 = dh0[i] * (1.f + P.t0[i] * (1.f - th * th));
    G[3 + i] += da0;                                 // b0
    G[i] += da0 * A.x;                               // M0
    dx += P.M0[i] * da0;
  }
  return dx;
}

constexpr int kFzBlocks = 256;

__global__ void __launch_bounds__(256) factorized_bwd_kernel(const float* zt, const float* params, float coef, float bound,
                                                             float* dz, float* partial, int64_t n, int C, const double* count = nullptr,
                                                             double num = 0.0, double mul = 0.0) {
  __shared__ float red[256];
  if (count) coef = (float)(num / (mul * *count));
  const int c = threadIdx.x % C;
  FzP P;
  float raw[44];
  fz_load(params, C, c, P, raw);
  float G[44];
#pragma unroll
  for (int k = 0; k < 44; ++k) G[k] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = zt[i];
    FzAct Al, Au;
    const float lo = fz_forward(P, v - 0.5f, Al), up = fz_forward(P, v + 0.5f, Au);
    const float t = lo + up;
    const float s = t > 0.f ? -1.f : (t < 0.f ? 1.f : 0.f);
    const float su = 1.f / (1.f + expf(-s * up)), sl = 1.f / (1.f + expf(-s * lo));
    const float delta = su - sl;
    const float p = fabsf(delta);
    float gz = 0.f;
    if (p >= bound && delta != 0.f) {
      const float k = coef / p * (delta > 0.f ? 1.f : -1.f);
      const float gup = k * su * (1.f - su) * s, glo = -k * sl * (1.f - sl) * s;
      gz = fz_backward(P, Au, gup, G) + fz_backward(P, Al, glo, G);
    }
    dz[i] = gz;
  }
  // chain rule to the raw variables: M = softplus(m) -> sigmoid(m); t = tanh(f) -> 1 - t^2
  const int is_m[44] = {1,1,1, 0,0,0, 0,0,0, 1,1,1,1,1,1,1,1,1, 0,0,0, 0,0,0, 1,1,1,1,1,1,1,1,1, 0,0,0, 0,0,0, 1,1,1, 0, 0};
  const int is_f[44] = {0,0,0, 0,0,0, 1,1,1, 0,0,0,0,0,0,0,0,0, 0,0,0, 1,1,1, 0,0,0,0,0,0,0,0,0, 0,0,0, 1,1,1, 0,0,0, 0, 1};
  if (C <= 64) {
    // threads with the same channel sit C lanes apart: a butterfly inside each wave (fixed order), then the four waves' sums
    // through LDS once for all 44 variables (was: 44 rounds of a serial 256 / C-term sum between two barriers, 74 us for the
    // 32 768 hyper-latents of a training batch)
    __shared__ float wsum[44 * 4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 44; ++k) {
      float g = G[k];
      if (is_m[k]) g *= 1.f / (1.f + expf(-raw[k]));
      if (is_f[k]) { const float th = tanhf(raw[k]); g *= (1.f - th * th); }
      for (int off = C; off < 64; off <<= 1) g += __shfl_xor(g, off, 64);
      if (lane < C) wsum[(k * 4 + wave) * C + lane] = g;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 44 * C; i += 256) {
      const int k = i / C, ch = i - k * C;
      const float s = ((wsum[(k * 4 + 0) * C + ch] + wsum[(k * 4 + 1) * C + ch]) + (wsum[(k * 4 + 2) * C + ch] + wsum[(k * 4 + 3) * C + ch]));
      partial[((size_t)blockIdx.x * C + ch) * 44 + k] = s;
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < 44; ++k) {
    float g = G[k];
    if (is_m[k]) g *= 1.f / (1.f + expf(-raw[k]));
    if (is_f[k]) { const float th = tanhf(raw[k]); g *= (1.f - th * th); }
    red[threadIdx.x] = g;
    __syncthreads();
    if (threadIdx.x < C) {
      float s = 0.f;
      for (int j = threadIdx.x; j < 256; j += C) s += red[j];      // fixed order
      partial[((size_t)blockIdx.x * C + threadIdx.x) * 44 + k] = s;
    }
    __syncthreads();
  }
}

// dparams in the packed tensor order; partial [blocks][C][44]
__global__ void factorized_bwd_final_kernel(const float* partial, float* dparams, int nblocks, int C) {
  const int idx = blockIdx.x * 64 + threadIdx.x;
  if (idx >= C * 44) return;
  const int c = idx / 44, k = idx % 44;
  float s = 0.f;
#pragma unroll 8
  for (int b = 0; b < nblocks; ++b) s += partial[((size_t)b * C + c) * 44 + k];     // loads ahead of the dependent adds
  // position of (c, k) in the packed list
  const int sizes[12] = {3, 3, 3, 9, 3, 3, 9, 3, 3, 3, 1, 1};
  int off = 0, kk = k, t = 0;
  for (; t < 12; ++t) { if (kk < sizes[t]) break; kk -= sizes[t]; off += sizes[t] * C; }
  dparams[off + c * sizes[t] + kk] = s;
}

// ---------------------------------------------------------------- BCE backward (loss.py:8-33)
// d/dpred of w0 * mean_{label=0}(-log(1-o)) + w1 * mean_{label>0}(-log o), o = clip(sigmoid(pred), 1e-7, 1-1e-7)
__global__ void bce_bwd_kernel(const float* pred, const float* label, float w0_over_n0, float w1_over_n1, float* dpred, int64_t n,
                               const double* sums4 = nullptr, double a0 = 0.0, double a1 = 0.0) {
  if (sums4) { w0_over_n0 = (float)(a0 / sums4[1]); w1_over_n1 = (float)(a1 / sums4[3]); }   // pcgc_bce_sums' counts, on the device
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float sgm = 1.0f / (1.0f + expf(-pred[i]));
    const bool inside = sgm >= 1e-7f && sgm <= 1.0f - 1e-7f;
    float g = 0.f;
    if (inside) g = label[i] > 0.f ? -w1_over_n1 * (1.f - sgm) : (label[i] == 0.f ? w0_over_n0 * sgm : 0.f);
    dpred[i] = g;
  }
}

// ---------------------------------------------------------------- sum of logs (deterministic, double)
__global__ void __launch_bounds__(256) sum_log_partial_kernel(const float* p, int64_t n, double* partial) {
  __shared__ double sh[256];
  sum_log_partial_body(p, n, partial, blockIdx.x, gridDim.x, sh);
}
__global__ void sum_final_kernel(const double* partial, int nb, double* out) {
  __shared__ double sh[64];
  sum_final_body(partial, nb, out, sh);
}
// The step's three reductions (BCE sums of the logits, log-likelihood sums of y and of z) in two launches instead of six:
// blocks [0, nb) run the BCE partial sums, the next kSumBlocks the sum of log over lik_y, the last kSumBlocks over lik_z —
// each exactly what its own launch would have run; the final kernel's three blocks finish them.
// ws: [nb * 4][kSumBlocks][kSumBlocks] doubles.
__global__ void __launch_bounds__(256) train_loss_partial_kernel(const float* pred, const float* label, int64_t n, int nb, const float* lik_y,
                                                                 int64_t n_y, const float* lik_z, int64_t n_z, double* ws) {
  __shared__ double sh[256];
  const int b = blockIdx.x;
  if (b < nb) bce_partial_body(pred, label, n, ws, b, nb, reinterpret_cast<double(*)[4]>(sh));
  else if (b < nb + kSumBlocks) sum_log_partial_body(lik_y, n_y, ws + (size_t)nb * 4, b - nb, kSumBlocks, sh);
  else sum_log_partial_body(lik_z, n_z, ws + (size_t)nb * 4 + kSumBlocks, b - nb - kSumBlocks, kSumBlocks, sh);
}
__global__ void train_loss_final_kernel(const double* ws, int nb, double* sums4, double* logs2) {
  __shared__ double sh[64];
  if (blockIdx.x == 0) bce_final_body(ws, nb, sums4, sh);
  else if (blockIdx.x == 1) sum_final_body(ws + (size_t)nb * 4, kSumBlocks, logs2, sh);
  else sum_final_body(ws + (size_t)nb * 4 + kSumBlocks, kSumBlocks, logs2 + 1, sh);
}

// ---------------------------------------------------------------- Adam (tf.train.AdamOptimizer, TF1 form)
// guard: the step's BCE sums {s0, n0, s1, n1} on the device (pcgc_train_loss_sums), or nullptr.  A batch without empty or
// without occupied voxels (n0 == 0 or n1 == 0) divided by zero in the reverse pass: its gradients are inf / NaN and the
// update is skipped — the host reads the sums AFTER queueing this launch (the read-back no longer idles the GPU) and raises.
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2, float eps,
                            const double* guard) {
  if (guard && (guard[1] == 0.0 || guard[3] == 0.0)) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
  }
}

static inline int grid_for(int64_t n, int cap = 4096) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace pcgc

using namespace pcgc;

extern "C" {

size_t pcgc_conv3d_bwd_workspace_bytes(int Cin, int Cout, int ksize) {
  const size_t wn = (size_t)ksize * ksize * ksize * Cin * Cout;
  size_t packed = 0;          // bwd-data runs the forward MFMA kernels on the adjoint filter, packed into the workspace
  for (int mode = 0; mode < 3; ++mode) {
    const size_t n = mfma_packed_floats(Cout, Cin, ksize, mode);
    packed = n > packed ? n : packed;
  }
  return (wn + (wn + 64) * kDwPartials + 4096 * 64 + 1024 + packed) * sizeof(float);
}

int pcgc_conv3d_bwd_data(const float* dz, const float* kernel, float* dx, int B, int D, int Cin, int Cout, int ksize,
                         int stride, int transposed, void* workspace, size_t workspace_bytes, pcgc_stream_t stream) {
  return pcgc_conv3d_bwd_data_fused(dz, kernel, dx, nullptr, nullptr, B, D, Cin, Cout, ksize, stride, transposed, workspace,
                                    workspace_bytes, stream);
}

int pcgc_conv3d_bwd_data_fused(const float* dz, const float* kernel, float* dx, const float* relu_mask, const float* add_to, int B,
                               int D, int Cin, int Cout, int ksize, int stride, int transposed, void* workspace,
                               size_t workspace_bytes, pcgc_stream_t stream) {
  PCGC_REQUIRE(dz && kernel && dx && workspace, "pcgc_conv3d_bwd_data: NULL argument");
  PCGC_REQUIRE(workspace_bytes >= pcgc_conv3d_bwd_workspace_bytes(Cin, Cout, ksize), "pcgc_conv3d_bwd_data: workspace too small");
  float* wt = reinterpret_cast<float*>(workspace);
  return bwd_data_impl(dz, kernel, nullptr, nullptr, dx, relu_mask, add_to, B, D, Cin, Cout, ksize, stride, transposed, wt,
                       wt + (size_t)ksize * ksize * ksize * Cin * Cout, (hipStream_t)stream);
}

}  // extern "C"

namespace pcgc {

// Gradient w.r.t. the input of one layer: the forward kernels on the adjoint filter.  wt_ready / packed_ready: the
// flipped filter (stride-1 layers) and its MFMA packing prepared beforehand (train_plan.hip); when nullptr they are
// produced here into wt_scratch / packed_scratch.
int bwd_data_impl(const float* dz, const float* kernel, const float* wt_ready, const float* packed_ready, float* dx,
                  const float* relu_mask, const float* add_to, int B, int D, int Cin, int Cout, int ksize, int stride, int transposed,
                  float* wt_scratch, float* packed_scratch, hipStream_t s, int x_q4, int dz_q4) {
  if (B == 0) return 0;
  if ((x_q4 || dz_q4) && !transposed && stride == 1) {
    // Q4 tensors (Trainer(q4=True)): of the stride-1 layers only deconv_out's reverse comes here (the blocks' layers have
    // fused reverse kernels): dz has one channel, dx 16 in Q4 = conv_in's row kernel on the flipped filter [27][1][16]
    if (!(Cin == 16 && Cout == 1 && ksize == 3 && D == 64 && x_q4 && wt_ready && !add_to)) {
      set_error("bwd-data on Q4 tensors: only deconv_out's shape (16 -> 1 at 64^3) has a kernel (Cin=%d Cout=%d k=%d D=%d)", Cin, Cout, ksize, D);
      return -1;
    }
    return launch_conv_in_row(dz, dx, wt_ready, nullptr, B, 0, s, nullptr, relu_mask);
  }
  ConvArgs a;
  a.x_q4 = dz_q4; a.y_q4 = x_q4;                          // the adjoint convolution reads dz and writes dx (mask / add_to laid out like dx)
  a.x = dz; a.bias = nullptr; a.y = dx; a.res = nullptr; a.B = B;
  a.relu = 0; a.absval = 0; a.lower_bound = 0.f; a.ksize = ksize;
  a.w2 = nullptr; a.bias2 = nullptr; a.y2 = nullptr; a.y2_cs = 0; a.cout2 = 0;
  a.Cin = Cout; a.Cout = Cin; a.x_cs = Cout; a.x_co = 0; a.y_cs = Cin; a.y_co = 0;
  a.mask = relu_mask; a.add_to = add_to;
  if (!transposed && stride == 1) {
    if (wt_ready) {
      a.w = wt_ready;
    } else {
      const int total = ksize * ksize * ksize * Cin * Cout;
      hipLaunchKernelGGL(flip_transpose_kernel, dim3((total + 255) / 256), dim3(256), 0, s, kernel, wt_scratch, ksize, Cin, Cout);
      a.w = wt_scratch;
    }
    a.mode = 0; a.Din = D; a.Dout = D;
  } else if (!transposed) {            // adjoint of the stride-2 conv = transposed conv with the same tensor
    a.w = kernel; a.mode = 2; a.Din = D / 2; a.Dout = D;
  } else {                             // adjoint of the transposed conv = stride-2 conv with the same tensor
    a.w = kernel; a.mode = 1; a.Din = 2 * D; a.Dout = D;
  }
  // same tile kernels as the forward pass: scalar-weight VALU for the 4/8-channel shapes, MFMA otherwise
  int rc = (x_q4 || dz_q4) ? 0 : launch_conv_valu(a, s, true);
  if (rc != 0) return rc < 0 ? rc : 0;
  if (launch_conv_mfma(a, nullptr, s, false) == 1) {
    const float* packed = packed_ready;
    if (!packed) {
      rc = pack_weights_mfma(a.w, packed_scratch, a.Cin, a.Cout, ksize, a.mode, s);
      if (rc) return rc;
      packed = packed_scratch;
    }
    rc = launch_conv_mfma(a, packed, s, true);
    return rc < 0 ? rc : 0;
  }
  if (x_q4 || dz_q4) { set_error("bwd-data on Q4 tensors: no MFMA kernel for Cin=%d Cout=%d stride=%d transposed=%d", Cin, Cout, stride, transposed); return -1; }
  if ((rc = launch_hyper_row_conv(a, s)) != 0) return rc < 0 ? rc : 0;        // the 8^3 hyper layers: row kernels
  return launch_conv_direct(a, s);
}

// floats of partial sums one layer's weight gradient needs at (B, D): *bias_floats more for the bias-only partials
size_t bwd_weight_partial_floats(int B, int D, int Cin, int Cout, int ksize, int stride, int transposed, size_t* bias_floats) {
  const size_t wn = (size_t)ksize * ksize * ksize * Cin * Cout;
  const int groups = transposed ? conv_dw_tile_groups_s2(B, D) : (stride == 2 ? conv_dw_tile_groups_s2(B, D / 2) : conv_dw_tile_groups(B, D));
  *bias_floats = (size_t)1024 * Cout;
  return (size_t)std::max(groups, kDwChunks) * (wn + Cout);
}

// Weight (and bias) gradient of one layer.  sink == nullptr: the final reductions are launched here; otherwise they
// are appended to *sink and the caller runs them later in one launch (launch_final_jobs) — `partial` must then stay
// untouched until that launch.
int bwd_weight_impl(const float* x, const float* dz, float* dkernel, float* dbias, int B, int D, int Cin, int Cout, int ksize,
                    int stride, int transposed, float* partial, float* bp, std::vector<FinalJob>* sink, hipStream_t s, int x_q4, int dz_q4) {
  const int mode = transposed ? 2 : (stride == 2 ? 1 : 0);
  const int Dout = transposed ? 2 * D : D / stride;
  const int taps = ksize * ksize * ksize;
  int nchunks = kDwChunks;
  const int wn = taps * Cin * Cout;
  const bool tile_bias = dbias && 256 % Cout == 0;
  int rc = 0, tile_groups = 0;
  bool tile_has_bias = tile_bias;
  if (mode == 0) {
    rc = launch_conv_dw_tile(x, dz, partial, B, D, Cin, Cout, ksize, tile_bias ? 1 : 0, s, x_q4, dz_q4);
    tile_groups = conv_dw_tile_groups(B, D);
  } else if (ksize == 3 && mode == 1) {       // stride-2 conv: x on the fine grid, result [tap][ci][co]
    if (dz_q4) { set_error("weight gradient of a stride-2 conv: the coarse operand cannot be Q4"); return -1; }
    rc = launch_conv_dw_tile_s2(x, dz, partial, B, Dout, Cin, Cout, tile_bias ? 1 : 0, s, x_q4);
    tile_groups = conv_dw_tile_groups_s2(B, Dout);
  } else if (ksize == 3 && mode == 2) {       // transposed conv: dz on the fine grid, result [tap][co][ci] = its TF layout
    if (x_q4) { set_error("weight gradient of a transposed conv: the coarse operand cannot be Q4"); return -1; }
    rc = launch_conv_dw_tile_s2(dz, x, partial, B, D, Cout, Cin, 0, s, dz_q4);
    tile_groups = conv_dw_tile_groups_s2(B, D);
    tile_has_bias = false;                    // the bias sums run over dz, which is the halo operand here
  }
  if (rc < 0) return rc;
  if (rc != 1 && (x_q4 || dz_q4)) { set_error("weight gradient on Q4 tensors: no tiled kernel for Cin=%d Cout=%d k=%d mode=%d", Cin, Cout, ksize, mode); return -1; }
  if (rc == 1) nchunks = tile_groups;
  auto final_dw = [&](float* db, int cin, int cout, int tr, int cstride) {
    if (sink) {
      sink->push_back(FinalJob{partial, dkernel, db, 0, taps, cin, cout, tr, nchunks, cstride, 0});
    } else {
      hipLaunchKernelGGL(conv_dw_final_kernel, dim3((cstride + 15) / 16), dim3(256), 0, s, partial, dkernel, db, taps, cin, cout, tr,
                         nchunks, cstride);
    }
  };
  if (rc == 1 && mode == 2) {
    final_dw(nullptr, Cout, Cin, 0, wn);
  } else if (rc == 1) {                 // tiled path: weights (and bias sums) in one partial buffer, one final reduction
    final_dw(tile_has_bias ? dbias : nullptr, Cin, Cout, transposed, wn + (tile_has_bias ? Cout : 0));
    if (!dbias || tile_has_bias) return launch_ok("conv bwd-weight kernels");
  } else {
    dim3 grid(taps, kDwChunks);
    const int pairs = (Cin * Cout + 255) / 256;
    if (pairs <= 1) hipLaunchKernelGGL(conv_dw_partial_kernel<1>, grid, dim3(256), 0, s, x, dz, partial, B, D, Dout, Cin, Cout, ksize, mode);
    else if (pairs <= 4) hipLaunchKernelGGL(conv_dw_partial_kernel<4>, grid, dim3(256), 0, s, x, dz, partial, B, D, Dout, Cin, Cout, ksize, mode);
    else hipLaunchKernelGGL(conv_dw_partial_kernel<16>, grid, dim3(256), 0, s, x, dz, partial, B, D, Dout, Cin, Cout, ksize, mode);
    final_dw(nullptr, Cin, Cout, transposed, wn);
  }
  if (dbias) {
    const int64_t nvox = (int64_t)B * Dout * Dout * Dout;
    const int nb = (int)std::min<int64_t>(1024, (nvox + 1023) / 1024);
    hipLaunchKernelGGL(bias_partial_kernel, dim3(nb), dim3(256), 0, s, dz, bp, nvox, Cout, dz_q4 ? Dout : 0);   // Cout divides 256 or is < 256
    if (sink) sink->push_back(FinalJob{bp, nullptr, dbias, 1, 0, 0, Cout, 0, nb, 0, 0});
    else hipLaunchKernelGGL(bias_final_kernel, dim3((Cout + 15) / 16), dim3(256), 0, s, bp, dbias, nb, Cout);
  }
  return launch_ok("conv bwd-weight kernels");
}

int launch_final_jobs(const std::vector<FinalJob>& jobs, hipStream_t s) {
  for (size_t at = 0; at < jobs.size(); at += FinalJobs::kMax) {
    FinalJobs fj;
    fj.n = (int)std::min<size_t>(FinalJobs::kMax, jobs.size() - at);
    int blocks = 0;
    for (int i = 0; i < fj.n; ++i) {
      fj.j[i] = jobs[at + i];
      fj.j[i].block0 = blocks;
      const FinalJob& j = fj.j[i];
      const int total = j.kind == 1 ? j.Cout : j.taps * j.Cin * j.Cout + (j.db ? j.Cout : 0);
      blocks += final_job_v4(j) ? (total + 255) / 256 : (total + 63) / 64;
    }
    hipLaunchKernelGGL(dw_final_jobs_kernel, dim3(blocks), dim3(256), 0, s, fj);
  }
  return launch_ok("dw_final_jobs_kernel");
}

}  // namespace pcgc

extern "C" {

int pcgc_conv3d_bwd_weight(const float* x, const float* dz, float* dkernel, float* dbias, int B, int D, int Cin, int Cout,
                           int ksize, int stride, int transposed, void* workspace, size_t workspace_bytes, pcgc_stream_t stream) {
  PCGC_REQUIRE(x && dz && dkernel && workspace, "pcgc_conv3d_bwd_weight: NULL argument");
  PCGC_REQUIRE(workspace_bytes >= pcgc_conv3d_bwd_workspace_bytes(Cin, Cout, ksize), "pcgc_conv3d_bwd_weight: workspace too small");
  PCGC_REQUIRE(Cin <= 64 && Cout <= 64, "pcgc_conv3d_bwd_weight: at most 64 channels");
  float* partial = reinterpret_cast<float*>(workspace);
  const size_t wn = (size_t)ksize * ksize * ksize * Cin * Cout;
  return bwd_weight_impl(x, dz, dkernel, dbias, B, D, Cin, Cout, ksize, stride, transposed, partial, partial + (size_t)kDwPartials * (wn + 64),
                         nullptr, (hipStream_t)stream);
}

int pcgc_relu_bwd(const float* dy, int dy_cs, int dy_co, const float* y, float* dz, int64_t nvox, int C, pcgc_stream_t stream) {
  PCGC_REQUIRE(dy && dz, "pcgc_relu_bwd: NULL argument");
  if (nvox * C == 0) return 0;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(nvox * C)), dim3(256), 0, (hipStream_t)stream, dy, dy_cs, dy_co, y, dz, nvox, C);
  return launch_ok("relu_bwd_kernel");
}

int pcgc_vrn_merge(const float* x, const float* t12, const float* t23, float* out, int64_t nvox, int C, pcgc_stream_t stream) {
  PCGC_REQUIRE(x && t12 && t23 && out && C > 0 && C % 8 == 0, "pcgc_vrn_merge: bad arguments (C must be a multiple of 8)");
  if (nvox == 0) return 0;
  hipLaunchKernelGGL(vrn_merge_kernel, dim3(grid_for(nvox * C / 4, 16384)), dim3(256), 0, (hipStream_t)stream, x, t12, t23, out, nvox, C);
  return launch_ok("vrn_merge_kernel");
}

int pcgc_vrn_bwd_split(const float* dout, const float* out, const float* t12, const float* t23, float* dpre, float* dz12,
                       float* dz23, int64_t nvox, int C, int premasked, pcgc_stream_t stream) {
  PCGC_REQUIRE(dout && t12 && dz12 && dz23 && (premasked || (out && dpre)) && C > 0 && C % 8 == 0, "pcgc_vrn_bwd_split: bad argument (C must be a multiple of 8)");
  if (nvox == 0) return 0;
  hipLaunchKernelGGL(vrn_bwd_split_kernel, dim3(grid_for(nvox * C / 4, 16384)), dim3(256), 0, (hipStream_t)stream, dout, out, t12, t23, dpre, dz12,
                     dz23, nvox, C, premasked);
  return launch_ok("vrn_bwd_split_kernel");
}

int pcgc_vrn_bwd_split_signs(const float* dout, const float* out, const int32_t* pre_signs, float* dpre, float* dz12, float* dz23,
                             int64_t nvox, int C, int premasked, pcgc_stream_t stream) {
  PCGC_REQUIRE(dout && pre_signs && dz12 && dz23 && (premasked || (out && dpre)) && C > 0 && C % 8 == 0 && C <= 32,
               "pcgc_vrn_bwd_split_signs: bad argument (C a multiple of 8, at most 32)");
  if (nvox == 0) return 0;
  hipLaunchKernelGGL(vrn_bwd_split_signs_kernel, dim3(grid_for(nvox * C / 4, 16384)), dim3(256), 0, (hipStream_t)stream, dout, out, pre_signs,
                     dpre, dz12, dz23, nvox, C, premasked);
  return launch_ok("vrn_bwd_split_signs_kernel");
}

int pcgc_vrn_bwd_tail_supported(int D, int C) { return (D == 64 && C == 16) || (D == 32 && C == 32); }

int pcgc_vrn_bwd_tail(const float* dz12, const float* dz23, const float* t11, const float* t21, const float* t22, const float* kernel12,
                      const float* kernel22, const float* kernel23, float* dt11, float* dt21, float* dt22, int B, int D, int C,
                      pcgc_stream_t stream) {
  PCGC_REQUIRE(pcgc_vrn_bwd_tail_supported(D, C), "pcgc_vrn_bwd_tail: D=%d C=%d has no fused kernel (D = 64 with C = 16, D = 32 with C = 32)", D, C);
  PCGC_REQUIRE(dz12 && dz23 && t11 && t21 && t22 && kernel12 && kernel22 && kernel23 && dt11 && dt21 && dt22 && B >= 0,
               "pcgc_vrn_bwd_tail: bad argument");
  if (B == 0) return 0;
  if (D == 32) return launch_vrn32_bwd_tail(dz12, dz23, t11, t21, t22, kernel12, kernel22, kernel23, dt11, dt21, dt22, B, (hipStream_t)stream);
  return launch_vrn16_bwd_tail(dz12, dz23, t11, t21, t22, kernel12, kernel22, kernel23, dt11, dt21, dt22, B, (hipStream_t)stream);
}

// (D = 32 / C = 32 measured and dropped: with the split folded in the pair-vector kernel reads the 32-channel gradient at a
// 128 B voxel stride and runs 72 us against 46 + 10 us for the two launches — profiles/HISTORY.md)
int pcgc_vrn_bwd_tail_split_supported(int D, int C) { return D == 64 && C == 16; }

// PCGC_DEBUG_SIGNS=1: the one-pass reverse takes the masks (t22 > 0), (t11 > 0), (t21 > 0) from bits 16-27 of the sign words
// (what pcgc_vrn_fwd_train_signs / _q4 write) and never reads t11 / t21 / t22.  A caller that built the words in the plain
// "bit c = pre[c] > 0" form would get dt11 / dt21 / dt22 silently zeroed; with the switch on the entry points compare the bits
// with the tensors first (one extra pass + a sync) and refuse words that do not carry the masks.
__global__ void signs_carry_masks_kernel(const int32_t* signs, const float4* t11, const float4* t21, const float4* t22, int64_t nvox,
                                         unsigned* bad) {
  unsigned n = 0;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * 256) {
    const unsigned w = (unsigned)signs[v];
    const float4 a = t22[v], b = t11[v], c = t21[v];
    const unsigned want = (a.x > 0.f) | (a.y > 0.f) << 1 | (a.z > 0.f) << 2 | (a.w > 0.f) << 3 | (b.x > 0.f) << 4 | (b.y > 0.f) << 5 |
                          (b.z > 0.f) << 6 | (b.w > 0.f) << 7 | (c.x > 0.f) << 8 | (c.y > 0.f) << 9 | (c.z > 0.f) << 10 | (c.w > 0.f) << 11;
    n += ((w >> 16) & 0xFFFu) != want;
  }
  if (n) atomicAdd(bad, n);
}

static int debug_check_signs(const char* who, const int32_t* signs, const float* t11, const float* t21, const float* t22, int B, int D,
                             hipStream_t st) {
  static const bool on = [] { const char* e = getenv("PCGC_DEBUG_SIGNS"); return e && e[0] == '1'; }();
  if (!on) return 0;
  unsigned* bad = nullptr;
  PCGC_CHECK_HIP(hipMalloc((void**)&bad, sizeof(unsigned)));
  hipError_t e = hipMemsetAsync(bad, 0, sizeof(unsigned), st);
  unsigned host = 0;
  if (e == hipSuccess) {
    const int64_t nvox = (int64_t)B * D * D * D;
    signs_carry_masks_kernel<<<dim3(2048), dim3(256), 0, st>>>(signs, (const float4*)t11, (const float4*)t21, (const float4*)t22, nvox, bad);
    e = hipMemcpyAsync(&host, bad, sizeof(unsigned), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  (void)hipFree(bad);
  PCGC_CHECK_HIP(e);
  PCGC_REQUIRE(host == 0, "%s: %u sign words do not carry (t22 > 0), (t11 > 0), (t21 > 0) in bits 16-27 — they were not written by "
               "pcgc_vrn_fwd_train_signs / pcgc_vrn_fwd_train_q4 (PCGC_DEBUG_SIGNS=1)", who, host);
  return 0;
}

int pcgc_vrn_bwd_tail_split(const float* dout, const int32_t* pre_signs, const float* t11, const float* t21, const float* t22,
                            const float* kernel12, const float* kernel22, const float* kernel23, float* dz12, float* dz23, float* dt11,
                            float* dt21, float* dt22, int B, int D, int C, pcgc_stream_t stream) {
  PCGC_REQUIRE(pcgc_vrn_bwd_tail_split_supported(D, C), "pcgc_vrn_bwd_tail_split: D=%d C=%d has no fused kernel (D = 64 with C = 16 only)", D, C);
  PCGC_REQUIRE(dout && pre_signs && t11 && t21 && t22 && kernel12 && kernel22 && kernel23 && dz12 && dz23 && dt11 && dt21 && dt22 && B >= 0,
               "pcgc_vrn_bwd_tail_split: bad argument");
  if (B == 0) return 0;
  if (int rc = debug_check_signs("pcgc_vrn_bwd_tail_split", pre_signs, t11, t21, t22, B, D, (hipStream_t)stream)) return rc;
  return launch_vrn16_bwd_tail_split(dout, pre_signs, t11, t21, t22, kernel12, kernel22, kernel23, dz12, dz23, dt11, dt21, dt22, B,
                                     (hipStream_t)stream);
}

int pcgc_vrn_bwd_tail_split_q4(const float* dout, const int32_t* pre_signs, const float* t11, const float* t21, const float* t22,
                               const float* kernel12, const float* kernel22, const float* kernel23, float* dz12, float* dz23, float* dt11,
                               float* dt21, float* dt22, int B, int D, int C, pcgc_stream_t stream) {
  PCGC_REQUIRE(D == 64 && C == 16, "pcgc_vrn_bwd_tail_split_q4: D=%d C=%d has no fused kernel (D = 64 with C = 16 only)", D, C);
  PCGC_REQUIRE(dout && pre_signs && t11 && t21 && t22 && kernel12 && kernel22 && kernel23 && dz12 && dz23 && dt11 && dt21 && dt22 && B >= 0,
               "pcgc_vrn_bwd_tail_split_q4: bad argument");
  if (B == 0) return 0;
  if (int rc = debug_check_signs("pcgc_vrn_bwd_tail_split_q4", pre_signs, t11, t21, t22, B, D, (hipStream_t)stream)) return rc;
  return launch_vrn16_bwd_tail_split(dout, pre_signs, t11, t21, t22, kernel12, kernel22, kernel23, dz12, dz23, dt11, dt21, dt22, B,
                                     (hipStream_t)stream, true);
}

int pcgc_vrn_bwd_input_supported(int D, int C) { return (D == 64 && C == 16) || (D == 32 && C == 32); }

int pcgc_vrn_bwd_input_q4(const float* dt11, const float* dt21, const float* dpre, const float* x_mask, const float* kernel11,
                          const float* kernel21, float* dx, int B, int D, int C, pcgc_stream_t stream) {
  PCGC_REQUIRE(D == 64 && C == 16, "pcgc_vrn_bwd_input_q4: D=%d C=%d has no fused kernel (D = 64 with C = 16 only)", D, C);
  PCGC_REQUIRE(dt11 && dt21 && dpre && kernel11 && kernel21 && dx && B >= 0, "pcgc_vrn_bwd_input_q4: bad argument");
  if (B == 0) return 0;
  return launch_vrn16_bwd_input(dt11, dt21, dpre, x_mask, kernel11, kernel21, dx, B, (hipStream_t)stream, true);
}

int pcgc_vrn_bwd_input(const float* dt11, const float* dt21, const float* dpre, const float* x_mask, const float* kernel11,
                       const float* kernel21, float* dx, int B, int D, int C, pcgc_stream_t stream) {
  PCGC_REQUIRE(pcgc_vrn_bwd_input_supported(D, C), "pcgc_vrn_bwd_input: D=%d C=%d has no fused kernel (D = 64 with C = 16, D = 32 with C = 32)", D, C);
  PCGC_REQUIRE(dt11 && dt21 && dpre && kernel11 && kernel21 && dx && B >= 0, "pcgc_vrn_bwd_input: bad argument");
  if (B == 0) return 0;
  if (D == 32) return launch_vrn32_bwd_input(dt11, dt21, dpre, x_mask, kernel11, kernel21, dx, B, (hipStream_t)stream);
  return launch_vrn16_bwd_input(dt11, dt21, dpre, x_mask, kernel11, kernel21, dx, B, (hipStream_t)stream);
}

int pcgc_add_inplace(float* a, const float* b, int64_t n, pcgc_stream_t stream) {
  PCGC_REQUIRE(a && b, "pcgc_add_inplace: NULL argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, n);
  return launch_ok("add_inplace_kernel");
}

int pcgc_abs_max(const float* s_raw, float lower_bound, const float* dscale, float* out, int64_t n, pcgc_stream_t stream) {
  PCGC_REQUIRE(s_raw && out, "pcgc_abs_max: NULL argument");
  if (n == 0) return 0;
  if (dscale) hipLaunchKernelGGL(abs_max_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dscale, s_raw, lower_bound, out, n);
  else hipLaunchKernelGGL(abs_max_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, s_raw, lower_bound, out, n);
  return launch_ok("abs_max kernel");
}

int pcgc_laplace_likelihood_bwd(const float* values, const float* loc, const float* scale, float coef, float likelihood_bound,
                                float* dvalues, float* dloc, float* dscale, int64_t n, pcgc_stream_t stream) {
  PCGC_REQUIRE(values && loc && scale && dvalues && dloc && dscale, "pcgc_laplace_likelihood_bwd: NULL argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(laplace_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, values, loc, scale, coef,
                     likelihood_bound, dvalues, dloc, dscale, n);
  return launch_ok("laplace_bwd_kernel");
}

int pcgc_laplace_likelihood_bwd_dev(const float* values, const float* loc, const float* scale, double num, double mul, const double* count,
                                    float likelihood_bound, float* dvalues, float* dloc, float* dscale, int64_t n, pcgc_stream_t stream) {
  PCGC_REQUIRE(values && loc && scale && dvalues && dloc && dscale && count, "pcgc_laplace_likelihood_bwd_dev: NULL argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(laplace_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, values, loc, scale, 0.f,
                     likelihood_bound, dvalues, dloc, dscale, n, count, num, mul);
  return launch_ok("laplace_bwd_kernel");
}

size_t pcgc_factorized_bwd_workspace_bytes(int C) { return (size_t)kFzBlocks * C * 44 * sizeof(float); }

int pcgc_factorized_likelihood_bwd_dev(const float* values, const float* params, double num, double mul, const double* count,
                                       float likelihood_bound, float* dvalues, float* dparams, int64_t n, int C, void* workspace,
                                       size_t workspace_bytes, pcgc_stream_t stream) {
  PCGC_REQUIRE(values && params && dvalues && dparams && workspace && count, "pcgc_factorized_likelihood_bwd_dev: NULL argument");
  PCGC_REQUIRE(C > 0 && 256 % C == 0 && n % C == 0, "pcgc_factorized_likelihood_bwd_dev: C=%d must divide 256", C);
  PCGC_REQUIRE(workspace_bytes >= pcgc_factorized_bwd_workspace_bytes(C), "pcgc_factorized_likelihood_bwd_dev: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(factorized_bwd_kernel, dim3(kFzBlocks), dim3(256), 0, s, values, params, 0.f, likelihood_bound, dvalues,
                     (float*)workspace, n, C, count, num, mul);
  hipLaunchKernelGGL(factorized_bwd_final_kernel, dim3((C * 44 + 63) / 64), dim3(64), 0, s, (const float*)workspace, dparams, kFzBlocks, C);
  return launch_ok("factorized bwd kernels");
}

int pcgc_bce_bwd_dev(const float* pred, const float* label, const double* sums4, double a0, double a1, float* dpred, int64_t n,
                     pcgc_stream_t stream) {
  PCGC_REQUIRE(pred && label && dpred && sums4, "pcgc_bce_bwd_dev: NULL argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, label, 0.f, 0.f, dpred, n, sums4, a0, a1);
  return launch_ok("bce_bwd_kernel");
}

int pcgc_factorized_likelihood_bwd(const float* values, const float* params, float coef, float likelihood_bound, float* dvalues,
                                   float* dparams, int64_t n, int C, void* workspace, size_t workspace_bytes,
                                   pcgc_stream_t stream) {
  PCGC_REQUIRE(values && params && dvalues && dparams && workspace, "pcgc_factorized_likelihood_bwd: NULL argument");
  PCGC_REQUIRE(C > 0 && 256 % C == 0 && n % C == 0, "pcgc_factorized_likelihood_bwd: C=%d must divide 256", C);
  PCGC_REQUIRE(workspace_bytes >= pcgc_factorized_bwd_workspace_bytes(C), "pcgc_factorized_likelihood_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(factorized_bwd_kernel, dim3(kFzBlocks), dim3(256), 0, s, values, params, coef, likelihood_bound, dvalues,
                     (float*)workspace, n, C);
  hipLaunchKernelGGL(factorized_bwd_final_kernel, dim3((C * 44 + 63) / 64), dim3(64), 0, s, (const float*)workspace, dparams, kFzBlocks, C);
  return launch_ok("factorized bwd kernels");
}

int pcgc_bce_bwd(const float* pred, const float* label, float w0_over_n0, float w1_over_n1, float* dpred, int64_t n,
                 pcgc_stream_t stream) {
  PCGC_REQUIRE(pred && label && dpred, "pcgc_bce_bwd: NULL argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, label, w0_over_n0, w1_over_n1, dpred, n);
  return launch_ok("bce_bwd_kernel");
}

size_t pcgc_sum_log_workspace_bytes(void) { return kSumBlocks * sizeof(double); }

int pcgc_sum_log(const float* p, int64_t n, double* out, void* workspace, size_t workspace_bytes, pcgc_stream_t stream) {
  PCGC_REQUIRE(p && out && workspace && workspace_bytes >= pcgc_sum_log_workspace_bytes(), "pcgc_sum_log: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sum_log_partial_kernel, dim3(kSumBlocks), dim3(256), 0, s, p, n, (double*)workspace);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, kSumBlocks, out);
  return launch_ok("sum_log kernels");
}

static int bce_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > kBceBlocks ? kBceBlocks : (b < 1 ? 1 : b));
}
size_t pcgc_train_loss_sums_workspace_bytes(int64_t n) { return ((size_t)bce_blocks(n) * 4 + 2 * kSumBlocks) * sizeof(double); }

int pcgc_train_loss_sums(const float* pred, const float* label, int64_t n, const float* lik_y, int64_t n_y, const float* lik_z, int64_t n_z,
                         double* sums4, double* logs2, void* workspace, size_t workspace_bytes, pcgc_stream_t stream) {
  PCGC_REQUIRE(pred && label && lik_y && lik_z && sums4 && logs2 && workspace && n >= 0 && n_y >= 0 && n_z >= 0 &&
                   workspace_bytes >= pcgc_train_loss_sums_workspace_bytes(n), "pcgc_train_loss_sums: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int nb = bce_blocks(n);
  hipLaunchKernelGGL(train_loss_partial_kernel, dim3(nb + 2 * kSumBlocks), dim3(256), 0, s, pred, label, n, nb, lik_y, n_y, lik_z, n_z,
                     (double*)workspace);
  hipLaunchKernelGGL(train_loss_final_kernel, dim3(3), dim3(64), 0, s, (const double*)workspace, nb, sums4, logs2);
  return launch_ok("train loss sums");
}

int pcgc_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                   float epsilon, pcgc_stream_t stream) {
  PCGC_REQUIRE(param && grad && m && v, "pcgc_adam_step: NULL argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr_t, beta1, beta2, epsilon,
                     (const double*)nullptr);
  return launch_ok("adam_kernel");
}

int pcgc_adam_step_guarded(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2,
                           float epsilon, const double* bce_sums4, pcgc_stream_t stream) {
  PCGC_REQUIRE(param && grad && m && v && bce_sums4, "pcgc_adam_step_guarded: NULL argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr_t, beta1, beta2, epsilon,
                     bce_sums4);
  return launch_ok("adam_kernel");
}

}  // extern "C"
