// The deterministic two-stage reductions of the training step's loss terms (loss.py:8-33 BCE sums; the log-likelihood sums of
// train_hyper.py:193-199) as device functions: pcgc_bce_sums (tail.hip), pcgc_sum_log (train.hip) and the fused
// pcgc_train_loss_sums (train.hip: all three in two launches) run the same bodies with the same block counts, so their
// results are bit-identical.
#pragma once
#include "common.h"

namespace pcgc {

constexpr int kBceBlocks = 1024;
constexpr int kSumBlocks = 512;

// block `bid` of `nb`: partial[bid * 4 + {0,1,2,3}] = {sum -log(1-o) over label == 0, their count, sum -log(o) over label > 0, count}
__device__ __forceinline__ void bce_partial_body(const float* pred, const float* label, int64_t n, double* partial, int bid, int nb,
                                                 double (*sh)[4]) {
  double s0 = 0, c0 = 0, s1 = 0, c1 = 0;
  for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) {
    float o = 1.0f / (1.0f + expf(-pred[i]));
    o = fminf(fmaxf(o, 1e-7f), 1.0f - 1e-7f);
    if (label[i] > 0.f) { s1 += (double)(-logf(o)); c1 += 1.0; }
    else if (label[i] == 0.f) { s0 += (double)(-logf(1.0f - o)); c0 += 1.0; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o); c0 += __shfl_xor(c0, o);
    s1 += __shfl_xor(s1, o); c1 += __shfl_xor(c1, o);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[w][0] = s0; sh[w][1] = c0; sh[w][2] = s1; sh[w][3] = c1; }
  __syncthreads();
  if (threadIdx.x < 4) partial[bid * 4 + threadIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// 64 lanes: lane = 4 * stripe + k sums the partials i = stripe, stripe + 16, ... of sum k (ascending), then lane k adds
// the 16 stripes in stripe order — a fixed order with a serial chain of nblocks / 16 + 16 instead of nblocks additions
// (the one-thread-per-sum loop took 120 us of the training step)
__device__ __forceinline__ void bce_final_body(const double* partial, int nblocks, double* sums4, double* sh) {
  const int k = threadIdx.x & 3, stripe = threadIdx.x >> 2;
  double a = 0;
  for (int i = stripe; i < nblocks; i += 16) a += partial[i * 4 + k];
  sh[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0;
    for (int j = 0; j < 16; ++j) t += sh[4 * j + threadIdx.x];
    sums4[threadIdx.x] = t;
  }
}

__device__ __forceinline__ void sum_log_partial_body(const float* p, int64_t n, double* partial, int bid, int nb, double* sh) {
  double a = 0.0;
  for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < n; i += (int64_t)nb * 256) a += (double)logf(p[i]);
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[bid] = sh[0];
}

__device__ __forceinline__ void sum_final_body(const double* partial, int nb, double* out, double* sh) {   // 64 stripes, then the stripes in order
  double a = 0.0;
  for (int i = threadIdx.x; i < nb; i += 64) a += partial[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int j = 0; j < 64; ++j) s += sh[j];
    *out = s;
  }
}

}  // namespace pcgc
