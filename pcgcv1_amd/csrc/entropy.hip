// Entropy-model kernels (gfx950): quantisation + ranges, Laplace / factorized
// likelihoods, and the per-row Laplace pmf -> 16-bit quantised CDF that feeds the
// host range coder.
//
// Restates, one thread per (voxel, channel) element:
//   models/conditional_entropy_model.py:21-32   _standardized_cumulative
//   models/conditional_entropy_model.py:34-56   _likelihood   (sign(2q - loc) quirk kept)
//   models/conditional_entropy_model.py:95-124  _get_cdf      (+ TF 1.13 pmf_to_quantized_cdf)
//   models/conditional_entropy_model.py:151-154 round, per-cube min / max
//   models/entropy_model.py:72-98, 114-151      _logits_cumulative, _likelihood
//   models/entropy_model.py:199-214             pmf over the integer support
// Built with -ffp-contract=off so every float op rounds once like the reference's
// separate TF ops, and exp / tanh / log / sigmoid are the fixed IEEE-op sequences of
// repro_math.h (not the device library's): a pmf, hence a CDF, is the same bits here,
// in the host library and in the CPU oracle, on any ROCm version.  The decoder
// regenerates the encoder's CDFs from the same (bit-identical) loc/scale, which is what
// keeps the range decoder in sync (README.md:111-114 describes the reference failing at this).
#include <climits>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "repro_math.h"

namespace pcgc {

// --------------------------------------------------------------------------
// round-half-even + per-segment integer min / max (integer atomics: order-free)
// --------------------------------------------------------------------------
__global__ void fill_minmax_kernel(int32_t* mn, int32_t* mx, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) { mn[i] = INT_MAX; mx[i] = INT_MIN; }
}

template <typename Q>
__global__ void __launch_bounds__(256) round_minmax_kernel(const float* x, Q* q, int32_t* seg_min, int32_t* seg_max,
                                                           int64_t seg_len, int blocks_per_seg) {
  const int seg = blockIdx.x / blocks_per_seg, part = blockIdx.x % blocks_per_seg;
  const int64_t base = (int64_t)seg * seg_len;
  int lo = INT_MAX, hi = INT_MIN;
  for (int64_t i = (int64_t)part * 256 + threadIdx.x; i < seg_len; i += (int64_t)blocks_per_seg * 256) {
    const float r = rintf(x[base + i]);
    const int v = (int)r;
    if (q) q[base + i] = (Q)(sizeof(Q) == 2 ? (float)max(-32768, min(32767, v)) : r);   // int16 form: the caller checks the range
    lo = min(lo, v);
    hi = max(hi, v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o));
    hi = max(hi, __shfl_xor(hi, o));
  }
  if ((threadIdx.x & 63) == 0 && lo <= hi) {
    atomicMin(&seg_min[seg], lo);
    atomicMax(&seg_max[seg], hi);
  }
}

// decoded symbols (int16, offset by min_v on the host side of the coder) -> float values: sym + offset
__global__ void __launch_bounds__(256) symbols_to_values_kernel(const int16_t* sym, float offset, float* out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = (float)sym[i] + offset;
}

// ... with one offset per segment (cube): conditional_entropy_model.py:196-199
__global__ void __launch_bounds__(256) symbols_to_values_seg_kernel(const int16_t* sym, const float* seg_offset, float* out, int64_t n,
                                                                    int64_t seg_len) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (float)sym[i] + seg_offset[i / seg_len];
}
// the same, eight symbols per thread (16-byte loads, two 16-byte stores): seg_len a multiple of 8, pointers 16-byte aligned
__global__ void __launch_bounds__(256) symbols_to_values_seg8_kernel(const int16_t* sym, const float* seg_offset, float* out, int64_t n8,
                                                                     int64_t seg_len8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const uint4 v = reinterpret_cast<const uint4*>(sym)[i];
    const float o = seg_offset[i / seg_len8];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    float4 a, b;
    a.x = (float)(int16_t)(w[0] & 0xffffu) + o; a.y = (float)(int16_t)(w[0] >> 16) + o;
    a.z = (float)(int16_t)(w[1] & 0xffffu) + o; a.w = (float)(int16_t)(w[1] >> 16) + o;
    b.x = (float)(int16_t)(w[2] & 0xffffu) + o; b.y = (float)(int16_t)(w[2] >> 16) + o;
    b.z = (float)(int16_t)(w[3] & 0xffffu) + o; b.w = (float)(int16_t)(w[3] >> 16) + o;
    reinterpret_cast<float4*>(out)[2 * i] = a;
    reinterpret_cast<float4*>(out)[2 * i + 1] = b;
  }
}

// --------------------------------------------------------------------------
// Laplace likelihood
// --------------------------------------------------------------------------
__device__ __forceinline__ float laplace_cdf(float x, float loc, float scale) {
  const float e = repro::expf_(-fabsf(x - loc) / scale);
  const float cl = 0.5f * e;
  const float cr = 1.0f - 0.5f * e;
  return (x <= loc) ? cl : ((x > loc) ? cr : 0.f);   // NaN input -> both masks 0 -> 0 like the reference
}

__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ float laplace_likelihood(float v, float loc, float scale) {
  float upper = v + 0.5f;
  float lower = v - 0.5f;
  const float sign = sgnf((upper + lower) - loc);
  upper = -sign * (upper - loc) + loc;
  lower = -sign * (lower - loc) + loc;
  return fabsf(laplace_cdf(upper, loc, scale) - laplace_cdf(lower, loc, scale));
}

__global__ void __launch_bounds__(256) laplace_likelihood_kernel(const float* y, const float* loc, const float* scale,
                                                                 const float* noise, float* values, float* lik,
                                                                 int64_t n, float bound) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = noise ? (y[i] + noise[i]) : rintf(y[i]);
    if (values) values[i] = v;
    if (lik) lik[i] = fmaxf(laplace_likelihood(v, loc[i], scale[i]), bound);
  }
}

// --------------------------------------------------------------------------
// Laplace pmf -> quantised CDF per row (TF 1.13 pmf_to_quantized_cdf restated).
// TF keeps a queue sorted by key and, after changing the head, re-inserts it behind
// every entry that does not compare strictly worse.  That queue is always sorted by
// (key, age) with age = original index, then a fresh stamp at every re-insertion, so
// "queue[0]" is an arg-min over (key, age): no sort needed, everything stays in
// statically indexed registers.
// --------------------------------------------------------------------------
template <int MAXN>
__global__ void __launch_bounds__(256) laplace_cdf_kernel(const float* loc, const float* scale, const int32_t* seg_min,
                                                          const int32_t* seg_max, int64_t rows, int64_t seg_rows,
                                                          int ncols, float bound, const float* symbols,
                                                          uint16_t* cdf_lower, uint32_t* lohi,
                                                          const double* __restrict__ lg /* lg[v] = log2(v), v in [0, 65537] */) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const int64_t seg = row / seg_rows;
  const int mn = seg_min[seg];
  int N = seg_max[seg] - mn + 1;
  if (N > MAXN) N = MAXN;       // host validates N <= ncols <= MAXN before launching
  const float l = loc[row], s = scale[row];

  int v[MAXN];
  float mass[MAXN];
  double key[MAXN];
  int age[MAXN];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < MAXN; ++k) {
    if (k < N) {
      const float p = fmaxf(laplace_likelihood((float)(mn + k), l, s), bound);
      mass[k] = p;
      int q = (int)rintf(p * 65536.0f);
      q = q < 1 ? 1 : q;
      v[k] = q;
      sum += q;
    } else {
      mass[k] = 0.f; v[k] = 0;
    }
    age[k] = k;
  }
  // The reference's greedy loop moves ONE count per step (queue head = arg-extreme of (key, age)): while the sum
  // is too large it takes a count from the item with the smallest penalty m*(log2 v - log2(v-1)), while it is too
  // small it gives one to the item with the largest gain m*(log2(v+1) - log2 v).  Both directions are the same
  // loop on c_k(v) = dir * m_k * (lg[v+o+1] - lg[v+o]) (dir = +1, o = 0 growing; dir = -1, o = -1 shrinking),
  // arg-max with the older item winning ties.
  //
  // Bulk form.  For one item the keys of its successive picks, s_k(j) = c_k(v_k + sd*j), strictly decrease in j
  // (consecutive table differences differ by >= 3e-10, the table's rounding noise is 2e-15), so the greedy loop is
  // a merge of N decreasing sequences: it consumes their elements in globally descending order.  Hence for ANY
  // threshold lambda with F = #{elements > lambda} <= D, the state after the first F steps is "every item advanced
  // past its elements above lambda", whatever the interleaving.  lambda comes from the continuous water-filling
  // level (s_k(j) ~ m_k / (ln2 (v_k + sd*j + 1/2))) aimed a few counts short of D; the per-item counts are then made
  // exact against the table, the ages are rebuilt from the order of the items' LAST picks (larger last key = picked
  // earlier = older; two items with identical mass and count have identical sequences and keep their relative
  // order), and the few remaining counts go through the plain loop.  A row whose deficit is thousands of counts
  // (little of the mass inside the support) costs one bulk step plus <= N+1 plain steps instead of one iteration
  // per leader change (up to 1 600 measured), which is what a 64-row wave used to wait for.  Results are identical
  // to the step-by-step loop (tests/test_gpu_parity.py::test_laplace_cdf_integer_algorithm_bit_exact).
  if (sum != 65536) {
    constexpr double kLn2 = 0.6931471805599453;
    const bool grow = sum < 65536;
    const double dir = grow ? 1.0 : -1.0;
    const int o = grow ? 0 : -1;
    const int sd = grow ? 1 : -1;
    auto ckey = [&](double m, int vv) -> double {
      return (!grow && vv <= 1) ? -__builtin_huge_val() : dir * (m * (lg[vv + o + 1] - lg[vv + o]));
    };
#pragma unroll
    for (int k = 0; k < MAXN; ++k) key[k] = (k < N) ? ckey((double)mass[k], v[k]) : -__builtin_huge_val();
    int stamp = MAXN;
    int D = grow ? 65536 - sum : sum - 65536;

    if (D > N + 3) {
      bool act[MAXN];
      int nact = 0;
#pragma unroll
      for (int k = 0; k < MAXN; ++k) {
        act[k] = (k < N) && (grow ? mass[k] > 0.f : v[k] >= 2);
        nact += act[k] ? 1 : 0;
      }
      double lvl = 0.0;
      bool ok = false;
      for (int round = 0; round < MAXN && nact > 0; ++round) {      // active-set water-filling: <= N rounds
        double sm = 0.0, sv = 0.0;
#pragma unroll
        for (int k = 0; k < MAXN; ++k)
          if (act[k]) { sm += (double)mass[k]; sv += (double)v[k] + (grow ? 0.5 : -0.5); }
        const double target = (double)(D - nact - 1);
        const double den = grow ? target + sv : sv - target;
        ok = den > 0.0 && sm > 0.0 && target > 0.0;
        if (!ok) break;
        lvl = sm / (kLn2 * den);
        const double inv = 1.0 / (kLn2 * lvl);
        bool removed = false;
#pragma unroll
        for (int k = 0; k < MAXN; ++k)
          if (act[k]) {
            const double e = grow ? (double)mass[k] * inv - (double)v[k] - 0.5 : (double)v[k] - 0.5 - (double)mass[k] * inv;
            if (e < 0.0) { act[k] = false; --nact; removed = true; }
          }
        if (!removed) break;
        ok = false;
      }
      if (ok && nact > 0 && lvl > 0.0 && lvl < __builtin_huge_val()) {
        const double lam = dir * lvl;
        const double inv = 1.0 / (kLn2 * lvl);
        int cnt[MAXN];
        int F = 0;
#pragma unroll
        for (int k = 0; k < MAXN; ++k) {
          int c = 0;
          if (k < N) {
            const double mk = (double)mass[k];
            const int cap = grow ? D : v[k] - 1;
            const double e = grow ? mk * inv - (double)v[k] - 0.5 : (double)v[k] - 0.5 - mk * inv;
            c = !(e > 0.0) ? 0 : (e >= (double)cap ? cap : (int)e + 1);
            c = c > cap ? cap : c;
            while (c > 0 && !(ckey(mk, v[k] + sd * (c - 1)) > lam)) --c;
            while (c < cap && (ckey(mk, v[k] + sd * c) > lam)) ++c;
          }
          cnt[k] = c;
          F += c;
        }
        if (F > 0 && F <= D) {
          double last[MAXN];
#pragma unroll
          for (int k = 0; k < MAXN; ++k) last[k] = cnt[k] > 0 ? ckey((double)mass[k], v[k] + sd * (cnt[k] - 1)) : 0.0;
          int rank[MAXN];
          int picked = 0;
          bool ambiguous = false;
#pragma unroll
          for (int k = 0; k < MAXN; ++k) {
            int r = 0;
            if (cnt[k] > 0) {
              ++picked;
#pragma unroll
              for (int i = 0; i < MAXN; ++i)
                if (i != k && cnt[i] > 0) {
                  if (last[i] > last[k]) ++r;
                  else if (last[i] == last[k]) {
                    if (mass[i] == mass[k] && v[i] == v[k]) r += age[i] < age[k] ? 1 : 0;    // identical sequences
                    else ambiguous = true;        // coincidental equal doubles: let the plain loop decide
                  }
                }
            }
            rank[k] = r;
          }
          if (!ambiguous) {
#pragma unroll
            for (int k = 0; k < MAXN; ++k)
              if (cnt[k] > 0) {
                age[k] = stamp + rank[k];
                v[k] += sd * cnt[k];
                key[k] = ckey((double)mass[k], v[k]);
              }
            stamp += picked;
            D -= F;
          }
        }
      }
    }
    while (D > 0) {                       // the reference's loop, one count per step
      int h = 0, ba = age[0];
      double bk = key[0];
#pragma unroll
      for (int k = 1; k < MAXN; ++k)
        if (k < N && (key[k] > bk || (key[k] == bk && age[k] < ba))) { h = k; bk = key[k]; ba = age[k]; }
#pragma unroll
      for (int k = 0; k < MAXN; ++k)
        if (k == h) {
          v[k] += sd;
          key[k] = ckey((double)mass[k], v[k]);
          age[k] = stamp;
        }
      ++stamp;
      --D;
    }
  }
  // prefix sums -> lower bounds
  int sym = -1;
  if (symbols) sym = (int)symbols[row] - mn;
  int acc = 0;
  uint32_t lh = 0;
#pragma unroll
  for (int k = 0; k < MAXN; ++k) {
    if (k < ncols && cdf_lower) cdf_lower[row * ncols + k] = (uint16_t)(k < N ? acc : 0xFFFF);
    if (k == sym) lh = (uint32_t)acc | ((uint32_t)(acc + v[k] - 1) << 16);
    acc += v[k];
  }
  if (lohi) lohi[row] = lh;
}

// --------------------------------------------------------------------------
// factorized prior (EntropyBottleneck)
// --------------------------------------------------------------------------
struct FactorizedParams {   // per channel, already transformed: softplus(matrix), bias, tanh(factor)
  float m0[3], b0[3], f0[3];
  float m1[9], b1[3], f1[3];
  float m2[9], b2[3], f2[3];
  float m3[3], b3[1], f3[1];
};

__device__ __forceinline__ float softplusf(float x) { return repro::softplusf_(x); }   // log(exp(x) + 1), stable form
__device__ __forceinline__ float tanhr(float x) { return repro::tanhf_(x); }

__device__ void load_factorized(const float* p, int C, int c, FactorizedParams& P) {
  // tensor order: matrix_0,bais_0,factor_0, matrix_1,... (entropy_model.py:50-66), each [C, rows, cols]
  const float* m0 = p;              const float* b0 = m0 + C * 3; const float* f0 = b0 + C * 3;
  const float* m1 = f0 + C * 3;     const float* b1 = m1 + C * 9; const float* f1 = b1 + C * 3;
  const float* m2 = f1 + C * 3;     const float* b2 = m2 + C * 9; const float* f2 = b2 + C * 3;
  const float* m3 = f2 + C * 3;     const float* b3 = m3 + C * 3; const float* f3 = b3 + C;
  for (int i = 0; i < 3; ++i) { P.m0[i] = softplusf(m0[c * 3 + i]); P.b0[i] = b0[c * 3 + i]; P.f0[i] = tanhr(f0[c * 3 + i]); }
  for (int i = 0; i < 9; ++i) { P.m1[i] = softplusf(m1[c * 9 + i]); P.m2[i] = softplusf(m2[c * 9 + i]); }
  for (int i = 0; i < 3; ++i) {
    P.b1[i] = b1[c * 3 + i]; P.f1[i] = tanhr(f1[c * 3 + i]);
    P.b2[i] = b2[c * 3 + i]; P.f2[i] = tanhr(f2[c * 3 + i]);
    P.m3[i] = softplusf(m3[c * 3 + i]);
  }
  P.b3[0] = b3[c]; P.f3[0] = tanhr(f3[c]);
}

__device__ __forceinline__ float logits_cumulative(const FactorizedParams& P, float x) {
  float a[3], t[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { a[i] = P.m0[i] * x + P.b0[i]; a[i] = a[i] + P.f0[i] * tanhr(a[i]); }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    t[i] = ((P.m1[i * 3 + 0] * a[0] + P.m1[i * 3 + 1] * a[1]) + P.m1[i * 3 + 2] * a[2]) + P.b1[i];
    t[i] = t[i] + P.f1[i] * tanhr(t[i]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    a[i] = ((P.m2[i * 3 + 0] * t[0] + P.m2[i * 3 + 1] * t[1]) + P.m2[i * 3 + 2] * t[2]) + P.b2[i];
    a[i] = a[i] + P.f2[i] * tanhr(a[i]);
  }
  float o = ((P.m3[0] * a[0] + P.m3[1] * a[1]) + P.m3[2] * a[2]) + P.b3[0];
  return o + P.f3[0] * tanhr(o);
}

__device__ __forceinline__ float sigmoidf_(float x) { return repro::sigmoidf_(x); }

__device__ __forceinline__ float factorized_likelihood(const FactorizedParams& P, float v) {
  const float lower = logits_cumulative(P, v - 0.5f);
  const float upper = logits_cumulative(P, v + 0.5f);
  const float sign = -sgnf(lower + upper);
  return fabsf(sigmoidf_(sign * upper) - sigmoidf_(sign * lower));
}

constexpr int kMaxFactorizedC = 64;

__global__ void __launch_bounds__(256) factorized_likelihood_kernel(const float* z, const float* params, const float* noise,
                                                                    float* values, float* lik, int64_t n, int C, float bound) {
  __shared__ FactorizedParams P[kMaxFactorizedC];
  for (int c = threadIdx.x; c < C; c += 256) load_factorized(params, C, c, P[c]);
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = noise ? (z[i] + noise[i]) : rintf(z[i]);
    if (values) values[i] = v;
    if (lik) lik[i] = fmaxf(factorized_likelihood(P[i % C], v), bound);
  }
}

__global__ void factorized_pmf_kernel(const float* params, int C, int min_v, int N, float bound, float* pmf) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= C * N) return;
  const int c = i / N, k = i % N;
  FactorizedParams P;
  load_factorized(params, C, c, P);
  pmf[i] = fmaxf(factorized_likelihood(P, (float)(min_v + k)), bound);
}

// log2 table: the reference's pmf_to_quantized_cdf ranks candidates by mass * (log2(v+1) - log2(v)) in double.
// Tabulating log2(v) for every reachable integer ON THE HOST (same libm as the host quantiser / oracle) makes the
// device keys bit-identical to the host's and replaces two software double log2 per key by two L2-resident loads.
// One 512 KiB table per device, created on first use and kept for the life of the process.
static std::mutex g_lg_mutex;
static std::map<int, double*> g_lg_tables;

static int get_log2_table(const double** out) {
  int dev = 0;
  PCGC_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_lg_mutex);
  auto it = g_lg_tables.find(dev);
  if (it == g_lg_tables.end()) {
    std::vector<double> h(65538);
    h[0] = 0.0;
    for (int v = 1; v < 65538; ++v) h[v] = std::log2((double)v);
    double* d = nullptr;
    PCGC_CHECK_HIP(hipMalloc(&d, h.size() * sizeof(double)));
    PCGC_CHECK_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
    it = g_lg_tables.emplace(dev, d).first;
  }
  *out = it->second;
  return 0;
}

}  // namespace pcgc

using namespace pcgc;

__global__ void __launch_bounds__(256) repro_eval_kernel(int fn, const float* x, float* y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    y[i] = fn == 0 ? repro::expf_(v) : fn == 1 ? repro::logf_(v) : fn == 2 ? repro::tanhf_(v) : fn == 3 ? repro::sigmoidf_(v)
                                                                                                     : repro::softplusf_(v);
  }
}

extern "C" {

int pcgc_repro_eval(int fn, const float* x, float* y, int64_t n, pcgc_stream_t stream) {
  PCGC_REQUIRE(fn >= 0 && fn <= 4 && n >= 0 && (n == 0 || (x && y)), "pcgc_repro_eval: bad arguments");
  if (n == 0) return 0;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(repro_eval_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, fn, x, y, n);
  return launch_ok("repro_eval_kernel");
}

int pcgc_round_minmax(const float* x, float* q, int32_t* seg_min, int32_t* seg_max, int64_t n, int64_t seg_len,
                      pcgc_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return 0;                                   // empty input: valid no-op, pointers may be NULL
  PCGC_REQUIRE(x && seg_min && seg_max && seg_len > 0 && n >= 0 && n % seg_len == 0,
               "pcgc_round_minmax: n=%lld must be a multiple of seg_len=%lld", (long long)n, (long long)seg_len);
  if (n == 0) return 0;
  const int nseg = (int)(n / seg_len);
  hipLaunchKernelGGL(fill_minmax_kernel, dim3((nseg + 255) / 256), dim3(256), 0, s, seg_min, seg_max, nseg);
  int bps = (int)((seg_len + 4095) / 4096);
  if (bps > 1024) bps = 1024;
  hipLaunchKernelGGL(round_minmax_kernel<float>, dim3(nseg * bps), dim3(256), 0, s, x, q, seg_min, seg_max, seg_len, bps);
  return launch_ok("round_minmax_kernel");
}

int pcgc_round_minmax_i16(const float* x, int16_t* q, int32_t* seg_min, int32_t* seg_max, int64_t n, int64_t seg_len,
                          pcgc_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return 0;
  PCGC_REQUIRE(x && q && seg_min && seg_max && seg_len > 0 && n >= 0 && n % seg_len == 0,
               "pcgc_round_minmax_i16: n=%lld must be a multiple of seg_len=%lld", (long long)n, (long long)seg_len);
  const int nseg = (int)(n / seg_len);
  hipLaunchKernelGGL(fill_minmax_kernel, dim3((nseg + 255) / 256), dim3(256), 0, s, seg_min, seg_max, nseg);
  int bps = (int)((seg_len + 4095) / 4096);
  if (bps > 1024) bps = 1024;
  hipLaunchKernelGGL(round_minmax_kernel<int16_t>, dim3(nseg * bps), dim3(256), 0, s, x, q, seg_min, seg_max, seg_len, bps);
  return launch_ok("round_minmax_kernel<int16>");
}

int pcgc_symbols_to_values(const int16_t* sym, int offset, float* out, int64_t n, pcgc_stream_t stream) {
  if (n == 0) return 0;
  PCGC_REQUIRE(sym && out && n >= 0, "pcgc_symbols_to_values: bad argument");
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(symbols_to_values_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, sym,
                     (float)offset, out, n);
  return launch_ok("symbols_to_values_kernel");
}

int pcgc_symbols_to_values_seg(const int16_t* sym, const float* seg_offset, float* out, int64_t n, int64_t seg_len, pcgc_stream_t stream) {
  if (n == 0) return 0;
  PCGC_REQUIRE(sym && seg_offset && out && n >= 0 && seg_len > 0 && n % seg_len == 0, "pcgc_symbols_to_values_seg: bad argument");
  if (seg_len % 8 == 0 && ((uintptr_t)sym | (uintptr_t)out) % 16 == 0) {
    const int64_t blocks8 = (n / 8 + 255) / 256;
    hipLaunchKernelGGL(symbols_to_values_seg8_kernel, dim3((unsigned)(blocks8 < 4096 ? blocks8 : 4096)), dim3(256), 0, (hipStream_t)stream, sym,
                       seg_offset, out, n / 8, seg_len / 8);
    return launch_ok("symbols_to_values_seg8_kernel");
  }
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(symbols_to_values_seg_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, sym,
                     seg_offset, out, n, seg_len);
  return launch_ok("symbols_to_values_seg_kernel");
}

int pcgc_laplace_likelihood(const float* y, const float* loc, const float* scale, const float* noise, float* values,
                            float* likelihood, int64_t n, float likelihood_bound, pcgc_stream_t stream) {
  if (n == 0) return 0;
  PCGC_REQUIRE(y && loc && scale && n >= 0, "pcgc_laplace_likelihood: NULL tensor");
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(laplace_likelihood_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, loc, scale, noise,
                     values, likelihood, n, likelihood_bound);
  return launch_ok("laplace_likelihood_kernel");
}

int pcgc_laplace_cdf(const float* loc, const float* scale, const int32_t* seg_min, const int32_t* seg_max, int64_t rows,
                     int64_t seg_rows, int ncols, float likelihood_bound, const float* symbols, uint16_t* cdf_lower,
                     uint32_t* lohi, pcgc_stream_t stream) {
  if (rows == 0) return 0;
  PCGC_REQUIRE(loc && scale && seg_min && seg_max && seg_rows > 0 && rows % seg_rows == 0, "pcgc_laplace_cdf: bad arguments");
  PCGC_REQUIRE(ncols >= 1 && ncols <= 32, "pcgc_laplace_cdf: ncols=%d outside [1,32] (the container stores |min|,|max| <= 15)", ncols);
  PCGC_REQUIRE(!lohi || symbols, "pcgc_laplace_cdf: lohi needs symbols");
  if (rows == 0) return 0;
  dim3 grid((unsigned)((rows + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const double* lg = nullptr;
  { int rc = get_log2_table(&lg); if (rc) return rc; }
  if (ncols <= 4)
    hipLaunchKernelGGL(laplace_cdf_kernel<4>, grid, block, 0, s, loc, scale, seg_min, seg_max, rows, seg_rows, ncols,
                       likelihood_bound, symbols, cdf_lower, lohi, lg);
  else if (ncols <= 8)
    hipLaunchKernelGGL(laplace_cdf_kernel<8>, grid, block, 0, s, loc, scale, seg_min, seg_max, rows, seg_rows, ncols,
                       likelihood_bound, symbols, cdf_lower, lohi, lg);
  else if (ncols <= 16)
    hipLaunchKernelGGL(laplace_cdf_kernel<16>, grid, block, 0, s, loc, scale, seg_min, seg_max, rows, seg_rows, ncols,
                       likelihood_bound, symbols, cdf_lower, lohi, lg);
  else
    hipLaunchKernelGGL(laplace_cdf_kernel<32>, grid, block, 0, s, loc, scale, seg_min, seg_max, rows, seg_rows, ncols,
                       likelihood_bound, symbols, cdf_lower, lohi, lg);
  return launch_ok("laplace_cdf_kernel");
}

int pcgc_factorized_likelihood(const float* z, const float* params, const float* noise, float* values, float* likelihood,
                               int64_t n, int C, float likelihood_bound, pcgc_stream_t stream) {
  if (n == 0) return 0;
  PCGC_REQUIRE(z && params && C > 0 && C <= kMaxFactorizedC && n % C == 0, "pcgc_factorized_likelihood: bad arguments (C=%d)", C);
  if (n == 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(factorized_likelihood_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, params, noise,
                     values, likelihood, n, C, likelihood_bound);
  return launch_ok("factorized_likelihood_kernel");
}

int pcgc_factorized_pmf(const float* params, int C, int min_v, int max_v, float likelihood_bound, float* pmf,
                        pcgc_stream_t stream) {
  PCGC_REQUIRE(params && pmf && C > 0 && max_v >= min_v, "pcgc_factorized_pmf: bad arguments");
  const int N = max_v - min_v + 1;
  hipLaunchKernelGGL(factorized_pmf_kernel, dim3((C * N + 63) / 64), dim3(64), 0, (hipStream_t)stream, params, C, min_v, N,
                     likelihood_bound, pmf);
  return launch_ok("factorized_pmf_kernel");
}

}  // extern "C"
