// libpcgc_host.so — the sequential host tail of the codec (plain C++17, no HIP).
//
//   * 32-bit range coder with 16-bit renormalisation and carry delay, and the
//     pmf -> quantised CDF step: the published algorithm of tensorflow 1.13.1
//     tensorflow/contrib/coder/kernels/{range_coder.cc, pmf_to_cdf_op.cc}, which the
//     reference calls through coder_ops at models/entropy_model.py:218,258,298 and
//     models/conditional_entropy_model.py:122,161,195 (TF is not vendored in the
//     reference tree; spec in SURVEY.md §8a row a12).
//   * batched forms: cubes are independent streams (transform.py:157-168,
//     238-248 loops over them one by one); each stream stays sequential, streams
//     run on a small thread pool.
//   * partition of a point cloud into cubes (dataprocess/inout_points.py:50-90).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <condition_variable>
#include <mutex>
#include <pthread.h>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/pcgc.h"
#include "repro_math.h"

namespace {

thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------- range encoder
struct Sink {
  uint8_t* out;
  int64_t cap;
  int64_t len = 0;   // keeps counting past cap
  inline void put(uint8_t b) {
    if (len < cap) out[len] = b;
    ++len;
  }
  inline void fill(uint64_t n, uint8_t b) {
    for (uint64_t i = 0; i < n; ++i) put(b);
  }
};

struct RangeEncoder {
  uint32_t base = 0;
  uint32_t size_minus1 = std::numeric_limits<uint32_t>::max();
  uint64_t delay = 0;

  inline void encode(uint32_t lower, uint32_t upper, int precision, Sink& sink) {
    const uint64_t size = static_cast<uint64_t>(size_minus1) + 1;
    const uint32_t a = static_cast<uint32_t>((size * lower) >> precision);
    const uint32_t b = static_cast<uint32_t>(((size * upper) >> precision) - 1);
    base += a;
    size_minus1 = b - a;
    const bool base_overflow = base < a;
    if (static_cast<uint32_t>(base + size_minus1) < base) {
      // the interval straddles 2^32: the carry is still undecided
      if ((size_minus1 >> 16) == 0) {
        base <<= 16;
        size_minus1 = (size_minus1 << 16) | 0xFFFF;
        delay += 0x20000;
      }
      return;
    }
    if (delay != 0) {
      if (base_overflow) {
        sink.put(static_cast<uint8_t>(delay >> 8));
        sink.put(static_cast<uint8_t>(delay));
        sink.fill(delay >> 16, 0x00);
      } else {
        --delay;
        sink.put(static_cast<uint8_t>(delay >> 8));
        sink.put(static_cast<uint8_t>(delay));
        sink.fill(delay >> 16, 0xFF);
      }
      delay = 0;
    }
    if ((size_minus1 >> 16) == 0) {
      const uint32_t top = base >> 16;
      base <<= 16;
      size_minus1 = (size_minus1 << 16) | 0xFFFF;
      if (base <= static_cast<uint32_t>(base + size_minus1)) {
        sink.put(static_cast<uint8_t>(top >> 8));
        sink.put(static_cast<uint8_t>(top));
      } else {
        delay = static_cast<uint64_t>(top) + 1;
      }
    }
  }

  inline void finalize(Sink& sink) {
    if (delay != 0) {
      sink.put(static_cast<uint8_t>(delay >> 8));
      if ((delay & 0xFF) != 0) sink.put(static_cast<uint8_t>(delay));
    } else if (base != 0) {
      const uint32_t mid = ((base - 1) >> 16) + 1;
      sink.put(static_cast<uint8_t>(mid >> 8));
      if ((mid & 0xFF) != 0) sink.put(static_cast<uint8_t>(mid));
    }
  }
};

// ---------------------------------------------------------------- range decoder
struct RangeDecoder {
  uint32_t base = 0;
  uint32_t size_minus1 = std::numeric_limits<uint32_t>::max();
  uint32_t value = 0;
  const uint8_t* cur;
  const uint8_t* end;

  RangeDecoder(const uint8_t* s, int64_t n) : cur(s), end(s + n) {
    read16();
    read16();
  }
  inline void read16() {
    value <<= 8;
    if (cur != end) value |= *cur++;
    value <<= 8;
    if (cur != end) value |= *cur++;
  }
  // cdf(k) for k in [0, n]; returns the symbol or -1
  template <typename CdfAt>
  inline int decode(int n, int precision, CdfAt cdf) {
    const uint64_t size = static_cast<uint64_t>(size_minus1) + 1;
    const uint64_t offset = ((static_cast<uint64_t>(static_cast<uint32_t>(value - base)) + 1) << precision) - 1;
    // smallest k in [1, n] with size * cdf(k) > offset
    int lo = 1;
    if (n <= 16) {
      // a CDF is non-decreasing, so the predicate holds for a prefix of k: count it, without branches (the latents' rows
      // have 3 .. 8 symbols; the binary search below mispredicts about once per symbol on them)
      for (int k = 1; k <= n; ++k) lo += (size * static_cast<uint64_t>(cdf(k)) <= offset) ? 1 : 0;
    } else {
      int len = n;
      while (len > 0) {
        const int half = len / 2;
        const int mid = lo + half;
        if (size * static_cast<uint64_t>(cdf(mid)) <= offset) {
          lo = mid + 1;
          len -= half + 1;
        } else {
          len = half;
        }
      }
    }
    if (lo > n) return -1;
    const uint32_t a = static_cast<uint32_t>((size * static_cast<uint64_t>(cdf(lo - 1))) >> precision);
    const uint32_t b = static_cast<uint32_t>(((size * static_cast<uint64_t>(cdf(lo))) >> precision) - 1);
    base += a;
    size_minus1 = b - a;
    if ((size_minus1 >> 16) == 0) {
      base <<= 16;
      size_minus1 = (size_minus1 << 16) | 0xFFFF;
      read16();
    }
    return lo - 1;
  }
};

// ---------------------------------------------------------------- pmf -> cdf
struct Item {
  int idx;
  double mass;
  double key;
};

// log2(v) for every integer the quantiser can reach: the greedy correction calls it twice per step and a
// truncated support can need tens of thousands of steps per row.
const double* log2_table() {
  static const std::vector<double> t = [] {
    std::vector<double> v(65538);
    v[0] = 0.0;
    for (int i = 1; i < 65538; ++i) v[i] = std::log2(double(i));
    return v;
  }();
  return t.data();
}

int pmf_to_cdf_row(const float* pmf, int n, int precision, int32_t* cdf) {
  const double* lg = log2_table();
  const int32_t normalizer = int32_t(1) << precision;
  int32_t* v = cdf + 1;
  int64_t sum = 0;
  for (int i = 0; i < n; ++i) {
    int32_t q = static_cast<int32_t>(std::rint(pmf[i] * static_cast<float>(normalizer)));
    q = std::max(q, 1);
    v[i] = q;
    sum += q;
  }
  if (sum != normalizer) {
    const bool shrink = sum > normalizer;
    auto key_of = [&](int i) -> double {
      const double m = pmf[i];
      if (shrink) return v[i] <= 1 ? std::numeric_limits<double>::infinity() : m * (lg[v[i]] - lg[v[i] - 1]);
      return m * (lg[v[i] + 1] - lg[v[i]]);
    };
    // "worse" = later in TF's queue: larger penalty when shrinking, smaller gain when growing
    auto before = [&](double a, double b) { return shrink ? a < b : a > b; };
    std::vector<Item> q(n);
    for (int i = 0; i < n; ++i) q[i] = {i, double(pmf[i]), key_of(i)};
    std::stable_sort(q.begin(), q.end(), [&](const Item& a, const Item& b) { return before(a.key, b.key); });
    while (sum != normalizer) {
      Item head = q[0];
      if (shrink) {
        if (v[head.idx] <= 1) { set_error("pmf_to_quantized_cdf: cannot shrink below 1"); return -2; }
        --v[head.idx];
        --sum;
      } else {
        ++v[head.idx];
        ++sum;
      }
      head.key = key_of(head.idx);
      int j = 1;
      while (j < n && !before(head.key, q[j].key)) { q[j - 1] = q[j]; ++j; }
      q[j - 1] = head;
    }
  }
  cdf[0] = 0;
  int32_t acc = 0;
  for (int i = 0; i < n; ++i) { acc += v[i]; v[i] = acc; }
  return 0;
}

// Persistent worker pool: the per-cube streams are coded by long-lived threads (creating 32+ std::threads per call
// cost more than the coding itself).  One job at a time; the caller takes part; work items are claimed dynamically.
class Pool {
 public:
  // kPools independent pools: the host pipelines of transform.py code their groups at the same time, each on its own
  // pool (a caller takes the first idle pool, or queues on the one its thread id hashes to)
  static constexpr int kPools = 4;
  static Pool& get() {
    Pool* pools;
    {
      static std::mutex create_mu;
      std::lock_guard<std::mutex> g(create_mu);
      if (!instance()) {
        static bool hooked = false;
        if (!hooked) {                     // a forked child has none of the workers: start over with fresh pools
          pthread_atfork(nullptr, nullptr, [] { instance() = nullptr; });
          hooked = true;
        }
        instance() = new Pool[kPools];     // leaked on purpose: detached workers outlive static destruction
      }
      pools = instance();
    }
    for (int i = 0; i < kPools; ++i)
      if (pools[i].job_mu_.try_lock()) return pools[i];
    Pool& p = pools[std::hash<std::thread::id>()(std::this_thread::get_id()) % kPools];
    p.job_mu_.lock();
    return p;                              // returned with job_mu_ held; run() releases it
  }
  static Pool*& instance() {
    static Pool* p = nullptr;
    return p;
  }
  template <typename F>
  void run(int n, int n_threads, F& f) {
    std::lock_guard<std::mutex> job_guard(job_mu_, std::adopt_lock);
    n_threads = std::min(std::min(n_threads, n), kMaxThreads);
    grow(n_threads - 1);
    next_.store(0);
    n_ = n;
    fn_ = [](void* ctx, int i) { (*static_cast<F*>(ctx))(i); };
    ctx_ = &f;
    {
      std::lock_guard<std::mutex> g(mu_);
      want_ = n_threads - 1;
      pending_ = n_threads - 1;
      ++gen_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> g(mu_);
    done_cv_.wait(g, [&] { return pending_ == 0; });
  }

 private:
  static constexpr int kMaxThreads = 256;
  void work() {
    for (int i = next_.fetch_add(1); i < n_; i = next_.fetch_add(1)) fn_(ctx_, i);
  }
  void grow(int workers) {
    while ((int)threads_.size() < workers) {
      const int id = (int)threads_.size();
      threads_.emplace_back([this, id] {
        uint64_t seen = 0;
        for (;;) {
          {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return gen_ != seen; });
            seen = gen_;
            if (id >= want_) continue;            // not needed for this job
          }
          work();
          {
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_cv_.notify_one();
          }
        }
      });
      threads_.back().detach();
    }
  }
  std::mutex job_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> threads_;
  std::atomic<int> next_{0};
  int n_ = 0, want_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
  void (*fn_)(void*, int) = nullptr;
  void* ctx_ = nullptr;
};

template <typename F>
void parallel_for(int n, int n_threads, F f) {
  if (n_threads <= 1 || n <= 1) {
    for (int i = 0; i < n; ++i) f(i);
    return;
  }
  Pool::get().run(n, n_threads, f);
}

template <typename T>
int range_encode_values(const T* data, int64_t rows, int cols, int offset, const int32_t* cdf, int n, int broadcast_rows,
                               int precision, uint8_t* out, int64_t cap, int64_t* out_len) {
  Sink sink{out, out ? cap : 0};
  RangeEncoder enc;
  for (int64_t r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const int32_t* row = cdf + ((broadcast_rows ? 0 : r * cols) + c) * int64_t(n + 1);
      const int v = int(data[r * cols + c]) - offset;
      if (v < 0 || v >= n) { set_error("pcgc_range_encode_values: symbol %d outside [0,%d) at (%lld,%d)", v, n, (long long)r, c); return -1; }
      enc.encode(uint32_t(row[v]), uint32_t(row[v + 1]), precision, sink);
    }
  enc.finalize(sink);
  *out_len = sink.len;
  if (sink.len > sink.cap) { set_error("pcgc_range_encode_values: output needs %lld bytes, capacity %lld", (long long)sink.len, (long long)cap); return -2; }
  return 0;
}


}  // namespace

extern "C" {

const char* pcgc_host_last_error(void) { return g_err; }

int pcgc_pmf_to_quantized_cdf(const float* pmf, int64_t rows, int n, int precision, int32_t* cdf) {
  if (!pmf || !cdf || n < 1 || precision < 1 || precision > 16) { set_error("pcgc_pmf_to_quantized_cdf: bad arguments"); return -1; }
  for (int64_t r = 0; r < rows; ++r) {
    int rc = pmf_to_cdf_row(pmf + r * n, n, precision, cdf + r * (n + 1));
    if (rc) return rc;
  }
  return 0;
}

int pcgc_range_encode(const int16_t* data, int64_t rows, int cols, const int32_t* cdf, int n, int broadcast_rows,
                      int precision, uint8_t* out, int64_t cap, int64_t* out_len) {
  if (!out_len || (rows * cols > 0 && (!data || !cdf))) { set_error("pcgc_range_encode: NULL argument"); return -1; }
  Sink sink{out, out ? cap : 0};
  RangeEncoder enc;
  for (int64_t r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c) {
      const int32_t* row = cdf + ((broadcast_rows ? 0 : r * cols) + c) * int64_t(n + 1);
      const int v = data[r * cols + c];
      if (v < 0 || v >= n) { set_error("pcgc_range_encode: symbol %d outside [0,%d) at (%lld,%d)", v, n, (long long)r, c); return -1; }
      enc.encode(uint32_t(row[v]), uint32_t(row[v + 1]), precision, sink);
    }
  enc.finalize(sink);
  *out_len = sink.len;
  if (sink.len > sink.cap) { set_error("pcgc_range_encode: output needs %lld bytes, capacity %lld", (long long)sink.len, (long long)cap); return -2; }
  return 0;
}

static int range_decode_impl(const uint8_t* str, int64_t len, int64_t rows, int cols, const int32_t* cdf, int n,
                             int broadcast_rows, int precision, int16_t* out, int64_t* progress) {
  if ((rows * cols > 0 && (!cdf || !out)) || (len > 0 && !str)) { set_error("pcgc_range_decode: NULL argument"); return -1; }
  RangeDecoder dec(str, len);
  for (int64_t r = 0; r < rows; ++r) {
    for (int c = 0; c < cols; ++c) {
      const int32_t* row = cdf + ((broadcast_rows ? 0 : r * cols) + c) * int64_t(n + 1);
      const int s = dec.decode(n, precision, [row](int k) { return uint32_t(row[k]); });
      if (s < 0) {
        set_error("pcgc_range_decode: corrupt stream at (%lld,%d)", (long long)r, c);
        if (progress) __atomic_store_n(progress, int64_t(-1), __ATOMIC_RELEASE);
        return -3;
      }
      out[r * cols + c] = int16_t(s);
    }
    if (progress && ((r & 1023) == 1023 || r + 1 == rows)) __atomic_store_n(progress, r + 1, __ATOMIC_RELEASE);
  }
  return 0;
}

int pcgc_range_encode_values(const void* data, int elem_bytes, int64_t rows, int cols, int offset, const int32_t* cdf, int n,
                             int broadcast_rows, int precision, uint8_t* out, int64_t cap, int64_t* out_len) {
  if (!out_len || (rows * cols > 0 && (!data || !cdf)) || (elem_bytes != 1 && elem_bytes != 2)) {
    set_error("pcgc_range_encode_values: bad argument");
    return -1;
  }
  if (elem_bytes == 1)
    return range_encode_values(static_cast<const int8_t*>(data), rows, cols, offset, cdf, n, broadcast_rows, precision, out, cap, out_len);
  return range_encode_values(static_cast<const int16_t*>(data), rows, cols, offset, cdf, n, broadcast_rows, precision, out, cap, out_len);
}

int pcgc_range_decode(const uint8_t* str, int64_t len, int64_t rows, int cols, const int32_t* cdf, int n,
                      int broadcast_rows, int precision, int16_t* out) {
  return range_decode_impl(str, len, rows, cols, cdf, n, broadcast_rows, precision, out, nullptr);
}

int pcgc_range_decode_progress(const uint8_t* str, int64_t len, int64_t rows, int cols, const int32_t* cdf, int n,
                               int broadcast_rows, int precision, int16_t* out, int64_t* progress) {
  if (!progress) { set_error("pcgc_range_decode_progress: progress is NULL"); return -1; }
  return range_decode_impl(str, len, rows, cols, cdf, n, broadcast_rows, precision, out, progress);
}

int pcgc_range_encode_lohi_batch(const uint32_t* lohi, int n_streams, int64_t sym_per_stream, int precision,
                                 uint8_t* out, int64_t cap_per_stream, int64_t* out_lens, int n_threads) {
  if (n_streams < 0 || (n_streams > 0 && (!lohi || !out || !out_lens))) { set_error("pcgc_range_encode_lohi_batch: NULL argument"); return -1; }
  std::atomic<int> bad(0);
  parallel_for(n_streams, n_threads, [&](int sidx) {
    Sink sink{out + int64_t(sidx) * cap_per_stream, cap_per_stream};
    RangeEncoder enc;
    const uint32_t* p = lohi + int64_t(sidx) * sym_per_stream;
    for (int64_t i = 0; i < sym_per_stream; ++i) {
      const uint32_t w = p[i];
      enc.encode(w & 0xFFFFu, (w >> 16) + 1u, precision, sink);
    }
    enc.finalize(sink);
    out_lens[sidx] = sink.len;
    if (sink.len > cap_per_stream) bad.store(1);
  });
  if (bad.load()) { set_error("pcgc_range_encode_lohi_batch: a stream exceeded cap_per_stream=%lld", (long long)cap_per_stream); return -2; }
  return 0;
}

int pcgc_range_decode_u16_batch(const uint8_t* strings, const int64_t* offsets, const int64_t* lens, int n_streams,
                                int64_t sym_per_stream, const uint16_t* cdf_lower, int ncols, const int32_t* n_sym,
                                int precision, int16_t* out, int n_threads) {
  if (n_streams < 0 || (n_streams > 0 && (!offsets || !lens || !cdf_lower || !n_sym || !out))) { set_error("pcgc_range_decode_u16_batch: NULL argument"); return -1; }
  std::atomic<int> bad(0);
  const uint32_t top = 1u << precision;
  parallel_for(n_streams, n_threads, [&](int sidx) {
    RangeDecoder dec(strings + offsets[sidx], lens[sidx]);
    const int n = n_sym[sidx];
    if (n < 1 || n > ncols) { bad.store(1); return; }
    const uint16_t* rows = cdf_lower + int64_t(sidx) * sym_per_stream * ncols;
    int16_t* o = out + int64_t(sidx) * sym_per_stream;
    for (int64_t i = 0; i < sym_per_stream; ++i) {
      const uint16_t* row = rows + i * ncols;
      const int s = dec.decode(n, precision, [row, n, top](int k) { return k >= n ? top : uint32_t(row[k]); });
      if (s < 0) { bad.store(2); return; }
      o[i] = int16_t(s);
    }
  });
  if (bad.load()) { set_error("pcgc_range_decode_u16_batch: %s", bad.load() == 1 ? "n_sym outside [1,ncols]" : "corrupt stream"); return -3; }
  return 0;
}

// ---------------------------------------------------------------- partition
int pcgc_partition(const int32_t* points, int64_t n, int cube_size, int min_num, int64_t* n_cubes,
                   int64_t* cube_positions, int64_t* sorted_positions, int32_t* cube_of_point) {
  if (!n_cubes || (n > 0 && !points) || cube_size <= 0) { set_error("pcgc_partition: bad arguments"); return -1; }
  auto fdiv = [cube_size](int32_t v) -> int64_t { return v >= 0 ? v / cube_size : -((-(int64_t)v + cube_size - 1) / cube_size); };
  struct Key { int64_t x, y, z; bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; } };
  struct H { size_t operator()(const Key& k) const { return size_t(k.x * 73856093LL ^ k.y * 19349663LL ^ k.z * 83492791LL); } };
  std::vector<Key> keys;
  std::vector<int64_t> counts;
  // scratch that survives the call (per calling thread): fresh 13 MB vectors cost their zero fill and page faults every time
  static thread_local std::vector<int> ord_buf;
  static thread_local std::vector<int32_t> kc_buf;
  if ((int64_t)ord_buf.size() < n) ord_buf.resize(size_t(n));
  if ((int64_t)kc_buf.size() < n * 3) kc_buf.resize(size_t(n) * 3);
  // ... but not without bound: scratch above 4 M points (64 MB; a vox10 cloud has 0.8 M, a vox12 cloud tens of millions)
  // is given back when the call returns instead of staying with the calling thread for the life of the process
  struct Release {
    std::vector<int>& a; std::vector<int32_t>& b;
    ~Release() {
      if (a.size() > (size_t(4) << 20)) { std::vector<int>().swap(a); std::vector<int32_t>().swap(b); }
    }
  } release{ord_buf, kc_buf};
  int* ord = ord_buf.data();
  // cube coordinates of every point and their bounding box; a cloud on a 1024^3 grid has 16^3 candidate cubes, so the
  // cube -> first-appearance ordinal map is a dense array (a hash map only for pathological extents)
  int32_t* kc = kc_buf.data();
  int64_t lo[3] = {std::numeric_limits<int64_t>::max(), std::numeric_limits<int64_t>::max(), std::numeric_limits<int64_t>::max()};
  int64_t hi[3] = {std::numeric_limits<int64_t>::min(), std::numeric_limits<int64_t>::min(), std::numeric_limits<int64_t>::min()};
  // floor division of every coordinate: slices of the cloud on the worker pool (it is 2.5 M divisions for a vox10 cloud),
  // an arithmetic shift where the cube size is a power of two (the usual 64)
  const int T = int(std::max<int64_t>(1, std::min<int64_t>(16, n / 65536)));
  const int shift = (cube_size & (cube_size - 1)) == 0 ? __builtin_ctz((unsigned)cube_size) : -1;
  std::vector<int64_t> tlo(size_t(T) * 3, std::numeric_limits<int64_t>::max()), thi(size_t(T) * 3, std::numeric_limits<int64_t>::min());
  parallel_for(T, T, [&](int t) {
    const int64_t i0 = n * t / T, i1 = n * (t + 1) / T;
    int64_t l[3] = {tlo[0], tlo[0], tlo[0]}, h[3] = {thi[0], thi[0], thi[0]};
    l[0] = l[1] = l[2] = std::numeric_limits<int64_t>::max();
    h[0] = h[1] = h[2] = std::numeric_limits<int64_t>::min();
    for (int64_t i = i0; i < i1; ++i)
      for (int a = 0; a < 3; ++a) {
        const int32_t v = points[i * 3 + a];
        const int64_t c = shift >= 0 ? int64_t(v >> shift) : fdiv(v);
        kc[i * 3 + a] = int32_t(c);
        l[a] = std::min(l[a], c);
        h[a] = std::max(h[a], c);
      }
    for (int a = 0; a < 3; ++a) { tlo[size_t(t) * 3 + a] = l[a]; thi[size_t(t) * 3 + a] = h[a]; }
  });
  for (int t = 0; t < T; ++t)
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], tlo[size_t(t) * 3 + a]); hi[a] = std::max(hi[a], thi[size_t(t) * 3 + a]); }
  const int64_t ex = n ? hi[0] - lo[0] + 1 : 0, ey = n ? hi[1] - lo[1] + 1 : 0, ez = n ? hi[2] - lo[2] + 1 : 0;
  if (n && ex * ey * ez <= (int64_t(1) << 16) && T > 1) {
    // small extents (a vox10 cloud: 16^3 candidate cubes), large clouds: every worker counts its slice of the points into
    // a dense table of its own and notes where each cube first appears; the slices are in point order, so the global
    // first-appearance order is the cubes sorted by their first point
    const size_t E = size_t(ex * ey * ez);
    std::vector<int64_t> first(size_t(T) * E, std::numeric_limits<int64_t>::max()), cnt(size_t(T) * E, 0);
    parallel_for(T, T, [&](int t) {
      int64_t* f = first.data() + size_t(t) * E;
      int64_t* c = cnt.data() + size_t(t) * E;
      for (int64_t i = n * t / T, i1 = n * (t + 1) / T; i < i1; ++i) {
        const size_t idx = size_t(((kc[i * 3] - lo[0]) * ey + (kc[i * 3 + 1] - lo[1])) * ez + (kc[i * 3 + 2] - lo[2]));
        if (c[idx]++ == 0) f[idx] = i;
        ord[i] = int(idx);                                   // the dense index for now
      }
    });
    std::vector<std::pair<int64_t, size_t>> seen;            // (first point, dense index)
    std::vector<int64_t> total(E, 0);
    for (size_t idx = 0; idx < E; ++idx) {
      int64_t f0 = std::numeric_limits<int64_t>::max();
      for (int t = 0; t < T; ++t) { total[idx] += cnt[size_t(t) * E + idx]; f0 = std::min(f0, first[size_t(t) * E + idx]); }
      if (total[idx]) seen.push_back({f0, idx});
    }
    std::sort(seen.begin(), seen.end());
    std::vector<int> slot_of(E, -1);
    for (size_t o = 0; o < seen.size(); ++o) {
      const size_t idx = seen[o].second;
      slot_of[idx] = int(o);
      const int64_t z = int64_t(idx % size_t(ez)), y = int64_t((idx / size_t(ez)) % size_t(ey)), x = int64_t(idx / size_t(ez * ey));
      keys.push_back(Key{x + lo[0], y + lo[1], z + lo[2]});
      counts.push_back(total[idx]);
    }
    parallel_for(T, T, [&](int t) {
      for (int64_t i = n * t / T, i1 = n * (t + 1) / T; i < i1; ++i) ord[i] = slot_of[size_t(ord[i])];
    });
  } else if (n && ex * ey * ez <= (int64_t(1) << 22)) {
    std::vector<int> dense(size_t(ex * ey * ez), -1);
    for (int64_t i = 0; i < n; ++i) {
      int& slot = dense[size_t(((kc[i * 3] - lo[0]) * ey + (kc[i * 3 + 1] - lo[1])) * ez + (kc[i * 3 + 2] - lo[2]))];
      if (slot < 0) { slot = int(keys.size()); keys.push_back(Key{kc[i * 3], kc[i * 3 + 1], kc[i * 3 + 2]}); counts.push_back(0); }
      ord[i] = slot;
      ++counts[slot];
    }
  } else {
    std::unordered_map<Key, int, H> index;   // cube -> first-appearance ordinal
    for (int64_t i = 0; i < n; ++i) {
      Key k{kc[i * 3], kc[i * 3 + 1], kc[i * 3 + 2]};
      auto it = index.find(k);
      int o;
      if (it == index.end()) { o = int(keys.size()); index.emplace(k, o); keys.push_back(k); counts.push_back(0); }
      else o = it->second;
      ord[i] = o;
      ++counts[o];
    }
  }
  // the reference counts a single-point cube as "3" (a 1-D array's shape[0], inout_points.py:66,72);
  // that only matters for min_num <= 3 and then crashes later, so plain counts are used.
  std::vector<int> kept;
  for (size_t o = 0; o < keys.size(); ++o) if (counts[o] >= min_num) kept.push_back(int(o));
  *n_cubes = int64_t(kept.size());
  if (!cube_positions && !sorted_positions && !cube_of_point) return 0;
  if (kept.empty()) { set_error("pcgc_partition: no cube holds at least min_num=%d points", min_num); return -4; }
  int64_t mx = std::numeric_limits<int64_t>::min();
  for (int o : kept) mx = std::max({mx, keys[o].x, keys[o].y, keys[o].z});
  const int64_t step = mx + 1;
  std::vector<std::pair<int64_t, int>> order;   // (x + y*step + z*step^2, ordinal)
  for (int o : kept) order.push_back({keys[o].x + keys[o].y * step + keys[o].z * step * step, o});
  std::sort(order.begin(), order.end());
  std::vector<int> sorted_slot(keys.size(), -1);
  for (size_t s = 0; s < order.size(); ++s) sorted_slot[order[s].second] = int(s);
  if (cube_positions)
    for (size_t i = 0; i < kept.size(); ++i) {
      cube_positions[i * 3] = keys[kept[i]].x; cube_positions[i * 3 + 1] = keys[kept[i]].y; cube_positions[i * 3 + 2] = keys[kept[i]].z;
    }
  if (sorted_positions)
    for (size_t s = 0; s < order.size(); ++s) {
      // the reference re-derives positions from the key (inout_points.py:83-86): x = n % step, ...
      const int64_t key = order[s].first;
      auto pmod = [](int64_t a, int64_t m) { int64_t r = a % m; return r < 0 ? r + m : r; };
      auto pdiv = [](int64_t a, int64_t m) { int64_t q = a / m; return (a % m != 0 && ((a < 0) != (m < 0))) ? q - 1 : q; };
      sorted_positions[s * 3] = pmod(key, step);
      sorted_positions[s * 3 + 1] = pmod(pdiv(key, step), step);
      sorted_positions[s * 3 + 2] = pdiv(pdiv(key, step), step);
    }
  if (cube_of_point)
    parallel_for(T, T, [&](int t) {
      for (int64_t i = n * t / T, i1 = n * (t + 1) / T; i < i1; ++i) cube_of_point[i] = sorted_slot[ord[i]];
    });
  return 0;
}

// ---------------------------------------------------------------- ply text (reader)
// One token the way Python's float() reads it after str.split(' '): optional surrounding whitespace, then a decimal
// number.  Plain "[+-]digits[.digits]" is converted by hand (exact for the coordinate magnitudes a ply holds, < 2^53
// with at most 15 significant digits: the integer mantissa and the power of ten are both exact doubles and one
// division is correctly rounded); anything else goes through strtod.  Returns false where float() raises ValueError.
static bool parse_float_token(const char* b, const char* e, double* out) {
  while (b < e && (*b == ' ' || (*b >= 9 && *b <= 13))) ++b;
  while (e > b && (e[-1] == ' ' || (e[-1] >= 9 && e[-1] <= 13))) --e;
  if (b == e) return false;
  const char* p = b;
  bool neg = false;
  if (*p == '+' || *p == '-') { neg = *p == '-'; ++p; }
  uint64_t mant = 0;
  int digits = 0, frac = 0;
  bool dot = false, simple = p < e;
  for (; p < e; ++p) {
    if (*p >= '0' && *p <= '9') {
      if (++digits > 15) { simple = false; break; }
      mant = mant * 10 + uint64_t(*p - '0');
      if (dot) ++frac;
    } else if (*p == '.' && !dot) {
      dot = true;
    } else {
      simple = false;
      break;
    }
  }
  if (simple && digits > 0) {
    static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    const double v = double(mant) / p10[frac];
    *out = neg ? -v : v;
    return true;
  }
  char buf[64];
  if (size_t(e - b) >= sizeof(buf)) return false;
  size_t n = 0;
  for (const char* q = b; q < e; ++q) {                    // float() allows single underscores between digits
    if (*q == '_') {
      const bool between = q > b && q + 1 < e && q[-1] >= '0' && q[-1] <= '9' && q[1] >= '0' && q[1] <= '9';
      if (!between) return false;
      continue;
    }
    buf[n++] = *q;
  }
  buf[n] = 0;
  for (size_t i = 0; i < n; ++i)
    if (buf[i] == 'x' || buf[i] == 'X' || buf[i] == 'p' || buf[i] == 'P') return false;   // strtod takes hex floats, float() does not
  char* endp = nullptr;
  const double v = std::strtod(buf, &endp);
  if (endp != buf + n) return false;
  *out = v;
  return true;
}

int pcgc_parse_ply_points(const char* text, int64_t len, int32_t* out, int64_t cap, int64_t* n_points, int n_threads) {
  if ((len > 0 && !text) || !n_points || (cap > 0 && !out)) { set_error("pcgc_parse_ply_points: NULL argument"); return -1; }
  n_threads = std::max(1, std::min(n_threads, 64));
  // chunk boundaries at line starts
  std::vector<int64_t> cut(size_t(n_threads) + 1, len);
  cut[0] = 0;
  for (int t = 1; t < n_threads; ++t) {
    int64_t p = len * t / n_threads;
    if (p < cut[t - 1]) p = cut[t - 1];
    while (p < len && text[p] != '\n') ++p;
    cut[t] = p < len ? p + 1 : len;
  }
  std::vector<std::vector<int32_t>> part(static_cast<size_t>(n_threads));
  parallel_for(n_threads, n_threads, [&](int t) {
    std::vector<int32_t>& v = part[t];
    v.reserve(size_t((cut[t + 1] - cut[t]) / 8));
    const char* p = text + cut[t];
    const char* end = text + cut[t + 1];
    while (p < end) {
      const char* nl = static_cast<const char*>(std::memchr(p, '\n', size_t(end - p)));
      const char* le = nl ? nl + 1 : end;                  // the line incl. its newline, like Python's iteration
      const char* tok = p;
      double xyz[3];
      int k = 0;
      bool ok = true;
      for (; k < 3 && ok; ++k) {
        const char* sp = static_cast<const char*>(std::memchr(tok, ' ', size_t(le - tok)));
        const char* te = sp ? sp : le;
        ok = parse_float_token(tok, te, &xyz[k]);
        if (!sp && k < 2) { ok = false; }                  // fewer than three tokens
        tok = sp ? sp + 1 : le;
      }
      if (ok && k == 3) {
        v.push_back(int32_t(xyz[0]));
        v.push_back(int32_t(xyz[1]));
        v.push_back(int32_t(xyz[2]));
      }
      p = le;
    }
  });
  int64_t total = 0;
  for (auto& v : part) total += int64_t(v.size() / 3);
  *n_points = total;
  if (total > cap) { set_error("pcgc_parse_ply_points: %lld points, buffer holds %lld", (long long)total, (long long)cap); return -2; }
  std::vector<int64_t> off(size_t(n_threads) + 1, 0);
  for (int t = 0; t < n_threads; ++t) off[size_t(t) + 1] = off[size_t(t)] + int64_t(part[size_t(t)].size());
  parallel_for(n_threads, n_threads, [&](int t) {          // every thread's points to their final place (file order)
    const std::vector<int32_t>& v = part[size_t(t)];
    if (!v.empty()) std::memcpy(out + off[size_t(t)], v.data(), v.size() * sizeof(int32_t));
  });
  return 0;
}

// ---------------------------------------------------------------- ply text
int pcgc_format_points_int(const int64_t* pts, int64_t n, char* out, int64_t cap, int64_t* out_len) {
  if ((n > 0 && !pts) || !out_len) { set_error("pcgc_format_points_int: NULL argument"); return -1; }
  // two passes over blocks of points: the exact text length of every block (digits + sign + separator per number), then
  // every block written at its final offset — no worst-case buffer, nothing moved afterwards
  const int n_blocks = int(std::max<int64_t>(1, std::min<int64_t>(32, n / 16384)));
  auto ndigits = [](uint64_t u) { int d = 1; while (u >= 10) { u /= 10; ++d; } return d; };
  std::vector<int64_t> len(size_t(n_blocks), 0);
  parallel_for(n_blocks, n_blocks, [&](int t) {
    int64_t l = 0;
    for (int64_t i = 3 * (n * t / n_blocks); i < 3 * (n * (t + 1) / n_blocks); ++i) {
      const int64_t v = pts[i];
      l += ndigits(v < 0 ? 0 - (uint64_t)v : (uint64_t)v) + (v < 0 ? 2 : 1);
    }
    len[size_t(t)] = l;
  });
  std::vector<int64_t> off(size_t(n_blocks) + 1, 0);
  for (int t = 0; t < n_blocks; ++t) off[size_t(t) + 1] = off[size_t(t)] + len[size_t(t)];
  const int64_t total = off[size_t(n_blocks)];
  *out_len = total;
  if (!out || cap < total) {
    set_error("pcgc_format_points_int: buffer of %lld bytes for %lld points (need %lld)", (long long)cap, (long long)n, (long long)total);
    return -2;
  }
  parallel_for(n_blocks, n_blocks, [&](int t) {
    char* p = out + off[size_t(t)];
    char tmp[24];
    for (int64_t i = 3 * (n * t / n_blocks); i < 3 * (n * (t + 1) / n_blocks); ++i) {
      const int64_t v = pts[i];
      uint64_t u = v < 0 ? 0 - (uint64_t)v : (uint64_t)v;
      if (v < 0) *p++ = '-';
      int k = 0;
      do { tmp[k++] = char('0' + u % 10); u /= 10; } while (u);
      while (k) *p++ = tmp[--k];
      *p++ = (i % 3 == 2) ? '\n' : ' ';
    }
  });
  return 0;
}

// ---------------------------------------------------------------- reproducible elementary functions (repro_math.h)
int pcgc_host_repro_eval(int fn, const float* x, float* y, int64_t n) {
  if (fn < 0 || fn > 4 || n < 0 || (n > 0 && (!x || !y))) { set_error("pcgc_host_repro_eval: bad arguments"); return -1; }
  for (int64_t i = 0; i < n; ++i) {
    const float v = x[i];
    y[i] = fn == 0 ? pcgc::repro::expf_(v) : fn == 1 ? pcgc::repro::logf_(v) : fn == 2 ? pcgc::repro::tanhf_(v)
         : fn == 3 ? pcgc::repro::sigmoidf_(v) : pcgc::repro::softplusf_(v);
  }
  return 0;
}

// ---------------------------------------------------------------- crc32c (tensor-bundle checkpoints)
uint32_t pcgc_crc32c(uint32_t crc_in, const void* data, int64_t n) {
  static uint32_t table[8][256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
  });
  const uint8_t* p = static_cast<const uint8_t*>(data);
  uint32_t c = ~crc_in;
  while (n >= 8) {                       // slicing-by-8
    uint32_t lo, hi;
    std::memcpy(&lo, p, 4);
    std::memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
        table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    p += 8; n -= 8;
  }
  while (n-- > 0) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  return ~c;
}

}  // extern "C"
