/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object.
 *
 * Plain-C restatement of the integer arithmetic behind
 *   tensorflow.contrib.coder.python.ops.coder_ops.{pmf_to_quantized_cdf,
 *   range_encode, range_decode}
 * which the reference calls at
 *   models/entropy_model.py:218,258,298 and
 *   models/conditional_entropy_model.py:122,161,195.
 *
 * The algorithm lives in a third-party dependency that is NOT vendored under
 * /root/reference: tensorflow-gpu==1.13.1 (pinned by README.md:19),
 * tensorflow/contrib/coder/kernels/{range_coder.cc, range_coder_ops.cc,
 * pmf_to_cdf_op.cc}.  TensorFlow cannot be installed here, the reference has
 * no tests or golden vectors for it, therefore:
 *
 *      *** PARITY UNPINNED ***
 *
 * What IS checked (tests/test_oracle_coder.py): encode->decode round trips,
 * structural pins from the reference's recorded byte counts (z header 12 B,
 * head = 2+B+sum(1|3)+10), CDF rows sum to 2^precision with every entry >= 1,
 * and agreement with an independent pure-Python restatement on small cases.
 *
 * Published algorithm restated (TF 1.13):
 *   pmf_to_quantized_cdf: v_i = max(1, rint(pmf_i * 2^p)); while sum > 2^p
 *     decrement the item with the smallest penalty
 *     pmf_i*(log2 v - log2(v-1)) (v>1); while sum < 2^p increment the item
 *     with the largest gain pmf_i*(log2(v+1) - log2 v).  TF keeps a sorted
 *     queue (std::sort, then after each step the head is moved behind every
 *     item that does not compare strictly worse).  std::sort's order among
 *     exactly equal keys is implementation-defined; this restatement fixes it
 *     to ascending index (a stable sort), and the product does the same.
 *   RangeEncoder/RangeDecoder: 32-bit range coder with 16-bit renormalisation
 *     and carry delay, exactly as described in SURVEY.md §8(a) row a12.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* pmf_to_quantized_cdf  (pmf_to_cdf_op.cc)                            */
/* ------------------------------------------------------------------ */

typedef struct {
  int idx;
  double mass;
  double key; /* penalty (ascending) or gain (descending) */
} q_item;

static double next_penalty(int32_t v, double mass) {
  if (v <= 1) return INFINITY;
  return mass * (log2((double)v) - log2((double)(v - 1)));
}

static double next_gain(int32_t v, double mass) {
  if (v < 1) return -INFINITY;
  return mass * (log2((double)(v + 1)) - log2((double)v));
}

/* stable insertion sort; less(a,b) = a.key < b.key (penalty) or a.key > b.key (gain) */
static void stable_sort_items(q_item* q, int n, int descending) {
  for (int i = 1; i < n; ++i) {
    q_item t = q[i];
    int j = i - 1;
    while (j >= 0 && (descending ? (t.key > q[j].key) : (t.key < q[j].key))) {
      q[j + 1] = q[j];
      --j;
    }
    q[j + 1] = t;
  }
}

/* one row: pmf[n] -> cdf[n+1]; returns 0 ok, <0 error */
int oracle_pmf_to_quantized_cdf_row(const float* pmf, int n, int precision,
                                    int32_t* cdf) {
  if (n <= 0 || n > 4096) return -1;
  const int32_t normalizer = (int32_t)1 << precision;
  int32_t* v = cdf + 1;
  int64_t sum = 0;
  for (int i = 0; i < n; ++i) {
    /* float multiply by a power of two is exact; rint = round-half-even */
    int32_t value = (int32_t)rintf(pmf[i] * (float)normalizer);
    if (value < 1) value = 1;
    v[i] = value;
    sum += value;
  }
  q_item stack_q[64];
  q_item* q = n <= 64 ? stack_q : (q_item*)malloc(sizeof(q_item) * (size_t)n);
  if (sum > normalizer) {
    for (int i = 0; i < n; ++i) {
      q[i].idx = i;
      q[i].mass = (double)pmf[i];
      q[i].key = next_penalty(v[i], q[i].mass);
    }
    stable_sort_items(q, n, 0);
    while (sum-- > normalizer) {
      if (v[q[0].idx] <= 1) { if (q != stack_q) free(q); return -2; } /* TF CHECK_GT(*pointer, 1) */
      v[q[0].idx] -= 1;
      q[0].key = next_penalty(v[q[0].idx], q[0].mass);
      /* move head behind every item that is not strictly greater */
      q_item head = q[0];
      int j = 1;
      while (j < n && !(head.key < q[j].key)) { q[j - 1] = q[j]; ++j; }
      q[j - 1] = head;
    }
  } else if (sum < normalizer) {
    for (int i = 0; i < n; ++i) {
      q[i].idx = i;
      q[i].mass = (double)pmf[i];
      q[i].key = next_gain(v[i], q[i].mass);
    }
    stable_sort_items(q, n, 1);
    while (sum++ < normalizer) {
      v[q[0].idx] += 1;
      q[0].key = next_gain(v[q[0].idx], q[0].mass);
      q_item head = q[0];
      int j = 1;
      /* GainItem operator<: lhs.gain > rhs.gain; head < q[j]  <=>  head.key > q[j].key */
      while (j < n && !(head.key > q[j].key)) { q[j - 1] = q[j]; ++j; }
      q[j - 1] = head;
    }
  }
  if (q != stack_q) free(q);
  cdf[0] = 0;
  int32_t acc = 0;
  for (int i = 0; i < n; ++i) { acc += v[i]; v[i] = acc; }
  return 0;
}

int oracle_pmf_to_quantized_cdf(const float* pmf, int64_t rows, int n,
                                int precision, int32_t* cdf) {
  for (int64_t r = 0; r < rows; ++r) {
    int rc = oracle_pmf_to_quantized_cdf_row(pmf + r * n, n, precision,
                                             cdf + r * (n + 1));
    if (rc) return rc;
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* RangeEncoder / RangeDecoder  (range_coder.cc)                       */
/* ------------------------------------------------------------------ */

typedef struct {
  uint32_t base;
  uint32_t size_minus1;
  uint64_t delay;
  uint8_t* out;
  int64_t cap;
  int64_t len; /* keeps counting past cap so the caller can detect overflow */
} enc_state;

static void put(enc_state* e, uint8_t b) {
  if (e->len < e->cap) e->out[e->len] = b;
  e->len++;
}
static void put_n(enc_state* e, uint64_t n, uint8_t b) {
  for (uint64_t i = 0; i < n; ++i) put(e, b);
}

static void enc_init(enc_state* e, uint8_t* out, int64_t cap) {
  e->base = 0; e->size_minus1 = 0xFFFFFFFFu; e->delay = 0;
  e->out = out; e->cap = cap; e->len = 0;
}

static void enc_encode(enc_state* e, int32_t lower, int32_t upper, int precision) {
  const uint64_t size = (uint64_t)e->size_minus1 + 1;
  const uint32_t a = (uint32_t)((size * (uint64_t)lower) >> precision);
  const uint32_t b = (uint32_t)(((size * (uint64_t)upper) >> precision) - 1);
  e->base += a;
  e->size_minus1 = b - a;
  const int base_overflow = (e->base < a);

  if ((uint32_t)(e->base + e->size_minus1) < e->base) {
    if ((e->size_minus1 >> 16) == 0) {
      e->base <<= 16;
      e->size_minus1 <<= 16;
      e->size_minus1 |= 0xFFFF;
      e->delay += 0x20000;
    }
    return;
  }
  if (e->delay != 0) {
    if (base_overflow) {
      put(e, (uint8_t)(e->delay >> 8));
      put(e, (uint8_t)(e->delay >> 0));
      put_n(e, e->delay >> 16, 0x00);
    } else {
      --e->delay;
      put(e, (uint8_t)(e->delay >> 8));
      put(e, (uint8_t)(e->delay >> 0));
      put_n(e, e->delay >> 16, 0xFF);
    }
    e->delay = 0;
  }
  if ((e->size_minus1 >> 16) == 0) {
    const uint32_t top = e->base >> 16;
    e->base <<= 16;
    e->size_minus1 <<= 16;
    e->size_minus1 |= 0xFFFF;
    if (e->base <= (uint32_t)(e->base + e->size_minus1)) {
      put(e, (uint8_t)(top >> 8));
      put(e, (uint8_t)(top));
    } else {
      e->delay = (uint64_t)top + 1;
    }
  }
}

static void enc_finalize(enc_state* e) {
  if (e->delay != 0) {
    put(e, (uint8_t)(e->delay >> 8));
    if ((e->delay & 0xFF) != 0) put(e, (uint8_t)(e->delay));
  } else if (e->base != 0) {
    const uint32_t mid = ((e->base - 1) >> 16) + 1;
    put(e, (uint8_t)(mid >> 8));
    if ((mid & 0xFF) != 0) put(e, (uint8_t)(mid));
  }
}

/*
 * range_encode(data int16 [rows, cols], cdf int32): row-major traversal, the
 * CDF row of element (r,c) is cdf[(r*cdf_row_stride + c) * (n+1) ...] with
 * cdf_row_stride = cols when the CDF has a leading dim equal to rows
 * (conditional_entropy_model.py:161) or 0 when it broadcasts over rows
 * (entropy_model.py:258: cdf shape [1, C, N+1]).
 * Returns encoded length (may exceed cap -> caller must retry), <0 on error.
 */
int64_t oracle_range_encode(const int16_t* data, int64_t rows, int cols,
                            const int32_t* cdf, int n, int broadcast_rows,
                            int precision, uint8_t* out, int64_t cap) {
  enc_state e;
  enc_init(&e, out, cap);
  for (int64_t r = 0; r < rows; ++r) {
    for (int c = 0; c < cols; ++c) {
      const int32_t* row = cdf + ((broadcast_rows ? 0 : r * cols) + c) * (int64_t)(n + 1);
      int v = data[r * cols + c];
      if (v < 0 || v >= n) return -1; /* TF: "value not in [0, m)" */
      enc_encode(&e, row[v], row[v + 1], precision);
    }
  }
  enc_finalize(&e);
  return e.len;
}

typedef struct {
  uint32_t base, size_minus1, value;
  const uint8_t* cur;
  const uint8_t* end;
} dec_state;

static void dec_read16(dec_state* d) {
  d->value <<= 8;
  if (d->cur != d->end) d->value |= *d->cur++;
  d->value <<= 8;
  if (d->cur != d->end) d->value |= *d->cur++;
}

int oracle_range_decode(const uint8_t* str, int64_t len, int64_t rows, int cols,
                        const int32_t* cdf, int n, int broadcast_rows,
                        int precision, int16_t* out) {
  dec_state d;
  d.base = 0; d.size_minus1 = 0xFFFFFFFFu; d.value = 0;
  d.cur = str; d.end = str + len;
  dec_read16(&d);
  dec_read16(&d);
  for (int64_t r = 0; r < rows; ++r) {
    for (int c = 0; c < cols; ++c) {
      const int32_t* row = cdf + ((broadcast_rows ? 0 : r * cols) + c) * (int64_t)(n + 1);
      const uint64_t size = (uint64_t)d.size_minus1 + 1;
      const uint64_t offset = (((uint64_t)(uint32_t)(d.value - d.base) + 1) << precision) - 1;
      const int32_t* pv = row + 1;
      int64_t l = n;
      do {
        const int64_t half = l / 2;
        const int32_t* mid = pv + half;
        if (size * (uint64_t)(*mid) <= offset) { pv = mid + 1; l -= half + 1; }
        else { l = half; }
      } while (l > 0);
      if (pv >= row + n + 1) return -1;
      const uint32_t a = (uint32_t)((size * (uint64_t)(*(pv - 1))) >> precision);
      const uint32_t b = (uint32_t)(((size * (uint64_t)(*pv)) >> precision) - 1);
      d.base += a;
      d.size_minus1 = b - a;
      if ((d.size_minus1 >> 16) == 0) {
        d.base <<= 16;
        d.size_minus1 <<= 16;
        d.size_minus1 |= 0xFFFF;
        dec_read16(&d);
      }
      out[r * cols + c] = (int16_t)(pv - row - 1);
    }
  }
  return 0;
}
