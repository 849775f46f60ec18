// Reproducible single-precision exp / log / tanh / sigmoid / softplus for everything that feeds a range-coder CDF.
//
// The decoder must rebuild the encoder's integer CDFs bit for bit (conditional_entropy_model.py:95-124,
// entropy_model.py:183-221; README.md:111-114 describes the reference failing at exactly this).  A CDF entry is
// rint(pmf * 65536) of a float32 pmf, so it depends on the last ulp of exp(), and a vendor expf differs between
// device libraries, host libms and ROCm versions.  These functions are therefore defined HERE, as fixed sequences
// of IEEE-754 binary32 operations (+, -, *, /, floor, compare, exponent-field edits; round-to-nearest-even, NO
// fused multiply-add — every file that includes this header is built with -ffp-contract=off), so that the HIP
// kernels (entropy.hip), the host library and the CPU oracle (oracle/entropy.py restates them in numpy) produce
// identical bits on any implementation.  The algorithms are the classic Cephes single-precision ones (the same
// family TensorFlow's Eigen backend uses for exp / tanh); accuracy <= 2 ulp, checked in tests/test_repro_math.py.
// Constants are written as hexadecimal floats: exact in C++ and in Python.
//
//   expf_(x):  x = min(max(x, -87), 88);  n = floor(x * LOG2E + 0.5);  r = (x - n*C1) - n*C2;
//              p = ((((E0*r + E1)*r + E2)*r + E3)*r + E4)*r + E5;  y = (p * (r*r) + r) + 1;  return y * 2^n
//   logf_(x):  x normal, > 0:  x = m * 2^e, m in [0.5, 1);  if (m < SQRTH) { e -= 1; m = (m + m) - 1 } else m = m - 1;
//              z = m*m;  y = ((((((((L0*m + L1)*m + L2)*m + L3)*m + L4)*m + L5)*m + L6)*m + L7)*m + L8) * m * z;
//              y = y + C2*e;  y = y - 0.5*z;  return (m + y) + C1*e
//   tanhf_(x): a = |x|;  a >= 0.625:  t = 1 - 2 / (expf_(a + a) + 1), sign of x;
//              else z = x*x;  ((((T0*z + T1)*z + T2)*z + T3)*z + T4) * z * x + x
//   sigmoidf_(x) = 1 / (1 + expf_(-x));   softplusf_(x) = max(x, 0) + log1p(e), e = expf_(-|x|), u = 1 + e,
//              log1p(e) = (u == 1) ? e : logf_(u) * (e / (u - 1))
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define PCGC_HD __host__ __device__ __forceinline__
#else
#define PCGC_HD inline
#endif

namespace pcgc {
namespace repro {

PCGC_HD float from_bits(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
PCGC_HD uint32_t to_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

constexpr float kLog2e = 0x1.715476p+0f, kC1 = 0x1.63p-1f, kC2 = -0x1.bd0106p-13f;

PCGC_HD float expf_(float x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  x = x > -87.0f ? x : -87.0f;            // also maps NaN to -87 (comparison false)
  x = x < 88.0f ? x : 88.0f;
  float n = x * kLog2e;
  n = n + 0.5f;
  n = __builtin_floorf(n);
  float r = n * kC1;
  r = x - r;
  float t = n * kC2;
  r = r - t;
  float p = 0x1.a0d2cep-13f * r;
  p = p + 0x1.6e879cp-10f;
  p = p * r; p = p + 0x1.111210p-7f;
  p = p * r; p = p + 0x1.555382p-5f;
  p = p * r; p = p + 0x1.555554p-3f;
  p = p * r; p = p + 0.5f;
  const float z = r * r;
  float y = p * z;
  y = y + r;
  y = y + 1.0f;
  const int e = (int)n + 127;             // 2 .. 254 for the clamped range
  return y * from_bits((uint32_t)e << 23);
}

PCGC_HD float logf_(float x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const uint32_t u = to_bits(x);
  int e = (int)((u >> 23) & 0xffu) - 126;
  float m = from_bits((u & 0x807fffffu) | 0x3f000000u);   // [0.5, 1)
  if (m < 0x1.6a09e6p-1f) {
    e -= 1;
    m = m + m;
    m = m - 1.0f;
  } else {
    m = m - 1.0f;
  }
  const float z = m * m;
  float y = 0x1.204376p-4f * m;
  y = y + -0x1.d7a370p-4f;
  y = y * m; y = y + 0x1.de4a34p-4f;
  y = y * m; y = y + -0x1.fcba9ep-4f;
  y = y * m; y = y + 0x1.23d37ep-3f;
  y = y * m; y = y + -0x1.555ca0p-3f;
  y = y * m; y = y + 0x1.999d58p-3f;
  y = y * m; y = y + -0x1.fffff8p-3f;
  y = y * m; y = y + 0x1.555554p-2f;
  y = y * m;
  y = y * z;
  const float fe = (float)e;
  float t = kC2 * fe;
  y = y + t;
  t = 0.5f * z;
  y = y - t;
  float r = m + y;
  t = kC1 * fe;
  r = r + t;
  return r;
}

PCGC_HD float tanhf_(float x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float a = x < 0.0f ? -x : x;
  if (a >= 0.625f) {
    float s = expf_(a + a);
    s = s + 1.0f;
    s = 2.0f / s;
    s = 1.0f - s;
    return x < 0.0f ? -s : s;
  }
  const float z = x * x;
  float p = -0x1.75e1d4p-8f * z;
  p = p + 0x1.52269cp-6f;
  p = p * z; p = p + -0x1.b83c5ap-5f;
  p = p * z; p = p + 0x1.110726p-3f;
  p = p * z; p = p + -0x1.555532p-2f;
  p = p * z;
  p = p * x;
  return p + x;
}

PCGC_HD float sigmoidf_(float x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  float s = expf_(-x);
  s = 1.0f + s;
  return 1.0f / s;
}

PCGC_HD float softplusf_(float x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float a = x < 0.0f ? -x : x;
  const float e = expf_(-a);
  const float u = 1.0f + e;
  float s = e;                           // log1p(e) = e when 1 + e rounds to 1, else log(u) * (e / (u - 1))
  if (u != 1.0f) {
    float q = u - 1.0f;
    q = e / q;
    s = logf_(u);
    s = s * q;
  }
  const float m = x > 0.0f ? x : 0.0f;
  return m + s;
}

}  // namespace repro
}  // namespace pcgc
