// Implicit-GEMM 3-D convolutions on the gfx950 fp32 matrix cores.
//
// Replaces tf.keras.layers.Conv3D / Conv3DTranspose (models/model_voxception.py:
// 21-54, 83-122, 153-192, 224-244, 263-297) for NDHWC fp32 tensors.
//
// Mapping onto v_mfma_f32_16x16x4_f32 (exact fp32, D = A*B + C, one wave):
//   M (rows of D)  = 16 output channels  -> A operand = weights
//   N (cols of D)  = 16 output voxels consecutive along W -> B operand = activations
//   K              = 4 input channels of one filter tap
// With this orientation lane (j = lane&15, g = lane>>4) ends up holding output
// channels 4g..4g+3 of voxel j, i.e. one float4 of the NDHWC output: the epilogue is
// a coalesced 16-byte store per lane, and the activation operand of lane (j,g) is
// VEC = min(Cin,16)/4 consecutive input channels of voxel j (+tap), one
// ds_read_b128/b64/b32 from an NDHWC tile staged in LDS with its halo.
//
// Workgroup = 256 threads = 4 waves; the workgroup owns a TD x TH x 16 block of
// output voxels (transposed conv: a block of INPUT voxels and the 2x2x2 output
// voxels each of them maps to) and all output channels; wave w owns TD*TH/4 rows
// of 16 voxels.  Input channels are processed in chunks of 16 through LDS.
// Summation order per output element is fixed (chunk, tap, channel) — no atomics,
// no split-K — so results do not depend on batch size or grid placement.
#include "common.h"

namespace pcgc {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int CIN>
struct Chunk {
  static constexpr int CK = CIN < 16 ? CIN : 16;           // channels per LDS chunk
  static constexpr int NCH = CIN / CK;                     // chunks
  static constexpr int VEC = CK / 4;                       // floats per lane per tap (K-steps)
  static constexpr int VS = CK == 16 ? 20 : (CK == 8 ? 12 : 4);  // LDS voxel stride (floats), 16-B multiple
};

template <int VEC>
__device__ __forceinline__ void lds_read_vec(const float* p, float (&v)[4]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void gl_read_vec(const float* p, float (&v)[4]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = *p;
  }
}

// epilogue for one accumulator: lane holds channels c0..c0+3 of one voxel
__device__ __forceinline__ void store_acc(const ConvArgs& a, int64_t vox, int c0, f32x4 acc) {
  if (c0 >= a.Cout) return;
  float v[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (a.bias) {
    const float4 bv = *reinterpret_cast<const float4*>(a.bias + c0);
    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (a.relu) v[r] = fmaxf(v[r], 0.f);
    if (a.absval) v[r] = fmaxf(fabsf(v[r]), a.lower_bound);
  }
  float* yp = a.y + vox * a.y_cs + a.y_co + c0;
  if (a.res) {
    const float4 rv = *reinterpret_cast<const float4*>(a.res + vox * a.y_cs + a.y_co + c0);
    v[0] = fmaxf(rv.x + v[0], 0.f); v[1] = fmaxf(rv.y + v[1], 0.f);
    v[2] = fmaxf(rv.z + v[2], 0.f); v[3] = fmaxf(rv.w + v[3], 0.f);
  }
  *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------
// stride-1 (KS = 1 or 3) and stride-2 (KS = 3) convolution
// ---------------------------------------------------------------------------
template <int CIN, int COUT, int KS, int STRIDE, int TD, int TH>
__global__ void __launch_bounds__(256) conv_mfma_kernel(ConvArgs a) {
  using C = Chunk<CIN>;
  constexpr int CK = C::CK, NCH = C::NCH, VEC = C::VEC, VS = C::VS;
  constexpr int MT = (COUT + 15) / 16;
  constexpr int NT = TD * TH / 4;
  constexpr int HALO = KS - 1;                                   // 0 or 2
  constexpr int ID = (TD - 1) * STRIDE + KS, IH = (TH - 1) * STRIDE + KS, IW = 15 * STRIDE + KS;
  constexpr int NVOX = ID * IH * IW;
  constexpr int TAPS = KS * KS * KS;
  __shared__ __attribute__((aligned(16))) float tile[NVOX * VS];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tw = a.Dout / 16, th = a.Dout / TH, td = a.Dout / TD;
  int bid = blockIdx.x;
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  const int b = bid;
  const int od0 = tx * TD, oh0 = ty * TH, ow0 = tz * 16;
  // input origin of the tile
  const int id0 = STRIDE == 1 ? od0 - HALO / 2 : od0 * 2;
  const int ih0 = STRIDE == 1 ? oh0 - HALO / 2 : oh0 * 2;
  const int iw0 = STRIDE == 1 ? ow0 - HALO / 2 : ow0 * 2;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* wl = a.w + (size_t)lane * VEC;                   // this lane's slot inside every packed block

  for (int cb = 0; cb < NCH; ++cb) {
    if (cb) __syncthreads();
    // ---- stage the input chunk (zero outside the volume = 'same' padding) ----
    constexpr int Q = CK / 4;
    for (int idx = threadIdx.x; idx < NVOX * Q; idx += 256) {
      const int v = idx / Q, q = idx - v * Q;
      const int zw = v % IW, zh = (v / IW) % IH, zd = v / (IW * IH);
      const int gd = id0 + zd, gh = ih0 + zh, gw = iw0 + zw;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)gd < (unsigned)a.Din && (unsigned)gh < (unsigned)a.Din && (unsigned)gw < (unsigned)a.Din) {
        val = *reinterpret_cast<const float4*>(
            a.x + ((((int64_t)b * a.Din + gd) * a.Din + gh) * a.Din + gw) * a.x_cs + a.x_co + cb * CK + q * 4);
      }
      *reinterpret_cast<float4*>(&tile[v * VS + q * 4]) = val;
    }
    __syncthreads();
    // ---- taps ----
    const float* wc = wl + (size_t)cb * TAPS * MT * 64 * VEC;
#pragma unroll 1
    for (int kd = 0; kd < KS; ++kd) {
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          const int tap = (kd * KS + kh) * KS + kw;
          float av[MT][4];
#pragma unroll
          for (int m = 0; m < MT; ++m) gl_read_vec<VEC>(wc + (size_t)(tap * MT + m) * 64 * VEC, av[m]);
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            const int nt = wv * NT + i;
            const int dd = nt / TH, hh = nt % TH;
            const int pos = ((dd * STRIDE + kd) * IH + (hh * STRIDE + kh)) * IW + (j * STRIDE + kw);
            float bv[4];
            lds_read_vec<VEC>(&tile[pos * VS + VEC * g], bv);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int r = 0; r < VEC; ++r) acc[m][i] = mfma4(av[m][r], bv[r], acc[m][i]);
          }
        }
      }
    }
  }
  // ---- epilogue ----
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int nt = wv * NT + i;
    const int dd = nt / TH, hh = nt % TH;
    const int64_t vox = (((int64_t)b * a.Dout + od0 + dd) * a.Dout + oh0 + hh) * a.Dout + ow0 + j;
#pragma unroll
    for (int m = 0; m < MT; ++m) store_acc(a, vox, m * 16 + 4 * g, acc[m][i]);
  }
}

// ---------------------------------------------------------------------------
// stride-2 transposed convolution: y[o] = b + sum_{2i+k=o} x[i] W[k]
// The workgroup owns TD x TH x 16 INPUT voxels (+1 low-side halo) and the
// (2TD) x (2TH) x 32 outputs they map to, one output-parity class at a time:
// per axis an even output o=2i has taps k=0 (input i) and k=2 (input i-1), an
// odd output o=2i+1 has tap k=1 (input i).  All CIN channels are staged at once.
// ---------------------------------------------------------------------------
template <int CIN, int COUT, int TD, int TH>
__global__ void __launch_bounds__(256) tconv_mfma_kernel(ConvArgs a) {
  constexpr int NCH = CIN / 16;
  constexpr int VS = CIN + 4;
  constexpr int MT = (COUT + 15) / 16;
  constexpr int NT = TD * TH / 4;
  constexpr int ID = TD + 1, IH = TH + 1, IW = 17;
  constexpr int NVOX = ID * IH * IW;
  __shared__ __attribute__((aligned(16))) float tile[NVOX * VS];

  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int tw = a.Din / 16, th = a.Din / TH, td = a.Din / TD;
  int bid = blockIdx.x;
  const int tz = bid % tw; bid /= tw;
  const int ty = bid % th; bid /= th;
  const int tx = bid % td; bid /= td;
  const int b = bid;
  const int id0 = tx * TD, ih0 = ty * TH, iw0 = tz * 16;

  constexpr int Q = CIN / 4;
  for (int idx = threadIdx.x; idx < NVOX * Q; idx += 256) {
    const int v = idx / Q, q = idx - v * Q;
    const int zw = v % IW, zh = (v / IW) % IH, zd = v / (IW * IH);
    const int gd = id0 + zd - 1, gh = ih0 + zh - 1, gw = iw0 + zw - 1;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gd >= 0 && gh >= 0 && gw >= 0)
      val = *reinterpret_cast<const float4*>(
          a.x + ((((int64_t)b * a.Din + gd) * a.Din + gh) * a.Din + gw) * a.x_cs + a.x_co + q * 4);
    *reinterpret_cast<float4*>(&tile[v * VS + q * 4]) = val;
  }
  __syncthreads();
  const float* wl = a.w + (size_t)lane * 4;

#pragma unroll 1
  for (int cls = 0; cls < 8; ++cls) {
    const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nd = pd ? 1 : 2, nh = ph ? 1 : 2, nw = pw ? 1 : 2;
#pragma unroll 1
    for (int td_ = 0; td_ < nd; ++td_) {
      const int kd = pd ? 1 : 2 * td_, offd = (kd == 2) ? 0 : 1;
#pragma unroll 1
      for (int th_ = 0; th_ < nh; ++th_) {
        const int kh = ph ? 1 : 2 * th_, offh = (kh == 2) ? 0 : 1;
#pragma unroll 1
        for (int tw_ = 0; tw_ < nw; ++tw_) {
          const int kw = pw ? 1 : 2 * tw_, offw = (kw == 2) ? 0 : 1;
          const int tap = (kd * 3 + kh) * 3 + kw;
#pragma unroll
          for (int cb = 0; cb < NCH; ++cb) {
            float av[MT][4];
#pragma unroll
            for (int m = 0; m < MT; ++m)
              gl_read_vec<4>(wl + (size_t)((cb * 27 + tap) * MT + m) * 256, av[m]);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
              const int nt = wv * NT + i;
              const int dd = nt / TH, hh = nt % TH;
              const int pos = ((dd + offd) * IH + (hh + offh)) * IW + (j + offw);
              float bv[4];
              lds_read_vec<4>(&tile[pos * VS + cb * 16 + 4 * g], bv);
#pragma unroll
              for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][i] = mfma4(av[m][r], bv[r], acc[m][i]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int nt = wv * NT + i;
      const int dd = nt / TH, hh = nt % TH;
      const int od = 2 * (id0 + dd) + pd, oh = 2 * (ih0 + hh) + ph, ow = 2 * (iw0 + j) + pw;
      const int64_t vox = (((int64_t)b * a.Dout + od) * a.Dout + oh) * a.Dout + ow;
#pragma unroll
      for (int m = 0; m < MT; ++m) store_acc(a, vox, m * 16 + 4 * g, acc[m][i]);
    }
  }
}

// ---------------------------------------------------------------------------
// weight packing: TF layout -> [chunk][tap][mtile][lane][VEC]
//   value = W[tap][ci = chunk*CK + VEC*(lane>>4) + r][co = mtile*16 + (lane&15)]   (0 past Cout)
// ---------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* w, float* p, int Cin, int Cout, int taps, int transposed, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int CK = Cin < 16 ? Cin : 16, VEC = CK / 4, MT = (Cout + 15) / 16;
  int t = idx;
  const int r = t % VEC; t /= VEC;
  const int lane = t % 64; t /= 64;
  const int m = t % MT; t /= MT;
  const int tap = t % taps; t /= taps;
  const int cb = t;
  const int ci = cb * CK + VEC * (lane >> 4) + r;
  const int co = m * 16 + (lane & 15);
  float v = 0.f;
  if (co < Cout) v = transposed ? w[((size_t)tap * Cout + co) * Cin + ci] : w[((size_t)tap * Cin + ci) * Cout + co];
  p[idx] = v;
}

size_t mfma_packed_floats(int Cin, int Cout, int ksize, int mode) {
  (void)mode;
  const int taps = ksize * ksize * ksize;
  const int MT = (Cout + 15) / 16;
  return (size_t)taps * Cin * MT * 16;      // = chunks * taps * MT * 64 * VEC
}

int pack_weights_mfma(const float* w_tf, float* packed, int Cin, int Cout, int ksize, int mode, hipStream_t s) {
  const int total = (int)mfma_packed_floats(Cin, Cout, ksize, mode);
  hipLaunchKernelGGL(pack_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w_tf, packed, Cin, Cout,
                     ksize * ksize * ksize, mode == 2 ? 1 : 0, total);
  return launch_ok("pack_weights_kernel");
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
template <int CIN, int COUT, int KS, int STRIDE, int TD, int TH>
static int run_conv(const ConvArgs& a, hipStream_t s) {
  const int blocks = a.B * (a.Dout / TD) * (a.Dout / TH) * (a.Dout / 16);
  hipLaunchKernelGGL((conv_mfma_kernel<CIN, COUT, KS, STRIDE, TD, TH>), dim3(blocks), dim3(256), 0, s, a);
  return launch_ok("conv_mfma_kernel");
}
template <int CIN, int COUT, int TD, int TH>
static int run_tconv(const ConvArgs& a, hipStream_t s) {
  const int blocks = a.B * (a.Din / TD) * (a.Din / TH) * (a.Din / 16);
  hipLaunchKernelGGL((tconv_mfma_kernel<CIN, COUT, TD, TH>), dim3(blocks), dim3(256), 0, s, a);
  return launch_ok("tconv_mfma_kernel");
}

#define CASE_S1(ci, co, ks)                                          \
  if (a.Cin == ci && a.Cout == co && a.ksize == ks) {                \
    if (!run) return 1;                                              \
    int rc = run_conv<ci, co, ks, 1, 4, 4>(b, s);                    \
    return rc ? rc : 1;                                              \
  }
#define CASE_S2(ci, co)                                              \
  if (a.Cin == ci && a.Cout == co) {                                 \
    if (!run) return 1;                                              \
    int rc = run_conv<ci, co, 3, 2, 2, 2>(b, s);                     \
    return rc ? rc : 1;                                              \
  }
#define CASE_T(ci, co, td, th)                                       \
  if (a.Cin == ci && a.Cout == co) {                                 \
    if (!run) return 1;                                              \
    int rc = run_tconv<ci, co, td, th>(b, s);                        \
    return rc ? rc : 1;                                              \
  }

int launch_conv_mfma(const ConvArgs& a, const float* packed_w, hipStream_t s, bool run) {
  ConvArgs b = a;
  b.w = packed_w;
  // alignment / geometry preconditions of the tiled kernels
  if (a.Cout % 4 || a.x_cs % 4 || a.x_co % 4 || a.y_cs % 4 || a.y_co % 4) return 0;
  if (a.mode == 0) {
    if (a.Dout % 16) return 0;
    CASE_S1(16, 16, 3) CASE_S1(16, 8, 3) CASE_S1(16, 4, 3) CASE_S1(16, 64, 3) CASE_S1(16, 32, 3)
    CASE_S1(4, 8, 3) CASE_S1(4, 4, 3) CASE_S1(8, 16, 3) CASE_S1(8, 8, 3)
    CASE_S1(32, 8, 3) CASE_S1(32, 16, 3) CASE_S1(64, 16, 3)
    CASE_S1(16, 4, 1) CASE_S1(4, 8, 1) CASE_S1(32, 8, 1) CASE_S1(8, 16, 1) CASE_S1(64, 16, 1) CASE_S1(16, 32, 1)
    return 0;
  }
  if (a.mode == 1) {
    if (a.Dout % 16 || a.ksize != 3) return 0;
    CASE_S2(16, 32) CASE_S2(32, 64) CASE_S2(16, 16)
    return 0;
  }
  if (a.mode == 2) {
    if (a.Din % 16 || a.ksize != 3) return 0;
    CASE_T(64, 32, 2, 4) CASE_T(32, 16, 4, 4) CASE_T(16, 16, 4, 4)
    return 0;
  }
  return 0;
}

}  // namespace pcgc
