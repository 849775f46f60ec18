// Device helpers shared by the row kernels on v_mfma_f32_4x4x1_16B_f32 (vrn_row.hip: 64^3 / C = 16, vrn_row32.hip:
// 32^3 / C = 32): the 16-block MFMA with A broadcast (cbsz = 4, abid = k), DPP lane shifts, raw buffer loads / stores
// whose out-of-range offsets read zeros / drop the store.  See vrn_row.hip for the mapping.
#pragma once
#include "common.h"

namespace pcgc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b128 to a ONE-dword load; bind the intrinsics directly
__device__ f32x4 raw_load4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ float raw_load1(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void raw_store4(f32x4 v, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");
__device__ void raw_store1i(int v, i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i32");

__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each with a private L2: give every XCD a
// CONTIGUOUS range of tiles, so that the halo rows neighbouring tiles share are fetched into one L2 once (speed only).
__device__ __forceinline__ int xcd_contiguous(int bid, int nblk) {
  return (nblk & 7) ? bid : (bid & 7) * (nblk >> 3) + (bid >> 3);
}
constexpr int kOOB = 0x7ffff000;   // byte offset past any cube: the buffer load returns 0

template <int ABID>
__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
}
// abid is a compile-time constant after unrolling; the switch folds away
__device__ __forceinline__ f32x4 mfa(int abid, float a, float b, f32x4 c) {
  switch (abid) {
    case 0: return mf<0>(a, b, c);
    case 1: return mf<1>(a, b, c);
    case 2: return mf<2>(a, b, c);
    case 3: return mf<3>(a, b, c);
    case 4: return mf<4>(a, b, c);
    case 5: return mf<5>(a, b, c);
    case 6: return mf<6>(a, b, c);
    case 7: return mf<7>(a, b, c);
    case 8: return mf<8>(a, b, c);
    case 9: return mf<9>(a, b, c);
    case 10: return mf<10>(a, b, c);
    case 11: return mf<11>(a, b, c);
    case 12: return mf<12>(a, b, c);
    case 13: return mf<13>(a, b, c);
    case 14: return mf<14>(a, b, c);
    default: return mf<15>(a, b, c);
  }
}

// The first MFMA of an accumulator: C = bias / zero, D = registers of their own.  The empty asm keeps a, b and d alive
// together, so D is never allocated on top of a dying operand.  History: in round 2 fresh accumulators came out wrong in
// lanes 12..15 of each row of 16 when a second wave shared the SIMD (cubes 4..7 of an 8-cube launch, run to run
// different) and this constraint made it go away; it was read as "a 4x4x1 MFMA must not have D over B".  Round 3 tested
// that directly (tools/exp/exp_mfma_overlap.hip: D over A, B or both, every position / abid / occupancy, 324 variants):
// no mismatch — the overlap is harmless.  What the constraint really changed was WHERE the new accumulator landed: without
// it the allocator reused the registers of the rows just handed to a 128-bit buffer_store with a register soffset, the one
// store form the compiler does not protect against an overwrite of its data (see rsrc_at in vrn_row.hip; tools/check_isa.py
// refuses that form in the object code).  Kept: it costs nothing and keeps the allocation away from in-flight store data.
// Validated with hipcc 7.2.26015 / clang 22.0.0git (roc-7.2.0); tests: test_every_row_kernel_is_slot_invariant_and_repeatable.
__device__ __forceinline__ f32x4 mfa_new(int abid, float a, float b, f32x4 c) {
  f32x4 d = mfa(abid, a, b, c);
  asm("" : "+v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float shr1(float v) {   // lane i <- lane i-1, lane 0 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float v) {   // lane i <- lane i+1, lane 63 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
}
__device__ __forceinline__ float comp(const f32x4& v, int c) { return v[c]; }

// lds[i] = f(i) for i < N by 256 threads, the loads of 16 elements per thread in flight before their LDS writes (the
// plain `for (i = tid; ...) lds[i] = w[index(i)]` loop waits for every load in turn)
template <int N, class F>
__device__ __forceinline__ void stage_indexed(float* lds, F f) {
  constexpr int IT = (N + 255) / 256;
#pragma unroll
  for (int k0 = 0; k0 < IT; k0 += 16) {
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = (k0 + k) * 256 + threadIdx.x;
      v[k] = (k0 + k < IT && i < N) ? f(i) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = (k0 + k) * 256 + threadIdx.x;
      if (k0 + k < IT && i < N) lds[i] = v[k];
    }
  }
}

// N floats global -> LDS by 256 threads: every 16-byte load is issued before the first LDS write, so the copy costs one
// memory round trip instead of one per loop iteration
template <int N>
__device__ __forceinline__ void stage_image(float* lds, const float* __restrict__ g) {
  constexpr int IT = (N / 4 + 255) / 256;
  float4 v[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = (k * 256 + threadIdx.x) * 4;
    v[k] = i < N ? *reinterpret_cast<const float4*>(g + i) : float4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = (k * 256 + threadIdx.x) * 4;
    if (i < N) *reinterpret_cast<float4*>(lds + i) = v[k];
  }
}

// the same for a workgroup of T threads
template <int N, int T>
__device__ __forceinline__ void stage_image_t(float* lds, const float* __restrict__ g) {
  constexpr int IT = (N / 4 + T - 1) / T;
  float4 v[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = (k * T + threadIdx.x) * 4;
    v[k] = i < N ? *reinterpret_cast<const float4*>(g + i) : float4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = (k * T + threadIdx.x) * 4;
    if (i < N) *reinterpret_cast<float4*>(lds + i) = v[k];
  }
}

}  // namespace pcgc
