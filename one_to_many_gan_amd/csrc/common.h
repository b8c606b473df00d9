// Shared device helpers for libo2m_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/o2m_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA 32x32x16 operand
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // one 32x32 accumulator tile
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define O2M_WAVE 64

__device__ __forceinline__ unsigned short f2bf(float x) {
  __bf16 h = (__bf16)x;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned short b) {
  return __builtin_bit_cast(float, (unsigned int)b << 16);
}
__device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) {
  return (unsigned int)f2bf(lo) | ((unsigned int)f2bf(hi) << 16);
}

// Element traits: storage type T is `unsigned short` (bf16 bits) or `float`.
template <typename T> struct Elem;
template <> struct Elem<unsigned short> {
  static constexpr int dtype = O2M_BF16;
  static __device__ __forceinline__ float ld(const unsigned short* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = f2bf(v); }
};
template <> struct Elem<float> {
  static constexpr int dtype = O2M_F32;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};

// 8 consecutive channels <-> 8 floats (16 B of bf16 or 32 B of fp32; pointer 16-B aligned)
__device__ __forceinline__ void load8(const unsigned short* p, float (&v)[8]) {
  u32x4 r = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __builtin_bit_cast(float, r[i] << 16);
    v[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void load8_stream(const unsigned short* p, float (&v)[8]) {
  u32x4 r = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __builtin_bit_cast(float, r[i] << 16);
    v[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void load8_stream(const float* p, float (&v)[8]) {
  f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void store8(unsigned short* p, const float (&v)[8]) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = r;
}
// Streaming form (non-temporal stores): for output rows far larger than the caches, which would otherwise evict
// the input lines the filter taps re-read through L2.
__device__ __forceinline__ void store8_stream(unsigned short* p, const float (&v)[8]) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
  __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(p));
}
__device__ __forceinline__ void store8_stream(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(p));
  __builtin_nontemporal_store(b, reinterpret_cast<f32x4*>(p + 4));
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

// 8 consecutive bf16 channels ADDED to memory: four packed atomics (global_atomic_pk_add_bf16, round to nearest even)
__device__ __forceinline__ void atomic_add8(unsigned short* p, const float (&v)[8]) {
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) s16x2_t gs16x2_t;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    __builtin_amdgcn_global_atomic_fadd_v2bf16((gs16x2_t*)(p + 2 * i), __builtin_bit_cast(s16x2_t, pack_bf2(v[2 * i], v[2 * i + 1])));
}
__device__ __forceinline__ void atomic_add8(float* p, const float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) unsafeAtomicAdd(p + i, v[i]);
}

// The activation code is uniform over a launch.  A per-element `switch` compiles into a chain of scalar
// compare-and-branch blocks PER ELEMENT (with the tanh expansion in the middle): measured 8 of the 11 us
// the p8 igemm's epilogue took.  So: ReLU / LeakyReLU / identity are one select each, driven by two
// uniform values, and tanh sits behind ONE uniform branch per call (act_fwd8: per 8 values).
// Streaming variants chosen at run time (block-uniform flag): tensors of >= 64 MiB that a kernel touches once.
// Read or written normally they only push the lines the concurrent MFMA kernels re-read out of L2.
constexpr size_t kStreamBytes = (size_t)64 << 20;
#ifndef O2M_NO_STREAMING
#define O2M_NO_STREAMING 0  // (1: every streaming load / store becomes a plain one -- A/B builds, tools/build_variant.sh)
#endif
template <typename T>
__device__ __forceinline__ void load8x(const T* p, float (&v)[8], bool stream) {
  if (stream && !O2M_NO_STREAMING) load8_stream(p, v);
  else load8(p, v);
}
template <typename T>
__device__ __forceinline__ void store8x(T* p, const float (&v)[8], bool stream) {
  if (stream && !O2M_NO_STREAMING) store8_stream(p, v);
  else store8(p, v);
}

__device__ __forceinline__ float act_fwd_piecewise(float v, bool relu, float neg) {
  return v > 0.f ? v : (relu ? 0.f : v * neg);
}
__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == O2M_ACT_TANH) return tanhf(v);
  return act_fwd_piecewise(v, act == O2M_ACT_RELU, act == O2M_ACT_LRELU ? 0.2f : 1.f);
}
__device__ __forceinline__ void act_fwd8(float (&v)[8], int act) {
  if (act == O2M_ACT_NONE) return;
  if (act == O2M_ACT_TANH) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = tanhf(v[q]);
    return;
  }
  const bool relu = act == O2M_ACT_RELU;
  const float neg = act == O2M_ACT_LRELU ? 0.2f : 1.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = act_fwd_piecewise(v[q], relu, neg);
}
// derivative expressed through the activation OUTPUT y (branch-free: selects on uniform values)
__device__ __forceinline__ float act_bwd_from_out(float y, int act) {
  const float neg = act == O2M_ACT_RELU ? 0.f : (act == O2M_ACT_LRELU ? 0.2f : 1.f);
  const float piece = y > 0.f ? 1.f : neg;
  return act == O2M_ACT_TANH ? 1.f - y * y : piece;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (bid % 8 labels
// the blocks that share an L2), so consecutive LOGICAL tiles -- which share operand panels --
// are handed to one XCD: logical = (chunk of that XCD) + position inside it.  Bijective for
// any grid size.  Speed only; nothing depends on the placement.
__device__ __forceinline__ int xcd_tile_order(int bid, int n) {
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = n >> 3, r = n & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Per-kernel launch timing (o2m_launch_timing in include/o2m_hip.h): when switched on, every MFMA conv
// launch site brackets its kernel with a HIP-event pair on the launch stream and files it under the
// kernel's own name.  Off: one load of a flag.  Definitions in pointwise.hip.
namespace o2m_timing {
extern bool g_on;
int open(const char* name, double flops, hipStream_t s);
void close(int slot, hipStream_t s);
}  // namespace o2m_timing
struct LaunchScope {
  hipStream_t s;
  int slot = -1;
  template <typename... A>
  LaunchScope(hipStream_t s_, double flops, const char* fmt, A... a) : s(s_) {
    if (o2m_timing::g_on) {
      char name[64];
      snprintf(name, sizeof(name), fmt, a...);
      slot = o2m_timing::open(name, flops, s);
    }
  }
  ~LaunchScope() {
    if (slot >= 0) o2m_timing::close(slot, s);
  }
};

#define O2M_LAUNCH_CHECK()                        \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)
