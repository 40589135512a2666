// Shared device helpers for libvapx (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VAPX_DIM 256
#define VAPX_FFN 768
#define VAPX_HEADS 4
#define VAPX_PAD 320

// Affine row addressing used for every activation tensor: logical row m lives at float offset
//   (m / R) * gs + (m % R) * rs.
// Contiguous [M][ld] is {gs = R*ld, rs = ld}.  A conv layer's implicit-GEMM operand is the same
// thing with rs = stride*256 (overlapping windows of the channels-last activation) and gs = the
// guarded per-(stream,channel) slab.
struct RowMap {
  long gs;
  long rs;
  int R;
};

__device__ __forceinline__ long row_off(const RowMap& r, int m) {
  int g = m / r.R;
  return (long)g * r.gs + (long)(m - g * r.R) * r.rs;
}

static inline RowMap contiguous_rows(long ld) {
  RowMap r;
  r.R = 1 << 30;
  r.gs = 0;
  r.rs = ld;
  return r;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// branch-free gate non-linearities for the LSTM recurrence (40 evaluations per lane per step): v_exp +
// v_rcp (1 ulp) instead of libm's branchy tanhf and IEEE division.  Absolute error ~1e-7.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  float t = __expf(-2.0f * fabsf(x));
  float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  return copysignf(r, x);
}

// 64-lane butterfly all-reduce (sum / max)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// all-reduce inside each 32-lane half (MFMA 32x32 accumulator rows live in one half)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
