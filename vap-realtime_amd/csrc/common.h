// Shared device helpers for libvapx (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
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
// Branch-free GELU for the FFN hot loop (98 k evaluations per stream-frame), ONE transcendental per value (round 5; rounds 1-4 used
// Abramowitz-Stegun 7.1.26: a reciprocal AND an exponential, |error| <= 1.5e-7):
//   gelu(x) = (x + |x| - |x| erfc(|x| / sqrt 2)) / 2,   erfc(z) = 2^(z q(z)),  q = degree-6 minimax fit of log2(erfc(z)) / z on [0, 4.2]
// (tools/fit_erfc_exp2.py: max |erf error| 1.8e-7 in float32 arithmetic, fp32-rounding level; beyond z = 4.2 erfc < 3e-9 and the clamp
// holds it there).  12 plain VALU operations + v_exp_f32 instead of 13 + v_rcp_f32 + v_exp_f32, no libm branches.  A NaN stays a NaN
// (v_min drops it from z, the final fma puts it back).
__device__ __forceinline__ float gelu_fast(float x) {
  const float ax = fabsf(x);
  const float z = fminf(ax * 0.70710678118654752440f, 4.2f);
  float q = fmaf(1.0022112e-4f, z, -4.6157415e-4f);
  q = fmaf(q, z, -2.3022329e-3f);
  q = fmaf(q, z, 2.9452506e-2f);
  q = fmaf(q, z, -1.4896366e-1f);
  q = fmaf(q, z, -9.1832864e-1f);
  q = fmaf(q, z, -1.6279137f);
  const float ec = __builtin_amdgcn_exp2f(q * z);     // erfc(|x| / sqrt 2)
  return 0.5f * fmaf(-ax, ec, x + ax);
}
// ReLU as torch.relu computes it (encoder_components.py:103): a NaN stays a NaN (fmaxf(NaN, 0) would return 0 and hide a poisoned
// sample that the reference carries into the LSTM state for good; tests/golden/poison20.npz pins that behaviour).
__device__ __forceinline__ float relu_nanprop(float x) { return x < 0.f ? 0.f : x; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// branch-free gate non-linearities for the LSTM recurrence (40 evaluations per lane per step): v_exp +
// v_rcp (1 ulp) instead of libm's branchy tanhf and IEEE division.  Absolute error ~1e-7.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  float t = __expf(-2.0f * fabsf(x));
  float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  return copysignf(r, x);
}

// Host side: runs `f` once per (call site, HIP device) — used to raise a kernel's dynamic-LDS limit before its first launch on that device.
// Guarded by a mutex so that a second host thread cannot launch before the first one has finished setting the attribute, and per device
// because the attribute lives with the device's copy of the function (an engine may be created on every GPU of a process).
struct PerDeviceOnce {
  std::mutex mu;
  unsigned long long done = 0;
  template <class F>
  void run(F&& f) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    std::lock_guard<std::mutex> lk(mu);
    if (done & bit) return;
    f();
    done |= bit;
  }
};

// compute units of the current device (cached per device): grid size of the persistent kernels
inline int device_cu_count() {
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  int& c = cus[dev & 63];
  if (c == 0 && (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0)) c = 256;
  return c;
}

// The same value as a NEW SSA definition the optimiser cannot look through: index expressions derived from it are not commoned with
// the ones of an earlier kernel phase, so they die with their phase instead of occupying registers across the whole kernel
// (hipcc otherwise keeps e.g. the 32 accumulator-row indices of a 64-row tile live from the softmax to the final stores and spills).
__device__ __forceinline__ int opaque_vgpr(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Split-precision path: power of two s with |x| s < 2^14 for every |x| <= bound (bound >= 0, finite).  s in [2^-100, 2^40]: a tiny or
// all-zero row is not blown up beyond what fp32 holds exactly.  Scaling an f16 (hi, lo) operand by a power of two and the fp32
// accumulator by its inverse changes no bit of the product as long as nothing overflows — which is what the scale guarantees.
__device__ __forceinline__ float pow2_scale_for(float bound) {
  const int eb = (__builtin_bit_cast(int, bound) >> 23) & 0xff;     // bound < 2^(eb - 126)
  int es = 267 - eb;                                                // biased exponent of 2^(13 - (eb - 127))
  es = es < 27 ? 27 : (es > 167 ? 167 : es);
  return __builtin_bit_cast(float, es << 23);
}

// value of lane `l` (wave-uniform index) in every lane: v_readlane_b32, no LDS
__device__ __forceinline__ float lane_bcast(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// p[idx] for a wave-uniform idx through the SCALAR cache.  Inside the persistent item loop hipcc reads the per-stream metadata (bn, ring_rot,
// ids) with vector loads — global stores precede them, so it cannot prove the words unclobbered — and waits vmcnt(0) for each: three
// dependent round trips per item that also sit out every store and prefetch in flight.
__device__ __forceinline__ int uniform_load(const int* p, int idx) {
  int v;
  asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p), "s"(idx * 4) : "memory");
  return v;
}
// workgroup barrier for LDS hand-offs WITHOUT the vmcnt(0) that __syncthreads' release fence brings: waiting for this wave's outstanding
// global STORES at a barrier (and for weight fragments fetched ahead) at every LDS hand-off serialises what the kernels work to overlap
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// 64-lane all-reduce (max; the sum is below, built on the DPP half_sum) without the LDS crossbar: four DPP steps make every 16-lane row
// uniform (xor 1 / 2 via quad_perm, then row_half_mirror and row_mirror), v_readlane fetches the four rows.  (Round 2 used six
// __shfl_xor = ds_bpermute round trips; the row statistics of a 64-row tile call this 8 times per wave.)
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));    // quad_perm [1,0,3,2]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));    // quad_perm [2,3,0,1]
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));   // row_half_mirror
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));   // row_mirror
  return fmaxf(fmaxf(lane_bcast(v, 0), lane_bcast(v, 16)), fmaxf(lane_bcast(v, 32), lane_bcast(v, 48)));
}
// all-reduce inside each 32-lane half (MFMA 32x32 accumulator rows live in one half).
// __shfl_xor lowers to ds_bpermute (an LDS-crossbar round trip per step); DPP adds are plain VALU:
// xor 1 / 2 via quad_perm, then row_half_mirror and row_mirror (valid because the value is already
// uniform inside each 4- / 8-lane group), and one ds_swizzle for the 16-lane exchange.
__device__ __forceinline__ float half_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));                   // lane ^ 16
  return v;
}
// 64-lane all-reduce: five DPP / swizzle steps inside each half, one ds_bpermute across the halves
__device__ __forceinline__ float wave_sum(float v) {
  v = half_sum(v);
  return lane_bcast(v, 0) + lane_bcast(v, 32);     // (the same two addends as v + __shfl_xor(v, 32), without the LDS round trip)
}
