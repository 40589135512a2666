// Micro-benchmark: how much independent VALU / transcendental work fits in the shadow of v_mfma_f32_32x32x16_f16 (8 passes = 32 clk) on one
// SIMD, with one or two waves per SIMD.  Per iteration: 8 MFMAs alternating over two (or four) accumulators and K independent v_fma_f32 (or
// v_exp_f32) per MFMA, placed right behind it (sched_group_barrier pins the order).  Prints clk per MFMA for K = 0 .. 10.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mfma_valu_overlap tools/microbench/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K, int TRANS, int NACC, int AGPR = 0>
__global__ __launch_bounds__(512) void overlap_kernel(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(1.0f + i * 0.25f); }
  f32x16 c[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
  float v[10];
  for (int i = 0; i < 10; ++i) v[i] = threadIdx.x * 0.001f + i;
  const float m = 0.999f, d = 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[u % NACC]) : "v"(a), "v"(b));   // accumulator in AccVGPRs
      else c[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u % NACC], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if constexpr (TRANS) v[k] = __builtin_amdgcn_exp2f(v[k]);
        else v[k] = __builtin_fmaf(v[k], m, d);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if constexpr (K > 0) __builtin_amdgcn_sched_group_barrier(TRANS ? 0x400 : 0x002, K, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += c[j][0] + c[j][15];
  for (int i = 0; i < 10; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int TRANS, int NACC, int AGPR = 0>
double run(int threads) {
  static float* out = nullptr;
  if (!out) hipMalloc(&out, 1 << 22);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((overlap_kernel<K, TRANS, NACC, AGPR>), dim3(256), dim3(threads), 0, 0, out, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((overlap_kernel<K, TRANS, NACC, AGPR>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = threads / 256.0;
  return ms * 1e6 / ((double)iters * 8 * waves_per_simd);   // ns per MFMA per SIMD
}

template <int TRANS, int NACC, int AGPR = 0>
void sweep(int threads) {
  const double base = run<0, TRANS, NACC, AGPR>(threads);
  printf("%s%d waves / SIMD, %d accumulators, %s per MFMA: ns per MFMA per SIMD (relative to K = 0: %.2f ns = 32 clk)\n", AGPR ? "[accumulators in AGPRs] " : "", threads / 256, NACC, TRANS ? "v_exp_f32" : "v_fma_f32", base);
  double r[6] = {run<1, TRANS, NACC, AGPR>(threads), run<2, TRANS, NACC, AGPR>(threads), run<4, TRANS, NACC, AGPR>(threads), run<6, TRANS, NACC, AGPR>(threads), run<8, TRANS, NACC, AGPR>(threads), run<10, TRANS, NACC, AGPR>(threads)};
  const int ks[6] = {1, 2, 4, 6, 8, 10};
  for (int i = 0; i < 6; ++i) printf("   K = %2d: %.2f ns  = %.1f clk per MFMA  (+%.1f clk for %d VALU)\n", ks[i], r[i], r[i] / base * 32, (r[i] / base - 1) * 32, ks[i]);
}

int main() {
  sweep<0, 2>(256); sweep<0, 2>(512); sweep<0, 4>(512);
  sweep<1, 2>(256); sweep<1, 2>(512);
  sweep<0, 2, 1>(256); sweep<0, 2, 1>(512); sweep<1, 2, 1>(512);
  return 0;
}
