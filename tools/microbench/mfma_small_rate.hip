// Micro-benchmark: issue rate of the small-tile fp32 MFMAs one wave per SIMD / two waves per SIMD can sustain, against v_mfma_f32_32x32x2_f32
// (64 FLOP / clk / SIMD = 64 clk per instruction): v_mfma_f32_4x4x1_16B_f32 (512 FLOP), v_mfma_f32_16x16x1_4B_f32 (2048 FLOP),
// v_mfma_f32_16x16x4_f32 (2048 FLOP).  Ten independent accumulators per shape, back to back.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mfma_small_rate tools/microbench/mfma_small_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void rate_kernel(float* out, int iters, unsigned long long* clk) {
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  f32x4 c4[10];
  f32x16 c16[4];
  for (int i = 0; i < 10; ++i) c4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c16[i][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (SHAPE == 0) {
#pragma unroll
      for (int i = 0; i < 10; ++i) c4[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c4[i], 0, 0, 0);
    } else if constexpr (SHAPE == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) c16[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c16[i], 0, 0, 0);
    } else if constexpr (SHAPE == 2) {
#pragma unroll
      for (int i = 0; i < 10; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) c16[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c16[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 10; ++i) s += c4[i][0] + c4[i][3];
  for (int i = 0; i < 4; ++i) s += c16[i][0] + c16[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int SHAPE>
void run(const char* name, int per_iter, int flop, int threads) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 1 << 20); hipMalloc(&clk, 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<SHAPE>, dim3(256), dim3(threads), 0, 0, out, 100, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(rate_kernel<SHAPE>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * per_iter;               // MFMAs per wave
  const double waves_per_simd = threads / 256.0;
  const double ns_per = ms * 1e6 / (n * waves_per_simd);   // per MFMA per SIMD
  printf("%-28s %d waves / SIMD: %.2f ns per instruction per SIMD = %.1f clk at 2.4 GHz, %.1f FLOP / clk / SIMD\n", name, (int)waves_per_simd, ns_per, ns_per * 2.4, flop / (ns_per * 2.4));
}

int main() {
  for (int threads : {256, 512}) {
    run<3>("v_mfma_f32_32x32x2_f32", 4, 4096, threads);
    run<2>("v_mfma_f32_16x16x4_f32", 10, 2048, threads);
    run<1>("v_mfma_f32_16x16x1_4B_f32", 4, 2048, threads);
    run<0>("v_mfma_f32_4x4x1_16B_f32", 10, 512, threads);
  }
  return 0;
}
