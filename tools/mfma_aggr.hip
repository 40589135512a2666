// Diagnostic only: another process that keeps every SIMD's matrix core busy from registers (no memory traffic).
// usage: mfma_aggr <f16|f32|bf16|fp8> <seconds> [waves_per_wg=4] [wgs=2048]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256) void burn(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float a = seed + threadIdx.x * 1e-3f, b = 0.5f - threadIdx.x * 1e-3f;
  h16x8 ah, bh; b16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b - i); ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
      else if (KIND == 1) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
      else acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[t], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const char* kind = argc > 1 ? argv[1] : "f16";
  const double secs = argc > 2 ? atof(argv[2]) : 10.0;
  const int wgs = argc > 3 ? atoi(argv[3]) : 2048;
  float* out;
  (void)hipMalloc(&out, (size_t)wgs * 256 * 4);
  printf("running\n"); fflush(stdout);
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    for (int k = 0; k < 8; ++k) {
      if (!strcmp(kind, "f16")) hipLaunchKernelGGL(burn<0>, dim3(wgs), dim3(256), 0, nullptr, out, 2000, 0.25f);
      else if (!strcmp(kind, "f32")) hipLaunchKernelGGL(burn<1>, dim3(wgs), dim3(256), 0, nullptr, out, 2000, 0.25f);
      else hipLaunchKernelGGL(burn<2>, dim3(wgs), dim3(256), 0, nullptr, out, 2000, 0.25f);
    }
    (void)hipDeviceSynchronize();
  }
  return 0;
}
