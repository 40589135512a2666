// Power probe: what the chip draws, and what clock it holds, when ONE resource is driven flat out on all 256 CUs for a few seconds — the f16 and
// fp32 matrix pipes, the LDS read path, the L2 -> VGPR path and the plain VALU.  Run under tools/clock_watch.sh-style sampling (the program
// prints its own achieved rates; `rocm-smi --showclocks --showpower` sampled beside it gives watts and MHz):
//   tools/microbench/power_probe <mode> <seconds>      mode: f16 | f32 | lds | l2 | valu | mix (f16 MFMA at ~45 % duty + LDS + L2, the FFN block's blend)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/power_probe tools/microbench/power_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k_f16(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(1.0f + 0.25f * i); }
  f32x16 c[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u & 3], 0, 0, 0);
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k_f32(float* out, int iters) {
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  f32x16 c[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[u & 3], 0, 0, 0);
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void k_lds(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[32 * 1024];   // 128 KB
  for (int i = threadIdx.x; i < 32 * 1024; i += 512) lds[i] = i;
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int base = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 1056;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += *(const f32x4*)&lds[(base + u * 264 + (it & 7) * 4096) & (32 * 1024 - 4)];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ __launch_bounds__(512) void k_l2(float* out, const f32x4* w, int iters) {   // every CU streams the same 2.3 MB (an FFN layer's fragments) over and over
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int n = 2304 * 1024 / 16;                           // 16-byte units
  int p = threadIdx.x + (blockIdx.x * 8191) % n;
  for (int it = 0; it < iters; ++it) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v[u] = w[p]; p += 512; p = p >= n ? p - n : p; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ __launch_bounds__(512) void k_valu(float* out, int iters) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float m = 0.999f, d = 1e-3f;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 64; ++u) v[u & 15] = __builtin_fmaf(v[u & 15], m, d);
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the FFN block's blend per k-chunk and wave: 6 f16 MFMAs, 4 ds_read_b128, 2 global 16-byte loads from an L2-resident stream, then idle VALU-ish filler
__global__ __launch_bounds__(512) void k_mix(float* out, const f32x4* w, int iters, int filler) {
  __shared__ __attribute__((aligned(16))) float lds[32 * 1024];
  for (int i = threadIdx.x; i < 32 * 1024; i += 512) lds[i] = (float)(i & 255) * 1e-3f;
  __syncthreads();
  f32x16 c[2];
  for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
  const int n = 2304 * 1024 / 16;
  int p = threadIdx.x + (blockIdx.x * 8191) % n;
  const int base = (threadIdx.x & 31) * 264 + (threadIdx.x >> 5 & 1) * 4;
  float fv[8];
  for (int i = 0; i < 8; ++i) fv[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < iters; ++it) {
    const f32x4 w0 = w[p]; p += 512; p = p >= n ? p - n : p;
    const f32x4 w1 = w[p]; p += 512; p = p >= n ? p - n : p;
    const f32x4 a0 = *(const f32x4*)&lds[(base + (it & 15) * 8) & (32 * 1024 - 4)];
    const f32x4 a1 = *(const f32x4*)&lds[(base + 8448 + (it & 15) * 8) & (32 * 1024 - 4)];
    const f32x4 a2 = *(const f32x4*)&lds[(base + 16896 + (it & 15) * 8) & (32 * 1024 - 4)];
    const f32x4 a3 = *(const f32x4*)&lds[(base + 25344 + (it & 15) * 8) & (32 * 1024 - 4)];
    const f16x8 bh = __builtin_bit_cast(f16x8, w0), bl = __builtin_bit_cast(f16x8, w1);
    c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, __builtin_bit_cast(f16x8, a0), c[0], 0, 0, 0);
    c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, __builtin_bit_cast(f16x8, a1), c[1], 0, 0, 0);
    c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, __builtin_bit_cast(f16x8, a2), c[0], 0, 0, 0);
    c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, __builtin_bit_cast(f16x8, a3), c[1], 0, 0, 0);
    c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, __builtin_bit_cast(f16x8, a0), c[0], 0, 0, 0);
    c[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, __builtin_bit_cast(f16x8, a1), c[1], 0, 0, 0);
    for (int f = 0; f < filler; ++f)
#pragma unroll
      for (int u = 0; u < 8; ++u) fv[u] = __builtin_fmaf(fv[u], 0.999f, 1e-3f);
  }
  float s = 0.f;
  for (int j = 0; j < 2; ++j) s += c[j][0] + c[j][15];
  for (int i = 0; i < 8; ++i) s += fv[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "f16";
  const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
  const int filler = argc > 3 ? atoi(argv[3]) : 0;
  float* out; f32x4* w;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&w, 2304 * 1024); hipMemset(w, 0, 2304 * 1024);
  const int iters = 20000;
  auto launch = [&]() {
    if (!strcmp(mode, "f16")) hipLaunchKernelGGL(k_f16, dim3(256), dim3(512), 0, 0, out, iters);
    else if (!strcmp(mode, "f32")) hipLaunchKernelGGL(k_f32, dim3(256), dim3(512), 0, 0, out, iters / 2);
    else if (!strcmp(mode, "lds")) hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 0, 0, out, iters);
    else if (!strcmp(mode, "l2")) hipLaunchKernelGGL(k_l2, dim3(256), dim3(512), 0, 0, out, w, iters / 4);
    else if (!strcmp(mode, "valu")) hipLaunchKernelGGL(k_valu, dim3(256), dim3(512), 0, 0, out, iters);
    else hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, out, w, iters, filler);
  };
  launch(); hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  while (el < seconds) {
    for (int i = 0; i < 4; ++i) launch();
    hipDeviceSynchronize();
    launches += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double per = el / launches;
  const double waves = 256.0 * 8;
  if (!strcmp(mode, "f16")) printf("f16 MFMA 32x32x16: %.1f TFLOP/s dense (%.2f ms / launch)\n", waves * iters * 16.0 * 32768 / per / 1e12, per * 1e3);
  else if (!strcmp(mode, "f32")) printf("fp32 MFMA 32x32x2: %.1f TFLOP/s (%.2f ms / launch)\n", waves * (iters / 2) * 16.0 * 4096 / per / 1e12, per * 1e3);
  else if (!strcmp(mode, "lds")) printf("LDS ds_read_b128: %.1f TB/s chip-wide = %.1f B / clk / CU at 2.4 GHz (%.2f ms / launch)\n", waves * iters * 16.0 * 1024 / per / 1e12, waves * iters * 16.0 * 1024 / per / 256 / 2.4e9, per * 1e3);
  else if (!strcmp(mode, "l2")) printf("L2 -> VGPR 16-byte loads: %.1f TB/s chip-wide = %.1f B / clk / CU at 2.4 GHz (%.2f ms / launch)\n", waves * (iters / 4) * 8.0 * 1024 / per / 1e12, waves * (iters / 4) * 8.0 * 1024 / per / 256 / 2.4e9, per * 1e3);
  else if (!strcmp(mode, "valu")) printf("v_fma_f32: %.1f TFLOP/s (%.2f ms / launch)\n", waves * iters * 64.0 * 128 / per / 1e12, per * 1e3);
  else printf("FFN-block blend (filler %d): %.1f TFLOP/s f16 dense, %.1f TB/s LDS, %.1f TB/s L2 (%.2f ms / launch)\n", filler, waves * iters * 6.0 * 32768 / per / 1e12, waves * iters * 4.0 * 1024 / per / 1e12, waves * iters * 2.0 * 1024 / per / 1e12, per * 1e3);
  return 0;
}
