// Diagnostic only: a deterministic fp32-MFMA workload checked bit-for-bit against its own first launch.  Run it next to
// tools/mfma_aggr (another process) to see whether co-running matrix-core work of another data type disturbs it.
// usage: mfma_victim <launches> <mode>
//   fp32 MFMA victims:  0 operands from registers, 1 from global memory, 2 A operand from LDS (per-lane rows, stride 260)
//   VALU FMA victims:   3 global + LDS + cross-lane reduction, 4 global + LDS, 5 global only, 7 registers only,
//                       6 activations by WAVE-UNIFORM 16-byte LDS reads   <- wrong data next to `mfma_aggr f16` / `bf16`
//                       11 the same by wave-uniform 8-byte reads          <- wrong data too
//                       8 wave-uniform 4-byte reads, 9 one word per lane + v_readlane, 10 per-lane distinct 16-byte reads
// prints how many launches differ bit-wise from the first one (alone: always 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void work(const float* __restrict__ w, float* out, int iters) {
  __shared__ float lds[64 * 260];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  if (MODE == 2) {
    for (int i = tid; i < 64 * 260; i += 256) lds[i] = w[(blockIdx.x * 131 + i) & 0xffff];
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
    f32x4 a, b;
    if (MODE == 0) { a = f32x4{0.5f + lane * 1e-3f, -0.25f, 0.125f + it * 1e-4f, 1.0f}; b = f32x4{1.0f, 0.5f - lane * 1e-3f, 0.25f, -0.5f}; }
    else if (MODE == 1) { a = *(const f32x4*)(w + (((it * 4 + wv) * 64 + lane) * 4 & 0xffff)); b = *(const f32x4*)(w + (((it * 4 + wv + 7) * 64 + lane) * 4 & 0xffff)); }
    else { a = *(const f32x4*)(lds + (lane & 31) * 260 + ((it * 8 + (lane >> 5) * 4) & 255)); b = *(const f32x4*)(w + (((it * 4 + wv) * 64 + lane) * 4 & 0xffff)); }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[3 - s], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3 - s], b[s], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3 - s], b[3 - s], acc[3], 0, 0, 0);
    }
  }
  for (int t = 0; t < 4; ++t)
    for (int r = 0; r < 16; ++r) out[((long)blockIdx.x * 64 + t * 16 + r) * 256 + tid] = acc[t][r];
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
// mode 3: packed-fp32 VALU FMA chains (v_pk_fma_f32) fed from global + LDS like a mat-vec, plus cross-lane DPP reductions
// FEAT bits: 1 weights from global memory, 2 activations from LDS, 4 cross-lane reduction at the end
// PAT (how the LDS activations are read): 0 wave-uniform 16-byte reads, 1 wave-uniform 4-byte reads (strided), 2 one distinct word
// per lane + v_readlane broadcast, 3 per-lane distinct 16-byte reads, 4 wave-uniform 8-byte reads
template <int FEAT, int PAT = 0>
__global__ __launch_bounds__(256) void valu_work(const float* __restrict__ w, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float xs[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 256) xs[i] = w[(blockIdx.x * 17 + i) & 0xffff];
  __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  for (int it = 0; it < iters; ++it) {
    f32x4 wv0, wv1, xv;
    if (FEAT & 1) { wv0 = *(const f32x4*)(w + ((it * 1024 + tid * 4) & 0xffff)); wv1 = *(const f32x4*)(w + ((it * 1024 + tid * 4 + 4096) & 0xffff)); }
    else { wv0 = f32x4{0.5f, -0.25f, 0.125f, 1.0f} * (1.0f + (it & 7) * 0.01f + tid * 1e-4f); wv1 = f32x4{-0.5f, 0.75f, 0.3f, -1.0f} * (1.0f - (it & 3) * 0.01f); }
    if (FEAT & 2) {
      if (PAT == 0) xv = *(const f32x4*)(xs + ((it * 4) & 1020));
      else if (PAT == 1) { const int i = it & 255; xv = f32x4{xs[i], xs[i + 256], xs[i + 512], xs[i + 768]}; }
      else if (PAT == 2) {
        const float mine = xs[(it * 64 + (tid & 63)) & 1023];
        xv = f32x4{__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), 3)),
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), 17)),
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), 40)),
                   __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), 61))};
      } else if (PAT == 3) xv = *(const f32x4*)(xs + ((it * 4 + (tid & 63) * 4) & 1020));
      else { const f32x2 lo = *(const f32x2*)(xs + ((it * 2) & 1022)), hi2 = *(const f32x2*)(xs + ((it * 2 + 512) & 1022)); xv = f32x4{lo[0], lo[1], hi2[0], hi2[1]}; }   // PAT 4: wave-uniform 8-byte reads
    }
    else xv = f32x4{0.01f, -0.02f, 0.03f, 0.015f} * (1.0f + (it & 15) * 0.001f);
    for (int u = 0; u < 4; ++u) { acc0 += wv0 * xv[u]; acc1 += wv1 * xv[3 - u]; }
  }
  float r = (acc0[0] + acc0[1]) + (acc0[2] + acc0[3]) + (acc1[0] + acc1[1]) + (acc1[2] + acc1[3]);
  if (FEAT & 4) for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o);
  out[(long)blockIdx.x * 256 + tid] = r + acc0[tid & 3];
}
int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;
  const int wgs = 2048;
  const size_t n = (size_t)wgs * 64 * 256;
  float *w, *out;
  (void)hipMalloc(&w, 65536 * 4 + 64); (void)hipMalloc(&out, n * 4);
  std::vector<float> hw(65536 + 16);
  srand(1);
  for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-3f;
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> ref(n), got(n);
  long bad_launches = 0, bad_words = 0;
  for (int l = 0; l < launches; ++l) {
    if (mode == 0) hipLaunchKernelGGL(work<0>, dim3(wgs), dim3(256), 0, nullptr, w, out, 512);
    else if (mode == 1) hipLaunchKernelGGL(work<1>, dim3(wgs), dim3(256), 0, nullptr, w, out, 512);
    else if (mode == 2) hipLaunchKernelGGL(work<2>, dim3(wgs), dim3(256), 0, nullptr, w, out, 512);
    else if (mode == 3) hipLaunchKernelGGL(valu_work<7>, dim3(wgs), dim3(256), 0, nullptr, w, out, 4096);
    else if (mode == 4) hipLaunchKernelGGL(valu_work<3>, dim3(wgs), dim3(256), 0, nullptr, w, out, 4096);
    else if (mode == 5) hipLaunchKernelGGL(valu_work<1>, dim3(wgs), dim3(256), 0, nullptr, w, out, 4096);
    else if (mode == 6) hipLaunchKernelGGL(valu_work<2>, dim3(wgs), dim3(256), 0, nullptr, w, out, 4096);
    else if (mode == 7) hipLaunchKernelGGL(valu_work<0>, dim3(wgs), dim3(256), 0, nullptr, w, out, 4096);
    else if (mode == 8) hipLaunchKernelGGL((valu_work<2, 1>), dim3(wgs), dim3(256), 0, nullptr, w, out, 4096);
    else if (mode == 9) hipLaunchKernelGGL((valu_work<2, 2>), dim3(wgs), dim3(256), 0, nullptr, w, out, 4096);
    else if (mode == 10) hipLaunchKernelGGL((valu_work<2, 3>), dim3(wgs), dim3(256), 0, nullptr, w, out, 4096);
    else hipLaunchKernelGGL((valu_work<2, 4>), dim3(wgs), dim3(256), 0, nullptr, w, out, 4096);
    (void)hipMemcpy(l == 0 ? ref.data() : got.data(), out, n * 4, hipMemcpyDeviceToHost);
    if (l > 0) {
      long d = 0;
      for (size_t i = 0; i < n; ++i) d += memcmp(&ref[i], &got[i], 4) != 0;
      if (d) { ++bad_launches; bad_words += d; }
    }
  }
  printf("mode %d: %ld of %d launches differ from the first (%ld words)\n", mode, bad_launches, launches - 1, bad_words);
  return 0;
}
