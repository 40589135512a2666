// Micro-benchmark: what does ONE compute unit's store path take on MI355X, and what does it do to loads that share the CU?
// (Evidence for DESIGN.md §5: a 64 KB output tile of the split-precision FFN block takes ~1.8 us to leave whatever its coalescing.)
//
//   mode 0  stores only: every workgroup writes `tiles` tiles of 64 KB (8 waves x 8 rows x 1 KB, 16 bytes per lane, whole rows) to its own region
//   mode 1  the same bytes as 16-byte-per-lane stores scattered over 64 rows per instruction (32-byte segments: the accumulator layout)
//   mode 2  loads only: every workgroup streams `tiles` x 256 KB of a shared 3 MB weight buffer (L2-resident) into registers
//   mode 3  both: per tile 256 KB of loads and 64 KB of stores issued back to back by the same waves (a contraction followed by its store phase)
// Prints bytes / clk / CU (clock from s_memtime is not the shader clock: wall time x 2.4 GHz is used) for grids of 32 / 128 / 256 / 512
// workgroups of 512 threads (one or two per CU).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/store_path tools/microbench/store_path.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void path_kernel(float* out, const float* wts, int tiles, int mode, float* sink) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float* base = out + (long)blockIdx.x * tiles * 16384;            // 64 KB per tile per workgroup
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < tiles; ++t) {
    if (mode >= 2) {                                               // 256 KB of "weight fragments": 8 waves x 32 loads x 1 KB
      const f32x4* wp = (const f32x4*)wts + ((long)((t * 7 + blockIdx.x) % 12) * 16384) + w * 2048 + lane;
#pragma unroll 8
      for (int i = 0; i < 32; ++i) acc += wp[i * 64];
    }
    if (mode != 2) {
      float* tp = base + (long)t * 16384;
      const f32x4 v = acc + (float)t;
      if (mode == 1) {                                             // lane -> (row = lane & 31 (+32), 16 bytes at column 8 j + 4 (lane >> 5) of the wave's 32 columns)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int j = 0; j < 4; ++j) *(f32x4*)(tp + (rt * 32 + (lane & 31)) * 256 + w * 32 + 8 * j + 4 * (lane >> 5)) = v;
      } else {                                                     // 8 rows x 128 contiguous bytes per instruction
#pragma unroll
        for (int i = 0; i < 8; ++i) *(f32x4*)(tp + ((lane >> 3) + 8 * i) * 256 + w * 32 + 4 * (lane & 7)) = v;
      }
    }
  }
  if (acc[0] == 12345.678f) sink[0] = acc[1] + acc[2] + acc[3];   // keep the loads
}

int main() {
  const int tiles = 64;
  float *out, *wts, *sink;
  const size_t out_bytes = (size_t)512 * tiles * 65536;
  hipMalloc(&out, out_bytes);
  hipMalloc(&wts, (size_t)12 * 65536 * sizeof(float));
  hipMalloc(&sink, 16);
  hipMemset(wts, 0, (size_t)12 * 65536 * sizeof(float));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char* names[4] = {"stores, whole rows   ", "stores, scattered 32B", "loads (L2 weights)   ", "loads + stores       "};
  printf("workgroups of 512 threads, %d tiles each; bytes / clk / CU at 2.4 GHz (loads: 256 KB per tile, stores: 64 KB per tile)\n", tiles);
  for (int mode = 0; mode < 4; ++mode)
    for (int grid : {32, 128, 256, 512}) {
      path_kernel<<<grid, 512>>>(out, wts, tiles, mode, sink);      // warm-up
      hipDeviceSynchronize();
      hipEventRecord(a);
      path_kernel<<<grid, 512>>>(out, wts, tiles, mode, sink);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, a, b);
      const double cus = grid < 256 ? grid : 256;
      const double st = mode == 2 ? 0.0 : (double)grid * tiles * 65536, ld = mode >= 2 ? (double)grid * tiles * 262144 : 0.0;
      const double clk = ms * 1e-3 * 2.4e9;
      printf("%s grid %3d: %7.3f ms  stores %5.1f B/clk/CU (%5.2f TB/s)  loads %5.1f B/clk/CU (%5.2f TB/s)\n", names[mode], grid, ms, st / clk / cus,
             st / (ms * 1e-3) / 1e12, ld / clk / cus, ld / (ms * 1e-3) / 1e12);
    }
  hipFree(out); hipFree(wts); hipFree(sink);
  return 0;
}
