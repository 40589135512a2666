// Micro-benchmark: how fast can ONE workgroup fill LDS from an L2-resident buffer, and by which path?
//   mode 0  global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave-instruction), `depth` pieces in flight per wave
//   mode 1  global_load_dwordx4 -> VGPR -> ds_write_b128, `depth` loads in flight per wave
//   mode 2  global_load_dwordx4 -> VGPR only (the L2 -> CU path without the LDS store)
// Every workgroup streams `kb` KB per iteration from a 3 MB buffer shared by all workgroups (weights of one layer), `iters` times.
// Prints bytes / clk / CU at 2.4 GHz for 1 workgroup and for one workgroup on every CU, with 4 and 8 waves.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/lds_fill_rate tools/microbench/lds_fill_rate.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void fill_kernel(const char* src, int bytes_per_iter, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const unsigned ring_base = (unsigned)(uintptr_t)(lds_u8*)ring;
  const int pieces = bytes_per_iter / 1024 / nw;                 // per wave per iteration
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    const char* base = src + ((size_t)((it * 5 + blockIdx.x) % 12) * 262144);   // walk over the 3 MB
    for (int p0 = 0; p0 < pieces; p0 += DEPTH) {
      if constexpr (MODE == 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int piece = (p0 + d) * nw + w;
          const char* s = base + (size_t)piece * 1024;
          const unsigned dst = ring_base + ((piece * 1024) & 0x1ffff);
          const unsigned voff = lane * 16;
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(dst), "s"(s) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        f32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int piece = (p0 + d) * nw + w;
          v[d] = *(const f32x4*)(base + (size_t)piece * 1024 + lane * 16);
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int piece = (p0 + d) * nw + w;
          if constexpr (MODE == 1) *(f32x4*)(ring + ((piece * 1024) & 0x1ffff) + lane * 16) = v[d];
          else acc += v[d];
        }
      }
    }
    __syncthreads();
  }
  if (MODE != 2) acc += *(const f32x4*)(ring + lane * 16);
  if (acc[0] == 123.456f) sink[0] = acc[1] + acc[2] + acc[3];
}

template <int MODE, int DEPTH>
void run(const char* name, const char* src, float* sink, int threads, int grid) {
  const int kb = 256, iters = 200;
  (void)hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(grid), dim3(threads), 128 * 1024, 0, src, kb * 1024, 20, sink);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(grid), dim3(threads), 128 * 1024, 0, src, kb * 1024, iters, sink);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)kb * 1024 * iters;
  printf("%-34s depth %2d  waves %d  grid %3d: %6.1f B/clk/CU  (%.3f ms)\n", name, DEPTH, threads / 64, grid, bytes / (ms * 1e-3 * 2.4e9), ms);
}

int main() {
  char* src;
  float* sink;
  (void)hipMalloc(&src, 4 << 20);
  (void)hipMemset(src, 1, 4 << 20);
  (void)hipMalloc(&sink, 64);
  for (int grid : {1, 256}) {
    for (int threads : {256, 512}) {
      run<0, 2>("LDS-DMA (global_load_lds_dwordx4)", src, sink, threads, grid);
      run<0, 4>("LDS-DMA (global_load_lds_dwordx4)", src, sink, threads, grid);
      run<0, 8>("LDS-DMA (global_load_lds_dwordx4)", src, sink, threads, grid);
      run<1, 2>("load -> VGPR -> ds_write_b128", src, sink, threads, grid);
      run<1, 4>("load -> VGPR -> ds_write_b128", src, sink, threads, grid);
      run<1, 8>("load -> VGPR -> ds_write_b128", src, sink, threads, grid);
      run<2, 4>("load -> VGPR", src, sink, threads, grid);
      run<2, 8>("load -> VGPR", src, sink, threads, grid);
    }
  }
  return 0;
}
