// Micro-benchmark: what does a per-step exchange between the workgroups of a small cluster cost on MI355X?
// C workgroups per cluster (consecutive blockIdx), each step: every workgroup writes a 1 KB slice of a row block, signals an
// agent-scope counter, waits until all C slices of the step are there, reads the whole block back.  Bounded spin (no hang).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench/xwg_sync tools/microbench/xwg_sync.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void xwg_kernel(float* xbuf, int* flags, int C, int steps, float* out, int* timeouts) {
  const int cluster = blockIdx.x / C, c = blockIdx.x % C;
  const int tid = threadIdx.x;
  float acc = 0.f;
  for (int t = 0; t < steps; ++t) {
    float* blk = xbuf + ((long)cluster * steps + t) * C * 256;       // fresh addresses every step: never cached stale
    blk[c * 256 + tid] = (float)(t + c) + acc * 1e-9f;
    __syncthreads();
    if (tid == 0) {
      __threadfence();                                                  // release: publish this workgroup's slice
      __hip_atomic_fetch_add(&flags[cluster * steps + t], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(&flags[cluster * steps + t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < C) {
        if (++spins > (1 << 22)) { atomicAdd(timeouts, 1); break; }
        __builtin_amdgcn_s_sleep(1);
      }
      __threadfence();
    }
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < C; ++i) s += __builtin_nontemporal_load(&blk[i * 256 + tid]);
    acc += s;
  }
  out[blockIdx.x * 256 + tid] = acc;
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 4, clusters = argc > 2 ? atoi(argv[2]) : 32, steps = argc > 3 ? atoi(argv[3]) : 200;
  float *xbuf, *out; int *flags, *timeouts;
  hipMalloc(&xbuf, sizeof(float) * (size_t)clusters * steps * C * 256);
  hipMalloc(&out, sizeof(float) * clusters * C * 256);
  hipMalloc(&flags, sizeof(int) * clusters * steps);
  hipMalloc(&timeouts, sizeof(int));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipMemset(flags, 0, sizeof(int) * clusters * steps); hipMemset(timeouts, 0, sizeof(int));
    hipEventRecord(a);
    hipLaunchKernelGGL(xwg_kernel, dim3(clusters * C), dim3(256), 0, 0, xbuf, flags, C, steps, out, timeouts);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  int to = 0; hipMemcpy(&to, timeouts, sizeof(int), hipMemcpyDeviceToHost);
  std::vector<float> h(clusters * C * 256); hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
  double expect = 0; for (int t = 0; t < steps; ++t) for (int i = 0; i < C; ++i) expect += t + i;
  printf("C=%d clusters=%d steps=%d: %.2f us per step (kernel %.3f ms), timeouts=%d, value %.1f (expect ~%.1f)\n", C, clusters, steps,
         best * 1e3 / steps, best, to, h[0], expect);
  return 0;
}
