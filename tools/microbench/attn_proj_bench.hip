// Micro-benchmark of attention_proj_f16x3_kernel (csrc/attention_proj_f16x3.hip) on synthetic LN rows: times the launch for B streams at window T
// and, built with -DVAPX_TRACE, prints the median phase timeline of one wave over all items.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ivap-realtime_amd/csrc [-DVAPX_TRACE [-DAP_TRACE_WAVE=k]] -o tools/microbench/attn_proj_bench tools/microbench/attn_proj_bench.hip
#include "../../vap-realtime_amd/csrc/attention_proj_f16x3.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, T = argc > 2 ? atoi(argv[2]) : 250, iters = argc > 3 ? atoi(argv[3]) : 5;
  const size_t rows = (size_t)B * 2 * T;
  float *xn, *out, *wq;
  int* bn;
  hipMalloc(&xn, rows * 256 * 4); hipMalloc(&out, rows * 256 * 4); hipMalloc(&wq, 3 * 65536 * 4); hipMalloc(&bn, B * 4);
  std::vector<float> hx(rows * 256);
  unsigned s = 99;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 2e-3f; }
  hipMemcpy(xn, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  std::vector<_Float16> hw((size_t)3 * 65536 * 2);
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 8) % 2001 - 1000) * 1e-2f); }
  hipMemcpy(wq, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  std::vector<int> hn(B, T);
  hipMemcpy(bn, hn.data(), B * 4, hipMemcpyHostToDevice);
  AttnProjArgs a{xn, wq, out, bn, T, 0};
#ifdef VAPX_TRACE
#ifndef AP_TRACE_WAVE
#define AP_TRACE_WAVE 0
#endif
  unsigned long long* tb;
  hipMalloc(&tb, (size_t)256 * 128 * 16 * 8);
  hipMemset(tb, 0, (size_t)256 * 128 * 16 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(ap_trace_buf), &tb, sizeof tb);
  const int tw = AP_TRACE_WAVE;
  hipMemcpyToSymbol(HIP_SYMBOL(ap_trace_wave), &tw, sizeof tw);
#endif
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) if (launch_attention_proj_f16x3(a, B, 0) != hipSuccess) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch_attention_proj_f16x3(a, B, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  const int items = B * 8, per_cu = (items + 255) / 256;
  printf("B=%d T=%d: %.3f ms per launch, %.2f us per (stream, channel, head) item per CU\n", B, T, ms, ms * 1e3 / per_cu);
#ifdef VAPX_TRACE
  {
    hipMemset(tb, 0, (size_t)256 * 128 * 16 * 8);
    launch_attention_proj_f16x3(a, B, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)256 * 128 * 16);
    hipMemcpy(h.data(), tb, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<std::vector<double>> d(16);
    std::vector<double> tot;
    for (int b = 0; b < 256; ++b)
      for (int t = 1; t < 127; ++t) {
        const unsigned long long* r = &h[((size_t)b * 128 + t) * 16];
        if (!r[0] || !r[7]) continue;
        for (int k = 1; k < 8; ++k) d[k].push_back((double)(r[k] - r[k - 1]));
        tot.push_back((double)(r[7] - r[0]));
      }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double tpu = 100.0;   // s_memtime runs at 100 MHz on this part? calibrated below against the launch time
    const double item_us = ms * 1e3 / per_cu;
    const double scale = item_us / (med(tot) > 0 ? med(tot) : 1.0);
    printf("  timeline of wave %d (us, scaled so that the stamped span = %.2f us per item; ticks per item %.0f):", AP_TRACE_WAVE, item_us, med(tot));
    const char* nm[8] = {"", "barrier A", "slot 0", "slots 1-7", "barrier B", "K/V/Q -> LDS + stores", "barrier C", "attention"};
    for (int k = 1; k < 8; ++k) printf(" [%s] %.2f", nm[k], med(d[k]) * scale);
    printf("\n");
    (void)tpu;
  }
#endif
  return 0;
}
