// Micro-benchmark of ffn_rows_f16x3_kernel (ffn_rows_f16x3.hip next to this file: the round-5 row-stationary experiment) on synthetic rows: times `iters` launches of one mode / tail
// combination over M rows and prints us per 128-row tile per CU.  Experiment knobs are compile-time (-DRS_EXP_...): timing only, the
// numbers a knob produces are wrong.  Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ivap-realtime_amd/csrc [-DRS_EXP_NO_DMA ...] -o tools/microbench/ffn_rows_bench tools/microbench/ffn_rows_bench.hip
#include "ffn_rows_f16x3.hip"

#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <vector>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096 * 2 * 50;
  const int mode = argc > 2 ? atoi(argv[2]) : 1;
  const int tail = argc > 3 ? atoi(argv[3]) : 1;
  const int iters = argc > 4 ? atoi(argv[4]) : 10;
  auto dalloc = [](size_t n) { float* p; if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) { printf("alloc failed\n"); exit(1); } return p; };
  std::vector<float> hx((size_t)M * 256);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
  float *att = dalloc((size_t)M * 256), *resid = dalloc((size_t)M * 256), *xout = dalloc((size_t)M * 256), *kvx = dalloc((size_t)M * 512),
        *qkv = dalloc((size_t)M * 768), *xn = dalloc((size_t)M * 256);
  hipMemcpy(att, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(resid, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  std::vector<_Float16> hw((size_t)12 * 65536 * 2);
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (_Float16)(((int)(s >> 8) % 2001 - 1000) * 1e-2f); }
  float* wrs = dalloc((size_t)12 * 65536);
  hipMemcpy(wrs, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  std::vector<float> ones(256, 1.0f), zeros(256, 0.0f);
  float *gam = dalloc(256), *bet = dalloc(256);
  hipMemcpy(gam, ones.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(bet, zeros.data(), 1024, hipMemcpyHostToDevice);
  RowsArgs a;
  memset(&a, 0, sizeof a);
  a.xmid = resid; a.lnf_g = gam; a.lnf_b = bet; a.xout = xout; a.ln_g = gam; a.ln_b = bet; a.M = M; a.hid_scale = 1.0f;
  a.mode = mode; a.att = att; a.resid = resid; a.wrs = wrs; a.wprojf = wrs;
  if (tail) { a.wqkvf = wrs; a.wkvxf = wrs; a.qkv = qkv; a.kvx = kvx; a.n_qkv_chunks = 3; } else { a.xn_out = xn; }
#ifdef RS_TRACE
  unsigned long long* tb;
  hipMalloc(&tb, (size_t)256 * 64 * 32 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(rs_trace_buf), &tb, sizeof tb);
#endif
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) if (launch_ffn_rows_f16x3(a, 0) != hipSuccess) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch_ffn_rows_f16x3(a, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  const int tiles = (M + 127) / 128, cus = device_cu_count();
  const int rounds = (tiles + cus - 1) / cus;
  const int ncontr = (mode == 1 ? 1 : 0) + 6 + (tail ? 5 : 0);
  const double flop = 2.0 * M * 65536.0 * ncontr;
  printf("M=%d mode=%d tail=%d: %.3f ms per launch, %.1f us per tile (%d rounds), %.1f TF (split roof 838.9), %d contractions: %.2f us each\n", M, mode, tail, ms,
         ms * 1e3 / rounds, rounds, flop / ms * 1e-9, ncontr, ms * 1e3 / rounds / ncontr);
#ifdef RS_TRACE
  {   // timeline: median phase durations over all (workgroup, tile) pairs of ONE launch
    const int cusn = std::min(tiles, cus), per = std::min(64, rounds);
    hipMemset(tb, 0, (size_t)256 * 64 * 32 * 8);
    launch_ffn_rows_f16x3(a, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)cusn * 64 * 32);
    hipMemcpy(h.data(), tb, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<std::vector<double>> d(32);
    std::vector<double> tot, gap, rate;
    for (int b = 0; b < cusn; ++b)
      for (int t = 0; t < per - 1; ++t) {
        const unsigned long long* r = &h[((size_t)b * 64 + t) * 32];
        int n = 0;
        while (n < 30 && r[n]) ++n;
        if (n < 2) continue;
        for (int k = 1; k < n; ++k) d[k].push_back((double)(r[k] - r[k - 1]));
        tot.push_back((double)(r[n - 1] - r[0]));
        if (r[31] > r[30]) rate.push_back((double)(r[n - 1] - r[0]) / ((double)(r[31] - r[30]) / 100.0));   // s_memrealtime: 100 MHz
        const unsigned long long* nx = &h[((size_t)b * 64 + t + 1) * 32];
        if (nx[0]) gap.push_back((double)(nx[0] - r[n - 1]));
      }
    auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double tick_per_us = med(rate);
    printf("  timeline (median over %zu tiles, us; s_memtime at %.0f ticks / us):", tot.size(), tick_per_us);
    for (int k = 1; k < 32; ++k) if (!d[k].empty()) printf(" [%d] %.2f", k, med(d[k]) / tick_per_us);
    printf("  | tile %.2f, gap to next tile %.2f\n", med(tot) / tick_per_us, med(gap) / tick_per_us);
  }
#endif
  float chk[4];
  hipMemcpy(chk, xout, 16, hipMemcpyDeviceToHost);
  printf("  xout[0..3] = %g %g %g %g\n", chk[0], chk[1], chk[2], chk[3]);
  return 0;
}
