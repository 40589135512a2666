// fp32-in / fp32-accumulate MFMA GEMM for every dense contraction of the VAP step
// (conv1-4 as implicit GEMM, LSTM input projection, downsample, QKV / proj / FFN, combinator).
//
//   C[m][n] = epilogue( sum_k A[m][k] * W[n][k] )
//
// Design (gfx950):
//   * v_mfma_f32_32x32x2_f32: exact f32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD = 157 TF peak.
//   * workgroup = 4 waves; tile = (32*WM) rows x 256 columns, WM*WN = 4.  A wave owns 32 rows x
//     (256/WN) columns = 8/WN accumulators of 32x32.  N = 256 is the model width, so with a
//     256-wide tile a whole output row lives in one workgroup and LayerNorm / ChannelNorm /
//     residual epilogues fuse into the GEMM.  Small-M layers (conv3/4, downsample) use WM=1 so the
//     grid still covers the 256 CUs.
//   * both operands are K-contiguous, so A and W tiles are staged [rows][32 k] (+4 pad -> 144 B
//     row stride, conflict-free for ds_read_b128) and each lane fetches its MFMA operands for four
//     consecutive k-pairs with ONE 16-byte LDS read: lane half h (= lane>>5) takes k = 8c+4h..8c+4h+3
//     of chunk c — a permutation of the k order that both operands share, which a dot product
//     does not care about.
//   * global -> register prefetch of tile t+1 is issued before the MFMAs of tile t, then written
//     to LDS after them (single LDS buffer, 55 KB at BM=128 -> 2 workgroups per CU).
#include "gemm_f32.h"

namespace {

constexpr int LDT = 36;  // LDS row stride in floats (32 + 4 pad)
// SPLIT variant (opt-in, GemmArgs::split): both operands are split into f16 (hi, lo) pairs while they are staged
// (x = hi + lo; weights as 2^8 w so that lo stays a normal f16, undone exactly in the epilogue) and every 16-wide
// k-chunk becomes three v_mfma_f32_32x32x16_f16: hi.hi + lo.hi + hi.lo, fp32 accumulate — fp32-accurate (dropped
// term 2^-22 relative), 3/16 of the MFMA time.  Tiles are [rows][32 k] halves with a 40-half (80 B) row stride.
constexpr int LDT16 = 40;
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int WM, int WN, int EPI, bool SPLIT = false>
__global__ __launch_bounds__(256, (WM == 4 ? 2 : 2)) void gemm_f32_kernel(const GemmArgs g) {
  constexpr int BM = 32 * WM;
  constexpr int NS = 8 / WN;    // 32-column sub-tiles per wave
  constexpr int NW = 256 / WN;  // columns per wave
  __shared__ __attribute__((aligned(16))) float lds[(BM + 256) * (SPLIT ? LDT16 : LDT)];
  float* sA = lds;
  float* sB = lds + BM * LDT;
  _Float16* sAh = (_Float16*)lds;             // SPLIT: hi / lo tiles of A, then of W
  _Float16* sAl = sAh + BM * LDT16;
  _Float16* sBh = sAl + BM * LDT16;
  _Float16* sBl = sBh + 256 * LDT16;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = g.N >> 8;
  const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
  const int m0 = mt * BM, n0 = nt * 256;

  // ---- staging coordinates: 8 threads cover one 128-byte k-row ----
  // SPLIT: a staged row is 32 halves at an 80-byte stride and leaves as 8-byte stores; a ds_write_b64 is served in groups of 16 lanes over
  // 32 banks, i.e. TWO rows per group, and rows r, r + 1 overlap in four banks (20 dwords apart: 28 % of the LDS-active cycles of
  // gemm_f32_kernel<4, 1, 4, true> were conflicts, profiles/r05_c3_split_pmc.txt).  Rows r and r + 4 are 80 dwords = 16 banks apart: the
  // thread -> row map pairs those (which 8 threads fetch which 128-byte k-row from global is free).
  const int srow = SPLIT ? ((tid >> 4) & 3) + 4 * ((tid >> 3) & 1) + 8 * (tid >> 6) : tid >> 3, skq = (tid & 7) * 4;
  const float* aptr[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    int m = m0 + i * 32 + srow;
    m = m < g.M ? m : g.M - 1;
    aptr[i] = g.A + row_off(g.am, m) + skq;
  }
  const bool pre16 = SPLIT && g.W16 != nullptr;           // weights arrive pre-split: same addressing, no conversion
  const float* bptr = (pre16 ? g.W16 : g.W) + (long)(n0 + srow) * g.K + skq;
  const long bstep = (long)32 * g.K;

  f32x4 ra[WM], rb[8];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < WM; ++i) ra[i] = *(const f32x4*)(aptr[i] + k0);
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = *(const f32x4*)(bptr + i * bstep + k0);
  };
  auto sstore = [&]() {
    if constexpr (SPLIT) {
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const f16x4 hi = __builtin_convertvector(ra[i], f16x4);
        const f16x4 lo = __builtin_convertvector(ra[i] - __builtin_convertvector(hi, f32x4), f16x4);
        *(f16x4*)&sAh[(i * 32 + srow) * LDT16 + skq] = hi;
        *(f16x4*)&sAl[(i * 32 + srow) * LDT16 + skq] = lo;
      }
      if (pre16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f16x8 hl = __builtin_bit_cast(f16x8, rb[i]);     // [hi x 4 | lo x 4]
          *(f16x4*)&sBh[(i * 32 + srow) * LDT16 + skq] = __builtin_shufflevector(hl, hl, 0, 1, 2, 3);
          *(f16x4*)&sBl[(i * 32 + srow) * LDT16 + skq] = __builtin_shufflevector(hl, hl, 4, 5, 6, 7);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const f32x4 wv = rb[i] * 256.0f;
          const f16x4 hi = __builtin_convertvector(wv, f16x4);
          const f16x4 lo = __builtin_convertvector(wv - __builtin_convertvector(hi, f32x4), f16x4);
          *(f16x4*)&sBh[(i * 32 + srow) * LDT16 + skq] = hi;
          *(f16x4*)&sBl[(i * 32 + srow) * LDT16 + skq] = lo;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < WM; ++i) *(f32x4*)&sA[(i * 32 + srow) * LDT + skq] = ra[i];
#pragma unroll
      for (int i = 0; i < 8; ++i) *(f32x4*)&sB[(i * 32 + srow) * LDT + skq] = rb[i];
    }
  };

  f32x16 acc[NS];
#pragma unroll
  for (int ns = 0; ns < NS; ++ns)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ns][r] = 0.f;

  const int l31 = lane & 31, kh = (lane >> 5) * 4;
  const float* pa = &sA[(wm * 32 + l31) * LDT + kh];
  const float* pb = &sB[(wn * NW + l31) * LDT + kh];

  const int nk = g.K >> 5;
  gload(0);
  sstore();
  __syncthreads();
  const int k8 = (lane >> 5) * 8;
  const _Float16* pah = &sAh[(wm * 32 + l31) * LDT16 + k8];
  const _Float16* pal = &sAl[(wm * 32 + l31) * LDT16 + k8];
  const _Float16* pbh = &sBh[(wn * NW + l31) * LDT16 + k8];
  const _Float16* pbl = &sBl[(wn * NW + l31) * LDT16 + k8];
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) gload((kt + 1) << 5);
    if constexpr (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const f16x8 ah = *(const f16x8*)(pah + ks * 16), al = *(const f16x8*)(pal + ks * 16);
        f16x8 bh[NS], bl[NS];
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          bh[ns] = *(const f16x8*)(pbh + ns * 32 * LDT16 + ks * 16);
          bl[ns] = *(const f16x8*)(pbl + ns * 32 * LDT16 + ks * 16);
        }
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ns] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ns], acc[ns], 0, 0, 0);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ns] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ns], acc[ns], 0, 0, 0);
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) acc[ns] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ns], acc[ns], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      f32x4 a = *(const f32x4*)(pa + kc * 8);
      f32x4 b[NS];
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) b[ns] = *(const f32x4*)(pb + ns * 32 * LDT + kc * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
          acc[ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[ns][s], acc[ns], 0, 0, 0);
    }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      sstore();
      __syncthreads();
    }
  }

  // ---- epilogue ----
  if constexpr (SPLIT) {
#pragma unroll
    for (int ns = 0; ns < NS; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ns][r] *= (1.0f / 256.0f);
  }
  // accumulator element (ns, r): row lr = (r&3) + 8*(r>>2) + 4*(lane>>5), col = ns*32 + (lane&31)
  const int cbase = n0 + wn * NW + l31;
  const int rbase = m0 + wm * 32 + 4 * (lane >> 5);
  float* red = lds;  // [WN][BM] partial row sums (the K loop ended with a barrier)

  auto row_allreduce = [&](float(&v)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = half_sum(v[r]);
    if constexpr (WN > 1) {
      if (l31 == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wn * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = v[r];
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) s += red[w * BM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
        v[r] = s;
      }
      __syncthreads();
    }
  };

  if constexpr (EPI == EPI_STORE || EPI == EPI_CN_RELU || EPI == EPI_BIAS_LN_GELU) {
    if (g.bias) {
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        float bv = g.bias[cbase + ns * 32];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ns][r] += bv;
      }
    }
  }
  if constexpr (EPI == EPI_RESID || EPI == EPI_RESID_LN) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // clamp instead of branching: a conditional element update makes hipcc copy the whole
      // accumulator vector per row (thousands of spills)
      int m = rbase + (r & 3) + 8 * (r >> 2);
      m = m < g.M ? m : g.M - 1;
      const float* rp = g.resid + row_off(g.rm, m) + cbase;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) acc[ns][r] += rp[ns * 32];
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the loads in flight (registers)
    }
  }

  // plain store of acc (all epilogues except the ones that normalise in place)
  if constexpr (EPI == EPI_STORE || EPI == EPI_GELU || EPI == EPI_RESID || EPI == EPI_RESID_LN) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = rbase + (r & 3) + 8 * (r >> 2);
      if (m < g.M) {
        float* cp = g.C + row_off(g.cm, m) + cbase;
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          if constexpr (EPI == EPI_GELU) cp[ns * 32] = gelu_erf(acc[ns][r]);
          else cp[ns * 32] = acc[ns][r];
        }
      }
      if constexpr (EPI == EPI_GELU) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (EPI == EPI_RESID_LN) __builtin_amdgcn_sched_barrier(0);
  }

  if constexpr (EPI == EPI_RESID_LN || EPI == EPI_CN_RELU || EPI == EPI_BIAS_LN_GELU) {
    // two-pass mean / variance over the 256 columns of each row
    float s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float t = 0.f;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) t += acc[ns][r];
      s[r] = t;
    }
    row_allreduce(s);
    float mean[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      mean[r] = s[r] * (1.0f / 256.0f);
      float t = 0.f;
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) {
        float d = acc[ns][r] - mean[r];
        t += d * d;
      }
      s[r] = t;
    }
    row_allreduce(s);
    constexpr float inv_n = (EPI == EPI_CN_RELU) ? (1.0f / 255.0f) : (1.0f / 256.0f);
    float gam[NS], bet[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      gam[ns] = g.gamma[cbase - n0 + ns * 32];
      bet[ns] = g.beta[cbase - n0 + ns * 32];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = rbase + (r & 3) + 8 * (r >> 2);
      float rstd = rsqrtf(s[r] * inv_n + 1e-5f);
      if (m < g.M) {
        float* op = (EPI == EPI_RESID_LN ? g.C2 + row_off(g.c2m, m) : g.C + row_off(g.cm, m)) + cbase;
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          float y = (acc[ns][r] - mean[r]) * rstd * gam[ns] + bet[ns];
          if constexpr (EPI == EPI_CN_RELU) y = relu_nanprop(y);
          if constexpr (EPI == EPI_BIAS_LN_GELU) y = gelu_erf(y);
          op[ns * 32] = y;
        }
      }
    }
  }
}

template <int WM, int WN, bool SPLIT>
hipError_t launch_wm(const GemmArgs& g, int epi, hipStream_t st) {
  constexpr int BM = 32 * WM;
  dim3 grid(((g.M + BM - 1) / BM) * (g.N >> 8)), block(256);
  switch (epi) {
    case EPI_STORE: hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, EPI_STORE, SPLIT>), grid, block, 0, st, g); break;
    case EPI_GELU: hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, EPI_GELU, SPLIT>), grid, block, 0, st, g); break;
    case EPI_RESID: hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, EPI_RESID, SPLIT>), grid, block, 0, st, g); break;
    case EPI_RESID_LN: hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, EPI_RESID_LN, SPLIT>), grid, block, 0, st, g); break;
    case EPI_CN_RELU: hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, EPI_CN_RELU, SPLIT>), grid, block, 0, st, g); break;
    case EPI_BIAS_LN_GELU: hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, EPI_BIAS_LN_GELU, SPLIT>), grid, block, 0, st, g); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm_f32(const GemmArgs& g, int epi, int tile_rows, hipStream_t stream) {
  if (g.M <= 0) return hipSuccess;
  if ((g.N & 255) || (g.K & 31) || g.K <= 0) return hipErrorInvalidValue;
  if ((epi == EPI_RESID_LN || epi == EPI_CN_RELU || epi == EPI_BIAS_LN_GELU) && g.N != 256) return hipErrorInvalidValue;
  if (tile_rows == 0) {
    // measured on MI355X (tools/gemm_sweep.py, profiles/r01_gemm_sweep.txt): 64-row tiles win for
    // N = 256 (more workgroups -> less tile quantisation, 3 waves/SIMD), 128-row tiles for the
    // plain-store wide GEMMs, 32-row tiles when M is too small to fill 256 CUs otherwise.
    if (g.M < 12000) tile_rows = 32;
    else if (g.N >= 512 && epi == EPI_STORE) tile_rows = 128;
    else if (g.K >= 2048 && g.M >= 200000) tile_rows = 128;
    else tile_rows = 64;
  }
  if (g.split) {
    switch (tile_rows) {
      case 128: return launch_wm<4, 1, true>(g, epi, stream);
      case 64: return launch_wm<2, 2, true>(g, epi, stream);
      case 32: return launch_wm<1, 4, true>(g, epi, stream);
      default: return hipErrorInvalidValue;
    }
  }
  switch (tile_rows) {
    case 128: return launch_wm<4, 1, false>(g, epi, stream);
    case 64: return launch_wm<2, 2, false>(g, epi, stream);
    case 32: return launch_wm<1, 4, false>(g, epi, stream);
    default: return hipErrorInvalidValue;
  }
}
