// ffn_block_f16x3_kernel: the fused flat-row blocks of fused_blocks.hip (FFN block, and for long windows the attention-projection
// modes 1 / 2) with every 256x256 contraction computed as an fp32-accurate SPLIT-PRECISION product on the f16 matrix cores (opt-in:
// VAPX_FLAG_SPLIT_F16, off by default).
//
//   x = x_hi + x_lo,  x_hi = f16(x),  x_lo = f16(x - x_hi)          (activations; 22 significant bits)
//   w' = 2^8 w = w_hi + w_lo                                        (weights, split once at pack time; the power-of-two
//                                                                    pre-scale keeps w_lo a NORMAL f16, undone exactly)
//   x.w = 2^-8 (x_hi.w_hi + x_lo.w_hi + x_hi.w_lo)  + O(2^-22 |x||w|)
// Each f16 product is exact in the MFMA's fp32 accumulator, so the result differs from an fp32 GEMM only in the
// dropped x_lo.w_lo term and the summation order: measured error vs float64 equals the fp32 path's
// (tools/split_precision_error.py, tests/test_split_precision_gpu.py).  Three v_mfma_f32_32x32x16_f16 (8 passes,
// K = 16) replace eight v_mfma_f32_32x32x2_f32 (16 passes, K = 2 each): 3/16 of the MFMA time for the same operand bytes.
//
// NO OPERAND CAN OVERFLOW f16 (|x| >= 65504), whatever the input — every A operand is one of
//   * a LayerNorm output: |y| <= 16 |gamma| + |beta| by construction;
//   * the GELU hidden row: bounded by the weights alone (its input is a LayerNorm output); weights.pack_blob computes the bound and
//     a static power-of-two scale `hid_scale` <= 1 per layer (1 for every sane checkpoint), undone exactly in the accumulator;
//   * a RAW residual-stream / attention-output row, whose magnitude the input decides: scaled per ROW by a power of two s_row so
//     that |x s_row| < 2^14 — from the row's own sum of squares (already needed for the next LayerNorm: no extra reduction) or,
//     where whole rows are staged by one wave, from the row maximum — and the accumulator row is multiplied by 1 / s_row (exact).
// Power-of-two scales commute with every rounding here, so the scaled product equals the unscaled one bit for bit wherever the
// latter does not overflow (tests/test_split_precision_gpu.py::test_split_f16_handles_activations_far_beyond_the_f16_range).
//
// Same structure as ffn_block_kernel<1, MODE>: 32-row tile, 4 waves x 64 columns, LN_ffn on load, hidden row in LDS, next
// layer's projections from the on-chip tile.  Differences: the A operands sit in LDS as f16 (hi, lo) row-major pairs
// ([32][264] halves each: one ds_read_b128 = the 8 k-values of a lane), weights stream from L2 as f16 fragment pairs
// ([4 w][16 kc][2 ns][2 hi/lo][64 lane][8 halves], weights.frag_pack_f16x3), C/D layout is the fp32 one (dtype
// independent on gfx950).  A and B use the same lane -> k mapping, so the k order inside a 16-chunk is immaterial.
#include "fused_blocks.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int LD16 = 264;                  // halves per LDS row: 256 + 8 pad (528 B: conflict-free 16-byte row reads)
constexpr float kWScaleInv = 1.0f / 256.0f;   // weights are packed as 2^8 w

__device__ __forceinline__ void split_store(_Float16* hi, _Float16* lo, int idx, float v) {
  const _Float16 h = (_Float16)v;
  hi[idx] = h;
  lo[idx] = (_Float16)(v - (float)h);
}

// MODE as FfnArgs::mode (0: xmid from global; 1: attention-output projection + the whole block; 2: projection + LN + wqkvf chunks only)
template <int MODE>
__global__ __launch_bounds__(256, 2) void ffn_block_f16x3_kernel(const FfnArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int BM = 32;
  _Float16* sXh = (_Float16*)lds_raw;     // LN_ffn(x) tile, hi / lo
  _Float16* sXl = sXh + BM * LD16;
  _Float16* sHh = sXl + BM * LD16;        // attention rows / gelu chunk / raw x / LN rows, hi / lo
  _Float16* sHl = sHh + BM * LD16;
  float* red = (float*)(sHl + BM * LD16); // [4][BM] row partials
  float* rinv = red + 4 * BM;             // [BM] 1 / s_row of the staged attention rows (modes 1, 2)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.x * BM;

  // weight fragments: ring of 4 k-chunks (16 x 16-byte fragments) ahead, running on into the next unit
  f32x4 ring[16];
  auto wbase = [&](const float* wfrag) { return (const f32x4*)wfrag + (long)w * 16 * 4 * 64; };   // wave-uniform
  auto fetch = [&](const float* wfrag) {
    const f32x4* wf = wbase(wfrag);
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[i] = wf[i * 64 + lane];
  };
  fetch(MODE == 0 ? g.w0f : g.wprojf);   // the first weight fragments fly while the tile is staged

  if constexpr (MODE == 0) {   // A operand of FFN1 = LayerNorm(xmid; ln_ffn), normalised and split while the tile is staged
    const f32x4 lg = *(const f32x4*)(g.lnf_g + lane * 4), lb = *(const f32x4*)(g.lnf_b + lane * 4);
    f32x4 xr[BM / 4];
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) {
      int m = m0 + (tid >> 6) + 4 * k;
      m = m < g.M ? m : g.M - 1;
      xr[k] = *(const f32x4*)(g.xmid + (long)m * 256 + lane * 4);
    }
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) {
      float sm = half_sum(xr[k][0] + xr[k][1] + xr[k][2] + xr[k][3]);
      sm += __shfl_xor(sm, 32);
      const float mean = sm * (1.0f / 256.0f);
      f32x4 d = xr[k] - mean;
      float sv = half_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
      sv += __shfl_xor(sv, 32);
      const float rstd = rsqrtf(sv * (1.0f / 256.0f) + 1e-5f);
      const f32x4 y = d * rstd * lg + lb;
      const int base = ((tid >> 6) + 4 * k) * LD16 + lane * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) split_store(sXh, sXl, base + e, y[e]);
    }
  } else {   // raw attention rows -> sH (A operand of the output projection): a wave stages WHOLE rows, so the row maximum is one
             // wave reduction; the row goes in as x * s_row (|.| < 2^14), 1 / s_row is kept for the accumulator rows
    f32x4 xr[BM / 4];
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) {
      int m = m0 + (tid >> 6) + 4 * k;
      m = m < g.M ? m : g.M - 1;
      xr[k] = *(const f32x4*)(g.att + (long)m * 256 + lane * 4);
    }
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) {
      const float mx = wave_max(fmaxf(fmaxf(fabsf(xr[k][0]), fabsf(xr[k][1])), fmaxf(fabsf(xr[k][2]), fabsf(xr[k][3]))));
      const float s = pow2_scale_for(mx);
      const int row = (tid >> 6) + 4 * k;
      if (lane == 0) rinv[row] = __builtin_amdgcn_rcpf(s);     // (exact: s is a power of two)
      const int base = row * LD16 + lane * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) split_store(sHh, sHl, base + e, xr[k][e] * s);
    }
  }
  __syncthreads();

  auto mm = [&](f32x16(&acc)[2], const _Float16* Ah, const _Float16* Al, const float* wfrag, const float* next_wfrag) {
    const _Float16* pah = Ah + l31 * LD16 + hi * 8;
    const _Float16* pal = Al + l31 * LD16 + hi * 8;
    const f32x4* wf = wbase(wfrag);
    const f32x4* wnext = next_wfrag ? wbase(next_wfrag) : wf;
    __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
      const f32x4* nx = blk < 3 ? wf + (blk + 1) * 16 * 64 : wnext;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int kc = blk * 4 + k4;
        const f16x8 ah = *(const f16x8*)(pah + kc * 16), al = *(const f16x8*)(pal + kc * 16);
        const f16x8 b0h = __builtin_bit_cast(f16x8, ring[k4 * 4 + 0]), b0l = __builtin_bit_cast(f16x8, ring[k4 * 4 + 1]);
        const f16x8 b1h = __builtin_bit_cast(f16x8, ring[k4 * 4 + 2]), b1l = __builtin_bit_cast(f16x8, ring[k4 * 4 + 3]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, acc[1], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[k4 * 4 + i] = nx[(k4 * 4 + i) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);   // keep the refills behind their MFMAs
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto zero = [](f32x16(&acc)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  };
  // accumulator (ns, r) <-> tile row lr = (r&3) + 8*(r>>2) + 4*hi, chunk column w*64 + ns*32 + l31
  const int ccol = w * 64 + l31;
  auto store_global = [&](const f32x16(&acc)[2], float* base, int ld, int col0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (m < g.M) {
        float* p = base + (long)m * ld + col0 + ccol;
        p[0] = acc[0][r];
        p[32] = acc[1][r];
      }
    }
  };
  // Row statistics of the tile held in accumulator layout (two-pass, partials across the 4 waves via `red`): mean and variance
  auto row_stats = [&](const f32x16(&v)[2], float (&mean)[16], float (&var)[16]) {
    float s[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = half_sum(v[0][r] + v[1][r]);
    __syncthreads();          // every wave is done reading `red` / the LDS tile of the previous phase
    if (l31 == 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[w * BM + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      mean[r] = (red[lr] + red[BM + lr] + red[2 * BM + lr] + red[3 * BM + lr]) * (1.0f / 256.0f);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float d0 = v[0][r] - mean[r], d1 = v[1][r] - mean[r];
      s[r] = half_sum(d0 * d0 + d1 * d1);
    }
    if (l31 == 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[w * BM + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      var[r] = (red[lr] + red[BM + lr] + red[2 * BM + lr] + red[3 * BM + lr]) * (1.0f / 256.0f);
    }
  };
  // LayerNorm rows (given their statistics) -> LDS tile (hi / lo) and optionally global
  auto ln_write = [&](const f32x16(&v)[2], const float (&mean)[16], const float (&var)[16], const float* gam, const float* bet,
                      _Float16* dh, _Float16* dl, float* gout) {
    const float g0 = gam[ccol], g1 = gam[ccol + 32], b0 = bet[ccol], b1 = bet[ccol + 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const float rstd = rsqrtf(var[r] + 1e-5f);
      const float y0 = (v[0][r] - mean[r]) * rstd * g0 + b0;
      const float y1 = (v[1][r] - mean[r]) * rstd * g1 + b1;
      split_store(dh, dl, lr * LD16 + ccol, y0);
      split_store(dh, dl, lr * LD16 + ccol + 32, y1);
      if (gout && m0 + lr < g.M) {
        gout[(long)(m0 + lr) * 256 + ccol] = y0;
        gout[(long)(m0 + lr) * 256 + ccol + 32] = y1;
      }
    }
  };

  const int nq = g.wqkvf ? g.n_qkv_chunks : 0;
  const float* after_ffn = g.wkvxf ? g.wkvxf : (nq ? g.wqkvf : nullptr);
  f32x16 out[2];
  if constexpr (MODE != 0) {
    // ---- attention output projection + residual: xmid = resid + att . Wproj^T (the separate GEMM of the long-window path) ----
    zero(out);
    mm(out, sHh, sHl, g.wprojf, MODE == 1 ? g.w0f : after_ffn);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      int m = m0 + lr;
      m = m < g.M ? m : g.M - 1;
      const float* rp;
      if (MODE == 1 && g.resid_rot) {   // layer 0: residual rows straight from the embedding ring
        const int T = g.resid_T;
        const int bc = m / T, i = m - bc * T, b = bc >> 1;
        const long slab = (long)(g.resid_ids ? g.resid_ids[b] : b) * 2 + (bc & 1);
        int rr = i + g.resid_rot[b];
        rr = rr >= T ? rr - T : rr;
        rp = g.resid + (slab * T + rr) * 256 + ccol;
      } else {
        rp = g.resid + (long)m * 256 + ccol;
      }
      const float inv = rinv[lr] * kWScaleInv;     // undo the row scale of the staged attention row and the weights' 2^8
      out[0][r] = out[0][r] * inv + rp[0];
      out[1][r] = out[1][r] * inv + rp[32];
    }
    store_global(out, g.xmid_out, 256, 0);
    if constexpr (MODE == 1) {   // A operand of FFN1 = LayerNorm(xmid; ln_ffn)
      float mean[16], var[16];
      row_stats(out, mean, var);
      ln_write(out, mean, var, g.lnf_g, g.lnf_b, sXh, sXl, nullptr);
      __syncthreads();
    }
  }
  if constexpr (MODE != 2) {
    // ---- feed-forward: x = xmid + gelu(xn W0^T) W3^T, hidden processed in 3 chunks of 256 ----
    // hid_scale: static power of two <= 1 from the weights' bound on |gelu(h)| (weights.pack_blob), 1 for every sane checkpoint
    const float hs = g.hid_scale > 0.f ? g.hid_scale : 1.0f;
    zero(out);
    for (int c = 0; c < 3; ++c) {
      f32x16 hacc[2];
      zero(hacc);
      mm(hacc, sXh, sXl, g.w0f + (long)c * 65536, g.w3f + (long)c * 65536);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hacc[0][r] = gelu_fast(hacc[0][r] * kWScaleInv) * hs;
        hacc[1][r] = gelu_fast(hacc[1][r] * kWScaleInv) * hs;
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();          // every wave is done reading the previous chunk from sH
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
        split_store(sHh, sHl, lr * LD16 + ccol, hacc[0][r]);
        split_store(sHh, sHl, lr * LD16 + ccol + 32, hacc[1][r]);
      }
      __syncthreads();
      mm(out, sHh, sHl, g.w3f + (long)c * 65536, c < 2 ? g.w0f + (long)(c + 1) * 65536 : after_ffn);
    }
    {
      const float inv = kWScaleInv * __builtin_amdgcn_rcpf(hs);
      const float* xm = MODE == 0 ? g.xmid : g.xmid_out;   // (mode 1: this lane's own xmid elements, written above)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        m = m < g.M ? m : g.M - 1;
        const float* rp = xm + (long)m * 256 + ccol;
        out[0][r] = out[0][r] * inv + rp[0];
        out[1][r] = out[1][r] * inv + rp[32];
      }
    }
    store_global(out, g.xout, 256, 0);
  }
  // ---- next layer's projections: cross K,V from the RAW layer output, self Q,K,V from LayerNorm(x) ----
  if (g.wkvxf || nq || g.xn_out) {
    float mean[16], var[16];
    row_stats(out, mean, var);   // (also fences the LDS tile: every wave is done with the previous contraction's reads)
    if constexpr (MODE != 2) {
      if (g.wkvxf) {
        // raw rows: |x| <= sqrt(sum x^2) = 16 sqrt(var + mean^2); scaled per row so that the f16 operand stays below 2^14
        float sinv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float s = pow2_scale_for(16.0f * sqrtf(var[r] + mean[r] * mean[r]) * 1.0001f);
          sinv[r] = __builtin_amdgcn_rcpf(s) * kWScaleInv;
          const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
          split_store(sHh, sHl, lr * LD16 + ccol, out[0][r] * s);
          split_store(sHh, sHl, lr * LD16 + ccol + 32, out[1][r] * s);
        }
        __syncthreads();
        for (int nc = 0; nc < 2; ++nc) {
          f32x16 acc[2];
          zero(acc);
          mm(acc, sHh, sHl, g.wkvxf + (long)nc * 65536, nc == 0 ? g.wkvxf + 65536 : (nq ? g.wqkvf : nullptr));
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[0][r] *= sinv[r]; acc[1][r] *= sinv[r]; }
          store_global(acc, g.kvx, 512, nc * 256);
        }
        __syncthreads();        // every wave is done reading the raw rows: the tile becomes the LayerNorm rows
      }
    }
    ln_write(out, mean, var, g.ln_g, g.ln_b, sHh, sHl, g.xn_out);
    __syncthreads();
    for (int nc = 0; nc < nq; ++nc) {
      f32x16 acc[2];
      zero(acc);
      mm(acc, sHh, sHl, g.wqkvf + (long)nc * 65536, nc + 1 < nq ? g.wqkvf + (long)(nc + 1) * 65536 : nullptr);
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] *= kWScaleInv; acc[1][r] *= kWScaleInv; }
      store_global(acc, g.qkv, nq * 256, nc * 256);
    }
  }
}

}  // namespace

hipError_t launch_ffn_block_f16x3(const FfnArgs& a, hipStream_t st) {
  if (a.M <= 0) return hipSuccess;
  static PerDeviceOnce attr_set;
  attr_set.run([] {
    (void)hipFuncSetAttribute((const void*)ffn_block_f16x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_block_f16x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_block_f16x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  // (64-row tiles halve the weight stream but leave one wave per SIMD: measured 8 % slower at 4096 streams, not built)
  const size_t lds = (size_t)4 * 32 * LD16 * sizeof(_Float16) + (4 * 32 + 32) * sizeof(float);
  const dim3 grid((a.M + 31) / 32), block(256);
  if (a.mode == 1 || a.mode == 2) {
    if (!a.att || !a.wprojf || !a.resid || !a.xmid_out) return hipErrorInvalidValue;
    if (a.mode == 1) hipLaunchKernelGGL(ffn_block_f16x3_kernel<1>, grid, block, lds, st, a);
    else hipLaunchKernelGGL(ffn_block_f16x3_kernel<2>, grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL(ffn_block_f16x3_kernel<0>, grid, block, lds, st, a);
  }
  return hipGetLastError();
}
