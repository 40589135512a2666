// ffn_block_f16x3_kernel: the fused FFN block of fused_blocks.hip with every 256x256 contraction computed as an
// fp32-accurate SPLIT-PRECISION product on the f16 matrix cores (opt-in: VAPX_FLAG_SPLIT_F16, off by default).
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
// Same structure as ffn_block_kernel<1>: 32-row tile, 4 waves x 64 columns, LN_ffn on load, hidden row in LDS, next
// layer's projections from the on-chip tile.  Differences: the A operands sit in LDS as f16 (hi, lo) row-major pairs
// ([32][264] halves each: one ds_read_b128 = the 8 k-values of a lane), weights stream from L2 as f16 fragment pairs
// ([4 w][16 kc][2 ns][2 hi/lo][64 lane][8 halves], weights.frag_pack_f16x3), C/D layout is the fp32 one (dtype
// independent on gfx950), so every epilogue is shared code.  A and B use the same lane -> k mapping, so the k order
// inside a 16-chunk is immaterial.
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

// MT = 32-row sub-tiles per workgroup: 1 (2 workgroups / CU, default) or 2 (64 rows, 1 workgroup / CU, every weight
// fragment feeds two row tiles)
template <int MT>
__global__ __launch_bounds__(256, MT == 1 ? 2 : 1) void ffn_block_f16x3_kernel(const FfnArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int BM = 32 * MT;
  _Float16* sXh = (_Float16*)lds_raw;     // LN_ffn(x) tile, hi / lo
  _Float16* sXl = sXh + BM * LD16;
  _Float16* sHh = sXl + BM * LD16;        // gelu chunk / raw x / LN_self(x), hi / lo
  _Float16* sHl = sHh + BM * LD16;
  float* red = (float*)(sHl + BM * LD16); // [4][BM] row partials
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.x * BM;

  {  // A operand of FFN1 = LayerNorm(xmid; ln_ffn), normalised and split while the tile is staged
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
  }
  __syncthreads();

  // weight fragments: ring of 4 k-chunks (16 x 16-byte fragments) ahead, running on into the next unit
  f32x4 ring[16];
  auto wbase = [&](const float* wfrag) { return (const f32x4*)wfrag + (long)w * 16 * 4 * 64; };   // wave-uniform
  auto fetch = [&](const float* wfrag) {
    const f32x4* wf = wbase(wfrag);
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[i] = wf[i * 64 + lane];
  };
  auto mm = [&](f32x16(&acc)[MT][2], const _Float16* Ah, const _Float16* Al, const float* wfrag, const float* next_wfrag) {
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
        f16x8 ah[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          ah[mt] = *(const f16x8*)(pah + mt * 32 * LD16 + kc * 16);
          al[mt] = *(const f16x8*)(pal + mt * 32 * LD16 + kc * 16);
        }
        const f16x8 b0h = __builtin_bit_cast(f16x8, ring[k4 * 4 + 0]), b0l = __builtin_bit_cast(f16x8, ring[k4 * 4 + 1]);
        const f16x8 b1h = __builtin_bit_cast(f16x8, ring[k4 * 4 + 2]), b1l = __builtin_bit_cast(f16x8, ring[k4 * 4 + 3]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], b0h, acc[mt][0], 0, 0, 0);
          acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], b1h, acc[mt][1], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], b0h, acc[mt][0], 0, 0, 0);
          acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mt], b1h, acc[mt][1], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], b0l, acc[mt][0], 0, 0, 0);
          acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mt], b1l, acc[mt][1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[k4 * 4 + i] = nx[(k4 * 4 + i) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);   // keep the refills behind their MFMAs
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto zero = [](f32x16(&acc)[MT][2]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[mt][0][r] = 0.f; acc[mt][1][r] = 0.f; }
  };
  auto scale = [](f32x16(&acc)[MT][2]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[mt][0][r] *= kWScaleInv; acc[mt][1][r] *= kWScaleInv; }
  };
  // accumulator (ns, r) <-> tile row lr = (r&3) + 8*(r>>2) + 4*hi, chunk column w*64 + ns*32 + l31
  const int ccol = w * 64 + l31;
  auto store_global = [&](const f32x16(&acc)[MT][2], float* base, int ld, int col0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (m < g.M) {
          float* p = base + (long)m * ld + col0 + ccol;
          p[0] = acc[mt][0][r];
          p[32] = acc[mt][1][r];
        }
      }
  };
  auto to_sH = [&](const f32x16(&acc)[MT][2]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int lr = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        split_store(sHh, sHl, lr * LD16 + ccol, acc[mt][0][r]);
        split_store(sHh, sHl, lr * LD16 + ccol + 32, acc[mt][1][r]);
      }
  };

  // ---- feed-forward: x = xmid + gelu(xn W0^T) W3^T, hidden processed in 3 chunks of 256 ----
  const int nq = g.wqkvf ? g.n_qkv_chunks : 0;
  const float* after_ffn = g.wkvxf ? g.wkvxf : (nq ? g.wqkvf : nullptr);
  f32x16 out[MT][2];
  zero(out);
  fetch(g.w0f);
  for (int c = 0; c < 3; ++c) {
    f32x16 hacc[MT][2];
    zero(hacc);
    mm(hacc, sXh, sXl, g.w0f + (long)c * 65536, g.w3f + (long)c * 65536);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hacc[mt][0][r] = gelu_fast(hacc[mt][0][r] * kWScaleInv);
        hacc[mt][1][r] = gelu_fast(hacc[mt][1][r] * kWScaleInv);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    __syncthreads();          // every wave is done reading the previous chunk from sH
    to_sH(hacc);
    __syncthreads();
    mm(out, sHh, sHl, g.w3f + (long)c * 65536, c < 2 ? g.w0f + (long)(c + 1) * 65536 : after_ffn);
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      m = m < g.M ? m : g.M - 1;
      const float* rp = g.xmid + (long)m * 256 + ccol;
      out[mt][0][r] = out[mt][0][r] * kWScaleInv + rp[0];
      out[mt][1][r] = out[mt][1][r] * kWScaleInv + rp[32];
    }
  store_global(out, g.xout, 256, 0);

  // ---- next layer's cross K,V from the RAW layer output ----
  if (g.wkvxf) {
    __syncthreads();
    to_sH(out);
    __syncthreads();
    for (int nc = 0; nc < 2; ++nc) {
      f32x16 acc[MT][2];
      zero(acc);
      mm(acc, sHh, sHl, g.wkvxf + (long)nc * 65536, nc == 0 ? g.wkvxf + 65536 : (nq ? g.wqkvf : nullptr));
      scale(acc);
      store_global(acc, g.kvx, 512, nc * 256);
    }
  }
  // ---- next layer's self Q,K,V from LayerNorm(x) (or just the normalised rows) ----
  if (nq || g.xn_out) {
    float s[MT][16], mean[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[mt][r] = half_sum(out[mt][0][r] + out[mt][1][r]);
    __syncthreads();          // also: every wave is done reading sH (cross K,V)
    if (l31 == 0)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[w * BM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[mt][r];
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int lr = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        mean[mt][r] = (red[lr] + red[BM + lr] + red[2 * BM + lr] + red[3 * BM + lr]) * (1.0f / 256.0f);
      }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float d0 = out[mt][0][r] - mean[mt][r], d1 = out[mt][1][r] - mean[mt][r];
        s[mt][r] = half_sum(d0 * d0 + d1 * d1);
      }
    if (l31 == 0)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[w * BM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[mt][r];
    __syncthreads();
    const float g0 = g.ln_g[ccol], g1 = g.ln_g[ccol + 32], b0 = g.ln_b[ccol], b1 = g.ln_b[ccol + 32];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int lr = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        float var = (red[lr] + red[BM + lr] + red[2 * BM + lr] + red[3 * BM + lr]) * (1.0f / 256.0f);
        float rstd = rsqrtf(var + 1e-5f);
        const float y0 = (out[mt][0][r] - mean[mt][r]) * rstd * g0 + b0;
        const float y1 = (out[mt][1][r] - mean[mt][r]) * rstd * g1 + b1;
        split_store(sHh, sHl, lr * LD16 + ccol, y0);
        split_store(sHh, sHl, lr * LD16 + ccol + 32, y1);
        if (g.xn_out && m0 + lr < g.M) {
          g.xn_out[(long)(m0 + lr) * 256 + ccol] = y0;
          g.xn_out[(long)(m0 + lr) * 256 + ccol + 32] = y1;
        }
      }
    __syncthreads();
    for (int nc = 0; nc < nq; ++nc) {
      f32x16 acc[MT][2];
      zero(acc);
      mm(acc, sHh, sHl, g.wqkvf + (long)nc * 65536, nc + 1 < nq ? g.wqkvf + (long)(nc + 1) * 65536 : nullptr);
      scale(acc);
      store_global(acc, g.qkv, nq * 256, nc * 256);
    }
  }
}

}  // namespace

hipError_t launch_ffn_block_f16x3(const FfnArgs& a, hipStream_t st) {
  if (a.M <= 0) return hipSuccess;
  static PerDeviceOnce attr_set;
  attr_set.run([] {
    (void)hipFuncSetAttribute((const void*)ffn_block_f16x3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  // (64-row tiles — MT = 2 — halve the weight stream but leave one wave per SIMD: measured 8 % slower at 4096 streams, not instantiated)
  const size_t lds = (size_t)4 * 32 * LD16 * sizeof(_Float16) + 4 * 32 * sizeof(float);
  hipLaunchKernelGGL(ffn_block_f16x3_kernel<1>, dim3((a.M + 31) / 32), dim3(256), lds, st, a);
  return hipGetLastError();
}
