// Vertically fused transformer blocks (gfx950).  At a few hundred streams a [rows x 256] activation
// is ~26 MB: every separate GEMM launch pays that in lock-step epilogue traffic plus launch /
// prologue / tail.  These kernels keep the row tile on-chip across several contractions.
//
//  ffn_block_kernel: for a 32-row tile
//      h      = gelu(xn . W0^T)                 [32 x 768]  (never leaves LDS, 256 columns at a time)
//      x_out  = xmid + h . W3^T                 [32 x 256]  -> global (layer output / next residual)
//      kvx    = x_out . Wkv_x(next)^T           [32 x 512]  -> global (next layer's cross K,V: RAW input)
//      qkv    = LN_self(next)(x_out) . Wqkv^T   [32 x 768]  -> global (next layer's self Q,K,V)
//  reference: TransformerLayer.forward modules.py:257-286 (ffnetwork :9-21 bias-free, exact GELU),
//  the q/k/v Linear layers of MultiHeadAttention :36-38 applied to LN(x) (self) and to the
//  un-normalised other-channel input (cross, :276-283).
//
// Same MFMA scheme as gemm_f32.hip: v_mfma_f32_32x32x2_f32, 4 waves, wave w owns 64 of the 256
// columns of a chunk (2 accumulators); the shared A rows are read from LDS with one ds_read_b128
// per 4 k-pairs, the weights stream straight from L2 in fragment order.
#include "fused_blocks.h"

namespace {

constexpr int LDH = 260;   // resident [32][256] buffer row stride (256 + 4 pad)

// Weights arrive pre-packed in MFMA-fragment-major order (weights.frag_pack): for a 256x256
// sub-matrix, [4 wave][32 kc][2 ns][64 lane][4].  A wave owns 64 distinct output columns, so its B
// fragments are shared with nobody: they go global -> VGPR directly (one coalesced 1 KiB load per
// 4 MFMAs) and the K loops contain no barrier and no LDS store at all.  Only the A operand (the 32
// activation rows all four waves share) lives in LDS.
__global__ __launch_bounds__(256, 2) void ffn_block_kernel(const FfnArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 32 * LDH + 128];
  float* sX = lds;                 // [32][260] LN_ffn(x) tile (A operand of FFN1)
  float* sH = lds + 32 * LDH;      // [32][260] gelu chunk / raw x / LN_self(x)
  float* red = sH + 32 * LDH;      // [4][32] row partials
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5, kh = hi * 4;
  const int m0 = blockIdx.x * 32;

  for (int i = tid; i < 32 * 64; i += 256) {
    int row = i >> 6, q = (i & 63) * 4;
    int m = m0 + row;
    m = m < g.M ? m : g.M - 1;
    *(f32x4*)&sX[row * LDH + q] = *(const f32x4*)(g.xn + (long)m * 256 + q);
  }
  __syncthreads();

  // acc[2] += A[32 x 256] (LDS) . Wsub^T for this wave's 64 columns; wfrag = 256x256 fragment block.
  // Explicit register pipeline: the fragments of the next 8-kc block (16 x 1 KiB per wave) are in
  // flight while the 64 MFMAs of the current block run (~4k cycles of cover for the L2 latency),
  // and the first block of the NEXT contraction is fetched during the last block of this one.
  f32x4 ring[16];   // weight fragments of 8 kc steps (2 per step), refilled in place one block ahead
  auto wbase = [&](const float* wfrag) { return (const f32x4*)wfrag + (long)w * 32 * 2 * 64 + lane; };
  auto fetch = [&](const float* wfrag, int) {
    const f32x4* wf = wbase(wfrag);
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[i] = wf[i * 64];
  };
  auto mm = [&](f32x16(&acc)[2], const float* A, const float* wfrag, const float* next_wfrag) {
    const float* pa = A + l31 * LDH + kh;
    const f32x4* wf = wbase(wfrag);
    const f32x4* wnext = next_wfrag ? wbase(next_wfrag) : wf;   // no successor: harmless re-read
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
      const f32x4* nx = blk < 3 ? wf + (blk + 1) * 16 * 64 : wnext;
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8) {
        f32x4 a = *(const f32x4*)(pa + (blk * 8 + k8) * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], ring[k8 * 2][s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], ring[k8 * 2 + 1][s], acc[1], 0, 0, 0);
        }
        ring[k8 * 2] = nx[(k8 * 2) * 64];
        ring[k8 * 2 + 1] = nx[(k8 * 2 + 1) * 64];
      }
    }
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
  auto to_sH = [&](const f32x16(&acc)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      sH[lr * LDH + ccol] = acc[0][r];
      sH[lr * LDH + ccol + 32] = acc[1][r];
    }
  };

  // ---- feed-forward: x = xmid + gelu(xn W0^T) W3^T, hidden processed in 3 chunks of 256 ----
  const int nq = g.wqkvf ? g.n_qkv_chunks : 0;
  const float* after_ffn = g.wkvxf ? g.wkvxf : (nq ? g.wqkvf : nullptr);
  f32x16 out[2];
  zero(out);
  fetch(g.w0f, 0);
  for (int c = 0; c < 3; ++c) {
    f32x16 hacc[2];
    zero(hacc);
    mm(hacc, sX, g.w0f + (long)c * 65536, g.w3f + (long)c * 65536);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      hacc[0][r] = gelu_erf(hacc[0][r]);
      hacc[1][r] = gelu_erf(hacc[1][r]);
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();          // every wave is done reading the previous chunk from sH
    to_sH(hacc);
    __syncthreads();
    mm(out, sH, g.w3f + (long)c * 65536, c < 2 ? g.w0f + (long)(c + 1) * 65536 : after_ffn);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    m = m < g.M ? m : g.M - 1;
    const float* rp = g.xmid + (long)m * 256 + ccol;
    out[0][r] += rp[0];
    out[1][r] += rp[32];
  }
  store_global(out, g.xout, 256, 0);

  // ---- next layer's cross K,V from the RAW layer output ----
  if (g.wkvxf) {
    __syncthreads();
    to_sH(out);
    __syncthreads();
    for (int nc = 0; nc < 2; ++nc) {
      f32x16 acc[2];
      zero(acc);
      mm(acc, sH, g.wkvxf + (long)nc * 65536, nc == 0 ? g.wkvxf + 65536 : (nq ? g.wqkvf : nullptr));
      store_global(acc, g.kvx, 512, nc * 256);
    }
  }
  // ---- next layer's self Q,K,V from LayerNorm(x) ----
  if (nq) {
    float s[16], mean[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = half_sum(out[0][r] + out[1][r]);
    __syncthreads();          // also: every wave is done reading sH (cross K,V)
    if (l31 == 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[w * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      mean[r] = (red[lr] + red[32 + lr] + red[64 + lr] + red[96 + lr]) * (1.0f / 256.0f);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float d0 = out[0][r] - mean[r], d1 = out[1][r] - mean[r];
      s[r] = half_sum(d0 * d0 + d1 * d1);
    }
    if (l31 == 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[w * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[r];
    __syncthreads();
    const float g0 = g.ln_g[ccol], g1 = g.ln_g[ccol + 32], b0 = g.ln_b[ccol], b1 = g.ln_b[ccol + 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float var = (red[lr] + red[32 + lr] + red[64 + lr] + red[96 + lr]) * (1.0f / 256.0f);
      float rstd = rsqrtf(var + 1e-5f);
      sH[lr * LDH + ccol] = (out[0][r] - mean[r]) * rstd * g0 + b0;
      sH[lr * LDH + ccol + 32] = (out[1][r] - mean[r]) * rstd * g1 + b1;
    }
    __syncthreads();
    for (int nc = 0; nc < nq; ++nc) {
      f32x16 acc[2];
      zero(acc);
      mm(acc, sH, g.wqkvf + (long)nc * 65536, nc + 1 < nq ? g.wqkvf + (long)(nc + 1) * 65536 : nullptr);
      store_global(acc, g.qkv, g.n_qkv_chunks * 256, nc * 256);
    }
  }
}

}  // namespace

hipError_t launch_ffn_block(const FfnArgs& a, hipStream_t st) {
  if (a.M <= 0) return hipSuccess;
  hipLaunchKernelGGL(ffn_block_kernel, dim3((a.M + 31) / 32), dim3(256), 0, st, a);
  return hipGetLastError();
}
