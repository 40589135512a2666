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
// columns of a chunk (2 accumulators), K-contiguous operands read from LDS with one ds_read_b128
// per 4 k-pairs, weight tiles [256 n][32 k] staged through LDS with register prefetch.
#include "fused_blocks.h"

namespace {

constexpr int LDT = 36;    // staged tile row stride (32 + 4 pad)
constexpr int LDH = 260;   // resident [32][256] buffer row stride (256 + 4 pad)

__global__ __launch_bounds__(256, 2) void ffn_block_kernel(const FfnArgs g) {
  __shared__ __attribute__((aligned(16))) float lds[32 * LDH + 32 * LDT + 256 * LDT];
  float* sH = lds;                 // [32][260] resident A operand (gelu chunk / raw x / LN(x))
  float* sA = lds + 32 * LDH;      // [32][36]  staged A tile (xn from global)
  float* sW = sA + 32 * LDT;       // [256][36] staged weight tile
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5, kh = hi * 4;
  const int m0 = blockIdx.x * 32;
  const int srow = tid >> 3, skq = (tid & 7) * 4;
  int arow = m0 + srow;
  arow = arow < g.M ? arow : g.M - 1;

  // One software pipeline runs through ALL contractions of the block: while the MFMAs of k-step t
  // execute, the weight (and A) tile of step t+1 is in flight in registers — including across the
  // boundary between two contractions (`nxt`), so the ~2 us global-load latency is paid once per
  // workgroup instead of once per contraction.
  struct Src { const float* a; const float* w; long ldw; };   // a == nullptr: A operand is resident in sH
  f32x4 ra, rb[8];
  auto gload = [&](const Src& sdesc, int k0) {
    if (sdesc.a) ra = *(const f32x4*)(sdesc.a + (long)arow * 256 + skq + k0);
    const float* bp = sdesc.w + (long)srow * sdesc.ldw + skq + k0;
    const long bstep = 32 * sdesc.ldw;
#pragma unroll
    for (int i = 0; i < 8; ++i) rb[i] = *(const f32x4*)(bp + i * bstep);
  };
  auto sstore = [&](bool with_a) {
    if (with_a) *(f32x4*)&sA[srow * LDT + skq] = ra;
#pragma unroll
    for (int i = 0; i < 8; ++i) *(f32x4*)&sW[(i * 32 + srow) * LDT + skq] = rb[i];
  };
  // acc[2] += A[32 x 256] . W[256 n x 256 k]^T for this wave's 64 columns.  Tile 0 of `cur` must
  // already be in (ra, rb); on return tile 0 of `nxt` is (when nxt.w != nullptr).
  auto mm = [&](f32x16(&acc)[2], const Src cur, const Src nxt) {
    const bool with_a = cur.a != nullptr;
    sstore(with_a);
    __syncthreads();
    const float* pb = &sW[(w * 64 + l31) * LDT + kh];
    for (int kt = 0; kt < 8; ++kt) {
      if (kt + 1 < 8) gload(cur, (kt + 1) * 32);
      else if (nxt.w) gload(nxt, 0);
      const float* pa = with_a ? &sA[l31 * LDT + kh] : &sH[l31 * LDH + kt * 32 + kh];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        f32x4 a = *(const f32x4*)(pa + kc * 8);
        f32x4 b0 = *(const f32x4*)(pb + kc * 8);
        f32x4 b1 = *(const f32x4*)(pb + 32 * LDT + kc * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1[s], acc[1], 0, 0, 0);
        }
      }
      __syncthreads();
      if (kt + 1 < 8) {
        sstore(with_a);
        __syncthreads();
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

  const Src none{nullptr, nullptr, 0};
  auto w0c = [&](int c) { return Src{g.xn, g.w0 + (long)c * 256 * 256, 256}; };
  auto w3c = [&](int c) { return Src{nullptr, g.w3 + c * 256, 768}; };
  auto kvxc = [&](int nc) { return Src{nullptr, g.wkvx + (long)nc * 256 * 256, 256}; };
  auto qkvc = [&](int nc) { return Src{nullptr, g.wqkv + (long)nc * 256 * 256, 256}; };
  const int nq = g.wqkv ? (g.n_qkv >> 8) : 0;
  // what follows the feed-forward
  const Src after_ffn = g.wkvx ? kvxc(0) : (nq ? qkvc(0) : none);

  // ---- feed-forward: x = xmid + gelu(xn W0^T) W3^T, hidden processed in 3 chunks of 256 ----
  f32x16 out[2];
  zero(out);
  gload(w0c(0), 0);
  for (int c = 0; c < 3; ++c) {
    f32x16 hacc[2];
    zero(hacc);
    mm(hacc, w0c(c), w3c(c));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      sH[lr * LDH + ccol] = gelu_erf(hacc[0][r]);
      sH[lr * LDH + ccol + 32] = gelu_erf(hacc[1][r]);
      if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    mm(out, w3c(c), c < 2 ? w0c(c + 1) : after_ffn);
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
  if (g.wkvx) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      sH[lr * LDH + ccol] = out[0][r];
      sH[lr * LDH + ccol + 32] = out[1][r];
    }
    __syncthreads();
    for (int nc = 0; nc < 2; ++nc) {
      f32x16 acc[2];
      zero(acc);
      mm(acc, kvxc(nc), nc == 0 ? kvxc(1) : (nq ? qkvc(0) : none));
      store_global(acc, g.kvx, 512, nc * 256);
    }
  }
  // ---- next layer's self Q,K,V from LayerNorm(x) ----
  if (nq) {
    float* red = sA;  // [4 waves][32 rows] row partials (sA is idle: A comes from sH here)
    float s[16], mean[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = half_sum(out[0][r] + out[1][r]);
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
      mm(acc, qkvc(nc), nc + 1 < nq ? qkvc(nc + 1) : none);
      store_global(acc, g.qkv, g.n_qkv, nc * 256);
    }
  }
}

}  // namespace

hipError_t launch_ffn_block(const FfnArgs& a, hipStream_t st) {
  if (a.M <= 0) return hipSuccess;
  hipLaunchKernelGGL(ffn_block_kernel, dim3((a.M + 31) / 32), dim3(256), 0, st, a);
  return hipGetLastError();
}
