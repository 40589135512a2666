// last_block_kernel: the WHOLE last transformer layer for the newest row of every (stream, channel).
//
// Reference: TransformerStereoLayer.forward modules.py:289-300 -> TransformerLayer.forward :257-286 (self-attention,
// cross-attention with Q = LN_src(x) and K = V = the other channel's raw input, FFN), of which process_vap consumes
// row [-1] only (vap_main.py:290-317).  With a single query row per (stream, channel) the key / value projections
// of the other T-1 rows never have to be formed:
//     score_j = q_h . (Wk_h xn_j) = (Wk_h^T q_h) . xn_j          (absorb Wk into the query: one 64x256 product per head)
//     out_h   = sum_j p_j (Wv_h xn_j) = Wv_h (sum_j p_j xn_j)    (apply Wv after the weighted sum)
// so the previous layer's FFN block only emits LN_self(x) rows (and the raw x rows it writes anyway) instead of four
// 256-wide projections of every row — exact in real arithmetic, -12 % of the FFN-block MFMA work per tick.
// Everything that depends on the newest row runs here in one launch: q -> Wk^T q -> attention over LN_self(x) rows ->
// Wv -> proj + residual -> LN_src -> q_x -> Wk_x^T q_x -> attention over the OTHER channel's raw rows -> Wv_x ->
// proj_x + residual -> LN_ffn -> FFN + residual.
//
// One workgroup = ROWS (16 or 8) rows, 8 waves.  The fourteen 256x256 contractions run as v_mfma_f32_16x16x4_f32 on
// a 16-row tile (half empty for ROWS = 8, which doubles the workgroup count of a small batch): wave w owns output
// columns 32w..32w+31, its B fragments stream from L2 in fragment-major order ([unit][8 w][16 kc][2 ns][64 lane][4],
// weights.py "L3.last16") through a 16-deep register ring (8 k-chunks = 64 MFMAs ahead) that runs on into the next
// unit.  The residual stream stays in registers (accumulator layout); LDS holds the A operand of the next contraction
// and a [16][4 heads][256] buffer (absorbed queries, then weighted row sums, later the FFN hidden row).
// Attention: lane = (key parity, 8-column chunk); a 1 KB row is one coalesced wave load per two keys, the four head
// scores are 32-lane DPP reductions, softmax is online per key parity and merged at the end.
#include "fused_blocks.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int LB_LD = 260;     // 256 + 4 pad floats
constexpr int LB_QLD = 1028;   // 4 heads x 256 + 4
constexpr int LB_HLD = 772;    // 768 + 4 (FFN hidden, aliases the head buffer)

template <int ROWS>
__global__ __launch_bounds__(512) void last_block_kernel(const LastBlockArgs a) {
  __shared__ __attribute__((aligned(16))) float xa[16 * LB_LD];      // A operand of the next contraction
  __shared__ __attribute__((aligned(16))) float big[16 * LB_QLD];    // [row][head][256]: Wk^T q, then sum_j p_j x_j; FFN hidden
  __shared__ float red[8 * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * ROWS, M = a.B * 2, T = a.T;
  const int oc = w * 32 + l15;   // this lane's output columns: oc, oc + 16

  // ---- weight stream: virtual k-chunk v = 16 u + kc of the 14 units, 8 chunks ahead ----
  const f32x4* wbase = (const f32x4*)a.wf + (long)w * 16 * 2 * 64;
  auto frag = [&](int v) {   // clamp: the tail prefetch of the last unit re-reads its own head
    v = v < 14 * 16 ? v : 14 * 16 - 1;
    return wbase + (long)(v >> 4) * 16384 + (v & 15) * 128;
  };
  f32x4 ring[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    ring[i * 2] = frag(i)[lane];
    ring[i * 2 + 1] = frag(i)[64 + lane];
  }
  // acc[ns] += A[16][256] . W_u^T for this wave's 32 columns
  auto unit = [&](int u, const float* A, int lda, f32x4v (&acc)[2]) {
    const float* pa = A + l15 * lda + kq * 4;
#pragma unroll 1
    for (int k8 = 0; k8 < 2; ++k8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kc = k8 * 8 + k;
        f32x4 av = *(const f32x4*)(pa + kc * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], ring[k * 2][s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], ring[k * 2 + 1][s], acc[1], 0, 0, 0);
        }
        const f32x4* nx = frag(u * 16 + kc + 8);
        ring[k * 2] = nx[lane];
        ring[k * 2 + 1] = nx[64 + lane];
        __builtin_amdgcn_sched_barrier(0);   // keep the refill behind its MFMAs
      }
    }
  };
  // the absorbed-query unit: four K = 64 products (head h = k-chunks 4h..4h+3), one accumulator pair per head
  auto unit_heads = [&](int u, const float* A, int lda, f32x4v (&acc)[4][2]) {
    const float* pa = A + l15 * lda + kq * 4;
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) {
      f32x4 av = *(const f32x4*)(pa + kc * 16);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[kc >> 2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], ring[(kc & 7) * 2][s], acc[kc >> 2][0], 0, 0, 0);
        acc[kc >> 2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], ring[(kc & 7) * 2 + 1][s], acc[kc >> 2][1], 0, 0, 0);
      }
      const f32x4* nx = frag(u * 16 + kc + 8);
      ring[(kc & 7) * 2] = nx[lane];
      ring[(kc & 7) * 2 + 1] = nx[64 + lane];
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // LayerNorm of the 16 rows held in accumulator layout (v[ns][reg]: row 4kq+reg, column oc + 16 ns) -> xa
  auto rows_ln = [&](const float (&v)[2][4], const float* g, const float* bta) {
    float s[4], mean[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      float t = v[0][reg] + v[1][reg];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o);
      s[reg] = t;
    }
    if (l15 == 0)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) red[w * 16 + kq * 4 + reg] = s[reg];
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) t += red[u * 16 + kq * 4 + reg];
      mean[reg] = t * (1.0f / 256.0f);
    }
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      float d0 = v[0][reg] - mean[reg], d1 = v[1][reg] - mean[reg];
      float t = d0 * d0 + d1 * d1;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o);
      s[reg] = t;
    }
    if (l15 == 0)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) red[w * 16 + kq * 4 + reg] = s[reg];
    __syncthreads();
    const float g0 = g[oc], g1 = g[oc + 16], b0 = bta[oc], b1 = bta[oc + 16];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) t += red[u * 16 + kq * 4 + reg];
      const float rstd = rsqrtf(t * (1.0f / 256.0f) + 1e-5f);
      float* dst = &xa[(kq * 4 + reg) * LB_LD + oc];
      dst[0] = (v[0][reg] - mean[reg]) * rstd * g0 + b0;
      dst[16] = (v[1][reg] - mean[reg]) * rstd * g1 + b1;
    }
    __syncthreads();
  };

  // single-query attention of this wave's row(s) over the rows of `src` ([B*2][T][256]); absorbed queries in big
  // [row][head][256] (pre-scaled by 1/16), replaced in place by sum_j softmax_j x_j per head
  auto attention = [&](const float* src, bool swap) {
    const int half = lane >> 5, c8 = (lane & 31) * 8;
#pragma unroll 1
    for (int rr = 0; rr < ROWS / 8; ++rr) {
      const int r = w * (ROWS / 8) + rr;
      int bc = m0 + r;
      bc = bc < M ? bc : M - 1;
      const int n = a.bn[bc >> 1];
      const float* rows = src + (long)(swap ? (bc ^ 1) : bc) * T * 256 + c8;
      float* qrow = &big[r * LB_QLD + c8];
      f32x4 qk[4][2];
#pragma unroll
      for (int h = 0; h < 4; ++h) { qk[h][0] = *(const f32x4*)(qrow + h * 256); qk[h][1] = *(const f32x4*)(qrow + h * 256 + 4); }
      float mx[4], l[4];
      f32x4 u[4][2];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        mx[h] = -1e30f; l[h] = 0.f;
        u[h][0] = f32x4{0.f, 0.f, 0.f, 0.f}; u[h][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll 1
      for (int j0 = 0; j0 < n; j0 += 8) {
        f32x4 x[4][2];
        float sc[4][4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int j = j0 + it * 2 + half;
          const long jr = j < n ? j : n - 1;
          x[it][0] = *(const f32x4*)(rows + jr * 256);
          x[it][1] = *(const f32x4*)(rows + jr * 256 + 4);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int j = j0 + it * 2 + half;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            f32x4 p = x[it][0] * qk[h][0] + x[it][1] * qk[h][1];
            float d = half_sum((p[0] + p[1]) + (p[2] + p[3]));
            sc[it][h] = j < n ? d + exp2f(-2.0f * (float)(h + 1)) * (float)j : -1e30f;
          }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const float mn = fmaxf(fmaxf(mx[h], fmaxf(sc[0][h], sc[1][h])), fmaxf(sc[2][h], sc[3][h]));
          const float corr = __expf(mx[h] - mn);
          mx[h] = mn;
          l[h] *= corr;
          u[h][0] *= corr; u[h][1] *= corr;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const float p = __expf(sc[it][h] - mn);   // 0 for padded keys (and for everything while mn = -1e30)
            l[h] += p;
            u[h][0] += x[it][0] * p; u[h][1] += x[it][1] * p;
          }
        }
      }
      // merge the two key parities; a parity that saw no key (n = 1) has mx = -1e30, l = u = 0 and contributes 0...
      // except that exp(-1e30 - (-1e30)) = 1 made its l count the padded keys: those entered with p = exp(0) = 1
      // only if mn = -1e30, i.e. only if the parity saw NO valid key; reset that case explicitly
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if (mx[h] < -1e29f) { l[h] = 0.f; u[h][0] = f32x4{0.f, 0.f, 0.f, 0.f}; u[h][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const float mo = __shfl_xor(mx[h], 32);
        const float mn = fmaxf(mx[h], mo);
        const float cs = __expf(mx[h] - mn);
        float lt = l[h] * cs;
        lt += __shfl_xor(lt, 32);
        const float inv = 1.0f / lt;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          f32x4 t = u[h][q] * cs;
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] += __shfl_xor(t[e], 32);
          if (half == 0) *(f32x4*)(qrow + h * 256 + q * 4) = t * inv;
        }
      }
    }
    __syncthreads();
  };

  // ---- newest row: LN_self(x) row (already formed by the previous FFN block) -> xa; raw row -> residual regs ----
  for (int i = tid; i < 16 * 64; i += 512) {
    const int r = i >> 6, c4 = (i & 63) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < ROWS) {
      int bc = m0 + r;
      bc = bc < M ? bc : M - 1;
      v = *(const f32x4*)(a.xn + ((long)bc * T + a.bn[bc >> 1] - 1) * 256 + c4);
    }
    *(f32x4*)&xa[r * LB_LD + c4] = v;
  }
  float xres[2][4];   // residual stream, accumulator layout
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    int bc = m0 + kq * 4 + reg;
    bc = (kq * 4 + reg < ROWS && bc < M) ? bc : (m0 < M ? m0 : M - 1);
    const float* xr = a.x + ((long)bc * T + a.bn[bc >> 1] - 1) * 256 + oc;
    xres[0][reg] = xr[0];
    xres[1][reg] = xr[16];
  }
  __syncthreads();

  f32x4v acc[4][2];
  auto zero = [&]() {
#pragma unroll
    for (int h = 0; h < 4; ++h) { acc[h][0] = f32x4v{0.f, 0.f, 0.f, 0.f}; acc[h][1] = f32x4v{0.f, 0.f, 0.f, 0.f}; }
  };
  auto acc_to_xa = [&]() {   // accumulator pair 0 -> xa (A operand of the next unit)
    __syncthreads();         // every wave is done reading xa / big
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) xa[(kq * 4 + reg) * LB_LD + oc + ns * 16] = acc[0][ns][reg];
    __syncthreads();
  };
  auto heads_to_big = [&]() {   // per-head accumulators (absorbed queries) -> big, scaled by 1/sqrt(256) (modules.py:52,98)
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int ns = 0; ns < 2; ++ns)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) big[(kq * 4 + reg) * LB_QLD + h * 256 + oc + ns * 16] = acc[h][ns][reg] * 0.0625f;
    __syncthreads();
  };
  auto add_resid = [&]() {
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) xres[ns][reg] += acc[0][ns][reg];
  };
  const float* uhead = big + (w >> 1) * 256;   // this wave's output columns belong to head w / 2

  // ---- self-attention ----
  zero(); unit(0, xa, LB_LD, acc[0]);       // q = LN_self(x) . Wq^T
  acc_to_xa();
  zero(); unit_heads(1, xa, LB_LD, acc);       // per head: Wk_h^T q_h  (K = 64)
  heads_to_big();
  attention(a.xn, false);                   // big <- sum_j p_j LN_self(x_j) per head
  zero(); unit(2, uhead, LB_QLD, acc[0]);   // Wv_h applied after the weighted sum
  acc_to_xa();
  zero(); unit(3, xa, LB_LD, acc[0]);       // proj
  add_resid();
  rows_ln(xres, a.ln_src_g, a.ln_src_b);    // (first barrier inside: xa reads finished)
  // ---- cross-attention: keys / values are the OTHER channel's raw rows ----
  zero(); unit(4, xa, LB_LD, acc[0]);       // q_x = LN_src(x) . Wq_x^T
  acc_to_xa();
  zero(); unit_heads(5, xa, LB_LD, acc);
  heads_to_big();
  attention(a.x, true);
  zero(); unit(6, uhead, LB_QLD, acc[0]);
  acc_to_xa();
  zero(); unit(7, xa, LB_LD, acc[0]);       // proj_x
  add_resid();
  rows_ln(xres, a.ln_ffn_g, a.ln_ffn_b);
  // ---- FFN ----
  float* hb = big;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {
    zero(); unit(8 + c, xa, LB_LD, acc[0]);
#pragma unroll
    for (int ns = 0; ns < 2; ++ns)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) hb[(kq * 4 + reg) * LB_HLD + c * 256 + oc + ns * 16] = gelu_fast(acc[0][ns][reg]);
  }
  __syncthreads();
  zero();
#pragma unroll 1
  for (int c = 0; c < 3; ++c) unit(11 + c, hb + c * 256, LB_HLD, acc[0]);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int m = m0 + kq * 4 + reg;
    if (kq * 4 + reg < ROWS && m < M) {
      a.out[(long)m * 256 + oc] = xres[0][reg] + acc[0][0][reg];
      a.out[(long)m * 256 + oc + 16] = xres[1][reg] + acc[0][1][reg];
    }
  }
}

}  // namespace

hipError_t launch_last_block(const LastBlockArgs& a, hipStream_t st) {
  const int M = a.B * 2;
  // 8-row workgroups while they still fit one per CU: twice the workgroups of a small batch
  if ((M + 7) / 8 <= 256) hipLaunchKernelGGL(last_block_kernel<8>, dim3((M + 7) / 8), dim3(512), 0, st, a);
  else hipLaunchKernelGGL(last_block_kernel<16>, dim3((M + 15) / 16), dim3(512), 0, st, a);
  return hipGetLastError();
}
