// Non-GEMM kernels of the VAP step (gfx950): frame assembly + conv0, LSTM recurrence, context-ring
// gather + LayerNorm, causal ALiBi attention, combinator + heads.
#include "vap_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------
// 1. frame assembly + conv0 (1 -> 256, k10 s5 p3) + ChannelNorm + ReLU
//    reference: proc_serv_in carry logic vap_main.py:397-409; CPCEncoder.conv0/batchNorm0
//    encoder_components.py:83-84,99; ChannelNorm 64-70 (unbiased variance).
//    K = 10 is too thin for MFMA: one wave computes one output position (256 channels, 4 per lane)
//    with the 40 taps*channels weights held in registers; the channel reduction is a wave butterfly.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv0_kernel(Conv0Args a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [L]
  const int bc = blockIdx.x, b = bc >> 1, c = bc & 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sid = a.ids ? a.ids[b] : b;
  const int L = a.L, hop = L - VAPX_PAD;
  float* cr = a.carry ? a.carry + ((long)sid * 2 + c) * VAPX_PAD : nullptr;
  if (a.spc == hop) {
    for (int i = tid; i < VAPX_PAD; i += 256) xs[i] = cr[i];
    const float* src = a.audio + ((long)b * 2 + c) * hop;
    for (int i = tid; i < hop; i += 256) xs[VAPX_PAD + i] = src[i];
  } else {
    const float* src = a.audio + ((long)b * 2 + c) * L;
    for (int i = tid; i < L; i += 256) xs[i] = src[i];
  }
  __syncthreads();
  if (cr)
    for (int i = tid; i < VAPX_PAD; i += 256) cr[i] = xs[L - VAPX_PAD + i];
  if (a.frames_seen && c == 0 && tid == 0) {
    int fs = a.frames_seen[sid];
    a.bn[b] = fs + 1 < a.T ? fs + 1 : a.T;
    a.bhead[b] = fs % a.T;
  }

  const int ch = lane * 4;
  f32x4 w[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) w[t] = *(const f32x4*)(a.w + t * 256 + ch);
  const f32x4 bias = *(const f32x4*)(a.bias + ch);
  const f32x4 gam = *(const f32x4*)(a.gamma + ch);
  const f32x4 bet = *(const f32x4*)(a.beta + ch);
  const int P0 = L / 5;
  float* outb = a.h0 + (long)bc * (P0 + 4) * 256 + 2 * 256;
  for (int p = wave; p < P0; p += 4) {
    f32x4 v = bias;
    const int x0 = p * 5 - 3;
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      int xi = x0 + t;
      float xv = (xi >= 0 && xi < L) ? xs[xi] : 0.f;
      v += w[t] * xv;
    }
    float s = wave_sum(v[0] + v[1] + v[2] + v[3]);
    float mean = s * (1.0f / 256.0f);
    f32x4 d = v - mean;
    float ss = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
    float rstd = rsqrtf(ss * (1.0f / 255.0f) + 1e-5f);
    f32x4 y = d * rstd * gam + bet;
    y[0] = fmaxf(y[0], 0.f); y[1] = fmaxf(y[1], 0.f); y[2] = fmaxf(y[2], 0.f); y[3] = fmaxf(y[3], 0.f);
    *(f32x4*)(outb + (long)p * 256 + ch) = y;
  }
}

// ------------------------------------------------------------------------------------------------
// 2. LSTM (256 -> 256, 1 layer), n_cpc sequential steps, (h, c) persistent per (stream, channel)
//    reference: CPCAR.forward encoder_components.py:140-159 (nn.LSTM, gate order i,f,g,o, both
//    biases), keepHidden=True (encoder.py:27).
//    One workgroup = 32 (stream,channel) rows x all 1024 gate columns; wave w owns hidden units
//    64w..64w+63 for all four gates (weight rows pre-permuted so its 256 columns are
//    [i|f|g|o] x 64), i.e. 8 MFMA accumulators of 32x32 — the cell update is then lane-local.
//    Per step the contraction is [z_t | h_{t-1}] (K = 512) . [W_ih | W_hh]^T; the A operand sits
//    in LDS, the weight fragments stream from L2 in a fragment-major layout (one coalesced
//    1 KiB load per wave per 4 MFMAs).  c stays in registers across steps, h in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int ZH_LD = 516;  // 512 + 4 pad floats

__global__ __launch_bounds__(256) void lstm_kernel(LstmArgs a) {
  __shared__ __attribute__((aligned(16))) float zh[32 * ZH_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5, kh = hi * 4;
  const int m0 = blockIdx.x * 32;

  // state row of a local row
  auto state_row = [&](int m) {
    m = m < a.M ? m : a.M - 1;
    int b = m >> 1;
    int sid = a.ids ? a.ids[b] : b;
    return (long)sid * 2 + (m & 1);
  };
  // h_{-1} -> zh[:, 256:512]
  for (int i = tid; i < 32 * 64; i += 256) {
    int row = i >> 6, q = (i & 63) * 4;
    *(f32x4*)&zh[row * ZH_LD + 256 + q] = *(const f32x4*)(a.h_state + state_row(m0 + row) * 256 + q);
  }
  // c in registers: element (hh, r) <-> row lr(r), hidden unit j = 64w + 32hh + l31
  float creg[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    long sr = state_row(m0 + row);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) creg[hh][r] = a.c_state[sr * 256 + w * 64 + hh * 32 + l31];
  }
  float bias[8];
#pragma unroll
  for (int ns = 0; ns < 8; ++ns) bias[ns] = a.bias[w * 256 + ns * 32 + l31];

  const f32x4* wf = (const f32x4*)a.wfrag + (long)w * 64 * 8 * 64 + lane;
  for (int t = 0; t < a.ncpc; ++t) {
    for (int i = tid; i < 32 * 64; i += 256) {
      int row = i >> 6, q = (i & 63) * 4;
      int m = m0 + row;
      m = m < a.M ? m : a.M - 1;
      *(f32x4*)&zh[row * ZH_LD + q] = *(const f32x4*)(a.z + ((long)m * a.ncpc + t) * 256 + q);
    }
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int ns = 0; ns < 8; ++ns)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ns][r] = bias[ns];
    const float* pa = &zh[l31 * ZH_LD + kh];
#pragma unroll 2
    for (int kc = 0; kc < 64; ++kc) {
      f32x4 av = *(const f32x4*)(pa + kc * 8);
      f32x4 bv[8];
#pragma unroll
      for (int ns = 0; ns < 8; ++ns) bv[ns] = wf[((long)kc * 8 + ns) * 64];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ns = 0; ns < 8; ++ns)
          acc[ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[ns][s], acc[ns], 0, 0, 0);
    }
    __syncthreads();  // every wave has finished reading zh for this step
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
      int m = m0 + row;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float ig = sigmoidf_(acc[0 + hh][r]);
        float fg = sigmoidf_(acc[2 + hh][r]);
        float gg = tanhf(acc[4 + hh][r]);
        float og = sigmoidf_(acc[6 + hh][r]);
        float cn = fg * creg[hh][r] + ig * gg;
        creg[hh][r] = cn;
        float hn = og * tanhf(cn);
        int j = w * 64 + hh * 32 + l31;
        zh[row * ZH_LD + 256 + j] = hn;
        if (m < a.M) a.out[((long)m * a.ncpc + t) * 256 + j] = hn;
      }
    }
    // the z-load + barrier of the next step (or the barrier below) orders these LDS writes
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    int m = m0 + row;
    if (m < a.M) {
      long sr = state_row(m);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        int j = w * 64 + hh * 32 + l31;
        a.c_state[sr * 256 + j] = creg[hh][r];
        a.h_state[sr * 256 + j] = zh[row * ZH_LD + 256 + j];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3. context-ring append + chronological gather + LayerNorm(ln_self of layer 0)
//    reference: vap_main.py:274-283 (append, keep last T, cat).  One wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_ln_kernel(GatherArgs a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)a.B * 2 * a.T) return;
  const int t = (int)(row % a.T);
  const int bc = (int)(row / a.T), b = bc >> 1, c = bc & 1;
  const int n = a.bn[b];
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (t < n) {
    if (a.ring) {
      const int sid = a.ids ? a.ids[b] : b;
      const int head = a.bhead[b];
      float* rb = a.ring + ((long)sid * 2 + c) * a.T * 256;
      if (t == n - 1) {
        v = *(const f32x4*)(a.e + (long)bc * 256 + lane * 4);
        *(f32x4*)(rb + (long)head * 256 + lane * 4) = v;
      } else {
        int slot = head + 1 - n + t;
        slot = slot < 0 ? slot + a.T : slot;
        v = *(const f32x4*)(rb + (long)slot * 256 + lane * 4);
      }
    } else {
      v = *(const f32x4*)(a.xin + ((long)bc * a.rows_in + t) * 256 + lane * 4);
    }
  }
  *(f32x4*)(a.x0 + row * 256 + lane * 4) = v;
  float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / 256.0f);
  f32x4 d = v - mean;
  float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.0f / 256.0f);
  float rstd = rsqrtf(var + 1e-5f);
  f32x4 g = *(const f32x4*)(a.gamma + lane * 4), be = *(const f32x4*)(a.beta + lane * 4);
  *(f32x4*)(a.xn + row * 256 + lane * 4) = d * rstd * g + be;
}

// ------------------------------------------------------------------------------------------------
// 4. causal multi-head attention with ALiBi key bias
//    reference: MultiHeadAttention.forward modules.py:82-110 (scale 1/sqrt(dim)=1/16, :52),
//    MultiHeadAttentionAlibi.get_alibi_mask 162-188 (bias m_h * j on the KEY index).
//    One workgroup per (stream, channel, head); K and V tiles [n][64] staged in LDS, one query
//    row per lane (64 q values in registers), online softmax; fp32 vector FMAs (4 % of the step's
//    MACs at T = 50, not worth reshaping for MFMA).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float kv[];  // K [T][64] then V [T][64]
  const int T = a.T;
  float* Ks = kv;
  float* Vs = kv + (long)T * 64;
  const int h = blockIdx.x & 3, bc = blockIdx.x >> 2, b = bc >> 1;
  const int n = a.bn[b];
  const int kvbc = a.swap_kv ? (bc ^ 1) : bc;
  const int tid = threadIdx.x, lane = tid & 63, qb = tid >> 6;
  const float* kp = a.k + (long)kvbc * T * a.ldkv + h * 64;
  const float* vp = a.v + (long)kvbc * T * a.ldkv + h * 64;
  for (int i = tid; i < n * 16; i += blockDim.x) {
    int j = i >> 4, q = (i & 15) * 4;
    *(f32x4*)&Ks[j * 64 + q] = *(const f32x4*)(kp + (long)j * a.ldkv + q);
    *(f32x4*)&Vs[j * 64 + q] = *(const f32x4*)(vp + (long)j * a.ldkv + q);
  }
  __syncthreads();
  const int i = qb * 64 + lane;
  if (i >= T) return;
  float* op = a.out + ((long)bc * T + i) * 256 + h * 64;
  if (i >= n) {
#pragma unroll
    for (int d = 0; d < 16; ++d) *(f32x4*)(op + d * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    return;
  }
  f32x4 q[16];
  const float* qp = a.q + ((long)bc * T + i) * a.ldq + h * 64;
#pragma unroll
  for (int d = 0; d < 16; ++d) q[d] = *(const f32x4*)(qp + d * 4) * 0.0625f;
  const float slope = exp2f(-2.0f * (float)(h + 1));  // [1/4, 1/16, 1/64, 1/256]
  f32x4 o[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mx = -1e30f, l = 0.f;
  int jend = (qb + 1) * 64;
  jend = jend < n ? jend : n;
  for (int j = 0; j < jend; ++j) {
    const f32x4* kr = (const f32x4*)&Ks[j * 64];
    f32x4 sv = q[0] * kr[0];
#pragma unroll
    for (int d = 1; d < 16; ++d) sv += q[d] * kr[d];
    float s = (sv[0] + sv[1]) + (sv[2] + sv[3]) + slope * (float)j;
    if (j <= i) {
      float mn = fmaxf(mx, s);
      float corr = expf(mx - mn);
      float p = expf(s - mn);
      l = l * corr + p;
      mx = mn;
      const f32x4* vr = (const f32x4*)&Vs[j * 64];
#pragma unroll
      for (int d = 0; d < 16; ++d) o[d] = o[d] * corr + vr[d] * p;
    }
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int d = 0; d < 16; ++d) *(f32x4*)(op + d * 4) = o[d] * inv;
}

// ------------------------------------------------------------------------------------------------
// 5. Combinator + heads on the newest row of each stream
//    reference: Combinator.forward modules.py:449-464; vap_head / va_classifier / softmax
//    vap_main.py:290-295,313-314; probs_next_speaker_aggregate objective.py:186-206;
//    bc / nod heads vap_bc_main.py:272-277, vap_nod_main.py:273-279.
//    8 streams per workgroup; thread j owns output feature j (weights pre-transposed [k][j] so
//    the 256 threads read one coalesced row per k); row-wise reductions via wave butterflies +
//    one LDS exchange.
// ------------------------------------------------------------------------------------------------
constexpr int HB = 8;

template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* red /* [4][N] */, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < N; ++i) red[wave * N + i] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = red[i] + red[N + i] + red[2 * N + i] + red[3 * N + i];
}
template <int N>
__device__ __forceinline__ void block_max(float (&v)[N], float* red, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_max(v[i]);
  __syncthreads();
  if (lane == 0)
#pragma unroll
    for (int i = 0; i < N; ++i) red[wave * N + i] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = fmaxf(fmaxf(red[i], red[N + i]), fmaxf(red[2 * N + i], red[3 * N + i]));
}

__global__ __launch_bounds__(256) void head_kernel(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[HB][2][256];  // newest rows of the two towers
  __shared__ __attribute__((aligned(16))) float hs[HB][256];
  __shared__ float red[4 * 2 * HB];
  const int j = threadIdx.x, lane = j & 63, wave = j >> 6;
  const int b0 = blockIdx.x * HB;
  int nb[HB];
#pragma unroll
  for (int s = 0; s < HB; ++s) {
    int b = b0 + s;
    b = b < a.B ? b : a.B - 1;
    nb[s] = a.bn[b];
#pragma unroll
    for (int c = 0; c < 2; ++c) xs[s][c][j] = a.x[(((long)b * 2 + c) * a.T + nb[s] - 1) * 256 + j];
  }
  __syncthreads();
  // combinator projections
  float ha[HB], hb[HB];
#pragma unroll
  for (int s = 0; s < HB; ++s) ha[s] = hb[s] = 0.f;
  for (int k = 0; k < 256; k += 4) {
    float wa[4], wb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      wa[u] = a.waT[(k + u) * 256 + j];
      wb[u] = a.wbT[(k + u) * 256 + j];
    }
#pragma unroll
    for (int s = 0; s < HB; ++s) {
      f32x4 xa = *(const f32x4*)&xs[s][0][k];
      f32x4 xb = *(const f32x4*)&xs[s][1][k];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ha[s] += wa[u] * xa[u];
        hb[s] += wb[u] * xb[u];
      }
    }
  }
  // shared LayerNorm on both, exact GELU, sum
  float v[2 * HB];
#pragma unroll
  for (int s = 0; s < HB; ++s) { v[2 * s] = ha[s]; v[2 * s + 1] = hb[s]; }
  block_sum<2 * HB>(v, red, wave, lane);
  float mean[2 * HB];
#pragma unroll
  for (int s = 0; s < HB; ++s) {
    mean[2 * s] = v[2 * s] * (1.0f / 256.0f);
    mean[2 * s + 1] = v[2 * s + 1] * (1.0f / 256.0f);
    float da = ha[s] - mean[2 * s], db = hb[s] - mean[2 * s + 1];
    v[2 * s] = da * da;
    v[2 * s + 1] = db * db;
  }
  block_sum<2 * HB>(v, red, wave, lane);
  const float cg = a.cg[j], cb = a.cb[j];
#pragma unroll
  for (int s = 0; s < HB; ++s) {
    float ya = (ha[s] - mean[2 * s]) * rsqrtf(v[2 * s] * (1.0f / 256.0f) + 1e-5f) * cg + cb;
    float yb = (hb[s] - mean[2 * s + 1]) * rsqrtf(v[2 * s + 1] * (1.0f / 256.0f) + 1e-5f) * cg + cb;
    hs[s][j] = gelu_erf(ya) + gelu_erf(yb);
  }
  __syncthreads();
  // vap_head logits
  float lg[HB];
  const float hbias = a.hb[j];
#pragma unroll
  for (int s = 0; s < HB; ++s) lg[s] = hbias;
  for (int k = 0; k < 256; k += 4) {
    float wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wv[u] = a.hwT[(k + u) * 256 + j];
#pragma unroll
    for (int s = 0; s < HB; ++s) {
      f32x4 hv = *(const f32x4*)&hs[s][k];
#pragma unroll
      for (int u = 0; u < 4; ++u) lg[s] += wv[u] * hv[u];
    }
  }
  float mxv[HB];
#pragma unroll
  for (int s = 0; s < HB; ++s) mxv[s] = lg[s];
  block_max<HB>(mxv, red, wave, lane);
  float ex[HB], den[HB];
#pragma unroll
  for (int s = 0; s < HB; ++s) { ex[s] = expf(lg[s] - mxv[s]); den[s] = ex[s]; }
  block_sum<HB>(den, red, wave, lane);
  // class j <-> 8 bits; channel c bins = bits 4c..4c+3; now = bins 0-1, future = bins 2-3
  const float n0w = (float)((j & 1) + ((j >> 1) & 1)), n1w = (float)(((j >> 4) & 1) + ((j >> 5) & 1));
  const float f0w = (float)(((j >> 2) & 1) + ((j >> 3) & 1)), f1w = (float)(((j >> 6) & 1) + ((j >> 7) & 1));
  float pn0[HB], pn1[HB], pf0[HB], pf1[HB];
#pragma unroll
  for (int s = 0; s < HB; ++s) {
    float p = ex[s] / den[s];
    pn0[s] = p * n0w; pn1[s] = p * n1w; pf0[s] = p * f0w; pf1[s] = p * f1w;
  }
  block_sum<HB>(pn0, red, wave, lane);
  block_sum<HB>(pn1, red, wave, lane);
  block_sum<HB>(pf0, red, wave, lane);
  block_sum<HB>(pf1, red, wave, lane);
  // VAD on the ar_channel outputs (note: o, not the stereo towers)  vap_main.py:292-293
  float vd[2 * HB];
  const float vw = a.vw[j];
#pragma unroll
  for (int s = 0; s < HB; ++s) {
    int b = b0 + s;
    b = b < a.B ? b : a.B - 1;
#pragma unroll
    for (int c = 0; c < 2; ++c) vd[2 * s + c] = vw * a.o[(((long)b * 2 + c) * a.T + nb[s] - 1) * 256 + j];
  }
  block_sum<2 * HB>(vd, red, wave, lane);
  // auxiliary heads (bc: rows 0..2 softmax; nod: rows 0..3 softmax)
  float ax[4][HB];
  if (a.mode != 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float wr = a.aw[r * 256 + j];
#pragma unroll
      for (int s = 0; s < HB; ++s) ax[r][s] = wr * hs[s][j];
      block_sum<HB>(ax[r], red, wave, lane);
    }
  }
#pragma unroll
  for (int s = 0; s < HB; ++s) {
    int b = b0 + s;
    if (b >= a.B) break;
    float* out = a.out + (long)b * a.out_stride;
    out[16 + j] = lg[s];
    out[272 + j] = a.e[((long)b * 2) * 256 + j];
    out[272 + 256 + j] = a.e[((long)b * 2 + 1) * 256 + j];
    if (j == 0) {
      float dn = pn0[s] + pn1[s] + 1e-5f, df = pf0[s] + pf1[s] + 1e-5f;
      out[0] = pn0[s] / dn; out[1] = pn1[s] / dn;
      out[2] = pf0[s] / df; out[3] = pf1[s] / df;
      out[4] = sigmoidf_(vd[2 * s] + a.vb[0]);
      out[5] = sigmoidf_(vd[2 * s + 1] + a.vb[0]);
      out[10] = (float)nb[s];
      if (a.mode != 0) {
        int nr = a.mode == 1 ? 3 : 4;
        float z[4], zm = -1e30f, zs = 0.f;
        for (int r = 0; r < nr; ++r) { z[r] = ax[r][s] + a.ab[r]; zm = fmaxf(zm, z[r]); }
        for (int r = 0; r < nr; ++r) { z[r] = expf(z[r] - zm); zs += z[r]; }
        for (int r = 0; r < 4; ++r) out[6 + r] = r < nr ? z[r] / zs : 0.f;
      }
      if (a.frames_seen) {
        int sid = a.ids ? a.ids[b] : b;
        a.frames_seen[sid] += 1;
      }
    }
  }
}

}  // namespace

hipError_t launch_conv0(const Conv0Args& a, int B, hipStream_t st) {
  hipLaunchKernelGGL(conv0_kernel, dim3(B * 2), dim3(256), a.L * sizeof(float), st, a);
  return hipGetLastError();
}
hipError_t launch_lstm(const LstmArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(lstm_kernel, dim3((a.M + 31) / 32), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_gather_ln(const GatherArgs& a, hipStream_t st) {
  long rows = (long)a.B * 2 * a.T;
  hipLaunchKernelGGL(gather_ln_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_attention(const AttnArgs& a, int B, hipStream_t st) {
  int nqb = (a.T + 63) / 64;
  size_t lds = (size_t)a.T * 64 * 2 * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(attention_kernel, dim3(B * 2 * 4), dim3(64 * nqb), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_head(const HeadArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(head_kernel, dim3((a.B + HB - 1) / HB), dim3(256), 0, st, a);
  return hipGetLastError();
}
