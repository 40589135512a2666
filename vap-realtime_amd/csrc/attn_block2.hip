// attn_block2_kernel — second generation of the fused short-window attention block (T <= 64), gfx950 only.
//
// Same arithmetic, element for element, as attn_block_kernel (fused_blocks.hip):
//     att = causal ALiBi attention of the 4 heads; x' = resid + att . Wproj^T -> xmid; LayerNorm(x'); optional qx = LN(x') . Wq_x^T
// reference: MultiHeadAttention.forward modules.py:82-110 (ALiBi :162-188), output projection :107, residual +
// ln_src_attn / ln_ffnetwork of TransformerLayer.forward :263-286, mha_cross.query.
//
// What changed is WHO does the work and WHEN.  Generation 1 gives every (stream, channel) window its own 4-wave workgroup, two
// per CU; at 256 streams that is exactly one round of 512 workgroups which all load, then all compute, in lock-step (HBM time
// and MFMA time add up), and at 4096 streams the two co-resident workgroups still pay each window's load latency in full.
// Here ONE persistent 8-wave workgroup per CU walks over its windows:
//   * wave (h, qt) = head h, 32-query tile qt: the two waves of a head sit on the same SIMD, so a SIMD always has a wave in
//     another phase of the same window (tile 0 has one key tile, tile 1 has two);
//   * the projection rows are split the same way (wave = 32 rows x 64 columns), so a window's projections take half as long;
//   * while a window's projection MFMAs run, the NEXT window's V tiles are already on their way (global -> registers before
//     the MFMAs, registers -> the other LDS buffer after them): a window's memory time hides behind its predecessor's compute.
// LDS: two [64][260] buffers (V tiles of the 4 heads, unpadded [64][64] each — their reads are lane-consecutive — then the
// attention output that feeds the projections, then the normalised rows for the cross-query projection) = 133 KB.
#include "fused_blocks.h"

namespace {

constexpr int LDA = 260;                 // sAtt row stride (256 + 4 pad: conflict-free ds_read_b128 A fragments)
constexpr int BUF = 64 * LDA;            // floats per buffer

__global__ __launch_bounds__(512, 1) void attn_block2_kernel(const AttnBlockArgs a, const int n_windows) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* red = lds + 2 * BUF;              // [2 qt][4 h][32 rows]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = w & 3, qt = w >> 2;        // head / column group, query-row tile
  const int l31 = lane & 31, hi = lane >> 5, kh = hi * 4;
  const int T = a.T;
  const bool ringed = a.ring_rot != nullptr;
  const float slope = exp2f(-2.0f * (float)(h + 1));
  const int ccol = h * 64 + l31;

  struct Win { int b, n, rot; long slab_q, slab_kv; };
  auto window = [&](int bc) {
    Win x;
    x.b = bc >> 1;
    x.n = a.bn[x.b];
    x.rot = ringed ? a.ring_rot[x.b] : 0;
    x.slab_q = ringed ? ((long)(a.ids ? a.ids[x.b] : x.b) * 2 + (bc & 1)) : (long)bc;
    x.slab_kv = ringed ? x.slab_q : (long)(a.swap_kv ? (bc ^ 1) : bc);
    return x;
  };
  // V tiles of a window: 4 heads x 64 rows x 64 floats = 4096 float4, 8 per thread; thread t, piece u -> float4 index
  // u*512 + t = (head, row j, quad q): rows >= n are zero
  auto v_load = [&](const Win& x, f32x4 (&vv)[8]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = u * 512 + tid, hh = idx >> 10, j = (idx >> 4) & 63, q4 = (idx & 15) * 4;
      const int jc = j < x.n ? j : x.n - 1;
      int r = jc + x.rot;
      r = r >= T ? r - T : r;
      vv[u] = *(const f32x4*)(a.v + (x.slab_kv * T + r) * a.ldkv + hh * 64 + q4);
    }
  };
  auto v_store = [&](const Win& x, const f32x4 (&vv)[8], float* buf) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = u * 512 + tid, hh = idx >> 10, j = (idx >> 4) & 63, q4 = (idx & 15) * 4;
      *(f32x4*)&buf[hh * 4096 + j * 64 + q4] = j < x.n ? vv[u] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x4 ring[16];
  auto wbase = [&](const float* wfrag) { return (const f32x4*)wfrag + (long)h * 32 * 2 * 64; };   // column group h, wave-uniform
  // acc[2] += A(32 rows of sA starting at row 32*qt, 256 k) . W^T for columns 64h .. 64h+63 (fragment-major weights through
  // the in-place register ring, as in ffn_block_kernel)
  auto mm = [&](f32x16 (&acc)[2], const float* sA, const float* wfrag, const float* next_wfrag) {
    const float* pa = sA + (qt * 32 + l31) * LDA + kh;
    const f32x4* wf = wbase(wfrag);
    const f32x4* wnext = next_wfrag ? wbase(next_wfrag) : wf;
    f32x4 p0 = *(const f32x4*)(pa), p1;
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
      const f32x4* nx = blk < 3 ? wf + (blk + 1) * 16 * 64 : wnext;
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8) {
        const int kn = (blk * 8 + k8 + 1) & 31;
        f32x4& ac = (k8 & 1) ? p1 : p0;
        f32x4& an = (k8 & 1) ? p0 : p1;
        an = *(const f32x4*)(pa + kn * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[s], ring[k8 * 2][s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[s], ring[k8 * 2 + 1][s], acc[1], 0, 0, 0);
        }
        ring[k8 * 2] = nx[(k8 * 2) * 64 + lane];
        ring[k8 * 2 + 1] = nx[(k8 * 2 + 1) * 64 + lane];
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep the refill behind its MFMAs (see ffn_block_kernel)
      }
    }
  };

  int cur = 0;
  int bc = blockIdx.x;
  if (bc >= n_windows) return;
  {   // first window: V tiles straight in
    const Win x = window(bc);
    f32x4 vv[8];
    v_load(x, vv);
    v_store(x, vv, lds);
  }
  for (; bc < n_windows; bc += gridDim.x, cur ^= 1) {
    float* buf = lds + cur * BUF;            // V tiles of this window, later its attention output / normalised rows
    float* nbuf = lds + (cur ^ 1) * BUF;
    const Win x = window(bc);
    const int n = x.n;
    const int nbc = bc + gridDim.x;
    const bool has_next = nbc < n_windows;
    auto prow = [&](int i) { int r = i + x.rot; return r >= T ? r - T : r; };
    const float* kp = a.k + x.slab_kv * T * a.ldkv + h * 64;
    // ---- this wave's (head, query tile): K fragments of key tile 0, Q fragments of tile qt ----
    const int i = qt * 32 + l31;             // query row of this lane
    f32x4 kf0[8], qf[8];
    {
      const int j0 = l31 < n ? l31 : n - 1;
      const float* k0 = kp + (long)prow(j0) * a.ldkv + kh;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) kf0[kc] = *(const f32x4*)(k0 + kc * 8);
      const int iq = i < n ? i : n - 1;
      const float* qp = a.q + (x.slab_q * T + prow(iq)) * a.ldq + h * 64 + kh;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) qf[kc] = *(const f32x4*)(qp + kc * 8);
    }
    __syncthreads();                         // V tiles of this window are in `buf` (and the previous window is done with it)
    const float* Vs = buf + h * 4096;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    const bool live = qt == 0 || n > 32;     // the second query tile holds valid rows only for n > 32
    if (live) {
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) qf[kc] *= 0.0625f;
      const bool use_j1 = qt == 1;
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int kc = 0; kc < 8; ++kc)
#pragma unroll
        for (int s = 0; s < 4; ++s) s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf0[kc][s], qf[kc][s], s0, 0, 0, 0);
      if (use_j1) {
        const int j1 = 32 + l31 < n ? 32 + l31 : n - 1;
        const float* k1 = kp + (long)prow(j1) * a.ldkv + kh;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          const f32x4 kf1 = *(const f32x4*)(k1 + kc * 8);
#pragma unroll
          for (int s = 0; s < 4; ++s) s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kf1[s], qf[kc][s], s1, 0, 0, 0);
        }
      }
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float v0 = s0[r] + slope * (float)j;
        v0 = ((j <= i) && (j < n)) ? v0 : -1e30f;
        s0[r] = v0;
        mx = fmaxf(mx, v0);
        const int j2 = j + 32;
        float v1 = s1[r] + slope * (float)j2;
        v1 = (use_j1 && (j2 <= i) && (j2 < n)) ? v1 : -1e30f;
        s1[r] = v1;
        mx = fmaxf(mx, v1);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p0 = s0[r] > -1e29f ? expf(s0[r] - mx) : 0.f;
        const float p1 = s1[r] > -1e29f ? expf(s1[r] - mx) : 0.f;
        s0[r] = p0; s1[r] = p1;
        sum += p0 + p1;
      }
      sum += __shfl_xor(sum, 32);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* va = &Vs[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + l31];
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[0], s0[r], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32], s0[r], o1, 0, 0, 0);
      }
      if (use_j1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* va = &Vs[(32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + l31];
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[0], s1[r], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32], s1[r], o1, 0, 0, 0);
        }
      }
      const float inv = i < n ? 1.0f / sum : 0.f;    // rows beyond the window -> zeros
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= inv; o1[r] *= inv; }
    }
    {   // projection weights: the ring fill flies under the barriers and the attention-output stores
      const f32x4* wf = wbase(a.wprojf);
#pragma unroll
      for (int u = 0; u < 16; ++u) ring[u] = wf[u * 64 + lane];
    }
    // next window's V tiles: requested now, they travel while this window's projections run
    Win xn = x;
    f32x4 vv[8];
    if (has_next) {
      xn = window(nbc);
      v_load(xn, vv);
    }
    __syncthreads();                         // every head is done with its V tile: `buf` becomes the attention output
    // O^T accumulator r <-> feature d = dt*32 + (r&3) + 8*(r>>2) + 4*hi of head h, query i = qt*32 + l31
    float* sAtt = buf;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      *(f32x4*)&sAtt[i * LDA + h * 64 + rr * 8 + kh] = f32x4{o0[rr * 4], o0[rr * 4 + 1], o0[rr * 4 + 2], o0[rr * 4 + 3]};
      *(f32x4*)&sAtt[i * LDA + h * 64 + 32 + rr * 8 + kh] = f32x4{o1[rr * 4], o1[rr * 4 + 1], o1[rr * 4 + 2], o1[rr * 4 + 3]};
    }
    __syncthreads();
    // ---- projection: rows 32qt .. 32qt+31 x columns 64h .. 64h+63 ----
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    mm(acc, sAtt, a.wprojf, a.wqxf ? a.wqxf : a.wprojf);
    if (has_next) v_store(xn, vv, nbuf);     // (nbuf's previous contents were last read before this window's first barrier)
    // residual
    {
      float rv[2][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int ii = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        ii = ii < T ? ii : T - 1;
        const float* rp = a.resid + (x.slab_q * T + prow(ii)) * 256 + ccol;
        rv[0][r] = rp[0];
        rv[1][r] = rp[32];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] += rv[0][r]; acc[1][r] += rv[1][r]; }
    }
    // LayerNorm over the 256 columns of each row (two-pass); partials of the four column groups of this row tile via LDS
    float* rd = red + qt * 128;
    float s[16], mean[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = half_sum(acc[0][r] + acc[1][r]);
    if (l31 == 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) rd[h * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      mean[r] = (rd[lr] + rd[32 + lr] + rd[64 + lr] + rd[96 + lr]) * (1.0f / 256.0f);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d0 = acc[0][r] - mean[r], d1 = acc[1][r] - mean[r];
      s[r] = half_sum(d0 * d0 + d1 * d1);
    }
    if (l31 == 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) rd[h * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[r];
    __syncthreads();                         // (also: every wave has finished reading sAtt in mm)
    const float g0 = a.ln_g[ccol], g1 = a.ln_g[ccol + 32], b0 = a.ln_b[ccol], b1 = a.ln_b[ccol + 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      const int ii = qt * 32 + lr;
      const float var = (rd[lr] + rd[32 + lr] + rd[64 + lr] + rd[96 + lr]) * (1.0f / 256.0f);
      const float rstd = rsqrtf(var + 1e-5f);
      const float y0 = (acc[0][r] - mean[r]) * rstd * g0 + b0;
      const float y1 = (acc[1][r] - mean[r]) * rstd * g1 + b1;
      if (ii < T) {
        float* xm = a.xmid + ((long)bc * T + ii) * 256 + ccol;
        xm[0] = acc[0][r]; xm[32] = acc[1][r];
        if (a.xn) {
          float* xo = a.xn + ((long)bc * T + ii) * 256 + ccol;
          xo[0] = y0; xo[32] = y1;
        }
      }
      if (a.wqxf) { sAtt[ii * LDA + ccol] = y0; sAtt[ii * LDA + ccol + 32] = y1; }
    }
    if (a.wqxf) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
      mm(acc, sAtt, a.wqxf, a.wprojf);       // (the ring refill of the last block already fetches the next window's Wproj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (ii < T) {
          float* qo = a.qx + ((long)bc * T + ii) * 256 + ccol;
          qo[0] = acc[0][r]; qo[32] = acc[1][r];
        }
      }
    }
  }
}

}  // namespace

hipError_t launch_attn_block2(const AttnBlockArgs& a, int B, hipStream_t st) {
  if (a.T > 64 || a.split) return hipErrorInvalidValue;
  static bool attr_set = false;
  const size_t lds = (size_t)(2 * BUF + 2 * 4 * 32) * sizeof(float);
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_block2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const int n_windows = B * 2;
  const int grid = n_windows < 256 ? n_windows : 256;   // one persistent workgroup per CU
  hipLaunchKernelGGL(attn_block2_kernel, dim3(grid), dim3(512), lds, st, a, n_windows);
  return hipGetLastError();
}
