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
// MT = 32-row sub-tiles per workgroup.  Shipped: MT = 1 (32 rows, 67 KB LDS, 2 workgroups / CU).  MT = 2 (64 rows, 133 KB, 1 workgroup / CU,
// every weight fragment feeds two MFMAs) measured slower at every batch size (one wave per SIMD: nothing fills the barrier phases) and is not
// instantiated.
template <int MT, int MODE = 0>
__global__ __launch_bounds__(256, MT == 1 ? 2 : 1) void ffn_block_kernel(const FfnArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BM = 32 * MT;
  float* sX = lds;                 // [BM][260] LN_ffn(x) tile (A operand of FFN1)
  float* sH = MODE == 2 ? lds : lds + BM * LDH;      // [BM][260] gelu chunk / raw x / LN_self(x).  MODE 2 (projection + LN + cross-q only) never touches sX: it is
                                                     // launched with ONE tile of LDS, and at 158 registers three of its workgroups share a CU (round 4)
  const int tid = threadIdx.x, lane = tid & 63;
  // wave index as an SGPR: weight-fragment addresses become scalar base + lane offset (SALU pointer
  // bumps, saddr loads) instead of per-lane 64-bit VALU adds — measured 2.8 -> ~1.6 VALU per MFMA
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5, kh = hi * 4;
  const int m0 = blockIdx.x * BM;
#ifdef VAPX_TRACE
  int stamp_k = 0;
  auto STAMP = [&]() {   // phase time stamps of wave 0 (debug build `make trace`: tools/ffn_trace.py)
    if (g.trace && tid == 0 && stamp_k < 32) g.trace[(long)blockIdx.x * 32 + stamp_k] = __builtin_amdgcn_s_memtime();
    ++stamp_k;
  };
  STAMP();
  if (g.trace && tid == 0) g.trace[(long)blockIdx.x * 32 + 28] = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz, chip-wide
#else
  auto STAMP = [] {};
#endif

  // acc[mt][2] += A[BM x 256] (LDS) . Wsub^T for this wave's 64 columns; wfrag = 256x256 fragment
  // block.  Weight fragments run through an in-place register ring one 8-kc block ahead (~4k MFMA
  // cycles of cover for the L2 latency), including across contraction boundaries (next_wfrag).
  f32x4 ring[16];
  auto wbase = [&](const float* wfrag) { return (const f32x4*)wfrag + (long)w * 32 * 2 * 64; };   // wave-uniform
  auto fetch = [&](const float* wfrag) {
    const f32x4* wf = wbase(wfrag);
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[i] = wf[i * 64 + lane];
  };
  fetch(MODE == 0 ? g.w0f : g.wprojf);   // the first weight fragments fly while the activation tile is loaded and normalised
  // layer 0 on long windows (mode 1 with resid_rot): the residual rows come straight from the embedding ring.  A tile is no longer than a window
  // (the launcher checks resid_T >= BM), so its rows lie in at most TWO (stream, channel) slabs: their ring slots and rotations are four
  // SCALAR loads, issued here, long before the projection needs them.  (Until round 5 every one of a lane's 16 rows looked its stream's slot
  // and rotation up by itself behind the projection: an integer division and two DEPENDENT vector loads per row, each waited for with
  // vmcnt(0) - 32 serialised L2 round trips per tile that also drained the next contraction's weight prefetch; tools/isa_waits.py.)
  int ring_bc0 = 0, ring_bc1 = 0, ring_slot0 = 0, ring_slot1 = 0, ring_rot0 = 0, ring_rot1 = 0;
  if (MODE == 1 && g.resid_rot) {
    ring_bc0 = __builtin_amdgcn_readfirstlane(m0 / g.resid_T);
    const int last_bc = (g.M - 1) / g.resid_T;
    ring_bc1 = ring_bc0 + 1 <= last_bc ? ring_bc0 + 1 : ring_bc0;
    ring_slot0 = g.resid_ids ? uniform_load(g.resid_ids, ring_bc0 >> 1) : ring_bc0 >> 1;
    ring_slot1 = g.resid_ids ? uniform_load(g.resid_ids, ring_bc1 >> 1) : ring_bc1 >> 1;
    ring_rot0 = uniform_load(g.resid_rot, ring_bc0 >> 1);
    ring_rot1 = uniform_load(g.resid_rot, ring_bc1 >> 1);
  }
  if constexpr (MODE != 0) {   // raw attention rows -> sH (A operand of the output projection)
    f32x4 xr[BM / 4];
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) {
      int m = m0 + (tid >> 6) + 4 * k;
      m = m < g.M ? m : g.M - 1;
      xr[k] = *(const f32x4*)(g.att + (long)m * 256 + lane * 4);
    }
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) *(f32x4*)&sH[((tid >> 6) + 4 * k) * LDH + lane * 4] = xr[k];
  } else
  {  // A operand of FFN1 = LayerNorm(xmid; ln_ffn), normalised while the tile is staged: every wave
     // loads whole rows (64 lanes x 16 B), so the row statistics are two wave reductions — the
     // producer (attention block) no longer writes a normalised copy to HBM at all
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
      *(f32x4*)&sX[((tid >> 6) + 4 * k) * LDH + lane * 4] = d * rstd * lg + lb;
    }
  }
  __syncthreads();
  STAMP();   // 1: tile staged + normalised

  auto mm = [&](f32x16(&acc)[MT][2], const float* A, const float* wfrag, const float* next_wfrag) {
    const float* pa = A + l31 * LDH + kh;
    const f32x4* wf = wbase(wfrag);
    const f32x4* wnext = next_wfrag ? wbase(next_wfrag) : wf;   // no successor: harmless re-read
    // A fragments ping-pong between two register sets (even / odd k-step) so that the LDS read of
    // step k+1 is in flight during the MFMAs of step k (with one set hipcc issues the read and
    // waits for it right away); sched_group_barrier fixes the order: 1 LDS read, 8 MFMAs, 2 loads.
    __builtin_amdgcn_s_setprio(2);   // MFMA phases outrank the co-resident workgroup's epilogues (measured +0.7 %)
    f32x4 a0[MT], a1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a0[mt] = *(const f32x4*)(pa + mt * 32 * LDH);
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
      const f32x4* nx = blk < 3 ? wf + (blk + 1) * 16 * 64 : wnext;
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8) {
        const int kn = (blk * 8 + k8 + 1) & 31;
        f32x4(&ac)[MT] = (k8 & 1) ? a1 : a0;
        f32x4(&an)[MT] = (k8 & 1) ? a0 : a1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) an[mt] = *(const f32x4*)(pa + mt * 32 * LDH + kn * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[mt][s], ring[k8 * 2][s], acc[mt][0], 0, 0, 0);
            acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[mt][s], ring[k8 * 2 + 1][s], acc[mt][1], 0, 0, 0);
          }
        ring[k8 * 2] = nx[(k8 * 2) * 64 + lane];
        ring[k8 * 2 + 1] = nx[(k8 * 2 + 1) * 64 + lane];
        __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);       // DS read(s) of the next A fragment first
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * MT, 0);   // then this step's MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);        // then the two ring refills
        // pin the step: left alone, hipcc sinks all 16 refills to the end of the block and waits for
        // them at the top of the next one (zero prefetch)
        __builtin_amdgcn_sched_barrier(0);
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
  // accumulator (mt, ns, r) <-> tile row lr = 32mt + (r&3) + 8*(r>>2) + 4*hi, chunk column w*64 + ns*32 + l31
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

  auto to_lds = [&](const f32x16(&acc)[MT][2], float* buf) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int lr = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        buf[lr * LDH + ccol] = acc[mt][0][r];
        buf[lr * LDH + ccol + 32] = acc[mt][1][r];
      }
  };
  auto to_sH = [&](const f32x16(&acc)[MT][2]) { to_lds(acc, sH); };
  // LayerNorm of the tile rows, ROW-PER-WAVE and in place on the fp32 tile in the LDS buffer `buf`: wave w owns rows w, w + 4, ..; a lane
  // holds 4 columns of a whole row, so mean and variance (two-pass) are wave reductions — no cross-wave partial sums through `red`, one
  // barrier pair instead of four, 16-byte LDS accesses, and the optional global copy leaves as 1 KiB-contiguous rows.  `parked`: the rows
  // are in `buf` already (the raw x_out tile the cross K,V contractions just read); otherwise the accumulator tile `v` is parked there first.
  // (Round 2 normalised in accumulator layout: 3.1 us of a 99 us tile alone, and the longest barrier chain of the block.)
  auto ln_rows = [&](const f32x16(&v)[MT][2], bool parked, float* buf, const float* gam, const float* bet, float* gout) {
    __syncthreads();          // every wave is done reading `buf` in the previous contraction
    if (!parked) {
      to_lds(v, buf);
      __syncthreads();
    }
    const f32x4 lg = *(const f32x4*)(gam + lane * 4), lb = *(const f32x4*)(bet + lane * 4);
    f32x4 x[BM / 4];
    float sm[BM / 4];
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) x[k] = *(const f32x4*)&buf[(w + 4 * k) * LDH + lane * 4];
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) sm[k] = wave_sum(x[k][0] + x[k][1] + x[k][2] + x[k][3]);
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) {
      x[k] = x[k] - sm[k] * (1.0f / 256.0f);
      sm[k] = x[k][0] * x[k][0] + x[k][1] * x[k][1] + x[k][2] * x[k][2] + x[k][3] * x[k][3];
    }
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) sm[k] = wave_sum(sm[k]);
#pragma unroll
    for (int k = 0; k < BM / 4; ++k) {
      const f32x4 y = x[k] * __builtin_amdgcn_rsqf(sm[k] * (1.0f / 256.0f) + 1e-5f) * lg + lb;   // (argument >= 1e-5: never denormal)
      const int row = w + 4 * k;
      *(f32x4*)&buf[row * LDH + lane * 4] = y;
      if (gout && m0 + row < g.M) *(f32x4*)(gout + (long)(m0 + row) * 256 + lane * 4) = y;
    }
    __syncthreads();
  };
  auto add_rows = [&](f32x16(&v)[MT][2], const float* src) {   // v += src tile (global [M][256] rows)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        m = m < g.M ? m : g.M - 1;
        const float* rp = src + (long)m * 256 + ccol;
        v[mt][0][r] += rp[0];
        v[mt][1][r] += rp[32];
      }
  };

  const int nq = g.wqkvf ? g.n_qkv_chunks : 0;
  const float* after_ffn = g.wkvxf ? g.wkvxf : (nq ? g.wqkvf : nullptr);
  f32x16 out[MT][2];
  if constexpr (MODE != 0) {
    // ---- attention output projection + residual: xmid = resid + att . Wproj^T (the separate GEMM of the long-window path) ----
    zero(out);
    mm(out, sH, g.wprojf, MODE == 1 ? g.w0f : after_ffn);
    if (MODE == 1 && g.resid_rot) {   // layer 0: residual rows straight from the embedding ring
      const int T = g.resid_T;
      const int bc0 = ring_bc0, bc1 = ring_bc1, slot0 = ring_slot0, slot1 = ring_slot1, rot0 = ring_rot0, rot1 = ring_rot1;   // (loaded at kernel entry)
      const int split = (bc0 + 1) * T;                           // first row of the second slab
      const float* rp[MT][16];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int m = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          m = m < g.M ? m : g.M - 1;
          const bool second = m >= split;
          const int bc = second ? bc1 : bc0, i = m - bc * T;
          int rr = i + (second ? rot1 : rot0);
          rr = rr >= T ? rr - T : rr;
          rp[mt][r] = g.resid + (((long)(second ? slot1 : slot0) * 2 + (bc & 1)) * T + rr) * 256 + ccol;
        }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          out[mt][0][r] += rp[mt][r][0];
          out[mt][1][r] += rp[mt][r][32];
        }
    } else
    add_rows(out, g.resid);
    // mode 2 hands xmid to the next block through HBM; in mode 1 its only consumer is this workgroup (the residual of the FFN), so it
    // stays where it is — in the accumulators the FFN2 products are added to — and never travels (2 x 2.1 GB per launch at C3)
    if constexpr (MODE == 2) store_global(out, g.xmid_out, 256, 0);
    if constexpr (MODE == 1) ln_rows(out, false, sX, g.lnf_g, g.lnf_b, nullptr);   // A operand of FFN1
  }
  if constexpr (MODE != 2) {
  // ---- feed-forward: x = xmid + gelu(xn W0^T) W3^T, hidden processed in 3 chunks of 256 ----
  if constexpr (MODE == 0) zero(out);   // (mode 1: the accumulators start from xmid)
  for (int c = 0; c < 3; ++c) {
    f32x16 hacc[MT][2];
    zero(hacc);
    mm(hacc, sX, g.w0f + (long)c * 65536, g.w3f + (long)c * 65536);
    STAMP();   // 2 + 4c: FFN1 chunk MFMAs done
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        hacc[mt][0][r] = gelu_fast(hacc[mt][0][r]);
        hacc[mt][1][r] = gelu_fast(hacc[mt][1][r]);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    STAMP();   // 3 + 4c: GELU done
    __syncthreads();          // every wave is done reading the previous chunk from sH
    to_sH(hacc);
    __syncthreads();
    STAMP();   // 4 + 4c: hidden chunk in LDS
    mm(out, sH, g.w3f + (long)c * 65536, c < 2 ? g.w0f + (long)(c + 1) * 65536 : after_ffn);
    STAMP();   // 5 + 4c: FFN2 chunk MFMAs done
  }
  if constexpr (MODE == 0) add_rows(out, g.xmid);
  STAMP();   // 14: residual added
  store_global(out, g.xout, 256, 0);
  STAMP();   // 15: x_out stored

  // ---- next layer's cross K,V from the RAW layer output ----
  if (g.wkvxf) {
    __syncthreads();
    to_sH(out);
    __syncthreads();
    STAMP();   // 16
    for (int nc = 0; nc < 2; ++nc) {
      f32x16 acc[MT][2];
      zero(acc);
      mm(acc, sH, g.wkvxf + (long)nc * 65536, nc == 0 ? g.wkvxf + 65536 : (nq ? g.wqkvf : nullptr));
      STAMP();   // 17, 19
      store_global(acc, g.kvx, 512, nc * 256);
      STAMP();   // 18, 20
    }
  }
  }   // MODE != 2
  // ---- next layer's self Q,K,V from LayerNorm(x) (or just the normalised rows) ----
  if (nq || g.xn_out) {
    ln_rows(out, MODE != 2 && g.wkvxf != nullptr, sH, g.ln_g, g.ln_b, g.xn_out);   // (after the cross K,V contractions the raw rows sit in sH already)
    STAMP();   // 21: LN_self rows in LDS
    for (int nc = 0; nc < nq; ++nc) {
      f32x16 acc[MT][2];
      zero(acc);
      mm(acc, sH, g.wqkvf + (long)nc * 65536, nc + 1 < nq ? g.wqkvf + (long)(nc + 1) * 65536 : nullptr);
      STAMP();   // 22, 24, 26
      store_global(acc, g.qkv, nq * 256, nc * 256);
      STAMP();   // 23, 25, 27
    }
  }
#ifdef VAPX_TRACE
  if (g.trace && tid == 0) g.trace[(long)blockIdx.x * 32 + 29] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------------------------------------
// attn_block_kernel: one workgroup per (stream, channel), wave = head.
//     att   = causal ALiBi attention of the 4 heads           (as attention_mfma_kernel)
//     x'    = resid + att . Wproj^T                           -> xmid (global)
//     xn    = LayerNorm(x'; g, b)                             -> xn   (global)
//     qx    = xn . Wq_x^T                (optional)           -> qx   (global; cross-attention queries)
// reference: MultiHeadAttention.forward modules.py:82-110 incl. the output projection :107, the
// residual + ln_src_attn / ln_ffnetwork of TransformerLayer.forward :263-286, and mha_cross.query.
// K fragments come straight from global (A operand, reused by both query tiles), V sits in LDS
// (zero rows beyond n), and once every head has finished P.V the same LDS bytes are re-used for
// the [64 x 256] attention output that feeds the projection MFMAs — 70 KB, two workgroups per CU.
// Only for T <= 64 (two 32-row tiles); longer windows use the unfused kernels.
// ------------------------------------------------------------------------------------------------
constexpr int KV_LD2 = 68;

// SPLIT (opt-in, VAPX_FLAG_SPLIT_F16): the two projections run as fp32-accurate 3-term split products on the f16 matrix
// cores (see ffn_block_f16x3.hip): the attention output / LN rows are kept in LDS as f16 (hi, lo) pairs, the weights
// arrive as pre-split f16 fragments (weights.frag_pack_f16x3).  The attention itself (S, softmax, P.V) stays fp32 MFMA.
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr int ALD16 = 264;   // halves per sAtt row in SPLIT mode

// QX: the launch also produces the cross-attention queries, qx = LayerNorm(x'; ln_src) . Wq_x^T (self-attention half of a stereo layer).
// Without QX nothing in the launch consumes LayerNorm(x') — the FFN block normalises xmid itself when it stages its tile — so the
// block stores x' straight from the accumulators and skips the LayerNorm altogether (3 of the 5 attention launches of a tick).
// R52 (round 5, fp32 path, T <= 52 — the reference's 2.5 s x 20 Hz window is T = 50): the projections run on 32 + 5 x 4 rows instead of
// 2 x 32.  Rows 32.. of the window go through v_mfma_f32_4x4x1_16B_f32 (sixteen 4 x 4 blocks per instruction, the same 64 FLOP / clk /
// SIMD as the 32x32x2 MFMA): block b of lane group 4 b .. 4 b + 3 pairs four ROWS (A: lane 4 b + i = row i) with four COLUMNS (B: lane
// 4 b + j = column j) — and a lane's B value is exactly what the 32x32x2 weight fragment already holds in that lane (column 32 ns + (lane &
// 31), k = 8 kc + 4 (lane >> 5) + s), so the SAME ring registers feed both shapes and nothing is streamed twice.  Lanes 0-31 and 32-63
// accumulate the two k-halves of a k-step for the same 32 columns; one cross-half add at the end.  40 small MFMAs (320 clk) replace the 8
// big ones (512 clk) of the second row tile in every k-step: -19 % of the projection's matrix-core time for T = 50 (14 of 64 rows were padding).
template <bool SPLIT, bool QX, bool R52 = false>
__global__ __launch_bounds__(256, 2) void attn_block_kernel(const AttnBlockArgs a) {
  static_assert(!(SPLIT && R52), "the 4-row tiles exist for the fp32 fragments only");
  __shared__ __attribute__((aligned(16))) float lds[4 * 64 * KV_LD2];
  __shared__ float vmx[4];                  // SPLIT: max |V| of each head's tile (bounds the attention output: a convex combination of V rows)
  float* sAtt = lds;                        // [64][260] (aliases the V tiles after a barrier)
  _Float16* sAh = (_Float16*)lds;           // SPLIT: [64][264] halves, hi then lo
  _Float16* sAl = sAh + 64 * ALD16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar wave / head index (see ffn_block_kernel)
  const int l31 = lane & 31, hi = lane >> 5, kh = hi * 4;
  const int bc = blockIdx.x, b = bc >> 1, T = a.T;
  const int n = a.bn[b];
  const int kvbc = a.swap_kv ? (bc ^ 1) : bc;
  float* Vs = lds + h * 64 * KV_LD2;
  // Layer 0 reads Q|K|V and the residual straight from the per-stream rings (no chronological copy):
  // slab = stream slot * 2 + channel, logical row i lives in ring slot (i + rot) % T.
  const bool ringed = a.ring_rot != nullptr;
  const int rot = ringed ? a.ring_rot[b] : 0;
  const long slab_q = ringed ? ((long)(a.ids ? a.ids[b] : b) * 2 + (bc & 1)) : (long)bc;
  const long slab_kv = ringed ? slab_q : (long)kvbc;
  auto prow = [&](int i) { int r = i + rot; return r >= T ? r - T : r; };
  const float* vp = a.v + slab_kv * T * a.ldkv + h * 64;
  const float* kp = a.k + slab_kv * T * a.ldkv + h * 64;
  const bool two = n > 32;                  // second query/key tile holds valid rows
#ifdef VAPX_TRACE
  int stamp_k = 0;
  auto STAMP = [&]() {   // phase time stamps of wave 0 (debug build `make trace`: tools/attn_trace.py)
    if (a.trace && tid == 0 && stamp_k < 28) a.trace[(long)blockIdx.x * 32 + stamp_k] = __builtin_amdgcn_s_memtime();
    ++stamp_k;
  };
  STAMP();   // 0: entry
  if (a.trace && tid == 0) a.trace[(long)blockIdx.x * 32 + 28] = __builtin_amdgcn_s_memrealtime();
#else
  auto STAMP = [] {};
#endif
  // EVERY global load of the attention phase is issued before anything waits (one memory latency for the whole phase): this head's
  // V tile (-> LDS), then the A operands of S^T = K.Q^T (key row j, k-slots kc*8 + 4*hi ..+3; rows >= n are clamped, their scores
  // are masked below) and the Q fragments of BOTH query tiles.  V goes first: loads return in order, so the LDS stores that wait
  // for V leave the 32 K / Q loads in flight.  (Round 2 fetched the second tile's K / Q fragments inside its MFMA loop: ten exposed
  // L2 / HBM latencies per workgroup, the largest single cost of the block.)
  // SPLIT (round 5): the attention phase runs on the f16 matrix cores too — S^T = K.Q^T and O^T = V^T.P^T as 3-term split products (12
  // v_mfma_f32_32x32x16_f16 per 32 x 32 tile pair instead of 32 v_mfma_f32_32x32x2_f32 of twice the length).  A lane half then holds 8 consecutive
  // features of its row per 16-feature chunk c: d = 16 c + 8 hi + 0..7, fetched as the two f32x4 of fragment slots 2 c, 2 c + 1.
  auto koff = [&](int kc) { return SPLIT ? (kc >> 1) * 16 + hi * 8 + (kc & 1) * 4 : kc * 8 + kh; };
  float vmax_h = 0.f;                       // SPLIT: max |V| of this head's tile
  f32x4 kf0[8], qf0[8], kf1[8], qf1[8];
  {
    f32x4 vv[16];
    const int q4 = (lane & 15) * 4, jb = lane >> 4;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int j = u * 4 + jb;
      int jc = j < n ? j : n - 1;
      vv[u] = *(const f32x4*)(vp + (long)prow(jc) * a.ldkv + q4);
    }
    {
      const int j0 = l31 < n ? l31 : n - 1;
      const float* k0 = kp + (long)prow(j0) * a.ldkv;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) kf0[kc] = *(const f32x4*)(k0 + koff(kc));
      const float* qp = a.q + (slab_q * T + prow(j0)) * a.ldq + h * 64;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) qf0[kc] = *(const f32x4*)(qp + koff(kc));
    }
    if (two) {
      const int j1 = 32 + l31 < n ? 32 + l31 : n - 1;
      const float* k1 = kp + (long)prow(j1) * a.ldkv;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) kf1[kc] = *(const f32x4*)(k1 + koff(kc));
      const float* qp = a.q + (slab_q * T + prow(j1)) * a.ldq + h * 64;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) qf1[kc] = *(const f32x4*)(qp + koff(kc));
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the K / Q loads above the LDS stores that wait for V
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      int j = u * 4 + jb;
      *(f32x4*)&Vs[j * KV_LD2 + q4] = j < n ? vv[u] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (SPLIT) {   // the cross-attention V rows come from RAW residual-stream rows: their magnitude is the input's to decide
      float m = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(vv[u][0]), fabsf(vv[u][1])), fmaxf(fabsf(vv[u][2]), fabsf(vv[u][3]))));
      m = wave_max(m);
      if (lane == 0) vmx[h] = m;
      vmax_h = m;
    }
  }
  STAMP();   // 1: V tile in LDS (first memory latency)
  const float slope = exp2f(-2.0f * (float)(h + 1));

  // S^T tile = K_tile . Q_tile^T (unscaled: the 1/16 is a power of two and rides in the softmax fma, bit-identical)
  auto scores = [&](f32x16& s, const f32x4 (&kf)[8], const f32x4 (&qf)[8]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kc = 0; kc < 8; ++kc)
#pragma unroll
      for (int c = 0; c < 4; ++c) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kc][c], qf[kc][c], s, 0, 0, 0);
  };
  // accumulator r of a key tile jt <-> key j = 32 jt + (r&3) + 8 (r>>2) + 4 hi, query i = 32 it + l31 (one query column per lane pair:
  // the softmax is lane-local + one exchange across the halves).  s := exp(score - max), returns the row sum.
  // (key j = C_r + 4 hi with C_r a compile-time constant: the causal / window tests compare C_r with the per-lane i - 4 hi, n - 4 hi, so no
  // per-element index register exists)
  const float hi4f = (float)kh;
  // (SPLIT: the scores carry the power-of-two scales of their K and Q tiles; uk un-scales the accumulator, uq rides with the 1/16 — both exact)
  auto masked = [&](f32x16& s, int jt, int i, float& mx, float uk = 1.0f, float uq = 1.0f) {
    const int i4 = i - kh, n4 = n - kh;
    const float sixteenth = SPLIT ? 0.0625f * uq : 0.0625f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = jt * 32 + (r & 3) + 8 * (r >> 2);
      float v = fmaf(SPLIT ? s[r] * uk : s[r], sixteenth, slope * ((float)c + hi4f));
      v = ((c <= i4) && (c < n4)) ? v : -1e30f;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
  };
  auto expsum = [&](f32x16& s, float mx) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = s[r] > -1e29f ? __expf(s[r] - mx) : 0.f;   // v_exp_f32 path: arguments <= 0, |rel err| ~1e-6 (as attention_long2_kernel)
      s[r] = pr;
      sum += pr;
    }
    return sum;
  };
  // O^T[d][i] += sum_j V[j][d] P^T[j][i] for the 32 keys of tile jt: accumulator register r of S^T IS the B operand of MFMA step r
  auto pv = [&](f32x16& o0, f32x16& o1, const f32x16& pr, int jt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* va = &Vs[(jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * KV_LD2 + l31];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[0], pr[r], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va[32], pr[r], o1, 0, 0, 0);
    }
  };
  f32x16 oa0, oa1, ob0, ob1;   // query tile 0 / 1, feature tile 0 / 1 (O^T, scaled by 1 / rowsum)
#pragma unroll
  for (int r = 0; r < 16; ++r) { oa0[r] = 0.f; oa1[r] = 0.f; ob0[r] = 0.f; ob1[r] = 0.f; }
  if constexpr (SPLIT) {
    // ---- f16 (hi, lo) operands of the three score tiles, each tile scaled by a power of two from its own maximum ----
    auto split8s = [](const f32x4& x0, const f32x4& x1, float sc, h16x8& fh, h16x8& fl) {
      const f32x4 y0 = x0 * sc, y1 = x1 * sc;
      const h16x4 h0 = __builtin_convertvector(y0, h16x4), h1 = __builtin_convertvector(y1, h16x4);
      const h16x4 l0 = __builtin_convertvector(y0 - __builtin_convertvector(h0, f32x4), h16x4);
      const h16x4 l1 = __builtin_convertvector(y1 - __builtin_convertvector(h1, f32x4), h16x4);
      fh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
      fl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto tile_scale = [&](const f32x4 (&f)[8]) {
      float m = 0.f;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) m = fmaxf(m, fmaxf(fmaxf(fabsf(f[kc][0]), fabsf(f[kc][1])), fmaxf(fabsf(f[kc][2]), fabsf(f[kc][3]))));
      return pow2_scale_for(wave_max(m));
    };
    auto conv = [&](const f32x4 (&f)[8], float sc, h16x8 (&fh)[4], h16x8 (&fl)[4]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) split8s(f[2 * c], f[2 * c + 1], sc, fh[c], fl[c]);
    };
    auto scores16 = [&](f32x16& s, const h16x8 (&Kh)[4], const h16x8 (&Kl)[4], const h16x8 (&Qh)[4], const h16x8 (&Ql)[4]) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kh[c], Qh[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kl[c], Qh[c], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_f16(Kh[c], Ql[c], s, 0, 0, 0);
      }
    };
    h16x8 K0h[4], K0l[4], Q0h[4], Q0l[4], K1h[4], K1l[4], Q1h[4], Q1l[4];
    const float sK0 = tile_scale(kf0), sQ0 = tile_scale(qf0);
    conv(kf0, sK0, K0h, K0l);
    conv(qf0, sQ0, Q0h, Q0l);
    float sK1 = 1.0f, sQ1 = 1.0f;
    if (two) {
      sK1 = tile_scale(kf1); sQ1 = tile_scale(qf1);
      conv(kf1, sK1, K1h, K1l);
      conv(qf1, sQ1, Q1h, Q1l);
    }
    const float uK0 = __builtin_amdgcn_rcpf(sK0), uQ0 = __builtin_amdgcn_rcpf(sQ0), uK1 = __builtin_amdgcn_rcpf(sK1), uQ1 = __builtin_amdgcn_rcpf(sQ1);   // exact: powers of two
    f32x16 sa, sb0, sb1;
    scores16(sa, K0h, K0l, Q0h, Q0l);
    if (two) { scores16(sb0, K0h, K0l, Q1h, Q1l); scores16(sb1, K1h, K1l, Q1h, Q1l); }
    // ---- O^T = V^T . P^T: the accumulator registers 8 c .. 8 c + 7 of a score tile are the 8 k-slots of this lane half in key chunk c, i.e. the
    // keys 32 jt + 16 c + 8 (e >> 2) + 4 hi + (e & 3), e = 0 .. 7; the V^T operand of lane (feature d, hi) gathers exactly those keys from the fp32
    // tile in LDS (the k order inside a chunk is free as long as both operands agree), scaled by a power of two from the head's max |V| ----
    const float sV = pow2_scale_for(vmax_h), uV = __builtin_amdgcn_rcpf(sV);
    auto vt_frags = [&](int jt, h16x8 (&Vh)[2][2], h16x8 (&Vl)[2][2]) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const float* vb = &Vs[(jt * 32 + c * 16 + 4 * hi) * KV_LD2 + dt * 32 + l31];
          const f32x4 v0 = {vb[0], vb[KV_LD2], vb[2 * KV_LD2], vb[3 * KV_LD2]};
          const f32x4 v1 = {vb[8 * KV_LD2], vb[9 * KV_LD2], vb[10 * KV_LD2], vb[11 * KV_LD2]};
          split8s(v0, v1, sV, Vh[c][dt], Vl[c][dt]);
        }
    };
    auto pv16 = [&](f32x16& o0, f32x16& o1, const f32x16& pr, const h16x8 (&Vh)[2][2], const h16x8 (&Vl)[2][2]) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 p0 = {pr[8 * c], pr[8 * c + 1], pr[8 * c + 2], pr[8 * c + 3]}, p1 = {pr[8 * c + 4], pr[8 * c + 5], pr[8 * c + 6], pr[8 * c + 7]};
        h16x8 ph, pl;
        split8s(p0, p1, 1.0f, ph, pl);          // P in [0, 1]: no scale
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Vh[c][0], ph, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Vh[c][1], ph, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Vl[c][0], ph, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Vl[c][1], ph, o1, 0, 0, 0);
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Vh[c][0], pl, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Vh[c][1], pl, o1, 0, 0, 0);
      }
    };
    h16x8 Vh[2][2], Vl[2][2];
    vt_frags(0, Vh, Vl);
    {
      const int i = l31;
      float mx = -1e30f;
      masked(sa, 0, i, mx, uK0, uQ0);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = expsum(sa, mx);
      sum += __shfl_xor(sum, 32);
      pv16(oa0, oa1, sa, Vh, Vl);
      const float inv = i < n ? uV / sum : 0.f;    // rows beyond the window -> zeros
#pragma unroll
      for (int r = 0; r < 16; ++r) { oa0[r] *= inv; oa1[r] *= inv; }
    }
    if (two) {
      const int i = 32 + l31;
      float mx = -1e30f;
      masked(sb0, 0, i, mx, uK0, uQ1);
      masked(sb1, 1, i, mx, uK1, uQ1);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = expsum(sb0, mx) + expsum(sb1, mx);
      sum += __shfl_xor(sum, 32);
      pv16(ob0, ob1, sb0, Vh, Vl);
      vt_frags(1, Vh, Vl);
      pv16(ob0, ob1, sb1, Vh, Vl);
      const float inv = i < n ? uV / sum : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { ob0[r] *= inv; ob1[r] *= inv; }
    }
  } else {
    // the three score tiles back to back (96 MFMAs): the softmax VALU of tile 0 then overlaps the matrix pipe draining
    f32x16 sa, sb0, sb1;
    scores(sa, kf0, qf0);
    if (two) { scores(sb0, kf0, qf1); scores(sb1, kf1, qf1); }
    {
      const int i = l31;
      float mx = -1e30f;
      masked(sa, 0, i, mx);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = expsum(sa, mx);
      sum += __shfl_xor(sum, 32);
      pv(oa0, oa1, sa, 0);
      const float inv = i < n ? 1.0f / sum : 0.f;    // rows beyond the window -> zeros
#pragma unroll
      for (int r = 0; r < 16; ++r) { oa0[r] *= inv; oa1[r] *= inv; }
    }
    if (two) {
      const int i = 32 + l31;
      float mx = -1e30f;
      masked(sb0, 0, i, mx);
      masked(sb1, 1, i, mx);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = expsum(sb0, mx) + expsum(sb1, mx);
      sum += __shfl_xor(sum, 32);
      pv(ob0, ob1, sb0, 0);
      pv(ob0, ob1, sb1, 1);
      const float inv = i < n ? 1.0f / sum : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { ob0[r] *= inv; ob1[r] *= inv; }
    }
  }
  STAMP();   // 2: attention (scores, softmax, P.V) done
  // projection weights: start the ring fill now, it flies under the barrier and the sAtt stores
  const int w = h;
  f32x4 ring[16];
  auto wbase = [&](const float* wfrag) { return (const f32x4*)wfrag + (long)w * 32 * 2 * 64; };   // wave-uniform
  {
    const f32x4* wf = wbase(a.wprojf);
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[i] = wf[i * 64 + lane];
  }
  // acc[mt*2 + ns][r] <-> row i = mt*32 + (r&3) + 8*(r>>2) + 4*hi, column 64w + 32ns + l31.  The projection accumulates ON TOP of the
  // residual rows: their 64 loads per lane are issued here, fly under the barriers and the sAtt stores, and are consumed by the first
  // MFMA of each accumulator — no separate load-wait-add phase after the projection (and 64 registers less at the LayerNorm).
  // SPLIT: the f16x3 weight fragments interleave the output columns of a wave's two tiles (weights.frag_pack_f16x3: tile ns holds columns
  // 2 l + ns), so a lane's two accumulators are adjacent columns; fp32 fragments keep tile ns = columns 32 ns + l
  const int ccol = SPLIT ? w * 64 + 2 * l31 : w * 64 + l31;
  constexpr int CS = SPLIT ? 1 : 32;   // column distance between acc[2 mt] and acc[2 mt + 1]
  f32x16 acc[4];
  f32x4 g4[5][2];                      // R52: rows 32 + 4 g + r (r = register), column 64 w + 32 ns + l31; the two lane halves hold the two k-halves
  __builtin_amdgcn_sched_barrier(0);   // (the row offsets below must not be hoisted above the attention phase: 32 live registers)
  {
    const float* rbase = a.resid + slab_q * T * 256;           // workgroup-uniform base + 32-bit per-lane offsets
    const int h4 = opaque_vgpr(kh);
    if constexpr (R52) {               // the residual rides in lane half 0 only (the halves are added at the end)
#pragma unroll
      for (int gI = 0; gI < 5; ++gI)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int i = 32 + 4 * gI + r;
          i = i < T ? i : T - 1;
          const unsigned off = (unsigned)prow(i) * 256u + (unsigned)ccol;
          const float v0 = rbase[off], v1 = rbase[off + CS];
          g4[gI][0][r] = hi ? 0.f : v0;
          g4[gI][1][r] = hi ? 0.f : v1;
        }
    }
#pragma unroll
    for (int mt = 0; mt < (R52 ? 1 : 2); ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int i = mt * 32 + (r & 3) + 8 * (r >> 2) + h4;
        i = i < T ? i : T - 1;
        const unsigned off = (unsigned)prow(i) * 256u + (unsigned)ccol;
        acc[mt * 2][r] = rbase[off];
        acc[mt * 2 + 1][r] = rbase[off + CS];
      }
  }
  STAMP();   // 3: residual + weight-ring loads issued
  __syncthreads();             // every head is done with its V tile: the bytes become sAtt
  STAMP();   // 4: barrier passed (all heads done)
  // SPLIT: |att| <= max |V| over the slab (softmax weights are a convex combination).  The attention output goes into the f16 (hi, lo)
  // tile as att * osc with osc a power of two keeping it below 2^14, the accumulator carries 2^8 osc x and is un-scaled exactly at the end.
  float osc = 1.0f;
  if constexpr (SPLIT) {
    osc = pow2_scale_for(fmaxf(fmaxf(vmx[0], vmx[1]), fmaxf(vmx[2], vmx[3])));
    const float up = 256.0f * osc;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] *= up;
  }
  // O^T accumulator r <-> feature d = dt*32 + (r&3) + 8*(r>>2) + 4*hi, query i = it*32 + l31
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const f32x4 v00 = f32x4{oa0[rr * 4], oa0[rr * 4 + 1], oa0[rr * 4 + 2], oa0[rr * 4 + 3]};
    const f32x4 v01 = f32x4{oa1[rr * 4], oa1[rr * 4 + 1], oa1[rr * 4 + 2], oa1[rr * 4 + 3]};
    const f32x4 v10 = f32x4{ob0[rr * 4], ob0[rr * 4 + 1], ob0[rr * 4 + 2], ob0[rr * 4 + 3]};
    const f32x4 v11 = f32x4{ob1[rr * 4], ob1[rr * 4 + 1], ob1[rr * 4 + 2], ob1[rr * 4 + 3]};
    if constexpr (SPLIT) {
      auto put = [&](int row, int col, f32x4 v) {
        v *= osc;
        const h16x4 hh = __builtin_convertvector(v, h16x4);
        const h16x4 ll = __builtin_convertvector(v - __builtin_convertvector(hh, f32x4), h16x4);
        *(h16x4*)&sAh[row * ALD16 + col] = hh;
        *(h16x4*)&sAl[row * ALD16 + col] = ll;
      };
      put(l31, h * 64 + rr * 8 + kh, v00);
      put(l31, h * 64 + 32 + rr * 8 + kh, v01);
      put(32 + l31, h * 64 + rr * 8 + kh, v10);
      put(32 + l31, h * 64 + 32 + rr * 8 + kh, v11);
    } else {
      *(f32x4*)&sAtt[l31 * 260 + h * 64 + rr * 8 + kh] = v00;
      *(f32x4*)&sAtt[l31 * 260 + h * 64 + 32 + rr * 8 + kh] = v01;
      *(f32x4*)&sAtt[(32 + l31) * 260 + h * 64 + rr * 8 + kh] = v10;
      *(f32x4*)&sAtt[(32 + l31) * 260 + h * 64 + 32 + rr * 8 + kh] = v11;
    }
  }
  __syncthreads();

  // ---- projection(s): [64 x 256] (LDS) . W^T, wave h owns columns 64h..64h+63 ----
  auto mm = [&](f32x16(&acc)[4], const float* wfrag, const float* next_wfrag, float split_inv) {
    const float* pa = sAtt + l31 * 260 + kh;
    const f32x4* wf = wbase(wfrag);
    const f32x4* wnext = next_wfrag ? wbase(next_wfrag) : wf;
    if constexpr (SPLIT) {
      // fragments [16 kc][2 ns][2 hi/lo][64 lane] x 16 B; ring slot k4 holds (ns0 hi, ns0 lo, ns1 hi, ns1 lo) of a k-chunk
      const _Float16* pah = sAh + l31 * ALD16 + hi * 8;
      const _Float16* pal = sAl + l31 * ALD16 + hi * 8;
#pragma unroll 1
      for (int blk = 0; blk < 4; ++blk) {
        const f32x4* nx = blk < 3 ? wf + (blk + 1) * 16 * 64 : wnext;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const int kc = blk * 4 + k4;
          const h16x8 a0h = *(const h16x8*)(pah + kc * 16), a0l = *(const h16x8*)(pal + kc * 16);
          const h16x8 a1h = *(const h16x8*)(pah + 32 * ALD16 + kc * 16), a1l = *(const h16x8*)(pal + 32 * ALD16 + kc * 16);
          const h16x8 b0h = __builtin_bit_cast(h16x8, ring[k4 * 4 + 0]), b0l = __builtin_bit_cast(h16x8, ring[k4 * 4 + 1]);
          const h16x8 b1h = __builtin_bit_cast(h16x8, ring[k4 * 4 + 2]), b1l = __builtin_bit_cast(h16x8, ring[k4 * 4 + 3]);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b0h, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b1h, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b0h, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b1h, acc[3], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b0h, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b1h, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b0h, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b1h, acc[3], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b0l, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b1l, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b0l, acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b1l, acc[3], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i) ring[k4 * 4 + i] = nx[(k4 * 4 + i) * 64 + lane];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= split_inv;   // weights are packed as 2^8 w (and the operand rows may carry a power-of-two scale)
      return;
    }
    if constexpr (R52) {
      // row tile 0 as below (A fragment one k-step ahead); rows 32.. in five 4-row groups: lane L reads row 32 + 4 g + (L & 3) at the k-half
      // (L >> 5) of the step — 8 distinct addresses per ds_read_b128 — at the top of the step, to land under tile 0's eight MFMAs
      const float* pg = sAtt + (32 + (lane & 3)) * 260 + kh;
      f32x4 q0, q1;
      q0 = *(const f32x4*)(pa);
#pragma unroll 1
      for (int blk = 0; blk < 4; ++blk) {
        const f32x4* nx = blk < 3 ? wf + (blk + 1) * 16 * 64 : wnext;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
          const int kc = blk * 8 + k8, kn = (kc + 1) & 31;
          f32x4& ac = (k8 & 1) ? q1 : q0;
          f32x4& an = (k8 & 1) ? q0 : q1;
          an = *(const f32x4*)(pa + kn * 8);
          f32x4 ag[5];
#pragma unroll
          for (int gI = 0; gI < 5; ++gI) ag[gI] = *(const f32x4*)(pg + gI * 4 * 260 + kc * 8);
#pragma unroll
          for (int sI = 0; sI < 4; ++sI) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[sI], ring[k8 * 2][sI], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[sI], ring[k8 * 2 + 1][sI], acc[1], 0, 0, 0);
          }
#pragma unroll
          for (int sI = 0; sI < 4; ++sI)
#pragma unroll
            for (int gI = 0; gI < 5; ++gI) {
              g4[gI][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ag[gI][sI], ring[k8 * 2][sI], g4[gI][0], 0, 0, 0);
              g4[gI][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ag[gI][sI], ring[k8 * 2 + 1][sI], g4[gI][1], 0, 0, 0);
            }
          ring[k8 * 2] = nx[(k8 * 2) * 64 + lane];
          ring[k8 * 2 + 1] = nx[(k8 * 2 + 1) * 64 + lane];
          __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 48, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the two k-halves of every 4-row group meet: afterwards BOTH lane halves hold the finished value of (row, column 32 ns + l31)
#pragma unroll
      for (int gI = 0; gI < 5; ++gI)
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
#pragma unroll
          for (int r = 0; r < 4; ++r) g4[gI][ns][r] += __shfl_xor(g4[gI][ns][r], 32);
      return;
    }
    // A fragments ping-pong between two register sets so the LDS reads of step k+1 fly under the
    // MFMAs of step k (see ffn_block_kernel)
    f32x4 p0[2], p1[2];
    p0[0] = *(const f32x4*)(pa);
    p0[1] = *(const f32x4*)(pa + 32 * 260);
#pragma unroll 1
    for (int blk = 0; blk < 4; ++blk) {
      const f32x4* nx = blk < 3 ? wf + (blk + 1) * 16 * 64 : wnext;
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8) {
        const int kn = (blk * 8 + k8 + 1) & 31;
        f32x4(&ac)[2] = (k8 & 1) ? p1 : p0;
        f32x4(&an)[2] = (k8 & 1) ? p0 : p1;
        an[0] = *(const f32x4*)(pa + kn * 8);
        an[1] = *(const f32x4*)(pa + 32 * 260 + kn * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[0][s], ring[k8 * 2][s], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[0][s], ring[k8 * 2 + 1][s], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[1][s], ring[k8 * 2][s], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[1][s], ring[k8 * 2 + 1][s], acc[3], 0, 0, 0);
        }
        ring[k8 * 2] = nx[(k8 * 2) * 64 + lane];
        ring[k8 * 2 + 1] = nx[(k8 * 2 + 1) * 64 + lane];
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_barrier(0);   // keep the refill behind its MFMAs (see ffn_block_kernel)
      }
    }
  };
  STAMP();   // 5: attention output in LDS
  f32x4 lg, lb;                 // LayerNorm weight / bias of this lane's 4 columns (row-per-wave form below): in flight under the projection
  if constexpr (QX) { lg = *(const f32x4*)(a.ln_g + lane * 4); lb = *(const f32x4*)(a.ln_b + lane * 4); }
  mm(acc, a.wprojf, QX ? a.wqxf : nullptr, (1.0f / 256.0f) * __builtin_amdgcn_rcpf(osc));   // acc = resid + att . Wproj^T
  STAMP();   // 6: projection MFMAs done
  // Accumulator (mt, r) <-> tile row C + 4 hi with C = 32 mt + (r&3) + 8 (r>>2) a compile-time constant: every address below is ONE
  // per-lane base register plus a constant, and the row tests compare C with the per-lane T - 4 hi.
  if constexpr (!QX) {
    const int h4 = opaque_vgpr(kh);
    float* xbase = a.xmid + (long)bc * T * 256;                  // workgroup-uniform base + 32-bit per-lane offsets
    const unsigned o0 = (unsigned)h4 * 256u + (unsigned)ccol;
    const int T4 = T - h4;
#pragma unroll
    for (int mt = 0; mt < (R52 ? 1 : 2); ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = mt * 32 + (r & 3) + 8 * (r >> 2);
        if (c < T4) {
          const unsigned off = o0 + (unsigned)c * 256u;
          xbase[off] = acc[mt * 2][r]; xbase[off + CS] = acc[mt * 2 + 1][r];
        }
      }
    if constexpr (R52) {               // rows 32..: lane half 0 stores (both halves hold the value): 32 lanes x 4 bytes = one 128-byte segment per row and tile
#pragma unroll
      for (int gI = 0; gI < 5; ++gI)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 32 + 4 * gI + r;
          if (row < T && !hi) {
            const unsigned off = (unsigned)row * 256u + (unsigned)ccol;
            xbase[off] = g4[gI][0][r]; xbase[off + CS] = g4[gI][1][r];
          }
        }
    }
    STAMP();   // 7: xmid stores issued
  } else {
    // x' tile -> LDS (fp32 rows), then ROW-PER-WAVE LayerNorm: wave w owns rows w, w + 4, ...; a lane holds 4 columns of the row, the
    // statistics are two wave reductions (two-pass, exactly the arithmetic of the FFN block's staging LayerNorm) — no cross-wave
    // exchange, no partial-sum buffer, and x' leaves as 1 KiB-contiguous 16-byte stores.  (Round 2 normalised in accumulator layout:
    // 256 LDS reads, 3 barriers and 64 scalar stores per lane — 7 us of a 49 us workgroup alone, 15-21 us next to a partner.)
    __syncthreads();            // every wave has finished reading sAtt in mm
    {
      const int h4 = opaque_vgpr(kh);
      float* ps = sAtt + h4 * 260 + ccol;
#pragma unroll
      for (int mt = 0; mt < (R52 ? 1 : 2); ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = mt * 32 + (r & 3) + 8 * (r >> 2);
          ps[c * 260] = acc[mt * 2][r]; ps[c * 260 + CS] = acc[mt * 2 + 1][r];
        }
      if constexpr (R52) {             // rows 32 .. 51 from the 4-row groups (lane half 0); rows 52 .. 63 keep the attention phase's finite bytes
#pragma unroll
        for (int gI = 0; gI < 5; ++gI)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (!hi) { sAtt[(32 + 4 * gI + r) * 260 + ccol] = g4[gI][0][r]; sAtt[(32 + 4 * gI + r) * 260 + ccol + CS] = g4[gI][1][r]; }
      }
    }
    __syncthreads();
    float* xbase = a.xmid + (long)bc * T * 256 + lane * 4;
    // All 16 row slots of the wave are normalised unconditionally (slots >= T hold finite garbage and are never stored): one basic
    // block, so the 32 reduction chains interleave instead of running one after the other behind a branch per row.
    f32x4 x[16], y[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = *(const f32x4*)&sAtt[(4 * k + w) * 260 + lane * 4];
    float sm[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) sm[k] = wave_sum(x[k][0] + x[k][1] + x[k][2] + x[k][3]);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      y[k] = x[k] - sm[k] * (1.0f / 256.0f);
      sm[k] = y[k][0] * y[k][0] + y[k][1] * y[k][1] + y[k][2] * y[k][2] + y[k][3] * y[k][3];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) sm[k] = wave_sum(sm[k]);
#pragma unroll
    for (int k = 0; k < 16; ++k) y[k] = y[k] * __builtin_amdgcn_rsqf(sm[k] * (1.0f / 256.0f) + 1e-5f) * lg + lb;   // (argument >= 1e-5: never denormal)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int row = 4 * k + w;                                 // wave-uniform
      if (row < T) *(f32x4*)(xbase + row * 256) = x[k];
    }
    if constexpr (SPLIT) __syncthreads();   // the f16 (hi, lo) rows alias OTHER waves' fp32 rows: every row must be in registers first
#pragma unroll
    for (int k = 0; k < 16; ++k) {          // all 64 rows are written (rows >= T: finite, never stored): the A tile of the next MFMAs is fully defined
      const int row = 4 * k + w;
      if constexpr (SPLIT) {
        const h16x4 hh = __builtin_convertvector(y[k], h16x4);
        const h16x4 ll = __builtin_convertvector(y[k] - __builtin_convertvector(hh, f32x4), h16x4);
        *(h16x4*)&sAh[row * ALD16 + lane * 4] = hh;
        *(h16x4*)&sAl[row * ALD16 + lane * 4] = ll;
      } else {
        *(f32x4*)&sAtt[row * 260 + lane * 4] = y[k];
      }
    }
    STAMP();   // 7: LayerNorm + xmid stores issued
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    if constexpr (R52) {
#pragma unroll
      for (int gI = 0; gI < 5; ++gI) { g4[gI][0] = f32x4{0.f, 0.f, 0.f, 0.f}; g4[gI][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    mm(acc, a.wqxf, nullptr, 1.0f / 256.0f);
    STAMP();   // 8: cross-q projection MFMAs done
    const int h4 = opaque_vgpr(kh);
    float* qbase = a.qx + (long)bc * T * 256;
    const unsigned o0 = (unsigned)h4 * 256u + (unsigned)ccol;
    const int T4 = T - h4;
#pragma unroll
    for (int mt = 0; mt < (R52 ? 1 : 2); ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = mt * 32 + (r & 3) + 8 * (r >> 2);
        if (c < T4) {
          const unsigned off = o0 + (unsigned)c * 256u;
          qbase[off] = acc[mt * 2][r]; qbase[off + CS] = acc[mt * 2 + 1][r];
        }
      }
    if constexpr (R52) {
#pragma unroll
      for (int gI = 0; gI < 5; ++gI)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 32 + 4 * gI + r;
          if (row < T && !hi) {
            const unsigned off = (unsigned)row * 256u + (unsigned)ccol;
            qbase[off] = g4[gI][0][r]; qbase[off + CS] = g4[gI][1][r];
          }
        }
    }
  }
#ifdef VAPX_TRACE
  STAMP();   // 8 or 9: stores issued
  if (a.trace && tid == 0) { a.trace[(long)blockIdx.x * 32 + 29] = __builtin_amdgcn_s_memrealtime(); a.trace[(long)blockIdx.x * 32 + 30] = (unsigned long long)stamp_k; }
#endif
}

// ------------------------------------------------------------------------------------------------
// conv_tail_kernel: conv2 -> conv3 -> conv4 of the CPC encoder for ONE stream (both channels) per
// workgroup, intermediates in LDS.
// reference: CPCEncoder.forward encoder_components.py:98-104 (Conv1d k4 s2 p1 + ChannelNorm
// (unbiased variance) + ReLU, three times); only conv4 positions 1..P4-2 survive z[:,1:-1]
// (encoder.py:76).  At a few hundred streams these three layers are tiny GEMMs (M = 28, 14, 5 rows
// per channel) that cannot fill 256 CUs one launch at a time; fused, every CU runs one stream's
// whole tail.  Implicit GEMM: the A row of output position p is the 4 input rows 2p..2p+3 of the
// guarded channels-last slab (tap t -> row 2p+t), K = 4 x 256; weights stream fragment-major per tap.
// Requires P2 <= 32, 2*P3 <= 32, 2*ncpc <= 32 (20 Hz and 50 Hz frames).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void conv_tail_kernel(const ConvTailArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int P1 = g.P1, P2 = P1 / 2, P3 = P2 / 2, ncpc = g.ncpc;
  const int rows_in = P1 + 2, rows2 = P2 + 2, rows3 = P3 + 2;
  float* sIn = lds;                              // [P1+2][260]   one channel's conv1 output (guarded)
  float* sH2 = sIn + rows_in * LDH;              // [2][P2+2][260] conv2 outputs of both channels (guarded)
  float* red = sH2 + 2 * rows2 * LDH;            // [4][32]
  float* sH3 = sIn;                              // [2][P3+2][260] conv3 outputs, aliases sIn once conv2 is done
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5, kh = hi * 4;
  const int b = blockIdx.x;                      // stream in the batch
  const int ccol = w * 64 + l31;

  f32x4 ring[16];
  auto wbase = [&](const float* wfrag) { return (const f32x4*)wfrag + (long)w * 32 * 2 * 64; };
  auto fetch = [&](const float* wfrag) {
    const f32x4* wf = wbase(wfrag);
#pragma unroll
    for (int i = 0; i < 16; ++i) ring[i] = wf[i * 64 + lane];
  };
  // acc += A(32 rows, this lane's row at `arow`, 256 k) . Wtap^T
  auto mm = [&](f32x16(&acc)[2], const float* arow, const float* wfrag, const float* next_wfrag) {
    const float* pa = arow + kh;
    const f32x4* wf = wbase(wfrag);
    const f32x4* wnext = next_wfrag ? wbase(next_wfrag) : wf;
    f32x4 p0 = *(const f32x4*)(pa), p1;      // ping-pong A fragments (see ffn_block_kernel)
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
  // bias + ChannelNorm (unbiased variance over the 256 channels) + ReLU on a 32-row tile
  auto cn_relu = [&](f32x16(&acc)[2], const float* bias, const float* gam, const float* bet) {
    const float bi0 = bias[ccol], bi1 = bias[ccol + 32];
    float s[16], mean[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[0][r] += bi0; acc[1][r] += bi1;
      s[r] = half_sum(acc[0][r] + acc[1][r]);
    }
    __syncthreads();
    if (l31 == 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[w * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      mean[r] = (red[lr] + red[32 + lr] + red[64 + lr] + red[96 + lr]) * (1.0f / 256.0f);
      float d0 = acc[0][r] - mean[r], d1 = acc[1][r] - mean[r];
      s[r] = half_sum(d0 * d0 + d1 * d1);
    }
    __syncthreads();
    if (l31 == 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[w * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = s[r];
    __syncthreads();
    const float g0 = gam[ccol], g1 = gam[ccol + 32], b0 = bet[ccol], b1 = bet[ccol + 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      float var = (red[lr] + red[32 + lr] + red[64 + lr] + red[96 + lr]) * (1.0f / 255.0f);
      float rstd = rsqrtf(var + 1e-5f);
      acc[0][r] = relu_nanprop((acc[0][r] - mean[r]) * rstd * g0 + b0);
      acc[1][r] = relu_nanprop((acc[1][r] - mean[r]) * rstd * g1 + b1);
    }
  };
  auto zero = [](f32x16(&acc)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  };

  // zero the guard rows of sH2 (rows 0 and P2+1 of each channel slab); sH3's are written later
  for (int i = tid; i < 4 * 64; i += 256) {
    int u = i >> 7, which = (i >> 6) & 1, q = (i & 63) * 4;
    *(f32x4*)&sH2[(u * rows2 + (which ? P2 + 1 : 0)) * LDH + q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  fetch(g.w2f);
  // ---- conv2: one 32-row tile per channel ----
  for (int u = 0; u < 2; ++u) {
    __syncthreads();                         // previous channel's MFMAs are done with sIn
    const float* src = g.h1 + ((long)b * 2 + u) * rows_in * 256;
    for (int i = tid; i < rows_in * 64; i += 256) {
      int row = i >> 6, q = (i & 63) * 4;
      *(f32x4*)&sIn[row * LDH + q] = *(const f32x4*)(src + (long)row * 256 + q);
    }
    __syncthreads();
    const int p = l31 < P2 ? l31 : P2 - 1;   // pad rows recompute the last position (discarded)
    f32x16 acc[2];
    zero(acc);
#pragma unroll 1
    for (int t = 0; t < 4; ++t)
      mm(acc, &sIn[(2 * p + t) * LDH], g.w2f + (long)t * 65536,
         t < 3 ? g.w2f + (long)(t + 1) * 65536 : (u == 0 ? g.w2f : g.w3f));
    cn_relu(acc, g.b2, g.g2, g.be2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (lr < P2) {
        sH2[(u * rows2 + 1 + lr) * LDH + ccol] = acc[0][r];
        sH2[(u * rows2 + 1 + lr) * LDH + ccol + 32] = acc[1][r];
      }
    }
  }
  __syncthreads();                           // sH2 complete; sIn is free -> becomes sH3
  for (int i = tid; i < 4 * 64; i += 256) {
    int u = i >> 7, which = (i >> 6) & 1, q = (i & 63) * 4;
    *(f32x4*)&sH3[(u * rows3 + (which ? P3 + 1 : 0)) * LDH + q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- conv3: both channels in one tile (rows r = u*P3 + p) ----
  {
    int r0 = l31 < 2 * P3 ? l31 : 2 * P3 - 1;
    const int u = r0 / P3, p = r0 - u * P3;
    f32x16 acc[2];
    zero(acc);
#pragma unroll 1
    for (int t = 0; t < 4; ++t)
      mm(acc, &sH2[(u * rows2 + 2 * p + t) * LDH], g.w3f + (long)t * 65536, t < 3 ? g.w3f + (long)(t + 1) * 65536 : g.w4f);
    cn_relu(acc, g.b3, g.g3, g.be3);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (lr < 2 * P3) {
        int uu = lr / P3, pp = lr - uu * P3;
        sH3[(uu * rows3 + 1 + pp) * LDH + ccol] = acc[0][r];
        sH3[(uu * rows3 + 1 + pp) * LDH + ccol + 32] = acc[1][r];
      }
    }
  }
  __syncthreads();
  // ---- conv4: positions 1..ncpc of both channels in one tile (rows r = u*ncpc + p') ----
  {
    int r0 = l31 < 2 * ncpc ? l31 : 2 * ncpc - 1;
    const int u = r0 / ncpc, pq = r0 - u * ncpc;          // output position pq + 1 -> slab rows 2pq+2 ..
    f32x16 acc[2];
    zero(acc);
#pragma unroll 1
    for (int t = 0; t < 4; ++t)
      mm(acc, &sH3[(u * rows3 + 2 * pq + 2 + t) * LDH], g.w4f + (long)t * 65536, t < 3 ? g.w4f + (long)(t + 1) * 65536 : nullptr);
    cn_relu(acc, g.b4, g.g4, g.be4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int lr = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (lr < 2 * ncpc) {
        int uu = lr / ncpc, pp = lr - uu * ncpc;
        float* zp = g.z + (((long)b * 2 + uu) * ncpc + pp) * 256 + ccol;
        zp[0] = acc[0][r];
        zp[32] = acc[1][r];
      }
    }
  }
}

}  // namespace

bool conv_tail_supported(int P1, int ncpc) {
  const int P2 = P1 / 2, P3 = P2 / 2;
  return P2 <= 32 && 2 * P3 <= 32 && 2 * ncpc <= 32 && ((size_t)((P1 + 2) + 2 * (P2 + 2)) * LDH + 128) * 4 <= 160 * 1024;
}

hipError_t launch_conv_tail(const ConvTailArgs& a, int B, hipStream_t st) {
  if (!conv_tail_supported(a.P1, a.ncpc)) return hipErrorInvalidValue;
  static PerDeviceOnce attr_set;
  attr_set.run([] {
    (void)hipFuncSetAttribute((const void*)conv_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = ((size_t)((a.P1 + 2) + 2 * (a.P1 / 2 + 2)) * LDH + 128) * sizeof(float);
  hipLaunchKernelGGL(conv_tail_kernel, dim3(B), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_attn_block(const AttnBlockArgs& a, int B, hipStream_t st) {
  if (a.T > 64) return hipErrorInvalidValue;
  const dim3 grid(B * 2), block(256);
  if (a.split) {
    if (a.wqxf) hipLaunchKernelGGL((attn_block_kernel<true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((attn_block_kernel<true, false>), grid, block, 0, st, a);
  } else if (a.T <= 52 && a.T > 32) {   // 32 + 5 x 4 row tiles (the reference's T = 50)
    if (a.wqxf) hipLaunchKernelGGL((attn_block_kernel<false, true, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((attn_block_kernel<false, false, true>), grid, block, 0, st, a);
  } else {
    if (a.wqxf) hipLaunchKernelGGL((attn_block_kernel<false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((attn_block_kernel<false, false>), grid, block, 0, st, a);
  }
  return hipGetLastError();
}

hipError_t launch_ffn_block(const FfnArgs& a, hipStream_t st) {
  if (a.M <= 0) return hipSuccess;
  static PerDeviceOnce attr_set;
  attr_set.run([] {
    (void)hipFuncSetAttribute((const void*)ffn_block_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = (size_t)(2 * 32 * LDH) * sizeof(float);
  if (a.mode == 1 || a.mode == 2) {
    static PerDeviceOnce attr2;
    attr2.run([] {
      (void)hipFuncSetAttribute((const void*)ffn_block_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)ffn_block_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (!a.att || !a.wprojf || !a.resid || !a.xmid_out) return hipErrorInvalidValue;
    if (a.mode == 1 && a.resid_rot && a.resid_T < 32) return hipErrorInvalidValue;   // a 32-row tile spans at most two windows (the kernel's scalar slot / rotation loads)
    if (a.mode == 1) hipLaunchKernelGGL((ffn_block_kernel<1, 1>), dim3((a.M + 31) / 32), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((ffn_block_kernel<1, 2>), dim3((a.M + 31) / 32), dim3(256), lds / 2, st, a);   // one tile: three workgroups per CU
    return hipGetLastError();
  }
  hipLaunchKernelGGL(ffn_block_kernel<1>, dim3((a.M + 31) / 32), dim3(256), lds, st, a);
  return hipGetLastError();
}
