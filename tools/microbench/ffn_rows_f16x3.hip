// ffn_rows_f16x3_kernel: the flat-row blocks of ffn_block_f16x3.hip (mode 0: FFN block; mode 1: attention output projection + the whole
// block; both with the next layer's projections behind them) rebuilt ROW-STATIONARY (round 5).  Reference ops: TransformerLayer.forward
// modules.py:266-297 (proj + residual, ln_ffnetwork, ffn_block :9-21, the next layer's ln_self_attn and query / key / value Linear
// :82-110, mha_cross key / value of the raw layer output :289-292).  Same split-precision arithmetic as ffn_block_f16x3.hip:
//   x s = hi + lo (f16),  w' = 2^8 w = hi + lo,  x.w = 2^-8 / s (x_hi.w_hi + x_lo.w_hi + x_hi.w_lo),  fp32 accumulation.
//
// WHY.  The 64-row block of rounds 3-4 kept its operand tiles in LDS and handed every intermediate (projection -> LayerNorm -> FFN1 ->
// GELU -> FFN2 -> LayerNorm -> Q|K|V) from one eight-wave phase to the next through LDS and barriers: 68 us per tile with 28 us of MFMA
// in it, each wave streaming its own weight fragments L2 -> VGPR (256 KB per contraction per 64 rows: 42 B / clk of the CU's 64).
// Here the roles are swapped:
//   * a WAVE OWNS 16 ROWS for the whole chain (v_mfma_f32_16x16x32_f16).  With the weight fragment as the MFMA's A operand (16 output
//     columns x 32 k) and the rows as its B operand, accumulator register r of lane (j, g) holds  y[row j][column 16 t + 4 g + r]  — and a
//     B operand wants, from lane (j, g), eight k-values of row j.  So the accumulators of two adjacent tiles ARE a B operand of the next
//     contraction, provided its weights are packed with the k order 16 (s >> 2) + 4 g + (s & 3) inside every 32-chunk
//     (weights.frag_pack_f16x3_rs).  Nothing a row needs ever leaves the four lanes that own it: LayerNorm statistics are a per-lane sum
//     and two cross-lane shuffles, the per-row power-of-two operand scale is a per-lane maximum, GELU + f16 split of the hidden row happen
//     in registers INSIDE the FFN2 k-loop (the 8 values of a k-step at a time).  No LDS tile, no parking, no phase barrier.
//   * the WEIGHTS are what is shared: eight waves = 128 rows per workgroup read every fragment from an LDS ring that LDS-DMA
//     (global_load_lds_dwordx4, no VGPRs) fills linearly from the layer's weight PROGRAM — all twelve 256x256 blocks of a tile in
//     consumption order.  256 KB per contraction per 128 ROWS: half the L2 -> CU stream of the 64-row block per row, none of it through
//     registers.  One s_barrier per 16 KB ring slot hands it over.
//   * ONE UNIFORM FRAGMENT STREAM.  Whatever the contraction, the unit of work is a "tile pair" (TP): 4 KB of the ring = the (hi, lo)
//     fragments of two 16-column tiles for one k-step = 4 ds_read_b128 + 6 MFMAs; a ring slot is four TPs.  The fragments run through a
//     four-deep REGISTER ring, read three TPs ahead of their use — across k-steps, contractions and tiles alike.
//   * TWO WAVES PER SIMD (256 registers each).  A first version of this kernel gave each wave 32 rows on the 32x32x16 MFMA, one wave per
//     SIMD with 512 registers: its bare MFMA stream ran at the matrix-core rate, but with nobody else on the SIMD every instruction whose
//     ISSUE stalls was paid in full — ~85 clk per LDS-DMA piece, ~370 clk per accumulator-layout store, ~170 clk per barrier: 183 us per
//     128 rows against 96 us for the stream alone (tools/microbench/ffn_rows_bench, profiles/r05_experiments).  With two waves per SIMD one
//     wave's stalls are the other's MFMA time; the price is twice the LDS fragment traffic per MFMA (16-row B operands).
//   * the hidden row is processed in six 128-wide sub-chunks (FFN1 on 8 tiles, then FFN2 over those 128 hidden values): 32 accumulator
//     registers; output tiles are stored straight from the accumulators (lane (j, g): 16 bytes, the row's four lanes 64 contiguous
//     bytes) UNDER the next contraction: the tail (next layer's cross K|V and Q|K|V) runs as 128-column halves into two alternating
//     32-register sets, one store of the previous half per k-step.
// Registers per lane: the residual-stream tile `out` (64 fp32), its (hi, lo) operand form X (64), hidden / projection accumulators
// (32 + 32), the fragment ring (64), addresses.
#include <algorithm>
#include <type_traits>

#include "fused_blocks.h"

// (experiment, not part of libvapx: FfnArgs + the layer's weight program, see tools/microbench/ffn_rows_pack.py)
struct RowsArgs : FfnArgs {
  const float* wrs;
};

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;

constexpr int kPairBytes = 16384;      // one ring slot: 8 tiles x (hi, lo) x 1 KB = a k-step of a 128-column half, or half a k-step of 256 columns
constexpr int kRingPairs = 8;          // 128 KB of LDS
constexpr int kAhead = 7;              // slots ahead of the one being read that the DMA stream runs
constexpr float kWScaleInv = 1.0f / 256.0f;

__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, f16x8& fh, f16x8& fl) {
  const h16x4 h0 = __builtin_convertvector(x0, h16x4), h1 = __builtin_convertvector(x1, h16x4);
  const h16x4 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f32x4), h16x4);
  const h16x4 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f32x4), h16x4);
  fh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
  fl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float max4abs(float mx, const f32x4& v) {
  return fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
// all-reduce over the four lanes (j, g = 0..3) that share a row
__device__ __forceinline__ float row4_sum(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}
__device__ __forceinline__ float row4_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}

// stores issued behind the six pairs before pair q of the tail (2 per pair in its first half — x_out leaves there —, 1 per pair later,
// none before the tail): they are younger than the LDS-DMA pieces pair q waits for
constexpr int tail_younger_stores(int q) {
  int n = 0;
  for (int pq = q - 6; pq < q; ++pq)
    if (pq >= 0) n += pq < 8 ? 2 : 1;
  return n;
}

#ifdef RS_TRACE
__device__ unsigned long long* rs_trace_buf;
#ifndef RS_TRACE_WAVE
#define RS_TRACE_WAVE 0
#endif
#endif

// MODE 0: xmid rows come from global (the fused short-window attention block wrote them).  MODE 1: att rows come from global, the
// output projection + residual run first (long windows).  TAIL: the next layer's cross K|V (raw rows) and Q|K|V (LN_self rows) follow;
// otherwise the block ends with x_out (+ LN_self(x_out) -> xn_out when asked for: the fused last-row block reads it).
template <int MODE, bool TAIL>
__global__ __launch_bounds__(512, 2) void ffn_rows_f16x3_kernel(const RowsArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7: rows 16 w .. 16 w + 15 of the 128-row tile
  const int j = lane & 15, gq = lane >> 4;                      // row within the wave's 16, column quad within a 16-column tile
  const unsigned ring_base = (unsigned)(uintptr_t)(lds_u8*)ring;

  // ---- the weight stream: pairs of 16 KB, program window [first, first + n) of the layer's program, wrapping (persistent tile loop) ----
  constexpr int kFirstPair = MODE == 0 ? 16 : 0;
  constexpr int kPairsPerTile = (MODE == 1 ? 16 : 0) + 96 + (TAIL ? 80 : 0);
  const char* const prog0 = (const char*)g.wrs + (size_t)kFirstPair * kPairBytes;
  const char* const prog_end = prog0 + (size_t)kPairsPerTile * kPairBytes;
  const char* dma_src = prog0;                                  // wave-uniform: next pair to fetch
  unsigned dma_slot = 0;                                        // its ring slot
  const unsigned dma_voff = (unsigned)(w * 2048 + lane * 16);   // this wave copies bytes [2048 w, 2048 w + 2048) of every pair
  auto dma_piece = [&](int e) {                                 // piece e (1 KB, e = 0 / 1) of this wave's share of the pair at dma_src
    unsigned keep;
    const char* src = dma_src + e * 1024;
    const unsigned d = ring_base + dma_slot * kPairBytes + w * 2048 + e * 1024;
#ifndef RS_EXP_NO_DMA
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(dma_voff), "s"(d), "s"(src) : "memory");
#endif
  };
  auto dma_next = [&]() {
    dma_src += kPairBytes;
    dma_src = dma_src == prog_end ? prog0 : dma_src;
    dma_slot = (dma_slot + 1) & (kRingPairs - 1);
  };
  unsigned rd_slot = 0;                                         // slot of the pair being consumed
  // top of a pair: the pair AFTER this one has landed (this wave's share: vmcnt; everyone's: the barrier), every wave is done reading the
  // pair before this one -> its slot takes the pair kAhead further on.  (Own pieces in flight after the wait: pairs +2 .. +6 = 10.)
  // vmcnt counts stores too, in order: a pair whose predecessors issued stores AFTER the pieces it waits for (they were issued six pairs
  // earlier) allows that many more operations in flight (`younger`) — otherwise the wait sits out store acknowledgements, which take
  // microseconds when every CU is writing.  Stores are unconditional for that reason (see store_x).
  auto pair_sync = [&](auto YOUNGERc) {
#ifndef RS_EXP_NO_DMA
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(10 + decltype(YOUNGERc)::value) : "memory");
#endif
#ifndef RS_EXP_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
  };

  // ---- the fragment pipeline: TP i of the stream sits in A[i & 3]; its reads were issued while TP i - 3 computed ----
  f16x8 A[4][4];                                                // [ring position][tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo]
  auto tp_read = [&](auto NBc, unsigned slot) {
    constexpr int nb = decltype(NBc)::value;
    const unsigned char* p = ring + slot * kPairBytes + nb * 4096 + lane * 16;
    A[nb][0] = *(const f16x8*)p;
    A[nb][1] = *(const f16x8*)(p + 1024);
    A[nb][2] = *(const f16x8*)(p + 2048);
    A[nb][3] = *(const f16x8*)(p + 3072);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  // one TP: position I4 within its pair; accumulators of its two tiles; B operand of its k-step
  auto tp = [&](auto I4c, auto YOUNGERc, f32x4& c0, f32x4& c1, const f16x8& bh, const f16x8& bl) {
    constexpr int I4 = decltype(I4c)::value;
    if constexpr (I4 == 0) pair_sync(YOUNGERc);
    if constexpr (I4 == 0) dma_piece(0);
    if constexpr (I4 == 2) { dma_piece(1); dma_next(); }
#ifndef RS_EXP_NO_LDS
    tp_read(std::integral_constant<int, (I4 + 3) & 3>{}, (rd_slot + (I4 + 3) / 4) & (kRingPairs - 1));
#endif
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[I4][0], bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[I4][2], bh, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[I4][0], bl, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[I4][2], bl, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[I4][1], bh, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[I4][3], bh, c1, 0, 0, 0);
    if constexpr (I4 == 3) rd_slot = (rd_slot + 1) & (kRingPairs - 1);
    // hipcc otherwise sinks every fragment read down to its first use (register pressure heuristics) and the pipeline is gone: reads,
    // MFMAs, DMA pieces and stores stay inside their TP; plain VALU / SALU / transcendental work (GELU, splits, address arithmetic) may move
    __builtin_amdgcn_sched_barrier(0x406);
  };
  // one ring slot = eight tiles of one k-step
  auto pair8y = [&](auto YOUNGERc, f32x4* acc, const f16x8& bh, const f16x8& bl) {
    tp(I0{}, YOUNGERc, acc[0], acc[1], bh, bl);
    tp(I1{}, YOUNGERc, acc[2], acc[3], bh, bl);
    tp(I2{}, YOUNGERc, acc[4], acc[5], bh, bl);
    tp(I3{}, YOUNGERc, acc[6], acc[7], bh, bl);
  };
  auto pair8 = [&](f32x4* acc, const f16x8& bh, const f16x8& bl) { pair8y(I0{}, acc, bh, bl); };

  // prologue: the first kAhead pairs fly, pairs 0 and 1 land, TPs 0..2 go to registers
#pragma unroll
  for (int q = 0; q < kAhead; ++q) {
    dma_piece(0);
    dma_piece(1);
    dma_next();
  }
  pair_sync(I0{});
  tp_read(I0{}, 0);
  tp_read(I1{}, 0);
  tp_read(I2{}, 0);
#ifdef RS_EXP_NO_LDS
  tp_read(I3{}, 0);
#endif

  f16x8 Xh[8], Xl[8];          // the rows' operand form: k-chunk kc (32 columns) = accumulator tiles 2 kc, 2 kc + 1
  f32x4 out[16];               // the rows' residual-stream tile (fp32): tile t, register r <-> column 16 t + 4 gq + r
  auto contract16 = [&](f32x4 (&acc)[16]) {                     // acc += rows(X) . W^T, 256 columns: 8 k-steps x 2 pairs
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
      pair8(&acc[0], Xh[kc], Xl[kc]);
      pair8(&acc[8], Xh[kc], Xl[kc]);
    }
  };
  auto zero8 = [](f32x4 (&a)[8]) {
#pragma unroll
    for (int t = 0; t < 8; ++t) a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // row statistics of a tile held in accumulator layout: a lane has 64 of its row's 256 values
  auto row_stats = [&](const f32x4 (&a)[16], float& mean, float& rstd, float& amax) {
    float s = 0.f, mx = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) { s += (a[t][0] + a[t][1]) + (a[t][2] + a[t][3]); mx = max4abs(mx, a[t]); }
    amax = row4_max(mx);
    mean = row4_sum(s) * (1.0f / 256.0f);
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const f32x4 d = a[t] - mean;
      v += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    rstd = rsqrtf(row4_sum(v) * (1.0f / 256.0f) + 1e-5f);
  };
  // X[kc] = LayerNorm of the 8 values (a0 = tile 2 kc, a1 = tile 2 kc + 1) (gamma / beta: 16 bytes per tile straight from global / L1)
  auto ln_chunk = [&](int kc, const f32x4& a0, const f32x4& a1, float mean, float rstd, const float* gm, const float* bt) {
    const int c0 = 32 * kc + 4 * gq;
    const f32x4 g0 = *(const f32x4*)(gm + c0), g1 = *(const f32x4*)(gm + c0 + 16);
    const f32x4 b0 = *(const f32x4*)(bt + c0), b1 = *(const f32x4*)(bt + c0 + 16);
    split8((a0 - mean) * rstd * g0 + b0, (a1 - mean) * rstd * g1 + b1, Xh[kc], Xl[kc]);
  };

#ifdef RS_EXP_SKEW   // timing experiment: workgroups start their tile loops RS_EXP_SKEW x 3.4 us x (blockIdx & 3) apart (write phases of the CUs spread)
  for (int i = 0; i < (int)(blockIdx.x & 3) * RS_EXP_SKEW; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  const int n_tiles = (g.M + 127) >> 7;
#ifdef RS_TRACE   // timeline build of tools/microbench/ffn_rows_bench: s_memtime stamps of wave RS_TRACE_WAVE at the phase boundaries of every tile
  int stamp_k = 0, stamp_tile = 0;
  auto STAMP = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if (w == RS_TRACE_WAVE && lane == 0 && stamp_k < 30) rs_trace_buf[((long)blockIdx.x * 64 + stamp_tile) * 32 + stamp_k] = __builtin_amdgcn_s_memtime();
    ++stamp_k;
    __builtin_amdgcn_sched_barrier(0);
  };
#else
  auto STAMP = [] {};
#endif
#pragma unroll 1
  for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
#ifdef RS_TRACE
    stamp_k = 0;
    stamp_tile = (tile - (int)blockIdx.x) / (int)gridDim.x;
    if (stamp_tile >= 64) stamp_tile = 63;
#endif
    STAMP();   // 0: tile start
#ifdef RS_TRACE
    if (w == RS_TRACE_WAVE && lane == 0) rs_trace_buf[((long)blockIdx.x * 64 + stamp_tile) * 32 + 30] = __builtin_amdgcn_s_memrealtime();
#endif
    const int mrow = tile * 128 + w * 16 + j;                   // this lane quad's row
    const bool live = mrow < g.M;
    const long mc = live ? mrow : g.M - 1;                      // rows beyond the matrix re-read its last row (never stored)

    if constexpr (MODE == 1) {
      // ---- attention rows -> operand form, each row scaled by a power of two so that |x s| < 2^14 (raw rows: the input decides) ----
      const float* ap = g.att + mc * 256 + 4 * gq;
      f32x4 raw[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) raw[t] = *(const f32x4*)(ap + 16 * t);
      float mx = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) mx = max4abs(mx, raw[t]);
      const float s = pow2_scale_for(row4_max(mx));
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) split8(raw[2 * kc] * s, raw[2 * kc + 1] * s, Xh[kc], Xl[kc]);
#pragma unroll
      for (int t = 0; t < 16; ++t) out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      STAMP();   // 1: attention rows in operand form
      contract16(out);
      STAMP();   // 2: projection
      // xmid = resid + att . Wproj^T
      const float* rp;
      if (g.resid_rot) {   // layer 0 of a long window: residual rows straight from the embedding ring
        const int T = g.resid_T;
        const int bc = (int)mc / T, i = (int)mc - bc * T, b = bc >> 1;
        const long slab = (long)(g.resid_ids ? g.resid_ids[b] : b) * 2 + (bc & 1);
        int rr = i + g.resid_rot[b];
        rr = rr >= T ? rr - T : rr;
        rp = g.resid + (slab * T + rr) * 256 + 4 * gq;
      } else {
        rp = g.resid + mc * 256 + 4 * gq;
      }
      const float sc = __builtin_amdgcn_rcpf(s) * kWScaleInv;    // (exact: powers of two)
#pragma unroll
      for (int t = 0; t < 16; ++t) out[t] = out[t] * sc + *(const f32x4*)(rp + 16 * t);
    } else {
      const float* xp = g.xmid + mc * 256 + 4 * gq;
#pragma unroll
      for (int t = 0; t < 16; ++t) out[t] = *(const f32x4*)(xp + 16 * t);
    }
    // ---- feed-forward: x = xmid + gelu(LN_ffn(xmid) W0^T) W3^T ----
    {
      float mean, rstd, amax;
      row_stats(out, mean, rstd, amax);
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) ln_chunk(kc, out[2 * kc], out[2 * kc + 1], mean, rstd, g.lnf_g, g.lnf_b);
    }
    STAMP();   // 3 (mode 0: 1): residual + LN_ffn -> X
    const float hs = g.hid_scale > 0.f ? g.hid_scale : 1.0f;   // static power of two <= 1 from the weights' bound on |gelu(h)| (vapx_create)
    {
      const float up = 256.0f * hs;                            // the accumulators start from xmid, in the products' units (exact)
#pragma unroll
      for (int t = 0; t < 16; ++t) out[t] *= up;
    }
    // gelu (A&S 7.1.26 erf, the constants of ffn_block_f16x3.hip: operand scale 2^-8, result scale hs, 1 / sqrt 2 and log2 e folded)
    constexpr double kSqrtLog2e = 1.2011224087864498;
    const float k1 = (float)(0.70710678118654752440 * kSqrtLog2e) * kWScaleInv;
    constexpr float k2 = (float)(0.3275911 / kSqrtLog2e);
    const float k3 = 0.5f * kWScaleInv * hs;
    auto gelu1 = [&](float a) {
#ifdef RS_EXP_NO_GELU
      return a * k3;
#endif
      const float z = fabsf(a) * k1;
      const float t = __builtin_amdgcn_rcpf(fmaf(k2, z, 1.0f));
      float pl = fmaf(1.061405429f, t, -1.453152027f);
      pl = fmaf(pl, t, 1.421413741f);
      pl = fmaf(pl, t, -0.284496736f);
      pl = fmaf(pl, t, 0.254829592f);
      const float er = fmaf(-(pl * t), __builtin_amdgcn_exp2f(-z * z), 1.0f);
      const float hc = a * k3;
      return fmaf(copysignf(er, a), hc, hc);
    };
    f32x4 hacc[8];
#pragma unroll 1
    for (int sc = 0; sc < 6; ++sc) {                            // six 128-wide sub-chunks of the hidden row
      zero8(hacc);
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) pair8(hacc, Xh[kc], Xl[kc]);   // FFN1: hidden columns 128 sc .. + 127
      if (sc == 0) STAMP();   // 4: FFN1 of sub-chunk 0
#pragma unroll
      for (int q = 0; q < 4; ++q) {                             // FFN2 over them: the hidden values of a k-step become its B operand in place
        f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v0[e] = gelu1(hacc[2 * q][e]); v1[e] = gelu1(hacc[2 * q + 1][e]); }
        f16x8 bh, bl;
        split8(v0, v1, bh, bl);
        pair8(&out[0], bh, bl);
        pair8(&out[8], bh, bl);
      }
      if (sc == 0) STAMP();   // 5: FFN2 of sub-chunk 0
    }
    STAMP();   // 6: FFN done
    {
      const float inv = kWScaleInv * __builtin_amdgcn_rcpf(hs);
#pragma unroll
      for (int t = 0; t < 16; ++t) out[t] *= inv;
    }
    // stores in accumulator layout: 16 bytes per tile per lane, the four lanes of a row 64 contiguous bytes
    // (UNCONDITIONAL: the lanes of rows beyond the matrix computed on its last row and hold that row's values bit for bit, so they store
    // to row M - 1 as well — the same bytes — and every wave issues the same number of stores, which pair_sync's counts rely on)
    float* const xo = g.xout + mc * 256 + 4 * gq;
    auto store_x = [&](int t) {                                 // t = 0..15
#ifdef RS_EXP_NO_STORE
      if (g.M < 0)
#endif
        *(f32x4*)(xo + 16 * t) = out[t];
    };
    float mean, rstd, amax;
    if (TAIL || g.xn_out) row_stats(out, mean, rstd, amax);
    if constexpr (!TAIL) {
#pragma unroll
      for (int t = 0; t < 16; ++t) store_x(t);
      if (g.xn_out) {
        float* const xn = g.xn_out + mc * 256 + 4 * gq;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int c0 = 16 * t + 4 * gq;
          const f32x4 y = (out[t] - mean) * rstd * *(const f32x4*)(g.ln_g + c0) + *(const f32x4*)(g.ln_b + c0);
#ifdef RS_EXP_NO_STORE
          if (g.M < 0)
#endif
            *(f32x4*)(xn + 16 * t) = y;
        }
      }
    } else {
      // ---- next layer's projections: cross K, V from the RAW rows (each scaled by a power of two), self Q, K, V from LN_self(x).
      //      Ten 128-column halves into two alternating 32-register sets; the previous half's stores ride in the current one's k-loop.
      //      x_out leaves under the first half ----
      const float s = pow2_scale_for(amax);
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) split8(out[2 * kc] * s, out[2 * kc + 1] * s, Xh[kc], Xl[kc]);
      const float osc = __builtin_amdgcn_rcpf(s) * kWScaleInv;
      STAMP();   // 7: row statistics + raw operand
      f32x4 accB[8];
      f32x4 (&accA)[8] = hacc;
      auto scale8 = [](f32x4 (&a)[8], float sc) {
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] *= sc;
      };
      // half hf (0..9): 0-3 = cross K lo/hi columns, cross V lo/hi -> kvx [M][512]; 4-9 = Q, K, V -> qkv [M][768]
      auto half_dst = [&](int hf) -> float* {
        return hf < 4 ? g.kvx + mc * 512 + hf * 128 + 4 * gq : g.qkv + mc * 768 + (hf - 4) * 128 + 4 * gq;
      };
      auto store_half = [&](const f32x4 (&a)[8], float* dst, int t) {   // t = 0..7
#ifdef RS_EXP_NO_STORE
        if (g.M < 0)
#endif
          *(f32x4*)(dst + 16 * t) = a[t];
      };
      // pair q = 8 hf + kc of the tail: the stores issued behind the previous six pairs are younger than the pieces it waits for
      // (2 per pair in half 0, 1 per pair later, none before the tail)
      auto half = [&](auto HFc, f32x4 (&acc)[8], auto&& side) {
        constexpr int hf = decltype(HFc)::value;
        zero8(acc);
        auto step = [&](auto KCc) {
          constexpr int kc = decltype(KCc)::value, q = 8 * hf + kc;
          constexpr int younger = tail_younger_stores(q);
          pair8y(std::integral_constant<int, younger>{}, acc, Xh[kc], Xl[kc]);
          side(kc);
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
      };
      half(std::integral_constant<int, 0>{}, accA, [&](int kc) { store_x(2 * kc); store_x(2 * kc + 1); });             // cross K, columns 0..127 (x_out leaves under it)
      scale8(accA, osc);
      half(std::integral_constant<int, 1>{}, accB, [&](int kc) { store_half(accA, half_dst(0), kc); });
      scale8(accB, osc);
      half(std::integral_constant<int, 2>{}, accA, [&](int kc) { store_half(accB, half_dst(1), kc); });
      scale8(accA, osc);
      half(std::integral_constant<int, 3>{}, accB, [&](int kc) { store_half(accA, half_dst(2), kc); });
      scale8(accB, osc);
      STAMP();   // 8: four cross K|V halves
      // X = LN_self(x): the rows come back from global (this lane's own x_out stores, L1 / L2) so that `out` is dead from the first half
      // on — 64 registers the two accumulator sets need
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) ln_chunk(kc, *(const f32x4*)(xo + 32 * kc), *(const f32x4*)(xo + 32 * kc + 16), mean, rstd, g.ln_g, g.ln_b);
      STAMP();   // 9: LN_self -> X
      half(std::integral_constant<int, 4>{}, accA, [&](int kc) { store_half(accB, half_dst(3), kc); });
      scale8(accA, kWScaleInv);
      half(std::integral_constant<int, 5>{}, accB, [&](int kc) { store_half(accA, half_dst(4), kc); });
      scale8(accB, kWScaleInv);
      half(std::integral_constant<int, 6>{}, accA, [&](int kc) { store_half(accB, half_dst(5), kc); });
      scale8(accA, kWScaleInv);
      half(std::integral_constant<int, 7>{}, accB, [&](int kc) { store_half(accA, half_dst(6), kc); });
      scale8(accB, kWScaleInv);
      half(std::integral_constant<int, 8>{}, accA, [&](int kc) { store_half(accB, half_dst(7), kc); });
      scale8(accA, kWScaleInv);
      half(std::integral_constant<int, 9>{}, accB, [&](int kc) { store_half(accA, half_dst(8), kc); });
      scale8(accB, kWScaleInv);
#pragma unroll
      for (int t = 0; t < 8; ++t) store_half(accB, half_dst(9), t);
    }
    STAMP();   // last: tile done
#ifdef RS_TRACE
    if (w == RS_TRACE_WAVE && lane == 0) rs_trace_buf[((long)blockIdx.x * 64 + stamp_tile) * 32 + 31] = __builtin_amdgcn_s_memrealtime();
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA piece may still be in flight when the workgroup's LDS is released
}

}  // namespace

bool ffn_rows_f16x3_supported(const RowsArgs& a) {
  if (!a.wrs || (a.mode != 0 && a.mode != 1)) return false;
  const bool tail = a.wkvxf && a.wqkvf && a.n_qkv_chunks == 3;
  const bool none = !a.wkvxf && !a.wqkvf;
  return tail || none;
}

hipError_t launch_ffn_rows_f16x3(const RowsArgs& a, hipStream_t st) {
  if (a.M <= 0) return hipSuccess;
  if (!ffn_rows_f16x3_supported(a)) return hipErrorInvalidValue;
  if (a.mode == 1 && (!a.att || !a.resid)) return hipErrorInvalidValue;
  static PerDeviceOnce attr_set;
  attr_set.run([] {
    (void)hipFuncSetAttribute((const void*)ffn_rows_f16x3_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_rows_f16x3_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_rows_f16x3_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_rows_f16x3_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = (size_t)kRingPairs * kPairBytes;
  const int n_tiles = (a.M + 127) / 128;
  const dim3 grid(std::min(n_tiles, device_cu_count())), block(512);   // persistent: one workgroup per CU walks over its tiles
  const bool tail = a.wkvxf != nullptr;
  if (a.mode == 1) {
    if (tail) hipLaunchKernelGGL((ffn_rows_f16x3_kernel<1, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((ffn_rows_f16x3_kernel<1, false>), grid, block, lds, st, a);
  } else {
    if (tail) hipLaunchKernelGGL((ffn_rows_f16x3_kernel<0, true>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((ffn_rows_f16x3_kernel<0, false>), grid, block, lds, st, a);
  }
  return hipGetLastError();
}
