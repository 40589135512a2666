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
// TILING (round 3).  The fp32 block streams its weights L2 -> VGPR once per 32-row tile: 256 KB per contraction, hidden under 16 k-cycles
// of fp32 MFMA.  With three f16 MFMAs per product the same contraction needs 3/16 of the matrix-core time (1.4 us) but the SAME 256 KB,
// and a CU takes at most 64 B / clk from its vector memory path: 1.85 us — the round-2 kernel (32-row tiles, 2 workgroups / CU) sat
// on that wall (measured 2.2 us per contraction alone, 3.7 with a partner; profiles/r03_experiments/ffn_split_trace_*).  So the tile is
// now 64 rows x 256 columns per workgroup, EIGHT waves, wave w = the 32 output columns 32 w .. 32 w + 31 of BOTH 32-row sub-tiles: every
// weight fragment feeds six MFMAs instead of three (half the bytes per row), and two waves share each SIMD, so one wave's epilogue
// VALU / LDS work runs under the other's MFMAs.  136 KB of LDS, one workgroup per CU.
//   * A operands in LDS as f16 (hi, lo) row-major pairs ([64][264] halves each: one ds_read_b128 = the 8 k-values of a lane);
//   * weights as f16 fragment pairs [8 w][16 kc][2 hi/lo][64 lane][8 halves] (weights.frag_pack_f16x3_w8), ring 8 k-chunks ahead;
//   * every row statistic (LayerNorm mean / variance, the row maximum behind the power-of-two operand scale) is computed ROW-PER-WAVE on
//     an fp32 copy of the tile parked in the LDS bytes of the (hi, lo) pair it is about to become — no cross-wave partial sums.
// C/D layout is the fp32 one (dtype independent on gfx950); A and B use the same lane -> k mapping, so the k order inside a 16-chunk is
// immaterial.
#include "fused_blocks.h"

#include <algorithm>
#include <cstdlib>

// Variants measured in round 5 and NOT in this file (one code path in the product; tools/microbench/patches/ffn_block_f16x3_knobs.patch re-creates
// them): a persistent tile loop for mode 1 without Q|K|V chunks (next tile's rows fetched behind the last contraction: staging 9.2 -> 2.1 us, frame
// rate +-1 % on a power-limited board), the GELU as a phase of its own instead of woven into the next FFN1 contraction (32 registers less, same
// time), a weight ring of 8 k-chunks instead of 4 (as fast, 32 registers dearer).  profiles/r05_experiments/README.md section 4.

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int LD16 = 264;                  // halves per LDS row: 256 + 8 pad (528 B: conflict-free 16-byte row reads)
constexpr float kWScaleInv = 1.0f / 256.0f;   // weights are packed as 2^8 w

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE as FfnArgs::mode (0: xmid from global; 1: attention-output projection + the whole block; 2: projection + LN + wqkvf chunks only)
// MODE 2 (projection + LayerNorm + cross-query projection: two contractions per tile, no FFN) never touches the X tile: it is launched with
// ONE (hi, lo) tile of LDS (68 KB) and compiled for 4 waves per SIMD, so that TWO workgroups share a CU and one's staging / statistics / store
// phases run under the other's contractions (round 4: 22 us per tile with 5 us of MFMA in it when it ran alone on its CU).  Each workgroup
// still feeds 64 rows per weight fragment, so the pair needs no more of the CU's vector-memory path per MFMA than one workgroup does.
// TAILQ: the tile ends with the next layer's Q|K|V chunks (short windows; long windows only with VAPX_FLAG_SPLIT_QKV_IN_FFN).  Without them
// (mode 1 by default since round 5: the attention kernel projects its own Q|K|V) the normalised rows leave straight for xn_out.
template <int MODE, bool TAILQ = true>
__global__ __launch_bounds__(512, MODE == 2 ? 2 : 1) void ffn_block_f16x3_kernel(const FfnArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int BM = 64;
  _Float16* sXh = (_Float16*)lds_raw;     // LN_ffn(x) tile, hi / lo (MODE 2: absent — sH starts here)
  _Float16* sXl = sXh + BM * LD16;
  _Float16* sHh = MODE == 2 ? sXh : sXl + BM * LD16;   // attention rows / gelu chunk / raw x / LN rows, hi / lo
  _Float16* sHl = sHh + BM * LD16;
  float* sHf = (float*)sHh;               // the same bytes as ONE fp32 tile [BM][LD16] (2 x BM x LD16 halves = BM x LD16 floats)
  float* rinv = (float*)(sHl + BM * LD16);   // [BM] 1 / s_row of the rows staged with a power-of-two scale
  const int tid = threadIdx.x;
  int lane = tid & 63;
  int w = __builtin_amdgcn_readfirstlane(tid >> 6);            // 0..7
  int l31 = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.x * BM;
#ifdef VAPX_TRACE
  int stamp_k = 0;
  auto STAMP = [&]() {   // phase time stamps of wave 0 (debug build `make trace`: tools/ffn_trace.py --split)
    if (g.trace && tid == 0 && stamp_k < 28) g.trace[(long)(m0 / BM) * 32 + stamp_k] = __builtin_amdgcn_s_memtime();
    ++stamp_k;
  };
  auto TILE_BEGIN = [&]() {
    stamp_k = 0;
    STAMP();
    if (g.trace && tid == 0) g.trace[(long)(m0 / BM) * 32 + 28] = __builtin_amdgcn_s_memrealtime();
  };
  auto TILE_END = [&]() {
    if (g.trace && tid == 0) { g.trace[(long)(m0 / BM) * 32 + 29] = __builtin_amdgcn_s_memrealtime(); g.trace[(long)(m0 / BM) * 32 + 30] = (unsigned long long)stamp_k; }
  };
  auto FINE = [&]() { __builtin_amdgcn_sched_barrier(0); STAMP(); __builtin_amdgcn_sched_barrier(0); };   // (tools/ffn_trace.py --split --fine)
#else
  auto STAMP = [] {};
  auto FINE = [] {};
  auto TILE_BEGIN = [] {};
  auto TILE_END = [] {};
#endif

  // weight fragments of this wave's 32 columns: ring of RD k-chunks (hi, lo) ahead, running on into the next unit
  constexpr int RD = 4;   // k-chunks of weight fragments in flight per wave (1.5 k cycles of cover)
  f32x4 ring[2 * RD];
  auto wbase = [&](const float* wfrag) { return (const f32x4*)wfrag + (long)w * 16 * 2 * 64; };   // wave-uniform
  auto fetch = [&](const float* wfrag) {
    const f32x4* wf = wbase(wfrag);
#pragma unroll
    for (int i = 0; i < 2 * RD; ++i) ring[i] = wf[i * 64 + lane];
  };
  // layer 0 (mode 1): the residual rows live in the per-stream embedding rings.  A 64-row tile of a long window (T >= 64, which is where mode 1
  // runs) touches at most TWO (stream, channel) slabs, so their slots and rotations are four SCALAR loads (no vmcnt, no per-lane chain)
  int bc0 = 0, slot0 = 0, slot1 = 0, rot0 = 0, rot1 = 0;
  auto load_slots = [&]() {
    if (MODE == 1 && g.resid_rot) {
      bc0 = m0 / g.resid_T;
      const int nb = (g.M / g.resid_T) >> 1;                     // streams in the batch
      const int b0 = bc0 >> 1, b1 = ((bc0 + 1) >> 1) < nb ? (bc0 + 1) >> 1 : b0;
      slot0 = g.resid_ids ? uniform_load(g.resid_ids, b0) : b0;
      slot1 = g.resid_ids ? uniform_load(g.resid_ids, b1) : b1;
      rot0 = uniform_load(g.resid_rot, b0);
      rot1 = uniform_load(g.resid_rot, b1);
    }
  };
  fetch(MODE == 0 ? g.w0f : g.wprojf);   // the first weight fragments fly while the tile is staged

  // wave w stages rows w, w + 8, ..: a lane holds 4 columns of a whole row, row statistics are wave reductions
  auto split_row = [&](_Float16* dh, _Float16* dl, int row, const f32x4& y) {
    const h16x4 hh = __builtin_convertvector(y, h16x4);
    const h16x4 ll = __builtin_convertvector(y - __builtin_convertvector(hh, f32x4), h16x4);
    *(h16x4*)&dh[row * LD16 + lane * 4] = hh;
    *(h16x4*)&dl[row * LD16 + lane * 4] = ll;
  };
  f32x4 xr[BM / 8];
  auto load_rows = [&]() {
    const float* src = MODE == 0 ? g.xmid : g.att;
#pragma unroll
    for (int k = 0; k < BM / 8; ++k) {
      int m = m0 + w + 8 * k;
      m = m < g.M ? m : g.M - 1;
      xr[k] = *(const f32x4*)(src + (long)m * 256 + lane * 4);
    }
  };
  // The residual rows of the output projection (modes 1, 2) are HBM misses — the layer input was written a whole launch ago: their loads go out
  // HERE, with the tile's own rows, and land while the tile is staged (round 5: issued behind the contraction they cost 2.4 us of latency
  // plus 3.2 us of barrier skew per tile; issued just ahead of it they would stall its weight ring instead — one in-order vmcnt).
  // (Mode 2 keeps them behind its contraction: 32 more live registers would cost it the second workgroup per CU, which hides the latency anyway.)
  f32x4 rs[2][4];
  auto load_resid = [&]() {
    const int rcol = w * 32 + 4 * hi;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int lr = rt * 32 + l31;
      int m = m0 + lr;
      m = m < g.M ? m : g.M - 1;
      const float* rp;
      if (MODE == 1 && g.resid_rot) {   // layer 0: residual rows straight from the embedding ring
        const int T = g.resid_T;
        const bool second = m >= (bc0 + 1) * T;                  // (m < (bc0 + 2) T: the tile is no longer than a window)
        const int bc = bc0 + (second ? 1 : 0), i = m - bc * T;
        const long slab = (long)(second ? slot1 : slot0) * 2 + (bc & 1);
        int rr = i + (second ? rot1 : rot0);
        rr = rr >= T ? rr - T : rr;
        rp = g.resid + (slab * T + rr) * 256 + rcol;
      } else {
        rp = g.resid + (long)m * 256 + rcol;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) rs[rt][j] = *(const f32x4*)(rp + 8 * j);
    }
  };
  // every global load of the tile's own rows is issued at kernel entry (mode 1: the projection's residual rows with them)
  load_slots();
  load_rows();
  if constexpr (MODE == 1) load_resid();
  TILE_BEGIN();
  {
    if constexpr (MODE == 0) {   // A operand of FFN1 = LayerNorm(xmid; ln_ffn), normalised and split while the tile is staged
      const f32x4 lg = *(const f32x4*)(g.lnf_g + lane * 4), lb = *(const f32x4*)(g.lnf_b + lane * 4);
      float sm[BM / 8];
#pragma unroll
      for (int k = 0; k < BM / 8; ++k) sm[k] = wave_sum(xr[k][0] + xr[k][1] + xr[k][2] + xr[k][3]);
#pragma unroll
      for (int k = 0; k < BM / 8; ++k) {
        xr[k] = xr[k] - sm[k] * (1.0f / 256.0f);
        sm[k] = xr[k][0] * xr[k][0] + xr[k][1] * xr[k][1] + xr[k][2] * xr[k][2] + xr[k][3] * xr[k][3];
      }
#pragma unroll
      for (int k = 0; k < BM / 8; ++k) sm[k] = wave_sum(sm[k]);
#pragma unroll
      for (int k = 0; k < BM / 8; ++k) split_row(sXh, sXl, w + 8 * k, xr[k] * rsqrtf(sm[k] * (1.0f / 256.0f) + 1e-5f) * lg + lb);
    } else {   // raw attention rows -> sH (A operand of the output projection), each as x * s_row with |.| < 2^14; 1 / s_row kept
      float mx[BM / 8];
#pragma unroll
      for (int k = 0; k < BM / 8; ++k) mx[k] = wave_max(fmaxf(fmaxf(fabsf(xr[k][0]), fabsf(xr[k][1])), fmaxf(fabsf(xr[k][2]), fabsf(xr[k][3]))));
#pragma unroll
      for (int k = 0; k < BM / 8; ++k) {
        const float s = pow2_scale_for(mx[k]);
        if (lane == 0) rinv[w + 8 * k] = __builtin_amdgcn_rcpf(s);     // (exact: s is a power of two)
        split_row(sHh, sHl, w + 8 * k, xr[k] * s);
      }
    }
  }
  lds_barrier();
  STAMP();   // 1: tile staged

  // acc[rt] += (A[rows 32 rt ..][256 k] (LDS hi / lo) . W^T for this wave's 32 columns)^T: the WEIGHT fragment is the MFMA's A operand
  // and the activation rows its B operand (the two operand layouts are the same), so a lane ends up with ONE ROW and 4 x 4 consecutive
  // columns of it — every epilogue below moves 16 bytes per instruction and needs one row index per lane instead of sixteen
  // one k-chunk: 6 MFMAs on the A fragments fetched a step ago, the next step's A fragments from LDS, two ring refills from `nx`
  f16x8 n0h, n0l, n1h, n1l;   // A fragments ping-pong one k-chunk ahead of their use
  auto kstep_body = [&](f32x16(&acc)[2], const _Float16* pah, const _Float16* pal, const f32x4* nx, int k8, int kn) {
    const f16x8 a0h = n0h, a0l = n0l, a1h = n1h, a1l = n1l;
    n0h = *(const f16x8*)(pah + kn); n0l = *(const f16x8*)(pal + kn);
    n1h = *(const f16x8*)(pah + 32 * LD16 + kn); n1l = *(const f16x8*)(pal + 32 * LD16 + kn);
    const f16x8 bh = __builtin_bit_cast(f16x8, ring[k8 * 2]), bl = __builtin_bit_cast(f16x8, ring[k8 * 2 + 1]);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, a0h, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, a1h, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, a0l, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, a1l, acc[1], 0, 0, 0);
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, a0h, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, a1h, acc[1], 0, 0, 0);
    ring[k8 * 2] = nx[(k8 * 2) * 64 + lane];
    ring[k8 * 2 + 1] = nx[(k8 * 2 + 1) * 64 + lane];
  };
  auto kstep = [&](f32x16(&acc)[2], const _Float16* pah, const _Float16* pal, const f32x4* nx, int k8, int kn) {
    kstep_body(acc, pah, pal, nx, k8, kn);
    // one memory instruction between two MFMAs (4 LDS reads, 2 weight loads per 6 MFMAs)
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    __builtin_amdgcn_sched_barrier(0);   // keep the refills of this chunk behind its MFMAs
  };
  auto mm = [&](f32x16(&acc)[2], const _Float16* Ah, const _Float16* Al, const float* wfrag, const float* next_wfrag) {
    const _Float16* pah = Ah + l31 * LD16 + hi * 8;
    const _Float16* pal = Al + l31 * LD16 + hi * 8;
    const f32x4* wf = wbase(wfrag);
    const f32x4* wnext = next_wfrag ? wbase(next_wfrag) : wf;
    __builtin_amdgcn_s_setprio(1);
    n0h = *(const f16x8*)(pah); n0l = *(const f16x8*)(pal);
    n1h = *(const f16x8*)(pah + 32 * LD16); n1l = *(const f16x8*)(pal + 32 * LD16);
#pragma unroll 1
    for (int blk = 0; blk < 16 / RD; ++blk) {
      const f32x4* nx = blk < 16 / RD - 1 ? wf + (blk + 1) * RD * 2 * 64 : wnext;
#pragma unroll
      for (int k8 = 0; k8 < RD; ++k8) kstep(acc, pah, pal, nx, k8, ((blk * RD + k8 + 1) & 15) * 16);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // the same contraction, fully unrolled, with `side(q)` — a slice of VALU / LDS-store work that does not depend on it: one of the eight
  // 4-value groups of the PREVIOUS hidden chunk's GELU — woven into every pair of k-chunks: behind each of the 12 MFMAs come five VALU
  // instructions, one transcendental and one memory instruction (sched_group_barrier), which is what fits in an MFMA's shadow with two
  // waves on the SIMD.  (As a block BEHIND the MFMAs the slice gains nothing: the SIMD's two waves run in lockstep, so both sit in their
  // VALU blocks at the same time and the matrix pipe idles — measured: 7.4 us for the pair, exactly the sum of its parts.)
  auto mm_side = [&](f32x16(&acc)[2], const _Float16* Ah, const _Float16* Al, const float* wfrag, const float* next_wfrag, auto&& side) {
    const _Float16* pah = Ah + l31 * LD16 + hi * 8;
    const _Float16* pal = Al + l31 * LD16 + hi * 8;
    const f32x4* wf = wbase(wfrag);
    const f32x4* wnext = next_wfrag ? wbase(next_wfrag) : wf;
    n0h = *(const f16x8*)(pah); n0l = *(const f16x8*)(pal);
    n1h = *(const f16x8*)(pah + 32 * LD16); n1l = *(const f16x8*)(pal + 32 * LD16);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int blk = ks / RD, k8 = ks % RD;
      kstep_body(acc, pah, pal, blk < 16 / RD - 1 ? wf + (blk + 1) * RD * 2 * 64 : wnext, k8, ((ks + 1) & 15) * 16);
      if (ks & 1) {
        side(ks >> 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);     // GELU arithmetic
          __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);     // its exp / rcp
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // an A-fragment LDS read
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // a weight refill
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto zero = [](f32x16(&acc)[2]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  };
  // accumulator (rt, r) <-> tile row 32 rt + l31, chunk column 32 w + 8 (r >> 2) + 4 hi + (r & 3)
  const int ccol = w * 32 + 4 * hi;
  auto quad = [](const f32x16& a, int j) { return f32x4{a[4 * j], a[4 * j + 1], a[4 * j + 2], a[4 * j + 3]}; };
  // Output tiles leave through a WAVE-PRIVATE transposition in LDS (round 4).  In the accumulator layout a lane holds 16 bytes of 32
  // different rows per store instruction: 64 partial cache lines, 1.8 us to issue the eight stores of a wave (round-3 timeline), six such
  // phases per tile.  The X tile's bytes are dead whenever a tile is stored (the FFN1 contractions are over; mode 2 never uses them), and
  // [64 rows][32 columns] fp32 = 8 KB per wave x 8 waves is exactly that region: a wave writes its quads (16-byte units, unit q of row r at
  // q ^ (r & 7): conflict-free both ways — ds_write_b128 is served in groups of 8 consecutive lanes over 32 banks, ds_read_b128 in
  // the groups of 16 of MI355X_MICROARCH.md over 64), reads them back as rows — no barrier: both sides are this
  // wave's, DS operations of a wave complete in order — and stores 8 rows x 128 contiguous bytes = 16 whole lines per instruction.
  static_assert(8 * 64 * 32 * sizeof(float) <= 2 * BM * LD16 * sizeof(_Float16), "the eight wave-private transposition tiles must fit inside the X tile they borrow");
  float* sT = (float*)sXh + w * (64 * 32);
  auto store_global = [&](const f32x16(&acc)[2], float* base, int ld, int col0) {
    if constexpr (MODE == 2) {               // no X tile to borrow (and no registers to spare at 4 waves per SIMD): straight from the accumulators
      float* bu = base + (long)m0 * ld + col0 + w * 32 + 4 * hi;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int lr = rt * 32 + l31;
        if (m0 + lr < g.M) {
          float* p = bu + (unsigned)(lr * ld);
#pragma unroll
          for (int j = 0; j < 4; ++j) *(f32x4*)(p + 8 * j) = quad(acc[rt], j);
        }
      }
      return;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = rt * 32 + l31;
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)&sT[row * 32 + 4 * ((2 * j + hi) ^ (row & 7))] = quad(acc[rt], j);
    }
    // the reads below fetch OTHER lanes' quads: the DS operations of a wave complete in order, but nothing in the language says so to the
    // compiler — a wavefront-scope release / acquire pair and a wave barrier pin "all eight stores, then the loads" (advisor r04)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float* bu = base + (long)m0 * ld + col0 + w * 32 + 4 * (lane & 7);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int R = (lane >> 3) + 8 * i;
      const f32x4 v = *(const f32x4*)&sT[R * 32 + 4 * ((lane & 7) ^ (R & 7))];
      if (m0 + R < g.M) *(f32x4*)(bu + (unsigned)(R * ld)) = v;
    }
  };
  // accumulator tile -> fp32 rows in the sH bytes (row-per-wave statistics follow); callers fence sH before and after
  auto park = [&](const f32x16(&acc)[2]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int j = 0; j < 4; ++j) *(f32x4*)&sHf[(rt * 32 + l31) * LD16 + ccol + 8 * j] = quad(acc[rt], j);
  };
  // rows w, w + 8, .. of the parked tile: mean, variance (two-pass) and max |x| of each, the rows themselves in registers
  auto parked_stats = [&](f32x4 (&x)[BM / 8], float (&mean)[BM / 8], float (&var)[BM / 8], float (&amax)[BM / 8]) {
#pragma unroll
    for (int k = 0; k < BM / 8; ++k) x[k] = *(const f32x4*)&sHf[(w + 8 * k) * LD16 + lane * 4];
#pragma unroll
    for (int k = 0; k < BM / 8; ++k) {
      mean[k] = wave_sum(x[k][0] + x[k][1] + x[k][2] + x[k][3]) * (1.0f / 256.0f);
      amax[k] = wave_max(fmaxf(fmaxf(fabsf(x[k][0]), fabsf(x[k][1])), fmaxf(fabsf(x[k][2]), fabsf(x[k][3]))));
    }
#pragma unroll
    for (int k = 0; k < BM / 8; ++k) {
      const f32x4 d = x[k] - mean[k];
      var[k] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    }
#pragma unroll
    for (int k = 0; k < BM / 8; ++k) var[k] = wave_sum(var[k]) * (1.0f / 256.0f);
  };

  const int nq = TAILQ ? (g.wqkvf ? g.n_qkv_chunks : 0) : 0;
  const float* after_ffn = g.wkvxf ? g.wkvxf : (nq ? g.wqkvf : nullptr);
  // When the tile's rows go through the row-per-wave pass at the end (statistics for the next projections / LayerNorm), the block's own
  // output (x, or mode 2's xmid) leaves FROM THAT PASS: a wave holds whole rows there — 1 KB contiguous per store instruction, and no trip
  // through the transposition tile (round 5: the separate store phase was 2.2 us of a 62 us tile).
  const bool rows_follow = g.wkvxf || nq || g.xn_out;
  f32x16 out[2];
  if constexpr (MODE != 0) {
    // ---- attention output projection + residual: xmid = resid + att . Wproj^T (the separate GEMM of the long-window path) ----
    zero(out);
    mm(out, sHh, sHl, g.wprojf, MODE == 1 ? g.w0f : after_ffn);
    FINE();
    if constexpr (MODE == 2) load_resid();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float sc = rinv[rt * 32 + l31] * kWScaleInv;     // undo the row scale of the staged attention row and the weights' 2^8
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) out[rt][4 * j + e] = out[rt][4 * j + e] * sc + rs[rt][j][e];
    }
    // mode 2 hands xmid to the next block through HBM; in mode 1 its only consumer is this workgroup (the residual of the FFN), so it
    // stays in the accumulators the FFN2 products are added to (below) and never travels
    if constexpr (MODE == 2) { if (!rows_follow) store_global(out, g.xmid_out, 256, 0); }
    STAMP();   // 2: projection + residual
    if constexpr (MODE == 1) {   // A operand of FFN1 = LayerNorm(xmid; ln_ffn): parked in sH (the attention rows are consumed), normalised into sX
      lds_barrier();           // every wave is done reading the attention rows
      FINE();
      park(out);
      lds_barrier();
      FINE();
      f32x4 x[BM / 8];
      float mean[BM / 8], var[BM / 8], amax[BM / 8];
      parked_stats(x, mean, var, amax);
      FINE();
      const f32x4 lg = *(const f32x4*)(g.lnf_g + lane * 4), lb = *(const f32x4*)(g.lnf_b + lane * 4);
#pragma unroll
      for (int k = 0; k < BM / 8; ++k) split_row(sXh, sXl, w + 8 * k, (x[k] - mean[k]) * rsqrtf(var[k] + 1e-5f) * lg + lb);
      lds_barrier();
      FINE();
    }
  }
  if constexpr (MODE != 2) {
    // ---- feed-forward: x = xmid + gelu(xn W0^T) W3^T, hidden processed in 3 chunks of 256 ----
    // hid_scale: static power of two <= 1 from the weights' bound on |gelu(h)| (vapx_create), 1 for every sane checkpoint
    const float hs = g.hid_scale > 0.f ? g.hid_scale : 1.0f;
    if constexpr (MODE == 0) {
      zero(out);
    } else {   // the accumulators start from xmid, in the products' units (2^8 from the weights, hs from the hidden row: exact)
      const float up = 256.0f * hs;
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[rt][r] *= up;
    }
    // gelu + f16 split of one 4-value group (q = 4 rt + j) of a hidden chunk's accumulators -> sH.  These VALU instructions are NOT free under
    // the contraction they are woven into: a SIMD issues them and the MFMAs from the same port (tools/microbench/mfma_valu_overlap: ten
    // independent v_fma_f32 per 32-clk MFMA stretch it to 46 clk with two waves on the SIMD; the woven phase measures as the SUM of its MFMA
    // and VALU time), so the cost of a hidden chunk is its instruction count.  Round 5: ONE transcendental per element instead of two —
    //   gelu(h) = (h + |h| - |h| erfc(|h| / sqrt 2)) / 2,   erfc(z) = 2^(z q(z)),  q = degree-6 minimax fit of log2(erfc(z)) / z on [0, 4.2]
    // (tools/fit_erfc_exp2.py: max |erf error| 1.8e-7 in float32 arithmetic — A&S 7.1.26, used by the fp32 path, has 1.5e-7; beyond
    // z = 4.2 erfc < 3e-9 and the clamp holds it there).  With the operand scale 2^-8 and the result scale hs folded into k1 / k3: 13 plain
    // operations + exp2 per element instead of 18 + rcp + exp2 (written on pairs; hipcc un-packs most v_pk_fma_f32 next to MFMAs on purpose).
    const float k1 = 0.70710678118654752440f * kWScaleInv;
    const float k3 = 0.5f * kWScaleInv * hs;
    const f32x2 C6 = f32x2{1.0022112e-4f, 1.0022112e-4f}, C5 = f32x2{-4.6157415e-4f, -4.6157415e-4f}, C4 = f32x2{-2.3022329e-3f, -2.3022329e-3f},
                C3 = f32x2{2.9452506e-2f, 2.9452506e-2f}, C2 = f32x2{-1.4896366e-1f, -1.4896366e-1f}, C1 = f32x2{-9.1832864e-1f, -9.1832864e-1f},
                C0 = f32x2{-1.6279137f, -1.6279137f}, ZMAX = f32x2{4.2f, 4.2f};
    auto gelu_group = [&](const f32x16(&h)[2], int q) {
      const int rt = q >> 2, j = q & 3;
      f32x4 y;
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {   // two elements per instruction wherever a packed form exists (v_pk_mul / v_pk_fma / v_pk_add)
        const f32x2 a = f32x2{h[rt][4 * j + 2 * e2], h[rt][4 * j + 2 * e2 + 1]};
        const f32x2 aa = f32x2{fabsf(a[0]), fabsf(a[1])};
        f32x2 z = aa * k1;
        z = __builtin_elementwise_min(z, ZMAX);
        f32x2 pl = __builtin_elementwise_fma(C6, z, C5);
        pl = __builtin_elementwise_fma(pl, z, C4);
        pl = __builtin_elementwise_fma(pl, z, C3);
        pl = __builtin_elementwise_fma(pl, z, C2);
        pl = __builtin_elementwise_fma(pl, z, C1);
        pl = __builtin_elementwise_fma(pl, z, C0);
        pl = pl * z;
        const f32x2 ec = f32x2{__builtin_amdgcn_exp2f(pl[0]), __builtin_amdgcn_exp2f(pl[1])};   // erfc(|h| / sqrt 2), arguments <= 0
        const f32x2 r = __builtin_elementwise_fma(-aa, ec, a + aa) * k3;
        y[2 * e2] = r[0]; y[2 * e2 + 1] = r[1];
      }
      const h16x4 hh = __builtin_convertvector(y, h16x4);
      const h16x4 ll = __builtin_convertvector(y - __builtin_convertvector(hh, f32x4), h16x4);
      *(h16x4*)&sHh[(rt * 32 + l31) * LD16 + ccol + 8 * j] = hh;
      *(h16x4*)&sHl[(rt * 32 + l31) * LD16 + ccol + 8 * j] = ll;
    };
    // weight order: W0.0, W0.1, W3.0, W0.2, W3.1, W3.2 — the GELU of chunk c rides in the FFN1 contraction of chunk c + 1 (mm_side)
    f32x16 hacc[2][2];
    zero(hacc[0]);
    mm(hacc[0], sXh, sXl, g.w0f, g.w0f + 65536);
    STAMP();   // FFN1 chunk 0
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (c > 0) lds_barrier();   // every wave is done reading the previous chunk from sH
      if (c < 2) {
        zero(hacc[(c + 1) & 1]);
        mm_side(hacc[(c + 1) & 1], sXh, sXl, g.w0f + (long)(c + 1) * 65536, g.w3f + (long)c * 65536, [&](int q) { gelu_group(hacc[c & 1], q); });
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) gelu_group(hacc[c & 1], q);
      }
      lds_barrier();
      STAMP();   // gelu(c) -> sH  (+ FFN1 chunk c + 1)
      mm(out, sHh, sHl, g.w3f + (long)c * 65536, c == 0 ? g.w0f + 2 * 65536 : (c == 1 ? g.w3f + 2 * 65536 : after_ffn));
      STAMP();   // FFN2 chunk c
    }
    {
      const float inv = kWScaleInv * __builtin_amdgcn_rcpf(hs);
      if constexpr (MODE == 0) {
        const float* xm = g.xmid + ccol;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          int m = m0 + rt * 32 + l31;
          m = m < g.M ? m : g.M - 1;                             // rows beyond the matrix re-read its last row (their results are not stored)
          const float* rp = xm + (long)m * 256;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 rs = *(const f32x4*)(rp + 8 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[rt][4 * j + e] = out[rt][4 * j + e] * inv + rs[e];
          }
        }
      } else {   // xmid is in the sum already
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int r = 0; r < 16; ++r) out[rt][r] *= inv;
      }
    }
    if (!rows_follow) store_global(out, g.xout, 256, 0);
    STAMP();   // residual (+ x_out store)
  }
  // ---- next layer's projections: cross K,V from the RAW layer output, self Q,K,V from LayerNorm(x) ----
  if (rows_follow) {
    lds_barrier();            // every wave is done reading sH in the previous contraction
    park(out);
    lds_barrier();
    f32x4 x[BM / 8];
    float mean[BM / 8], var[BM / 8], amax[BM / 8];
    parked_stats(x, mean, var, amax);
    if (rows_follow) {
      float* dst = (MODE == 2 ? g.xmid_out : g.xout) + (long)m0 * 256 + lane * 4;
#pragma unroll
      for (int k = 0; k < BM / 8; ++k)
        if (m0 + w + 8 * k < g.M) *(f32x4*)(dst + (w + 8 * k) * 256) = x[k];
    }
    auto ln_rows = [&]() {     // LayerNorm(x; next layer's ln_self) of the wave's rows: to xn_out, and to sH when the Q|K|V chunks follow
      const f32x4 lg = *(const f32x4*)(g.ln_g + lane * 4), lb = *(const f32x4*)(g.ln_b + lane * 4);
#pragma unroll
      for (int k = 0; k < BM / 8; ++k) {
        const f32x4 y = (x[k] - mean[k]) * rsqrtf(var[k] + 1e-5f) * lg + lb;
        if (nq) split_row(sHh, sHl, w + 8 * k, y);   // (A operand of the Q|K|V chunks; without them only xn_out wants the rows)
        const int m = m0 + w + 8 * k;
        if (g.xn_out && m < g.M) *(f32x4*)(g.xn_out + (long)m * 256 + lane * 4) = y;
      }
    };
    // Without Q|K|V chunks nothing needs the normalised rows in LDS: they leave for xn_out NOW, and the row registers (x, mean, var: 48) are free
    // through the cross K|V contractions — which is what lets the next tile's rows be fetched behind the last of them.
    if constexpr (!TAILQ) ln_rows();
    STAMP();   // row statistics
    lds_barrier();            // the (hi, lo) rows written next alias OTHER waves' fp32 rows: every row is in registers first
    if (MODE != 2 && g.wkvxf) {
      // raw rows, each scaled by a power of two so that the f16 operand stays below 2^14 (exact row maximum: the wave holds the row)
#pragma unroll
      for (int k = 0; k < BM / 8; ++k) {
        const float s = pow2_scale_for(amax[k]);
        if (lane == 0) rinv[w + 8 * k] = __builtin_amdgcn_rcpf(s);
        split_row(sHh, sHl, w + 8 * k, x[k] * s);
      }
      lds_barrier();
      for (int nc = 0; nc < 2; ++nc) {
        f32x16 acc[2];
        zero(acc);
        mm(acc, sHh, sHl, g.wkvxf + (long)nc * 65536, nc == 0 ? g.wkvxf + 65536 : (nq ? g.wqkvf : nullptr));
        if (nc == 1) FINE();    // kvx1 mm done (stores follow)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const float sc = rinv[rt * 32 + l31] * kWScaleInv;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][r] *= sc;
        }
        store_global(acc, g.kvx, 512, nc * 256);
        STAMP();   // kvx mm + store
      }
      lds_barrier();          // every wave is done reading the raw rows: the tile becomes the LayerNorm rows
    }
    if constexpr (TAILQ) {
      ln_rows();
      lds_barrier();
    }
    for (int nc = 0; nc < nq; ++nc) {
      f32x16 acc[2];
      zero(acc);
      mm(acc, sHh, sHl, g.wqkvf + (long)nc * 65536, nc + 1 < nq ? g.wqkvf + (long)(nc + 1) * 65536 : nullptr);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][r] *= kWScaleInv;
      store_global(acc, g.qkv, nq * 256, nc * 256);
      STAMP();   // q / k / v mm + store
    }
  }
  TILE_END();
}

}  // namespace

hipError_t launch_ffn_block_f16x3(const FfnArgs& a, hipStream_t st) {
  if (a.M <= 0) return hipSuccess;
  static PerDeviceOnce attr_set;
  attr_set.run([] {
    (void)hipFuncSetAttribute((const void*)ffn_block_f16x3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_block_f16x3_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_block_f16x3_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)ffn_block_f16x3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  const size_t lds = (size_t)(a.mode == 2 ? 2 : 4) * 64 * LD16 * sizeof(_Float16) + 64 * sizeof(float);   // mode 2: one (hi, lo) tile, two workgroups per CU
  const dim3 grid((a.M + 63) / 64), block(512);
  if (a.mode == 1 || a.mode == 2) {
    if (!a.att || !a.wprojf || !a.resid || !a.xmid_out) return hipErrorInvalidValue;
    if (a.mode == 1 && a.resid_rot && a.resid_T < 64) return hipErrorInvalidValue;   // a tile spans at most two windows (see the kernel's scalar slot / rotation loads)
    if (a.mode == 1) {
      if (a.wqkvf && a.n_qkv_chunks > 0) hipLaunchKernelGGL((ffn_block_f16x3_kernel<1, true>), grid, block, lds, st, a);
      else hipLaunchKernelGGL((ffn_block_f16x3_kernel<1, false>), grid, block, lds, st, a);
    }
    else hipLaunchKernelGGL(ffn_block_f16x3_kernel<2>, grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL(ffn_block_f16x3_kernel<0>, grid, block, lds, st, a);
  }
  return hipGetLastError();
}
