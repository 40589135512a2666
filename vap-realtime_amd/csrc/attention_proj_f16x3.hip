// attention_proj_f16x3_kernel (round 5): the long-window SELF-attention of the split-precision path with the query / key / value projections
// INSIDE the kernel — reference MultiHeadAttention.forward modules.py:82-110 computes k = key(K), q = query(Q), v = value(V) right where it uses
// them, too.  Until round 4 the flat-row FFN block of the previous layer produced Q|K|V for all rows (three contractions and three store phases
// of its tile), wrote 3 x 2.1 GB per C3 layer to HBM and attention_long_f16x3_kernel read them back: 12.6 of the 42 GB a stereo layer moves,
// and store phases whose acknowledgements the FFN block's next weight loads have to sit out (vmcnt is one in-order counter:
// profiles/r05_experiments/README.md).  Here a workgroup item is still one (stream, channel, head):
//   PROLOGUE  K_h, V_h (all rows of the window) and Q_h = LN_self(x) . W{q,k,v}_h^T as 3-term split products (v_mfma_f32_32x32x16_f16).
//     * eight waves, wave w owns the 32 rows of its query tile: they are its MFMA B operand for Q and K (weights = A: a lane ends up with ONE
//       row and 16 features of it — Q stays in registers in exactly the order the score MFMAs want, K goes to LDS in 8-byte pieces) and its A
//       operand for V (weights = B: a lane ends up with one FEATURE and 16 keys — the transposed V^T the P.V MFMAs read).  The row fragments
//       come straight from global / L2 (LN_self(x) rows written by the previous layer's FFN block: |x| <= 16 |gamma| + |beta|, no scaling
//       needed), 32 bytes per lane per k-step, one slot ahead.
//     * the head's 192 weight rows are shared by all eight waves: an LDS ring of 4 x 24 KB (two k-steps per slot, in the bytes the K / V
//       operands take afterwards) filled by LDS-DMA from the layer's per-head weight stream (weights.frag_pack_f16x3_qkv_heads), one
//       s_barrier per slot.  The vmcnt waits count exactly the operations issued behind the pieces they wait for (the row loads are issued
//       BEFORE the pieces of their slot so that hipcc's own waits for them never include a piece that has just left).
//   ATTENTION  as attention_long_f16x3_kernel (S^T = K.Q^T, online softmax, O^T = V^T.P^T, all as split products), K / V^T from LDS.
//   The output stores of an item are issued AFTER the next item's prologue: a wait for a load that was issued behind a store cannot complete
//   before the store is acknowledged, and the prologue is full of such waits.
// Items are mapped so that the four heads of a (stream, channel) run on one XCD at the same time: its 250 x 256 LN rows are read from HBM once.
#include <algorithm>
#include <type_traits>

#include "vap_kernels.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_u8;
constexpr int LDK = 72;             // halves per LDS row of K: 64 features + 8 pad (144 B: conflict-free 16-byte reads down a column)
constexpr int LDV = 264;            // halves per LDS row of V^T: 256 key slots + 8 pad (528 B)
constexpr float kPScale = 4096.0f;  // P rides as 2^12 P
constexpr int kSlotBytes = 24576;   // two k-steps of the head's weight stream: 2 x 6 tiles x (hi, lo) x 1 KB
constexpr float kWScaleInv = 1.0f / 256.0f;

__device__ __forceinline__ void split4(const f32x4& y, h16x4& hh, h16x4& ll) {
  hh = __builtin_convertvector(y, h16x4);
  ll = __builtin_convertvector(y - __builtin_convertvector(hh, f32x4), h16x4);
}
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, float s, f16x8& fh, f16x8& fl) {
  h16x4 h0, h1, l0, l1;
  split4(x0 * s, h0, l0);
  split4(x1 * s, h1, l1);
  fh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
  fl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ float max16abs(float mx, const f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(v[r]));
  return mx;
}
__device__ __forceinline__ f32x4 quad(const f32x16& a, int q) { return f32x4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]}; }
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#ifdef VAPX_TRACE   // debug build only (tools/microbench/attn_proj_bench sets the two symbols): s_memtime stamps of wave ap_trace_wave at the phase boundaries of every item
__device__ unsigned long long* ap_trace_buf;
__device__ int ap_trace_wave;
#endif

__global__ __launch_bounds__(512, 1) void attention_proj_f16x3_kernel(AttnProjArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* Kh = (_Float16*)lds_raw;          // [256][LDK] K hi   (the weight ring lives in the first 96 KB of these bytes during the prologue)
  _Float16* Kl = Kh + 256 * LDK;              // [256][LDK] K lo
  _Float16* Vh = Kl + 256 * LDK;              // [64][LDV] V^T hi
  _Float16* Vl = Vh + 64 * LDV;               // [64][LDV] V^T lo
  float* sred = (float*)(Vl + 64 * LDV);      // [16] max |K|, max |V| per wave
  const unsigned ring_base = (unsigned)(uintptr_t)(lds_u8*)lds_raw;
  const int T = a.T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
  const int it = w < 4 ? w : 11 - w;                          // this wave's 32-row tile: the two waves of a SIMD (w, w + 4) own 9 key tiles together
  // item mapping: blocks x, x + 8, x + 16, x + 24 (one XCD) take the four heads of one (stream, channel)
  const int G = (int)gridDim.x;
  int h, bc, bc_step;
  if ((G & 31) == 0) {
    const int x = (int)blockIdx.x & 7, y = (int)blockIdx.x >> 3;
    h = y & 3;
    bc = (y >> 2) * 8 + x;
    bc_step = G >> 2;
  } else {                                                    // small grids: plain round-robin over (stream, channel, head)
    h = (int)blockIdx.x & 3;
    bc = (int)blockIdx.x >> 2;
    bc_step = (G + 3) >> 2;
    if (G & 3) { h = 0; bc = (int)blockIdx.x; bc_step = G; }  // (grid not a multiple of 4: one block walks over the four heads itself, see below)
  }
  const bool walk_heads = (G & 31) != 0 && (G & 3) != 0;
  const int n_bc = a.n_items >> 2;
  const unsigned dma_voff = (unsigned)(w * 3072 + lane * 16);  // this wave copies bytes [3072 w, 3072 w + 3072) of every slot

  // deferred output of the previous item
  f32x16 o0, o1;
  float o_scl = 0.f;
  long o_row = -1;                                             // < 0: nothing pending
  int o_h = 0;
  auto store_pending = [&]() {
    if (o_row < 0) return;
    float* op = a.out + o_row * 256 + o_h * 64;
    const int hi = lane >> 5;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      *(f32x4*)(op + rr * 8 + hi * 4) = quad(o0, rr) * o_scl;
      *(f32x4*)(op + 32 + rr * 8 + hi * 4) = quad(o1, rr) * o_scl;
    }
  };

#ifdef VAPX_TRACE
  int stamp_k = 0, stamp_item = 0;
  auto STAMP = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if (ap_trace_buf && w == ap_trace_wave && lane == 0 && stamp_k < 16 && stamp_item < 128) ap_trace_buf[((long)blockIdx.x * 128 + stamp_item) * 16 + stamp_k] = __builtin_amdgcn_s_memtime();
    ++stamp_k;
    __builtin_amdgcn_sched_barrier(0);
  };
#else
  auto STAMP = [] {};
#endif
#pragma unroll 1
  for (int hw = 0; hw < (walk_heads ? 4 : 1); ++hw) {
  if (walk_heads) h = hw;
#pragma unroll 1
  for (int cur = bc; cur < n_bc; cur += bc_step) {
    const int b = cur >> 1;
    const int n = uniform_load(a.bn, b);
    const int nt_valid = (n + 31) >> 5;
    const bool act = it < nt_valid;
    const float slope = exp2f(-2.0f * (float)(h + 1));  // [1/4, 1/16, 1/64, 1/256]
    const int l31 = opaque_vgpr(lane & 31), hi = opaque_vgpr(lane >> 5);
    const float hi4f = (float)(4 * hi);
    const int i = it * 32 + l31;                                 // this lane pair's row of the window
#ifdef VAPX_TRACE
    stamp_k = 0;
#endif
    STAMP();           // 0: item start
    lds_barrier();     // A: every wave is done with the previous item's K / V^T in LDS: the bytes become the weight ring
    STAMP();           // 1: barrier A passed

    // ================= PROLOGUE: Q_h (registers), K_h, V_h^T (LDS) for this head =================
    f32x16 accQ[2], accK[2], accV[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accQ[t][r] = 0.f; accK[t][r] = 0.f; accV[t][r] = 0.f; }
    {
      const int ic = i < n ? i : (n > 0 ? n - 1 : 0);           // rows beyond the window: the last valid row (finite; masked / zeroed below); an empty window reads row 0 of its own slab
      const float* xp = a.xn + ((long)cur * T + ic) * 256 + hi * 8;
      const char* wsrc = (const char*)a.wqkvp + (size_t)h * (16 * 12288);
      auto dma_slot = [&](int s) {                               // this wave's three 1 KB pieces of slot s -> ring position s & 3
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          unsigned keep;
          const char* src = wsrc + (size_t)s * kSlotBytes + e * 1024;
          const unsigned d = ring_base + (unsigned)(s & 3) * kSlotBytes + w * 3072 + e * 1024;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(dma_voff), "s"(d), "s"(src) : "memory");
        }
      };
      f32x4 rowbuf[2][2][2];                                     // [slot parity][k-step of the slot][first / second 16 bytes]
      f16x8 fr[2][4];                                            // fragment ping-pong: [buffer][tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo]
      auto load_rows = [&](int s) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          rowbuf[s & 1][kk][0] = *(const f32x4*)(xp + (2 * s + kk) * 16);
          rowbuf[s & 1][kk][1] = *(const f32x4*)(xp + (2 * s + kk) * 16 + 4);
        }
      };
      // issue order: L0 P0 P1 | slot s: top: everything issued so far has landed (vmcnt 0: this wave's pieces of slot s + 1 are the youngest
      // operations), barrier, L(s+1), P(s+2).  The weights are L2-resident (every CU streams the same 768 KB): one slot of lead is ample.
      // (Measured, tools/microbench/attn_proj_bench: an LDS-DMA piece holds its wave for ~400 clk at issue whether the eight waves issue
      // theirs in one burst or one wave after the other, waited for or not — 4.5 of the 13.4 us of slots 1-7; the same bytes through registers
      // (global_load_dwordx4 -> ds_write_b128) were slower still: every wait for a weight load sits behind the older ROW loads, which miss to
      // HBM.  Hence the L2 prefetch of the next item's rows during the attention phase, below.)
      load_rows(0);
      asm volatile("" ::: "memory");
      dma_slot(0); dma_slot(1);
      auto slot = [&](auto Sc) {
        constexpr int s = decltype(Sc)::value;
        // the pieces of slot s + 1 have landed (this wave's: vmcnt(0); everyone's: the barrier) and every wave is done reading slot s - 1
        if constexpr (s < 7) wait_vm<0>();
        if constexpr (s < 7) __builtin_amdgcn_s_barrier();
        if constexpr (s + 1 < 8) load_rows(s + 1);
        asm volatile("" ::: "memory");
        if constexpr (s + 2 < 8) dma_slot(s + 2);
        if constexpr (s == 0) {                                  // prime the pipeline: Q fragments of k-step 0
          const unsigned char* p0 = lds_raw + lane * 16;
          fr[0][0] = *(const f16x8*)p0; fr[0][1] = *(const f16x8*)(p0 + 1024); fr[0][2] = *(const f16x8*)(p0 + 2048); fr[0][3] = *(const f16x8*)(p0 + 3072);
        }
        // three "tile pairs" per k-step (Q, K, V: four fragments and six MFMAs each); the fragments of the NEXT pair are read while the current
        // one computes (ping-pong `fr`), across k-steps and slots alike: slot s + 1 is complete since this slot's top.  hipcc would sink the
        // reads down to their first use: the sched_barrier keeps reads and MFMAs inside their pair.
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          f16x8 xh, xl;
          split8(rowbuf[s & 1][kk][0], rowbuf[s & 1][kk][1], 1.0f, xh, xl);
          const unsigned char* fp = lds_raw + (s & 3) * kSlotBytes + kk * 12288 + lane * 16;
          const unsigned char* fn = kk == 0 ? fp + 12288 : lds_raw + ((s + 1) & 3) * kSlotBytes + lane * 16;   // the next k-step's fragments
          auto pair_read = [&](f16x8 (&f)[4], const unsigned char* p) {
            f[0] = *(const f16x8*)p; f[1] = *(const f16x8*)(p + 1024); f[2] = *(const f16x8*)(p + 2048); f[3] = *(const f16x8*)(p + 3072);
          };
          // Q (weights = A, rows = B) from fr[0]; K's fragments go to fr[1]
          pair_read(fr[1], fp + 4096);
          accQ[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0][0], xh, accQ[0], 0, 0, 0);
          accQ[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0][2], xh, accQ[1], 0, 0, 0);
          accQ[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0][1], xh, accQ[0], 0, 0, 0);
          accQ[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0][3], xh, accQ[1], 0, 0, 0);
          accQ[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0][0], xl, accQ[0], 0, 0, 0);
          accQ[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[0][2], xl, accQ[1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0x406);
          // K (weights = A, rows = B) from fr[1]; V's fragments go to fr[0]
          pair_read(fr[0], fp + 8192);
          accK[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[1][0], xh, accK[0], 0, 0, 0);
          accK[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[1][2], xh, accK[1], 0, 0, 0);
          accK[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[1][1], xh, accK[0], 0, 0, 0);
          accK[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[1][3], xh, accK[1], 0, 0, 0);
          accK[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[1][0], xl, accK[0], 0, 0, 0);
          accK[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[1][2], xl, accK[1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0x406);
          // V^T (rows = A, weights = B) from fr[0]; the next k-step's Q fragments go to fr[1] ... and are moved to fr[0] for the next round
          if (!(s == 7 && kk == 1)) pair_read(fr[1], fn);
          accV[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, fr[0][0], accV[0], 0, 0, 0);
          accV[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, fr[0][2], accV[1], 0, 0, 0);
          accV[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, fr[0][0], accV[0], 0, 0, 0);
          accV[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, fr[0][2], accV[1], 0, 0, 0);
          accV[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, fr[0][1], accV[0], 0, 0, 0);
          accV[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, fr[0][3], accV[1], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0x406);
#pragma unroll
          for (int f = 0; f < 4; ++f) fr[0][f] = fr[1][f];
        }
      };
      slot(std::integral_constant<int, 0>{});
      STAMP();         // 2: slot 0 done (first rows + pieces landed)
      slot(std::integral_constant<int, 1>{}); slot(std::integral_constant<int, 2>{});
      slot(std::integral_constant<int, 3>{}); slot(std::integral_constant<int, 4>{}); slot(std::integral_constant<int, 5>{});
      slot(std::integral_constant<int, 6>{}); slot(std::integral_constant<int, 7>{});
    }
    STAMP();           // 3: projection k-loop done
    // maxima of this wave's K rows / V^T keys -> per-item power-of-two operand scales (the weights' 2^8 is still in the accumulators)
    {
      float mk = max16abs(max16abs(0.f, accK[0]), accK[1]), mv = max16abs(max16abs(0.f, accV[0]), accV[1]);
      mk = wave_max(mk);
      mv = wave_max(mv);
      if (lane == 0) { sred[w] = act ? mk : 0.f; sred[8 + w] = act ? mv : 0.f; }
    }
    lds_barrier();     // B: every wave has read its last weight fragments
    STAMP();           // 4: barrier B passed (the ring's bytes become K / V^T) and the maxima are visible
    float kinv, vinv;
    {
      float mk = sred[0], mv = sred[8];
#pragma unroll
      for (int k = 1; k < 8; ++k) { mk = fmaxf(mk, sred[k]); mv = fmaxf(mv, sred[8 + k]); }
      const float ks = pow2_scale_for(mk), vs = pow2_scale_for(mv);
      kinv = __builtin_amdgcn_rcpf(ks);                         // (exact: powers of two)
      vinv = __builtin_amdgcn_rcpf(vs);
      if (act) {
        // K row-major [key][feature slots]: this lane = key row i, accumulator quad q of tile t = features 32 t + 8 q + 4 hi .. + 3, which the
        // score MFMAs consume as k-chunk c = 2 t + (q >> 1), lane half hi, slots 4 (q & 1) .. + 3
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            h16x4 hh, ll;
            split4(quad(accK[t], q) * ks, hh, ll);
            const int pos = i * LDK + (2 * t + (q >> 1)) * 16 + 8 * hi + 4 * (q & 1);
            *(h16x4*)&Kh[pos] = hh;
            *(h16x4*)&Kl[pos] = ll;
          }
        // V^T [feature][key slots]: this lane = feature 32 t + l31, accumulator quad q = keys 32 it + 8 q + 4 hi .. + 3 = key quad
        // G = 8 it + 2 q + hi -> slot (G >> 2) 16 + (G & 1) 8 + ((G >> 1) & 1) 4   (keys >= n: zeros; 0 x garbage must stay 0)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int key0 = it * 32 + 8 * q + 4 * hi;
            f32x4 y = quad(accV[t], q) * vs;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = key0 + e < n ? y[e] : 0.f;
            h16x4 hh, ll;
            split4(y, hh, ll);
            const int slot = (2 * it + (q >> 1)) * 16 + hi * 8 + (q & 1) * 4;
            *(h16x4*)&Vh[(t * 32 + l31) * LDV + slot] = hh;
            *(h16x4*)&Vl[(t * 32 + l31) * LDV + slot] = ll;
          }
      }
    }
    // Q fragments: k-chunk c of the score MFMAs = accumulator registers 8 (c & 1) .. + 7 of tile c >> 1, scaled per 32-query tile
    f16x8 qh[4], ql[4];
    float qk = 0.f;
    if (act) {
      const float sq = pow2_scale_for(wave_max(max16abs(max16abs(0.f, accQ[0]), accQ[1])));
#pragma unroll
      for (int c = 0; c < 4; ++c) split8(quad(accQ[c >> 1], 2 * (c & 1)), quad(accQ[c >> 1], 2 * (c & 1) + 1), sq, qh[c], ql[c]);
      qk = 0.0625f * __builtin_amdgcn_rcpf(sq) * kinv;           // 1 / sqrt(256) x the inverse operand scales (the weights' 2^16 = 2^8 x 2^8 below)
      qk *= kWScaleInv * kWScaleInv;
    }
    // the previous item's output leaves now: nothing behind it waits for a global load before the next item's prologue
    store_pending();
    STAMP();           // 5: K / V^T written, Q converted, previous item's stores issued
    lds_barrier();     // C: K, V^T of this item in LDS
    STAMP();           // 6: barrier C passed

    // L2 prefetch of the NEXT item's LN rows (this wave's 32 rows = 256 lines of 128 bytes, one dword per lane and line): they miss to HBM, and
    // in the prologue every wait for a weight piece would sit behind them (vmcnt is in order).  Nothing waits on vmcnt during the attention
    // phase; the values are consumed (by nothing) at its end.
    f32x4 pf = {0.f, 0.f, 0.f, 0.f};
    if (cur + bc_step < n_bc) {
      const float* nx = a.xn + (long)(cur + bc_step) * T * 256;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int line = e * 64 + lane;
        int r = it * 32 + (line >> 3);
        r = r < T ? r : T - 1;
        pf[e] = nx[r * 256 + (line & 7) * 32];
      }
    }
    // ================= ATTENTION (as attention_long_f16x3_kernel) =================
    float m = -1e30f, lp = 0.f;                                  // running maximum, PER-LANE partial sum (its own 16 keys per tile)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    if (act) {
#pragma unroll 1
      for (int jt = 0; jt <= it; ++jt) {
        f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
        {
          const _Float16* kph = Kh + (jt * 32 + l31) * LDK + hi * 8;
          const _Float16* kpl = Kl + (jt * 32 + l31) * LDK + hi * 8;
          f16x8 kh[4], kl[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) { kh[c] = *(const f16x8*)(kph + c * 16); kl[c] = *(const f16x8*)(kpl + c * 16); }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[c], qh[c], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[c], qh[c], sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[c], ql[c], sc, 0, 0, 0);
          }
        }
        float cm = -1e30f;
        const float jb = (float)(jt * 32) + hi4f;
        const bool masked = jt == it || (jt + 1) * 32 > n;
        if (masked) {
          const int i4 = i - jt * 32 - 4 * hi, n4 = n - jt * 32 - 4 * hi;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2);
            float v = fmaf(sc[r], qk, slope * ((float)c + jb));
            v = ((c <= i4) && (c < n4)) ? v : -1e30f;
            sc[r] = v;
            cm = fmaxf(cm, v);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = (r & 3) + 8 * (r >> 2);
            const float v = fmaf(sc[r], qk, slope * ((float)c + jb));
            sc[r] = v;
            cm = fmaxf(cm, v);
          }
        }
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);
        const float alpha = __expf(m - mn);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float p = __expf(sc[r] - mn);
          if (masked) p = sc[r] > -1e29f ? p : 0.f;
          sc[r] = p;
          sum += p;
        }
        lp = lp * alpha + sum;
        m = mn;
        const _Float16* vh = Vh + l31 * LDV + jt * 32 + hi * 8;
        const _Float16* vl = Vl + l31 * LDV + jt * 32 + hi * 8;
        f16x8 v0h[2], v1h[2], v0l[2], v1l[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          v0h[c] = *(const f16x8*)(vh + c * 16); v1h[c] = *(const f16x8*)(vh + 32 * LDV + c * 16);
          v0l[c] = *(const f16x8*)(vl + c * 16); v1l[c] = *(const f16x8*)(vl + 32 * LDV + c * 16);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          f16x8 ph, pl;
          split8(f32x4{sc[8 * c], sc[8 * c + 1], sc[8 * c + 2], sc[8 * c + 3]}, f32x4{sc[8 * c + 4], sc[8 * c + 5], sc[8 * c + 6], sc[8 * c + 7]},
                 kPScale, ph, pl);
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h[c], ph, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h[c], ph, o1, 0, 0, 0);
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0l[c], ph, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1l[c], ph, o1, 0, 0, 0);
          o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h[c], pl, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h[c], pl, o1, 0, 0, 0);
        }
      }
    }
    asm volatile("" ::"v"(pf[0] + pf[1] + pf[2] + pf[3]));   // (the prefetch loads end here)
    STAMP();           // 7: attention done
#ifdef VAPX_TRACE
    ++stamp_item;
#endif
    // this item's output: accumulator r of o0 / o1 <-> feature (r&3) + 8 (r>>2) + 4 hi (+ 32) of query row i; kept in registers until the
    // next item's prologue is through.  Rows >= T of the last tile: stored to row T - 1 with row T - 1's own value (they computed on the
    // clamped last row: the same bytes); rows in [n, T) and tiles beyond the window: zeros (the weights' 2^8 of V rides in the scale)
    {
      const float l = lp + __shfl_xor(lp, 32);
      const int ir = i < T ? i : T - 1;
      o_scl = (act && ir < n) ? vinv * kWScaleInv * (1.0f / kPScale) / l : 0.f;
      o_row = (long)cur * T + ir;
      o_h = h;
      if (it * 32 >= T) o_row = -1;                              // (tiles wholly beyond the window's capacity: nothing to write)
    }
  }
  }
  store_pending();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

hipError_t launch_attention_proj_f16x3(const AttnProjArgs& a, int B, hipStream_t st) {
  const int n_tiles = (a.T + 31) / 32;
  if (n_tiles > 8 || !a.xn || !a.wqkvp) return hipErrorInvalidValue;          // T <= 256 (vapx_create enforces it)
  if (B <= 0) return hipSuccess;
  const size_t lds = (size_t)2 * 256 * LDK * sizeof(_Float16) + (size_t)2 * 64 * LDV * sizeof(_Float16) + 16 * sizeof(float);
  static PerDeviceOnce attr_set;
  attr_set.run([] { (void)hipFuncSetAttribute((const void*)attention_proj_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  AttnProjArgs b = a;
  b.n_items = B * 8;                                       // (stream, channel, head)
  const int grid = std::min(b.n_items, device_cu_count());         // persistent: one workgroup per CU
  hipLaunchKernelGGL(attention_proj_f16x3_kernel, dim3(grid), dim3(512), lds, st, b);
  return hipGetLastError();
}
