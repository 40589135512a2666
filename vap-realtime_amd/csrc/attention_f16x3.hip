// attention_long_f16x3_kernel: the long-window attention of vap_kernels.hip (attention_long2_kernel: causal multi-head attention with
// the ALiBi key bias, 64 < T <= 256; reference MultiHeadAttention.forward modules.py:82-110, get_alibi_mask :162-188) with both
// contractions — S^T = K.Q^T over the 64 features of a head and O^T = V^T.P^T over the keys — as fp32-accurate SPLIT-PRECISION
// products on the f16 matrix cores (opt-in with the rest of the split path: VAPX_FLAG_SPLIT_F16).
//
//   x s = hi + lo (f16),   a.b = (hi_a.hi_b + lo_a.hi_b + hi_a.lo_b) / (s_a s_b)  + O(2^-22 |a||b|)        (ffn_block_f16x3.hip)
//
// Twelve v_mfma_f32_32x32x16_f16 per 32 x 32 score tile instead of thirty-two v_mfma_f32_32x32x2_f32 of twice the duration, the same again
// for P.V: 3/16 of the matrix-core time.  What is left is VALU work (softmax, operand conversion) and memory latency, and the kernel is
// built around those two:
//   * K AND V are converted ONCE per (stream, channel, head) item and kept in LDS as MFMA-ready f16 (hi, lo) operands — K row-major
//     [256 keys][64 features], V TRANSPOSED [64 features][256 key slots] because it is the A operand of O^T = V^T.P^T (lane = feature,
//     8 keys per lane and k-chunk) — 141 KB, ONE workgroup per CU, eight waves.  (A first version kept the fp32
//     kernel's structure — K fragments from global per (query tile, key tile) pair, converted in registers, two workgroups per CU — and
//     spent its time converting every K tile 4.5 times and waiting for first round trips: 14.1 ms per C3 tick against 17.4 fp32.)
//   * the kernel is PERSISTENT (one workgroup per CU walks over its items) and the next item's raw K / V rows are in flight — 16 x 16 bytes
//     per lane, in registers — while the current item computes: no workgroup ever waits for a cold first load except the very first.
//   * eight waves = one 32-query tile each (v_mfma_f32_32x32x16_f16); wave w takes query tile w (w < 4) or 11 - w, so the two waves of a SIMD
//     (w, w + 4) together own 9 causal key tiles, every SIMD the same.  (Tried and dropped: SIXTEEN-query tiles on v_mfma_f32_16x16x32_f16,
//     two per wave — tiles w and 15 - w, 9 units per wave whatever w, K / V fragments shared where both tiles need the same key tile.  Perfectly
//     balanced, but a 16 x 32 unit costs a wave as long as a 32 x 32 pair does — 0.82 us against 0.81: the step is bound by its fixed VALU /
//     latency chain, not by the tile area — so the balanced waves were as slow as this kernel's critical one: 11.2 ms per C3 tick against
//     10.7; git history has it.)
//
// No operand can overflow f16, whatever the input.  Q, K and V rows are raw projections (the cross-attention K / V of the RAW residual
// stream), so every operand carries a power-of-two scale from its own maximum: K and V per item (one block reduction each, sharing the
// barrier that hands the LDS over), Q per 32-query tile (wave-local); P is in [0, 1] and rides as 2^12 P so that its low half stays
// clear of the f16 denormals.  All scales are undone in fp32 (the score scale inside the softmax's fma, the V and P scales in the final
// 1 / l), exactly.  Inside each 16-key chunk the key slots of V^T are permuted so that the 8 keys a lane half needs (the keys whose
// probabilities sit in its 8 accumulator registers of S^T: 16c + 4 hi + {0..3} and 16c + 8 + 4 hi + {0..3}) are one 16-byte read.
// P never moves: accumulator registers 8c .. 8c+7 of S^T, converted, ARE the B operand of k-chunk c.  The row sum is kept PER LANE and the
// two lane halves of a query are added once per item.
#include <algorithm>
#include <type_traits>

#include "vap_kernels.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 72;             // halves per LDS row of K: 64 features + 8 pad (144 B: conflict-free 16-byte reads down a column)
constexpr int LDV = 264;            // halves per LDS row of V^T: 256 key slots + 8 pad (528 B)
constexpr float kPScale = 4096.0f;  // P rides as 2^12 P

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
__device__ __forceinline__ float max4abs(float mx, const f32x4& v) {
  return fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}
__global__ __launch_bounds__(512, 1) void attention_long_f16x3_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* Kh = (_Float16*)lds_raw;          // [256][LDK] K hi
  _Float16* Kl = Kh + 256 * LDK;              // [256][LDK] K lo
  _Float16* Vh = Kl + 256 * LDK;              // [64][LDV] V^T hi
  _Float16* Vl = Vh + 64 * LDV;               // [64][LDV] V^T lo
  float* sred = (float*)(Vl + 64 * LDV);      // [16] max |K|, max |V| per wave
  const int T = a.T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
  const int it = w < 4 ? w : 11 - w;                          // this wave's 32-query tile: the two waves of a SIMD (w, w + 4) own 9 key tiles together
  const bool ringed = a.ring_rot != nullptr;
  const int dq = tid & 15;                                    // K: feature quad 4 dq .. 4 dq + 3 of the rows this thread stages
  // V: thread -> (feature quad dqv, key quad gv) chosen so that the TRANSPOSED LDS stores are bank-conflict-free.  A ds_write_b64 is served
  // in groups of 16 consecutive lanes over 32 banks; a thread writes 8 bytes of row 4 dqv + dd at key slot(G), and rows 4 apart differ by
  // 16 banks only — so with lanes = dqv (round 3) a group's 16 stores fell on two bank pairs: 8-way conflicts, 59 % of the kernel's
  // LDS-array cycles (profiles/r03_c3_split_pmc.txt).  Now a group holds 2 feature quads (banks +0 / +16) x 8 key quads, whose slots
  // (G>>2)*16 + (G&1)*8 + ((G>>1)&1)*4 halves land on dword offsets {0,4,2,6,8,12,10,14}: 16 distinct bank pairs.  Global loads: one
  // wave instruction still reads whole 64-byte lines (8 keys x 128 contiguous bytes).
  const int dqv = (tid & 1) | (((tid >> 4) & 3) << 1) | (((tid >> 6) & 1) << 3);
  const int gv = ((tid >> 1) & 7) | (((tid >> 7) & 3) << 3);   // 0..31: key quad within a round of 32

  // raw K / V rows of the item being staged next: thread (key = (u*512 + tid) >> 4, dq) for K; for V four CONSECUTIVE keys 4 G .. 4 G + 3
  // (G = u*32 + gv) x feature quad dqv per round (4 x 4 block, transposed in registers when it goes to LDS)
  f32x4 kr[8], vr[8];
  auto issue_kv = [&](int item) {
    const int h = item & 3, bc = item >> 2, b = bc >> 1;
    const int n = uniform_load(a.bn, b);
    const int rot = ringed ? uniform_load(a.ring_rot, b) : 0;
    const int kvbc = a.swap_kv ? (bc ^ 1) : bc;
    const long slab_kv = ringed ? ((long)(a.ids ? uniform_load(a.ids, b) : b) * 2 + (bc & 1)) : (long)kvbc;
    const float* kp = a.k + slab_kv * T * a.ldkv + h * 64 + dq * 4;     // item-uniform part + this thread's feature quad
    const float* vp = a.v + slab_kv * T * a.ldkv + h * 64 + dqv * 4;
    auto row_off = [&](int j) {                                          // 32-bit offset of logical row j (clamped to the window)
      j = j < n ? j : n - 1;
      int r = j + rot;
      r = r >= T ? r - T : r;
      return (unsigned)(r * a.ldkv);
    };
#pragma unroll
    for (int u = 0; u < 8; ++u) kr[u] = *(const f32x4*)(kp + row_off((u * 512 + tid) >> 4));
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) vr[u * 4 + e] = *(const f32x4*)(vp + row_off(4 * (u * 32 + gv) + e));
  };

  // maxima of the rows in kr / vr -> sred (the power-of-two scales of the item they belong to).  Called at the END of the previous item's
  // compute phase, BEFORE its output stores: vmcnt is one in-order counter, so a wait for these loads issued after the stores would also
  // sit out the stores' acknowledgements (2.7 us per item when the wait was at the top of the loop)
  auto take_maxima = [&]() {
    float mk = 0.f, mv = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) { mk = max4abs(mk, kr[u]); mv = max4abs(mv, vr[u]); }
    mk = wave_max(mk);
    mv = wave_max(mv);
    if (lane == 0) { sred[w] = mk; sred[8 + w] = mv; }
  };
  int item = (int)blockIdx.x;
  if (item < a.n_items) { issue_kv(item); take_maxima(); }
#pragma unroll 1
  for (; item < a.n_items; item += (int)gridDim.x) {
    const int h = item & 3, bc = item >> 2, b = bc >> 1;
    const int n = uniform_load(a.bn, b);
    const int rot = ringed ? uniform_load(a.ring_rot, b) : 0;
    const long slab_q = ringed ? ((long)(a.ids ? uniform_load(a.ids, b) : b) * 2 + (bc & 1)) : (long)bc;
    const int nt_valid = (n + 31) >> 5;
    const bool act = it < nt_valid;
    const float slope = exp2f(-2.0f * (float)(h + 1));  // [1/4, 1/16, 1/64, 1/256]
    // (fresh per item: hipcc otherwise hoists every lane-constant table derived from these — key indices, ALiBi biases, mask bounds, LDS
    // addresses — out of the item loop and keeps them live, i.e. spilled, across it)
    const int l31 = opaque_vgpr(lane & 31), hi = opaque_vgpr(lane >> 5);
    const float hi4f = (float)(4 * hi);
#ifdef VAPX_TRACE
    int stamp_k = 0;
    auto STAMP = [&]() {
      if (a.trace && tid == 0 && item < 16384) a.trace[(long)item * 32 + stamp_k] = __builtin_amdgcn_s_memtime();
      ++stamp_k;
    };
    STAMP();   // 0: item start
    if (a.trace && tid == 0 && item < 16384) a.trace[(long)item * 32 + 28] = __builtin_amdgcn_s_memrealtime();
    auto FINE = [&]() { __builtin_amdgcn_sched_barrier(0); STAMP(); __builtin_amdgcn_sched_barrier(0); };
#else
    auto STAMP = [] {};
    auto FINE = [] {};
#endif
    // this wave's query rows: raw fragments (row 32 it + l31 clamped; qraw[2c], qraw[2c+1] = features 16 c + 8 hi .. + 7, the 8 k-values of this
    // lane half in the 16-feature chunk c — the same 16 bytes of a row-major K row in LDS below) fly while K / V are converted
    // (issued unconditionally — rows beyond the window are clamped anyway — so that hipcc can count them: behind a branch the wait for
    // the K / V rows below becomes vmcnt(0) and sits out a fresh round trip of these eight loads, 2.7 us per item)
    f32x4 qraw[8];
    {
      int i = it * 32 + l31;
      i = i < n ? i : n - 1;
      int r = i + rot;
      r = r >= T ? r - T : r;
      const float* qp = a.q + (slab_q * T + r) * a.ldq + h * 64 + hi * 8;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) qraw[kc] = *(const f32x4*)(qp + (kc >> 1) * 16 + (kc & 1) * 4);
    }
    FINE();            // 1: Q loads issued
    __syncthreads();   // A: maxima (taken at the end of the previous item) visible; every wave is done with the previous item's K / V in LDS
    FINE();            // 2: barrier A passed
    float kinv, vinv;
    {
      float mk = sred[0], mv = sred[8];
#pragma unroll
      for (int i = 1; i < 8; ++i) { mk = fmaxf(mk, sred[i]); mv = fmaxf(mv, sred[8 + i]); }
      const float ks = pow2_scale_for(mk), vs = pow2_scale_for(mv);
      kinv = __builtin_amdgcn_rcpf(ks);                     // (exact: powers of two)
      vinv = __builtin_amdgcn_rcpf(vs);
      // K rows (rows >= n hold the clamped last row: finite, masked in the softmax)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int key = (u * 512 + tid) >> 4;
        if (key < nt_valid * 32) {
          h16x4 hh, ll;
          split4(kr[u] * ks, hh, ll);
          *(h16x4*)&Kh[key * LDK + dq * 4] = hh;
          *(h16x4*)&Kl[key * LDK + dq * 4] = ll;
        }
      }
      // V^T (keys >= n: zeros; masked keys have P = 0 exactly, and 0 x garbage must stay 0)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int G = u * 32 + gv;                           // keys 4 G .. 4 G + 3
        if (4 * G < nt_valid * 32) {
          const int slot = (G >> 2) * 16 + (G & 1) * 8 + ((G >> 1) & 1) * 4;   // 16-key chunk, lane half that consumes the quad, first / second quad
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) {
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = 4 * G + e < n ? vr[u * 4 + e][dd] * vs : 0.f;
            h16x4 hh, ll;
            split4(y, hh, ll);
            *(h16x4*)&Vh[(dqv * 4 + dd) * LDV + slot] = hh;
            *(h16x4*)&Vl[(dqv * 4 + dd) * LDV + slot] = ll;
          }
        }
      }
    }
    FINE();            // 3: converted, LDS stores issued
    __syncthreads();   // B: K, V^T of this item in LDS
    STAMP();   // 4: staged
    f16x8 qh[4], ql[4];
    float qk = 0.f;
    if (act) {
      float mx = 0.f;
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) mx = max4abs(mx, qraw[kc]);
      const float sq = pow2_scale_for(wave_max(mx));
#pragma unroll
      for (int c = 0; c < 4; ++c) split8(qraw[2 * c], qraw[2 * c + 1], sq, qh[c], ql[c]);
      qk = 0.0625f * __builtin_amdgcn_rcpf(sq) * kinv;       // 1 / sqrt(256) x the inverse operand scales
    }
    FINE();            // 5: Q converted
    // the next item's rows fly under this item's MFMAs (unconditional — the last item re-reads itself — so that hipcc can count the loads)
    issue_kv(item + (int)gridDim.x < a.n_items ? item + (int)gridDim.x : item);
    float m = -1e30f, lp = 0.f;                              // running maximum, PER-LANE partial sum (its own 16 keys per tile)
    f32x16 o0, o1;
    const int i = it * 32 + l31;
    if (act) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
      FINE();          // 6
#pragma unroll 1
      for (int jt = 0; jt <= it; ++jt) {
        // ---- S^T tile = K_tile . Q^T (x s_k s_q): 12 MFMAs into one accumulator ----
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
        // ---- online softmax update for this 32-key tile (as attention_long2_kernel): sc := P.  accumulator r <-> key
        //      j = 32 jt + C_r + 4 hi with C_r = (r&3) + 8 (r>>2).  MASKED only for the diagonal tile (causal) and the tile with the window end ----
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
        // ---- O^T = alpha O^T + V_tile^T . P^T   (x s_v 2^12) ----
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
      FINE();          // 7: key tiles done
    }
    take_maxima();     // (next item's rows: issued a whole compute phase ago)
    FINE();            // 8: next item's rows arrived, maxima taken
    if (act) {
      // ---- store: accumulator r of o0 / o1 <-> feature (r&3) + 8 (r>>2) + 4 hi (+ 32) of query row i ----
      const float l = lp + __shfl_xor(lp, 32);
      if (i < T) {
        float* op = a.out + ((long)bc * T + i) * 256 + h * 64;
        const float scl = i < n ? vinv * (1.0f / kPScale) / l : 0.f;         // rows beyond the valid window: deterministic zeros
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          f32x4 v0 = {o0[rr * 4 + 0], o0[rr * 4 + 1], o0[rr * 4 + 2], o0[rr * 4 + 3]};
          f32x4 v1 = {o1[rr * 4 + 0], o1[rr * 4 + 1], o1[rr * 4 + 2], o1[rr * 4 + 3]};
          *(f32x4*)(op + rr * 8 + hi * 4) = v0 * scl;
          *(f32x4*)(op + 32 + rr * 8 + hi * 4) = v1 * scl;
        }
      }
    } else if (it * 32 < T) {                               // whole tile beyond the valid rows: deterministic zeros
      if (i < T) {
        float* op = a.out + ((long)bc * T + i) * 256 + h * 64;
#pragma unroll
        for (int d = 0; d < 8; ++d) *(f32x4*)(op + hi * 32 + d * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    STAMP();   // 9: outputs stored
#ifdef VAPX_TRACE
    if (a.trace && tid == 0 && item < 16384) { a.trace[(long)item * 32 + 29] = __builtin_amdgcn_s_memrealtime(); a.trace[(long)item * 32 + 30] = (unsigned long long)stamp_k; }
#endif
  }
}

}  // namespace

hipError_t launch_attention_f16x3(const AttnArgs& a, int B, hipStream_t st) {
  const int n_tiles = (a.T + 31) / 32;
  if (n_tiles > 8) return hipErrorInvalidValue;          // T <= 256 (vapx_create enforces it)
  if (B <= 0) return hipSuccess;
  const size_t lds = (size_t)2 * 256 * LDK * sizeof(_Float16) + (size_t)2 * 64 * LDV * sizeof(_Float16) + 16 * sizeof(float);
  static PerDeviceOnce attr_set;
  attr_set.run([] { (void)hipFuncSetAttribute((const void*)attention_long_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  AttnArgs b = a;
  b.n_items = B * 8;                                       // (stream, channel, head)
  const int grid = std::min(b.n_items, device_cu_count());         // persistent: one workgroup per CU
  hipLaunchKernelGGL(attention_long_f16x3_kernel, dim3(grid), dim3(512), lds, st, b);
  return hipGetLastError();
}
