// attention_long3_kernel: causal multi-head attention with the ALiBi key bias for long windows (64 < T <= 256), fp32 MFMA — the default
// path's attention of C3 (T = 250) and of the reference's bc / nod settings (T = 100).
//   reference: MultiHeadAttention.forward modules.py:82-110 (scale 1/sqrt(dim) = 1/16, :52), MultiHeadAttentionAlibi.get_alibi_mask
//   :162-188 (bias m_h * j on the KEY index).
// Round 3 rebuilt it on the structure the split-precision kernel (attention_f16x3.hip) arrived at:
//   * PERSISTENT: one 8-wave workgroup per CU walks over its (stream, channel, head) items; the next item's raw K / V rows are in flight —
//     16 x 16 bytes per lane, in registers — while the current item computes, so no item waits for a cold first round trip (the round-2
//     kernel, one workgroup per item and two per CU, spent 13-19 us of every 42 us workgroup on it: profiles/r03_experiments/attn_long_trace_*).
//   * K AND V sit in LDS for the whole item (K row-major [256][64], V TRANSPOSED [64][256]: 136 KB), written once; the round-2 kernel took its K
//     fragments from global / L2 per (query tile, key tile) pair — 4.5 reads of every K row, 9.1 GB of fetches per C3 launch against 3.4 now.
//   * v_mfma_f32_16x16x4_f32 (32 cycles, the same rate per flop as 32x32x2): SIXTEEN queries per tile, so the causal triangle splits evenly
//     over the eight waves with no merging — wave w owns query tiles w and 15 - w, 9 (16 queries x 32 keys) units each, whatever w — and
//     the two tiles of a wave walk the key tiles together: K and V fragments are fetched once for both.  The two waves of a SIMD always have
//     work of the same length, so one's softmax runs under the other's MFMAs for the whole item.
//   * S^T = K.Q^T as before: a query column lives in the 4 lanes r16 + 16 g, the softmax statistics need two shuffles per key tile, the row
//     sum is kept PER LANE and reduced once per item, and the accumulators of S^T are, untouched, the B operand of O^T = V^T.P^T (lane
//     group g holds keys 4 g + r and 16 + 4 g + r: k-slot g of MFMA step (accumulator, r)).
// Per-stream metadata goes through the scalar cache (uniform_load: inside the item loop hipcc would use vector loads and vmcnt(0)); the
// prefetched rows are touched BEFORE the item's output stores (vmcnt is one in-order counter).
#include <algorithm>
#include <type_traits>

#include "vap_kernels.h"

namespace {

constexpr int LDKF = 68;    // floats per LDS row of K: 64 features + 4 pad (272 B: conflict-free 16-byte reads down a column)
constexpr int LDVF = 260;   // floats per LDS row of V^T: 256 keys + 4 pad

// p[idx] for a wave-uniform idx through the SCALAR cache (see attention_f16x3.hip)
__device__ __forceinline__ int uniform_load(const int* p, int idx) {
  int v;
  asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p), "s"(idx * 4) : "memory");
  return v;
}

__global__ __launch_bounds__(512, 1) void attention_long3_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  float* Kf = lds_f;                          // [256][LDKF] K
  float* Vt = Kf + 256 * LDKF;                // [64][LDVF] V^T
  const int T = a.T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..7
  // this wave's two 16-query tiles: slot 0 = tile w, slot 1 = tile 15 - w (the longer one: key tiles 0 .. (15 - w) / 2)
  const int qt[2] = {w, 15 - w};
  const bool ringed = a.ring_rot != nullptr;
  const int dq = tid & 15;                                    // feature quad 4 dq .. 4 dq + 3 of the rows this thread stages

  // raw K / V rows of the item being staged next: thread (key = (u*512 + tid) >> 4, dq) for K; for V four CONSECUTIVE keys per round
  // (4 x 4 block, transposed in registers when it goes to LDS)
  f32x4 kr[8], vr[8];
  auto issue_kv = [&](int item) {
    const int h = item & 3, bc = item >> 2, b = bc >> 1;
    const int n = uniform_load(a.bn, b);
    const int rot = ringed ? uniform_load(a.ring_rot, b) : 0;
    const int kvbc = a.swap_kv ? (bc ^ 1) : bc;
    const long slab_kv = ringed ? ((long)(a.ids ? uniform_load(a.ids, b) : b) * 2 + (bc & 1)) : (long)kvbc;
    const float* kp = a.k + slab_kv * T * a.ldkv + h * 64 + dq * 4;     // item-uniform part + this thread's feature quad
    const float* vp = a.v + slab_kv * T * a.ldkv + h * 64 + dq * 4;
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
      for (int e = 0; e < 4; ++e) vr[u * 4 + e] = *(const f32x4*)(vp + row_off(4 * (u * 32 + (tid >> 4)) + e));
  };
  // wait for the rows in kr / vr here and now (called BEFORE an item's output stores: a wait issued after them would also sit out the
  // stores' acknowledgements — one in-order vmcnt)
  auto touch_kv = [&]() {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("" : "+v"(kr[u]));
      asm volatile("" : "+v"(vr[u]));
    }
  };

  int item = (int)blockIdx.x;
  if (item < a.n_items) issue_kv(item);
#pragma unroll 1
  for (; item < a.n_items; item += (int)gridDim.x) {
    const int h = item & 3, bc = item >> 2, b = bc >> 1;
    const int n = uniform_load(a.bn, b);
    const int rot = ringed ? uniform_load(a.ring_rot, b) : 0;
    const long slab_q = ringed ? ((long)(a.ids ? uniform_load(a.ids, b) : b) * 2 + (bc & 1)) : (long)bc;
    const int nt_valid = (n + 31) >> 5;
    const int n16 = (n + 15) >> 4;                       // 16-query tiles with valid rows
    const bool act[2] = {qt[0] < n16, qt[1] < n16};
    const float slope = exp2f(-2.0f * (float)(h + 1));  // [1/4, 1/16, 1/64, 1/256]
    // (fresh per item: hipcc otherwise hoists every lane-constant table derived from these — key indices, ALiBi biases, mask bounds, LDS
    // addresses — out of the item loop and keeps them live, i.e. spilled, across it)
    const int r16 = opaque_vgpr(lane & 15), g = opaque_vgpr(lane >> 4);
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
    // this wave's query rows (row 16 t + r16 clamped): lane group g holds features 16 g .. 16 g + 15 — k-slot g of the sixteen MFMA steps of a
    // score tile; K below uses the same mapping, so the order of the features inside the contraction is immaterial.  Issued unconditionally
    // (rows beyond the window are clamped anyway) so that hipcc can count the loads in flight
    f32x4 qf[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      int i = qt[s] * 16 + r16;
      i = i < n ? i : n - 1;
      int r = i + rot;
      r = r >= T ? r - T : r;
      const float* qp = a.q + (slab_q * T + r) * a.ldq + h * 64 + g * 16;
#pragma unroll
      for (int c = 0; c < 4; ++c) qf[s][c] = *(const f32x4*)(qp + c * 4);
    }
    FINE();            // 1: Q loads issued
    __syncthreads();   // A: every wave is done with the previous item's K / V in LDS
    FINE();            // 2: barrier A passed
    // K rows (rows >= n hold the clamped last row: finite, masked in the softmax)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int key = (u * 512 + tid) >> 4;
      if (key < nt_valid * 32) *(f32x4*)&Kf[key * LDKF + dq * 4] = kr[u];
    }
    // V^T (keys >= n: zeros; masked keys have P = 0 exactly, and 0 x garbage must stay 0)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int G = u * 32 + (tid >> 4);                   // keys 4 G .. 4 G + 3
      if (4 * G < nt_valid * 32) {
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
          f32x4 y;
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = 4 * G + e < n ? vr[u * 4 + e][dd] : 0.f;
          *(f32x4*)&Vt[(dq * 4 + dd) * LDVF + 4 * G] = y;
        }
      }
    }
    FINE();            // 3: LDS stores issued
    __syncthreads();   // B: K, V^T of this item in LDS
    STAMP();           // 4: staged
    // the next item's rows fly under this item's MFMAs (unconditional — the last item re-reads itself — so that hipcc can count the loads)
    issue_kv(item + (int)gridDim.x < a.n_items ? item + (int)gridDim.x : item);

    float m[2] = {-1e30f, -1e30f}, lp[2] = {0.f, 0.f};          // running maximum, PER-LANE partial sum (its own 8 keys per tile)
    f32x4 o[2][4];
    if (act[0]) {   // (slot 1 holds the later rows: act[1] implies act[0])
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) o[s][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int jlast[2] = {qt[0] >> 1, qt[1] >> 1};              // last (diagonal) key tile of each slot
      // one key tile for the slots S0 .. S1 - 1 ((0, 2): both query tiles, K / V fragments shared; (1, 2): only the longer one; (0, 1): only
      // the earlier one — the window is still filling and tile 15 - w has no valid row yet)
      auto step = [&](int jt, auto s0_tag, auto s1_tag) {
        constexpr int S0 = decltype(s0_tag)::value, S1 = decltype(s1_tag)::value;
        // ---- S^T = K_tile . Q^T: lane (query r16, group g) gets keys 4 g + r (first accumulator) and 16 + 4 g + r ----
        f32x4 sc[2][2];
#pragma unroll
        for (int s = S0; s < S1; ++s)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) sc[s][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4 kf[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) kf[c] = *(const f32x4*)&Kf[(jt * 32 + mt * 16 + r16) * LDKF + g * 16 + c * 4];
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int s = S0; s < S1; ++s) sc[s][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][e], qf[s][c][e], sc[s][mt], 0, 0, 0);
        }
        // ---- online softmax (running max / sum per 32-key tile), masked only on the diagonal tile of a slot and the tile with the window end ----
        const float jb = (float)(jt * 32 + 4 * g);
#pragma unroll
        for (int s = S0; s < S1; ++s) {
          float cm = -1e30f;
          const bool masked = jt == jlast[s] || (jt + 1) * 32 > n;
          if (masked) {
            const int i4 = qt[s] * 16 + r16 - jt * 32 - 4 * g, n4 = n - jt * 32 - 4 * g;     // key C + 4 g is visible iff C <= i4 and C < n4
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int c = mt * 16 + r;
                float v = fmaf(sc[s][mt][r], 0.0625f, slope * ((float)c + jb));
                v = ((c <= i4) && (c < n4)) ? v : -1e30f;
                sc[s][mt][r] = v;
                cm = fmaxf(cm, v);
              }
          } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float v = fmaf(sc[s][mt][r], 0.0625f, slope * ((float)(mt * 16 + r) + jb));
                sc[s][mt][r] = v;
                cm = fmaxf(cm, v);
              }
          }
          cm = fmaxf(cm, __shfl_xor(cm, 16));
          cm = fmaxf(cm, __shfl_xor(cm, 32));
          const float mn = fmaxf(m[s], cm);
          const float alpha = __expf(m[s] - mn);
          float sum = 0.f;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float pv = __expf(sc[s][mt][r] - mn);
              if (masked) pv = sc[s][mt][r] > -1e29f ? pv : 0.f;
              sc[s][mt][r] = pv;
              sum += pv;
            }
          lp[s] = lp[s] * alpha + sum;
          m[s] = mn;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) o[s][mt] *= alpha;
        }
        // ---- O^T = alpha O^T + V_tile^T . P^T: lane (query r16, group g) gets features 16 dt + 4 g + r; P comes straight from the score accumulators ----
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const f32x4 vf = *(const f32x4*)&Vt[(dt * 16 + r16) * LDVF + jt * 32 + mt * 16 + g * 4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int s = S0; s < S1; ++s) o[s][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[r], sc[s][mt][r], o[s][dt], 0, 0, 0);
          }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      FINE();          // 5: next item's loads issued
      if (act[1]) {
#pragma unroll 1
        for (int jt = 0; jt <= jlast[0]; ++jt) step(jt, I0{}, I2{});
        FINE();        // 6: key tiles shared by both query tiles done
#pragma unroll 1
        for (int jt = jlast[0] + 1; jt <= jlast[1]; ++jt) step(jt, I1{}, I2{});
        FINE();        // 7: remaining key tiles of the longer query tile done
      } else {
#pragma unroll 1
        for (int jt = 0; jt <= jlast[0]; ++jt) step(jt, I0{}, I1{});
      }
    }
    touch_kv();        // (next item's rows: issued a whole compute phase ago)
    FINE();            // 8: next item's rows arrived
    if (act[0]) {
      // ---- store: accumulator (dt, r) <-> feature 16 dt + 4 g + r of query row 16 t + r16 ----
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (act[s]) {
          float l = lp[s];
          l += __shfl_xor(l, 16);
          l += __shfl_xor(l, 32);
          const int i = qt[s] * 16 + r16;
          if (i < T) {
            float* op = a.out + ((long)bc * T + i) * 256 + h * 64 + g * 4;
            const float scl = i < n ? 1.0f / l : 0.f;         // rows beyond the valid window: deterministic zeros
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(f32x4*)(op + dt * 16) = o[s][dt] * scl;
          }
        }
    }
    // whole 16-query tiles beyond the valid rows: deterministic zeros
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (!act[s]) {
        const int i = qt[s] * 16 + r16;
        if (i < T) {
          float* op = a.out + ((long)bc * T + i) * 256 + h * 64 + g * 4;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) *(f32x4*)(op + dt * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    STAMP();   // 9: outputs stored
#ifdef VAPX_TRACE
    if (a.trace && tid == 0 && item < 16384) { a.trace[(long)item * 32 + 29] = __builtin_amdgcn_s_memrealtime(); a.trace[(long)item * 32 + 30] = (unsigned long long)stamp_k; }
#endif
  }
}

struct DeviceCUs {
  std::mutex mu;
  int cus[64] = {0};
  int get() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    int& c = cus[dev & 63];
    if (c == 0) {
      if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    }
    return c;
  }
};

}  // namespace

hipError_t launch_attention_long(const AttnArgs& a, int B, hipStream_t st) {
  const int n_tiles = (a.T + 31) / 32;
  if (n_tiles > 8) return hipErrorInvalidValue;          // T <= 256 (vapx_create enforces it)
  if (B <= 0) return hipSuccess;
  const size_t lds = ((size_t)256 * LDKF + (size_t)64 * LDVF) * sizeof(float);
  static PerDeviceOnce attr_set;
  attr_set.run([] { (void)hipFuncSetAttribute((const void*)attention_long3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  static DeviceCUs cus;
  AttnArgs b = a;
  b.n_items = B * 8;                                       // (stream, channel, head)
  const int grid = std::min(b.n_items, cus.get());         // persistent: one workgroup per CU
  hipLaunchKernelGGL(attention_long3_kernel, dim3(grid), dim3(512), lds, st, b);
  return hipGetLastError();
}
