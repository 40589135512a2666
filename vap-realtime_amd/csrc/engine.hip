// libvapx engine: owns device weights, per-stream state and scratch; orchestrates one VAP frame for
// a batch of streams (the C ABI of include/vapx.h).  Compiled with hipcc for gfx950 only.
//
// Reference path being replaced: VAPRealTime.__init__/process_vap, rvap/vap_main/vap_main.py:192-335.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/vapx.h"
#include "fused_blocks.h"
#include "gemm_f32.h"
#include "vap_kernels.h"
#include "vapx_layout.h"

namespace {

thread_local std::string g_create_error;

struct Layer {
  const float *ln_self_g, *ln_self_b, *wqkv, *wproj;
  const float *ln_src_g, *ln_src_b, *wq_x, *wkv_x, *wproj_x;
  const float *ln_ffn_g, *ln_ffn_b, *w0, *w3;
  const float *w0f, *w3f, *wqkvf, *wkvxf, *wprojf, *wqxf, *wprojxf;   // fragment-major copies (fused blocks)
  const float *w0h, *w3h, *wqkvh, *wkvxh, *wprojh, *wqxh, *wprojxh;                             // split-precision (f16 hi/lo) fragment copies
  const float *wproj8, *wqx8, *wprojx8;   // the attention projections in the 8-wave format of the 64-row flat-row blocks (long windows)
  const float* wqkvp;                     // per-head Q|K|V weight stream of attention_proj_f16x3_kernel (layers >= 1)
  float hid_scale = 1.0f;   // split-precision path: static power-of-two scale of the GELU hidden row (1 unless the weights allow |gelu(h)| >= 2^15)
};

}  // namespace

struct Scratch {
  int *bn = nullptr, *bhead = nullptr, *rot = nullptr;
  float *h0 = nullptr, *h1 = nullptr, *h2 = nullptr, *h3 = nullptr, *z = nullptr, *gx = nullptr, *lstm_out = nullptr, *e = nullptr;
  float* xl[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // layer inputs/outputs: x0, o, stereo0..2
  float *xn = nullptr, *xmid = nullptr, *att = nullptr, *qkv = nullptr, *qx = nullptr, *kvx = nullptr, *ffn = nullptr;
  float* last[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float *en = nullptr, *qkv_new = nullptr;   // [B*2][256], [B*2][768]: LN0(e) and layer-0 Q|K|V of the new row
  float* lffn = nullptr;                     // [B*2][768] FFN hidden of the last-row path  // [B*2][256] each: x, xn, q, att, xmid, out (last-row path)
  Scratch slice(size_t b0, const int* P, int ncpc, int T) const {
    Scratch s = *this;
    const size_t bc = b0 * 2, rows = bc * T;
    s.bn += b0; s.bhead += b0; s.rot += b0;
    s.h0 += bc * (P[0] + 4) * 256; s.h1 += bc * (P[1] + 2) * 256; s.h2 += bc * (P[2] + 2) * 256; s.h3 += bc * (P[3] + 2) * 256;
    s.z += bc * ncpc * 256; s.gx += bc * ncpc * 1024; s.lstm_out += bc * ncpc * 256; s.e += bc * 256;
    for (int i = 0; i < 5; ++i) s.xl[i] += rows * 256;
    s.xn += rows * 256; s.xmid += rows * 256; s.att += rows * 256; s.qkv += rows * 768; s.qx += rows * 256;
    s.kvx += rows * 512; s.ffn += rows * 768;
    for (int i = 0; i < 6; ++i) s.last[i] += bc * 256;
    s.en += bc * 256; s.qkv_new += bc * 768; s.lffn += bc * 768;
    return s;
  }
};

// per-stream state bases; for identity stream ids of a sub-batch starting at slot b0 the bases are
// simply advanced by b0 streams
struct StateView {
  float *ring, *ring_qkv, *h_state, *c_state, *carry;
  int* frames_seen;
};

struct vapx_engine {
  vapx_config cfg;
  int hop, L, P[5], ncpc, T, K;
  std::string err;

  float* w = nullptr;  // weight blob
  size_t w_floats = 0;
  const vapx_layout::Entry* lay = nullptr;
  size_t lay_n = 0;
  Layer layer[4];

  // per-stream state
  float *ring = nullptr, *ring_qkv = nullptr, *h_state = nullptr, *c_state = nullptr, *carry = nullptr;
  int* frames_seen = nullptr;

  // scratch (max_batch); every buffer is linear in the batch index, so a sub-batch starting at
  // stream slot b0 is just the same struct with offset pointers (Scratch::slice)
  float *audio_dev = nullptr, *out_dev = nullptr;
  int* ids_dev = nullptr;
  Scratch sc;
  // intra-tick overlap: the batch is split into groups that run on their own HIP streams
  static constexpr int kMaxGroups = 8;
  hipStream_t gstream[kMaxGroups] = {};
  hipEvent_t gdone[kMaxGroups] = {};
  hipEvent_t gstart = nullptr;
  int n_groups = 1;
  // debug knob (env VAPX_POISON_SCRATCH): every scratch buffer is refilled with NaN bit patterns before each step and the rings start as
  // NaNs, so a kernel that consumes anything it (or an earlier kernel of the same tick) did not write shows up as a non-finite output
  std::vector<std::pair<void*, size_t>> poison;
#ifdef VAPX_TRACE   // debug build only (make trace -> libvapx_trace.so): per-workgroup phase stamps, tools/ffn_trace.py / tools/attn_trace.py
  unsigned long long* ffn_trace = nullptr;   // env VAPX_FFN_TRACE=<file>: phase stamps of the layer-0 FFN block's workgroups
  unsigned long long* attn_trace = nullptr;  // env VAPX_ATTN_TRACE=<file>: phase stamps of the layer-1 self-attention block's workgroups
  size_t ffn_trace_wgs = 0, attn_trace_wgs = 0;
  std::string ffn_trace_path, attn_trace_path;
#endif
  float* out_pinned = nullptr;
  float* audio_pinned = nullptr;          // staging for pageable host audio (callers holding vapx_host_alloc memory skip it)
  hipEvent_t audio_evt = nullptr;         // the H2D copy out of audio_pinned has completed
  int* ids_pinned = nullptr;
  hipEvent_t ids_evt = nullptr;
  int last_B = 0, last_G = 1;
  bool deferred_pending = false;          // the latest step left its overlap groups un-joined (VAPX_DEFER_JOIN)
  std::vector<int32_t> pending_resets;    // vapx_reset_stream requests, applied stream-ordered by the next step
  std::vector<uint32_t> id_stamp;         // duplicate-id check: id_stamp[sid] == id_gen <=> sid already in this batch
  uint32_t id_gen = 0;
  std::vector<int32_t> bad_slots;         // batch slots of the latest host-output step whose results were not finite

  // shared trunk (vapx_attach_trunk): followers take the leader's LSTM outputs instead of running the CPC encoder
  vapx_engine* trunk = nullptr;           // set on a follower
  std::vector<vapx_engine*> followers;    // set on the leader
  bool orphaned = false;                  // follower whose leader was destroyed
  uint64_t tick = 0, followed_tick = 0;
  const int* last_ids = nullptr;          // device ids of the latest step (null = identity)

  // optional per-kernel-class HIP-event timing (vapx_profile_*): events are recorded on the launch
  // stream around the launches whose class bit is set in prof_mask
  uint32_t prof_mask = 0;
  struct ProfRec { hipEvent_t a, b; int cls; };
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> prof_pool;

  const float* W(const char* name) const {
    for (size_t i = 0; i < lay_n; ++i)
      if (!strcmp(lay[i].name, name)) return w + lay[i].off;
    return nullptr;
  }
};

namespace {

int fail(vapx_engine* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  else g_create_error = buf;
  return code;
}

#define HIPCHK(h, expr)                                                                          \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) return fail(h, VAPX_E_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// debug knob (env VAPX_GUARD_ZONES): every device allocation of the engine gets a 4 KiB canary zone on both sides; vapx_peek("guard_violations")
// counts the canary bytes that changed, i.e. out-of-bounds writes just past (or before) a buffer
constexpr size_t kGuardBytes = 4096;
bool guard_enabled() { static const bool on = getenv("VAPX_GUARD_ZONES") != nullptr; return on; }
struct GuardRec { char* base; size_t bytes; };
std::mutex g_guard_mu;
std::vector<GuardRec> g_guards;

template <typename T>
hipError_t dalloc(T** p, size_t n, bool zero = true) {
  const size_t bytes = n * sizeof(T);
  if (!guard_enabled()) {
    hipError_t e = hipMalloc((void**)p, bytes);
    if (e != hipSuccess) return e;
    return zero ? hipMemset(*p, 0, bytes) : hipSuccess;
  }
  char* base = nullptr;
  hipError_t e = hipMalloc((void**)&base, bytes + 2 * kGuardBytes);
  if (e != hipSuccess) return e;
  if ((e = hipMemset(base, 0xA5, kGuardBytes)) != hipSuccess) return e;
  if ((e = hipMemset(base + kGuardBytes + bytes, 0xA5, kGuardBytes)) != hipSuccess) return e;
  if (zero && bytes && (e = hipMemset(base + kGuardBytes, 0, bytes)) != hipSuccess) return e;
  *p = (T*)(base + kGuardBytes);
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guards.push_back({base, bytes});
  return hipSuccess;
}
void dfree(void* p) {
  if (!p) return;
  if (guard_enabled()) {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    for (size_t i = 0; i < g_guards.size(); ++i)
      if (g_guards[i].base + kGuardBytes == (char*)p) {
        (void)hipFree(g_guards[i].base);
        g_guards.erase(g_guards.begin() + (long)i);
        return;
      }
  }
  (void)hipFree(p);
}
// canary bytes overwritten around ANY live allocation of the process's engines (device must be idle)
long guard_violations() {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  std::vector<unsigned char> host(kGuardBytes);
  long bad = 0;
  for (const GuardRec& r : g_guards)
    for (int side = 0; side < 2; ++side) {
      if (hipMemcpy(host.data(), side ? r.base + kGuardBytes + r.bytes : r.base, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return -1;
      for (unsigned char c : host) bad += c != 0xA5;
    }
  return bad;
}

// true when p is page-locked host memory known to HIP (hipHostMalloc / hipHostRegister), i.e. a real async copy source / target
bool is_pinned_host(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

int rate_ok(int hz) { return hz == 5 || hz == 10 || hz == 20 || hz == 50; }

void geometry(int hz, int* hop, int* L, int P[5], int* ncpc) {
  *hop = 16000 / hz;
  *L = *hop + VAPX_PAD;
  P[0] = *L / 5;
  P[1] = P[0] / 4;
  P[2] = P[1] / 2;
  P[3] = P[2] / 2;
  P[4] = P[3] / 2;
  *ncpc = P[4] - 2;
}

// kernel classes for profiling: 0..4 = GEMM by epilogue (EPI_STORE .. EPI_CN_RELU), then the rest; EPI_BIAS_LN_GELU (= 5 as an epilogue id)
// is booked as class 13 — 5 is the fused conv tail (round 3 booked both under 5: a follower's downsample GEMM was priced as conv_tail)
enum { CLS_CONVTAIL = 5, CLS_FFN = 6, CLS_LASTROW = 7, CLS_CONV0 = 8, CLS_LSTM = 9, CLS_GATHER = 10, CLS_ATTN = 11, CLS_HEAD = 12,
       CLS_GEMM_BIAS_LN_GELU = 13,
       CLS_FFN_PROJ = 14,   // long windows: the mode-2 flat-row block (attention output projection + LN_src + cross-attention query): two
                            // contractions per row where the FFN block proper has seven to twelve — its own class so that a class's
                            // FLOPs, time and HBM bytes describe the same launches (bench.py roofline)
       CLS_COUNT = 15 };
static inline int gemm_class(int epi) { return epi == EPI_BIAS_LN_GELU ? CLS_GEMM_BIAS_LN_GELU : epi; }

struct ProfScope {
  vapx_engine* h; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; int cls;
  ProfScope(vapx_engine* h_, int cls_, hipStream_t st_) : h(h_), st(st_), cls(cls_) {
    if (!(h->prof_mask & (1u << cls))) return;
    auto get = [&]() {
      hipEvent_t e = nullptr;
      if (!h->prof_pool.empty()) { e = h->prof_pool.back(); h->prof_pool.pop_back(); }
      else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
      return e;
    };
    a = get(); b = get();
    if (a) (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (!a || !b) return;
    (void)hipEventRecord(b, st);
    h->prof_recs.push_back({a, b, cls});
  }
};

// bounded_A: the A operand is bounded by construction (a LayerNorm / ChannelNorm+ReLU output, LSTM outputs): only then may the opt-in
// split-precision product run (f16 operands overflow at 65504); GEMMs on RAW residual-stream / attention rows stay on the fp32 MFMA
hipError_t gemm(vapx_engine* h, const GemmArgs& g, int epi, hipStream_t st, bool bounded_A = true) {
  ProfScope ps(h, gemm_class(epi), st);
  if ((h->cfg.flags & VAPX_FLAG_SPLIT_F16) && bounded_A) {
    GemmArgs gs = g;
    gs.split = 1;
    return launch_gemm_f32(gs, epi, 0, st);
  }
  return launch_gemm_f32(g, epi, 0, st);
}

GemmArgs gemm_args(const float* A, RowMap am, const float* W, int M, int N, int K, float* C, RowMap cm) {
  GemmArgs g;
  memset(&g, 0, sizeof g);
  g.A = A; g.am = am; g.W = W; g.M = M; g.N = N; g.K = K; g.C = C; g.cm = cm;
  g.rm = cm; g.c2m = cm;
  return g;
}

// ---- the CPC encoder on B streams: frames -> e [B*2][256] -------------------------------------
int run_encoder(vapx_engine* h, const Scratch& sc, const StateView& sv, int B, const int* ids_dev, const float* audio,
                int spc, bool use_state_meta, hipStream_t st) {
  const int* P = h->P;
  Conv0Args c0;
  c0.audio = audio; c0.ids = ids_dev; c0.carry = use_state_meta ? sv.carry : nullptr;
  c0.h0 = sc.h0; c0.w = h->W("conv0.w"); c0.bias = h->W("conv0.b"); c0.gamma = h->W("cn0.g"); c0.beta = h->W("cn0.b");
  c0.frames_seen = use_state_meta ? sv.frames_seen : nullptr; c0.bn = sc.bn; c0.bhead = sc.bhead;
  c0.L = h->L; c0.spc = spc; c0.T = h->T;
  { ProfScope ps(h, CLS_CONV0, st); HIPCHK(h, launch_conv0(c0, B, st)); }

  struct ConvSpec { const float* in; int Pin, guard_in, k, s; float* out; int Pout, guard_out; const char* idx; };
  const ConvSpec cs[3] = {
      {sc.h0, P[0], 2, 8, 4, sc.h1, P[1], 1, "1"},
      {sc.h1, P[1], 1, 4, 2, sc.h2, P[2], 1, "2"},
      {sc.h2, P[2], 1, 4, 2, sc.h3, P[3], 1, "3"},
  };
  // one stream per workgroup pays 27 % row padding: worth it only while the three GEMMs cannot fill the chip
  // (on the split-precision path the three GEMMs are 2x cheaper and beat the fp32 fused tail even at 256 streams)
  const bool fused_tail = conv_tail_supported(P[1], h->ncpc) && !(h->cfg.flags & VAPX_FLAG_UNFUSED_CONV) &&
                          !(h->cfg.flags & VAPX_FLAG_SPLIT_F16) && B <= 512;
  char nm[32];
  for (int i = 0; i < (fused_tail ? 1 : 3); ++i) {
    const ConvSpec& c = cs[i];
    RowMap am{(long)(c.Pin + 2 * c.guard_in) * 256, (long)c.s * 256, c.Pout};
    RowMap cm{(long)(c.Pout + 2 * c.guard_out) * 256, 256, c.Pout};
    snprintf(nm, sizeof nm, "conv%s.w", c.idx);
    GemmArgs g = gemm_args(c.in, am, h->W(nm), B * 2 * c.Pout, 256, c.k * 256, c.out + c.guard_out * 256, cm);
    snprintf(nm, sizeof nm, "conv%s.w16", c.idx); g.W16 = h->W(nm);
    snprintf(nm, sizeof nm, "conv%s.b", c.idx); g.bias = h->W(nm);
    snprintf(nm, sizeof nm, "cn%s.g", c.idx); g.gamma = h->W(nm);
    snprintf(nm, sizeof nm, "cn%s.b", c.idx); g.beta = h->W(nm);
    HIPCHK(h, gemm(h, g, EPI_CN_RELU, st));
  }
  if (fused_tail) {
    // conv2 -> conv3 -> conv4 for one stream per workgroup, intermediates in LDS
    ConvTailArgs ct;
    ct.h1 = sc.h1; ct.w2f = h->W("conv2.wf"); ct.w3f = h->W("conv3.wf"); ct.w4f = h->W("conv4.wf");
    ct.b2 = h->W("conv2.b"); ct.g2 = h->W("cn2.g"); ct.be2 = h->W("cn2.b");
    ct.b3 = h->W("conv3.b"); ct.g3 = h->W("cn3.g"); ct.be3 = h->W("cn3.b");
    ct.b4 = h->W("conv4.b"); ct.g4 = h->W("cn4.g"); ct.be4 = h->W("cn4.b");
    ct.z = sc.z; ct.P1 = P[1]; ct.ncpc = h->ncpc;
    { ProfScope ps(h, CLS_CONVTAIL, st); HIPCHK(h, launch_conv_tail(ct, B, st)); }
  } else {  // conv4: only positions 1..P4-2 survive z[:, 1:-1] (encoder.py:76)
    RowMap am{(long)(P[3] + 2) * 256, 2 * 256, h->ncpc};
    GemmArgs g = gemm_args(sc.h3 + 2 * 256, am, h->W("conv4.w"), B * 2 * h->ncpc, 256, 4 * 256, sc.z, contiguous_rows(256));
    g.W16 = h->W("conv4.w16");
    g.bias = h->W("conv4.b"); g.gamma = h->W("cn4.g"); g.beta = h->W("cn4.b");
    HIPCHK(h, gemm(h, g, EPI_CN_RELU, st));
  }
  {  // LSTM input projection for all n_cpc steps at once: gx = z.W_ih^T + (b_ih + b_hh)
    GemmArgs g = gemm_args(sc.z, contiguous_rows(256), h->W("lstm.wih"), B * 2 * h->ncpc, 1024, 256, sc.gx, contiguous_rows(1024));
    g.W16 = h->W("lstm.wih16");
    g.bias = h->W("lstm.b");
    HIPCHK(h, gemm(h, g, EPI_STORE, st));
  }
  LstmArgs la;
  la.gx = sc.gx; la.ids = ids_dev; la.h_state = sv.h_state; la.c_state = sv.c_state;
  la.wfrag = h->W("lstm.whh"); la.out = sc.lstm_out; la.M = B * 2; la.ncpc = h->ncpc;
  // downsample (single-output Conv1d == dense [ncpc*256 -> 256]) + LN + GELU fused into the LSTM kernel
  la.down_wf = h->W("down.wf"); la.down_b = h->W("down.b"); la.down_g = h->W("down.g"); la.down_beta = h->W("down.beta");
  la.e = sc.e;
  la.ln0_g = h->layer[0].ln_self_g; la.ln0_b = h->layer[0].ln_self_b; la.en = use_state_meta ? sc.en : nullptr;
  { ProfScope ps(h, CLS_LSTM, st); HIPCHK(h, launch_lstm(la, st)); }
  return VAPX_OK;
}

struct RingView { const float* ring; const float* ring_qkv; const int* ids; };   // layer 0 reads the rings directly

// ---- 1 self + 3 self/cross layers on x0 = xl[l_begin] (LN_self already in xn) ---------------------
// Per layer: [QKV (+cross KV) projections] -> self-attention -> proj+residual+LN -> (cross: q GEMM,
// cross-attention, proj+residual+LN) -> fused FFN block, which also emits the NEXT layer's
// projections so that only the first executed layer needs stand-alone projection GEMMs.
int run_layers(vapx_engine* h, const Scratch& sc, int B, hipStream_t st, int l_begin = 0, int l_end = 4,
               bool prune_last = false, bool qkv0_ready = false, const RingView* rv = nullptr) {
  const int T = h->T;
  const int M = B * 2 * T;
  const RowMap r256 = contiguous_rows(256), r768 = contiguous_rows(768), r512 = contiguous_rows(512);
  if (prune_last && (l_end != 4 || l_begin > 2)) prune_last = false;
  const int l_full_end = prune_last ? 3 : l_end;
  // split path, long windows: the self-attention of a layer whose LN_self rows were written by the previous layer's flat-row block projects
  // its own Q|K|V (attention_proj_f16x3_kernel); that block then skips the three contractions and never writes sc.qkv
  // (windows of 257 .. 512 frames: attention_xl_kernel — fp32, Q|K|V from the flat-row blocks — on both paths)
  const bool qkv_in_attn = (h->cfg.flags & VAPX_FLAG_SPLIT_F16) && !(h->cfg.flags & (VAPX_FLAG_SPLIT_QKV_IN_FFN | VAPX_FLAG_UNFUSED_PROJ)) && T > 64 && T <= 256;
  bool xn_ready = false;                   // sc.xn holds LN_self(layer l)(x) of every row, sc.qkv does NOT hold this layer's Q|K|V
  for (int l = l_begin; l < l_full_end; ++l) {
    const Layer& Lw = h->layer[l];
    const float* xin = sc.xl[l];
    float* xout = sc.xl[l + 1];
    GemmArgs g;
    const bool split = (h->cfg.flags & VAPX_FLAG_SPLIT_F16) != 0;   // fp32-accurate products on the f16 matrix cores
    const float *pre_att = nullptr, *pre_w = nullptr, *pre_resid = nullptr;   // long window: projection fused into the FFN block
    bool pre_ring = false;
    if (l == l_begin && !(l == 0 && qkv0_ready)) {
      g = gemm_args(sc.xn, r256, Lw.wqkv, M, 768, 256, sc.qkv, r768);
      HIPCHK(h, gemm(h, g, EPI_STORE, st));
      if (l > 0) {
        g = gemm_args(xin, r256, Lw.wkv_x, M, 512, 256, sc.kvx, r512);
        HIPCHK(h, gemm(h, g, EPI_STORE, st, /*bounded_A=*/false));   // raw residual rows
      }
    }
    if (T <= 64) {
      // fused: attention + output projection + residual + LayerNorm (+ cross-attention queries)
      AttnBlockArgs ab;
      memset(&ab, 0, sizeof ab);
      ab.q = sc.qkv; ab.k = sc.qkv + 256; ab.v = sc.qkv + 512; ab.ldq = 768; ab.ldkv = 768; ab.swap_kv = 0;
      const bool asplit = (h->cfg.flags & VAPX_FLAG_SPLIT_F16) != 0;
      ab.split = asplit ? 1 : 0;
      ab.bn = sc.bn; ab.T = T; ab.wprojf = asplit ? Lw.wprojh : Lw.wprojf; ab.resid = xin; ab.xmid = sc.xmid;
      if (l == 0 && rv && rv->ring) {   // Q|K|V and the residual straight from the per-stream rings
        ab.q = rv->ring_qkv; ab.k = rv->ring_qkv + 256; ab.v = rv->ring_qkv + 512; ab.resid = rv->ring;
        ab.ring_rot = sc.rot; ab.ids = rv->ids;
      }
      if (l == 0) { ab.ln_g = Lw.ln_ffn_g; ab.ln_b = Lw.ln_ffn_b; }
      else { ab.ln_g = Lw.ln_src_g; ab.ln_b = Lw.ln_src_b; ab.wqxf = asplit ? Lw.wqxh : Lw.wqxf; ab.qx = sc.qx; }
#ifdef VAPX_TRACE
      if (h->attn_trace && l == 1 && (size_t)B * 2 <= 16384) { ab.trace = h->attn_trace; h->attn_trace_wgs = (size_t)B * 2; }
#endif
      { ProfScope ps(h, CLS_ATTN, st); HIPCHK(h, launch_attn_block(ab, B, st)); }
#ifdef VAPX_TRACE
      ab.trace = nullptr;
#endif
      if (l > 0) {
        ab.q = sc.qx; ab.k = sc.kvx; ab.v = sc.kvx + 256; ab.ldq = 256; ab.ldkv = 512; ab.swap_kv = 1;
        ab.wprojf = asplit ? Lw.wprojxh : Lw.wprojxf; ab.resid = sc.xmid; ab.ln_g = Lw.ln_ffn_g; ab.ln_b = Lw.ln_ffn_b;
        ab.wqxf = nullptr; ab.qx = nullptr; ab.ring_rot = nullptr; ab.ids = nullptr;
        { ProfScope ps(h, CLS_ATTN, st); HIPCHK(h, launch_attn_block(ab, B, st)); }
      }
    } else if (!(h->cfg.flags & VAPX_FLAG_UNFUSED_PROJ)) {
      // long window: plain attention kernels; every projection rides in a fused flat-row block (no [rows x 256] GEMM launches)
      AttnArgs aa{sc.qkv, sc.qkv + 256, sc.qkv + 512, sc.att, sc.bn, T, 768, 768, 0};
      const bool ring0 = l == 0 && rv && rv->ring;
      if (ring0) {   // Q|K|V straight from the per-stream rings (no chronological gather)
        aa.q = rv->ring_qkv; aa.k = rv->ring_qkv + 256; aa.v = rv->ring_qkv + 512;
        aa.ring_rot = sc.rot; aa.ids = rv->ids;
      }
      const bool proj_here = xn_ready && l > 0 && Lw.wqkvp;
#ifdef VAPX_TRACE
      if (h->attn_trace && l == 1) { aa.trace = h->attn_trace; h->attn_trace_wgs = std::min<size_t>(16384, (size_t)B * 8); }
#endif
      if (proj_here) {
        AttnProjArgs ap{sc.xn, Lw.wqkvp, sc.att, sc.bn, T, 0};
        ProfScope ps(h, CLS_ATTN, st);
        HIPCHK(h, launch_attention_proj_f16x3(ap, B, st));
      } else {
        ProfScope ps(h, CLS_ATTN, st);
        HIPCHK(h, split && T <= 256 ? launch_attention_f16x3(aa, B, st) : launch_attention(aa, B, st));
      }
      pre_att = sc.att; pre_w = split ? Lw.wproj8 : Lw.wprojf; pre_resid = ring0 ? rv->ring : xin;
      pre_ring = ring0;
      if (l > 0) {
        // self half: xmid = xin + att.Wproj^T ; qx = LN_src(xmid).Wq_x^T
        FfnArgs fp;
        memset(&fp, 0, sizeof fp);
        fp.mode = 2; fp.M = M; fp.att = sc.att; fp.wprojf = split ? Lw.wproj8 : Lw.wprojf; fp.resid = xin; fp.xmid_out = sc.xmid;
        fp.ln_g = Lw.ln_src_g; fp.ln_b = Lw.ln_src_b; fp.wqkvf = split ? Lw.wqx8 : Lw.wqxf; fp.n_qkv_chunks = 1; fp.qkv = sc.qx;
        { ProfScope ps(h, CLS_FFN_PROJ, st); HIPCHK(h, split ? launch_ffn_block_f16x3(fp, st) : launch_ffn_block(fp, st)); }
        AttnArgs ax{sc.qx, sc.kvx, sc.kvx + 256, sc.att, sc.bn, T, 256, 512, 1};
        { ProfScope ps(h, CLS_ATTN, st); HIPCHK(h, split && T <= 256 ? launch_attention_f16x3(ax, B, st) : launch_attention(ax, B, st)); }
        pre_w = split ? Lw.wprojx8 : Lw.wprojxf; pre_resid = sc.xmid;
      }
    } else {
    // self attention
      AttnArgs aa{sc.qkv, sc.qkv + 256, sc.qkv + 512, sc.att, sc.bn, T, 768, 768, 0};
      { ProfScope ps(h, CLS_ATTN, st); HIPCHK(h, split && T <= 256 ? launch_attention_f16x3(aa, B, st) : launch_attention(aa, B, st)); }
      g = gemm_args(sc.att, r256, Lw.wproj, M, 256, 256, sc.xmid, r256);
      g.resid = xin; g.C2 = sc.xn;
      if (l == 0) { g.gamma = Lw.ln_ffn_g; g.beta = Lw.ln_ffn_b; }
      else { g.gamma = Lw.ln_src_g; g.beta = Lw.ln_src_b; }
      HIPCHK(h, gemm(h, g, EPI_RESID_LN, st, /*bounded_A=*/false));   // raw attention rows
      if (l > 0) {
        // cross attention: Q from LN_src(x), K/V from the OTHER channel's raw layer input
        g = gemm_args(sc.xn, r256, Lw.wq_x, M, 256, 256, sc.qx, r256);
        HIPCHK(h, gemm(h, g, EPI_STORE, st));
        AttnArgs ax{sc.qx, sc.kvx, sc.kvx + 256, sc.att, sc.bn, T, 256, 512, 1};
        { ProfScope ps(h, CLS_ATTN, st); HIPCHK(h, split && T <= 256 ? launch_attention_f16x3(ax, B, st) : launch_attention(ax, B, st)); }
        g = gemm_args(sc.att, r256, Lw.wproj_x, M, 256, 256, sc.xmid, r256);
        g.resid = sc.xmid; g.C2 = sc.xn; g.gamma = Lw.ln_ffn_g; g.beta = Lw.ln_ffn_b;
        HIPCHK(h, gemm(h, g, EPI_RESID_LN, st, /*bounded_A=*/false));
      }
    }
    // feed-forward (+ next layer's projections)
    FfnArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.xmid = sc.xmid; fa.lnf_g = Lw.ln_ffn_g; fa.lnf_b = Lw.ln_ffn_b; fa.xout = xout; fa.M = M;
    if (pre_att) { fa.mode = 1; fa.att = pre_att; fa.wprojf = pre_w; fa.resid = pre_resid; fa.xmid_out = sc.xmid; }
    if (pre_ring) { fa.resid_rot = sc.rot; fa.resid_ids = rv->ids; fa.resid_T = T; }
    fa.w0f = split ? Lw.w0h : Lw.w0f; fa.w3f = split ? Lw.w3h : Lw.w3f; fa.hid_scale = Lw.hid_scale;
    xn_ready = false;
    if (l + 1 < l_end) {
      const Layer& Ln = h->layer[l + 1];
      const float* nqkv = split ? Ln.wqkvh : Ln.wqkvf;
      fa.ln_g = Ln.ln_self_g; fa.ln_b = Ln.ln_self_b; fa.wqkvf = nqkv; fa.qkv = sc.qkv; fa.n_qkv_chunks = 3;
      fa.wkvxf = split ? Ln.wkvxh : Ln.wkvxf; fa.kvx = sc.kvx;
      if (qkv_in_attn && pre_att && Ln.wqkvp && !(prune_last && l + 1 == 3)) {   // the next layer's self-attention projects Q|K|V itself
        fa.wqkvf = nullptr; fa.n_qkv_chunks = 0; fa.xn_out = sc.xn;
        xn_ready = true;
      }
      if (prune_last && l + 1 == 3) {
        if (h->cfg.flags & VAPX_FLAG_UNFUSED_LAST_ROW) {   // the pruned layer needs K,V of every row but Q of one row only
          fa.wqkvf = nqkv + 65536; fa.n_qkv_chunks = 2;
        } else {   // fused last-row block: K / V projections are absorbed into the single query (csrc/last_block.hip);
                   // it only needs LN_self(x) of every row next to the raw rows
          fa.wqkvf = nullptr; fa.n_qkv_chunks = 0; fa.wkvxf = nullptr; fa.xn_out = sc.xn;
        }
      }
    }
#ifdef VAPX_TRACE
    if (h->ffn_trace && l == 0) {
      fa.trace = h->ffn_trace;
      h->ffn_trace_wgs = std::min<size_t>(16384, (size_t)(M + 31) / 32);
      if ((size_t)(M + 31) / 32 > 16384) fa.trace = nullptr;
    }
#endif
    if (split) { ProfScope ps(h, CLS_FFN, st); HIPCHK(h, launch_ffn_block_f16x3(fa, st)); }
    else
    { ProfScope ps(h, CLS_FFN, st); HIPCHK(h, launch_ffn_block(fa, st)); }
  }
  if (prune_last) {
    // ---- layer 3 on the newest row of every (stream, channel) only; K/V (all rows) came from the
    //      previous layer's FFN block: sc.qkv = [K | V] [M][512], sc.kvx = cross [K | V] [M][512] ----
    const Layer& Lw = h->layer[3];
    const int Ml = B * 2;
    if (!(h->cfg.flags & VAPX_FLAG_UNFUSED_LAST_ROW)) {
      LastBlockArgs lb;
      lb.x = sc.xl[3]; lb.xn = sc.xn; lb.bn = sc.bn; lb.wf = h->W("L3.last16");
      lb.ln_src_g = Lw.ln_src_g; lb.ln_src_b = Lw.ln_src_b;
      lb.ln_ffn_g = Lw.ln_ffn_g; lb.ln_ffn_b = Lw.ln_ffn_b; lb.out = sc.last[5]; lb.B = B; lb.T = T;
      ProfScope ps(h, CLS_LASTROW, st);
      HIPCHK(h, launch_last_block(lb, st));
      return VAPX_OK;
    }
    float *lx = sc.last[0], *lxn = sc.last[1], *lq = sc.last[2], *latt = sc.last[3], *lxmid = sc.last[4], *lout = sc.last[5];
    ProfScope ps(h, CLS_LASTROW, st);
    LastRowArgs lr{sc.xl[3], sc.bn, lx, lxn, Lw.ln_self_g, Lw.ln_self_b, B, T};
    HIPCHK(h, launch_gather_last_ln(lr, st));
    GemmArgs g = gemm_args(lxn, r256, Lw.wqkv, Ml, 256, 256, lq, r256);            // Q rows of Wqkv
    HIPCHK(h, launch_gemm_f32(g, EPI_STORE, 0, st));
    AttnArgs aa{lq, sc.qkv, sc.qkv + 256, latt, sc.bn, T, 256, 512, 0};
    HIPCHK(h, launch_attention_last(aa, B, st));
    g = gemm_args(latt, r256, Lw.wproj, Ml, 256, 256, lxmid, r256);
    g.resid = lx; g.C2 = lxn; g.gamma = Lw.ln_src_g; g.beta = Lw.ln_src_b;
    HIPCHK(h, launch_gemm_f32(g, EPI_RESID_LN, 0, st));
    g = gemm_args(lxn, r256, Lw.wq_x, Ml, 256, 256, lq, r256);
    HIPCHK(h, launch_gemm_f32(g, EPI_STORE, 0, st));
    AttnArgs ax{lq, sc.kvx, sc.kvx + 256, latt, sc.bn, T, 256, 512, 1};
    HIPCHK(h, launch_attention_last(ax, B, st));
    g = gemm_args(latt, r256, Lw.wproj_x, Ml, 256, 256, lxmid, r256);
    g.resid = lxmid; g.C2 = lxn; g.gamma = Lw.ln_ffn_g; g.beta = Lw.ln_ffn_b;
    HIPCHK(h, launch_gemm_f32(g, EPI_RESID_LN, 0, st));
    // FFN on 2B rows: two plain GEMMs spread over more workgroups than one fused 32-row block would
    g = gemm_args(lxn, r256, Lw.w0, Ml, 768, 256, sc.lffn, r768);
    HIPCHK(h, launch_gemm_f32(g, EPI_GELU, 0, st));
    g = gemm_args(sc.lffn, r768, Lw.w3, Ml, 256, 768, lout, r256);
    g.resid = lxmid;
    HIPCHK(h, launch_gemm_f32(g, EPI_RESID, 0, st));
  }
  return VAPX_OK;
}

// nod variant: p_bc = sigmoid(bc_head(comb)) for EVERY row of the window (vap_nod_main.py:276 indexes the
// batch dim, so all n rows are emitted); one wave per (stream, row); written over the (unused in
// nod mode) logits slots of the output row.
__global__ void pbc_rows_kernel(const float* comb, const float* w, const float* bias, const int* bn, float* out,
                                int B, int T, int out_stride) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * T) return;
  const int b = (int)(row / T), t = (int)(row - (long)b * T);
  float v = 0.f;
  if (t < bn[b]) {
    f32x4 x = *(const f32x4*)(comb + row * 256 + lane * 4), ww = *(const f32x4*)(w + lane * 4);
    float d = wave_sum(x[0] * ww[0] + x[1] * ww[1] + x[2] * ww[2] + x[3] * ww[3]);
    v = 1.0f / (1.0f + expf(-(d + bias[0])));
  }
  if (lane == 0 && t < 256) out[(long)b * out_stride + VAPX_OUT_LOGITS + t] = v;
}

__global__ void fill_int_kernel(int* p, int v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// stream-ordered reset of up to 16 stream slots: LSTM h/c, carry and window fill back to zero (the rings need no
// clearing: rows beyond frames_seen are never read).  One workgroup per slot.
struct ResetList { int n; int ids[16]; int* fs[4]; };   // fs[0] = this engine's frames_seen, fs[1..] = trunk followers'
// ids[k] < 0 encodes "carry only" for stream -ids[k] - 1 (what a reconnect does in the reference, vap_main.py:368-369)
__global__ void reset_streams_kernel(ResetList r, float* h_state, float* c_state, float* carry) {
  const int raw = r.ids[blockIdx.x];
  const bool carry_only = raw < 0;
  const int sid = carry_only ? -raw - 1 : raw;
  for (int i = threadIdx.x; i < 2 * VAPX_PAD; i += blockDim.x) carry[(long)sid * 2 * VAPX_PAD + i] = 0.f;
  if (carry_only) return;
  for (int i = threadIdx.x; i < 512; i += blockDim.x) { h_state[(long)sid * 512 + i] = 0.f; c_state[(long)sid * 512 + i] = 0.f; }
  if (threadIdx.x < 4 && r.fs[threadIdx.x]) r.fs[threadIdx.x][sid] = 0;
}
__global__ void add_kernel(float* o, const float* a, const float* b, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}
// ---- level-1 head surface (vap_main.py:290-307 calls these on tensors of ANY row count): one wave per row ----
// y[r] = x[r] . w + b                                   (va_classifier: Linear(256, 1), vap_main.py:142,292-293)
__global__ void rowdot_kernel(const float* x, const float* w, const float* bias, float* y, long rows) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  f32x4 xv = *(const f32x4*)(x + r * 256 + lane * 4), wv = *(const f32x4*)(w + lane * 4);
  const float d = wave_sum(xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3]);
  if (lane == 0) y[r] = d + bias[0];
}
// y[r][o] = x[r] . w[o] + b[o], o < nout <= 4   (bc_head: Linear(256, 3) / Linear(256, 1), nod_head: Linear(256, 4);
// vap_realtime/vap_models.py:220,328-329, applied to out["x"] of ANY row count at vap_realtime/model.py:197,217-218)
__global__ void rowheads_kernel(const float* x, const float* w, const float* bias, float* y, long rows, int nout) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const f32x4 xv = *(const f32x4*)(x + r * 256 + lane * 4);
  for (int o = 0; o < nout; ++o) {
    const f32x4 wv = *(const f32x4*)(w + o * 256 + lane * 4);
    const float d = wave_sum(xv[0] * wv[0] + xv[1] * wv[1] + xv[2] * wv[2] + xv[3] * wv[3]);
    if (lane == 0) y[r * nout + o] = d + bias[o];
  }
}
// softmax over the 256 classes of a row (probs = logits.softmax(-1), vap_main.py:295)
__global__ void softmax256_kernel(const float* x, float* y, long rows) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  f32x4 v = *(const f32x4*)(x + r * 256 + lane * 4);
  const float mx = wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  f32x4 e = {expf(v[0] - mx), expf(v[1] - mx), expf(v[2] - mx), expf(v[3] - mx)};
  const float inv = 1.0f / wave_sum(e[0] + e[1] + e[2] + e[3]);
  *(f32x4*)(y + r * 256 + lane * 4) = e * inv;
}
// ObjectiveVAP.probs_next_speaker_aggregate (objective.py:186-206): class i <-> 8 bits, speaker c owns bits 4c..4c+3;
// p[c] = sum_i probs[i] * #set bits of i among bins [from, to] of speaker c;  p /= p0 + p1 + 1e-5
__global__ void aggregate_kernel(const float* probs, float* out, long rows, int from_bin, int to_bin) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  f32x4 p = *(const f32x4*)(probs + r * 256 + lane * 4);
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = lane * 4 + u;
    int c0 = 0, c1 = 0;
    for (int b = from_bin; b <= to_bin; ++b) { c0 += (i >> b) & 1; c1 += (i >> (4 + b)) & 1; }
    a0 += p[u] * (float)c0;
    a1 += p[u] * (float)c1;
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1);
  if (lane == 0) { const float d = a0 + a1 + 1e-5f; out[r * 2] = a0 / d; out[r * 2 + 1] = a1 / d; }
}

// compact [B*2][T][256] scratch rows (t < rows) into [B*2][rows][256]
__global__ void compact_rows_kernel(float* dst, const float* src, int T, int rows, long nrows_out) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
  if (i >= nrows_out * 64) return;
  long r = i >> 6;
  int q = (int)(i & 63);
  long bc = r / rows;
  int t = (int)(r - bc * rows);
  ((f32x4*)dst)[i] = ((const f32x4*)src)[(bc * T + t) * 64 + q];
}

int upload_ids(vapx_engine* h, int n, const int32_t* ids, int flags, hipStream_t st, const int** out) {
  if (!ids) { *out = nullptr; return VAPX_OK; }
  if (flags & VAPX_IDS_DEVICE) { *out = ids; return VAPX_OK; }
  if (++h->id_gen == 0) { std::fill(h->id_stamp.begin(), h->id_stamp.end(), 0u); h->id_gen = 1; }
  for (int i = 0; i < n; ++i) {
    if (ids[i] < 0 || ids[i] >= h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "stream id %d out of range [0,%d)", ids[i], h->cfg.max_streams);
    // two batch slots on one stream would race on its ring slot, LSTM state, carry and frame counter
    if (h->id_stamp[ids[i]] == h->id_gen) return fail(h, VAPX_E_INVAL, "stream id %d appears twice in one step (batch slot %d)", ids[i], i);
    h->id_stamp[ids[i]] = h->id_gen;
  }
  HIPCHK(h, hipEventSynchronize(h->ids_evt));  // previous async copy out of the pinned buffer is done
  memcpy(h->ids_pinned, ids, n * sizeof(int));
  HIPCHK(h, hipMemcpyAsync(h->ids_dev, h->ids_pinned, n * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipEventRecord(h->ids_evt, st));
  *out = h->ids_dev;
  return VAPX_OK;
}


// apply the vapx_reset_stream requests collected since the last step, ordered on `st` (no host or device sync)
int flush_resets(vapx_engine* h, hipStream_t st) {
  if (h->pending_resets.empty()) return VAPX_OK;
  ResetList r;
  memset(&r, 0, sizeof r);
  r.fs[0] = h->frames_seen;
  size_t nf = 0;
  for (vapx_engine* f : h->followers)
    if (nf + 1 < 4) r.fs[++nf] = f->frames_seen;   // a fourth and later follower gets its own tiny launches below
  for (size_t i = 0; i < h->pending_resets.size(); i += 16) {
    r.n = (int)std::min<size_t>(16, h->pending_resets.size() - i);
    for (int k = 0; k < r.n; ++k) r.ids[k] = h->pending_resets[i + k];
    hipLaunchKernelGGL(reset_streams_kernel, dim3(r.n), dim3(256), 0, st, r, h->h_state, h->c_state, h->carry);
    for (size_t f = 3; f < h->followers.size(); ++f)
      for (int k = 0; k < r.n; ++k)
        if (r.ids[k] >= 0) hipLaunchKernelGGL(fill_int_kernel, dim3(1), dim3(64), 0, st, h->followers[f]->frames_seen + r.ids[k], 0, 1);
  }
  HIPCHK(h, hipGetLastError());
  h->pending_resets.clear();
  return VAPX_OK;
}

// quiesce the device and apply queued resets (state import / export, peeks)
int quiesce(vapx_engine* h) {
  (void)hipGetLastError();   // a stale error of an earlier, unrelated HIP call (this library's or anyone's) is not this call's
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipDeviceSynchronize());
  vapx_engine* lead = h->trunk ? h->trunk : h;
  if (!lead->pending_resets.empty()) {
    int rc = flush_resets(lead, nullptr);
    if (rc) return rc;
    HIPCHK(h, hipDeviceSynchronize());
  }
  h->deferred_pending = false;
  return VAPX_OK;
}

// make `st` wait for overlap groups a VAPX_DEFER_JOIN step left running
int join_deferred(vapx_engine* h, hipStream_t st) {
  if (!h->deferred_pending) return VAPX_OK;
  for (int g = 0; g < h->last_G; ++g) HIPCHK(h, hipStreamWaitEvent(st, h->gdone[g], 0));
  h->deferred_pending = false;
  return VAPX_OK;
}

// Combinator on ALL rows: comb = gelu(LN(a.Wa^T)) + gelu(LN(b.Wb^T)), shared LN (modules.py:449-464).
// Tower rows of channel c of stream b sit at ((b*2+c)*T + t).  Result in sc.xmid as [n][T][256].
int run_combinator_all_rows(vapx_engine* h, const Scratch& sc, int n, hipStream_t st) {
  const int T = h->T, M = n * T;
  for (int c = 0; c < 2; ++c) {
    RowMap am{(long)2 * T * 256, 256, T};
    GemmArgs g = gemm_args(sc.xl[4] + (long)c * T * 256, am, h->W(c ? "comb.wb" : "comb.wa"), M, 256, 256,
                           c ? sc.qx : sc.att, contiguous_rows(256));
    g.gamma = h->W("comb.g"); g.beta = h->W("comb.b");
    HIPCHK(h, gemm(h, g, EPI_BIAS_LN_GELU, st, /*bounded_A=*/false));   // raw last-layer rows
  }
  const long tot = (long)M * 256;
  hipLaunchKernelGGL(add_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, sc.xmid, sc.att, sc.qx, tot);
  return VAPX_OK;
}

// one sub-batch of a tick: encoder -> ring -> transformer -> heads.  `b0` is the offset of the
// group inside the caller's batch (identity stream ids when ids == nullptr start at b0).
int step_group(vapx_engine* h, const Scratch& sc_own, int nb, int b0, const int* ids, const float* audio, int spc,
               float* out, hipStream_t st, const Scratch* lead = nullptr) {
  Scratch sc = sc_own;
  if (lead) { sc.bn = lead->bn; sc.bhead = lead->bhead; }   // window fill / ring slot: decided by the leader's conv0
  // identity ids: kernels index state by batch slot, so advance the state bases by b0 streams
  const size_t s0 = ids ? 0 : (size_t)b0;
  const StateView sv{h->ring + s0 * 2 * h->T * 256, h->ring_qkv + s0 * 2 * h->T * 768, h->h_state + s0 * 512, h->c_state + s0 * 512,
                     h->carry + s0 * 2 * VAPX_PAD, h->frames_seen + s0};
  int rc = VAPX_OK;
  if (!lead) {
    rc = run_encoder(h, sc, sv, nb, ids, audio, spc, true, st);
    if (rc) return rc;
  } else {
    // this weight set's own downsample on the shared LSTM outputs: e = gelu(LN(Conv1d_K(lstm_out))), en = LN0(e)
    GemmArgs g = gemm_args(lead->lstm_out, contiguous_rows(h->ncpc * 256), h->W("down.w"), nb * 2, 256, h->ncpc * 256, sc.e,
                           contiguous_rows(256));
    g.bias = h->W("down.b"); g.gamma = h->W("down.g"); g.beta = h->W("down.beta");
    HIPCHK(h, gemm(h, g, EPI_BIAS_LN_GELU, st));
    HIPCHK(h, launch_ln_rows(sc.e, sc.en, h->layer[0].ln_self_g, h->layer[0].ln_self_b, nb * 2, st));
  }
  {  // layer-0 Q|K|V of the NEW row only: they are per-row functions of the embedding, so the other
     // rows' values are cached next to the ring (exact; saves the [rows x 768] GEMM every tick)
    GemmArgs g = gemm_args(sc.en, contiguous_rows(256), h->layer[0].wqkv, nb * 2, 768, 256, sc.qkv_new, contiguous_rows(768));
    g.W16 = h->W("L0.wqkv16");
    HIPCHK(h, gemm(h, g, EPI_STORE, st));
  }
  GatherArgs ga;
  memset(&ga, 0, sizeof ga);
  ga.ring = sv.ring; ga.ring_qkv = sv.ring_qkv; ga.qkv_new = sc.qkv_new; ga.qkv = sc.qkv;
  ga.e = sc.e; ga.xin = nullptr; ga.ids = ids; ga.bn = sc.bn; ga.bhead = sc.bhead;
  ga.x0 = sc.xl[0]; ga.xn = sc.xn; ga.gamma = h->layer[0].ln_self_g; ga.beta = h->layer[0].ln_self_b;
  ga.B = nb; ga.T = h->T; ga.rows_in = 0;
  // layer 0 reads the rings in place (no chronological copy): always for the fused short-window block, and for long windows
  // when they run the fused long-window chain (the GEMM-chain variants need x0 as a plain buffer)
  const bool ring_direct = !(h->cfg.flags & VAPX_FLAG_MATERIALIZE_X0) &&
                           (h->T <= 64 || !(h->cfg.flags & VAPX_FLAG_UNFUSED_PROJ));
  ga.rot = sc.rot;
  if (ring_direct) { ProfScope ps(h, CLS_GATHER, st); HIPCHK(h, launch_ring_append(ga, st)); }
  else { ProfScope ps(h, CLS_GATHER, st); HIPCHK(h, launch_gather_ln(ga, st)); }
  // the nod variant emits p_bc for every row of the window, which needs the whole last layer
  const bool prune = !(h->cfg.flags & VAPX_FLAG_FULL_LAST_LAYER) && h->cfg.mode != VAPX_MODE_NOD;
  RingView rv{ring_direct ? sv.ring : nullptr, sv.ring_qkv, ids};
  rc = run_layers(h, sc, nb, st, 0, 4, prune, /*qkv0_ready=*/true, ring_direct ? &rv : nullptr);
  if (rc) return rc;
  HeadArgs ha;
  ha.x = prune ? sc.last[5] : sc.xl[4]; ha.x_last_only = prune ? 1 : 0; ha.o = sc.xl[1]; ha.e = sc.e; ha.bn = sc.bn; ha.ids = ids; ha.frames_seen = sv.frames_seen;
  ha.waT = h->W("comb.waT"); ha.wbT = h->W("comb.wbT"); ha.cg = h->W("comb.g"); ha.cb = h->W("comb.b");
  ha.hwT = h->W("head.wT"); ha.hb = h->W("head.b"); ha.vw = h->W("vad.w"); ha.vb = h->W("vad.b");
  ha.aw = h->W("aux.w"); ha.ab = h->W("aux.b"); ha.out = out; ha.B = nb; ha.T = h->T; ha.mode = h->cfg.mode;
  ha.out_stride = VAPX_OUT_STRIDE;
  { ProfScope ps(h, CLS_HEAD, st); HIPCHK(h, launch_head(ha, st)); }
  if (h->cfg.mode == VAPX_MODE_NOD) {
    rc = run_combinator_all_rows(h, sc, nb, st);
    if (rc) return rc;
    const long rows = (long)nb * h->T;
    hipLaunchKernelGGL(pbc_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, sc.xmid, h->W("aux.w") + 4 * 256,
                       h->W("aux.b") + 4, sc.bn, out, nb, h->T, VAPX_OUT_STRIDE);
    HIPCHK(h, hipGetLastError());
  }
  return VAPX_OK;
}

}  // namespace

extern "C" {

int32_t vapx_abi_version(void) { return VAPX_ABI_VERSION; }

size_t vapx_blob_floats(int32_t frame_hz) {
  if (!rate_ok(frame_hz)) return 0;
  int hop, L, P[5], ncpc;
  geometry(frame_hz, &hop, &L, P, &ncpc);
  size_t n = 0;
  const vapx_layout::Entry* lay = vapx_layout::layout_for_K(ncpc, &n);
  if (!lay) return 0;
  return lay[n - 1].off;  // "__total__"
}

const char* vapx_last_error(vapx_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

void vapx_destroy(vapx_handle h) {
  if (!h) return;
  if (h->trunk) {
    auto& fl = h->trunk->followers;
    for (size_t i = 0; i < fl.size(); ++i)
      if (fl[i] == h) { fl.erase(fl.begin() + i); break; }
  }
  for (vapx_engine* f : h->followers) { f->trunk = nullptr; f->orphaned = true; }
  (void)hipSetDevice(h->cfg.device_id);
  (void)hipDeviceSynchronize();
#ifdef VAPX_TRACE
  auto dump_trace = [&](unsigned long long* buf, size_t wgs, const std::string& path) {   // stamps of the last traced launch: [wgs][32] u64
    if (!buf) return;
    std::vector<unsigned long long> host(wgs * 32);
    if (!host.empty() && hipMemcpy(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
      if (FILE* f = fopen(path.c_str(), "wb")) { fwrite(host.data(), 8, host.size(), f); fclose(f); }
    }
    dfree(buf);
  };
  dump_trace(h->ffn_trace, h->ffn_trace_wgs, h->ffn_trace_path);
  dump_trace(h->attn_trace, h->attn_trace_wgs, h->attn_trace_path);
#endif
  float* fp[] = {h->w, h->ring, h->ring_qkv, h->h_state, h->c_state, h->carry, h->audio_dev, h->out_dev, h->sc.h0, h->sc.h1, h->sc.h2, h->sc.h3,
                 h->sc.z, h->sc.lstm_out, h->sc.e, h->sc.xl[0], h->sc.xl[1], h->sc.xl[2], h->sc.xl[3], h->sc.xl[4], h->sc.xn, h->sc.xmid, h->sc.att,
                 h->sc.qkv, h->sc.qx, h->sc.kvx, h->sc.ffn, h->sc.gx, h->sc.last[0], h->sc.last[1], h->sc.last[2],
                 h->sc.last[3], h->sc.last[4], h->sc.last[5], h->sc.en, h->sc.qkv_new, h->sc.lffn};
  for (float* p : fp) dfree(p);
  int* ip[] = {h->frames_seen, h->ids_dev, h->sc.bn, h->sc.bhead, h->sc.rot};
  for (int* p : ip) dfree(p);
  if (h->out_pinned) (void)hipHostFree(h->out_pinned);
  if (h->ids_pinned) (void)hipHostFree(h->ids_pinned);
  if (h->audio_pinned) (void)hipHostFree(h->audio_pinned);
  if (h->ids_evt) (void)hipEventDestroy(h->ids_evt);
  if (h->audio_evt) (void)hipEventDestroy(h->audio_evt);
  if (h->gstart) (void)hipEventDestroy(h->gstart);
  for (int g = 0; g < vapx_engine::kMaxGroups; ++g) {
    if (h->gdone[g]) (void)hipEventDestroy(h->gdone[g]);
    if (h->gstream[g]) (void)hipStreamDestroy(h->gstream[g]);
  }
  for (auto& r : h->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : h->prof_pool) (void)hipEventDestroy(e);
  delete h;
}

int vapx_create(const vapx_config* cfg, const float* blob, size_t n_floats, vapx_handle* out) {
  if (!cfg || !blob || !out) return fail(nullptr, VAPX_E_INVAL, "null argument");
  if (cfg->struct_size != (int32_t)sizeof(vapx_config)) return fail(nullptr, VAPX_E_INVAL, "vapx_config.struct_size mismatch");
  if (!rate_ok(cfg->frame_hz)) return fail(nullptr, VAPX_E_INVAL, "frame_hz must be 5, 10, 20 or 50");
  if (cfg->ctx_frames < 1 || cfg->ctx_frames > 512) return fail(nullptr, VAPX_E_INVAL, "ctx_frames must be in [1,512]");
  if (cfg->max_streams < 1 || cfg->max_batch < 1 || cfg->max_batch > cfg->max_streams)
    return fail(nullptr, VAPX_E_INVAL, "need 1 <= max_batch <= max_streams");
  if (cfg->mode < 0 || cfg->mode > 2) return fail(nullptr, VAPX_E_INVAL, "bad mode");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(nullptr, VAPX_E_NODEVICE, "no HIP device visible");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, VAPX_E_INVAL, "device_id out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess) return fail(nullptr, VAPX_E_HIP, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, VAPX_E_NODEVICE, "device %d is %s; libvapx is built for gfx950 only", cfg->device_id, prop.gcnArchName);

  vapx_engine* h = new vapx_engine();
  h->cfg = *cfg;
  geometry(cfg->frame_hz, &h->hop, &h->L, h->P, &h->ncpc);
  h->T = cfg->ctx_frames;
  h->K = h->ncpc;
  h->lay = vapx_layout::layout_for_K(h->ncpc, &h->lay_n);
  const size_t need = vapx_blob_floats(cfg->frame_hz);
  if (n_floats != need) {
    int rc = fail(nullptr, VAPX_E_INVAL, "weights blob has %zu floats, expected %zu for %d Hz", n_floats, need, cfg->frame_hz);
    delete h;
    return rc;
  }
#define CR(expr)                                                                                   \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      int rc = fail(nullptr, _e == hipErrorOutOfMemory ? VAPX_E_NOMEM : VAPX_E_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
      (void)hipGetLastError();   /* do not leave the error for the next launch check to find */     \
      vapx_destroy(h);                                                                             \
      return rc;                                                                                   \
    }                                                                                              \
  } while (0)
  CR(hipSetDevice(cfg->device_id));
  CR(dalloc(&h->w, need, false));
  CR(hipMemcpy(h->w, blob, need * sizeof(float), hipMemcpyHostToDevice));
  h->w_floats = need;
  char nm[64];
  for (int l = 0; l < 4; ++l) {
    Layer& Lw = h->layer[l];
    auto get = [&](const char* suffix) { snprintf(nm, sizeof nm, "L%d.%s", l, suffix); return h->W(nm); };
    Lw.ln_self_g = get("ln_self.g"); Lw.ln_self_b = get("ln_self.b"); Lw.wqkv = get("wqkv"); Lw.wproj = get("wproj");
    Lw.ln_src_g = get("ln_src.g"); Lw.ln_src_b = get("ln_src.b"); Lw.wq_x = get("wq_x"); Lw.wkv_x = get("wkv_x");
    Lw.wproj_x = get("wproj_x"); Lw.ln_ffn_g = get("ln_ffn.g"); Lw.ln_ffn_b = get("ln_ffn.b"); Lw.w0 = get("w0"); Lw.w3 = get("w3");
    Lw.w0f = get("w0f"); Lw.w3f = get("w3f"); Lw.wqkvf = get("wqkvf"); Lw.wkvxf = get("wkvxf");
    Lw.wprojf = get("wprojf"); Lw.wqxf = get("wqxf"); Lw.wprojxf = get("wprojxf");
    Lw.w0h = get("w0h"); Lw.w3h = get("w3h"); Lw.wqkvh = get("wqkvh"); Lw.wkvxh = get("wkvxh");
    Lw.wprojh = get("wprojh"); Lw.wqxh = get("wqxh"); Lw.wprojxh = get("wprojxh");
    Lw.wproj8 = get("wproj8"); Lw.wqx8 = get("wqx8"); Lw.wprojx8 = get("wprojx8");
    Lw.wqkvp = get("wqkvp");
  }
  if (cfg->flags & VAPX_FLAG_SPLIT_F16) {
    // Static guarantees of the split-precision path, from the weights alone (host copy of the blob):
    //  * every weight w satisfies |2^8 w| < 65504 (its f16 hi part is finite);
    //  * the GELU hidden row h = LN_ffn(x) . W0^T obeys |h_j| <= sum_k |W0[j][k]| (16 |gamma_k| + |beta_k|) because a LayerNorm output is
    //    bounded by 16 |gamma| + |beta|: a layer whose bound reaches 2^15 gets a power-of-two down-scale of the hidden operand.
    // every weight that reaches the split-precision GEMM (gemm(): conv1-4, LSTM input projection, downsample, Combinator) is stored /
    // converted as 2^8 w in f16 as well
    for (const char* nm : {"conv1.w", "conv2.w", "conv3.w", "conv4.w", "lstm.wih", "down.w", "comb.wa", "comb.wb"})
      for (size_t i = 0; i < h->lay_n; ++i)
        if (!strcmp(h->lay[i].name, nm))
          for (size_t k = 0; k < h->lay[i].n; ++k)
            if (!(fabsf(blob[h->lay[i].off + k]) < 255.0f)) {
              int rc = fail(nullptr, VAPX_E_INVAL, "VAPX_FLAG_SPLIT_F16: a weight of %s has |w| >= 255 (or is not finite); use the fp32 path", nm);
              vapx_destroy(h);
              return rc;
            }
    for (int l = 0; l < 4; ++l) {
      Layer& Lw = h->layer[l];
      auto host = [&](const float* dev) { return blob + (dev - h->w); };
      const struct { const float* p; size_t n; } mats[] = {{Lw.w0, 768 * 256}, {Lw.w3, 256 * 768}, {Lw.wqkv, 768 * 256}, {Lw.wproj, 65536},
                                                          {Lw.wq_x, 65536}, {Lw.wkv_x, 512 * 256}, {Lw.wproj_x, 65536}};
      for (const auto& m : mats) {
        if (!m.p) continue;
        const float* q = host(m.p);
        for (size_t i = 0; i < m.n; ++i)
          if (!(fabsf(q[i]) < 255.0f)) {
            int rc = fail(nullptr, VAPX_E_INVAL, "VAPX_FLAG_SPLIT_F16: a transformer weight of layer %d has |w| >= 255 (or is not finite); use the fp32 path", l);
            vapx_destroy(h);
            return rc;
          }
      }
      const float *w0 = host(Lw.w0), *gm = host(Lw.ln_ffn_g), *bt = host(Lw.ln_ffn_b);
      double worst = 0.0;
      for (int j = 0; j < 768; ++j) {
        double b = 0.0;
        for (int k = 0; k < 256; ++k) b += fabs((double)w0[j * 256 + k]) * (16.0 * fabs((double)gm[k]) + fabs((double)bt[k]));
        worst = std::max(worst, b);
      }
      float hs = 1.0f;
      while ((double)hs * worst >= 32768.0 && hs > 1e-30f) hs *= 0.5f;
      Lw.hid_scale = hs;
    }
  }
  const size_t S = cfg->max_streams, B = cfg->max_batch, T = h->T;
  const int* P = h->P;
  CR(dalloc(&h->ring, S * 2 * T * 256));
  CR(dalloc(&h->ring_qkv, S * 2 * T * 768));   // layer-0 Q|K|V cache, one entry per ring row
  CR(dalloc(&h->h_state, S * 2 * 256));
  CR(dalloc(&h->c_state, S * 2 * 256));
  CR(dalloc(&h->carry, S * 2 * VAPX_PAD));
  CR(dalloc(&h->frames_seen, S));
  CR(dalloc(&h->audio_dev, B * 2 * h->L));
  CR(dalloc(&h->out_dev, B * VAPX_OUT_STRIDE));
  CR(dalloc(&h->ids_dev, B));
  CR(dalloc(&h->sc.bn, B));
  CR(dalloc(&h->sc.bhead, B));
  CR(dalloc(&h->sc.rot, B));
  CR(dalloc(&h->sc.h0, B * 2 * (P[0] + 4) * 256));  // guard rows stay zero forever
  CR(dalloc(&h->sc.h1, B * 2 * (P[1] + 2) * 256));
  CR(dalloc(&h->sc.h2, B * 2 * (P[2] + 2) * 256));
  CR(dalloc(&h->sc.h3, B * 2 * (P[3] + 2) * 256));
  CR(dalloc(&h->sc.z, B * 2 * h->ncpc * 256));
  CR(dalloc(&h->sc.lstm_out, B * 2 * h->ncpc * 256));
  CR(dalloc(&h->sc.gx, B * 2 * h->ncpc * 1024));
  CR(dalloc(&h->sc.e, B * 2 * 256));
  const size_t rows = B * 2 * T;
  for (int i = 0; i < 5; ++i) CR(dalloc(&h->sc.xl[i], rows * 256));
  CR(dalloc(&h->sc.xn, rows * 256));
  CR(dalloc(&h->sc.xmid, rows * 256));
  CR(dalloc(&h->sc.att, rows * 256));
  CR(dalloc(&h->sc.qkv, rows * 768));
  CR(dalloc(&h->sc.qx, rows * 256));
  CR(dalloc(&h->sc.kvx, rows * 512));
  h->sc.ffn = nullptr;  // FFN hidden activations never leave the fused FFN block
  for (int i = 0; i < 6; ++i) CR(dalloc(&h->sc.last[i], B * 2 * 256));
  CR(dalloc(&h->sc.en, B * 2 * 256));
  CR(dalloc(&h->sc.qkv_new, B * 2 * 768));
  CR(dalloc(&h->sc.lffn, B * 2 * 768));
  CR(hipHostMalloc((void**)&h->out_pinned, B * VAPX_OUT_STRIDE * sizeof(float), hipHostMallocDefault));
  CR(hipHostMalloc((void**)&h->ids_pinned, B * sizeof(int), hipHostMallocDefault));
  CR(hipHostMalloc((void**)&h->audio_pinned, B * 2 * h->L * sizeof(float), hipHostMallocDefault));
  h->id_stamp.assign(S, 0u);
  if (getenv("VAPX_POISON_SCRATCH")) {
    CR(hipMemset(h->ring, 0xFF, S * 2 * T * 256 * sizeof(float)));          // rows beyond frames_seen are never read: prove it
    CR(hipMemset(h->ring_qkv, 0xFF, S * 2 * T * 768 * sizeof(float)));
    auto add = [&](float* q, size_t n) { h->poison.push_back({q, n * sizeof(float)}); };
    add(h->sc.z, B * 2 * h->ncpc * 256); add(h->sc.lstm_out, B * 2 * h->ncpc * 256); add(h->sc.gx, B * 2 * h->ncpc * 1024); add(h->sc.e, B * 2 * 256);
    for (int i = 0; i < 5; ++i) add(h->sc.xl[i], rows * 256);
    add(h->sc.xn, rows * 256); add(h->sc.xmid, rows * 256); add(h->sc.att, rows * 256); add(h->sc.qkv, rows * 768); add(h->sc.qx, rows * 256);
    add(h->sc.kvx, rows * 512);
    for (int i = 0; i < 6; ++i) add(h->sc.last[i], B * 2 * 256);
    add(h->sc.en, B * 2 * 256); add(h->sc.qkv_new, B * 2 * 768); add(h->sc.lffn, B * 2 * 768); add(h->out_dev, B * VAPX_OUT_STRIDE);
  }
#ifdef VAPX_TRACE
  if (const char* ev = getenv("VAPX_FFN_TRACE")) {
    h->ffn_trace_path = ev;
    CR(dalloc(&h->ffn_trace, (size_t)16384 * 32));
  }
  if (const char* ev = getenv("VAPX_ATTN_TRACE")) {
    h->attn_trace_path = ev;
    CR(dalloc(&h->attn_trace, (size_t)16384 * 32));
  }
#endif
  h->n_groups = cfg->flags & 0xF;
  if (h->n_groups == 0) h->n_groups = 1;   // measured: no gain at 256 streams, +2 % at 4096 with 2
  if (h->n_groups > vapx_engine::kMaxGroups) h->n_groups = vapx_engine::kMaxGroups;
  for (int g = 0; g < h->n_groups; ++g) {
    CR(hipStreamCreateWithFlags(&h->gstream[g], hipStreamNonBlocking));
    CR(hipEventCreateWithFlags(&h->gdone[g], hipEventDisableTiming));
  }
  CR(hipEventCreateWithFlags(&h->gstart, hipEventDisableTiming));
  CR(hipEventCreateWithFlags(&h->ids_evt, hipEventDisableTiming));
  CR(hipEventRecord(h->ids_evt, nullptr));
  CR(hipEventCreateWithFlags(&h->audio_evt, hipEventDisableTiming));
  CR(hipEventRecord(h->audio_evt, nullptr));
  CR(hipDeviceSynchronize());
#undef CR
  *out = h;
  return VAPX_OK;
}

int vapx_step(vapx_handle h, int32_t n, const int32_t* stream_ids, const float* audio, int32_t spc, float* out,
              int32_t flags, void* hip_stream) {
  if (!h) return VAPX_E_INVAL;
  if (n < 1 || n > h->cfg.max_batch) return fail(h, VAPX_E_RANGE, "n=%d outside [1,%d]", n, h->cfg.max_batch);
  if (!out) return fail(h, VAPX_E_INVAL, "null out");
  vapx_engine* lead = h->trunk;
  if (h->orphaned) return fail(h, VAPX_E_INVAL, "the trunk leader of this engine was destroyed");
  if (lead) {
    if (audio) return fail(h, VAPX_E_INVAL, "a trunk follower takes no audio (pass NULL): it consumes its leader's encoder output");
    if (lead->tick == 0 || lead->tick == h->followed_tick)
      return fail(h, VAPX_E_INVAL, "step the trunk leader first: no new encoder output since this follower's last step");
    if (n != lead->last_B) return fail(h, VAPX_E_INVAL, "n=%d differs from the leader's latest step (%d streams)", n, lead->last_B);
  } else {
    if (!audio) return fail(h, VAPX_E_INVAL, "null audio");
    if (spc != h->hop && spc != h->L) return fail(h, VAPX_E_INVAL, "samples_per_ch must be %d (hop) or %d (full frame)", h->hop, h->L);
  }
  if (!stream_ids && n > h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "n exceeds max_streams");
  hipStream_t st = (hipStream_t)hip_stream;
  (void)hipGetLastError();   // a stale error of an earlier, unrelated HIP call (this library's or anyone's) is not this call's
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  // split the batch into groups on separate HIP streams: streams are independent, and a second
  // group's kernels fill the prologue / epilogue / tail bubbles of the first group's kernels
  int G = h->n_groups;
  while (G > 1 && n / G < 32) --G;
  // free-running groups are only safe when nothing of this step is staged through engine-owned buffers on `st`
  // (host audio / host ids would be overwritten under a still-running group of the previous tick)
  const bool all_device = (flags & VAPX_OUT_DEVICE) && (lead || (flags & VAPX_AUDIO_DEVICE)) && (!stream_ids || (flags & VAPX_IDS_DEVICE));
  const bool defer_join = G > 1 && (flags & VAPX_DEFER_JOIN) && all_device;
  int rc = VAPX_OK;
  // a different batch split re-slices the shared scratch, and a reset touches state a running group may still use:
  // in both cases the previous tick's groups are joined first
  if (h->deferred_pending && (!defer_join || n != h->last_B || G != h->last_G || !h->pending_resets.empty())) {
    rc = join_deferred(h, st);
    if (rc) return rc;
  }
  if (!h->poison.empty()) {
    rc = join_deferred(h, st);
    if (rc) return rc;
    for (auto& pb : h->poison) HIPCHK(h, hipMemsetAsync(pb.first, 0xFF, pb.second, st));
  }
  if (!lead) { rc = flush_resets(h, st); if (rc) return rc; }
  const int* ids = nullptr;
  if (lead) ids = lead->last_ids;   // same streams, same order as the leader's step (stream_ids is ignored)
  else rc = upload_ids(h, n, stream_ids, flags, st, &ids);
  if (rc) return rc;
  const float* ad = audio;
  if (!lead && !(flags & VAPX_AUDIO_DEVICE)) {
    const size_t bytes = (size_t)n * 2 * spc * sizeof(float);
    const float* src = audio;
    if (!is_pinned_host(audio)) {   // pageable memory: stage through the engine's pinned buffer (an async copy from
                                    // pageable memory is a hidden synchronous staging copy inside the runtime)
      HIPCHK(h, hipEventSynchronize(h->audio_evt));
      memcpy(h->audio_pinned, audio, bytes);
      src = h->audio_pinned;
    }
    HIPCHK(h, hipMemcpyAsync(h->audio_dev, src, bytes, hipMemcpyHostToDevice, st));
    if (src == h->audio_pinned) HIPCHK(h, hipEventRecord(h->audio_evt, st));
    ad = h->audio_dev;
  }
  float* od = (flags & VAPX_OUT_DEVICE) ? out : h->out_dev;
  if (G > 1) {
    HIPCHK(h, hipEventRecord(h->gstart, st));
    for (int g = 0; g < G; ++g) HIPCHK(h, hipStreamWaitEvent(h->gstream[g], h->gstart, 0));
  }
  for (int g = 0; g < G; ++g) {
    int b0 = (int)((long)n * g / G), b1 = (int)((long)n * (g + 1) / G);
    const int nb = b1 - b0;
    hipStream_t gs = G > 1 ? h->gstream[g] : st;
    const Scratch sc = h->sc.slice(b0, h->P, h->ncpc, h->T);
    const int* gids = ids ? ids + b0 : nullptr;
    Scratch lsc;
    if (lead) lsc = lead->sc.slice(b0, h->P, h->ncpc, h->T);
    rc = step_group(h, sc, nb, b0, gids, lead ? nullptr : ad + (size_t)b0 * 2 * spc, spc, od + (size_t)b0 * VAPX_OUT_STRIDE, gs,
                    lead ? &lsc : nullptr);
    if (rc) return rc;
    if (G > 1) HIPCHK(h, hipEventRecord(h->gdone[g], gs));
  }
  if (G > 1 && !defer_join)
    for (int g = 0; g < G; ++g) HIPCHK(h, hipStreamWaitEvent(st, h->gdone[g], 0));
  h->deferred_pending = defer_join;
  h->last_G = G;
  h->last_B = n;
  h->last_ids = ids;
  if (lead) h->followed_tick = lead->tick; else ++h->tick;
  if (!(flags & VAPX_OUT_DEVICE)) {
    const size_t bytes = (size_t)n * VAPX_OUT_STRIDE * sizeof(float);
    if (is_pinned_host(out)) {   // vapx_host_alloc memory: the D2H lands in the caller's buffer directly
      HIPCHK(h, hipMemcpyAsync(out, h->out_dev, bytes, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
    } else {
      HIPCHK(h, hipMemcpyAsync(h->out_pinned, h->out_dev, bytes, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      memcpy(out, h->out_pinned, bytes);
    }
    // Fail loudly rather than hand NaNs to a dialogue system — but per stream: the out block is complete, the healthy rows
    // are valid, the offending batch slots carry VAPX_OUT_STATUS = 1 and are listed by vapx_bad_slots().
    h->bad_slots.clear();
    for (int i = 0; i < n; ++i)
      if (out[(size_t)i * VAPX_OUT_STRIDE + VAPX_OUT_STATUS] != 0.f) h->bad_slots.push_back(i);
    if (!h->bad_slots.empty()) {
      const int i = h->bad_slots[0];
      return fail(h, VAPX_E_NUMERIC, "non-finite outputs for batch slot %d (stream %d) and %zu more; the other rows are valid; these streams' state is "
                  "poisoned: reset them (vapx_bad_slots lists the slots)", i,
                  (stream_ids && !(flags & VAPX_IDS_DEVICE)) ? stream_ids[i] : i, h->bad_slots.size() - 1);
    }
  }
  return VAPX_OK;
}

int32_t vapx_bad_slots(vapx_handle h, int32_t* slots, int32_t max_slots) {
  if (!h) return VAPX_E_INVAL;
  const int32_t n = (int32_t)h->bad_slots.size();
  for (int32_t i = 0; i < n && i < max_slots && slots; ++i) slots[i] = h->bad_slots[i];
  return n;
}

void* vapx_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void vapx_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int vapx_join(vapx_handle h, void* hip_stream) {
  if (!h) return VAPX_E_INVAL;
  (void)hipGetLastError();   // a stale error of an earlier, unrelated HIP call (this library's or anyone's) is not this call's
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  for (int g = 0; g < h->last_G && h->last_G > 1; ++g) HIPCHK(h, hipStreamWaitEvent((hipStream_t)hip_stream, h->gdone[g], 0));
  return VAPX_OK;   // deferred_pending stays set: only `hip_stream` waited, a later step on another stream still has to
}

int vapx_attach_trunk(vapx_handle f, vapx_handle lead) {
  if (!f || !lead) return VAPX_E_INVAL;
  if (f == lead || lead->trunk || f->trunk || !f->followers.empty())
    return fail(f, VAPX_E_INVAL, "attach a stand-alone engine to a leader that is not itself a follower");
  if (f->tick != 0 || lead->tick != 0) return fail(f, VAPX_E_INVAL, "attach before the first step of either engine");
  if (f->cfg.device_id != lead->cfg.device_id || f->cfg.frame_hz != lead->cfg.frame_hz || f->T != lead->T ||
      f->cfg.max_streams != lead->cfg.max_streams || f->cfg.max_batch != lead->cfg.max_batch)
    return fail(f, VAPX_E_INVAL, "device, frame_hz, ctx_frames, max_streams and max_batch must match the leader's");
  HIPCHK(f, hipSetDevice(f->cfg.device_id));
  HIPCHK(f, hipDeviceSynchronize());
  {  // the CPC CNN + LSTM weights must be the same tensors (they come from the common cpc_model file, vap_main.py:199-201)
    const float* a0 = f->W("conv0.w"); const float* a1 = f->W("down.w");
    const float* b0 = lead->W("conv0.w");
    const size_t nfl = (size_t)(a1 - a0);
    std::vector<float> ha(nfl), hb(nfl);
    HIPCHK(f, hipMemcpy(ha.data(), a0, nfl * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(f, hipMemcpy(hb.data(), b0, nfl * sizeof(float), hipMemcpyDeviceToHost));
    if (memcmp(ha.data(), hb.data(), nfl * sizeof(float)) != 0)
      return fail(f, VAPX_E_INVAL, "CPC encoder weights differ from the leader's: nothing to share");
  }
  // a follower never runs the encoder: release its encoder scratch and LSTM / carry state
  float** drop[] = {&f->sc.h0, &f->sc.h1, &f->sc.h2, &f->sc.h3, &f->sc.z, &f->sc.gx, &f->sc.lstm_out, &f->audio_dev,
                    &f->h_state, &f->c_state, &f->carry};
  for (float** p : drop) {
    if (*p) {
      f->poison.erase(std::remove_if(f->poison.begin(), f->poison.end(), [&](const std::pair<void*, size_t>& pb) { return pb.first == (void*)*p; }),
                      f->poison.end());
      dfree(*p);
    }
    *p = nullptr;
  }
  f->trunk = lead;
  lead->followers.push_back(f);
  return VAPX_OK;
}

int vapx_reset_stream(vapx_handle h, int32_t sid) {
  if (!h) return VAPX_E_INVAL;
  if (sid < 0 || sid >= h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "stream id out of range");
  if (h->trunk) return fail(h, VAPX_E_INVAL, "reset the trunk leader: it resets its followers too");
  // No device work and no synchronisation here: the request is queued and the NEXT vapx_step applies it with one tiny
  // kernel ordered on its HIP stream, before that step touches any state — a joining client costs the other streams nothing.
  for (int32_t& q : h->pending_resets) {
    if (q == sid) return VAPX_OK;
    if (q == -sid - 1) { q = sid; return VAPX_OK; }   // a queued carry-only reset is subsumed
  }
  h->pending_resets.push_back(sid);
  return VAPX_OK;
}

int vapx_reset_carry(vapx_handle h, int32_t sid) {
  if (!h) return VAPX_E_INVAL;
  if (sid < 0 || sid >= h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "stream id out of range");
  if (h->trunk) return fail(h, VAPX_E_INVAL, "the carry lives in the trunk leader");
  for (int32_t q : h->pending_resets)
    if (q == sid || q == -sid - 1) return VAPX_OK;   // a full reset (or the same request) is already queued
  h->pending_resets.push_back(-sid - 1);
  return VAPX_OK;
}

int vapx_get_config(vapx_handle h, vapx_config* out) {
  if (!h || !out) return VAPX_E_INVAL;
  *out = h->cfg;
  return VAPX_OK;
}

int vapx_get_state(vapx_handle h, int32_t sid, float* ring, int32_t* n_frames, float* lstm, float* carry) {
  if (!h) return VAPX_E_INVAL;
  if (sid < 0 || sid >= h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "stream id out of range");
  { int rc = quiesce(h); if (rc) return rc; }
  int fs = 0;
  HIPCHK(h, hipMemcpy(&fs, h->frames_seen + sid, sizeof(int), hipMemcpyDeviceToHost));
  const int T = h->T, n = fs < T ? fs : T;
  if (n_frames) *n_frames = n;
  if (ring) {
    std::vector<float> tmp((size_t)2 * T * 256);
    HIPCHK(h, hipMemcpy(tmp.data(), h->ring + (size_t)sid * 2 * T * 256, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int c = 0; c < 2; ++c)
      for (int t = 0; t < n; ++t) {
        int slot = ((fs - n + t) % T + T) % T;
        memcpy(ring + ((size_t)c * T + t) * 256, tmp.data() + ((size_t)c * T + slot) * 256, 256 * sizeof(float));
      }
  }
  if (h->trunk && (lstm || carry)) return fail(h, VAPX_E_INVAL, "LSTM / carry state lives in the trunk leader");
  if (lstm) {
    for (int c = 0; c < 2; ++c) {
      HIPCHK(h, hipMemcpy(lstm + (c * 2 + 0) * 256, h->h_state + ((size_t)sid * 2 + c) * 256, 256 * sizeof(float), hipMemcpyDeviceToHost));
      HIPCHK(h, hipMemcpy(lstm + (c * 2 + 1) * 256, h->c_state + ((size_t)sid * 2 + c) * 256, 256 * sizeof(float), hipMemcpyDeviceToHost));
    }
  }
  if (carry) HIPCHK(h, hipMemcpy(carry, h->carry + (size_t)sid * 2 * VAPX_PAD, 2 * VAPX_PAD * sizeof(float), hipMemcpyDeviceToHost));
  return VAPX_OK;
}

int vapx_set_state(vapx_handle h, int32_t sid, const float* ring, int32_t n_frames, const float* lstm, const float* carry) {
  if (!h) return VAPX_E_INVAL;
  if (sid < 0 || sid >= h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "stream id out of range");
  if (n_frames < 0 || n_frames > h->T) return fail(h, VAPX_E_INVAL, "n_frames outside [0,T]");
  if (h->trunk && (lstm || carry)) return fail(h, VAPX_E_INVAL, "LSTM / carry state lives in the trunk leader");
  { int rc = quiesce(h); if (rc) return rc; }
  const int T = h->T;
  if (ring) {
    // chronological rows land in slots 0..n-1 and frames_seen = n, so the next append goes to slot n % T
    for (int c = 0; c < 2; ++c)
      HIPCHK(h, hipMemcpy(h->ring + ((size_t)sid * 2 + c) * T * 256, ring + (size_t)c * T * 256, (size_t)n_frames * 256 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->frames_seen + sid, &n_frames, sizeof(int), hipMemcpyHostToDevice));
    if (n_frames > 0) {
      // rebuild the layer-0 Q|K|V cache of the imported rows: LN(ring rows) . Wqkv^T (slots 0..n-1)
      hipLaunchKernelGGL(fill_int_kernel, dim3(1), dim3(64), 0, nullptr, h->sc.bn, T, 1);
      GatherArgs ga;
      memset(&ga, 0, sizeof ga);
      ga.xin = h->ring + (size_t)sid * 2 * T * 256; ga.bn = h->sc.bn; ga.bhead = h->sc.bhead;
      ga.x0 = h->sc.xl[0]; ga.xn = h->sc.xn; ga.gamma = h->layer[0].ln_self_g; ga.beta = h->layer[0].ln_self_b;
      ga.B = 1; ga.T = T; ga.rows_in = T;
      HIPCHK(h, launch_gather_ln(ga, nullptr));
      GemmArgs g = gemm_args(h->sc.xn, contiguous_rows(256), h->layer[0].wqkv, 2 * T, 768, 256,
                             h->ring_qkv + (size_t)sid * 2 * T * 768, contiguous_rows(768));
      HIPCHK(h, launch_gemm_f32(g, EPI_STORE, 0, nullptr));
      HIPCHK(h, hipDeviceSynchronize());
    }
  }
  if (lstm) {
    for (int c = 0; c < 2; ++c) {
      HIPCHK(h, hipMemcpy(h->h_state + ((size_t)sid * 2 + c) * 256, lstm + (c * 2 + 0) * 256, 256 * sizeof(float), hipMemcpyHostToDevice));
      HIPCHK(h, hipMemcpy(h->c_state + ((size_t)sid * 2 + c) * 256, lstm + (c * 2 + 1) * 256, 256 * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  if (carry) HIPCHK(h, hipMemcpy(h->carry + (size_t)sid * 2 * VAPX_PAD, carry, 2 * VAPX_PAD * sizeof(float), hipMemcpyHostToDevice));
  return VAPX_OK;
}

int vapx_encode_audio(vapx_handle h, int32_t n, const int32_t* stream_ids, const float* frames, float* e, void* hip_stream) {
  if (!h) return VAPX_E_INVAL;
  if (h->trunk || h->orphaned) return fail(h, VAPX_E_INVAL, "a trunk follower has no encoder; call the leader");
  if (n < 1 || n > h->cfg.max_batch) return fail(h, VAPX_E_RANGE, "n=%d outside [1,%d]", n, h->cfg.max_batch);
  if (!frames || !e) return fail(h, VAPX_E_INVAL, "null frames/e");
  hipStream_t st = (hipStream_t)hip_stream;
  (void)hipGetLastError();   // a stale error of an earlier, unrelated HIP call (this library's or anyone's) is not this call's
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int* ids = nullptr;
  int rc = join_deferred(h, st);
  if (rc) return rc;
  rc = flush_resets(h, st);
  if (rc) return rc;
  rc = upload_ids(h, n, stream_ids, 0, st, &ids);
  if (rc) return rc;
  const StateView sv{h->ring, h->ring_qkv, h->h_state, h->c_state, h->carry, h->frames_seen};
  rc = run_encoder(h, h->sc, sv, n, ids, frames, h->L, false, st);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(e, h->sc.e, (size_t)n * 2 * 256 * sizeof(float), hipMemcpyDeviceToDevice, st));
  h->last_B = n;
  return VAPX_OK;
}

int vapx_transformer(vapx_handle h, int32_t n, int32_t rows, const float* x, float* o, float* x12, float* comb,
                     int32_t stage, void* hip_stream) {
  if (!h) return VAPX_E_INVAL;
  if (n < 1 || n > h->cfg.max_batch) return fail(h, VAPX_E_RANGE, "n=%d outside [1,%d]", n, h->cfg.max_batch);
  if (rows < 1 || rows > h->T) return fail(h, VAPX_E_RANGE, "rows=%d outside [1,%d]", rows, h->T);
  if (!x) return fail(h, VAPX_E_INVAL, "null x");
  if (stage < 0 || stage > 2) return fail(h, VAPX_E_INVAL, "stage must be 0 (all), 1 (ar_channel) or 2 (ar)");
  if (stage == 1 && (x12 || comb)) return fail(h, VAPX_E_INVAL, "stage 1 produces only o");
  if (stage == 2 && o) return fail(h, VAPX_E_INVAL, "stage 2 does not produce o");
  hipStream_t st = (hipStream_t)hip_stream;
  (void)hipGetLastError();   // a stale error of an earlier, unrelated HIP call (this library's or anyone's) is not this call's
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int T = h->T;
  const int l_begin = stage == 2 ? 1 : 0, l_end = stage == 1 ? 1 : 4;
  { int rc0 = join_deferred(h, st); if (rc0) return rc0; }   // the stage API shares the step's scratch
  hipLaunchKernelGGL(fill_int_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h->sc.bn, rows, n);
  GatherArgs ga;
  memset(&ga, 0, sizeof ga);
  ga.ring = nullptr; ga.e = nullptr; ga.xin = x; ga.ids = nullptr; ga.bn = h->sc.bn; ga.bhead = h->sc.bhead;
  ga.x0 = h->sc.xl[l_begin]; ga.xn = h->sc.xn; ga.gamma = h->layer[l_begin].ln_self_g; ga.beta = h->layer[l_begin].ln_self_b;
  ga.B = n; ga.T = T; ga.rows_in = rows;
  { ProfScope ps(h, CLS_GATHER, st); HIPCHK(h, launch_gather_ln(ga, st)); }
  int rc = run_layers(h, h->sc, n, st, l_begin, l_end);
  if (rc) return rc;
  const long nro = (long)n * 2 * rows;
  const unsigned cgrid = (unsigned)((nro * 64 + 255) / 256);
  if (o) hipLaunchKernelGGL(compact_rows_kernel, dim3(cgrid), dim3(256), 0, st, o, h->sc.xl[1], T, rows, nro);
  if (x12) hipLaunchKernelGGL(compact_rows_kernel, dim3(cgrid), dim3(256), 0, st, x12, h->sc.xl[4], T, rows, nro);
  if (comb) {
    rc = run_combinator_all_rows(h, h->sc, n, st);
    if (rc) return rc;
    const long nrc = (long)n * rows;
    // xmid is [n][T][256]; compact with "2 channels" folded: treat as bc = stream
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)((nrc * 64 + 255) / 256)), dim3(256), 0, st, comb, h->sc.xmid, T, rows, nrc);
  }
  HIPCHK(h, hipGetLastError());
  h->last_B = n;
  return VAPX_OK;
}

int64_t vapx_peek(vapx_handle h, const char* name, float* dst, size_t max_floats) {
  if (!h || !name || !dst) return VAPX_E_INVAL;
  { int rc = quiesce(h); if (rc) return rc; }
  const size_t B = h->last_B, T = h->T;
  const int* P = h->P;
  const float* src = nullptr;
  size_t n = 0;
  if (!strcmp(name, "guard_violations")) {   // debug: see VAPX_GUARD_ZONES
    dst[0] = guard_enabled() ? (float)guard_violations() : -1.f;
    return 1;
  }
  if (!strcmp(name, "h0")) { src = h->sc.h0; n = B * 2 * (P[0] + 4) * 256; }
  else if (!strcmp(name, "h1")) { src = h->sc.h1; n = B * 2 * (P[1] + 2) * 256; }
  else if (!strcmp(name, "h2") || !strcmp(name, "h3")) {
    if (conv_tail_supported(P[1], h->ncpc) && !(h->cfg.flags & VAPX_FLAG_UNFUSED_CONV) && B <= 512)
      return fail(h, VAPX_E_INVAL, "\"%s\" stays in LDS in the fused conv tail; create the engine with VAPX_FLAG_UNFUSED_CONV to peek it", name);
    if (name[1] == '2') { src = h->sc.h2; n = B * 2 * (P[2] + 2) * 256; }
    else { src = h->sc.h3; n = B * 2 * (P[3] + 2) * 256; }
  }
  else if (!strcmp(name, "z")) { src = h->sc.z; n = B * 2 * h->ncpc * 256; }
  else if (!strcmp(name, "lstm_out")) { src = h->sc.lstm_out; n = B * 2 * h->ncpc * 256; }
  else if (!strcmp(name, "e")) { src = h->sc.e; n = B * 2 * 256; }
  else if (!strcmp(name, "x0")) {
    if (!(h->cfg.flags & VAPX_FLAG_MATERIALIZE_X0) && (h->T <= 64 || !(h->cfg.flags & VAPX_FLAG_UNFUSED_PROJ)))
      return fail(h, VAPX_E_INVAL, "\"x0\" is read straight from the ring; create the engine with VAPX_FLAG_MATERIALIZE_X0 to peek it");
    src = h->sc.xl[0]; n = B * 2 * T * 256;
  }
  else if (!strcmp(name, "o")) { src = h->sc.xl[1]; n = B * 2 * T * 256; }
  else if (!strcmp(name, "stereo0")) { src = h->sc.xl[2]; n = B * 2 * T * 256; }
  else if (!strcmp(name, "stereo1")) { src = h->sc.xl[3]; n = B * 2 * T * 256; }
  else if (!strcmp(name, "stereo2")) {
    if (!(h->cfg.flags & VAPX_FLAG_FULL_LAST_LAYER) && h->cfg.mode != VAPX_MODE_NOD)
      return fail(h, VAPX_E_INVAL, "\"stereo2\" is only materialised with VAPX_FLAG_FULL_LAST_LAYER (default: last layer runs on the newest row only)");
    src = h->sc.xl[4]; n = B * 2 * T * 256;
  }
  else return fail(h, VAPX_E_INVAL, "unknown buffer '%s'", name);
  if (n > max_floats) n = max_floats;
  HIPCHK(h, hipMemcpy(dst, src, n * sizeof(float), hipMemcpyDeviceToHost));
  return (int64_t)n;
}

int vapx_profile_enable(vapx_handle h, uint32_t class_mask) {
  if (!h) return VAPX_E_INVAL;
  h->prof_mask = class_mask;
  return VAPX_OK;
}

int vapx_profile_read(vapx_handle h, double* total_ms, int64_t* launches, int32_t n_classes) {
  if (!h || !total_ms || !launches) return VAPX_E_INVAL;
  (void)hipGetLastError();   // a stale error of an earlier, unrelated HIP call (this library's or anyone's) is not this call's
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipDeviceSynchronize());
  for (int i = 0; i < n_classes; ++i) { total_ms[i] = 0.0; launches[i] = 0; }
  for (auto& r : h->prof_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.cls < n_classes) { total_ms[r.cls] += ms; launches[r.cls] += 1; }
    h->prof_pool.push_back(r.a);
    h->prof_pool.push_back(r.b);
  }
  h->prof_recs.clear();
  return VAPX_OK;
}

int vapx_vap_head(vapx_handle h, int64_t rows, const float* x, float* logits, void* hip_stream) {
  if (!h || !x || !logits || rows < 1) return VAPX_E_INVAL;
  if (h->cfg.mode != VAPX_MODE_VAP) return fail(h, VAPX_E_INVAL, "this weight set has no vap_head (bc / nod variant)");
  (void)hipGetLastError();   // a stale error of an earlier, unrelated HIP call (this library's or anyone's) is not this call's
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  GemmArgs g = gemm_args(x, contiguous_rows(256), h->W("head.w"), (int)rows, 256, 256, logits, contiguous_rows(256));
  g.bias = h->W("head.b");
  HIPCHK(h, launch_gemm_f32(g, EPI_STORE, 0, (hipStream_t)hip_stream));
  return VAPX_OK;
}

int vapx_va_classifier(vapx_handle h, int64_t rows, const float* x, float* y, void* hip_stream) {
  if (!h || !x || !y || rows < 1) return VAPX_E_INVAL;
  (void)hipGetLastError();   // a stale error of an earlier, unrelated HIP call (this library's or anyone's) is not this call's
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)hip_stream, x, h->W("vad.w"), h->W("vad.b"), y, (long)rows);
  HIPCHK(h, hipGetLastError());
  return VAPX_OK;
}

int vapx_aux_head(vapx_handle h, int32_t which, int64_t rows, const float* x, float* y, void* hip_stream) {
  if (!h || !x || !y || rows < 1) return VAPX_E_INVAL;
  // aux.w rows: bc variant = bc_head rows 0..2; nod variant = nod_head rows 0..3, bc_head row 4 (weights.pack_blob)
  int row0, nout;
  if (h->cfg.mode == VAPX_MODE_BC && which == VAPX_AUX_BC_HEAD) { row0 = 0; nout = 3; }
  else if (h->cfg.mode == VAPX_MODE_NOD && which == VAPX_AUX_BC_HEAD) { row0 = 4; nout = 1; }
  else if (h->cfg.mode == VAPX_MODE_NOD && which == VAPX_AUX_NOD_HEAD) { row0 = 0; nout = 4; }
  else return fail(h, VAPX_E_INVAL, "this weight set has no such head (bc_head: bc / nod variants, nod_head: nod variant)");
  (void)hipGetLastError();
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  hipLaunchKernelGGL(rowheads_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)hip_stream, x, h->W("aux.w") + row0 * 256,
                     h->W("aux.b") + row0, y, (long)rows, nout);
  HIPCHK(h, hipGetLastError());
  return VAPX_OK;
}

int vapx_softmax256(int64_t rows, const float* x, float* y, void* hip_stream) {
  if (!x || !y || rows < 1) return VAPX_E_INVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(softmax256_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)hip_stream, x, y, (long)rows);
  return hipGetLastError() == hipSuccess ? VAPX_OK : VAPX_E_HIP;
}

int vapx_aggregate(int64_t rows, const float* probs, int32_t from_bin, int32_t to_bin, float* out, void* hip_stream) {
  if (!probs || !out || rows < 1 || from_bin < 0 || to_bin > 3 || from_bin > to_bin) return VAPX_E_INVAL;
  (void)hipGetLastError();
  hipLaunchKernelGGL(aggregate_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)hip_stream, probs, out, (long)rows, from_bin, to_bin);
  return hipGetLastError() == hipSuccess ? VAPX_OK : VAPX_E_HIP;
}

int vapx_gemm(void* hip_stream, int32_t M, int32_t N, int32_t K, const float* A, const float* W, float* C, int32_t epi,
              const float* bias, const float* gamma, const float* beta, const float* resid, float* C2, int32_t tile_rows) {
  (void)hipGetLastError();
  GemmArgs g = gemm_args(A, contiguous_rows(K), W, M, N, K, C, contiguous_rows(N));
  g.bias = bias; g.gamma = gamma; g.beta = beta; g.resid = resid; g.C2 = C2;
  hipError_t e = launch_gemm_f32(g, epi, tile_rows, (hipStream_t)hip_stream);
  if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? VAPX_E_INVAL : VAPX_E_HIP, "vapx_gemm: %s", hipGetErrorString(e));
  return VAPX_OK;
}

}  // extern "C"
