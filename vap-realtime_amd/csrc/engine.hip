// libvapx engine: owns device weights, per-stream state and scratch; orchestrates one VAP frame for
// a batch of streams (the C ABI of include/vapx.h).  Compiled with hipcc for gfx950 only.
//
// Reference path being replaced: VAPRealTime.__init__/process_vap, rvap/vap_main/vap_main.py:192-335.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vapx.h"
#include "gemm_f32.h"
#include "vap_kernels.h"
#include "vapx_layout.h"

namespace {

thread_local std::string g_create_error;

struct Layer {
  const float *ln_self_g, *ln_self_b, *wqkv, *wproj;
  const float *ln_src_g, *ln_src_b, *wq_x, *wkv_x, *wproj_x;
  const float *ln_ffn_g, *ln_ffn_b, *w0, *w3;
};

}  // namespace

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
  float *ring = nullptr, *h_state = nullptr, *c_state = nullptr, *carry = nullptr;
  int* frames_seen = nullptr;

  // scratch (max_batch)
  float *audio_dev = nullptr, *out_dev = nullptr;
  int *ids_dev = nullptr, *bn = nullptr, *bhead = nullptr;
  float *h0 = nullptr, *h1 = nullptr, *h2 = nullptr, *h3 = nullptr, *z = nullptr, *lstm_out = nullptr, *e = nullptr;
  float* xl[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // layer inputs/outputs: x0, o, stereo0..2
  float *xn = nullptr, *xmid = nullptr, *att = nullptr, *qkv = nullptr, *qx = nullptr, *kvx = nullptr, *ffn = nullptr;
  float *gx = nullptr;
  float* out_pinned = nullptr;
  int* ids_pinned = nullptr;
  hipEvent_t ids_evt = nullptr;
  int last_B = 0;

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

template <typename T>
hipError_t dalloc(T** p, size_t n, bool zero = true) {
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  if (e != hipSuccess) return e;
  return zero ? hipMemset(*p, 0, n * sizeof(T)) : hipSuccess;
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

// kernel classes for profiling: 0..5 = GEMM by epilogue, then the rest
enum { CLS_CONV0 = 8, CLS_LSTM = 9, CLS_GATHER = 10, CLS_ATTN = 11, CLS_HEAD = 12, CLS_COUNT = 13 };

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

hipError_t gemm(vapx_engine* h, const GemmArgs& g, int epi, hipStream_t st) {
  ProfScope ps(h, epi, st);
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
int run_encoder(vapx_engine* h, int B, const int* ids_dev, const float* audio, int spc, bool use_state_meta,
                hipStream_t st) {
  const int* P = h->P;
  Conv0Args c0;
  c0.audio = audio; c0.ids = ids_dev; c0.carry = use_state_meta ? h->carry : nullptr;
  c0.h0 = h->h0; c0.w = h->W("conv0.w"); c0.bias = h->W("conv0.b"); c0.gamma = h->W("cn0.g"); c0.beta = h->W("cn0.b");
  c0.frames_seen = use_state_meta ? h->frames_seen : nullptr; c0.bn = h->bn; c0.bhead = h->bhead;
  c0.L = h->L; c0.spc = spc; c0.T = h->T;
  { ProfScope ps(h, CLS_CONV0, st); HIPCHK(h, launch_conv0(c0, B, st)); }

  struct ConvSpec { const float* in; int Pin, guard_in, k, s; float* out; int Pout, guard_out; const char* idx; };
  const ConvSpec cs[3] = {
      {h->h0, P[0], 2, 8, 4, h->h1, P[1], 1, "1"},
      {h->h1, P[1], 1, 4, 2, h->h2, P[2], 1, "2"},
      {h->h2, P[2], 1, 4, 2, h->h3, P[3], 1, "3"},
  };
  char nm[32];
  for (int i = 0; i < 3; ++i) {
    const ConvSpec& c = cs[i];
    RowMap am{(long)(c.Pin + 2 * c.guard_in) * 256, (long)c.s * 256, c.Pout};
    RowMap cm{(long)(c.Pout + 2 * c.guard_out) * 256, 256, c.Pout};
    snprintf(nm, sizeof nm, "conv%s.w", c.idx);
    GemmArgs g = gemm_args(c.in, am, h->W(nm), B * 2 * c.Pout, 256, c.k * 256, c.out + c.guard_out * 256, cm);
    snprintf(nm, sizeof nm, "conv%s.b", c.idx); g.bias = h->W(nm);
    snprintf(nm, sizeof nm, "cn%s.g", c.idx); g.gamma = h->W(nm);
    snprintf(nm, sizeof nm, "cn%s.b", c.idx); g.beta = h->W(nm);
    HIPCHK(h, gemm(h, g, EPI_CN_RELU, st));
  }
  {  // conv4: only positions 1..P4-2 survive z[:, 1:-1] (encoder.py:76)
    RowMap am{(long)(P[3] + 2) * 256, 2 * 256, h->ncpc};
    GemmArgs g = gemm_args(h->h3 + 2 * 256, am, h->W("conv4.w"), B * 2 * h->ncpc, 256, 4 * 256, h->z, contiguous_rows(256));
    g.bias = h->W("conv4.b"); g.gamma = h->W("cn4.g"); g.beta = h->W("cn4.b");
    HIPCHK(h, gemm(h, g, EPI_CN_RELU, st));
  }
  {  // LSTM input projection for all n_cpc steps at once: gx = z.W_ih^T + (b_ih + b_hh)
    GemmArgs g = gemm_args(h->z, contiguous_rows(256), h->W("lstm.wih"), B * 2 * h->ncpc, 1024, 256, h->gx, contiguous_rows(1024));
    g.bias = h->W("lstm.b");
    HIPCHK(h, gemm(h, g, EPI_STORE, st));
  }
  LstmArgs la;
  la.gx = h->gx; la.ids = ids_dev; la.h_state = h->h_state; la.c_state = h->c_state;
  la.wfrag = h->W("lstm.whh"); la.out = h->lstm_out; la.M = B * 2; la.ncpc = h->ncpc;
  { ProfScope ps(h, CLS_LSTM, st); HIPCHK(h, launch_lstm(la, st)); }
  {  // downsample: single-output Conv1d == dense [ncpc*256 -> 256] + LN + GELU
    GemmArgs g = gemm_args(h->lstm_out, contiguous_rows((long)h->ncpc * 256), h->W("down.w"), B * 2, 256, h->ncpc * 256,
                           h->e, contiguous_rows(256));
    g.bias = h->W("down.b"); g.gamma = h->W("down.g"); g.beta = h->W("down.beta");
    HIPCHK(h, gemm(h, g, EPI_BIAS_LN_GELU, st));
  }
  return VAPX_OK;
}

// ---- 1 self + 3 self/cross layers on x0 = xl[0] (LN_self(L0) already in xn) ---------------------
int run_layers(vapx_engine* h, int B, hipStream_t st) {
  const int T = h->T;
  const int M = B * 2 * T;
  const RowMap r256 = contiguous_rows(256), r768 = contiguous_rows(768), r512 = contiguous_rows(512);
  for (int l = 0; l < 4; ++l) {
    const Layer& Lw = h->layer[l];
    const float* xin = h->xl[l];
    float* xout = h->xl[l + 1];
    // self attention
    GemmArgs g = gemm_args(h->xn, r256, Lw.wqkv, M, 768, 256, h->qkv, r768);
    HIPCHK(h, gemm(h, g, EPI_STORE, st));
    AttnArgs aa{h->qkv, h->qkv + 256, h->qkv + 512, h->att, h->bn, T, 768, 768, 0};
    { ProfScope ps(h, CLS_ATTN, st); HIPCHK(h, launch_attention(aa, B, st)); }
    g = gemm_args(h->att, r256, Lw.wproj, M, 256, 256, h->xmid, r256);
    g.resid = xin; g.C2 = h->xn;
    if (l == 0) { g.gamma = Lw.ln_ffn_g; g.beta = Lw.ln_ffn_b; }
    else { g.gamma = Lw.ln_src_g; g.beta = Lw.ln_src_b; }
    HIPCHK(h, gemm(h, g, EPI_RESID_LN, st));
    if (l > 0) {
      // cross attention: Q from LN_src(x), K/V from the OTHER channel's raw layer input
      g = gemm_args(h->xn, r256, Lw.wq_x, M, 256, 256, h->qx, r256);
      HIPCHK(h, gemm(h, g, EPI_STORE, st));
      g = gemm_args(xin, r256, Lw.wkv_x, M, 512, 256, h->kvx, r512);
      HIPCHK(h, gemm(h, g, EPI_STORE, st));
      AttnArgs ax{h->qx, h->kvx, h->kvx + 256, h->att, h->bn, T, 256, 512, 1};
      { ProfScope ps(h, CLS_ATTN, st); HIPCHK(h, launch_attention(ax, B, st)); }
      g = gemm_args(h->att, r256, Lw.wproj_x, M, 256, 256, h->xmid, r256);
      g.resid = h->xmid; g.C2 = h->xn; g.gamma = Lw.ln_ffn_g; g.beta = Lw.ln_ffn_b;
      HIPCHK(h, gemm(h, g, EPI_RESID_LN, st));
    }
    // feed-forward
    g = gemm_args(h->xn, r256, Lw.w0, M, 768, 256, h->ffn, r768);
    HIPCHK(h, gemm(h, g, EPI_GELU, st));
    g = gemm_args(h->ffn, r768, Lw.w3, M, 256, 768, xout, r256);
    g.resid = h->xmid;
    if (l < 3) {
      g.C2 = h->xn; g.gamma = h->layer[l + 1].ln_self_g; g.beta = h->layer[l + 1].ln_self_b;
      HIPCHK(h, gemm(h, g, EPI_RESID_LN, st));
    } else {
      HIPCHK(h, gemm(h, g, EPI_RESID, st));
    }
  }
  return VAPX_OK;
}

__global__ void fill_int_kernel(int* p, int v, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void add_kernel(float* o, const float* a, const float* b, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
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
  for (int i = 0; i < n; ++i)
    if (ids[i] < 0 || ids[i] >= h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "stream id %d out of range [0,%d)", ids[i], h->cfg.max_streams);
  HIPCHK(h, hipEventSynchronize(h->ids_evt));  // previous async copy out of the pinned buffer is done
  memcpy(h->ids_pinned, ids, n * sizeof(int));
  HIPCHK(h, hipMemcpyAsync(h->ids_dev, h->ids_pinned, n * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(h, hipEventRecord(h->ids_evt, st));
  *out = h->ids_dev;
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
  (void)hipSetDevice(h->cfg.device_id);
  (void)hipDeviceSynchronize();
  float* fp[] = {h->w, h->ring, h->h_state, h->c_state, h->carry, h->audio_dev, h->out_dev, h->h0, h->h1, h->h2, h->h3,
                 h->z, h->lstm_out, h->e, h->xl[0], h->xl[1], h->xl[2], h->xl[3], h->xl[4], h->xn, h->xmid, h->att,
                 h->qkv, h->qx, h->kvx, h->ffn, h->gx};
  for (float* p : fp)
    if (p) (void)hipFree(p);
  int* ip[] = {h->frames_seen, h->ids_dev, h->bn, h->bhead};
  for (int* p : ip)
    if (p) (void)hipFree(p);
  if (h->out_pinned) (void)hipHostFree(h->out_pinned);
  if (h->ids_pinned) (void)hipHostFree(h->ids_pinned);
  if (h->ids_evt) (void)hipEventDestroy(h->ids_evt);
  for (auto& r : h->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : h->prof_pool) (void)hipEventDestroy(e);
  delete h;
}

int vapx_create(const vapx_config* cfg, const float* blob, size_t n_floats, vapx_handle* out) {
  if (!cfg || !blob || !out) return fail(nullptr, VAPX_E_INVAL, "null argument");
  if (cfg->struct_size != (int32_t)sizeof(vapx_config)) return fail(nullptr, VAPX_E_INVAL, "vapx_config.struct_size mismatch");
  if (!rate_ok(cfg->frame_hz)) return fail(nullptr, VAPX_E_INVAL, "frame_hz must be 5, 10, 20 or 50");
  if (cfg->ctx_frames < 1 || cfg->ctx_frames > 256) return fail(nullptr, VAPX_E_INVAL, "ctx_frames must be in [1,256]");
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
  }
  const size_t S = cfg->max_streams, B = cfg->max_batch, T = h->T;
  const int* P = h->P;
  CR(dalloc(&h->ring, S * 2 * T * 256));
  CR(dalloc(&h->h_state, S * 2 * 256));
  CR(dalloc(&h->c_state, S * 2 * 256));
  CR(dalloc(&h->carry, S * 2 * VAPX_PAD));
  CR(dalloc(&h->frames_seen, S));
  CR(dalloc(&h->audio_dev, B * 2 * h->L));
  CR(dalloc(&h->out_dev, B * VAPX_OUT_STRIDE));
  CR(dalloc(&h->ids_dev, B));
  CR(dalloc(&h->bn, B));
  CR(dalloc(&h->bhead, B));
  CR(dalloc(&h->h0, B * 2 * (P[0] + 4) * 256));  // guard rows stay zero forever
  CR(dalloc(&h->h1, B * 2 * (P[1] + 2) * 256));
  CR(dalloc(&h->h2, B * 2 * (P[2] + 2) * 256));
  CR(dalloc(&h->h3, B * 2 * (P[3] + 2) * 256));
  CR(dalloc(&h->z, B * 2 * h->ncpc * 256));
  CR(dalloc(&h->lstm_out, B * 2 * h->ncpc * 256));
  CR(dalloc(&h->gx, B * 2 * h->ncpc * 1024));
  CR(dalloc(&h->e, B * 2 * 256));
  const size_t rows = B * 2 * T;
  for (int i = 0; i < 5; ++i) CR(dalloc(&h->xl[i], rows * 256));
  CR(dalloc(&h->xn, rows * 256));
  CR(dalloc(&h->xmid, rows * 256));
  CR(dalloc(&h->att, rows * 256));
  CR(dalloc(&h->qkv, rows * 768));
  CR(dalloc(&h->qx, rows * 256));
  CR(dalloc(&h->kvx, rows * 512));
  CR(dalloc(&h->ffn, rows * 768));
  CR(hipHostMalloc((void**)&h->out_pinned, B * VAPX_OUT_STRIDE * sizeof(float), hipHostMallocDefault));
  CR(hipHostMalloc((void**)&h->ids_pinned, B * sizeof(int), hipHostMallocDefault));
  CR(hipEventCreateWithFlags(&h->ids_evt, hipEventDisableTiming));
  CR(hipEventRecord(h->ids_evt, nullptr));
  CR(hipDeviceSynchronize());
#undef CR
  *out = h;
  return VAPX_OK;
}

int vapx_step(vapx_handle h, int32_t n, const int32_t* stream_ids, const float* audio, int32_t spc, float* out,
              int32_t flags, void* hip_stream) {
  if (!h) return VAPX_E_INVAL;
  if (n < 1 || n > h->cfg.max_batch) return fail(h, VAPX_E_RANGE, "n=%d outside [1,%d]", n, h->cfg.max_batch);
  if (!audio || !out) return fail(h, VAPX_E_INVAL, "null audio/out");
  if (spc != h->hop && spc != h->L) return fail(h, VAPX_E_INVAL, "samples_per_ch must be %d (hop) or %d (full frame)", h->hop, h->L);
  if (!stream_ids && n > h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "n exceeds max_streams");
  hipStream_t st = (hipStream_t)hip_stream;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int* ids = nullptr;
  int rc = upload_ids(h, n, stream_ids, flags, st, &ids);
  if (rc) return rc;
  const float* ad = audio;
  if (!(flags & VAPX_AUDIO_DEVICE)) {
    HIPCHK(h, hipMemcpyAsync(h->audio_dev, audio, (size_t)n * 2 * spc * sizeof(float), hipMemcpyHostToDevice, st));
    ad = h->audio_dev;
  }
  rc = run_encoder(h, n, ids, ad, spc, true, st);
  if (rc) return rc;
  GatherArgs ga;
  ga.ring = h->ring; ga.e = h->e; ga.xin = nullptr; ga.ids = ids; ga.bn = h->bn; ga.bhead = h->bhead;
  ga.x0 = h->xl[0]; ga.xn = h->xn; ga.gamma = h->layer[0].ln_self_g; ga.beta = h->layer[0].ln_self_b;
  ga.B = n; ga.T = h->T; ga.rows_in = 0;
  { ProfScope ps(h, CLS_GATHER, st); HIPCHK(h, launch_gather_ln(ga, st)); }
  rc = run_layers(h, n, st);
  if (rc) return rc;
  float* od = (flags & VAPX_OUT_DEVICE) ? out : h->out_dev;
  HeadArgs ha;
  ha.x = h->xl[4]; ha.o = h->xl[1]; ha.e = h->e; ha.bn = h->bn; ha.ids = ids; ha.frames_seen = h->frames_seen;
  ha.waT = h->W("comb.waT"); ha.wbT = h->W("comb.wbT"); ha.cg = h->W("comb.g"); ha.cb = h->W("comb.b");
  ha.hwT = h->W("head.wT"); ha.hb = h->W("head.b"); ha.vw = h->W("vad.w"); ha.vb = h->W("vad.b");
  ha.aw = h->W("aux.w"); ha.ab = h->W("aux.b"); ha.out = od; ha.B = n; ha.T = h->T; ha.mode = h->cfg.mode;
  ha.out_stride = VAPX_OUT_STRIDE;
  { ProfScope ps(h, CLS_HEAD, st); HIPCHK(h, launch_head(ha, st)); }
  h->last_B = n;
  if (!(flags & VAPX_OUT_DEVICE)) {
    HIPCHK(h, hipMemcpyAsync(h->out_pinned, h->out_dev, (size_t)n * VAPX_OUT_STRIDE * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    memcpy(out, h->out_pinned, (size_t)n * VAPX_OUT_STRIDE * sizeof(float));
  }
  return VAPX_OK;
}

int vapx_reset_stream(vapx_handle h, int32_t sid) {
  if (!h) return VAPX_E_INVAL;
  if (sid < 0 || sid >= h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "stream id out of range");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemset(h->h_state + (size_t)sid * 512, 0, 512 * sizeof(float)));
  HIPCHK(h, hipMemset(h->c_state + (size_t)sid * 512, 0, 512 * sizeof(float)));
  HIPCHK(h, hipMemset(h->carry + (size_t)sid * 2 * VAPX_PAD, 0, 2 * VAPX_PAD * sizeof(float)));
  HIPCHK(h, hipMemset(h->frames_seen + sid, 0, sizeof(int)));
  return VAPX_OK;
}

int vapx_get_state(vapx_handle h, int32_t sid, float* ring, int32_t* n_frames, float* lstm, float* carry) {
  if (!h) return VAPX_E_INVAL;
  if (sid < 0 || sid >= h->cfg.max_streams) return fail(h, VAPX_E_RANGE, "stream id out of range");
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipDeviceSynchronize());
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
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipDeviceSynchronize());
  const int T = h->T;
  if (ring) {
    // chronological rows land in slots 0..n-1 and frames_seen = n, so the next append goes to slot n % T
    for (int c = 0; c < 2; ++c)
      HIPCHK(h, hipMemcpy(h->ring + ((size_t)sid * 2 + c) * T * 256, ring + (size_t)c * T * 256, (size_t)n_frames * 256 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->frames_seen + sid, &n_frames, sizeof(int), hipMemcpyHostToDevice));
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
  if (n < 1 || n > h->cfg.max_batch) return fail(h, VAPX_E_RANGE, "n=%d outside [1,%d]", n, h->cfg.max_batch);
  if (!frames || !e) return fail(h, VAPX_E_INVAL, "null frames/e");
  hipStream_t st = (hipStream_t)hip_stream;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int* ids = nullptr;
  int rc = upload_ids(h, n, stream_ids, 0, st, &ids);
  if (rc) return rc;
  rc = run_encoder(h, n, ids, frames, h->L, false, st);
  if (rc) return rc;
  HIPCHK(h, hipMemcpyAsync(e, h->e, (size_t)n * 2 * 256 * sizeof(float), hipMemcpyDeviceToDevice, st));
  h->last_B = n;
  return VAPX_OK;
}

int vapx_transformer(vapx_handle h, int32_t n, int32_t rows, const float* x, float* o, float* x12, float* comb, void* hip_stream) {
  if (!h) return VAPX_E_INVAL;
  if (n < 1 || n > h->cfg.max_batch) return fail(h, VAPX_E_RANGE, "n=%d outside [1,%d]", n, h->cfg.max_batch);
  if (rows < 1 || rows > h->T) return fail(h, VAPX_E_RANGE, "rows=%d outside [1,%d]", rows, h->T);
  if (!x) return fail(h, VAPX_E_INVAL, "null x");
  hipStream_t st = (hipStream_t)hip_stream;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  const int T = h->T;
  hipLaunchKernelGGL(fill_int_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h->bn, rows, n);
  GatherArgs ga;
  ga.ring = nullptr; ga.e = nullptr; ga.xin = x; ga.ids = nullptr; ga.bn = h->bn; ga.bhead = h->bhead;
  ga.x0 = h->xl[0]; ga.xn = h->xn; ga.gamma = h->layer[0].ln_self_g; ga.beta = h->layer[0].ln_self_b;
  ga.B = n; ga.T = T; ga.rows_in = rows;
  { ProfScope ps(h, CLS_GATHER, st); HIPCHK(h, launch_gather_ln(ga, st)); }
  int rc = run_layers(h, n, st);
  if (rc) return rc;
  const long nro = (long)n * 2 * rows;
  const unsigned cgrid = (unsigned)((nro * 64 + 255) / 256);
  if (o) hipLaunchKernelGGL(compact_rows_kernel, dim3(cgrid), dim3(256), 0, st, o, h->xl[1], T, rows, nro);
  if (x12) hipLaunchKernelGGL(compact_rows_kernel, dim3(cgrid), dim3(256), 0, st, x12, h->xl[4], T, rows, nro);
  if (comb) {
    // Combinator on all rows: gelu(LN(a.Wa^T)) + gelu(LN(b.Wb^T)), shared LN (modules.py:449-464).
    // Tower rows of channel c of stream b sit at ((b*2+c)*T + t): address them with a RowMap.
    const int M = n * T;
    for (int c = 0; c < 2; ++c) {
      RowMap am{(long)2 * T * 256, 256, T};
      GemmArgs g = gemm_args(h->xl[4] + (long)c * T * 256, am, h->W(c ? "comb.wb" : "comb.wa"), M, 256, 256,
                             c ? h->qx : h->att, contiguous_rows(256));
      g.gamma = h->W("comb.g"); g.beta = h->W("comb.b");
      HIPCHK(h, gemm(h, g, EPI_BIAS_LN_GELU, st));
    }
    const long tot = (long)M * 256;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, h->xmid, h->att, h->qx, tot);
    const long nrc = (long)n * rows;
    // xmid is [n][T][256]; compact with "2 channels" folded: treat as bc = stream
    hipLaunchKernelGGL(compact_rows_kernel, dim3((unsigned)((nrc * 64 + 255) / 256)), dim3(256), 0, st, comb, h->xmid, T, rows, nrc);
  }
  HIPCHK(h, hipGetLastError());
  h->last_B = n;
  return VAPX_OK;
}

int64_t vapx_peek(vapx_handle h, const char* name, float* dst, size_t max_floats) {
  if (!h || !name || !dst) return VAPX_E_INVAL;
  HIPCHK(h, hipSetDevice(h->cfg.device_id));
  HIPCHK(h, hipDeviceSynchronize());
  const size_t B = h->last_B, T = h->T;
  const int* P = h->P;
  const float* src = nullptr;
  size_t n = 0;
  if (!strcmp(name, "h0")) { src = h->h0; n = B * 2 * (P[0] + 4) * 256; }
  else if (!strcmp(name, "h1")) { src = h->h1; n = B * 2 * (P[1] + 2) * 256; }
  else if (!strcmp(name, "h2")) { src = h->h2; n = B * 2 * (P[2] + 2) * 256; }
  else if (!strcmp(name, "h3")) { src = h->h3; n = B * 2 * (P[3] + 2) * 256; }
  else if (!strcmp(name, "z")) { src = h->z; n = B * 2 * h->ncpc * 256; }
  else if (!strcmp(name, "lstm_out")) { src = h->lstm_out; n = B * 2 * h->ncpc * 256; }
  else if (!strcmp(name, "e")) { src = h->e; n = B * 2 * 256; }
  else if (!strcmp(name, "x0")) { src = h->xl[0]; n = B * 2 * T * 256; }
  else if (!strcmp(name, "o")) { src = h->xl[1]; n = B * 2 * T * 256; }
  else if (!strcmp(name, "stereo0")) { src = h->xl[2]; n = B * 2 * T * 256; }
  else if (!strcmp(name, "stereo1")) { src = h->xl[3]; n = B * 2 * T * 256; }
  else if (!strcmp(name, "stereo2")) { src = h->xl[4]; n = B * 2 * T * 256; }
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

int vapx_gemm(void* hip_stream, int32_t M, int32_t N, int32_t K, const float* A, const float* W, float* C, int32_t epi,
              const float* bias, const float* gamma, const float* beta, const float* resid, float* C2, int32_t tile_rows) {
  GemmArgs g = gemm_args(A, contiguous_rows(K), W, M, N, K, C, contiguous_rows(N));
  g.bias = bias; g.gamma = gamma; g.beta = beta; g.resid = resid; g.C2 = C2;
  hipError_t e = launch_gemm_f32(g, epi, tile_rows, (hipStream_t)hip_stream);
  if (e != hipSuccess) return fail(nullptr, e == hipErrorInvalidValue ? VAPX_E_INVAL : VAPX_E_HIP, "vapx_gemm: %s", hipGetErrorString(e));
  return VAPX_OK;
}

}  // extern "C"
