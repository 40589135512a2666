// Argument blocks + launchers of the non-GEMM kernels (see vap_kernels.hip).
#pragma once
#include "common.h"

struct Conv0Args {
  const float* audio;   // [B][2][spc]
  const int* ids;       // [B] stream ids (device) or null = identity
  float* carry;         // [S][2][320] or null (stage API: frames carry their own context)
  float* h0;            // [B*2][P0+4][256] channels-last, 2 zero guard rows each side
  const float* w;       // [10][256]
  const float* bias;    // [256]
  const float* gamma;   // [256]
  const float* beta;    // [256]
  int* frames_seen;     // [S] or null
  int* bn;              // [B] out: rows in the window this frame
  int* bhead;           // [B] out: ring slot the new embedding goes to
  int L, spc, T;
};

struct LstmArgs {
  const float* gx;      // [M][ncpc][1024] input projection + both biases, permuted gate columns
  const int* ids;       // [M/2] or null
  float* h_state;       // [S*2][256]
  float* c_state;       // [S*2][256]
  const float* wfrag;   // W_hh, fragment-major [8 w][16 kc][8 ns][64 lane][4]
  float* out;           // [M][ncpc][256]
  const float* down_wf; // downsample weight, fragment-major [ncpc][8 w][16 kc][2 ns][64 lane][4] (null: skip)
  const float* down_b;  // [256] conv bias
  const float* down_g;  // [256] LayerNorm weight
  const float* down_beta;
  float* e;             // [M][256] embeddings (written when down_wf != null)
  const float* ln0_g;   // transformer layer-0 ln_self (for en)
  const float* ln0_b;
  float* en;            // [M][256] LayerNorm(e; ln0) or null
  int M, ncpc;
};

struct GatherArgs {
  float* ring;          // [S*2][T][256] or null (then xin is used)
  float* ring_qkv;      // [S*2][T][768] cached layer-0 Q|K|V per ring row, or null
  const float* qkv_new; // [B*2][768] this frame's layer-0 Q|K|V (when ring_qkv)
  float* qkv;           // [B*2][T][768] chronological gather target (when ring_qkv)
  int* rot;             // [B] out (ring_append_kernel): ring slot of the oldest row
  const float* e;       // [B*2][256] this frame's embeddings
  const float* xin;     // [B*2][rows_in][256] explicit context (stage API)
  const int* ids;
  const int* bn;
  const int* bhead;
  float* x0;            // [B*2][T][256] chronological, zero beyond n
  float* xn;            // LayerNorm(x0; gamma, beta)
  const float* gamma;
  const float* beta;
  int B, T, rows_in;
};

struct AttnArgs {
  const float* q;       // row (bc*T + i), stride ldq, head h at column h*64
  const float* k;
  const float* v;       // rows of channel (bc ^ swap_kv), stride ldkv
  float* out;           // [B*2*T][256]
  const int* bn;
  int T, ldq, ldkv, swap_kv;
  const int* ring_rot;  // [B] or null.  Non-null (long-window layer 0): q/k/v are the per-stream Q|K|V RINGS (slab = slot*2+channel,
                        // logical row i in ring slot (i + ring_rot[b]) % T) instead of chronological batch buffers
  const int* ids;       // [B] stream slots (null: identity); only used with ring_rot
  int n_items;          // set by launch_attention_f16x3 (persistent kernel): B * 2 channels * 4 heads
#ifdef VAPX_TRACE
  unsigned long long* trace;   // debug build: optional [grid][32] s_memtime stamps of workgroup phases (env VAPX_ATTN_TRACE, long windows)
#endif
};

struct AttnProjArgs {   // attention_proj_f16x3.hip: long-window self-attention with the Q|K|V projection inside (split path, layers >= 1)
  const float* xn;      // [B*2*T][256] LayerNorm(ln_self_attn)(x) rows of the layer (the previous layer's FFN block wrote them)
  const float* wqkvp;   // the layer's per-head weight stream (weights.frag_pack_f16x3_qkv_heads)
  float* out;           // [B*2*T][256] heads merged
  const int* bn;        // [B] valid rows
  int T;
  int n_items;          // set by the launcher: B * 2 channels * 4 heads
};

struct LastRowArgs {
  const float* x;       // [B*2][T][256]
  const int* bn;
  float* xlast;         // [B*2][256] row n-1 of every (stream, channel)
  float* xnlast;        // LayerNorm(xlast; gamma, beta)
  const float* gamma;
  const float* beta;
  int B, T;
};

struct HeadArgs {
  const float* x;       // [B*2][T][256] stereo tower outputs (a = ch 0, b = ch 1); [B*2][256] if x_last_only
  const float* o;       // [B*2][T][256] ar_channel outputs
  const float* e;       // [B*2][256]
  const int* bn;
  const int* ids;
  int* frames_seen;     // incremented when non-null
  const float* waT;     // [256 k][256 j]
  const float* wbT;
  const float* cg;
  const float* cb;
  const float* hwT;     // [256 k][256 j]
  const float* hb;
  const float* vw;      // [256]
  const float* vb;      // [1]
  const float* aw;      // [8][256]
  const float* ab;      // [8]
  float* out;           // [B][out_stride]
  int B, T, mode, out_stride, x_last_only;
};

hipError_t launch_conv0(const Conv0Args& a, int B, hipStream_t st);
hipError_t launch_lstm(const LstmArgs& a, hipStream_t st);
hipError_t launch_ring_append(const GatherArgs& a, hipStream_t st);
hipError_t launch_gather_ln(const GatherArgs& a, hipStream_t st);
hipError_t launch_attention(const AttnArgs& a, int B, hipStream_t st);
hipError_t launch_attention_f16x3(const AttnArgs& a, int B, hipStream_t st);   // split-precision variant (attention_f16x3.hip), same arguments
hipError_t launch_attention_proj_f16x3(const AttnProjArgs& a, int B, hipStream_t st);   // split path, layers >= 1: Q|K|V projected inside (attention_proj_f16x3.hip)
hipError_t launch_gather_last_ln(const LastRowArgs& a, hipStream_t st);
hipError_t launch_ln_rows(const float* x, float* y, const float* gamma, const float* beta, int rows, hipStream_t st);
hipError_t launch_attention_last(const AttnArgs& a, int B, hipStream_t st);
hipError_t launch_head(const HeadArgs& a, hipStream_t st);
