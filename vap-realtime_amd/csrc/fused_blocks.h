// Vertically fused transformer blocks (see fused_blocks.hip).
#pragma once
#include "common.h"

struct FfnArgs {
  const float* xmid;   // [M][256] residual stream before the FFN (LayerNorm(ln_ffn) is applied on load)
  const float* lnf_g;  // ln_ffnetwork weight / bias
  const float* lnf_b;
  const float* w0f;    // W0 [768][256], fragment-major: 3 column chunks
  const float* w3f;    // W3 [256][768], fragment-major: 3 k chunks
  float* xout;         // [M][256] layer output
  const float* ln_g;   // next layer's ln_self (used when wqkvf or xn_out is set)
  const float* ln_b;
  float* xn_out;       // [M][256] LayerNorm(x_out; ln_g, ln_b) rows, or null (the fused last-row block consumes these)
  const float* wqkvf;  // next layer's self projections, fragment-major chunks (null: skip)
  float* qkv;          // [M][256 * n_qkv_chunks]
  const float* wkvxf;  // next layer's cross K,V projections, 2 fragment-major chunks (null: skip)
  float* kvx;          // [M][512]
  int n_qkv_chunks;    // 3 (Q,K,V) or 2 (K,V only: pass wqkvf + 65536)
  int M;
  float hid_scale;     // split-precision path only: static power-of-two scale (<= 1) of the GELU hidden row, from the weights' bound on
                       // |gelu(h)| (engine.hip, vapx_create); 0 = 1
#ifdef VAPX_TRACE
  unsigned long long* trace;   // debug build: optional [grid][32] s_memtime stamps of workgroup phases (env VAPX_FFN_TRACE)
#endif
  // Long-window path (T > 64): the attention output projection rides in front of the block instead of a separate GEMM:
  //   xmid = resid + att . Wproj^T   (the residual stream), then LayerNorm as usual.  Mode 2 writes it to xmid_out (the next block's
  //   `resid`); in mode 1 its only consumer is the block itself, so it stays in the accumulators and xmid_out is NOT written.
  // mode 0: xmid is read from global (fused attention block wrote it).  mode 1: pre-projection + the whole block.
  // mode 2: pre-projection + LN(ln_g, ln_b) + the n_qkv_chunks contractions of wqkvf -> qkv only (self-attention half of a
  //         stereo layer: LN_src + cross-attention query projection), no FFN.
  int mode;
  const float* att;    // [M][256] attention output (heads merged)
  const float* wprojf; // output projection, fragment-major
  const float* resid;  // [M][256]
  float* xmid_out;     // [M][256] (mode 2)
  // mode 1 on layer 0 of a long window: `resid` is the per-stream embedding RING (slab = slot*2+channel, logical row i of the
  // window in ring slot (i + resid_rot[b]) % resid_T) instead of a chronological [M][256] buffer
  const int* resid_rot; // [B] or null
  const int* resid_ids; // [B] stream slots (null: identity)
  int resid_T;
};

struct AttnBlockArgs {
  const float* q;       // row (bc*T + i), stride ldq, head h at column 64h
  const float* k;
  const float* v;       // rows of channel (bc ^ swap_kv), stride ldkv
  const int* bn;
  const float* wprojf;  // output projection, fragment-major 256x256
  const float* resid;   // [B*2*T][256]
  const float* ln_g;    // ln_src_attn (only read with wqxf)
  const float* ln_b;
  float* xmid;          // resid + att.Wproj^T
  const float* wqxf;    // optional: cross-attention query projection (fragment-major), null to skip (then no LayerNorm either:
                        // the FFN block normalises xmid itself)
  float* qx;            // [B*2*T][256]
  const int* ring_rot;  // [B] or null.  Non-null: q/k/v/resid are per-stream RINGS (slab = slot*2+channel, logical
                        // row i in ring slot (i + ring_rot[b]) % T) instead of chronological batch buffers
  const int* ids;       // [B] stream slots (null: identity); only used with ring_rot
  int T, ldq, ldkv, swap_kv;
  int split;            // 1: wprojf / wqxf are f16 hi/lo fragment copies, projections run as 3-term split products
#ifdef VAPX_TRACE
  unsigned long long* trace;   // debug build: optional [grid][32] s_memtime stamps of workgroup phases (env VAPX_ATTN_TRACE)
#endif
};

struct ConvTailArgs {
  const float* h1;     // [B*2][P1+2][256] conv1 output, channels-last, one zero guard row each side
  const float* w2f;    // conv2 weight: 4 taps x fragment-major 256x256 block (tap t = W[:, :, t])
  const float* w3f;
  const float* w4f;
  const float *b2, *g2, *be2;   // conv bias, ChannelNorm weight / bias
  const float *b3, *g3, *be3;
  const float *b4, *g4, *be4;
  float* z;            // [B*2][ncpc][256] conv4 positions 1..ncpc
  int P1, ncpc;
};

struct LastBlockArgs {
  const float* x;      // [B*2][T][256] raw input rows of the last layer (residual of row n-1; cross-attention keys / values)
  const float* xn;     // [B*2][T][256] LN_self(last layer)(x) of every row (emitted by the previous layer's FFN block)
  const int* bn;       // [B] valid rows
  const float* wf;     // "L3.last16": Wq, Wk^T (per head), Wv, Wproj, Wq_x, Wk_x^T, Wv_x, Wproj_x, W0 x3, W3 x3; 16x16x4 fragment-major
  const float *ln_src_g, *ln_src_b, *ln_ffn_g, *ln_ffn_b;
  float* out;          // [B*2][256] layer output at the newest row
  int B, T;
};

bool conv_tail_supported(int P1, int ncpc);
hipError_t launch_last_block(const LastBlockArgs& a, hipStream_t st);
hipError_t launch_conv_tail(const ConvTailArgs& a, int B, hipStream_t st);
hipError_t launch_attn_block(const AttnBlockArgs& a, int B, hipStream_t st);   // T <= 64 only
hipError_t launch_ffn_block(const FfnArgs& a, hipStream_t st);
hipError_t launch_ffn_block_f16x3(const FfnArgs& a, hipStream_t st);   // w0f/w3f/wqkvf/wkvxf point to the *h (f16 hi/lo) copies
