// Vertically fused transformer blocks (see fused_blocks.hip).
#pragma once
#include "common.h"

struct FfnArgs {
  const float* xn;     // [M][256] LN_ffn(xmid)
  const float* xmid;   // [M][256] residual stream before the FFN
  const float* w0f;    // W0 [768][256], fragment-major: 3 column chunks
  const float* w3f;    // W3 [256][768], fragment-major: 3 k chunks
  float* xout;         // [M][256] layer output
  const float* ln_g;   // next layer's ln_self (used when wqkv != null)
  const float* ln_b;
  const float* wqkvf;  // next layer's self projections, fragment-major chunks (null: skip)
  float* qkv;          // [M][256 * n_qkv_chunks]
  const float* wkvxf;  // next layer's cross K,V projections, 2 fragment-major chunks (null: skip)
  float* kvx;          // [M][512]
  int n_qkv_chunks;    // 3 (Q,K,V) or 2 (K,V only: pass wqkvf + 65536)
  int M;
};

hipError_t launch_ffn_block(const FfnArgs& a, hipStream_t st);
