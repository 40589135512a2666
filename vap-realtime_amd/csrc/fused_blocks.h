// Vertically fused transformer blocks (see fused_blocks.hip).
#pragma once
#include "common.h"

struct FfnArgs {
  const float* xn;     // [M][256] LN_ffn(xmid)
  const float* xmid;   // [M][256] residual stream before the FFN
  const float* w0;     // [768][256]
  const float* w3;     // [256][768]
  float* xout;         // [M][256] layer output
  const float* ln_g;   // next layer's ln_self (used when wqkv != null)
  const float* ln_b;
  const float* wqkv;   // [n_qkv][256] next layer's self projections (null: skip)
  float* qkv;          // [M][n_qkv]
  const float* wkvx;   // [512][256] next layer's cross K,V projections (null: skip)
  float* kvx;          // [M][512]
  int n_qkv;           // 768 (Q,K,V) or 512 (K,V only: pass wqkv + 256*256)
  int M;
};

hipError_t launch_ffn_block(const FfnArgs& a, hipStream_t st);
