// fp32-MFMA GEMM with fused epilogues — declaration shared by the engine and the C ABI.
#pragma once
#include "common.h"

enum GemmEpi {
  EPI_STORE = 0,         // C = acc (+ bias)
  EPI_GELU = 1,          // C = gelu(acc)
  EPI_RESID = 2,         // C = resid + acc
  EPI_RESID_LN = 3,      // C = resid + acc ; C2 = LayerNorm(C; gamma, beta)
  EPI_CN_RELU = 4,       // C = relu(ChannelNorm(acc + bias; gamma, beta))   (unbiased variance)
  EPI_BIAS_LN_GELU = 5,  // C = gelu(LayerNorm(acc + bias; gamma, beta))
};

struct GemmArgs {
  const float* A;
  RowMap am;
  const float* W;  // [N][K], K contiguous (nn.Linear layout)
  int M, N, K;
  float* C;
  RowMap cm;
  const float* bias;
  const float* gamma;
  const float* beta;
  const float* resid;
  RowMap rm;
  float* C2;
  RowMap c2m;
  int split;   // 1: fp32-accurate 3-term split products on the f16 matrix cores (operands split while staged), see gemm_f32.hip
  const float* W16;   // split only, optional: pre-split copy of W (weights.split16_pack: same [N][K] addressing, 16 bytes = 4 hi + 4 lo halves
                      // of 2^8 w) — staged as it is instead of converting W in every k-tile of every workgroup; null: convert on the fly
};

// tile_rows: 0 = choose from M, else 32 / 64 / 128 rows per workgroup.
hipError_t launch_gemm_f32(const GemmArgs& g, int epi, int tile_rows, hipStream_t stream);
