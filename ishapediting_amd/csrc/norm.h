#pragma once
#include "common.h"

static inline int gn_rows_per_block(int HW) {
  int r = HW / 256;
  if (r < 8) r = 8;
  if (r > HW) r = HW;
  return r;
}
static inline size_t gn_partial_floats(int N, int HW, int C) {
  return (size_t)N * (HW / gn_rows_per_block(HW)) * C * 2;
}
// stats[n][32][2] = (mean, rstd); partial is scratch of gn_partial_floats()
int gn_stats_launch(const half_t* x, float* partial, float* stats, int N, int HW, int C, hipStream_t s);

struct GnApplyArgs {
  const half_t* x = nullptr;     // [N][H*W][C]
  half_t* out = nullptr;         // [N][HWo][C]  (or [N][HW][3C] when split)
  half_t* xpool = nullptr;       // pool: pooled raw x (may be null)
  const float* stats = nullptr;  // [N][32][2]
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* emb = nullptr;    // film: per image (scale[C] | shift[C]) fp32, images emb_ld floats apart
  int emb_ld = 0;
  int N = 1, H = 0, W = 0, C = 0;
  int film = 0, act = 1, pool = 0, split = 0;
};
int gn_apply_launch(const GnApplyArgs& a, hipStream_t s);
