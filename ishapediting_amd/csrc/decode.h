#pragma once
#include "common.h"
#include <algorithm>

struct DecodeArgs {
  const float* planes = nullptr;   // [3][S][S][32] un-normalised, channels-last
  int S = 0;
  const float* B = nullptr;        // net.0._B [32][64]
  const float* W1 = nullptr;       // net.1.weight [128][128]
  const float* b1 = nullptr;
  const float* W2 = nullptr;       // net.3.weight
  const float* b2 = nullptr;
  const float* w3 = nullptr;       // net.5.weight [1][128]
  const float* b3 = nullptr;       // [1]
  const float* coords = nullptr;   // [npts][3] or null for the dense grid
  const float* lin = nullptr;      // grid axis values [res]
  int res = 0;
  long long npts = 0;
  float* out = nullptr;            // [npts] logits
};
int triplane_decode_launch(const DecodeArgs& a, hipStream_t s);

struct DecBwdArgs {
  const float* planes = nullptr;   // [3][S][S][32]
  int S = 0;
  const float *B = nullptr, *W1 = nullptr, *b1 = nullptr, *W2 = nullptr, *b2 = nullptr, *w3 = nullptr, *b3 = nullptr;
  const float *W1T = nullptr, *W2T = nullptr;   // transposed copies (coalesced forward mat-vecs)
  const float* coords = nullptr;   // [npts][3]
  const float* gt = nullptr;       // [npts] occupancy targets
  long long npts = 0;
  float* dplanes = nullptr;        // [3][S][S][32] d(-BCE)/d planes (zeroed by the launch)
  float* loss = nullptr;           // [1] = -BCEWithLogits mean
  float* logits = nullptr;         // optional [npts]
};
int decode_points_bwd_launch(const DecBwdArgs& a, hipStream_t s);
int x0_grad_launch(const float* dplanes, const float* rng, const float* x, const float* model_out, float sr, float srm1,
                   int clip, int S, float* g_direct, float* cot_out, hipStream_t s);
int planes_prepare_launch(const float* latent, const float* rng, const float* mid, float* planes, int S, hipStream_t s);
