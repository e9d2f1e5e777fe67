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
int planes_prepare_launch(const float* latent, const float* rng, const float* mid, float* planes, int S, hipStream_t s);
