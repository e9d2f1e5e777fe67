#pragma once
#include "common.h"
#include <algorithm>

struct DragArgs {
  const half_t* edit = nullptr;    // current tap, NHWC fp16 [W*W][ld]
  const half_t* orig = nullptr;    // cached guidance tap, same layout
  int W = 0, ld = 0;               // feature-map side, channels of the tap
  int Cc = 0;                      // channels per plane after resize_feat_align (170 for a 512-ch tap)
  const int* chmap = nullptr;      // [3][Cc] -> channel of the tap
  const float* sources = nullptr;  // [B][3]
  const float* targets = nullptr;  // [B][3]
  int B = 0, r = 0;
  float voxel = 0.f, cof = 0.f;
  int l1 = 0;
  unsigned char* touched = nullptr;  // [3][W][W]
  int* nmask = nullptr;              // [1]
  unsigned char* chw = nullptr;      // [3][ld]: (plane, c) pairs mapped onto each tap channel (set up by drag_setup_launch)
  float* grad = nullptr;             // fp32 [W*W][ld] (d loss / d tap)
  long long* gfx = nullptr;          // scratch [W*W][ld]: the scatter accumulates here in 64-bit fixed point (DRAG_FX_SCALE)
  long long* acc = nullptr;          // [2] loss sums, 64-bit fixed point (DRAG_ACC_SCALE)
  float* loss = nullptr;             // [1]
};
// Integer atomics commute, so the scattered gradient and the loss are bitwise reproducible (fp32 atomics are not).
constexpr float DRAG_FX_SCALE = 17592186044416.f;    // 2^44: |sum| < 5e5, resolution 6e-14
constexpr float DRAG_ACC_SCALE = 16777216.f;         // 2^24
int drag_setup_launch(const DragArgs& a, hipStream_t s);        // touched bitmap + mask count (once per edit)
int drag_loss_grad_launch(const DragArgs& a, hipStream_t s);
int drag_loss_cotangent_launch(const DragArgs& a, half_t* cot, unsigned* bits, float* scale2, hipStream_t s);
int grad_to_scaled_f16_launch(const float* g, half_t* o, unsigned* bits, float* scale2, long long n, hipStream_t s);
