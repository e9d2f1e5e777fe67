#pragma once
#include "common.h"
#include <algorithm>

enum { DDPM_MODE_SQRT_VAR = 0, DDPM_MODE_EXP_HALF_LOGVAR = 1, DDPM_MODE_VARIANCE_NOISE = 2, DDPM_MODE_DDIM = 3 };

struct DdpmStepArgs {
  const float* x = nullptr;          // [N][C][HW]
  const float* model_out = nullptr;  // [N][2C][HW]  (eps | v)
  const float* noise = nullptr;      // [N][C][HW]: randn, or variance_noise in mode 2; null = zeros
  const float* variance_in = nullptr;  // optional override (p_sample_guidance `variance=`)
  float* sample = nullptr;
  float* pred_xstart = nullptr;
  float* variance = nullptr;
  float* mean = nullptr;
  int N = 1, C = 0, HW = 0;
  float min_log = 0, max_log = 0, sqrt_recip = 0, sqrt_recipm1 = 0, coef1 = 0, coef2 = 0;
  float nonzero = 1.f;
  float ddim_a = 0, ddim_b = 0, ddim_sigma = 0;   // mode 3: sqrt(abar_prev), sqrt(1 - abar_prev - sigma^2), sigma
  int clip = 1;
  int mode = DDPM_MODE_SQRT_VAR;
  // guided update folded into the step (round 5): guided = sample + variance * (guide_scale * (guide_grad * guide_mul))
  // (drag_utils.py:384-392) -- the arithmetic of guided_update_kernel on the values this kernel has in registers
  const float* guide_grad = nullptr;   // [N][C][HW] or null
  const float* guide_mul = nullptr;    // optional device scalar (1 / loss scale of the fp16 backward pass)
  float guide_scale = 0.f;
  float* guided = nullptr;
  // in-kernel noise (include/ishap.h, ishap_step_coefs::rng): used when rng != 0 and noise == nullptr
  int rng = 0;
  unsigned long long rng_seed = 0, rng_offset = 0;
  float* noise_out = nullptr;
};
int ddpm_step_launch(const DdpmStepArgs& a, hipStream_t s);
int guided_update_launch(const float* sample, const float* variance, const float* grad, float* out, float scale,
                         const float* grad_mul_dev, long long n, hipStream_t s);
int axpby_launch(const float* x, const float* y, float* o, float a, float b, long long n, hipStream_t s);
