// One reverse-diffusion step as a single coalesced pass (HBM-bound elementwise).
// Reference: guided_diffusion/gaussian_diffusion.py:265-279 (learned-range variance),
// :333-338 (eps -> x0, clamp), :216-219 (posterior mean), :498-510 (sample variants of
// p_sample_guidance), :443 (p_sample uses exp(0.5*logvar)).  The reference materialises
// 6-8 full tensors per step; here x, eps, v, noise are read once and each requested output
// is written once (float4 per lane).
#include "ddpm.h"

// Philox4x32-10 (J. Salmon, M. Moraes, R. Dror, D. Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator
// behind torch's CUDA randn as well, here with a counter layout of its own): ten rounds of two 32 x 32 -> 64 multiplies.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
// four standard normals for the 4-float vector `vec` of the tensor: two Box-Muller pairs
__device__ __forceinline__ f32x4 step_noise4(unsigned long long vec, unsigned long long seed, unsigned long long offset) {
  const unsigned long long key = seed ^ 0x9E3779B97F4A7C15ull;      // not torch's own stream of the same seed
  unsigned w[4];
  philox4x32_10((unsigned)vec, (unsigned)(vec >> 32), (unsigned)offset, (unsigned)(offset >> 32), (unsigned)key, (unsigned)(key >> 32), w);
  f32x4 n;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u0 = ((float)(w[2 * h] >> 8) + 0.5f) * 0x1p-24f, u1 = ((float)(w[2 * h + 1] >> 8) + 0.5f) * 0x1p-24f;   // in (0, 1)
    const float rad = sqrtf(-2.f * logf(u0));
    float sn, cs;
    sincosf(6.283185307179586f * u1, &sn, &cs);
    n[2 * h] = rad * cs;
    n[2 * h + 1] = rad * sn;
  }
  return n;
}

__global__ __launch_bounds__(256) void ddpm_step_kernel(DdpmStepArgs a) {
  const float gm = (a.guided && a.guide_mul) ? *a.guide_mul : 1.f;
  const long long per_img = (long long)a.C * a.HW;           // floats per image in x
  const long long nvec = (long long)a.N * per_img / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const long long n = e / per_img;
    const long long r = e - n * per_img;
    const f32x4 x = *reinterpret_cast<const f32x4*>(a.x + e);
    const f32x4 eps = *reinterpret_cast<const f32x4*>(a.model_out + n * 2 * per_img + r);
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.model_out + n * 2 * per_img + per_img + r);
    // every operand as an UNCONDITIONAL load (an absent one reads x again and is not used): `if (p) v = p[e]` closes with an
    // s_waitcnt vmcnt(0) per block -- four dependent round trips through a kernel that is one streaming pass (round 6)
    const f32x4 nz_l = *reinterpret_cast<const f32x4*>((a.noise ? a.noise : a.x) + e);
    const f32x4 var_l = *reinterpret_cast<const f32x4*>((a.variance_in ? a.variance_in : a.x) + e);
    const f32x4 gr_l = *reinterpret_cast<const f32x4*>((a.guided ? a.guide_grad : a.x) + e);
    f32x4 nz = {0.f, 0.f, 0.f, 0.f};
    if (a.noise) nz = nz_l;
    else if (a.rng) {
      nz = step_noise4((unsigned long long)i, a.rng_seed, a.rng_offset);
      if (a.noise_out) *reinterpret_cast<f32x4*>(a.noise_out + e) = nz;
    }
    f32x4 var_in = {0.f, 0.f, 0.f, 0.f};
    if (a.variance_in) var_in = var_l;
    f32x4 gr = {0.f, 0.f, 0.f, 0.f};
    if (a.guided) gr = gr_l;
    f32x4 o_sample, o_x0, o_var, o_mean, o_guided;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float frac = (v[k] + 1.f) / 2.f;
      float logvar = frac * a.max_log + (1.f - frac) * a.min_log;
      float var = expf(logvar);
      float x0 = a.sqrt_recip * x[k] - a.sqrt_recipm1 * eps[k];
      if (a.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
      float mean = a.coef1 * x0 + a.coef2 * x[k];
      float s;
      if (a.mode == DDPM_MODE_DDIM) {
        // ddim_sample (:654-705): eps re-derived from the (clipped) x0 (:350-354), Equation 12
        const float e2 = (a.sqrt_recip * x[k] - x0) / a.sqrt_recipm1;
        s = (x0 * a.ddim_a + a.ddim_b * e2) + a.nonzero * a.ddim_sigma * nz[k];
      } else if (a.mode == DDPM_MODE_VARIANCE_NOISE) s = mean + nz[k];                           // :498-499
      else if (a.mode == DDPM_MODE_EXP_HALF_LOGVAR) s = mean + a.nonzero * expf(0.5f * logvar) * nz[k];   // :443
      else s = mean + a.nonzero * sqrtf(a.variance_in ? var_in[k] : var) * nz[k];          // :503 / :508
      o_sample[k] = s; o_x0[k] = x0; o_var[k] = var; o_mean[k] = mean;
      // the dict's "variance" is the caller's override when one was given (gaussian_diffusion.py:510), else the learned one
      o_guided[k] = s + (a.variance_in ? var_in[k] : var) * (a.guide_scale * (gr[k] * gm));
    }
    if (a.guided) *reinterpret_cast<f32x4*>(a.guided + e) = o_guided;
    if (a.sample) *reinterpret_cast<f32x4*>(a.sample + e) = o_sample;
    if (a.pred_xstart) *reinterpret_cast<f32x4*>(a.pred_xstart + e) = o_x0;
    if (a.variance) *reinterpret_cast<f32x4*>(a.variance + e) = o_var;
    if (a.mean) *reinterpret_cast<f32x4*>(a.mean + e) = o_mean;
  }
}

int ddpm_step_launch(const DdpmStepArgs& a, hipStream_t s) {
  ISHAP_REQUIRE(((long long)a.C * a.HW) % 4 == 0, "C*H*W must be a multiple of 4");
  long long nvec = (long long)a.N * a.C * a.HW / 4;
  int blocks = (int)std::min<long long>((nvec + 255) / 256, 2048);
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(blocks), dim3(256), 0, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// img = sample + variance * (scale * grad * grad_mul)        drag_utils.py:384-392
//   grad_mul carries 1/loss-scale of the fp16 backward pass; read from device memory when given
__global__ void guided_update_kernel(const float* __restrict__ sample, const float* __restrict__ variance,
                                     const float* __restrict__ grad, float* __restrict__ out, float scale,
                                     const float* __restrict__ grad_mul_dev, long long nvec) {
  const float gm = grad_mul_dev ? *grad_mul_dev : 1.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    f32x4 s = reinterpret_cast<const f32x4*>(sample)[i];
    f32x4 v = reinterpret_cast<const f32x4*>(variance)[i];
    f32x4 g = reinterpret_cast<const f32x4*>(grad)[i];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = s[k] + v[k] * (scale * (g[k] * gm));
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}
int guided_update_launch(const float* sample, const float* variance, const float* grad, float* out, float scale,
                         const float* grad_mul_dev, long long n, hipStream_t s) {
  ISHAP_REQUIRE(n % 4 == 0, "n % 4");
  long long nvec = n / 4;
  int blocks = (int)std::min<long long>((nvec + 255) / 256, 2048);
  hipLaunchKernelGGL(guided_update_kernel, dim3(blocks), dim3(256), 0, s, sample, variance, grad, out, scale, grad_mul_dev, nvec);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// forward-noising chain of ddpm_inversion (gaussian_diffusion.py:520-522): x' = sqrt(cof) x + sqrt(1-cof) eps
__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ o, float a,
                             float b, long long nvec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    f32x4 xv = reinterpret_cast<const f32x4*>(x)[i], yv = reinterpret_cast<const f32x4*>(y)[i];
    reinterpret_cast<f32x4*>(o)[i] = a * xv + b * yv;
  }
}
int axpby_launch(const float* x, const float* y, float* o, float a, float b, long long n, hipStream_t s) {
  ISHAP_REQUIRE(n % 4 == 0, "n % 4");
  long long nvec = n / 4;
  int blocks = (int)std::min<long long>((nvec + 255) / 256, 2048);
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks), dim3(256), 0, s, x, y, o, a, b, nvec);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
