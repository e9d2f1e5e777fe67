// Activation arithmetic shared by every GroupNorm route (norm.hip, norm_local.hip, norm_bwd.hip via gn_bwd_terms.h, the
// implicit-GEMM epilogues): one definition, so that the routes round identically.
// Reference: GroupNorm32 output cast back to half (gd/nn.py:16-18), FiLM `h * (1 + scale) + shift` and SiLU on half tensors
// (gd/unet.py:245-252).  Round 4: these kernels turned out VALU-bound, not HBM-bound (a 128x128x256 apply pass: ~27
// instructions per element, 7.6 us of issue time per SIMD against 3.4 us of HBM time): the IEEE division of the sigmoid
// (v_div_scale / v_rcp / 4 v_fma / v_div_fmas / v_div_fixup) became v_rcp_f32 (1 ulp in fp32, far below the fp16 rounding that
// follows), and FiLM's product runs in fp16 arithmetic -- a correctly rounded fp16 product IS fp16(fp32 product) (the
// 22-bit product is exact in fp32).  The sum stays fp16(fp32 sum), the order the reference's half tensors take on the CPU
// the golden fixtures were generated on (an fp16 add would round once instead of twice: a different value on rare ties).
#pragma once
#include "common.h"

// fp16((x - mean) * rstd * gamma + beta): ONE rounding sequence (no contraction) for every route -- the apply kernels, the
// group-local kernels, the convolution epilogue that applies the norm itself and the backward pass's recomputation agree bitwise
__device__ __forceinline__ half_t gn_affine(float x, float mean, float rstd, float gamma, float beta) {
#pragma clang fp contract(off)
  const float xh = (x - mean) * rstd;
  return (half_t)(xh * gamma + beta);
}
__device__ __forceinline__ float gn_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float gn_silu(float v) { return v * gn_sigmoid(v); }
// fp16(fp16(y * sc) + sh) on fp16 operands (sc = fp16(1 + fp16(scale)), sh = fp16(shift), prepared by the caller)
__device__ __forceinline__ half_t gn_film(half_t y, half_t sc, half_t sh) {
#pragma clang fp contract(off)
  const half_t t = y * sc;
  return (half_t)((float)t + (float)sh);
}
