// One element of the GroupNorm input-gradient:  y = act(film(GN(x)))  ->  dyh = upstream * act'(pre) * (1+scale) * gamma
// and the normalised input xhat.  Shared by norm_bwd.hip (stand-alone partial/apply kernels) and by the implicit-GEMM
// epilogue / split-K reduce, which accumulate sum(dyh) and sum(dyh*xhat) per channel while the gradient tile is still
// in registers.  Rounding points mirror the forward kernel (norm.hip) so the recomputed pre-activation is the forward's.
#pragma once
#include "common.h"

__device__ __forceinline__ float gnb_rh(float v) { return (float)(half_t)v; }
__device__ __forceinline__ float gnb_silu_grad(float v) {
  float sg = 1.f / (1.f + __expf(-v));
  return sg * (1.f + v * (1.f - sg));
}
// emb_sc / emb_sh: the raw FiLM row entries (only read when film)
__device__ __forceinline__ void gn_bwd_term(float up, float x, float mu, float rs, float gam, float bet, float emb_sc,
                                            float emb_sh, bool film, bool act, float& dyh, float& xhat) {
  xhat = (x - mu) * rs;
  float u = up, mult = gam;
  if (film || act) {
    float pre = gnb_rh(xhat * gam + bet);
    if (film) {
      const float sc = gnb_rh(1.f + gnb_rh(emb_sc));
      const float sh = gnb_rh(emb_sh);
      pre = gnb_rh(gnb_rh(pre * sc) + sh);
      mult *= sc;
    }
    if (act) u *= gnb_silu_grad(pre);
  }
  dyh = u * mult;
}
