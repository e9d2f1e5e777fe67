// One element of the GroupNorm input-gradient:  y = act(film(GN(x)))  ->  dyh = upstream * act'(pre) * (1+scale) * gamma
// and the normalised input xhat.  Shared by norm_bwd.hip (stand-alone partial/apply kernels) and by the implicit-GEMM
// epilogue / split-K reduce, which accumulate sum(dyh) and sum(dyh*xhat) per channel while the gradient tile is still
// in registers.  Rounding points mirror the forward kernel (norm.hip) so the recomputed pre-activation is the forward's.
#pragma once
#include "common.h"
#include "gn_act.h"

__device__ __forceinline__ float gnb_rh(float v) { return (float)(half_t)v; }
// fp contract(off) in these helpers: they are inlined at several call sites of one kernel (a thread's first unit keeps dyh /
// xhat from the statistics pass, later units recompute them), and the sites must round identically whatever the compiler
// would fuse around them -- the group-local kernels' results must not depend on how many units a thread owns (parts = 1 vs > 1).
__device__ __forceinline__ float gnb_silu_grad(float v) {
#pragma clang fp contract(off)
  const float sg = gn_sigmoid(v);
  return sg * (1.f + v * (1.f - sg));
}
// emb_sc / emb_sh: the raw FiLM row entries (only read when film)
__device__ __forceinline__ void gn_bwd_term(float up, float x, float mu, float rs, float gam, float bet, float emb_sc,
                                            float emb_sh, bool film, bool act, float& dyh, float& xhat) {
#pragma clang fp contract(off)
  xhat = (x - mu) * rs;
  float u = up, mult = gam;
  if (film || act) {
    half_t preh = gn_affine(x, mu, rs, gam, bet);
    if (film) {
      const half_t sc = (half_t)(1.f + gnb_rh(emb_sc));
      preh = gn_film(preh, sc, (half_t)emb_sh);
      mult *= (float)sc;
    }
    if (act) u *= gnb_silu_grad((float)preh);
  }
  dyh = u * mult;
}
// dx = rstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat)), one rounding sequence for every call site
__device__ __forceinline__ float gn_bwd_dx(float rs, float dyh, float xh, float m1, float m2) {
#pragma clang fp contract(off)
  return rs * ((dyh - m1) - xh * m2);
}
