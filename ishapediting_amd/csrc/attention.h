#pragma once
#include "common.h"

struct AttnArgs {
  const half_t* qkv = nullptr;   // [N][T][3C], head h at [h*3d,(h+1)*3d): q | k | v
  half_t* out = nullptr;         // [N][T][C] attention output a (written by forward, read by backward)
  const half_t* dout = nullptr;  // backward: gradient of a
  half_t* dqkv = nullptr;        // backward: gradient of qkv
  float* lse = nullptr;          // [N*heads][T]
  float* Dbuf = nullptr;         // backward scratch [N*heads][T]
  int N = 1, T = 0, C = 0, heads = 0, d = 0;
  float alpha = 1.f;             // 1/sqrt(d)  (= s*s with s = d^-1/4, unet.py:348-351)
  int xcd_map = 1;               // XCD-aware workgroup -> (tile, head) mapping (attention.hip: attn_xcd_item); set by the launchers
  // forward only, proj_parts > 0 (round 5): proj_out (unet.py:304) leaves the attention launch as per-head fp32 K slices
  // slices[h][n * T + q][c] = sum_d a[q][h*d + d'] Wproj[c][h*d + d'] -- the consuming GroupNorm pass adds the heads up with proj's
  // bias and the residual (common.h SlabSrc).  Every query tile is then computed by `proj_parts` workgroups that differ only in
  // the C / proj_parts output columns of the slice they produce (the attention itself is recomputed: T <= 256, it is latency).
  const half_t* wproj = nullptr; // [C rows (c_out, padded)][ldp]
  int ldp = 0;
  float* slices = nullptr;       // [heads][N * T][C]
  int proj_parts = 0;
};
// proj_parts with which attn_forward_launch emits proj_out as per-head slices for this shape, 0 = it cannot
// (ISHAP_ATTN_PROJ: 0 = never, the default -- measured, no gain; 1 = wherever built; 256 = only up to 256 tokens)
int attn_proj_parts(int N, int T, int C, int d);
int attn_forward_launch(const AttnArgs& a, hipStream_t s);
int attn_backward_launch(const AttnArgs& a, hipStream_t s);

// Round 5: the body of an AttentionBlock on an 8x8 map (T = 64 tokens = one tile, head width 64) after its GroupNorm as ONE launch:
// qkv = Wqkv xn + b (gd/unet.py:299-301), QKVAttentionLegacy per head (:337-354), and proj_out as per-head fp32 K slices that the
// consuming GroupNorm pass adds up with proj's bias and the residual (the pending-tensor mechanism, common.h SlabSrc).
struct Attn8Args {
  const half_t* xn = nullptr;     // [N][64][C] GroupNorm output
  const half_t* wqkv = nullptr;   // [3C rows (padded)][C], row = qkv channel (legacy order: head h at rows [h*192, h*192 + 192): q | k | v)
  const float* bqkv = nullptr;    // [3C]
  const half_t* wproj = nullptr;  // [C rows (padded)][C]
  half_t* qkv = nullptr;          // [N][64][3C]   kept for the backward pass
  half_t* aout = nullptr;         // [N][64][C]    kept for the backward pass
  float* lse = nullptr;           // [N][heads][64]
  float* slices = nullptr;        // [heads][N * 64][C] fp32: slice h = a_h Wproj[:, h*64 .. h*64+63]^T
  unsigned* flags = nullptr;      // [N][heads][16], zero before the launch
  unsigned* status = nullptr;     // device status word (common.h)
  int spin_limit = 1 << 22;
  int N = 1, C = 0, heads = 0;
  float alpha = 1.f;              // 1 / sqrt(64)
};
bool attn8_applicable(int N, int T, int C, int d);
int attn8_fused_launch(const Attn8Args& a, hipStream_t s);

// The backward twin (round 5): d proj_out -> attention backward (both roles) -> d qkv of an 8x8-map AttentionBlock in one launch.
// No workgroup waits for another: every one of a head's 12 workgroups recomputes the head's dA (a 64 x 64 x C product) and its
// attention backward (T = 64: one tile) itself and differs only in the columns of the input-gradient slice it produces.
struct Attn8BwdArgs {
  const half_t* dy = nullptr;      // [N][64][C] gradient arriving at proj_out's output
  const half_t* wprojT = nullptr;  // proj_out's input-gradient operand [C rows (c_in)][ldp] (ConvW::wT)
  const half_t* wqkvT = nullptr;   // qkv's input-gradient operand [C rows (c_in)][ldq]
  int ldp = 0, ldq = 0;
  const half_t* qkv = nullptr;     // forward tensors (AttnSaved)
  const half_t* aout = nullptr;
  const float* lse = nullptr;
  half_t* dA = nullptr;            // [N][64][C] scratch: every part of a head writes the same values
  half_t* dqkv = nullptr;          // [N][64][3C] scratch, likewise
  float* slices = nullptr;         // [heads][N * 64][C] fp32: slice h = dqkv_h Wqkv[h*192 .. +191][:]  (the gradient at the GroupNorm output)
  int N = 1, C = 0, heads = 0;
  float alpha = 1.f;
};
bool attn8_bwd_applicable(int N, int T, int C, int d);
int attn8_bwd_fused_launch(const Attn8BwdArgs& a, hipStream_t s);
