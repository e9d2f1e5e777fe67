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
};
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
  int phases = 3;                 // set by the launcher: 3 = the whole block in one launch (flags); 1 = qkv only; 2 = attention + proj_out only
  int N = 1, C = 0, heads = 0;
  float alpha = 1.f;              // 1 / sqrt(64)
};
bool attn8_applicable(int N, int T, int C, int d);
// one_launch: the sequence holds the device's rendezvous tenancy (workgroups may wait for each other); false: two launches of the
// same kernel, bitwise the same results
int attn8_fused_launch(const Attn8Args& a, hipStream_t s, bool one_launch);

