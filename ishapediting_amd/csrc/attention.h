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
