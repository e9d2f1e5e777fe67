// conv8.hip: convolutions on 8x8 maps with the consumer's GroupNorm in the epilogue (see the header there).
#pragma once
#include "../../ishapediting_amd/csrc/common.h"
#include "../../ishapediting_amd/csrc/norm.h"

struct Conv8Args {
  // operands
  const half_t* X = nullptr;     // activation the 9-tap (or, taps == 1, the only) part reads: [N*64][ldx] fp16, Cin channels
  const half_t* X2 = nullptr;    // taps == 9 only: second source of a folded 1x1 convolution, [N*64][ldx2], K2 channels
  const half_t* W8 = nullptr;    // fragment-packed weights (conv8_pack)
  int Cin = 0, K2 = 0, Cout = 0, taps = 9;
  int ldx = 0, ldx2 = 0;
  int N = 1;                     // images (64 pixels each)
  const float* bias = nullptr;
  const float* bias2 = nullptr;
  const half_t* res = nullptr;   // residual [N*64][ldr] or null
  int ldr = 0;
  half_t* y = nullptr;           // convolution output [N*64][ldy], fp16 (what skip connections / the backward pass read)
  int ldy = 0;
  // the consumer's GroupNorm32 (gamma == null: none).  This tensor's channels are channels [norm_c0, norm_c0 + Cout) of the
  // consumer's input (norm_c0 > 0: it is the second part of a skip concatenation), norm_C channels in all, cpg per group
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* emb = nullptr;    // FiLM rows (scale at c, shift at norm_C + c), images emb_ld floats apart
  int emb_ld = 0, film = 0, act = 0;
  int cpg = 0, group_base = 0, norm_c0 = 0, norm_C = 0;
  half_t* a_out = nullptr;       // act(film(gn(y))) [N*64][lda], written at channel norm_c0 + n
  int lda = 0;
  float* stats_out = nullptr;    // [N][32][2] (mean, rstd) of the consumer's groups this tensor covers
  unsigned long long* rec = nullptr;   // zeroed rendezvous record [N][32][GN_REC_STRIDE]
  unsigned* status = nullptr;    // set by the launcher
  int spin_limit = 0;
};
bool conv8_shape_ok(int H, int W, int Cin, int K2, int Cout, int taps);
size_t conv8_packed_halfs(int Cout, int Cin, int K2, int taps);
// src: the row-major packed operand [rows >= Cout][ld] (k = tap * Cin + c, then the K2 columns of a folded 1x1 part)
int conv8_pack(const half_t* src, int ld, int Cout, int Cin, int K2, int taps, half_t* dst, hipStream_t s);
int conv8_launch(const Conv8Args& a, hipStream_t s);
