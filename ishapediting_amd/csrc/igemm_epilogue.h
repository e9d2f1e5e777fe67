// Shared epilogue of the implicit-GEMM kernels (igemm.hip register-staged, igemm2.hip LDS-DMA ring).
// A lane holds, for each (i, j) MFMA tile, output channels n..n+3 of one pixel m.  Besides bias / residual /
// layout handling, the epilogue can accumulate the per-channel sum and sum of squares of the fp16 values it
// stores (`stat_out`): these are the GroupNorm statistics of the NEXT layer (gd/nn.py:16-18), gathered here so no
// separate pass has to re-read the tensor.
#pragma once
#include "common.h"
#include "gn_bwd_terms.h"

template <int MT, int NT, int TMW, int TNW, int BN>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x4 (&acc)[NT][MT], int m0, int n0, int wm, int wn,
                                               int lane, int batch, int ks_id, float* lds_f, bool active = true) {
  // `active` = false: a loader wave of a producer/consumer kernel -- it owns no outputs but must take part in the
  // workgroup barriers of the statistics reduction below
  const int HW = a.H * a.W;
  float ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) { ssum[i][c] = 0.f; ssq[i][c] = 0.f; }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    if (!active) break;
    const int m = m0 + wm * TMW + j * 16 + (lane & 15);
    int n_img = 0, py = 0, px = 0;
    if (a.res_ups || a.out_mode == IG_OUT_NCHW_F32) {
      n_img = m / HW;
      const int p = m - n_img * HW;
      py = p / a.W;
      px = p - py * a.W;
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n0 + wn * TNW + i * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      f32x4 v = acc[i][j];
      if (a.ksplit > 1) {
        float* dst = a.ws + (((long long)ks_id * a.nbatch + batch) * a.M + m) * a.N + n;
        *reinterpret_cast<f32x4*>(dst) = v;
        continue;
      }
      v *= a.alpha;
      if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
      if (a.bias2) v += *reinterpret_cast<const f32x4*>(a.bias2 + n);
      if (a.res) {
        long long rrow = a.res_ups ? ((long long)n_img * (HW >> 2) + (py >> 1) * (a.W >> 1) + (px >> 1)) : m;
        half4 r = *reinterpret_cast<const half4*>(a.res + rrow * a.ldr + n);
        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
      }
      if (a.out_mode == IG_OUT_F16) {
        half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<half4*>((half_t*)a.out + (long long)batch * a.bso + (long long)m * a.ldo + n) = o;
        if (a.stat_out) {
#pragma unroll
          for (int c = 0; c < 4; ++c) { float f = (float)o[c]; ssum[i][c] += f; ssq[i][c] += f * f; }
        }
        if (a.gb_x) acc[i][j] = (f32x4){(float)o[0], (float)o[1], (float)o[2], (float)o[3]};   // the stored gradient, for the pass below
      } else if (a.out_mode == IG_OUT_F32) {
        *reinterpret_cast<f32x4*>((float*)a.out + (long long)batch * a.bso + (long long)m * a.ldo + n) = v;
      } else {
        float* o = (float*)a.out + ((long long)n_img * a.N + n) * HW + (py * a.W + px);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[(long long)r * HW] = v[r];
      }
    }
  }
  if (a.gb_x && a.ksplit == 1 && active) {
    // GroupNorm-backward sums of this gradient tile (see common.h): channel parameters once per i, pixels over j
    const int n_img = m0 / HW;
    const int cpg = a.N / 32;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int n = n0 + wn * TNW + i * 16 + (lane >> 4) * 4;
      if (n >= a.N) continue;
      float mu[4], rs[4], gam[4], bet[4], esc[4], esh[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int g = (n + c) / cpg;
        mu[c] = a.gb_stats[(n_img * 32 + g) * 2];
        rs[c] = a.gb_stats[(n_img * 32 + g) * 2 + 1];
        gam[c] = a.gb_gamma[n + c];
        bet[c] = a.gb_beta[n + c];
        esc[c] = a.gb_film ? a.gb_emb[(long long)n_img * a.gb_emb_ld + n + c] : 0.f;
        esh[c] = a.gb_film ? a.gb_emb[(long long)n_img * a.gb_emb_ld + a.N + n + c] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int m = m0 + wm * TMW + j * 16 + (lane & 15);
        const half4 xv = *reinterpret_cast<const half4*>(a.gb_x + (long long)m * a.N + n);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float dyh, xhat;
          gn_bwd_term(acc[i][j][c], (float)xv[c], mu[c], rs[c], gam[c], bet[c], esc[c], esh[c], a.gb_film != 0, a.gb_act != 0, dyh, xhat);
          ssum[i][c] += dyh;
          ssq[i][c] += dyh * xhat;
        }
      }
    }
  }
  if ((a.stat_out || a.gb_x) && a.ksplit == 1) {
    long long* const sdst = a.gb_x ? a.gb_csums : a.stat_out;
    const float scale_q = a.gb_x ? STAT_SCALE_SUM : STAT_SCALE_SQ;
    // reduce over the 16 pixel-lanes, stage per-wave channel sums in LDS, one atomic per (channel, stat) per block
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ssum[i][c] = row16_sum(ssum[i][c]);
        ssq[i][c] = row16_sum(ssq[i][c]);
      }
    __syncthreads();                       // the K-loop's LDS tiles are dead from here on
    if (active && (lane & 15) == 0) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int nl = wn * TNW + i * 16 + (lane >> 4) * 4 + c;      // channel within the block tile
          lds_f[(wm * BN + nl) * 2 + 0] = ssum[i][c];
          lds_f[(wm * BN + nl) * 2 + 1] = ssq[i][c];
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < BN * 2) {
      const int nl = t >> 1;
      if (n0 + nl < a.N) {
        const float v = lds_f[t] + lds_f[BN * 2 + t];                   // the two wm waves
        const int n_img = m0 / HW;                                        // a tile never straddles images (HW % BM == 0)
        const long long fx = __float2ll_rn(v * ((t & 1) ? scale_q : STAT_SCALE_SUM));
        atomicAdd(reinterpret_cast<unsigned long long*>(sdst + ((long long)n_img * a.N + n0 + nl) * 2 + (t & 1)),
                  (unsigned long long)fx);
      }
    }
  }
}
