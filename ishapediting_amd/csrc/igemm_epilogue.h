// Shared epilogue of the implicit-GEMM kernels (igemm.hip register-staged, igemm2.hip LDS-DMA ring).
// A lane holds, for each (i, j) MFMA tile, output channels n..n+3 of one pixel m.  Besides bias / residual /
// layout handling, the epilogue can accumulate the per-channel sum and sum of squares of the fp16 values it
// stores (`stat_out`): these are the GroupNorm statistics of the NEXT layer (gd/nn.py:16-18), gathered here so no
// separate pass has to re-read the tensor.
#pragma once
#include "common.h"
#include "gn_bwd_terms.h"
#include <type_traits>

// STAGE_THREADS > 0 (the LDS-DMA kernels; = threads of the workgroup, all of which call this function): fp16 outputs
// of an unsplit launch are collected in LDS and leave as full 16-byte-per-lane row segments written by every wave,
// loader waves included.  The fragment layout's own stores are 8 bytes per lane over 16 rows: issue-bound, ~9 us of
// a 128^2-map launch against ~2 us this way (tools/fixed_cost_probe2.sh).  Values and their order of evaluation are
// unchanged, so results are bit-identical to the direct path.
template <int MT, int NT, int TMW, int TNW, int BN, int STAGE_THREADS = 0>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x4 (&acc)[NT][MT], int m0, int n0, int wm, int wn,
                                               int lane, int batch, int ks_id, float* lds_f, bool active = true) {
  // `active` = false: a loader wave of a producer/consumer kernel -- it owns no outputs but must take part in the
  // workgroup barriers of the statistics reduction below
  // fields used inside the unrolled tile loops, read once: left as `a.field` the compiler kept {alpha, out_mode, stat_out}
  // in a private copy of the argument block and re-read it from scratch behind an s_waitcnt vmcnt(0) before every tile
  // -- each output store then waited for the previous one to be acknowledged
  const float alpha = a.alpha;
  const int out_mode = a.out_mode;
  const float* const bias = a.bias;
  const float* const bias2 = a.bias2;
  const half_t* const resp = a.res;
  const int HW = a.H * a.W;
  constexpr int BM_T = 2 * TMW;                  // rows of the workgroup tile
  constexpr int LDT = BN + 8;                    // staged-tile row pitch in halfs (conflict-free 8-byte fragment writes)
  half_t* const out16 = (half_t*)a.out + (long long)batch * a.bso;
  const bool staged = STAGE_THREADS > 0 && out_mode == IG_OUT_F16 && a.ksplit == 1 && (a.N & 7) == 0 && (a.ldo & 7) == 0 &&
                      (reinterpret_cast<unsigned long long>(out16) & 15) == 0;
  half_t* const tile = reinterpret_cast<half_t*>(lds_f);
  if (staged) __syncthreads();                   // the K loop's LDS tiles are dead from here on
  // Forward statistics are gathered about a PIVOT: every lane takes the first value it stores in a channel as that channel's
  // pivot p and accumulates sum(x - p), sum((x - p)^2) -- small numbers even when the channel sits at |mean| >> std, where
  // the plain fp32 sums of x and x^2 cancel (128 squares of 100.1 +- 0.1 carry an fp32 error of ~0.1 against a variance
  // contribution of 1.3: the GroupNorm32 of gd/nn.py:16-18 would come out several per cent off).  The pivots are
  // reconciled below (a lane row to its largest pivot, then the tile's waves in double), before the fixed-point atomics.
  float ssum[NT][4], ssq[NT][4], piv[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) { ssum[i][c] = 0.f; ssq[i][c] = 0.f; piv[i][c] = 0.f; }
  // One loop over the wave's MT x NT fragment tiles per output mode, the mode tested OUTSIDE the loop: with the tests
  // inside, every tile jumped over the other modes' code and the launch paid an instruction-cache miss per jump
  // (~4 us of a 128x128-tile launch, tools/fixed_cost_probe3.sh).
  auto col = [&](int i) { return n0 + wn * TNW + i * 16 + (lane >> 4) * 4; };
  auto row = [&](int j) { return m0 + wm * TMW + j * 16 + (lane & 15); };
  if (!active) {
    // nothing to compute
  } else if (a.ksplit > 1) {
    // a K slice: the fp32 partial tile goes to the workspace as it is
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      float* dst = a.ws + (((long long)ks_id * a.nbatch + batch) * a.M + row(j)) * a.N;
#pragma unroll
      for (int i = 0; i < NT; ++i)
        if (col(i) < a.N) *reinterpret_cast<f32x4*>(dst + col(i)) = acc[i][j];
    }
  } else {
    // per-channel operands of the whole tile first; the residual is fetched one 16-pixel row of fragments ahead of the
    // arithmetic (all of it at once cost 32 more live registers and spilled)
    f32x4 b1[NT], b2[NT];
    half4 rr[2][NT];
    int pimg[MT], ppy[MT], ppx[MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      b1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      b2[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (col(i) < a.N) {
        if (bias) b1[i] = *reinterpret_cast<const f32x4*>(bias + col(i));
        if (bias2) b2[i] = *reinterpret_cast<const f32x4*>(bias2 + col(i));
      }
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int m = row(j);
      pimg[j] = 0; ppy[j] = 0; ppx[j] = 0;
      if (a.res_ups || out_mode == IG_OUT_NCHW_F32) {
        pimg[j] = m / HW;
        const int p = m - pimg[j] * HW;
        ppy[j] = p / a.W;
        ppx[j] = p - ppy[j] * a.W;
      }
    }
    auto fetch_res = [&](int j) {
      if (!resp || j >= MT) return;
      const long long rrow = a.res_ups ? ((long long)pimg[j] * (HW >> 2) + (ppy[j] >> 1) * (a.W >> 1) + (ppx[j] >> 1)) : row(j);
#pragma unroll
      for (int i = 0; i < NT; ++i)
        if (col(i) < a.N) rr[j & 1][i] = *reinterpret_cast<const half4*>(resp + rrow * a.ldr + col(i));
    };
    fetch_res(0);
    // alpha * acc (+ bias) (+ bias2) (+ residual), in this order
    auto value = [&](int i, int j) {
      f32x4 v = acc[i][j];
      v *= alpha;
      if (bias) v += b1[i];
      if (bias2) v += b2[i];
      if (resp) {
        const half4 r = rr[j & 1][i];
        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
      }
      return v;
    };
    if (out_mode == IG_OUT_F16) {
      // two copies of the loop (LDS tile / straight to memory) so that each uses its own address space's stores
      auto f16_tiles = [&](auto to_lds) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          fetch_res(j + 1);
#pragma unroll
          for (int i = 0; i < NT; ++i) {
            if (col(i) >= a.N) continue;
            const f32x4 v = value(i, j);
            const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            if constexpr (decltype(to_lds)::value) *reinterpret_cast<half4*>(tile + (row(j) - m0) * LDT + (col(i) - n0)) = o;
            else *reinterpret_cast<half4*>(out16 + (long long)row(j) * a.ldo + col(i)) = o;
            // the stored (rounded) values: GroupNorm statistics of the next layer / operand of the backward sums below
            const f32x4 fo = {(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
            if (j == 0) {
#pragma unroll
              for (int c = 0; c < 4; ++c) piv[i][c] = fo[c];
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c) { const float d = fo[c] - piv[i][c]; ssum[i][c] += d; ssq[i][c] = fmaf(d, d, ssq[i][c]); }
            }
            acc[i][j] = fo;
          }
        }
      };
      if (staged) f16_tiles(std::true_type{});
      else f16_tiles(std::false_type{});
    } else if (out_mode == IG_OUT_F32) {
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        fetch_res(j + 1);
        float* const drow = (float*)a.out + (long long)batch * a.bso + (long long)row(j) * a.ldo;
#pragma unroll
        for (int i = 0; i < NT; ++i)
          if (col(i) < a.N) *reinterpret_cast<f32x4*>(drow + col(i)) = value(i, j);
      }
    } else {
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        fetch_res(j + 1);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          if (col(i) >= a.N) continue;
          const f32x4 v = value(i, j);
          float* o = (float*)a.out + ((long long)pimg[j] * a.N + col(i)) * HW + (ppy[j] * a.W + ppx[j]);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[(long long)r * HW] = v[r];
        }
      }
    }
  }
  if (staged) {
    __syncthreads();
    constexpr int CPRW = BN / 8;                 // 16-byte chunks per tile row
    for (int c = threadIdx.x; c < BM_T * CPRW; c += STAGE_THREADS) {
      const int row = c / CPRW, ch = c - row * CPRW;
      const int n = n0 + ch * 8;
#ifdef ABL_EPI_NOSTORE
      if (n < a.N && alpha == 12345.f)
#else
      if (n < a.N)
#endif
        *reinterpret_cast<half8*>(out16 + (long long)(m0 + row) * a.ldo + n) = *reinterpret_cast<const half8*>(tile + row * LDT + ch * 8);
    }
  }
  if (a.gb_x && a.ksplit == 1 && active) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) { ssum[i][c] = 0.f; ssq[i][c] = 0.f; piv[i][c] = 0.f; }     // O(1) terms: no pivot
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
    // reduce over the 16 pixel-lanes, stage per-wave channel sums in LDS, one atomic per (channel, stat) per block.
    // A lane row first moves to its largest pivot P (e = p - P is exact: both are fp16 values):
    //   sum(x - P) = sum(x - p) + n e,   sum((x - P)^2) = sum((x - p)^2) + 2 e sum(x - p) + n e^2,   n = MT values per lane
    const bool pivoted = !a.gb_x;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (pivoted) {
          const float P = row16_max(piv[i][c]);
          const float e = piv[i][c] - P;
          ssq[i][c] = fmaf(e, fmaf((float)MT, e, 2.f * ssum[i][c]), ssq[i][c]);
          ssum[i][c] = fmaf((float)MT, e, ssum[i][c]);
          piv[i][c] = P;
        }
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
          lds_f[(wm * BN + nl) * 3 + 0] = ssum[i][c];
          lds_f[(wm * BN + nl) * 3 + 1] = ssq[i][c];
          lds_f[(wm * BN + nl) * 3 + 2] = piv[i][c];
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < BN * 2) {
      const int nl = t >> 1;
      if (n0 + nl < a.N) {
        // the two wm waves, each about its own pivot P over nW = 16 * MT pixels, brought to pivot 0 in double:
        //   sum x = s + nW P,   sum x^2 = q + 2 P s + nW P^2
        constexpr double nW = 16.0 * MT;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const double sw = (double)lds_f[(w * BN + nl) * 3 + 0], qw = (double)lds_f[(w * BN + nl) * 3 + 1];
          const double P = (double)lds_f[(w * BN + nl) * 3 + 2];
          v += (t & 1) ? qw + P * (2.0 * sw + nW * P) : sw + nW * P;
        }
        const int n_img = m0 / HW;                                        // a tile never straddles images (HW % BM == 0)
        const long long fx = __double2ll_rn(v * (double)((t & 1) ? scale_q : STAT_SCALE_SUM));
#ifdef ABL_STAT_COPIES      // harness probe: spread the same-address atomics over ABL_STAT_COPIES copies of the table
        atomicAdd(reinterpret_cast<unsigned long long*>(sdst + (long long)((m0 / BM_T) % ABL_STAT_COPIES) * a.N * 2 +
                                                        ((long long)n_img * a.N + n0 + nl) * 2 + (t & 1)),
#else
        atomicAdd(reinterpret_cast<unsigned long long*>(sdst + ((long long)n_img * a.N + n0 + nl) * 2 + (t & 1)),
#endif
                  (unsigned long long)fx);
      }
    }
  }
}
