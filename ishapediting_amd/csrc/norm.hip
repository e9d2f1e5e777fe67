// GroupNorm(32) over NHWC fp16 maps, statistics in fp32/fp64, and the fused
// normalise -> (FiLM) -> (SiLU) -> (2x2 average pool) pass that feeds each convolution.
// Reference: guided_diffusion/nn.py:16-18,92-99 (GroupNorm32 computes on x.float() and casts
// back), unet.py:236-252 (ResBlock: in_layers / h_upd / FiLM `gn(h)*(1+scale)+shift` / SiLU),
// unet.py:129-140 (AvgPool2d(2,2) used by resblock_updown).
//
// HBM-bound elementwise work: every thread moves 16-byte vectors of 8 channels; statistics are a
// two-stage deterministic reduction (per-row-chunk partial sums per channel, then one block per
// (image, group) combining them in double) -- no float atomics, bitwise reproducible.
#include "norm.h"
#include "gn_act.h"
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// statistics
// ---------------------------------------------------------------------------------------------
// partial[n][blk][C][2]  (sum, sum of squares) over this block's rows, kept in DOUBLE end to end: with |mean| >> std the
// one-pass E[x^2] - E[x]^2 needs every bit of the sums (a group at mean 100 / std 0.1 loses several per cent of its
// variance to fp32 partial sums); products of fp16 values are exact in double.  `partial` is typed float for the
// workspace bookkeeping (gn_partial_floats counts two floats per double).
__global__ __launch_bounds__(256) void gn_partial_kernel(const half_t* __restrict__ x, float* __restrict__ partial_f,
                                                         int HW, int C, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) char red_raw[];
  double* red = reinterpret_cast<double*>(red_raw);   // [rpi][C][2]
  double* partial = reinterpret_cast<double*>(partial_f);
  const int CV = C >> 3;
  const int rpi = 256 / CV > 0 ? 256 / CV : 1;   // rows handled per iteration
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int blk = blockIdx.x;
  const int row0 = blk * rows_per_block;
  const bool active = tid < rpi * CV;
  const int cv = tid % CV, r0 = tid / CV;
  double s[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i] = 0.0; q[i] = 0.0; }
  if (active) {
    const half_t* base = x + ((long long)n * HW + row0) * C + cv * 8;
    for (int r = r0; r < rows_per_block; r += rpi) {
      half8 v = *reinterpret_cast<const half8*>(base + (long long)r * C);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const double f = (double)(float)v[i]; s[i] += f; q[i] += f * f; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      red[((r0 * C) + cv * 8 + i) * 2 + 0] = s[i];
      red[((r0 * C) + cv * 8 + i) * 2 + 1] = q[i];
    }
  }
  __syncthreads();
  double* out = partial + ((long long)n * gridDim.x + blk) * C * 2;
  for (int c = tid; c < C * 2; c += 256) {
    double acc = 0.0;
    for (int r = 0; r < rpi; ++r) acc += red[r * C * 2 + c];
    out[c] = acc;
  }
}

// stats[n][32][2] = (mean, rstd)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial_f, float* __restrict__ stats,
                                                          int nblk, int C, int HW, float eps) {
  __shared__ double sh[2][256];
  const double* partial = reinterpret_cast<const double*>(partial_f);
  const int g = blockIdx.x, n = blockIdx.y;
  const int cpg = C / 32;
  const int tid = threadIdx.x;
  double s = 0.0, q = 0.0;
  const int total = nblk * cpg;
  for (int i = tid; i < total; i += 256) {
    int b = i / cpg, c = g * cpg + i % cpg;
    const double* p = partial + (((long long)n * nblk + b) * C + c) * 2;
    s += p[0];
    q += p[1];
  }
  sh[0][tid] = s; sh[1][tid] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    double cnt = (double)HW * cpg;
    double mean = sh[0][0] / cnt;
    double var = sh[1][0] / cnt - mean * mean;
    if (var < 0) var = 0;
    stats[(n * 32 + g) * 2 + 0] = (float)mean;
    stats[(n * 32 + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

int gn_stats_launch(const half_t* x, float* partial, float* stats, int N, int HW, int C, hipStream_t s) {
  ISHAP_REQUIRE(C % 32 == 0 && C / 8 <= 256, "GroupNorm channels: multiple of 32, at most 2048");
  int rpb = gn_rows_per_block(HW);
  int nblk = HW / rpb;
  int CV = C / 8;
  int rpi = 256 / CV > 0 ? 256 / CV : 1;
  size_t smem = (size_t)rpi * C * 2 * sizeof(double);
  ISHAP_REQUIRE(smem <= 160 * 1024, "gn_partial LDS");
  ISHAP_TRY(ishap_set_max_lds((const void*)gn_partial_kernel, 160 * 1024));
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, N), dim3(256), smem, s, x, partial, HW, C, rpb);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(32, N), dim3(256), 0, s, partial, stats, nblk, C, HW, 1e-5f);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// apply
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rh(float v) { return (float)(half_t)v; }   // round through fp16

// One thread = 8 channels of one OUTPUT pixel.
//  FILM: y = fp16(fp16(gn)*fp16(1+scale)) + shift   (fp16 arithmetic like the reference's half tensors)
//  ACT : SiLU
//  POOL: output pixel = mean of the 2x2 activated input pixels; also writes pooled raw x to xpool
//  SPLIT: fp32-grade output for the fp32 head: writes [hi | lo | hi] fp16 channel blocks (3*C wide)
template <bool FILM, bool ACT, bool POOL, bool SPLIT>
__global__ __launch_bounds__(256) void gn_apply_kernel(int h_main_blocks, int h_C, int h_H, int h_W, int h_N, GnApplyArgs a) {
  // h_*: copies of h_main_blocks, h_C, h_H, h_W, h_N as leading scalar parameters, preloaded into SGPRs at dispatch
  // (common.h, IgemmHot): the index arithmetic runs under the s_load of `a`
  const int CV = h_C >> 3;
  const int HWo = POOL ? (h_H >> 1) * (h_W >> 1) : h_H * h_W;
  const long long total = (long long)h_N * HWo * CV;
  const int cpg = h_C / 32;
  // statistics: either finalised already (a.stats) or finalised here from the producer's per-channel sums:
  // 8 lanes per (image, group) add up the group's channels, then mean / rstd (gd/nn.py:16-18, eps 1e-5)
  __shared__ float sh_stats[16 * 32 * 2];
  const float* stats = a.stats;
  if (a.sums) {
    const float cnt = (float)(h_H * h_W) * (float)cpg;
    for (int idx = threadIdx.x; idx < h_N * 32 * 8; idx += 256) {
      const int part = idx & 7, g = (idx >> 3) & 31, n = idx >> 8;
      // exact integer adds of the fixed-point channel sums, then mean / variance in double: the one-pass
      // E[x^2] - E[x]^2 cancels badly in fp32 once |mean| >> std (gn_finalize_kernel does the same arithmetic)
      long long si = 0, qi = 0;
      for (int c = g * cpg + part; c < (g + 1) * cpg; c += 8) {
        const long long* sp = a.sums + ((long long)n * h_C + c) * 2;
        if (a.x2) sp = c < a.csplit ? a.sums + ((long long)n * a.csplit + c) * 2
                                    : a.sums2 + ((long long)n * (h_C - a.csplit) + (c - a.csplit)) * 2;
        si += sp[0];
        qi += sp[1];
      }
      si = lanes8_sum(si);                          // DPP, not ds_bpermute: this prologue stands in front of every load
      qi = lanes8_sum(qi);
      if (part == 0) {
        const double md = (double)si * (1.0 / (double)STAT_SCALE_SUM) / (double)cnt;
        double vd = (double)qi * (1.0 / (double)STAT_SCALE_SQ) / (double)cnt - md * md;
        vd = vd < 0.0 ? 0.0 : vd;
        const float mean = (float)md;
        const float rstd = (float)(1.0 / sqrt(vd + 1e-5));
        sh_stats[(n * 32 + g) * 2] = mean;
        sh_stats[(n * 32 + g) * 2 + 1] = rstd;
        if (a.stats_out && blockIdx.x == 0) {
          a.stats_out[(n * 32 + g) * 2] = mean;
          a.stats_out[(n * 32 + g) * 2 + 1] = rstd;
        }
      }
    }
    __syncthreads();
    stats = sh_stats;
  }
  // The launcher makes the thread count a multiple of CV, so a thread keeps ONE 8-channel vector for all its pixels:
  // gamma/beta are loaded once, the per-image values (mean, rstd, FiLM) only when the image changes.
  // 32-bit index arithmetic (N * H * W <= 16 * 128 * 128 pixels): a 64-bit division per pixel cost more than the arithmetic
  const int tg = blockIdx.x * blockDim.x + threadIdx.x;
  const int nth = h_main_blocks * blockDim.x;
  const int cv = tg % CV, c0 = cv * 8;
  const int pstep = nth / CV, npix = h_N * HWo;
  float gam[8], bet[8], mu[8], rs[8];
  half_t sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { gam[i] = a.gamma[c0 + i]; bet[i] = a.beta[c0 + i]; }
  int cur_n = -1;
  auto image_params = [&](int n) __attribute__((always_inline)) {
    if (n == cur_n) return;
    cur_n = n;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i, g = c / cpg;
      mu[i] = stats[(n * 32 + g) * 2];
      rs[i] = stats[(n * 32 + g) * 2 + 1];
      if (FILM) {
        sc[i] = (half_t)(1.f + rh(a.emb[(long long)n * a.emb_ld + c]));
        sh[i] = (half_t)a.emb[(long long)n * a.emb_ld + h_C + c];
      }
    }
  };
  auto one = [&](const half8& v, float* o) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float y;
      if (SPLIT) {                                  // fp32 head: no fp16 rounding points
        y = ((float)v[i] - mu[i]) * rs[i] * gam[i] + bet[i];
        if (ACT) y = gn_silu(y);
      } else {
        half_t yh = gn_affine((float)v[i], mu[i], rs[i], gam[i], bet[i]);
        if (FILM) yh = gn_film(yh, sc[i], sh[i]);
        y = (float)yh;
        if (ACT) y = rh(gn_silu(y));
      }
      o[i] = y;
    }
  };
  if (POOL) {
    for (int pix = tg / CV; pix < npix; pix += pstep) {
      const int n = h_N == 1 ? 0 : pix / HWo;
      const int p = pix - n * HWo;
      image_params(n);
      const int Wo = h_W >> 1;
      const int yo = p / Wo, xo = p % Wo;
      float acc[8], xacc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc[i] = 0.f; xacc[i] = 0.f; }
      half8 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {                  // the four inputs of the 2x2 cell in flight together
        const long long src = (long long)n * h_H * h_W + (2 * yo + (q >> 1)) * h_W + 2 * xo + (q & 1);
        v[q] = *reinterpret_cast<const half8*>(a.x + src * h_C + c0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t[8];
        one(v[q], t);
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] += t[i]; xacc[i] += (float)v[q][i]; }
      }
      half8 ov, xv;
#pragma unroll
      for (int i = 0; i < 8; ++i) { ov[i] = (half_t)(acc[i] * 0.25f); xv[i] = (half_t)(xacc[i] * 0.25f); }
      *reinterpret_cast<half8*>(a.out + (long long)pix * h_C + c0) = ov;
      if (a.xpool) *reinterpret_cast<half8*>(a.xpool + (long long)pix * h_C + c0) = xv;
    }
  } else {
    // Round 5: a thread's pixels are fetched U at a time BEFORE any of them is processed (one 16-byte load per wave in flight in
    // the old loop).  Measured (profiles/round5_ab_gn_inflight.txt): 8.99 -> 8.64 us (plain + SiLU, 128^2 x 256) and 7.32 -> 6.62
    // (FiLM variant) -- a few per cent, not the 2x a bytes-in-flight limit would have given: the launch is a chain of ~2 us phases
    // (sums -> finalise -> barrier -> loads -> stores), not a stream.
#ifndef GN_APPLY_U
#define GN_APPLY_U 4
#endif
    constexpr int U = SPLIT ? 1 : GN_APPLY_U;      // the head's variant writes three vectors per pixel: batching it doubled its time (8.8 -> 15.8 us)
    for (int pix0 = tg / CV; pix0 < npix; pix0 += U * pstep) {
      half8 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pix = pix0 + u * pstep;
        if (pix < npix) {
          if (a.x2) {                               // two-source input (skip concatenation)
            v[u] = c0 < a.csplit ? *reinterpret_cast<const half8*>(a.x + (long long)pix * a.csplit + c0)
                                 : *reinterpret_cast<const half8*>(a.x2 + (long long)pix * (h_C - a.csplit) + (c0 - a.csplit));
          } else {
            v[u] = *reinterpret_cast<const half8*>(a.x + (long long)pix * h_C + c0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pix = pix0 + u * pstep;
        if (pix >= npix) break;
        image_params(h_N == 1 ? 0 : pix / HWo);
        if (a.x2) *reinterpret_cast<half8*>(a.xcopy + (long long)pix * h_C + c0) = v[u];      // also emit the raw concatenation
        float o[8];
        one(v[u], o);
        if (SPLIT) {
          half8 hi, lo;
#pragma unroll
          for (int i = 0; i < 8; ++i) { hi[i] = (half_t)o[i]; lo[i] = (half_t)(o[i] - (float)hi[i]); }
          half_t* dst = a.out + (long long)pix * (3LL * h_C);
          *reinterpret_cast<half8*>(dst + c0) = hi;
          *reinterpret_cast<half8*>(dst + h_C + c0) = lo;
          *reinterpret_cast<half8*>(dst + 2 * h_C + c0) = hi;
        } else {
          half8 ov;
#pragma unroll
          for (int i = 0; i < 8; ++i) ov[i] = (half_t)o[i];
          store_out16(a.out + (long long)pix * h_C + c0, ov);
        }
      }
    }
  }
}

int gn_apply_launch(const GnApplyArgs& a, hipStream_t s) {
  ISHAP_REQUIRE(!a.x2 || (a.sums && a.sums2 && a.xcopy && !a.pool && !a.split && a.csplit % 8 == 0 && a.csplit > 0 && a.csplit < a.C),
                "two-source GroupNorm: sums of both halves, a copy target, plain variant");
  const int HWo = a.pool ? (a.H / 2) * (a.W / 2) : a.H * a.W;
  long long total = (long long)a.N * HWo * (a.C / 8);
  int blocks = (int)((total + 255) / 256);
  constexpr int cap_sums = 1024;          // in-situ sweep (tools/experiments/gn_apply_probe.sh): flat 512 .. 2048
  if (blocks > (a.sums ? cap_sums : 4096)) blocks = a.sums ? cap_sums : 4096;   // fewer, fatter blocks amortise the finalise prologue
  // thread count = multiple of CV (a thread owns one 8-channel vector): blocks = multiple of CV / gcd(CV, 256)
  const int CV = a.C / 8;
  int gcd = CV, r256 = 256;
  while (r256) { const int t = gcd % r256; gcd = r256; r256 = t; }
  const int unit = CV / gcd;
  blocks = blocks < unit ? unit : blocks / unit * unit;
  GnApplyArgs a2 = a;
  a2.main_blocks = blocks;
  dim3 g(blocks), b(256);
  if (a.split) hipLaunchKernelGGL((gn_apply_kernel<false, true, false, true>), g, b, 0, s, a2.main_blocks, a2.C, a2.H, a2.W, a2.N, a2);
  else if (a.pool) hipLaunchKernelGGL((gn_apply_kernel<false, true, true, false>), g, b, 0, s, a2.main_blocks, a2.C, a2.H, a2.W, a2.N, a2);
  else if (a.film) hipLaunchKernelGGL((gn_apply_kernel<true, true, false, false>), g, b, 0, s, a2.main_blocks, a2.C, a2.H, a2.W, a2.N, a2);
  else if (a.act) hipLaunchKernelGGL((gn_apply_kernel<false, true, false, false>), g, b, 0, s, a2.main_blocks, a2.C, a2.H, a2.W, a2.N, a2);
  else hipLaunchKernelGGL((gn_apply_kernel<false, false, false, false>), g, b, 0, s, a2.main_blocks, a2.C, a2.H, a2.W, a2.N, a2);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
