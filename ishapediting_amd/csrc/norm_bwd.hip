// Input-gradient of  y = act(film(GroupNorm32(x)))  on NHWC fp16 maps.
// Reference: what autograd derives for nn.py:16-18 + unet.py:248-252 when loss.backward() runs at
// drag_utils.py:383.  No parameter gradients are formed (the reference computes and discards them).
// Same two-stage deterministic reduction as the forward statistics; the forward's activations are
// recomputed from the saved GN input and (mean, rstd) instead of being stored.
#include "norm.h"
#include "gn_bwd_terms.h"

// gradient arriving at pixel (n, y, x) of the GN-input resolution, 8 channels
__device__ __forceinline__ void load_upstream(const half_t* g, int gmode, int n, int y, int x, int H, int W, int C, int c0,
                                              float* o) {
  if (gmode == GB_SAME) {
    half8 v = *reinterpret_cast<const half8*>(g + ((long long)n * H * W + y * W + x) * C + c0);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  } else if (gmode == GB_UNPOOL) {      // forward was AvgPool2d(2,2): each input pixel gets 1/4 of its cell's gradient
    const int Wh = W >> 1;
    half8 v = *reinterpret_cast<const half8*>(g + ((long long)n * (H >> 1) * Wh + (y >> 1) * Wh + (x >> 1)) * C + c0);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.25f * (float)v[i];
  } else {                              // forward was nearest x2: the gradients of the 4 copies add up
    const int W2 = W << 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      half8 v = *reinterpret_cast<const half8*>(
          g + ((long long)n * (H << 1) * W2 + (2 * y + (q >> 1)) * W2 + 2 * x + (q & 1)) * C + c0);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += (float)v[i];
    }
  }
}

// dyh[8] = upstream * act'(pre) * (1+scale) * gamma,  xh[8] = normalised input, for 8 channels of one pixel
template <bool FILM, bool ACT>
__device__ __forceinline__ void bwd_terms(const GnBwdArgs& a, int n, int y, int x, int c0, float* dyh, float* xh) {
  const int cpg = a.C / 32;
  float up[8];
  load_upstream(a.g, a.gmode, n, y, x, a.H, a.W, a.C, c0, up);
  half8 xv = *reinterpret_cast<const half8*>(a.x + ((long long)n * a.H * a.W + y * a.W + x) * a.C + c0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = c0 + i;
    const int g = c / cpg;
    const float mu = a.stats[(n * 32 + g) * 2], rs = a.stats[(n * 32 + g) * 2 + 1];
    gn_bwd_term(up[i], (float)xv[i], mu, rs, a.gamma[c], (FILM || ACT) ? a.beta[c] : 0.f,
                FILM ? a.emb[(long long)n * a.emb_ld + c] : 0.f, FILM ? a.emb[(long long)n * a.emb_ld + a.C + c] : 0.f, FILM, ACT,
                dyh[i], xh[i]);
  }
}

template <bool FILM, bool ACT>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(GnBwdArgs a, int rows_per_block) {
  extern __shared__ float red[];
  const int C = a.C, CV = C >> 3;
  const int rpi = 256 / CV > 0 ? 256 / CV : 1;
  const int tid = threadIdx.x, n = blockIdx.y, blk = blockIdx.x;
  const int row0 = blk * rows_per_block;
  const bool active = tid < rpi * CV;
  const int cv = tid % CV, r0 = tid / CV;
  float s[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
  if (active) {
    for (int r = r0; r < rows_per_block; r += rpi) {
      const int p = row0 + r;
      float dyh[8], xh[8];
      bwd_terms<FILM, ACT>(a, n, p / a.W, p % a.W, cv * 8, dyh, xh);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s[i] += dyh[i]; q[i] += dyh[i] * xh[i]; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      red[((r0 * C) + cv * 8 + i) * 2 + 0] = s[i];
      red[((r0 * C) + cv * 8 + i) * 2 + 1] = q[i];
    }
  }
  __syncthreads();
  // one fixed-point atomic per (channel, term) per block: integer adds commute -> bitwise reproducible sums
  for (int c = tid; c < C * 2; c += 256) {
    float acc = 0.f;
    for (int r = 0; r < rpi; ++r) acc += red[r * C * 2 + c];
    atomicAdd(reinterpret_cast<unsigned long long*>(a.csums + (long long)n * C * 2 + c),
              (unsigned long long)__float2ll_rn(acc * STAT_SCALE_SUM));
  }
}

template <bool FILM, bool ACT>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const long long* h_csums, int h_main_blocks, int h_C, int h_H, int h_W, int h_N, GnBwdArgs a) {
  // h_*: preloaded copies of h_csums, h_main_blocks, h_C, h_H, h_W, h_N (see gn_apply_kernel)
  const int CV = h_C >> 3, HW = h_H * h_W, cpg = h_C / 32;
  const long long total = (long long)h_N * HW * CV;
  // group means of dyh and dyh*xh from the per-channel sums (8 lanes per (image, group))
  __shared__ float sh_m[16 * 32 * 2];
  {
    const float cnt = (float)HW * (float)cpg;
    for (int idx = threadIdx.x; idx < h_N * 32 * 8; idx += 256) {
      const int part = idx & 7, g = (idx >> 3) & 31, n = idx >> 8;
      float s1 = 0.f, s2 = 0.f;
      for (int c = g * cpg + part; c < (g + 1) * cpg; c += 8) {
        s1 += (float)h_csums[((long long)n * h_C + c) * 2] * (1.f / STAT_SCALE_SUM);
        s2 += (float)h_csums[((long long)n * h_C + c) * 2 + 1] * (1.f / STAT_SCALE_SUM);
      }
      s1 = lanes8_sum(s1);
      s2 = lanes8_sum(s2);
      if (part == 0) { sh_m[(n * 32 + g) * 2] = s1 / cnt; sh_m[(n * 32 + g) * 2 + 1] = s2 / cnt; }
    }
    __syncthreads();
  }
  // thread count is a multiple of CV (launcher): one 8-channel vector per thread, parameters loaded once per image
  // 32-bit index arithmetic (N * H * W <= 16 * 128 * 128 pixels; shifts for the power-of-two maps of the UNet): the 64-bit
  // divisions per pixel cost more issue time than the gradient arithmetic
  const int tg = blockIdx.x * blockDim.x + threadIdx.x;
  const int nth = h_main_blocks * blockDim.x;
  const int cv = tg % CV, c0 = cv * 8;
  const int pstep = nth / CV, npix = h_N * HW;
  const bool wpow2 = (h_W & (h_W - 1)) == 0;
  const int wsh = 31 - __builtin_clz(h_W);
  float gam[8], bet[8], mu[8], rs[8], esc[8], esh[8], m1[8], m2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { gam[i] = a.gamma[c0 + i]; bet[i] = (FILM || ACT) ? a.beta[c0 + i] : 0.f; esc[i] = 0.f; esh[i] = 0.f; }
  int cur_n = -1;
  // (round 5: the loop can fetch the operands of U pixels before it processes the first -- see gn_apply_kernel, norm.hip; here a
  // pixel already has 2-4 loads in flight and U = 2 was slower)
#ifndef GN_BWD_U
#define GN_BWD_U 1      // measured: 2 pixels per batch 8.93 -> 9.47 us (170-188 registers), not kept (profiles/round5_ab_gn_inflight.txt)
#endif
  constexpr int U = GN_BWD_U;
  for (int pix0 = tg / CV; pix0 < npix; pix0 += U * pstep) {
    float up[U][8], ad[U][8];
    half8 xv[U], a2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pix = pix0 + u * pstep;
      a2[u] = (half8){0, 0, 0, 0, 0, 0, 0, 0};
      if (pix < npix) {
        const int n = h_N == 1 ? 0 : pix / HW, p = pix - n * HW;
        const int y = wpow2 ? p >> wsh : p / h_W, x = p - y * h_W;
        if (a.gmode == GB_SAME) {
          // round 6: the four vectors of a pixel as UNCONDITIONAL loads (a missing addend reads the gradient / the input again and
          // is not used) -- `if (a.add) load` and `if (a.add2) load` each closed with an s_waitcnt vmcnt(0): three dependent round
          // trips per pixel where one does
          const long long o8 = (long long)pix * h_C + c0;
          const half8 gv = *reinterpret_cast<const half8*>(a.g + o8);
          xv[u] = *reinterpret_cast<const half8*>(a.x + o8);
          const half8 av = *reinterpret_cast<const half8*>((a.add ? a.add : a.g) + o8);
          a2[u] = *reinterpret_cast<const half8*>((a.add2 ? a.add2 : a.x) + o8);
#pragma unroll
          for (int i = 0; i < 8; ++i) { up[u][i] = (float)gv[i]; ad[u][i] = (float)av[i]; }
        } else {
          load_upstream(a.g, a.gmode, n, y, x, h_H, h_W, h_C, c0, up[u]);
          xv[u] = *reinterpret_cast<const half8*>(a.x + (long long)pix * h_C + c0);
          if (a.add) load_upstream(a.add, a.gmode, n, y, x, h_H, h_W, h_C, c0, ad[u]);
          if (a.add2) a2[u] = *reinterpret_cast<const half8*>(a.add2 + (long long)pix * h_C + c0);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pix = pix0 + u * pstep;
      if (pix >= npix) break;
      const int n = h_N == 1 ? 0 : pix / HW;
      if (n != cur_n) {
        cur_n = n;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int c = c0 + i, g = c / cpg;
          mu[i] = a.stats[(n * 32 + g) * 2];
          rs[i] = a.stats[(n * 32 + g) * 2 + 1];
          m1[i] = sh_m[(n * 32 + g) * 2];
          m2[i] = sh_m[(n * 32 + g) * 2 + 1];
          if (FILM) { esc[i] = a.emb[(long long)n * a.emb_ld + c]; esh[i] = a.emb[(long long)n * a.emb_ld + h_C + c]; }
        }
      }
      half8 o;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float dyh, xh;
        gn_bwd_term(up[u][i], (float)xv[u][i], mu[i], rs[i], gam[i], bet[i], esc[i], esh[i], FILM, ACT, dyh, xh);
        float v = gn_bwd_dx(rs[i], dyh, xh, m1[i], m2[i]);
        if (a.add) v += ad[u][i];
        if (a.add2) v = (float)(half_t)v + (float)a2[u][i];      // same rounding as a separate fp16 add of the two gradient maps
        o[i] = (half_t)v;
      }
      if (a.csplit == 0) *reinterpret_cast<half8*>(a.dx + (long long)pix * h_C + c0) = o;
      else if (c0 < a.csplit) *reinterpret_cast<half8*>(a.dx + (long long)pix * a.csplit + c0) = o;
      else *reinterpret_cast<half8*>(a.dx2 + (long long)pix * (h_C - a.csplit) + (c0 - a.csplit)) = o;
    }
  }
}

int gn_backward_launch(const GnBwdArgs& a, hipStream_t s) {
  ISHAP_REQUIRE(a.C % 32 == 0 && a.C / 8 <= 256, "GroupNorm channels");
  ISHAP_REQUIRE(a.csplit == 0 || (a.dx2 && a.csplit % 8 == 0 && a.csplit < a.C), "split output");
  const int HW = a.H * a.W;
  const int rpb = gn_rows_per_block(HW), nblk = HW / rpb, CV = a.C / 8;
  const int rpi = 256 / CV > 0 ? 256 / CV : 1;
  const size_t smem = (size_t)rpi * a.C * 2 * sizeof(float);
  long long total = (long long)a.N * HW * CV;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  int gcd = CV, r256 = 256;                      // apply kernel: thread count = multiple of CV
  while (r256) { const int t = gcd % r256; gcd = r256; r256 = t; }
  const int unit = CV / gcd;
  blocks = blocks < unit ? unit : blocks / unit * unit;
  GnBwdArgs a2 = a;
  a2.main_blocks = blocks;
  const int grid_apply = blocks;
#define GB_LAUNCH(F, A)                                                                                             \
  do {                                                                                                              \
    if (!a.sums_ready)                                                                                              \
      hipLaunchKernelGGL((gn_bwd_partial_kernel<F, A>), dim3(nblk, a.N), dim3(256), smem, s, a, rpb);               \
    hipLaunchKernelGGL((gn_bwd_apply_kernel<F, A>), dim3(grid_apply), dim3(256), 0, s, (const long long*)a2.csums,   \
                       a2.main_blocks, a2.C, a2.H, a2.W, a2.N, a2);                                               \
  } while (0)
  if (a.film) GB_LAUNCH(true, true);
  else if (a.act) GB_LAUNCH(false, true);
  else GB_LAUNCH(false, false);
#undef GB_LAUNCH
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
