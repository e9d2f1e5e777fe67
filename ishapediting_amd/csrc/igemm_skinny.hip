// Small-map convolutions / GEMMs (8x8 and 16x16 feature maps at batch 1: M = 64 or 256 pixels).
// These layers stream 10-20 MB of weights for a few GFLOP; with the tiled kernels they needed split-K across
// workgroups plus a reduce launch (two ~5 us launch floors and an fp32 round trip through HBM).  Here ONE launch does
// it: a 1024-thread workgroup owns a (16*MT pixels) x (16 channels) output tile and its 16 waves split K between
// them.  Every wave loads its MFMA fragments straight from global memory into registers (a weight row is one full
// 128-byte line per 64-wide K-step; the im2col gather is per-lane address arithmetic with zeros for the padding), so
// the K loop has no LDS staging and no workgroup barrier; the 16 partial accumulators meet in LDS once, are summed in
// wave order (deterministic) and go through the same epilogue duties as igemm_epilogue.h (bias, residual, fp16/fp32/NCHW
// stores, GroupNorm statistics or GroupNorm-backward sums as 64-bit fixed-point atomics).
#include "common.h"
#include "gn_bwd_terms.h"

namespace {

constexpr int SK_WAVES = 16;
#ifndef SK_DEPTH
#define SK_DEPTH 3                       // K-steps of fragment loads in flight per wave
#endif
__device__ __attribute__((aligned(128))) half_t g_zero_line_sk[64];   // zero-initialised: what padded taps read

template <int MT, bool CONV3>
__global__ __launch_bounds__(SK_WAVES * 64) void igemm_skinny_kernel(const void* hX, const void* hWt, int hK, int hCin, int hldx, int hldw,
                                                                     int hH, int hW, int hN, IgemmArgs a0) {
  // leading scalar parameters are preloaded into SGPRs at dispatch (common.h, IgemmHot): the fragment loads need nothing else
  IgemmArgs a = a0;
  a.X = reinterpret_cast<const half_t*>(hX); a.Wt = reinterpret_cast<const half_t*>(hWt);
  a.K = hK; a.Cin = hCin; a.ldx = hldx; a.ldw = hldw; a.H = hH; a.W = hW; a.N = hN;
  __shared__ f32x4 red[SK_WAVES][MT][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, col = lane & 15;
  const int m0 = blockIdx.x * (16 * MT), n0 = blockIdx.y * 16;
  const int HW = a.H * a.W;
  const int KS = a.K / 64;
  const int per = (KS + SK_WAVES - 1) / SK_WAVES;
  const int s0 = wave * per, s1 = min(KS, s0 + per);
  const int steps_per_tap = CONV3 ? a.Cin / 64 : KS;

  // this lane's pixel in each 16-pixel sub-tile (B-operand column) and its weight row (A-operand row)
  int pn[MT], py[MT], px[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + j * 16 + col;
    if (CONV3) {
      pn[j] = m / HW;
      const int p = m - pn[j] * HW;
      py[j] = p / a.W;
      px[j] = p - py[j] * a.W;
    } else {
      pn[j] = m; py[j] = 0; px[j] = 0;
    }
  }
  const half_t* wrow = a.Wt + (long long)(n0 + col) * a.ldw + 8 * g;

  struct Frag { half8 w[2]; half8 x[MT][2]; };
  auto load = [&](int s, Frag& f) {
    int tap = 0, c0;
    if (CONV3) { tap = s / steps_per_tap; c0 = (s - tap * steps_per_tap) * 64; }
    else c0 = s * 64;
    const int dy = CONV3 ? tap / 3 - 1 : 0, dx = CONV3 ? tap % 3 - 1 : 0;
    const int kofs = CONV3 ? tap * a.Cin + c0 : c0;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) f.w[kk] = *reinterpret_cast<const half8*>(wrow + kofs + kk * 32);
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const half_t* src = g_zero_line_sk + 8 * g;       // unconditional loads: a padded tap reads zeros
      if (CONV3) {
        const int yy = py[j] + dy, xc = px[j] + dx;
        const bool in = yy >= 0 && yy < a.H && xc >= 0 && xc < a.W;
        const long long row = a.ups ? (long long)pn[j] * (HW >> 2) + (yy >> 1) * (a.W >> 1) + (xc >> 1)
                                    : (long long)pn[j] * HW + yy * a.W + xc;
        if (in) src = a.X + row * a.ldx + c0 + 8 * g;
      } else {
        src = a.X + (long long)pn[j] * a.ldx + c0 + 8 * g;
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) f.x[j][kk] = *reinterpret_cast<const half8*>(src + kk * 32);
    }
  };

  f32x4 acc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    // ring of SK_DEPTH fragment sets; the slot index is static (the loop is unrolled by the ring size)
    Frag ring[SK_DEPTH];
#pragma unroll
    for (int d = 0; d < SK_DEPTH - 1; ++d)
      if (s0 + d < s1) load(s0 + d, ring[d]);
    for (int sb = s0; sb < s1; sb += SK_DEPTH) {
#pragma unroll
      for (int d = 0; d < SK_DEPTH; ++d) {
        const int s = sb + d;
        if (s < s1) {
          if (s + SK_DEPTH - 1 < s1) load(s + SK_DEPTH - 1, ring[(d + SK_DEPTH - 1) % SK_DEPTH]);
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < MT; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[d].w[kk], ring[d].x[j][kk], acc[j], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) red[wave][j][lane] = acc[j];
  __syncthreads();
  if (wave >= MT) return;

  // ---- wave j finishes sub-tile j: lane = (pixel m, channels n .. n+3) ----
  const int j = wave;
  f32x4 v = red[0][j][lane];
#pragma unroll
  for (int w = 1; w < SK_WAVES; ++w) v += red[w][j][lane];
  const int m = m0 + j * 16 + col;
  const int n = n0 + g * 4;
  const bool ok = n < a.N;
  int n_img = 0, oy = 0, ox = 0;
  if (a.res_ups || a.out_mode == IG_OUT_NCHW_F32 || a.stat_out || a.gb_x) {
    n_img = m / HW;
    const int p = m - n_img * HW;
    oy = p / a.W;
    ox = p - oy * a.W;
  }
  float s1v[4] = {0.f, 0.f, 0.f, 0.f}, s2v[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    v *= a.alpha;
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.bias2) v += *reinterpret_cast<const f32x4*>(a.bias2 + n);
    if (a.res) {
      const long long rrow = a.res_ups ? ((long long)n_img * (HW >> 2) + (oy >> 1) * (a.W >> 1) + (ox >> 1)) : m;
      const half4 r = *reinterpret_cast<const half4*>(a.res + rrow * a.ldr + n);
      v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
    }
    if (a.out_mode == IG_OUT_F16) {
      const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4*>((half_t*)a.out + (long long)m * a.ldo + n) = o;
      if (a.stat_out) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float f = (float)o[c]; s1v[c] = f; s2v[c] = f * f; }
      } else if (a.gb_x) {
        const int cpg = a.N / 32;
        const half4 xv = *reinterpret_cast<const half4*>(a.gb_x + (long long)m * a.N + n);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int grp = (n + c) / cpg;
          float dyh, xhat;
          gn_bwd_term((float)o[c], (float)xv[c], a.gb_stats[(n_img * 32 + grp) * 2], a.gb_stats[(n_img * 32 + grp) * 2 + 1],
                      a.gb_gamma[n + c], a.gb_beta[n + c], a.gb_film ? a.gb_emb[(long long)n_img * a.gb_emb_ld + n + c] : 0.f,
                      a.gb_film ? a.gb_emb[(long long)n_img * a.gb_emb_ld + a.N + n + c] : 0.f, a.gb_film != 0, a.gb_act != 0,
                      dyh, xhat);
          s1v[c] = dyh;
          s2v[c] = dyh * xhat;
        }
      }
    } else if (a.out_mode == IG_OUT_F32) {
      *reinterpret_cast<f32x4*>((float*)a.out + (long long)m * a.ldo + n) = v;
    } else {
      float* o = (float*)a.out + ((long long)n_img * a.N + n) * HW + (oy * a.W + ox);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[(long long)r * HW] = v[r];
    }
  }
  if (a.stat_out || a.gb_x) {
    // 16 pixels of one image per sub-tile (HW % 16 == 0): row sums over the pixel lanes, one atomic per (channel, term)
    long long* const sdst = a.gb_x ? a.gb_csums : a.stat_out;
    const float scale_q = a.gb_x ? STAT_SCALE_SUM : STAT_SCALE_SQ;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float t1 = row16_sum(s1v[c]), t2 = row16_sum(s2v[c]);
      if (col == 0 && n + c < a.N) {
        long long* d = sdst + ((long long)n_img * a.N + n + c) * 2;
        atomicAdd(reinterpret_cast<unsigned long long*>(d), (unsigned long long)__float2ll_rn(t1 * STAT_SCALE_SUM));
        atomicAdd(reinterpret_cast<unsigned long long*>(d + 1), (unsigned long long)__float2ll_rn(t2 * scale_q));
      }
    }
  }
}

template <int MT>
int launch_skinny(const IgemmArgs& a, hipStream_t s) {
  dim3 grid(a.M / (16 * MT), (a.N + 15) / 16);
  if (a.conv3) {
    if (g_igemm_prof_start) hipExtLaunchKernelGGL((igemm_skinny_kernel<MT, true>), grid, dim3(SK_WAVES * 64), 0, s, g_igemm_prof_start, g_igemm_prof_stop, 0,
                                                   (const void*)a.X, (const void*)a.Wt, a.K, a.Cin, a.ldx, a.ldw, a.H, a.W, a.N, a);
    else hipLaunchKernelGGL((igemm_skinny_kernel<MT, true>), grid, dim3(SK_WAVES * 64), 0, s, (const void*)a.X, (const void*)a.Wt, a.K, a.Cin, a.ldx,
                            a.ldw, a.H, a.W, a.N, a);
  } else {
    if (g_igemm_prof_start) hipExtLaunchKernelGGL((igemm_skinny_kernel<MT, false>), grid, dim3(SK_WAVES * 64), 0, s, g_igemm_prof_start, g_igemm_prof_stop, 0,
                                                   (const void*)a.X, (const void*)a.Wt, a.K, a.Cin, a.ldx, a.ldw, a.H, a.W, a.N, a);
    else hipLaunchKernelGGL((igemm_skinny_kernel<MT, false>), grid, dim3(SK_WAVES * 64), 0, s, (const void*)a.X, (const void*)a.Wt, a.K, a.Cin, a.ldx,
                            a.ldw, a.H, a.W, a.N, a);
  }
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// shapes this kernel takes: one image-aligned pixel count <= 256, 64-wide K-steps, dense fp16 weights padded to 128 rows
bool igemm_skinny_applicable(const IgemmArgs& a) {
  if (a.nbatch != 1 || a.ksplit != 1 || a.M > 256 || a.M % 16 != 0 || a.N % 4 != 0) return false;
  if (a.conv3 ? (a.Cin % 64 != 0 || a.K != 9 * a.Cin) : (a.K % 64 != 0)) return false;
  if ((a.H * a.W) % 16 != 0) return false;                    // a 16-pixel sub-tile must not straddle images
  return true;
}

// mt: pixels per workgroup / 16 (1, 2 or 4); 0 = pick
int igemm_skinny_launch(const IgemmArgs& a, int mt, hipStream_t s) {
  ISHAP_REQUIRE(igemm_skinny_applicable(a), "skinny kernel: shape");
  if (mt == 0) mt = (a.M % 64 == 0 && a.M > 64) ? 4 : (a.M % 32 == 0 ? 2 : 1);
  ISHAP_REQUIRE(a.M % (16 * mt) == 0, "skinny kernel: tile");
  switch (mt) {
    case 1: return launch_skinny<1>(a, s);
    case 2: return launch_skinny<2>(a, s);
    default: return launch_skinny<4>(a, s);
  }
}
