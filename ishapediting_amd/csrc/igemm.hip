// Implicit-GEMM on the CDNA4 matrix cores: every 3x3 / 1x1 convolution of the UNet
// (reference: guided_diffusion/unet.py:185,211,222,286,294,482,615 -- all stock
// torch conv ops there) and every attention matmul (unet.py:349-353) runs through
// this one kernel family.
//
//   D[n][m] = sum_k Wt[n][k] * X(m,k)        (fp16 operands, fp32 accumulate)
//
// m = output pixel (NHWC row), n = output channel, k = tap*Cin + c.  The weight tile is
// the MFMA "A" operand and the pixel tile the "B" operand, so each lane ends up holding
// 4 consecutive output channels of one pixel: an 8-byte NHWC store, no shuffle.
//
// Block = 256 threads = 4 waves (WM x WN).  Tiles are staged global -> registers -> LDS
// (double-buffered, one barrier per K-step); the 16-byte chunk index is XOR-swizzled with
// (row>>1) so both the ds_write_b128 and the ds_read_b128 fragment reads are
// bank-conflict free (lane-group table of the gfx950 LDS).
#include "common.h"
#include "igemm_epilogue.h"
#include "gn_bwd_terms.h"

template <int BM, int BN, int BK, int WM, int WN, bool CONV3>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmArgs a) {
  constexpr int CPR = BK / 8;               // 16-byte chunks per tile row
  constexpr int XL = BM * CPR / 256;        // chunks per thread, X tile
  constexpr int WL = BN * CPR / 256;
  constexpr int TMW = BM / WM, TNW = BN / WN;
  constexpr int MT = TMW / 16, NT = TNW / 16;
  static_assert(XL >= 1 && WL >= 1, "tile too small for 256 threads");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  half_t* sX = reinterpret_cast<half_t*>(smem_raw);          // [2][BM*BK]
  half_t* sW = sX + 2 * BM * BK;                             // [2][BN*BK]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int batch = blockIdx.z / a.ksplit;
  const int ks_id = blockIdx.z % a.ksplit;

  const half_t* X = a.X + (long long)batch * a.bsx;
  const half_t* Wt = a.Wt + (long long)batch * a.bsw;

  const int KS = a.K / BK;
  const int per = (KS + a.ksplit - 1) / a.ksplit;
  const int ks0 = ks_id * per;
  const int ks1 = min(KS, ks0 + per);
  const int steps_per_tap = CONV3 ? a.Cin / BK : KS;
  const int HW = a.H * a.W;

  // per-thread loader coordinates
  int xrow[XL], xch[XL], xn[XL], xy[XL], xx[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    int idx = tid + i * 256;
    xrow[i] = idx / CPR;
    xch[i] = idx % CPR;
    int m = m0 + xrow[i];
    if (CONV3) {
      xn[i] = m / HW;
      int p = m - xn[i] * HW;
      xy[i] = p / a.W;
      xx[i] = p - xy[i] * a.W;
    } else {
      xn[i] = m; xy[i] = 0; xx[i] = 0;
    }
  }
  int wrow[WL], wch[WL];
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    int idx = tid + i * 256;
    wrow[i] = idx / CPR;
    wch[i] = idx % CPR;
  }

  half8 rx[XL], rw[WL];
  auto load_tiles = [&](int ks) {
    int tap = 0, c0;
    if (CONV3) { tap = ks / steps_per_tap; c0 = (ks - tap * steps_per_tap) * BK; }
    else c0 = ks * BK;
    const int dy = CONV3 ? tap / 3 - 1 : 0;
    const int dx = CONV3 ? tap % 3 - 1 : 0;
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (CONV3) {
        int yy = xy[i] + dy, xc = xx[i] + dx;
        if (yy >= 0 && yy < a.H && xc >= 0 && xc < a.W) {
          long long src;
          if (a.ups) src = (long long)xn[i] * (HW >> 2) + (yy >> 1) * (a.W >> 1) + (xc >> 1);
          else src = (long long)xn[i] * HW + yy * a.W + xc;
          v = *reinterpret_cast<const half8*>(X + src * a.ldx + c0 + xch[i] * 8);
        }
      } else {
        v = *reinterpret_cast<const half8*>(X + (long long)xn[i] * a.ldx + c0 + xch[i] * 8);
      }
      rx[i] = v;
    }
    const int kofs = CONV3 ? tap * a.Cin + c0 : c0;
#pragma unroll
    for (int i = 0; i < WL; ++i)
      rw[i] = *reinterpret_cast<const half8*>(Wt + (long long)(n0 + wrow[i]) * a.ldw + kofs + wch[i] * 8);
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      int ph = xch[i] ^ ((xrow[i] >> 1) & (CPR - 1));
      *reinterpret_cast<half8*>(sX + buf * BM * BK + xrow[i] * BK + ph * 8) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      int ph = wch[i] ^ ((wrow[i] >> 1) & (CPR - 1));
      *reinterpret_cast<half8*>(sW + buf * BN * BK + wrow[i] * BK + ph * 8) = rw[i];
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (ks0 < ks1) {
    load_tiles(ks0);
    store_tiles(0);
    __syncthreads();
    int buf = 0;
    for (int ks = ks0; ks < ks1; ++ks) {
      const bool more = ks + 1 < ks1;
      if (more) load_tiles(ks + 1);
      const half_t* bx = sX + buf * BM * BK;
      const half_t* bw = sW + buf * BN * BK;
#pragma unroll
      for (int kk = 0; kk < BK / 32; ++kk) {
        half8 xf[MT], wf[NT];
        const int ch = (lane >> 4) + 4 * kk;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          int row = wm * TMW + j * 16 + (lane & 15);
          xf[j] = *reinterpret_cast<const half8*>(bx + row * BK + ((ch ^ ((row >> 1) & (CPR - 1))) * 8));
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          int row = wn * TNW + i * 16 + (lane & 15);
          wf[i] = *reinterpret_cast<const half8*>(bw + row * BK + ((ch ^ ((row >> 1) & (CPR - 1))) * 8));
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = 0; j < MT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
      if (more) store_tiles(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  igemm_epilogue<MT, NT, TMW, TNW, BN>(a, acc, m0, n0, wm, wn, lane, batch, ks_id, reinterpret_cast<float*>(smem_raw));
}

// out = alpha * sum_z ws[z] (+bias)(+res).  Block = 16 rows x 64 channels (thread = one row, 4 channels); when
// `stat_out` is set the block also adds its per-channel (sum, sum of squares) of the stored fp16 values.
__global__ __launch_bounds__(256) void igemm_splitk_reduce(IgemmArgs a) {
  __shared__ float red[16][64][2];
  const int nq = (a.N + 63) / 64;                 // channel groups of 64
  const int rb = a.M / 16;                        // row blocks
  const int bid = blockIdx.x;
  const int cg = bid % nq;
  const int rbi = (bid / nq) % rb;
  const int batch = bid / (nq * rb);
  const int r = threadIdx.x >> 4, qd = threadIdx.x & 15;
  const int m = rbi * 16 + r;
  const int n = cg * 64 + qd * 4;
  const bool ok = n < a.N;
  const int HW = a.H * a.W;
  half4 o = {0, 0, 0, 0};
  if (ok) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const float* wp = a.ws + ((long long)batch * a.M + m) * a.N + n;
    const long long zs = (long long)a.nbatch * a.M * a.N;
    int z = 0;
    for (; z + 4 <= a.ksplit; z += 4) {             // four slices in flight; summed in slice order
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(wp + (z + 0) * zs), p1 = *reinterpret_cast<const f32x4*>(wp + (z + 1) * zs);
      const f32x4 p2 = *reinterpret_cast<const f32x4*>(wp + (z + 2) * zs), p3 = *reinterpret_cast<const f32x4*>(wp + (z + 3) * zs);
      v += p0; v += p1; v += p2; v += p3;
    }
    for (; z < a.ksplit; ++z) v += *reinterpret_cast<const f32x4*>(wp + z * zs);
    v *= a.alpha;
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.bias2) v += *reinterpret_cast<const f32x4*>(a.bias2 + n);
    int n_img = 0, py = 0, px = 0;
    if (a.res_ups || a.out_mode == IG_OUT_NCHW_F32) {
      n_img = m / HW;
      int p = m - n_img * HW;
      py = p / a.W;
      px = p - py * a.W;
    }
    if (a.res) {
      long long rrow = a.res_ups ? ((long long)n_img * (HW >> 2) + (py >> 1) * (a.W >> 1) + (px >> 1)) : m;
      half4 rr = *reinterpret_cast<const half4*>(a.res + rrow * a.ldr + n);
      v[0] += (float)rr[0]; v[1] += (float)rr[1]; v[2] += (float)rr[2]; v[3] += (float)rr[3];
    }
    if (a.out_mode == IG_OUT_F16) {
      o = (half4){(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      *reinterpret_cast<half4*>((half_t*)a.out + (long long)batch * a.bso + (long long)m * a.ldo + n) = o;
    } else if (a.out_mode == IG_OUT_F32) {
      *reinterpret_cast<f32x4*>((float*)a.out + (long long)batch * a.bso + (long long)m * a.ldo + n) = v;
    } else {
      float* op = (float*)a.out + ((long long)n_img * a.N + n) * HW + (py * a.W + px);
      for (int k = 0; k < 4; ++k) op[(long long)k * HW] = v[k];
    }
  }
  if (a.stat_out || a.gb_x) {
    long long* const sdst = a.gb_x ? a.gb_csums : a.stat_out;
    const float scale_q = a.gb_x ? STAT_SCALE_SUM : STAT_SCALE_SQ;
    if (a.gb_x) {                              // GroupNorm-backward sums of the stored gradient (see common.h)
      const int n_img = m / HW, cpg = a.N / 32;
      half4 xv = {0, 0, 0, 0};
      if (ok) xv = *reinterpret_cast<const half4*>(a.gb_x + (long long)m * a.N + n);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float dyh = 0.f, xhat = 0.f;
        if (ok) {
          const int g = (n + c) / cpg;
          gn_bwd_term((float)o[c], (float)xv[c], a.gb_stats[(n_img * 32 + g) * 2], a.gb_stats[(n_img * 32 + g) * 2 + 1],
                      a.gb_gamma[n + c], a.gb_beta[n + c], a.gb_film ? a.gb_emb[(long long)n_img * a.gb_emb_ld + n + c] : 0.f,
                      a.gb_film ? a.gb_emb[(long long)n_img * a.gb_emb_ld + a.N + n + c] : 0.f, a.gb_film != 0, a.gb_act != 0, dyh, xhat);
        }
        red[r][qd * 4 + c][0] = dyh;
        red[r][qd * 4 + c][1] = dyh * xhat;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float f = (float)o[c];
        red[r][qd * 4 + c][0] = f;
        red[r][qd * 4 + c][1] = f * f;
      }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 128) {
      const int ch = t >> 1, k = t & 1;
      // double: f and f*f of an fp16 value are exact, and so is their sum -- an fp32 sum of 16 squares at |mean| >> std
      // already loses the variance (see igemm_epilogue.h)
      double acc = 0.0;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc += (double)red[rr][ch][k];
      const int nn = cg * 64 + ch;
      if (nn < a.N)
        atomicAdd(reinterpret_cast<unsigned long long*>(sdst + ((long long)((rbi * 16) / HW) * a.N + nn) * 2 + k),
                  (unsigned long long)__double2ll_rn(acc * (double)(k ? scale_q : STAT_SCALE_SUM)));
    }
  }
}

// ---- optional per-launch timing (bench.py's roofline leg): HIP events on the launch stream around the main kernel ----
#include <vector>
struct ProfRec { hipEvent_t a, b, c; double flops; int variant; int M, N, K, conv3, big, ksplit; };
static bool g_prof_on = false;
hipEvent_t g_igemm_prof_start = nullptr, g_igemm_prof_stop = nullptr;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;
static hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
extern "C" int ishap_profile_begin(void) {
  for (auto& r : g_prof) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); if (r.c) g_prof_pool.push_back(r.c); }
  g_prof.clear();
  g_prof_on = true;
  return 0;
}
// out[v*3 + {0,1,2}] = {launches, total milliseconds, algorithmic FLOPs} for variant v:
//   0 conv3x3 128x128 tile, 1 conv3x3 64x64 tile, 2 GEMM 128x128 tile, 3 GEMM 64x64 tile, 4 conv3x3 64x64 two-team,
//   5 small-map (skinny) GEMM kernel, 6 register-staged (BK = 32) kernel, 7 unused (was conv3_small, removed in round 6),
//   8 / 9 / 10 the dx-reuse conv kernel (igemm4.hip): 128x128 tile / 64x64 tile / 64x64 two-team, 11 its sliced launches on
//   the 8x8 maps, 12 its 128x64 tiles
extern "C" int ishap_profile_end(double* out, int nvar) {
  g_prof_on = false;
  for (int i = 0; i < nvar * 3; ++i) out[i] = 0.0;
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.b) != hipSuccess) return -1;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return -1;
    if (r.variant < nvar) { out[r.variant * 3] += 1.0; out[r.variant * 3 + 1] += ms; out[r.variant * 3 + 2] += r.flops; }
  }
  return 0;
}

// Per-shape breakdown of the launches recorded since ishap_profile_begin (call before ishap_profile_end resets
// nothing; both may be called): CSV lines "M,N,K,conv3,tile,ksplit,launches,main_ms,reduce_ms,gflop" into buf.
#include <map>
#include <cstring>
#include <array>
#include <string>
extern "C" int ishap_profile_shapes(char* buf, int cap) {
  std::map<std::array<int, 6>, std::array<double, 4>> agg;
  for (auto& r : g_prof) {
    if (hipEventSynchronize(r.c ? r.c : r.b) != hipSuccess) return -1;
    float ms = 0.f, ms2 = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return -1;
    if (r.c && hipEventElapsedTime(&ms2, r.b, r.c) != hipSuccess) return -1;
    auto& v = agg[{r.M, r.N, r.K, r.conv3, r.big ? 128 : 64, r.ksplit}];
    v[0] += 1.0; v[1] += ms; v[2] += ms2; v[3] += r.flops * 1e-9;
  }
  std::string out;
  char line[160];
  for (auto& kv : agg) {
    snprintf(line, sizeof line, "%d,%d,%d,%d,%d,%d,%.0f,%.4f,%.4f,%.3f\n", kv.first[0], kv.first[1], kv.first[2], kv.first[3],
             kv.first[4], kv.first[5], kv.second[0], kv.second[1], kv.second[2], kv.second[3]);
    out += line;
  }
  if ((int)out.size() + 1 > cap) return -2;
  memcpy(buf, out.c_str(), out.size() + 1);
  return (int)agg.size();
}

int igemm2_launch_main(const IgemmArgs& a, bool big, hipStream_t s);   // igemm2.hip (LDS-DMA ring, BK = 64)
bool igemm2_two_teams(const IgemmArgs& a, bool big);
bool igemm4_applicable(const IgemmArgs& a, bool big);                  // igemm4.hip (3x3 with each activation slab staged once for dx = -1, 0, +1)
bool igemm4_two_teams(const IgemmArgs& a, bool big);
bool igemm4_tall_tiles(const IgemmArgs& a, bool big);
int igemm4_launch_main(const IgemmArgs& a, bool big, hipStream_t s);
// ISHAP_IGEMM4: 0 = never, 1 = 128x128 tiles only, 2 (default) = every shape igemm4 takes
static bool igemm4_wanted(const IgemmArgs& a, bool big) {
  static const int use4 = [] { const char* e = getenv("ISHAP_IGEMM4"); return e ? atoi(e) : 2; }();
  return use4 && (use4 > 1 || big) && igemm4_applicable(a, big);
}

template <int BM, int BN, int BK, int WM, int WN, bool CONV3>
static int launch_cfg(const IgemmArgs& a, hipStream_t s) {
  int prof_slot = -1;
  auto fire = [&]() -> int {
    if constexpr (BK == 64) {
      if (CONV3 && igemm4_wanted(a, BM == 128)) return igemm4_launch_main(a, BM == 128, s);
      return igemm2_launch_main(a, BM == 128, s);
    } else {
      // K not a multiple of 64 (tiny configurations; the full model's stem is padded to 128 channels): the register-staged kernel
      constexpr size_t smem = 2 * (size_t)(BM + BN) * BK * sizeof(half_t);
      auto kern = igemm_kernel<BM, BN, BK, WM, WN, CONV3>;
      ISHAP_TRY(ishap_set_max_lds((const void*)kern, (int)smem));
      const dim3 grid(a.M / BM, ceil_div(a.N, BN), a.nbatch * a.ksplit);
      if (g_igemm_prof_start) hipExtLaunchKernelGGL(kern, grid, dim3(256), smem, s, g_igemm_prof_start, g_igemm_prof_stop, 0, a);
      else hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, a);
      return 0;
    }
  };
  if (g_prof_on) {
    ProfRec r;
    r.a = prof_event(); r.b = prof_event(); r.c = nullptr;
    r.flops = 2.0 * a.M * a.N * a.K * a.nbatch * a.flops_scale;
    // one variant per kernel symbol: 0/1 conv 128/64 tile, 2/3 GEMM 128/64 tile, 4 two-team 64-tile conv,
    // 5 small-map kernel (set in igemm_launch), 6 register-staged BK=32 kernel (stem conv)
    r.variant = BK == 64 ? ((CONV3 ? 0 : 2) + (BM == 128 ? 0 : 1)) : 6;
    if (BK == 64 && igemm2_two_teams(a, BM == 128)) r.variant = 4;
    if (BK == 64 && CONV3 && igemm4_wanted(a, BM == 128))
      r.variant = BM == 128 ? 8 : (a.W == 8 ? 11 : (igemm4_tall_tiles(a, false) ? 12 : (igemm4_two_teams(a, false) ? 10 : 9)));
    r.M = a.M * a.nbatch; r.N = a.N; r.K = a.K; r.conv3 = CONV3; r.big = BM == 128; r.ksplit = a.ksplit;
    g_igemm_prof_start = r.a; g_igemm_prof_stop = r.b;      // attached to the dispatch: kernel begin / end timestamps
    const int rc = fire();
    g_igemm_prof_start = nullptr; g_igemm_prof_stop = nullptr;
    if (rc) return rc;
    prof_slot = (int)g_prof.size();
    g_prof.push_back(r);
  } else {
    ISHAP_TRY(fire());
  }
  ISHAP_CHECK_HIP(hipGetLastError());
  if (a.ksplit > 1 && !a.defer_reduce) {
    const unsigned rblocks = (unsigned)((long long)a.nbatch * (a.M / 16) * ((a.N + 63) / 64));
    hipLaunchKernelGGL(igemm_splitk_reduce, dim3(rblocks), dim3(256), 0, s, a);
    ISHAP_CHECK_HIP(hipGetLastError());
    if (prof_slot >= 0) {
      g_prof[prof_slot].c = prof_event();
      (void)hipEventRecord(g_prof[prof_slot].c, s);
    }
  }
  return 0;
}

// Tile and split-K policy.  256 CUs: prefer the 128x128 tile when it alone yields >= ~200 workgroups, otherwise the
// 64x64 tile; split K while the grid stays below ~224 workgroups (the small-map, weight-streaming layers),
// keeping >= 6 K-steps of 64 per slice so the fp32 partial traffic stays below the weight traffic.
static bool igemm_use_big(int M, int N, int nbatch) {
  if (M % 128 != 0 || N < 128) return false;
  long long blocks = (long long)(M / 128) * ceil_div(N, 128) * nbatch;
  static const int big_min = [] { const char* e = getenv("ISHAP_BIG_MIN"); return e ? atoi(e) : 192; }();
  return blocks >= big_min;
}
int igemm_pick_ksplit(int M, int N, int K, int nbatch, bool pending) {
  const bool big = igemm_use_big(M, N, nbatch);
  const int bm = big ? 128 : 64, bn = big ? 128 : 64;
  long long blocks = (long long)(M / bm) * ceil_div(N, bn) * nbatch;
  int ks = K / 64;
  // policy constants, each swept in situ (tools/experiments/sweep_split_policy.sh, profiles/round4_env_ab_pending_split.txt,
  // profiles/round5_ab_policy_resweep.txt: all flat within +-0.5 % around these values)
  constexpr int nosplit = 36;       // K-steps below which a launch followed by a reduce launch is not split (24 / 48: +0.5 %)
  constexpr int fill = 224;         // split while the grid stays below this many workgroups (plateau 208 .. 256)
  constexpr int minsteps = 6;       // K-steps left per slice at least
  // slices whose consumer adds them up cost no reduce launch: thresholds of their own (12 / 3 against 36 / 6: 0.1783 -> 0.1777 s/shape)
  constexpr int p_nosplit = 12, p_minsteps = 3;
  if (ks < (pending ? p_nosplit : nosplit)) return 1;        // below ~36 K-steps the extra reduce launch (~5.5 us) costs more than the split saves (harness sweep: ~48; in situ: 36)
  int split = 1;
  while (blocks * split < fill && ks / (split * 2) >= (pending ? p_minsteps : minsteps) && split < 32) split *= 2;
  return split;
}

// the stand-alone reduce of deferred split-K slices (a.ws, a.ksplit, bias / residual / output fields as in the main launch)
int igemm_reduce_launch(const IgemmArgs& a, hipStream_t s) {
  ISHAP_REQUIRE(a.ksplit > 1 && a.ws && a.M % 16 == 0 && a.N % 4 == 0, "reduce: split-K slices of a finished launch");
  const unsigned rblocks = (unsigned)((long long)a.nbatch * (a.M / 16) * ((a.N + 63) / 64));
  hipLaunchKernelGGL(igemm_splitk_reduce, dim3(rblocks), dim3(256), 0, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

bool igemm_skinny_applicable(const IgemmArgs& a);                      // igemm_skinny.hip (small maps, one launch)
int igemm_skinny_launch(const IgemmArgs& a, int mt, hipStream_t s);

// 1x1 GEMMs on the 8x8 maps: the one-launch skinny kernel beats the tiled one there (measured in situ: 7.8 vs 11.4 us
// at 64 x 1024 x 1024); for 3x3 layers it re-reads the im2col fragments once per 16-channel tile and loses.
static bool use_skinny(const IgemmArgs& a) {
  static const int on = [] { const char* e = getenv("ISHAP_SKINNY"); return e ? atoi(e) : 1; }();   // 0: the tiled kernel + reduce instead
  // in situ it wins at M = 64 (-3..5 us per launch) and loses at M = 256; 3x3 layers never (L2-bound fragment re-reads)
  return on && !a.conv3 && a.M <= 64 && igemm_skinny_applicable(a);
}

int igemm_launch(const IgemmArgs& a, hipStream_t s) {
  ISHAP_REQUIRE(a.M % 64 == 0, "M must be a multiple of 64");
  ISHAP_REQUIRE(a.N % 4 == 0, "N must be a multiple of 4");
  ISHAP_REQUIRE(a.K % 32 == 0, "K must be a multiple of 32");
  ISHAP_REQUIRE(!a.conv3 || (a.Cin % 32 == 0 && a.K == 9 * a.Cin + a.K2), "conv3: K = 9*Cin (+ K2), Cin % 32 == 0");
  ISHAP_REQUIRE(a.K2 == 0 || (a.conv3 && a.X2 && a.K2 % 64 == 0 && a.Cin % 64 == 0 && a.ldx2 % 8 == 0 && a.nbatch == 1),
                "folded 1x1 source: 3x3 launch, 64-wide K-steps");
  ISHAP_REQUIRE(a.ksplit == 1 || a.ws != nullptr, "split-K needs a workspace");
  ISHAP_REQUIRE(!a.gb_x || (!a.stat_out && a.out_mode == IG_OUT_F16 && a.ldo == a.N && a.N % 32 == 0 && a.nbatch == 1 && a.gb_csums &&
                            a.gb_stats && a.gb_gamma && a.gb_beta && (!a.gb_film || a.gb_emb)),
                "fused GroupNorm-backward sums: fp16 dense output, N % 32 == 0, no forward statistics");
  ISHAP_REQUIRE(a.ldx % 8 == 0 && a.ldw % 8 == 0, "row strides must keep 16-byte alignment");
  if (use_skinny(a)) {
    if (!g_prof_on) return igemm_skinny_launch(a, 0, s);
    ProfRec r;
    r.a = prof_event(); r.b = prof_event(); r.c = nullptr;
    r.flops = 2.0 * a.M * a.N * a.K * a.flops_scale;
    r.variant = 5;
    r.M = a.M; r.N = a.N; r.K = a.K; r.conv3 = a.conv3; r.big = false; r.ksplit = 0;      // ksplit 0 marks the skinny kernel
    g_igemm_prof_start = r.a; g_igemm_prof_stop = r.b;
    const int rc = igemm_skinny_launch(a, 0, s);
    g_igemm_prof_start = nullptr; g_igemm_prof_stop = nullptr;
    g_prof.push_back(r);
    return rc;
  }
  const bool k64 = a.conv3 ? (a.Cin % 64 == 0) : (a.K % 64 == 0);
  bool big = igemm_use_big(a.M, a.N, a.nbatch) && k64;      // the register-staged BK = 32 kernel (tiny configurations only) is built with 64x64 tiles
  if (a.stat_out || a.gb_x) {
    // the statistics epilogues file a whole tile under image m0 / HW: a tile must not straddle two images
    const int hw = a.H * a.W;
    ISHAP_REQUIRE(hw > 0 && hw % 64 == 0, "fused GroupNorm sums need H*W to be a multiple of the 64-row tile");
    if (hw % 128 != 0) big = false;
  }
#define IG_DISPATCH(BM, BN, WM_, WN_)                                                          \
  do {                                                                                         \
    if (a.conv3) {                                                                             \
      if (k64) return launch_cfg<BM, BN, 64, WM_, WN_, true>(a, s);                            \
      return launch_cfg<BM, BN, 32, WM_, WN_, true>(a, s);                                     \
    } else {                                                                                   \
      if (k64) return launch_cfg<BM, BN, 64, WM_, WN_, false>(a, s);                           \
      return launch_cfg<BM, BN, 32, WM_, WN_, false>(a, s);                                    \
    }                                                                                          \
  } while (0)
  if (big) {
    if (a.conv3) return launch_cfg<128, 128, 64, 2, 2, true>(a, s);
    return launch_cfg<128, 128, 64, 2, 2, false>(a, s);
  }
  IG_DISPATCH(64, 64, 2, 2);
#undef IG_DISPATCH
}
