// Shared epilogue of the implicit-GEMM kernels (igemm.hip register-staged, igemm2.hip LDS-DMA ring).
// A lane holds, for each (i, j) MFMA tile, output channels n..n+3 of one pixel m.  Besides bias / residual /
// layout handling, the epilogue can accumulate the per-channel sum and sum of squares of the fp16 values it
// stores (`stat_out`): these are the GroupNorm statistics of the NEXT layer (gd/nn.py:16-18), gathered here so no
// separate pass has to re-read the tensor.
#pragma once
#include "common.h"
#include "gn_bwd_terms.h"
#include "gn_act.h"
#include <type_traits>

// Diagnostic build only (tools/bench_igemm.hip -DIG_STAMPS): s_memtime stamps of one wave per workgroup into a buffer of
// their own (cdna_hip_programming.md 7, In-kernel stamps).  In the library the macro expands to nothing.
#ifdef IG_STAMPS
extern __device__ unsigned long long* g_ig_stamps;      // [workgroup][16]
#define IG_STAMP(slot, cond)                                                                                   \
  do {                                                                                                         \
    if (cond) {                                                                                                \
      unsigned long long t_;                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
      if ((threadIdx.x & 63) == 0)                                                                             \
        g_ig_stamps[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16 + (slot)] = t_; \
    }                                                                                                          \
  } while (0)
#else
#define IG_STAMP(slot, cond) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// Staged epilogue of the LDS-DMA kernels (round 3).  The round-2 form did everything on the four MFMA waves in the
// fragment layout: scale / bias / residual / convert / statistics for MT x NT tiles of 4 values each, ~60 instructions
// per tile behind run-time selects, then row sums by DPP chains -- 10 300 cycles of a 128x128-tile launch and 4 400 of a
// 64x64 one (s_memtime stamps, tools/bench_igemm.hip -DIG_STAMPS), on one wave per SIMD while the loader waves idled.
// Now the MFMA waves only park their raw fp32 accumulators in LDS (the ring is dead by then), and EVERY wave of the
// workgroup -- loaders and the second team included -- finishes the tile row-wise: a thread owns one 8-channel chunk of a
// few rows, so bias / FiLM-free epilogue operands are loaded once per thread, the residual and the output are whole
// 16-byte row segments (coalesced both ways).  Values are formed in the same order as before (alpha * acc, + bias,
// + bias2, + residual, one rounding to fp16), so outputs are bit-identical to the fragment-layout path.
// Statistics: forward launches (sum x, sum x^2 of the stored fp16 values = GroupNorm statistics of the next layer,
// gd/nn.py:16-18) re-read the rounded tile from LDS by column, a thread per channel pair; input-gradient launches
// (sum dyh, sum dyh*xhat, gn_bwd_terms.h) add their rows' terms in registers and meet by butterfly.  Either way the
// workgroup's totals go out as 64-bit fixed point (integer atomics commute: bitwise reproducible).
// ---------------------------------------------------------------------------------------------------------------
// block-uniform (kernel arguments only): dense fp16 outputs of an unsplit launch take the staged, all-waves epilogue
__device__ __forceinline__ bool igemm_epilogue_is_staged(const IgemmArgs& a, int batch) {
  const half_t* out16 = (const half_t*)a.out + (long long)batch * a.bso;
  const bool aligned16 = ((reinterpret_cast<unsigned long long>(out16) | reinterpret_cast<unsigned long long>(a.res)) & 15) == 0;
  return a.out_mode == IG_OUT_F16 && a.ksplit == 1 && (a.N & 7) == 0 && (a.ldo & 7) == 0 && (!a.res || (a.ldr & 7) == 0) && aligned16 &&
         (!a.gb_x || (a.N & 31) == 0);
}

template <int MT, int NT, int TMW, int TNW, int BN, int T>
__device__ __forceinline__ void igemm_epilogue_staged(const IgemmArgs& a, f32x4 (&acc)[NT][MT], int m0, int n0, int wm, int wn,
                                                      int lane, int batch, float* lds_f, bool active, int team2) {
  constexpr int BM_T = 2 * TMW;
  constexpr int LDF = BN + 4;                      // fp32 tile pitch: conflict-free ds_write_b128 of the fragments and row reads
  constexpr int LDH = BN + 8;                      // fp16 tile pitch (halfs)
  constexpr int CPRW = BN / 8;                     // 8-channel chunks per tile row
  constexpr int ITEMS = BM_T * CPRW;               // (row, chunk) items of the tile
  constexpr int RPT = (ITEMS + T - 1) / T;         // items per thread (rows RSTEP apart, same chunk)
  constexpr int RSTEP = T / CPRW;
  constexpr int NW = T / 64;
  constexpr int PAIRS = BN / 2;                    // statistics pass: a thread owns a channel pair over RPP rows
  constexpr int PARTS = (T / PAIRS) < (BM_T / 8) ? (T / PAIRS) : (BM_T / 8);
  constexpr int RPP = BM_T / PARTS;
  static_assert(T % CPRW == 0 && (ITEMS % T == 0 || ITEMS < T) && BM_T % PARTS == 0, "thread <-> tile mappings");
  const bool two = team2 >= 0;                                              // block-uniform
  float* const tileF = lds_f;                                               // [BM_T][LDF] raw accumulators (two teams: team 1's tile follows)
  half_t* const tileH = reinterpret_cast<half_t*>(lds_f + (two ? 2 : 1) * BM_T * LDF);   // [BM_T][LDH] the stored (rounded) values
  float* const slots = reinterpret_cast<float*>(tileH + BM_T * LDH);        // partial sums: [PARTS][PAIRS][4] or [NW][BN][2]
  const int t = threadIdx.x;
  const bool stamp_wave = active && wm == 0 && wn == 0;
  (void)stamp_wave;
  IG_STAMP(5, stamp_wave);
  __syncthreads();                                 // the K loop's LDS tiles are dead from here on
  if (active) {
    float* const mine = tileF + (two ? team2 * BM_T * LDF : 0);
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int i = 0; i < NT; ++i)
        *reinterpret_cast<f32x4*>(mine + (wm * TMW + j * 16 + (lane & 15)) * LDF + wn * TNW + i * 16 + (lane >> 4) * 4) = acc[i][j];
  }
  // this thread's chunk: epilogue operands once per thread, issued before the barrier so that their latency overlaps it
  const int chunk = t % CPRW, row_first = t / CPRW;
  const int n = n0 + chunk * 8;
  const bool has = row_first < BM_T && n < a.N;    // N % 8 == 0: a chunk is inside or outside as a whole
  const float alpha = a.alpha;
  const int HW = a.H * a.W;
  half_t* const out16 = (half_t*)a.out + (long long)batch * a.bso;
  f32x4 b1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, b2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  half8 rr[RPT];
  const bool want_fwd = a.stat_out != nullptr, want_gb = a.gb_x != nullptr;
  if (has) {
    if (a.bias) { b1[0] = *reinterpret_cast<const f32x4*>(a.bias + n); b1[1] = *reinterpret_cast<const f32x4*>(a.bias + n + 4); }
    if (a.bias2) { b2[0] = *reinterpret_cast<const f32x4*>(a.bias2 + n); b2[1] = *reinterpret_cast<const f32x4*>(a.bias2 + n + 4); }
    if (a.res) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int m = m0 + row_first + k * RSTEP;
        long long rrow = m;
        if (a.res_ups) {
          int img, py, px;
          if (a.hw_shift >= 0) { img = m >> a.hw_shift; const int p = m & (HW - 1); py = p >> a.w_shift; px = p & (a.W - 1); }
          else { img = m / HW; const int p = m - img * HW; py = p / a.W; px = p - py * a.W; }
          rrow = (long long)img * (HW >> 2) + (py >> 1) * (a.W >> 1) + (px >> 1);
        }
        rr[k] = *reinterpret_cast<const half8*>(a.res + rrow * a.ldr + n);
      }
    }
  }
  // GroupNorm-backward operands of this chunk (input-gradient launches only)
  float g_mu[8], g_rs[8], g_gam[8], g_bet[8], g_esc[8], g_esh[8];
  half8 gx[RPT];
  if (want_gb && has) {
    const int n_img = m0 / HW, cpg = a.N / 32;
    // every load below is UNCONDITIONAL (without FiLM the two FiLM rows read gamma and are ignored by gn_bwd_term): written as
    // `a.gb_film ? a.gb_emb[...] : 0.f` each of the eight iterations ended in an s_waitcnt vmcnt(0) -- the compiler closes every
    // conditional block that holds a load with one -- i.e. eight dependent round trips in front of this epilogue's barrier (round 6)
    const float* const ep = a.gb_film ? a.gb_emb + (long long)n_img * a.gb_emb_ld + n : a.gb_gamma + n;
    const float* const eq = a.gb_film ? ep + a.N : a.gb_gamma + n;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int g = (n + c) / cpg;
      g_mu[c] = a.gb_stats[(n_img * 32 + g) * 2];
      g_rs[c] = a.gb_stats[(n_img * 32 + g) * 2 + 1];
      g_gam[c] = a.gb_gamma[n + c];
      g_bet[c] = a.gb_beta[n + c];
      g_esc[c] = ep[c];
      g_esh[c] = eq[c];
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      gx[k] = *reinterpret_cast<const half8*>(a.gb_x + (long long)(m0 + row_first + k * RSTEP) * a.N + n);
  }
  IG_STAMP(6, stamp_wave);
  __syncthreads();
  float gs1[8], gs2[8];                            // input-gradient launches: sum dyh, sum dyh*xhat of this thread's rows (O(1) terms)
#pragma unroll
  for (int c = 0; c < 8; ++c) { gs1[c] = 0.f; gs2[c] = 0.f; }
  if (has) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int row = row_first + k * RSTEP;
      const float* src = tileF + row * LDF + chunk * 8;
      f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
      if (two) {                                   // team 0 + team 1, the order of the old merge pass
        v0 += *reinterpret_cast<const f32x4*>(src + BM_T * LDF);
        v1 += *reinterpret_cast<const f32x4*>(src + BM_T * LDF + 4);
      }
      v0 *= alpha; v1 *= alpha;                    // alpha * acc (+ bias) (+ bias2) (+ residual), in this order
      if (a.bias) { v0 += b1[0]; v1 += b1[1]; }
      if (a.bias2) { v0 += b2[0]; v1 += b2[1]; }
      if (a.res) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { v0[c] += (float)rr[k][c]; v1[c] += (float)rr[k][4 + c]; }
      }
      half8 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) { o[c] = (half_t)v0[c]; o[4 + c] = (half_t)v1[c]; }
#ifdef ABL_EPI_NOSTORE
      if (alpha == 12345.f)
#endif
      store_out16(out16 + (long long)(m0 + row) * a.ldo + n, o);
      if (want_fwd) {
        *reinterpret_cast<half8*>(tileH + row * LDH + chunk * 8) = o;    // the statistics pass reads the stored values by column
      } else if (want_gb) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float dyh, xhat;
          gn_bwd_term((float)o[c], (float)gx[k][c], g_mu[c], g_rs[c], g_gam[c], g_bet[c], g_esc[c], g_esh[c], a.gb_film != 0,
                      a.gb_act != 0, dyh, xhat);
          gs1[c] += dyh;
          gs2[c] = fmaf(dyh, xhat, gs2[c]);
        }
      }
    }
  }
  IG_STAMP(7, stamp_wave);
  if (!(want_fwd || want_gb)) { IG_STAMP(10, stamp_wave); return; }      // block-uniform (kernel arguments)
  const int n_img = a.hw_shift >= 0 ? (m0 >> a.hw_shift) : m0 / HW;     // a tile never straddles images (HW % BM == 0)
  if (want_fwd) {
    // Per-channel sum and sum of squares of the stored values about a PIVOT P = the channel's value in the tile's first row
    // (the same for every thread, so partial sums simply add): sum(x - P), sum((x - P)^2) stay small when the channel sits
    // at |mean| >> std, where fp32 sums of x^2 cancel (128 squares of 100.1 +- 0.1: error ~0.1 against a variance
    // contribution of 1.3 -- GroupNorm32, gd/nn.py:16-18, would come out several per cent off).
    __syncthreads();
    if (t < PARTS * PAIRS) {
      const int pair = t % PAIRS, part = t / PAIRS;
      const half2v pv = *reinterpret_cast<const half2v*>(tileH + pair * 2);
      const float P0 = (float)pv[0], P1 = (float)pv[1];
      float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
      for (int r = 0; r < RPP; ++r) {
        const half2v x = *reinterpret_cast<const half2v*>(tileH + (part * RPP + r) * LDH + pair * 2);
        const float d0 = (float)x[0] - P0, d1 = (float)x[1] - P1;
        s0 += d0; q0 = fmaf(d0, d0, q0);
        s1 += d1; q1 = fmaf(d1, d1, q1);
      }
      *reinterpret_cast<f32x4*>(slots + (part * PAIRS + pair) * 4) = (f32x4){s0, q0, s1, q1};
    }
    IG_STAMP(8, stamp_wave);
    __syncthreads();
    IG_STAMP(9, stamp_wave);
    if (t < BN * 2) {
      const int nl = t >> 1, k = t & 1;
      if (n0 + nl < a.N) {
        // parts add in double (fixed order); then to pivot 0:  sum x = S + n P,  sum x^2 = Q + P (2 S + n P)
        double S = 0.0, Q = 0.0;
#pragma unroll
        for (int part = 0; part < PARTS; ++part) {
          const f32x4 r = *reinterpret_cast<const f32x4*>(slots + (part * PAIRS + (nl >> 1)) * 4);
          S += (double)r[(nl & 1) * 2];
          Q += (double)r[(nl & 1) * 2 + 1];
        }
        const double P = (double)(float)tileH[nl], nn = (double)BM_T;
        const double v = k ? Q + P * (2.0 * S + nn * P) : S + nn * P;
#ifdef ABL_STAT_COPIES      // harness probe (tools/experiments/stat_probe2.sh): spread the same-address atomics over copies of the table
        atomicAdd(reinterpret_cast<unsigned long long*>(a.stat_out + (long long)((m0 / BM_T) % ABL_STAT_COPIES) * a.N * 2 +
                                                        ((long long)n_img * a.N + n0 + nl) * 2 + k),
#else
        atomicAdd(reinterpret_cast<unsigned long long*>(a.stat_out + ((long long)n_img * a.N + n0 + nl) * 2 + k),
#endif
                  (unsigned long long)__double2ll_rn(v * (double)(k ? STAT_SCALE_SQ : STAT_SCALE_SUM)));
      }
    }
  } else {
    // the lanes of a wave that share a chunk (lane, lane + CPRW, ...) add by butterfly, one slot per (wave, channel, term)
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int o = CPRW; o < 64; o <<= 1) { gs1[c] += __shfl_xor(gs1[c], o); gs2[c] += __shfl_xor(gs2[c], o); }
    }
    IG_STAMP(8, stamp_wave);
    if (lane < CPRW) {
      float* dst = slots + ((t >> 6) * BN + lane * 8) * 2;
#pragma unroll
      for (int c = 0; c < 8; ++c) { dst[c * 2] = gs1[c]; dst[c * 2 + 1] = gs2[c]; }
    }
    __syncthreads();
    if (t < BN * 2) {
      const int nl = t >> 1, k = t & 1;
      if (n0 + nl < a.N) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += (double)slots[(w * BN + nl) * 2 + k];
        atomicAdd(reinterpret_cast<unsigned long long*>(a.gb_csums + ((long long)n_img * a.N + n0 + nl) * 2 + k),
                  (unsigned long long)__double2ll_rn(v * (double)STAT_SCALE_SUM));
      }
    }
  }
  IG_STAMP(10, stamp_wave);
}

// STAGE_THREADS > 0 (the LDS-DMA kernels; = threads of the workgroup, all of which call this function): fp16 outputs
// of an unsplit launch are collected in LDS and leave as full 16-byte-per-lane row segments written by every wave,
// loader waves included.  The fragment layout's own stores are 8 bytes per lane over 16 rows: issue-bound, ~9 us of
// a 128^2-map launch against ~2 us this way (tools/experiments/fixed_cost_probe2.sh).  Values and their order of evaluation are
// unchanged, so results are bit-identical to the direct path.
// `team2` >= 0 (two-team kernels on the staged path only): this wave belongs to team `team2`, and BOTH teams hold partial
// accumulators that the staged epilogue adds up (team 0 + team 1) while it finishes the rows -- the separate merge pass
// through LDS (two barriers, 1 100 cycles of a 64x64 launch) is gone.  -1: `acc` is complete in the waves flagged `active`.
template <int MT, int NT, int TMW, int TNW, int BN, int STAGE_THREADS = 0>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, f32x4 (&acc)[NT][MT], int m0, int n0, int wm, int wn,
                                               int lane, int batch, int ks_id, float* lds_f, bool active = true, int team2 = -1) {
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
  if constexpr (STAGE_THREADS > 0) {
    if (igemm_epilogue_is_staged(a, batch)) {
      igemm_epilogue_staged<MT, NT, TMW, TNW, BN, STAGE_THREADS>(a, acc, m0, n0, wm, wn, lane, batch, lds_f, active, team2);
      return;
    }
  }
  constexpr bool staged = false;                 // the fragment-layout path below: split-K slices, fp32 / NCHW outputs, odd shapes
  half_t* const tile = reinterpret_cast<half_t*>(lds_f);
  const bool stamp_wave = active && wm == 0 && wn == 0;
  (void)stamp_wave;
  IG_STAMP(5, stamp_wave);
  if (staged) __syncthreads();                   // the K loop's LDS tiles are dead from here on
  IG_STAMP(6, stamp_wave);
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
  // (~4 us of a 128x128-tile launch, tools/experiments/fixed_cost_probe3.sh).
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
  IG_STAMP(7, stamp_wave);
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
  IG_STAMP(8, stamp_wave);
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
    IG_STAMP(9, stamp_wave);
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
  IG_STAMP(10, stamp_wave);
}
