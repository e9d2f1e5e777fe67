// Implicit GEMM, second generation: LDS-DMA ring.
// Same contract and output mapping as igemm.hip (weights = MFMA A operand, pixels = B operand, a lane
// owns 4 consecutive output channels of one pixel), but operand tiles go global -> LDS directly with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write), into an NST-deep ring.  A wave's instruction writes
// 1 KiB linearly (wave-uniform base + lane*16), so the bank swizzle is applied to the per-lane SOURCE chunk and
// again on the fragment read.  Zero padding of the 3x3 gather points the lane at a zero line instead of
// predicating.  One raw s_barrier per K-step; loads stay in flight across it behind a counted s_waitcnt vmcnt.
#include "common.h"
#include "igemm_epilogue.h"
#include <type_traits>

__device__ __attribute__((aligned(128))) half_t g_zero_line[64];   // zero-initialised: source of padded rows

typedef __attribute__((address_space(3))) void lds_void;

#ifndef IG2_LOADERS
#define IG2_LOADERS 4          // loader waves per workgroup (4 measured faster than 8)
#endif
// L2 prefetch of the weight panel (round 4).  The CUs of an XCD walk K in step, so every ring stage holds lines no CU of
// the XCD has touched before (this K-step's slab of the weight panel): each stage lands at the latency of an L2 MISS and
// the K-step takes miss latency / (NST - 1).  MFMA wave 0 (1: the second 64 rows of a 128-row panel) therefore touches one
// dword of every 128-byte weight line IG2_PF_DIST K-steps ahead of the step being multiplied, by LDS-DMA into a 256-byte
// scratch (no register to keep alive, no effect on the loader waves' vmcnt counting): by the time the loaders stage that
// step its lines are L2 hits.  IG2_PF_SHARE: the m-tiles of an XCD that share a panel take turns (step % sharers).
#ifndef IG2_PF_DIST
#define IG2_PF_DIST 0
#endif
#ifndef IG2_PF_SHARE
#define IG2_PF_SHARE 0
#endif

// HALVES = 2: the workgroup is two such 8-wave teams, each with its own ring, working on the two halves of the K range
// and meeting once in LDS at the end -- split-K by two without partial tiles in HBM or a reduce launch.  It gives a CU
// that holds a single workgroup (grids of <= 256 tiles) the DMA/MFMA overlap two co-resident workgroups would have.
template <int BM, int BN, int NST, bool CONV3, int HALVES = 1>
__global__ __launch_bounds__((256 + 64 * IG2_LOADERS) * HALVES) void igemm2_kernel(const void* hX, const void* hWt, int hK, int hCin, int hldx, int hldw, int hH, int hW,
                                                                                   int hksplit, int hnwg, unsigned hpacked, IgemmArgs a) {
  // plain leading scalars (14 dwords): only those are preloaded into SGPRs (a by-value struct is not); see IgemmHot
  // (void pointers: the profiler's demangler does not know _Float16 and would list the kernels by their mangled names)
  const IgemmHot h{(const half_t*)hX, (const half_t*)hWt, hK, hCin, hldx, hldw, hH, hW, hksplit, hnwg, hpacked};
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses device-only builtins; the host pass only needs the launch stub
  constexpr int BK = 64;
  constexpr int CPR = 8;                 // 16-byte chunks per tile row
  constexpr int RPI = 8;                 // rows covered by one wave-instruction (64 lanes / CPR)
  constexpr int XI = BM / RPI / 4;       // wave-instructions per wave per stage, X tile
  constexpr int WI = BN / RPI / 4;
  constexpr int STAGE = (BM + BN) * BK;  // halfs per ring slot
  constexpr int TMW = BM / 2, TNW = BN / 2;
  constexpr int MT = TMW / 16, NT = TNW / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  // 8 waves: 0-3 compute (2x2 MFMA sub-tiles), 4-7 only issue the global->LDS DMA.  A DMA instruction costs its wave
  // ~150 cycles of issue time; on dedicated loader waves that time overlaps the consumers' MFMAs instead of preceding them.
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  IG_STAMP(0, wave_all == 0);
  const int team = HALVES == 2 ? wave_all / (4 + IG2_LOADERS) : 0;
  const int wave8 = wave_all - team * (4 + IG2_LOADERS);
  half_t* lds = reinterpret_cast<half_t*>(smem_raw) + team * (NST * STAGE);
  const bool loader = wave8 >= 4;
  constexpr int LSPLIT = IG2_LOADERS / 4;               // loader waves sharing one consumer-wave's worth of DMA instructions
  const int wave = loader ? (wave8 - 4) / LSPLIT : wave8;
  const int lpart = loader ? (wave8 - 4) % LSPLIT : 0;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with a private 4 MiB L2); remap the
  // linear id so that one XCD gets a contiguous run of tiles (n fastest, then m): neighbouring image rows, whose 3x3
  // halos overlap, and both n-tiles of the same rows then share an L2.  Pure speed: any placement is correct.
  int tile_m, tile_n, tile_z;
  int pf_turn = 0, pf_mask = 0;          // weight-prefetch duty among the m-tiles of this XCD that share a panel
  {
    // grid extents: from the preloaded arguments when they are powers of two (gridDim.* is a load from the hidden arguments)
    const int nx = h.ny_shift() >= 0 ? (1 << h.nx_shift()) : (int)gridDim.x, ny = h.ny_shift() >= 0 ? (1 << h.ny_shift()) : (int)gridDim.y;
    const int nwg = h.nwg;
    const int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7, pos = lin >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    if (h.ny_shift() >= 0) {             // power-of-two grid (nearly every launch): no integer divisions in the prologue
      tile_n = swz & (ny - 1);
      tile_m = (swz >> h.ny_shift()) & (nx - 1);
      tile_z = swz >> (h.ny_shift() + h.nx_shift());
      if (IG2_PF_SHARE && r == 0 && (q >> h.ny_shift()) >= 2 && ((q >> h.ny_shift()) & ((q >> h.ny_shift()) - 1)) == 0 &&
          (q & (ny - 1)) == 0 && (q >> h.ny_shift()) <= nx) {
        pf_mask = (q >> h.ny_shift()) - 1;       // q tiles per XCD, n fastest: q / ny m-tiles of one z share a weight panel
        pf_turn = (pos >> h.ny_shift()) & pf_mask;
      }
    } else {
      tile_n = swz % ny;
      tile_m = (swz / ny) % nx;
      tile_z = swz / (ny * nx);
    }
  }
#ifdef ABL_SAMEX
  const int m0 = (tile_m & 1) * BM;      // every block reads the same two pixel tiles (L2-hot probe; wrong results)
#else
  const int m0 = tile_m * BM;
#endif
  const int n0 = tile_n * BN;
  const int batch = h.ksplit == 1 ? tile_z : tile_z / h.ksplit;
  const int ks_id = h.ksplit == 1 ? 0 : tile_z % h.ksplit;
  const half_t* X = h.X;
  const half_t* Wt = h.Wt;
  if (batch != 0) {                      // batch strides live in the argument block proper (not preloaded): first batch needs none
    X += (long long)batch * a.bsx;
    Wt += (long long)batch * a.bsw;
  }
  const int KS = h.K / BK;
  const int per = h.ksplit == 1 ? KS : (KS + h.ksplit - 1) / h.ksplit;
  int ks0 = ks_id * per;
  int ks1 = min(KS, ks0 + per);
  int nk_loop = ks1 - ks0;                       // barrier count: identical for every wave of the workgroup
  if (HALVES == 2) {
    const int half_n = (ks1 - ks0 + 1) / 2;      // team 0 takes the larger half
    nk_loop = half_n;
    ks0 += team * half_n;
    ks1 = min(ks1, ks0 + half_n);
  }
  const int nk = max(0, ks1 - ks0);
  const int HW = h.H * h.W;

  // ---- loader state: instruction i of this wave covers tile rows (wave*XI + i)*8 .. +7.  K order is tap-outer,
  //      channel-inner (k = tap*Cin + c, linear in the K-step).  Source pointers are incremental: they advance by BK
  //      halfs per K-step and are recomputed only when the 3x3 tap changes (measured: the chunk-outer order that
  //      re-reads an activation line on consecutive steps is not faster -- the DMA rate, not L2 locality, bounds it). ----
  const int steps_per_tap = CONV3 ? h.Cin / BK : KS;
  const int lrow = lane >> 3;            // row within the instruction
  const int pch = lane & 7;              // physical chunk this lane fills
  int xn[XI], xy[XI], xx[XI], xsc[XI];
  const half_t* xp[XI];
  int xstep[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = (wave * XI + i) * RPI + lrow;
    xsc[i] = (pch ^ ((row >> 1) & 7)) * 8;     // source chunk (halfs) for this physical slot
    const int m = m0 + row;
    if (CONV3) {
      if (h.hw_shift() >= 0) {                   // power-of-two maps (every map of the UNet): shifts, block-uniform branch
        xn[i] = m >> h.hw_shift();
        const int p = m & (HW - 1);
        xy[i] = p >> h.w_shift();
        xx[i] = p & (h.W - 1);
      } else {
        xn[i] = m / HW;
        const int p = m - xn[i] * HW;
        xy[i] = p / h.W;
        xx[i] = p - xy[i] * h.W;
      }
      xp[i] = g_zero_line;
      xstep[i] = 0;
    } else {
      xn[i] = m; xy[i] = 0; xx[i] = 0;
      xp[i] = X + (long long)m * h.ldx + (long long)ks0 * BK + xsc[i];
      xstep[i] = BK;
    }
  }
  const half_t* wp[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = (wave * WI + i) * RPI + lrow;
    wp[i] = Wt + (long long)(n0 + row) * h.ldw + (long long)ks0 * BK + (pch ^ ((row >> 1) & 7)) * 8;
  }
  int tap_left = 0;                      // K-steps left before the tap changes (conv)
  int next_ks = ks0;

  auto retap = [&]() {
    if (next_ks >= 9 * steps_per_tap) {          // second source: the folded 1x1 convolution (plain rows of X2)
      const int c0 = (next_ks - 9 * steps_per_tap) * BK;
      tap_left = KS - next_ks;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const long long m = (long long)xn[i] * HW + xy[i] * h.W + xx[i];
        xp[i] = a.X2 + m * a.ldx2 + c0 + xsc[i];
        xstep[i] = BK;
      }
      return;
    }
    const int tap = next_ks / steps_per_tap;
    const int c0 = (next_ks - tap * steps_per_tap) * BK;
    tap_left = steps_per_tap - (next_ks - tap * steps_per_tap);
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int yy = xy[i] + dy, xc = xx[i] + dx;
      const bool ok = yy >= 0 && yy < h.H && xc >= 0 && xc < h.W;
      const int yc = ok ? yy : 0, xq = ok ? xc : 0;
      const long long pix = h.ups() ? ((long long)xn[i] * (HW >> 2) + (yc >> 1) * (h.W >> 1) + (xq >> 1))
                                    : ((long long)xn[i] * HW + yc * h.W + xq);
      const half_t* src = X + pix * h.ldx + c0 + xsc[i];
      xp[i] = ok ? src : (const half_t*)g_zero_line;
      xstep[i] = ok ? BK : 0;
    }
  };

  auto issue = [&](int slot) {
    if (CONV3) {
      if (tap_left == 0) retap();
      --tap_left;
    }
    ++next_ks;
    half_t* sx = lds + slot * STAGE;
    half_t* sw = sx + BM * BK;
#ifndef ABL_NOLOAD
#pragma unroll
    for (int i = 0; i < XI; ++i) {
#ifndef ABL_NOX
      if (i % LSPLIT == lpart)
        __builtin_amdgcn_global_load_lds(xp[i], (lds_void*)(sx + (wave * XI + i) * RPI * BK), 16, 0, 0);
#endif
      xp[i] += xstep[i];
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
#ifndef ABL_NOW
      if (i % LSPLIT == lpart)
        __builtin_amdgcn_global_load_lds(wp[i], (lds_void*)(sw + (wave * WI + i) * RPI * BK), 16, 0, 0);
#endif
      wp[i] += BK;
    }
#endif
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  IG_STAMP(1, wave_all == 0);
  // ---- prologue: NST-1 stages in flight ----
  if (loader) {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < nk) issue(s);
  }

  if (loader) {
    for (int k = 0; k < nk_loop; ++k) {
      if (k < nk) {
        // stage k must have landed: stages up to min(nk, k+NST-1)-1 are issued, (XI+WI) instructions each
        if (k + NST - 1 <= nk) {
#if defined(ABL_NOX)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (WI)) : "memory");
#elif defined(ABL_NOW)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (XI)) : "memory");
#else
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (XI + WI) / LSPLIT) : "memory");
#endif
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (k + NST - 1 < nk) issue((k + NST - 1) % NST);
    }
  } else {
    // MFMA waves.  A K-step is two 32-deep halves; the fragments of the second half are read while the MFMAs of the
    // first run, and the first half of step k+1 is read (right after that step's barrier) under the MFMAs of the second
    // half of step k: the matrix pipe never waits for an LDS round trip (the straightforward loop -- read, wait, 16
    // MFMAs, read, wait ... -- spent 1170 cycles per step on 512 cycles of MFMA work, tools/experiments/fixed_cost_probe2.sh).
    // All fragment reads of step k have returned (lgkmcnt(0)) before this wave enters barrier k+1, after which the
    // loaders may overwrite that ring slot.
    // The reads are inline asm with hand-counted s_waitcnt lgkmcnt: hipcc's own counting falls back to lgkmcnt(0) for
    // reads that are in flight across the loop's back edge, which would expose a full LDS round trip per half step.
    unsigned xo[2][MT], wo[2][NT];               // fragment byte addresses within ring slot 0, per 32-deep half
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_void*)lds;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = (lane >> 4) + 4 * kk;
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int row = wm * TMW + j * 16 + (lane & 15);
        xo[kk][j] = lds_base + 2 * (row * BK + ((ch ^ ((row >> 1) & 7)) * 8));
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int row = wn * TNW + i * 16 + (lane & 15);
        wo[kk][i] = lds_base + 2 * (BM * BK + row * BK + ((ch ^ ((row >> 1) & 7)) * 8));
      }
    }
    half8 xa[MT], wa[NT], xb[MT], wb[NT];
    auto read_half = [&](int k, int kk, half8 (&xf)[MT], half8 (&wf)[NT]) {
      const unsigned slot = (unsigned)(k % NST) * (STAGE * 2);
#pragma unroll
      for (int j = 0; j < MT; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[j]) : "v"(xo[kk][j] + slot) : "memory");
#pragma unroll
      for (int i = 0; i < NT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(wf[i]) : "v"(wo[kk][i] + slot) : "memory");
    };
    // wait until at most PENDING LDS reads are outstanding; the fragments are tied to the statement so that no MFMA on
    // them can be scheduled above it
    auto wait_frags = [&](auto pending, half8 (&xf)[MT], half8 (&wf)[NT]) {
      static_assert(MT <= 4 && NT <= 4, "operand list");
      if constexpr (MT == 4 && NT == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3])
                     : "n"(decltype(pending)::value) : "memory");
      else
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(wf[0]), "+v"(wf[1]) : "n"(decltype(pending)::value) : "memory");
    };
    auto mfma_half = [&](half8 (&xf)[MT], half8 (&wf)[NT]) {
#ifndef ABL_NOMFMA
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
#else
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i][0][0] += (float)wf[i][0];
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[0][j][1] += (float)xf[j][0];
#endif
    };
    // Measured (tools/experiments/fixed_cost_probe2.sh, loads ablated): 0.44 us per 128x128x64 step = 1.23 PFLOP/s chip-wide, the
    // rate an MFMA-dense loop on random data sustains at the clock the chip holds under that load (MI355X_MICROARCH.md,
    // DVFS); spreading the reads one per MFMA gap instead of in a burst changed nothing.
    using std::integral_constant;
#if IG2_PF_DIST > 0
    __shared__ int pf_scratch[64];
    const bool pf_wave = wave8 < BN / 64;
    const half_t* pfp = Wt + (long long)(n0 + wave8 * 64 + lane) * h.ldw + (long long)ks0 * BK;
    auto pf_issue = [&](int step) {
      if (((ks0 + step) & pf_mask) == pf_turn)
        __builtin_amdgcn_global_load_lds(pfp + step * BK, (lds_void*)pf_scratch, 4, 0, 0);
    };
    if (pf_wave)
      for (int s2 = NST - 1; s2 < min(nk, IG2_PF_DIST); ++s2) pf_issue(s2);
#endif
    __builtin_amdgcn_s_barrier();                // step 0 has landed
    asm volatile("" ::: "memory");
    IG_STAMP(2, wave_all == 0);
    if (nk > 0) read_half(0, 0, xa, wa);
    for (int k = 0; k < nk_loop; ++k) {
      const bool act = k < nk;                   // the shorter half (odd step count) idles through the last barrier
#if IG2_PF_DIST > 0
      if (pf_wave && k + IG2_PF_DIST < nk) pf_issue(k + IG2_PF_DIST);
#endif
      if (act) {
        read_half(k, 1, xb, wb);
        wait_frags(integral_constant<int, MT + NT>{}, xa, wa);
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(xa, wa);
        __builtin_amdgcn_sched_barrier(0);
        wait_frags(integral_constant<int, 0>{}, xb, wb);
      }
      if (k + 1 < nk_loop) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (k + 1 < nk) read_half(k + 1, 0, xa, wa);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (act) mfma_half(xb, wb);
    }
  }

  IG_STAMP(3, wave_all == 0);
  // two teams on the staged path: both teams' partial tiles go straight into the epilogue, which adds them row by row
  const bool merge_in_epilogue = HALVES == 2 && igemm_epilogue_is_staged(a, batch);
  if (HALVES == 2 && !merge_in_epilogue) {
    // team 1 hands its accumulators to team 0 through (its own, now idle) ring memory
    f32x4* red = reinterpret_cast<f32x4*>(reinterpret_cast<half_t*>(smem_raw) + NST * STAGE);
    __syncthreads();                             // every fragment read of the K loop is done
    if (!loader && team == 1) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) red[((i * MT + j) * 4 + wave) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    if (!loader && team == 0) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] += red[((i * MT + j) * 4 + wave) * 64 + lane];
    }
  }
  IG_STAMP(4, wave_all == 0);
#ifdef ABL_NOEPI
  if (a.alpha == 12345.f)                        // harness probe: the launch without its epilogue (never true)
#endif
  igemm_epilogue<MT, NT, TMW, TNW, BN, (256 + 64 * IG2_LOADERS) * HALVES>(a, acc, m0, n0, wm, wn, lane, batch, ks_id,
                                                                        reinterpret_cast<float*>(smem_raw),
                                                                        !loader && (team == 0 || merge_in_epilogue),
                                                                        merge_in_epilogue ? team : -1);
#endif
}

template <int BM, int BN, int NST, bool CONV3, int HALVES = 1>
static int launch2(const IgemmArgs& a, hipStream_t s) {
  constexpr size_t smem = (size_t)HALVES * NST * (BM + BN) * 64 * sizeof(half_t);
  static_assert(smem <= 163840, "LDS");
  auto kern = igemm2_kernel<BM, BN, NST, CONV3, HALVES>;
  ISHAP_TRY(ishap_set_max_lds((const void*)kern, (int)smem));
  dim3 grid(a.M / BM, ceil_div(a.N, BN), a.nbatch * a.ksplit);
  IgemmArgs b = a;
  auto lg2 = [](int v) { int k = 0; while ((1 << k) < v) ++k; return (1 << k) == v ? k : -1; };
  b.w_shift = a.W > 0 ? lg2(a.W) : -1;
  b.hw_shift = (a.W > 0 && a.H > 0) ? lg2(a.H * a.W) : -1;
  if (b.w_shift < 0 || b.hw_shift < 0) b.w_shift = b.hw_shift = -1;
  b.nx_shift = lg2((int)grid.x);
  b.ny_shift = lg2((int)grid.y);
  if (b.nx_shift < 0 || b.ny_shift < 0) b.nx_shift = b.ny_shift = -1;
  IgemmHot h;
  h.X = b.X; h.Wt = b.Wt; h.K = b.K; h.Cin = b.Cin; h.ldx = b.ldx; h.ldw = b.ldw; h.H = b.H; h.W = b.W; h.ksplit = b.ksplit;
  h.nwg = (int)(grid.x * grid.y * grid.z);
  h.packed = (unsigned)(b.w_shift & 0x3f) | (unsigned)(b.hw_shift & 0x3f) << 6 | (unsigned)(b.nx_shift & 0x3f) << 12 |
             (unsigned)(b.ny_shift & 0x3f) << 18 | (b.ups ? 1u << 24 : 0u);
  if (g_igemm_prof_start) hipExtLaunchKernelGGL(kern, grid, dim3((256 + 64 * IG2_LOADERS) * HALVES), smem, s, g_igemm_prof_start, g_igemm_prof_stop, 0,
                                                (const void*)h.X, (const void*)h.Wt, h.K, h.Cin, h.ldx, h.ldw, h.H, h.W, h.ksplit, h.nwg, h.packed, b);
  else hipLaunchKernelGGL(kern, grid, dim3((256 + 64 * IG2_LOADERS) * HALVES), smem, s,
                          (const void*)h.X, (const void*)h.Wt, h.K, h.Cin, h.ldx, h.ldw, h.H, h.W, h.ksplit, h.nwg, h.packed, b);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// one workgroup per CU at most and a K slice long enough to split: the two-team variant of the 64x64 conv kernel
// (measured: +4..16 % there, a loss on short slices and 1x1)
bool igemm2_two_teams(const IgemmArgs& a, bool big) {
  static const int halves = [] { const char* e = getenv("ISHAP_HALVES"); return e ? atoi(e) : 2; }();
  if (big || halves != 2 || !a.conv3) return false;
  const long long tiles = (long long)(a.M / 64) * ((a.N + 63) / 64) * a.nbatch * a.ksplit;
  const int steps = (a.K / 64 + a.ksplit - 1) / a.ksplit;
  return tiles <= 256 && steps >= 16;
}

// main kernel only (the caller adds the split-K reduce); big = 128x128 tile, else 64x64
int igemm2_launch_main(const IgemmArgs& a, bool big, hipStream_t s) {
#ifndef IG2_BIG_NST
#define IG2_BIG_NST 4
#endif
  if (big) return a.conv3 ? launch2<128, 128, IG2_BIG_NST, true>(a, s) : launch2<128, 128, IG2_BIG_NST, false>(a, s);
#ifndef IG2_SMALL_NST
#define IG2_SMALL_NST 4
#endif
  if (igemm2_two_teams(a, big)) return launch2<64, 64, 4, true, 2>(a, s);
  return a.conv3 ? launch2<64, 64, IG2_SMALL_NST, true>(a, s) : launch2<64, 64, IG2_SMALL_NST, false>(a, s);
}
