// 3x3 convolution, fourth generation (round 4): the LDS-DMA ring of igemm2.hip with each activation tile staged ONCE for
// the three horizontal taps of a kernel row.
//
// Why: the K loop of the LDS-DMA kernels runs at the rate the CU's load path delivers bytes into LDS, ~36 B/clk per CU
// whatever the hit rates (tools/experiments/pf_probe.sh, profiles/round4_kstep_ablation.txt: without the activation loads a 64x64-tile
// launch loses 25-37 % of its time, without the weight loads 14-33 %; an L2 prefetch of the weight panel changes nothing) --
// so the lever is staged bytes per FLOP.  igemm2 stages a BM x 64 activation slab per K-step, i.e. nine times per
// (pixel, channel); here K runs (chunk, dy, dx) with dx innermost and the slab of (chunk, dy) is staged once and read at
// three pixel shifts: activation traffic / 3, operand traffic / 1.5.
// What igemm3.hip (rounds 1-3, the same idea, never faster than igemm2; removed in round 4) got wrong and this kernel does differently:
//   * no halo columns or padded pitch: the slab is exactly the BM pixels of the tile (BM / 8 DMA instructions, as in igemm2);
//     a lane whose shifted pixel falls off the image row reads a zero row kept at the end of every slab slot;
//   * every fragment address (3 shifts x 2 K halves) is precomputed once per lane; a K-step costs one v_add per ds_read
//     (igemm3 rebuilt row, swizzle and address per fragment per step: ~100 VALU instructions per step on the MFMA waves);
//   * hand-pipelined fragment reads across the barrier, loader / MFMA wave split, SGPR-preloaded arguments, XCD tile
//     remap and the staged all-waves epilogue are igemm2's.
// Swizzle of the activation slab: 16-byte chunk ^ (row & 7) -- conflict-free for ds_read_b128 fragments starting at ANY
// row under the lane grouping of MI355X_MICROARCH.md (LDS table), which the shifted reads need; the weight slab keeps
// igemm2's chunk ^ ((row >> 1) & 7).
// Tile shapes (igemm4_launch_main): 128x128 on maps 16 ... 128 wide (several image rows per tile below 128; fragment reads
// interleaved into the MFMA blocks), 64x64 one- and two-team on maps 16 ... 64 wide, 128 pixels x 64 channels on the 64-wide
// maps, and on the 8x8 maps 64x64 tiles (= one image) with K cut into ~16 slices whose fp32 partial tiles are left for the
// consuming GroupNorm kernel to add up (igemm4_small_map_slices; took the level over from a one-launch small-map kernel late in round 4, removed in round 6).
// Reference arithmetic: conv2d 3x3, padding 1 (gd/unet.py ResBlock in_layers / out_layers, :236-256) and its input
// gradient (flipped, transposed weights).  Same products as igemm2, another order of the K sum inside the fp32
// accumulators: the two kernels agree to summation order (tests/test_gpu_fullsize.py: <= 2e-3 relative over the full model).
#include "common.h"
#include "igemm_epilogue.h"
#include <type_traits>

__device__ __attribute__((aligned(128))) half_t g_zero_line4[64];   // zero-initialised: source of out-of-image rows
typedef __attribute__((address_space(3))) void lds_void4;

// loads a loader wave issues for K-steps a .. b of a slice's 3x3 part (WI weight instructions each, XI more where a slab starts)
constexpr int ig4_loads(int a, int b, int wi, int xi) {
  int c = 0;
  for (int u = a; u <= b; ++u) c += wi + ((u % 3 == 0) ? xi : 0);
  return c;
}

#ifndef IG4_PRO
#define IG4_PRO 2              // K-steps issued before the first wait; the ring then fills two steps per iteration
#endif

#ifdef ABL_NOLOAD             // harness probe: the K loop without its DMA (addresses still computed; wrong results)
#define IG4_DMA(src, dst) asm volatile("" ::"v"(src), "v"((unsigned)(unsigned long long)(lds_void4*)(dst)))
#else
#define IG4_DMA(src, dst) __builtin_amdgcn_global_load_lds((src), (lds_void4*)(dst), 16, 0, 0)
#endif

// s_waitcnt vmcnt takes an immediate; the loader waves know the count only at run time (block-uniform), hence the switch
__device__ __forceinline__ void ig4_wait_vm(int n) {
#define IG4_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    IG4_W(0) IG4_W(1) IG4_W(2) IG4_W(3) IG4_W(4) IG4_W(5) IG4_W(6) IG4_W(7) IG4_W(8) IG4_W(9) IG4_W(10) IG4_W(11) IG4_W(12)
    IG4_W(13) IG4_W(14) IG4_W(15) IG4_W(16) IG4_W(17) IG4_W(18) IG4_W(19) IG4_W(20) IG4_W(21) IG4_W(22) IG4_W(23) IG4_W(24)
    IG4_W(25) IG4_W(26) IG4_W(27) IG4_W(28) IG4_W(29) IG4_W(30) IG4_W(31) IG4_W(32) IG4_W(33) IG4_W(34) IG4_W(35) IG4_W(36)
    IG4_W(37) IG4_W(38) IG4_W(39) IG4_W(40)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;     // never more than 40 outstanding by construction; safe anyway
  }
#undef IG4_W
}

// HALVES = 2: two 8-wave teams with their own rings split the K range and meet in the epilogue (igemm2's two-team form: a
// CU that holds one workgroup gets two rings' worth of loads in flight and two MFMA waves per SIMD).
// Folded second source (IgemmArgs::X2 / K2, a ResBlock's 1x1 skip convolution riding on its conv2): after the 3x3 part
// the K loop goes on with K2 / 64 steps that each stage their own slab of X2 rows and read it unshifted.
template <int BM, int BN, int WD, int NSTW, int NSTX, int HALVES = 1>
__global__ __launch_bounds__(512 * HALVES) void igemm4_kernel(const void* hX, const void* hWt, int hK, int hCin, int hldx, int hldw, int hH, int hW,
                                                              int hksplit, int hnwg, unsigned hpacked, int hbase, IgemmArgs a) {
  const IgemmHot h{(const half_t*)hX, (const half_t*)hWt, hK, hCin, hldx, hldw, hH, hW, hksplit, hnwg, hpacked};
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BK = 64;
  constexpr int XI = BM / 8 / 4;                   // activation DMA instructions per loader wave per slab
  constexpr int WI = BN / 8 / 4;                   // weight DMA instructions per loader wave per K-step
  constexpr int WSLOT = BN * BK;                   // halfs
  constexpr int XSLOT = (BM + 8) * BK;             // BM pixel rows + the zero row (row BM) + padding to a 1 KiB multiple
  constexpr int RING = NSTW * WSLOT + NSTX * XSLOT;   // halfs per team
  constexpr int TMW = BM / 2, TNW = BN / 2, MT = TMW / 16, NT = TNW / 16;
  constexpr int DIST = NSTW - 1;                   // K-steps in flight in the steady state
  constexpr int PRO = DIST < IG4_PRO ? DIST : IG4_PRO;
  constexpr int RAMP = DIST - PRO;                 // iterations that issue two K-steps
  static_assert(RAMP <= 6, "ramp iterations are peeled by hand");
  static_assert(NSTX * 3 >= DIST + 3, "activation ring too shallow for the prefetch distance");
  static_assert(BM % WD == 0, "a tile is whole image rows (fragment addresses are per lane, so a 16-pixel MFMA sub-tile may span rows)");
  static_assert(DIST * (WI + XI) <= 40, "ig4_wait_vm covers up to 40 outstanding loads");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = HALVES == 2 ? wave_all >> 3 : 0;
  const int wave8 = wave_all & 7;
  half_t* const sW = reinterpret_cast<half_t*>(smem_raw) + team * RING;      // [NSTW][WSLOT]
  half_t* const sX = sW + NSTW * WSLOT;                                      // [NSTX][XSLOT]
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int wm = wave >> 1, wn = wave & 1;
  IG_STAMP(0, wave_all == 0);
  // zero rows (never written by the DMA): the source of a shifted pixel that falls off its image row
  if (tid < HALVES * NSTX * 8) {
    const int tm = tid / (NSTX * 8), k = tid - tm * (NSTX * 8);
    *reinterpret_cast<f32x4*>(reinterpret_cast<half_t*>(smem_raw) + tm * RING + NSTW * WSLOT + (k >> 3) * XSLOT + BM * BK + (k & 7) * 8) =
        (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int tile_m, tile_n, tile_z;
  {
    const int nx = h.ny_shift() >= 0 ? (1 << h.nx_shift()) : (int)gridDim.x, ny = h.ny_shift() >= 0 ? (1 << h.ny_shift()) : (int)gridDim.y;
    const int nwg = h.nwg;
    // hbase (round 5): a launch may cover only tiles [hbase, hbase + grid) of the layer -- the overlapped forward tail runs a
    // layer as several one-dimensional launches of at most ISHAP_TAIL_WGS tiles, so that it never holds the LDS of more compute
    // units than that and the backward chain on the caller's stream keeps the rest (launch4)
    // bit 30 of hbase: the tile order WITHIN an XCD's share.  0: n-tiles fastest (an XCD holds few m-tiles x all weight panels:
    // right when the activations dominate, the 64^2 / 128^2 maps); 1: m-tiles fastest, then K slices, n-tiles slowest (an XCD
    // holds ONE or two weight panels x every m-tile and slice: right on the 32^2 / 16^2 maps, where the weights are most of the
    // bytes and every XCD used to pull all of them: 29.4 MB fetched per 32^2 launch for 5.7 MB of operands, round 4 PMC pass)
    const bool n_outer = (hbase >> 30) & 1;
    const int lin = (hbase & 0x3fffffff) + blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7, pos = lin >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    if (n_outer) {                                   // launcher: only with power-of-two nx
      tile_m = swz & (nx - 1);
      const int rest = swz >> h.nx_shift();
      tile_n = rest / h.ksplit;
      tile_z = rest - tile_n * h.ksplit;
    } else if (h.ny_shift() >= 0) {
      tile_n = swz & (ny - 1);
      tile_m = (swz >> h.ny_shift()) & (nx - 1);
      tile_z = swz >> (h.ny_shift() + h.nx_shift());
    } else {
      tile_n = swz % ny;
      tile_m = (swz / ny) % nx;
      tile_z = swz / (ny * nx);
    }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ks_id = tile_z;                        // nbatch == 1 (launcher)
  const int H = h.H, HW = H * WD;
  const int NC = h.Cin / BK;                       // 64-channel chunks
  const int NC2 = (int)(h.packed >> 25);           // 64-channel chunks of the folded second source (launcher: K2 / 64)
  const int G = 3 * NC;                            // slab groups (chunk, dy), three K-steps (dx) each
  // this K slice: groups [g0, g1) of the 3x3 part and chunks [c20, c21) of the second source, both dealt out evenly
  const int per = (G + h.ksplit - 1) / h.ksplit, per2 = (NC2 + h.ksplit - 1) / h.ksplit;
  int g0 = min(G, ks_id * per), g1 = min(G, g0 + per);
  int c20 = min(NC2, ks_id * per2), c21 = min(NC2, c20 + per2);
  int nk_loop = (g1 - g0) * 3 + (c21 - c20);       // barrier count: identical for every wave of the workgroup
  if (HALVES == 2) {
    // team 0: the first ~half of the steps as whole groups; team 1: the other groups and the second source
    const int ga = min(g1 - g0, (nk_loop + 3) / 6);
    const int n0s = ga * 3, n1s = nk_loop - n0s;
    nk_loop = max(n0s, n1s);
    if (team == 0) { g1 = g0 + ga; c21 = c20; }
    else g0 += ga;
  }
  const int ns3 = (g1 - g0) * 3, ns = ns3 + (c21 - c20);
  const int n_img = h.hw_shift() >= 0 ? (m0 >> h.hw_shift()) : m0 / HW;
  const int y0 = (m0 - n_img * HW) / WD;           // first image row of this tile
  __syncthreads();                                 // the zero rows are in place

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  IG_STAMP(1, wave_all == 0);
  if (loader) {
    // ---- loader waves.  They pace the K loop (every step ends at a barrier they reach after their waits and issues), and a
    // wave retires an instruction every ~5 cycles at best: the per-step path is kept to the DMA instructions, one 64-bit add
    // per pointer and a few scalar instructions -- a first generic version of this loop (run-time load counts, addresses
    // rebuilt per slab) cost 10-25 % of the whole launch (profiles/round4_igemm4_probe_v3.txt, column noload). ----
    const int lrow = lane >> 3, pch = lane & 7;
    // slab source of the issue head (chunk hc, kernel row hdy): xp = this lane's 16 bytes of pixel (y + hdy - 1, x), advanced by
    // block-uniform deltas from slab to slab whether or not the row exists; a row outside the image reads zp (a zero line)
    int hc = g0 / 3, hdy = g0 - hc * 3, hdx = 0;   // issue head of the 3x3 part: chunk, kernel row, column
    const half_t* xp[XI];
    const half_t* zp[XI];
    int yr[XI], xc[XI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int row = (wave * XI + i) * 8 + lrow;  // pixel of the tile = LDS row of the slab
      yr[i] = y0 + row / WD;
      xc[i] = row % WD;
      const int sc = (pch ^ (row & 7)) * 8;
      zp[i] = (const half_t*)g_zero_line4 + sc;
      xp[i] = h.X + ((long long)n_img * HW + (long long)(yr[i] + hdy - 1) * WD + xc[i]) * h.ldx + hc * BK + sc;
    }
    const long long x_row = (long long)WD * h.ldx;          // one image row down
    const half_t* wp[WI];
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int row = (wave * WI + i) * 8 + lrow;
      wp[i] = h.Wt + (long long)(n0 + row) * h.ldw + (pch ^ ((row >> 1) & 7)) * 8 + (long long)(3 * hdy) * h.Cin + hc * BK;
    }
    int hws = 0, hxs = 0;                          // ring slots of the head
    int issued = 0, tot = 0;                       // K-steps / DMA instructions issued by this wave
    auto issue3 = [&]() __attribute__((always_inline)) {      // one K-step of the 3x3 part
      if (hdx == 0) {                              // a new slab: image rows y + hdy - 1 of chunk hc, staged once for dx = -1, 0, +1
        half_t* x = sX + hxs * XSLOT;
        const int dy = hdy - 1;
        if (h.ups()) {                             // nearest-x2 upsampled source (4 launches per forward): addresses from scratch
#pragma unroll
          for (int i = 0; i < XI; ++i) {
            const int yy = yr[i] + dy;
            const bool ok = yy >= 0 && yy < H;
            const int yc = ok ? yy : 0;
            const long long pix = (long long)n_img * (HW >> 2) + (yc >> 1) * (WD >> 1) + (xc[i] >> 1);
            const half_t* src = ok ? h.X + pix * h.ldx + hc * BK + (zp[i] - (const half_t*)g_zero_line4) : zp[i];
            IG4_DMA(src, x + (wave * XI + i) * 8 * BK);
          }
        } else {
          const long long dxp = hdy == 2 ? BK - 2 * x_row : x_row;      // to the next slab: a row down, or up two and a chunk on
#pragma unroll
          for (int i = 0; i < XI; ++i) {
            const half_t* src = (unsigned)(yr[i] + dy) < (unsigned)H ? xp[i] : zp[i];
            IG4_DMA(src, x + (wave * XI + i) * 8 * BK);
            xp[i] += dxp;
          }
        }
        hxs = hxs + 1 == NSTX ? 0 : hxs + 1;
        tot += XI;
      }
      half_t* w = sW + hws * WSLOT;
#pragma unroll
      for (int i = 0; i < WI; ++i) IG4_DMA(wp[i], w + (wave * WI + i) * 8 * BK);
      hws = hws + 1 == NSTW ? 0 : hws + 1;
      long long dw = h.Cin;                        // taps are consecutive Cin-wide column blocks of the weight row
      if (++hdx == 3) {
        hdx = 0;
        if (++hdy == 3) { hdy = 0; ++hc; dw = BK - 8LL * h.Cin; }
      }
#pragma unroll
      for (int i = 0; i < WI; ++i) wp[i] += dw;
      tot += WI;
      ++issued;
    };
    auto issue2 = [&]() __attribute__((always_inline)) {      // one K-step of the folded second source: its own slab of X2 rows, unshifted
      if (issued == ns3) {                         // entering it: re-aim the pointers (X2 rows; the last columns of the weight rows)
#pragma unroll
        for (int i = 0; i < XI; ++i) {
          const int row = (wave * XI + i) * 8 + lrow;
          xp[i] = a.X2 + (long long)(m0 + row) * a.ldx2 + c20 * BK + (pch ^ (row & 7)) * 8;
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
          const int row = (wave * WI + i) * 8 + lrow;
          wp[i] = h.Wt + (long long)(n0 + row) * h.ldw + (pch ^ ((row >> 1) & 7)) * 8 + 9LL * h.Cin + (long long)c20 * BK;
        }
      }
      half_t* x = sX + hxs * XSLOT;
#pragma unroll
      for (int i = 0; i < XI; ++i) { IG4_DMA(xp[i], x + (wave * XI + i) * 8 * BK); xp[i] += BK; }
      hxs = hxs + 1 == NSTX ? 0 : hxs + 1;
      half_t* w = sW + hws * WSLOT;
#pragma unroll
      for (int i = 0; i < WI; ++i) { IG4_DMA(wp[i], w + (wave * WI + i) * 8 * BK); wp[i] += BK; }
      hws = hws + 1 == NSTW ? 0 : hws + 1;
      tot += XI + WI;
      ++issued;
    };
    using std::integral_constant;
    // Everything that may be issued once the reads of K-steps < s are complete (s = -1: the prologue).  Step u needs its
    // weight slot (u <= s + DIST) and, when it starts a slab, a dead slab slot: always true in the 3x3 part (3 NSTX >= DIST + 3),
    // u <= s + NSTX - 1 for the one-step slabs of the second source.  The ring is NOT filled at once: a prologue of DIST
    // stages from every CU at the same moment (HBM-cold weights) comes back at the burst rate of the whole chip and stage 0
    // lands last-ish (9 800 cycles for 112 KB per CU against 4 400 for igemm2's 96 KB, in-kernel stamps) -- PRO steps first,
    // then two more per iteration until DIST are in flight.
    auto issue_after = [&](int s) __attribute__((always_inline)) {
      while (issued < ns && issued <= s + DIST && issued < PRO + 2 * (s + 1) && (issued < ns3 || issued <= s + NSTX - 1)) {
        if (issued < ns3) issue3();
        else issue2();
      }
    };
    int s = 0;
    if (ns3 >= 2 * DIST + 3) {
      // the common case, with compile-time load counts: PRO steps, RAMP peeled iterations that issue two steps each, then
      // the steady loop (one step per iteration, DIST in flight) for as long as the step it issues is a 3x3 step
#pragma unroll
      for (int u = 0; u < PRO; ++u) issue3();
      auto ramp_iter = [&](auto S) __attribute__((always_inline)) {
        constexpr int s0 = decltype(S)::value;
        constexpr int hd = PRO + 2 * s0;           // K-steps issued before barrier s0
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ig4_loads(s0 + 1, hd - 1, WI, XI)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr int upto = s0 + 1 + DIST < PRO + 2 * (s0 + 1) ? s0 + 1 + DIST : PRO + 2 * (s0 + 1);
#pragma unroll
        for (int u = hd; u < upto; ++u) issue3();
      };
      if constexpr (RAMP > 0) ramp_iter(integral_constant<int, 0>{});
      if constexpr (RAMP > 1) ramp_iter(integral_constant<int, 1>{});
      if constexpr (RAMP > 2) ramp_iter(integral_constant<int, 2>{});
      if constexpr (RAMP > 3) ramp_iter(integral_constant<int, 3>{});
      if constexpr (RAMP > 4) ramp_iter(integral_constant<int, 4>{});
      if constexpr (RAMP > 5) ramp_iter(integral_constant<int, 5>{});
      int ph = RAMP % 3;
      for (s = RAMP; s + DIST < ns3; ++s) {
        // K-step s has landed when at most the loads issued after its weights are outstanding: steps s+1 .. s+DIST-1
        if (ph == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ig4_loads(1, DIST - 1, WI, XI)) : "memory");          // s = 0 (mod 3)
        else if (ph == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ig4_loads(2, DIST, WI, XI)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ig4_loads(3, DIST + 1, WI, XI)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue3();
#ifdef ABL_XFORM
        // HARNESS TIMING PROBE (tools/bench_igemm.hip -DABL_XFORM; wrong results): what a GroupNorm apply + SiLU
        // of the NEXT activation slab, done in LDS by the loader waves, costs the K loop.  The slab of steps 3(j+1) .. 3(j+1)+2 is
        // rewritten in two halves, in the iterations of steps 3j+1 and 3j+2 (it has landed once at most the weights of the three
        // steps issued after it are outstanding); the arithmetic is gn_act.h's sequence with per-lane stand-in parameters
        if (ph != 0) {
          if (ph == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * WI) : "memory");
          const int nslot = (s / 3 + 1) % NSTX;
          half_t* xs = sX + nslot * XSLOT;
          const float x_mu = 0.01f * (float)(lane & 7), x_rs = 1.05f, x_g = 0.9f + 1e-3f * (float)lane, x_b = 0.02f;
#pragma unroll
          for (int i2 = 0; i2 < (XI + 1) / 2; ++i2) {
            const int i = (ph - 1) * ((XI + 1) / 2) + i2;
            if (i < XI) {
              half8* p = reinterpret_cast<half8*>(xs + (wave * XI + i) * 8 * BK + lane * 8);
              half8 v = *p;
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                const half_t yh = gn_affine((float)v[c], x_mu, x_rs, x_g, x_b);
#ifdef ABL_XFORM_FILM
                const half_t yf = gn_film(yh, (half_t)1.01f, (half_t)0.01f);
                v[c] = (half_t)gn_silu((float)yf);
#else
                v[c] = (half_t)gn_silu((float)yh);
#endif
              }
              *p = v;
            }
          }
        }
#endif
        ph = ph == 2 ? 0 : ph + 1;
      }
    } else {
      issue_after(-1);
    }
    // the rest -- the tail of the 3x3 part, the second source, a slice too short for the loop above, the idle barriers of the
    // shorter team -- with run-time counts: K-step s has landed when only the loads issued after its weights are outstanding
    // (loads complete in order), i.e. everything issued so far minus everything up to and including step s
    for (; s < nk_loop; ++s) {
      if (s < ns) {
        const int slabs = s < ns3 ? s / 3 + 1 : ns3 / 3 + (s - ns3 + 1);
        ig4_wait_vm(tot - ((s + 1) * WI + slabs * XI));
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_after(s);
    }
  } else {
    // ---- MFMA waves: fragment byte addresses within slot 0 of each ring, per 32-deep half; activations per shift ----
    unsigned wo[2][NT], xo[3][2][MT];
    const unsigned wbase = (unsigned)(unsigned long long)(lds_void4*)sW, xbase = (unsigned)(unsigned long long)(lds_void4*)sX;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int ch = (lane >> 4) + 4 * kk;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int row = wn * TNW + i * 16 + (lane & 15);
        wo[kk][i] = wbase + 2 * (row * BK + ((ch ^ ((row >> 1) & 7)) * 8));
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int pl = wm * TMW + j * 16 + (lane & 15);
        const int x = pl % WD;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const int xs = x + d - 1;
          const int row = (xs >= 0 && xs < WD) ? pl + d - 1 : BM;       // BM = the zero row
          xo[d][kk][j] = xbase + 2 * (row * BK + (row == BM ? ch * 8 : ((ch ^ (row & 7)) * 8)));
        }
      }
    }
    half8 xa[MT], wa[NT], xb[MT], wb[NT];
    unsigned sw_off = 0, sx_off = 0;               // byte offsets of the current W / X ring slots
    auto read_frags = [&](const unsigned (&xs)[2][MT], int kk, unsigned swo, unsigned sxo, half8 (&xf)[MT], half8 (&wf)[NT]) {
#pragma unroll
      for (int j = 0; j < MT; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[j]) : "v"(xs[kk][j] + sxo) : "memory");
#pragma unroll
      for (int i = 0; i < NT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(wf[i]) : "v"(wo[kk][i] + swo) : "memory");
    };
    auto read_half = [&](auto DX, int kk, unsigned swo, unsigned sxo, half8 (&xf)[MT], half8 (&wf)[NT]) {
      read_frags(xo[decltype(DX)::value], kk, swo, sxo, xf, wf);
    };
    auto wait_frags = [&](auto pending, half8 (&xf)[MT], half8 (&wf)[NT]) {
      static_assert((MT == 2 || MT == 4) && (NT == 2 || NT == 4), "operand list");
      if constexpr (MT == 4 && NT == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3])
                     : "n"(decltype(pending)::value) : "memory");
      else if constexpr (MT == 4 && NT == 2)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(wf[0]), "+v"(wf[1])
                     : "n"(decltype(pending)::value) : "memory");
      else if constexpr (MT == 2 && NT == 4)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(xf[0]), "+v"(xf[1]), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3])
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
    // MFMAs of one 32-deep half with the fragment reads of the NEXT half in their gaps (one read behind each of the first MT + NT
    // MFMAs, pinned by scheduling barriers): issued in a burst the eight reads and their address adds kept the MFMA pipe idle for
    // ~100 cycles twice per K-step (the wave issues in order, the reads sat behind the sixteenth MFMA)
    auto read_one = [&](const unsigned (&xs)[2][MT], int kk, unsigned swo, unsigned sxo, half8 (&xf)[MT], half8 (&wf)[NT], int r) __attribute__((always_inline)) {
      if (r < MT) asm volatile("ds_read_b128 %0, %1" : "=v"(xf[r]) : "v"(xs[kk][r] + sxo) : "memory");
      else asm volatile("ds_read_b128 %0, %1" : "=v"(wf[r - MT]) : "v"(wo[kk][r - MT] + swo) : "memory");
    };
    auto mfma_reads = [&](half8 (&xf)[MT], half8 (&wf)[NT], const unsigned (&xs)[2][MT], int kk, unsigned swo, unsigned sxo, half8 (&xn)[MT],
                          half8 (&wn)[NT]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
          const int r = i * MT + j;
          if (r < MT + NT) {
            read_one(xs, kk, swo, sxo, xn, wn, r);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    };
    using std::integral_constant;
    // one K-step at shift DX (= dx + 1): second half read under the MFMAs of the first, first half of the next step read
    // right after its barrier.  SLAB_ENDS: the step is the last one on its slab (dx = +1, or any step of the second
    // source).  The next step reads at shift (DX + 1) % 3 inside the 3x3 part and unshifted (1) in the second source.
    // 128x128 tiles (MFMA-paced K loop): the interleaved form, -1.5 / -2.2 % per launch (128^2: 256->256 26.2 -> 25.8 us, 512->256
    // 43.2 -> 42.2; profiles/round4_igemm4_interleave_probe.txt); on the 64x64 tiles (paced by the staging) it is 1-3 % slower in situ
    constexpr bool INTERLEAVE = BM == 128 && BN == 128;
    auto step_il = [&](auto DX, auto SLAB_ENDS, int s) {
      constexpr int d = decltype(DX)::value;
      wait_frags(integral_constant<int, 0>{}, xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      mfma_reads(xa, wa, xo[d], 1, sw_off, sx_off, xb, wb);
      __builtin_amdgcn_sched_barrier(0);
      wait_frags(integral_constant<int, 0>{}, xb, wb);
      sw_off = sw_off + WSLOT * 2 == NSTW * WSLOT * 2 ? 0u : sw_off + WSLOT * 2;
      if (decltype(SLAB_ENDS)::value) sx_off = sx_off + XSLOT * 2 == NSTX * XSLOT * 2 ? 0u : sx_off + XSLOT * 2;
      if (s + 1 < nk_loop) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 1 < ns) {
          __builtin_amdgcn_sched_barrier(0);
          if (decltype(SLAB_ENDS)::value && s + 1 >= ns3) mfma_reads(xb, wb, xo[1], 0, sw_off, sx_off, xa, wa);
          else mfma_reads(xb, wb, xo[(d + 1) % 3], 0, sw_off, sx_off, xa, wa);
          __builtin_amdgcn_sched_barrier(0);
          return;
        }
      }
      mfma_half(xb, wb);
    };
    auto step_burst = [&](auto DX, auto SLAB_ENDS, int s) {
      constexpr int d = decltype(DX)::value;
      read_half(DX, 1, sw_off, sx_off, xb, wb);
      wait_frags(integral_constant<int, MT + NT>{}, xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      wait_frags(integral_constant<int, 0>{}, xb, wb);
      sw_off = sw_off + WSLOT * 2 == NSTW * WSLOT * 2 ? 0u : sw_off + WSLOT * 2;
      if (decltype(SLAB_ENDS)::value) sx_off = sx_off + XSLOT * 2 == NSTX * XSLOT * 2 ? 0u : sx_off + XSLOT * 2;
      if (s + 1 < nk_loop) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 1 < ns) {
          if (decltype(SLAB_ENDS)::value && s + 1 >= ns3) read_half(integral_constant<int, 1>{}, 0, sw_off, sx_off, xa, wa);
          else read_half(integral_constant<int, (d + 1) % 3>{}, 0, sw_off, sx_off, xa, wa);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      mfma_half(xb, wb);
    };
    auto step = [&](auto DX, auto SLAB_ENDS, int s) __attribute__((always_inline)) {
      if constexpr (INTERLEAVE) step_il(DX, SLAB_ENDS, s);
      else step_burst(DX, SLAB_ENDS, s);
    };
    if (nk_loop > 0) __builtin_amdgcn_s_barrier();      // step 0 has landed (every wave passes nk_loop barriers)
    asm volatile("" ::: "memory");
    IG_STAMP(2, wave_all == 0);
    if (ns > 0) {
      if (ns3 > 0) read_half(integral_constant<int, 0>{}, 0, sw_off, sx_off, xa, wa);
      else read_half(integral_constant<int, 1>{}, 0, sw_off, sx_off, xa, wa);
    }
    int s = 0;
    for (; s < ns3; s += 3) {
      step(integral_constant<int, 0>{}, std::false_type{}, s);
      step(integral_constant<int, 1>{}, std::false_type{}, s + 1);
      step(integral_constant<int, 2>{}, std::true_type{}, s + 2);
    }
    for (; s < ns; ++s) step(integral_constant<int, 1>{}, std::true_type{}, s);
    for (; s < nk_loop; ++s)                       // the shorter team of a two-team workgroup idles through the last barriers
      if (s + 1 < nk_loop) __builtin_amdgcn_s_barrier();
  }
  IG_STAMP(3, wave_all == 0);
  // two teams on the staged path: both teams' partial tiles go straight into the epilogue, which adds them row by row
  const bool merge_in_epilogue = HALVES == 2 && igemm_epilogue_is_staged(a, 0);
  if (HALVES == 2 && !merge_in_epilogue) {
    // team 1 hands its accumulators to team 0 through (its own, now idle) ring memory
    f32x4* red = reinterpret_cast<f32x4*>(reinterpret_cast<half_t*>(smem_raw) + RING);
    __syncthreads();                               // every fragment read of the K loop is done
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
  igemm_epilogue<MT, NT, TMW, TNW, BN, 512 * HALVES>(a, acc, m0, n0, wm, wn, lane, 0, ks_id, reinterpret_cast<float*>(smem_raw),
                                                     !loader && (team == 0 || merge_in_epilogue), merge_in_epilogue ? team : -1);
#endif
}

template <int BM, int BN, int WD, int NSTW, int NSTX, int HALVES = 1>
static int launch4(const IgemmArgs& a, hipStream_t s) {
  constexpr size_t ring = (size_t)HALVES * (NSTW * BN * 64 + NSTX * (BM + 8) * 64) * sizeof(half_t);
  constexpr size_t epi = (size_t)HALVES * BM * (BN + 4) * 4 + (size_t)BM * (BN + 8) * 2 + 16384;      // staged epilogue: fp32 tile(s) + fp16 tile + partial sums
  static_assert((ring > epi ? ring : epi) <= 163840, "LDS");
  const size_t smem = ring > epi ? ring : epi;
  auto kern = igemm4_kernel<BM, BN, WD, NSTW, NSTX, HALVES>;
  ISHAP_TRY(ishap_set_max_lds((const void*)kern, (int)smem));
  const dim3 grid(a.M / BM, ceil_div(a.N, BN), a.ksplit);
  IgemmArgs b = a;
  auto lg2 = [](int v) { int k = 0; while ((1 << k) < v) ++k; return (1 << k) == v ? k : -1; };
  b.w_shift = lg2(a.W);
  b.hw_shift = lg2(a.H * a.W);
  if (b.w_shift < 0 || b.hw_shift < 0) b.w_shift = b.hw_shift = -1;
  b.nx_shift = lg2((int)grid.x);
  b.ny_shift = lg2((int)grid.y);
  if (b.nx_shift < 0 || b.ny_shift < 0) b.nx_shift = b.ny_shift = -1;
  IgemmHot h;
  h.X = b.X; h.Wt = b.Wt; h.K = b.K; h.Cin = b.Cin; h.ldx = b.ldx; h.ldw = b.ldw; h.H = b.H; h.W = b.W; h.ksplit = b.ksplit;
  h.nwg = (int)(grid.x * grid.y * grid.z);
  h.packed = (unsigned)(b.w_shift & 0x3f) | (unsigned)(b.hw_shift & 0x3f) << 6 | (unsigned)(b.nx_shift & 0x3f) << 12 |
             (unsigned)(b.ny_shift & 0x3f) << 18 | (b.ups ? 1u << 24 : 0u) | (unsigned)(b.K2 / 64) << 25;
  // chunked form (IgemmArgs::chunk_tiles > 0, a multiple of 8 so that a tile keeps the `lin & 7` = XCD of the remap): the layer as
  // ceil(tiles / chunk) launches of at most `chunk` tiles each, back to back on the stream
  // tile order within an XCD's share (kernel: bit 30 of hbase).  Bytes an XCD's L2 has to pull under each order, from the tiles it
  // is dealt (q = tiles / 8 consecutive ids): a weight panel per distinct (n-tile, K slice), an activation slab per distinct
  // (m-tile, K slice); neighbouring m-tiles share their halo rows inside one L2.  ISHAP_IG4_NOUTER=0 / 1 forces an order.
  int n_outer = 0;
  if (b.nx_shift >= 0) {
    static const int force = [] { const char* e = getenv("ISHAP_IG4_NOUTER"); return e ? atoi(e) : 2; }();
    const int nx = (int)grid.x, ny = (int)grid.y, nz = (int)grid.z;
    const int q = (h.nwg + 7) / 8;
    const double wp = (double)BN * (b.K / (double)nz) * 2.0;                 // one weight panel of one slice
    const double xs = (double)BM * b.Cin * 2.0 / nz + (b.K2 ? (double)BM * b.K2 * 2.0 / nz : 0.0);   // one tile's pixels, one slice's channels
    auto cdiv = [](int a, int d) { return (a + d - 1) / d; };
    // order 0: n fastest, then m, then z
    const int nA = ny < q ? ny : q, mzA = cdiv(q, ny), mA = nx < mzA ? nx : mzA, zA = cdiv(mzA, nx);
    const double costA = (double)nA * zA * wp + (double)mA * zA * xs;
    // order 1: m fastest, then z, then n
    const int mB = nx < q ? nx : q, znB = cdiv(q, nx), zB = nz < znB ? nz : znB;
    const double costB = (double)znB * wp + (double)mB * zB * xs;
    n_outer = force == 2 ? (costB < 0.8 * costA) : force;
  }
  const int order_bit = n_outer ? (1 << 30) : 0;
  const int tiles = h.nwg;
  const int chunk = a.chunk_tiles;
  const bool chunked = chunk > 0 && chunk % 8 == 0 && tiles > chunk && b.nx_shift >= 0 && !g_igemm_prof_start;
  if (chunked) {
    for (int base = 0; base < tiles; base += chunk) {
      const int n = tiles - base < chunk ? tiles - base : chunk;
      hipLaunchKernelGGL(kern, dim3(n), dim3(512 * HALVES), smem, s,
                         (const void*)h.X, (const void*)h.Wt, h.K, h.Cin, h.ldx, h.ldw, h.H, h.W, h.ksplit, h.nwg, h.packed, base | order_bit, b);
    }
  } else if (g_igemm_prof_start) hipExtLaunchKernelGGL(kern, grid, dim3(512 * HALVES), smem, s, g_igemm_prof_start, g_igemm_prof_stop, 0,
                                                (const void*)h.X, (const void*)h.Wt, h.K, h.Cin, h.ldx, h.ldw, h.H, h.W, h.ksplit, h.nwg, h.packed, order_bit, b);
  else hipLaunchKernelGGL(kern, grid, dim3(512 * HALVES), smem, s,
                          (const void*)h.X, (const void*)h.Wt, h.K, h.Cin, h.ldx, h.ldw, h.H, h.W, h.ksplit, h.nwg, h.packed, order_bit, b);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// the shapes this kernel takes: 3x3 (optionally with the folded 1x1 second source), Cin % 64 == 0, one image per tile, a tile =
// whole image rows of a map 16 / 32 / 64 / 128 pixels wide
bool igemm4_applicable(const IgemmArgs& a, bool big) {
  const int BM = big ? 128 : 64;
  if (!a.conv3 || a.nbatch != 1 || a.Cin % 64 != 0 || a.K != 9 * a.Cin + a.K2) return false;
  if (a.K2 && (a.K2 % 64 != 0 || a.K2 / 64 > 127 || !a.X2)) return false;
  // the folded second source runs with a short lookahead (one slab per step, NSTX - 1 ahead): worth it on the 128-tiles
  // (-11 %), a loss on the 64-tiles (+4..18 %, profiles/round4_igemm4_probe_v4.txt) -- those stay with igemm2
  // (the sliced launches on the 8x8 maps excepted: 9.5 us against 23.9 for the one-launch kernel they replaced, round 4)
  const bool w8 = !big && a.W == 8 && a.H == 8 && a.ksplit > 1;
  if (a.K2 && !big && !w8) return false;
  // (128-pixel tiles on maps narrower than 128: several image rows per tile -- the batched generate path, where M = batch * H * W
  // fills the chip with 128x128 tiles on the 64^2 ... 16^2 maps)
  if (big ? (a.W != 128 && a.W != 64 && a.W != 32 && a.W != 16) : (a.W != 16 && a.W != 32 && a.W != 64 && !w8)) return false;
  if (BM % a.W != 0 || (a.H * a.W) % BM != 0 || a.M % BM != 0) return false;
  return true;
}

// The 8x8 maps (one tile = one image): K slices for a sliced launch whose consumer adds the slices up, 0 = not taken (the shape
// then goes through the generic tile + split policy).  Enough slices for ~one workgroup per CU: the harness has 16 x 16 workgroups at
// 8.1 us against 9.1 for the one-launch small-map kernel this replaced (1024 -> 1024), 11.3 against 15.1 (2048 -> 1024), 11.1
// against 16.4 (1024 -> 2048), 9.5 against 23.9 with the folded skip (profiles/round4_igemm4_w8_probe.txt, _w8_k2_probe.txt)
int igemm4_small_map_slices(const IgemmArgs& a) {
  static const int on = [] { const char* all = getenv("ISHAP_IGEMM4"); return (all ? atoi(all) : 2) > 1; }();      // igemm.hip: 2 (default) = every shape igemm4 takes
  constexpr int target = 256;        // workgroups (in-situ sweep 256 .. 640: flat)
  if (!on || !a.conv3 || a.W != 8 || a.H != 8 || a.nbatch != 1 || a.Cin % 64 != 0 || a.M % 64 != 0 || a.K != 9 * a.Cin + a.K2) return 0;
  if (a.K2 && (a.K2 % 64 != 0 || a.K2 / 64 > 127 || !a.X2)) return 0;
  const int tiles = (a.M / 64) * ((a.N + 63) / 64), G = 3 * (a.Cin / 64);
  int ks = (target + tiles / 2) / tiles;
  if (ks > 16) ks = 16;
  if (ks > G) ks = G;
  if (ks < 2) return 0;
  const int per = (G + ks - 1) / ks;
  return (G + per - 1) / per;                      // no slice without 3x3 groups (the second source's chunks are dealt out likewise)
}

#ifndef IG4_BIG_W
#define IG4_BIG_W 5
#define IG4_BIG_X 3
#endif
#ifndef IG4_SMALL_W
#define IG4_SMALL_W 6
#define IG4_SMALL_X 3
#endif
#ifndef IG4_W8_W               // the 8x8 maps' sliced launches: 9-18 K-steps per workgroup -- a 4-slot weight ring (3 steps in flight) lets
#define IG4_W8_W 4             // a 9-step slice take the loader's compile-time path and beats the 6-slot one by 4-6 % per launch
#define IG4_W8_X 3             // (profiles/round4_igemm4_w8_ring_probe.txt)
#endif
#ifndef IG4_TEAM_W
#define IG4_TEAM_W 6
#define IG4_TEAM_X 3
#endif
// two teams: one workgroup per CU at most (<= 256 tiles) and a K slice long enough to halve (measured break-even: ~40 steps)
bool igemm4_two_teams(const IgemmArgs& a, bool big) {
  static const int on = [] { const char* e = getenv("ISHAP_IG4_TEAMS"); return e ? atoi(e) : 2; }();
  if (big || on != 2) return false;
  const long long tiles = (long long)(a.M / 64) * ((a.N + 63) / 64) * a.ksplit;
  const int groups = (3 * (a.Cin / 64) + a.ksplit - 1) / a.ksplit;
  return tiles <= 256 && 3 * groups + a.K2 / 64 >= 48;
}
// 128-pixel x 64-channel tiles (two image rows of a 64-wide map): for the 64^2 layers with >= 512 output channels the grid still
// fills the chip (32 x 8 = 256 workgroups) and a K-step stages 13.3 KB for twice the FLOPs of a 64x64 tile's 10.7 KB; with only 128
// such tiles (the 64^2 256->256 layers) it loses (0.1804 -> 0.182 s/shape): at least 224 tiles
bool igemm4_tall_tiles(const IgemmArgs& a, bool big) {
  static const int on = [] { const char* e = getenv("ISHAP_IG4_TEAMS"); return e ? atoi(e) : 2; }();      // 0: plain one-team 64x64 tiles everywhere
  if (big || !on || a.W != 64 || a.K2 != 0 || a.ksplit != 1 || a.M % 128 != 0 || (a.H * a.W) % 128 != 0) return false;
  const long long tiles = (long long)(a.M / 128) * ((a.N + 63) / 64);
  return tiles >= 224 && tiles <= 512;
}
int igemm4_launch_main(const IgemmArgs& a, bool big, hipStream_t s) {
  if (big) {
    if (a.W == 128) return launch4<128, 128, 128, IG4_BIG_W, IG4_BIG_X>(a, s);
    if (a.W == 64) return launch4<128, 128, 64, IG4_BIG_W, IG4_BIG_X>(a, s);
    if (a.W == 32) return launch4<128, 128, 32, IG4_BIG_W, IG4_BIG_X>(a, s);
    return launch4<128, 128, 16, IG4_BIG_W, IG4_BIG_X>(a, s);
  }
  if (igemm4_tall_tiles(a, big)) return launch4<128, 64, 64, IG4_SMALL_W, IG4_SMALL_X>(a, s);
#ifdef IG4_TALL_PROBE      // harness only (tools/experiments/tall_probe.sh; profiles/round6_tall_tiles_sliced_launches.txt): 128 x 64 tiles on the 32^2 / 16^2 maps
  if ((a.W == 32 || a.W == 16) && a.K2 == 0 && a.M % 128 == 0 && (a.H * a.W) % 128 == 0) {
    const int st = 3 * ((3 * (a.Cin / 64) + a.ksplit - 1) / a.ksplit);
    if (st >= 9 && st <= 12) return a.W == 32 ? launch4<128, 64, 32, IG4_W8_W, IG4_W8_X>(a, s) : launch4<128, 64, 16, IG4_W8_W, IG4_W8_X>(a, s);
    return a.W == 32 ? launch4<128, 64, 32, IG4_SMALL_W, IG4_SMALL_X>(a, s) : launch4<128, 64, 16, IG4_SMALL_W, IG4_SMALL_X>(a, s);
  }
#endif
  if (igemm4_two_teams(a, big)) {
    if (a.W == 64) return launch4<64, 64, 64, IG4_TEAM_W, IG4_TEAM_X, 2>(a, s);
    if (a.W == 32) return launch4<64, 64, 32, IG4_TEAM_W, IG4_TEAM_X, 2>(a, s);
    if (a.W == 8) return launch4<64, 64, 8, IG4_W8_W, IG4_W8_X>(a, s);
    return launch4<64, 64, 16, IG4_TEAM_W, IG4_TEAM_X, 2>(a, s);
  }
  if (a.W == 8) return launch4<64, 64, 8, IG4_W8_W, IG4_W8_X>(a, s);
  // slices of 9-12 K-steps: too short for the 6-slot ring's compile-time loader path (13 steps), long enough for the 4-slot
  // ring's (9) -- -13 % per launch there (16^2 512->512 in 8 slices 7.7 -> 6.7 us, 32^2 256->512 in 4 slices 9.9 -> 8.6;
  // profiles/round4_igemm4_ring_by_slice_length_probe.txt); longer slices keep the deeper ring (+5 % at 36 steps with 4 slots)
  const int steps = 3 * ((3 * (a.Cin / 64) + a.ksplit - 1) / a.ksplit);
  if (a.K2 == 0 && steps >= 9 && steps <= 12) {
    if (a.W == 64) return launch4<64, 64, 64, IG4_W8_W, IG4_W8_X>(a, s);
    if (a.W == 32) return launch4<64, 64, 32, IG4_W8_W, IG4_W8_X>(a, s);
    return launch4<64, 64, 16, IG4_W8_W, IG4_W8_X>(a, s);
  }
  if (a.W == 64) return launch4<64, 64, 64, IG4_SMALL_W, IG4_SMALL_X>(a, s);
  if (a.W == 32) return launch4<64, 64, 32, IG4_SMALL_W, IG4_SMALL_X>(a, s);
  return launch4<64, 64, 16, IG4_SMALL_W, IG4_SMALL_X>(a, s);
}
