// 3x3 convolution, fourth generation (round 4): the LDS-DMA ring of igemm2.hip with each activation tile staged ONCE for
// the three horizontal taps of a kernel row.
//
// Why: the K loop of the LDS-DMA kernels runs at the rate the CU's load path delivers bytes into LDS, ~36 B/clk per CU
// whatever the hit rates (tools/pf_probe.sh, profiles/round4_kstep_ablation.txt: without the activation loads a 64x64-tile
// launch loses 25-37 % of its time, without the weight loads 14-33 %; an L2 prefetch of the weight panel changes nothing) --
// so the lever is staged bytes per FLOP.  igemm2 stages a BM x 64 activation slab per K-step, i.e. nine times per
// (pixel, channel); here K runs (chunk, dy, dx) with dx innermost and the slab of (chunk, dy) is staged once and read at
// three pixel shifts: activation traffic / 3, operand traffic / 1.5.
// What igemm3.hip (round 1-3, the same idea) got wrong and this kernel does differently:
//   * no halo columns or padded pitch: the slab is exactly the BM pixels of the tile (BM / 8 DMA instructions, as in igemm2);
//     a lane whose shifted pixel falls off the image row reads a zero row kept at the end of every slab slot;
//   * every fragment address (3 shifts x 2 K halves) is precomputed once per lane; a K-step costs one v_add per ds_read
//     (igemm3 rebuilt row, swizzle and address per fragment per step: ~100 VALU instructions per step on the MFMA waves);
//   * hand-pipelined fragment reads across the barrier, loader / MFMA wave split, SGPR-preloaded arguments, XCD tile
//     remap and the staged all-waves epilogue are igemm2's.
// Swizzle of the activation slab: 16-byte chunk ^ (row & 7) -- conflict-free for ds_read_b128 fragments starting at ANY
// row under the lane grouping of MI355X_MICROARCH.md (LDS table), which the shifted reads need; the weight slab keeps
// igemm2's chunk ^ ((row >> 1) & 7).
// Reference arithmetic: conv2d 3x3, padding 1 (gd/unet.py ResBlock in_layers / out_layers, :236-256) and its input
// gradient (flipped, transposed weights); results are bit-identical to igemm2's (same products, same fp32 MFMA order of K
// within a 64-channel chunk is NOT the same -- the K order differs, so values agree to fp32 summation order only).
#include "common.h"
#include "igemm_epilogue.h"
#include <type_traits>

__device__ __attribute__((aligned(128))) half_t g_zero_line4[64];   // zero-initialised: source of out-of-image rows
typedef __attribute__((address_space(3))) void lds_void4;

// activation slabs whose first K-step u lies in [s + 1, s + dist - 1], for s = ph (mod 3)
constexpr int ig4_slabs_ahead(int ph, int dist) {
  int c = 0;
  for (int j = 1; j < dist; ++j) c += ((ph + j) % 3 == 0) ? 1 : 0;
  return c;
}

template <int BM, int BN, int WD, int NSTW, int NSTX>
__global__ __launch_bounds__(512) void igemm4_kernel(const void* hX, const void* hWt, int hK, int hCin, int hldx, int hldw, int hH, int hW,
                                                     int hksplit, int hnwg, unsigned hpacked, IgemmArgs a) {
  const IgemmHot h{(const half_t*)hX, (const half_t*)hWt, hK, hCin, hldx, hldw, hH, hW, hksplit, hnwg, hpacked};
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BK = 64;
  constexpr int XI = BM / 8 / 4;                   // activation DMA instructions per loader wave per slab
  constexpr int WI = BN / 8 / 4;                   // weight DMA instructions per loader wave per K-step
  constexpr int WSLOT = BN * BK;                   // halfs
  constexpr int XSLOT = (BM + 8) * BK;             // BM pixel rows + the zero row (row BM) + padding to a 1 KiB multiple
  constexpr int TMW = BM / 2, TNW = BN / 2, MT = TMW / 16, NT = TNW / 16;
  constexpr int DIST = NSTW - 1;                   // K-steps in flight
  static_assert(NSTX * 3 >= DIST + 3, "activation ring too shallow for the prefetch distance");
  static_assert(BM % WD == 0 && WD % 16 == 0, "a tile is whole image rows; a 16-pixel MFMA sub-tile stays inside a row");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  half_t* const sW = reinterpret_cast<half_t*>(smem_raw);      // [NSTW][WSLOT]
  half_t* const sX = sW + NSTW * WSLOT;                        // [NSTX][XSLOT]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int wm = wave >> 1, wn = wave & 1;
  IG_STAMP(0, wave8 == 0);
  // zero rows (never written by the DMA): the source of a shifted pixel that falls off its image row
  if (tid < NSTX * 8) *reinterpret_cast<f32x4*>(sX + (tid >> 3) * XSLOT + BM * BK + (tid & 7) * 8) = (f32x4){0.f, 0.f, 0.f, 0.f};
  int tile_m, tile_n, tile_z;
  {
    const int nx = h.ny_shift() >= 0 ? (1 << h.nx_shift()) : (int)gridDim.x, ny = h.ny_shift() >= 0 ? (1 << h.ny_shift()) : (int)gridDim.y;
    const int nwg = h.nwg;
    const int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7, pos = lin >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    if (h.ny_shift() >= 0) {
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
  const int G = 3 * NC;                            // slab groups (chunk, dy), three K-steps (dx) each
  const int per = (G + h.ksplit - 1) / h.ksplit;
  const int g0 = ks_id * per, g1 = min(G, g0 + per);
  const int ns = max(0, g1 - g0) * 3;
  const int n_img = h.hw_shift() >= 0 ? (m0 >> h.hw_shift()) : m0 / HW;
  const int y0 = (m0 - n_img * HW) / WD;           // first image row of this tile
  __syncthreads();                                 // the zero rows are in place

  // ---- loader state ----
  const int lrow = lane >> 3, pch = lane & 7;
  int xyr[XI], xcol[XI], xsc[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int row = (wave * XI + i) * 8 + lrow;    // pixel of the tile = LDS row of the slab
    xyr[i] = y0 + row / WD;
    xcol[i] = row % WD;
    xsc[i] = (pch ^ (row & 7)) * 8;
  }
  const half_t* wp[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = (wave * WI + i) * 8 + lrow;
    wp[i] = h.Wt + (long long)(n0 + row) * h.ldw + (pch ^ ((row >> 1) & 7)) * 8;
  }
  // issue head: next K-step to stage = (chunk hc, kernel row hdy, column hdx); weight columns (3 hdy + hdx) * Cin + 64 hc
  int hc = g0 / 3, hdy = g0 - hc * 3, hdx = 0;
  int hws = 0, hxs = 0;                            // ring slots of the head
#pragma unroll
  for (int i = 0; i < WI; ++i) wp[i] += (long long)(3 * hdy) * h.Cin + hc * BK;

  auto issue = [&]() {
    if (hdx == 0) {                                // a new slab: image rows y + hdy - 1 of chunk hc, staged once for dx = -1, 0, +1
      half_t* x = sX + hxs * XSLOT;
      const int dy = hdy - 1;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int yy = xyr[i] + dy;
        const bool ok = yy >= 0 && yy < H;
        const int yc = ok ? yy : 0;
        const long long pix = h.ups() ? ((long long)n_img * (HW >> 2) + (yc >> 1) * (WD >> 1) + (xcol[i] >> 1))
                                      : ((long long)n_img * HW + yc * WD + xcol[i]);
        const half_t* src = ok ? h.X + pix * h.ldx + hc * BK + xsc[i] : (const half_t*)g_zero_line4 + xsc[i];
        __builtin_amdgcn_global_load_lds(src, (lds_void4*)(x + (wave * XI + i) * 8 * BK), 16, 0, 0);
      }
      hxs = hxs + 1 == NSTX ? 0 : hxs + 1;
    }
    half_t* w = sW + hws * WSLOT;
#pragma unroll
    for (int i = 0; i < WI; ++i) __builtin_amdgcn_global_load_lds(wp[i], (lds_void4*)(w + (wave * WI + i) * 8 * BK), 16, 0, 0);
    hws = hws + 1 == NSTW ? 0 : hws + 1;
    // advance the head: dx, then dy, then the chunk (taps are consecutive Cin-wide column blocks of the weight row)
    long long dw = h.Cin;
    if (++hdx == 3) {
      hdx = 0;
      if (++hdy == 3) { hdy = 0; ++hc; dw = BK - 8 * h.Cin; }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) wp[i] += dw;
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  IG_STAMP(1, wave8 == 0);
  if (loader) {
#pragma unroll
    for (int s = 0; s < DIST; ++s)
      if (s < ns) issue();
    int ph = 0;
    for (int s = 0; s < ns; ++s) {
      // K-step s has landed when at most the loads issued after its weights are outstanding: steps s+1 .. s+DIST-1,
      // WI each plus XI for every slab that starts among them
      if (s + DIST <= ns) {
        if (ph == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * WI + ig4_slabs_ahead(0, DIST) * XI) : "memory");
        else if (ph == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * WI + ig4_slabs_ahead(1, DIST) * XI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * WI + ig4_slabs_ahead(2, DIST) * XI) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + DIST < ns) issue();
      ph = ph == 2 ? 0 : ph + 1;
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
    using std::integral_constant;
    // one K-step at shift DX (= dx + 1): second half read under the MFMAs of the first, first half of the next step
    // (shift (DX + 1) % 3, next weight slot, next slab slot when DX == 2) read right after its barrier
    auto step = [&](auto DX, int s) {
      constexpr int d = decltype(DX)::value;
      read_half(DX, 1, sw_off, sx_off, xb, wb);
      wait_frags(integral_constant<int, MT + NT>{}, xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(xa, wa);
      __builtin_amdgcn_sched_barrier(0);
      wait_frags(integral_constant<int, 0>{}, xb, wb);
      sw_off = sw_off + WSLOT * 2 == NSTW * WSLOT * 2 ? 0u : sw_off + WSLOT * 2;
      if (d == 2) sx_off = sx_off + XSLOT * 2 == NSTX * XSLOT * 2 ? 0u : sx_off + XSLOT * 2;
      if (s + 1 < ns) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_half(integral_constant<int, (d + 1) % 3>{}, 0, sw_off, sx_off, xa, wa);
        __builtin_amdgcn_sched_barrier(0);
      }
      mfma_half(xb, wb);
    };
    if (ns > 0) __builtin_amdgcn_s_barrier();      // step 0 has landed (the loader waves pass ns barriers, so do these)
    asm volatile("" ::: "memory");
    IG_STAMP(2, wave8 == 0);
    if (ns > 0) read_half(integral_constant<int, 0>{}, 0, sw_off, sx_off, xa, wa);
    for (int s = 0; s < ns; s += 3) {
      step(integral_constant<int, 0>{}, s);
      step(integral_constant<int, 1>{}, s + 1);
      step(integral_constant<int, 2>{}, s + 2);
    }
  }
  IG_STAMP(3, wave8 == 0);
  IG_STAMP(4, wave8 == 0);
  igemm_epilogue<MT, NT, TMW, TNW, BN, 512>(a, acc, m0, n0, wm, wn, lane, 0, ks_id, reinterpret_cast<float*>(smem_raw), !loader);
#endif
}

template <int BM, int BN, int WD, int NSTW, int NSTX>
static int launch4(const IgemmArgs& a, hipStream_t s) {
  constexpr size_t ring = (size_t)(NSTW * BN * 64 + NSTX * (BM + 8) * 64) * sizeof(half_t);
  constexpr size_t epi = (size_t)BM * (BN + 4) * 4 + (size_t)BM * (BN + 8) * 2 + 16384;      // staged epilogue: fp32 tile + fp16 tile + partial sums
  constexpr size_t smem = ring > epi ? ring : epi;
  static_assert(smem <= 163840, "LDS");
  auto kern = igemm4_kernel<BM, BN, WD, NSTW, NSTX>;
  ISHAP_TRY(ishap_set_max_lds((const void*)kern, (int)smem));
  dim3 grid(a.M / BM, ceil_div(a.N, BN), a.ksplit);
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
             (unsigned)(b.ny_shift & 0x3f) << 18 | (b.ups ? 1u << 24 : 0u);
  if (g_igemm_prof_start) hipExtLaunchKernelGGL(kern, grid, dim3(512), smem, s, g_igemm_prof_start, g_igemm_prof_stop, 0,
                                                (const void*)h.X, (const void*)h.Wt, h.K, h.Cin, h.ldx, h.ldw, h.H, h.W, h.ksplit, h.nwg, h.packed, b);
  else hipLaunchKernelGGL(kern, grid, dim3(512), smem, s,
                          (const void*)h.X, (const void*)h.Wt, h.K, h.Cin, h.ldx, h.ldw, h.H, h.W, h.ksplit, h.nwg, h.packed, b);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// the shapes this kernel takes: plain 3x3 (no folded second source), Cin % 64 == 0, one image per tile, a tile = whole image
// rows of a map 32 / 64 / 128 pixels wide
bool igemm4_applicable(const IgemmArgs& a, bool big) {
  const int BM = big ? 128 : 64;
  if (!a.conv3 || a.nbatch != 1 || a.Cin % 64 != 0 || a.K2 != 0 || a.K != 9 * a.Cin) return false;
  if (big ? a.W != 128 : (a.W != 32 && a.W != 64)) return false;
  if (BM % a.W != 0 || (a.H * a.W) % BM != 0 || a.M % BM != 0) return false;
  return true;
}

#ifndef IG4_BIG_W
#define IG4_BIG_W 6
#define IG4_BIG_X 3
#endif
#ifndef IG4_SMALL_W
#define IG4_SMALL_W 8
#define IG4_SMALL_X 4
#endif
int igemm4_launch_main(const IgemmArgs& a, bool big, hipStream_t s) {
  if (big) return launch4<128, 128, 128, IG4_BIG_W, IG4_BIG_X>(a, s);
  if (a.W == 64) return launch4<64, 64, 64, IG4_SMALL_W, IG4_SMALL_X>(a, s);
  return launch4<64, 64, 32, IG4_SMALL_W, IG4_SMALL_X>(a, s);
}
