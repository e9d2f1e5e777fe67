// 3x3 convolution, third generation: LDS-DMA ring with activation reuse across the three horizontal taps.
// The implicit-GEMM kernels are bound by the global->LDS DMA rate (~10 TB/s chip-wide measured), not by MFMA, so the
// lever is bytes per FLOP.  A pixel tile here is a set of whole image rows; for each (dy, 64-channel chunk) the rows
// y+dy are staged ONCE with a one-pixel halo on both sides (out-of-image pixels come from a zero line) and the taps
// dx = -1, 0, +1 read them at shifted offsets.  Activation traffic drops 3x, total operand traffic ~1.5x.
// K order: (dy, chunk, dx), dx innermost.  Weights stream through a 4-slot ring (one slot per K-step), activation
// tiles through a 2-slot ring (one slot per three K-steps); everything is prefetched three K-steps ahead behind a
// counted s_waitcnt vmcnt and one raw barrier per K-step.  Swizzle: 16-byte chunk ^ (row & 7) -- conflict-free for
// fragment reads starting at ANY row, which the shifted reads need.  Same epilogue as igemm.hip / igemm2.hip.
#include "common.h"
#include "igemm_epilogue.h"

__device__ __attribute__((aligned(128))) half_t g_zero_line3[64];   // zero-initialised: source of padded pixels
typedef __attribute__((address_space(3))) void lds_void3;

// number of group starts (K-steps = 0 mod 3) among steps s+1 .. s+DIST-1 when s = ph (mod 3)
constexpr int starts_ahead(int ph, int dist) {
  int c = 0;
  for (int j = 1; j < dist; ++j) c += ((ph + j) % 3 == 0) ? 1 : 0;
  return c;
}

template <int BM, int BN, int NSTW, int NSTX>
__global__ __launch_bounds__(512) void igemm3_kernel(IgemmArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BK = 64;
  constexpr int WI = BN / 8 / 4;                   // weight DMA instructions per wave per K-step
  constexpr int XI = (BM == 128) ? 5 : 3;          // activation DMA instructions per wave per group (padded)
  constexpr int XROWS = XI * 4 * 8;                // LDS rows of one activation slot (>= R*(W+2))
  constexpr int WSLOT = BN * BK, XSLOT = XROWS * BK;
  constexpr int TMW = BM / 2, TNW = BN / 2, MT = TMW / 16, NT = TNW / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int DIST = NSTW - 1;                   // prefetch distance in K-steps
  static_assert(NSTX * 3 >= DIST + 3, "activation ring too shallow for the prefetch distance");
  half_t* sW = reinterpret_cast<half_t*>(smem_raw);      // [NSTW][BN*BK]
  half_t* sX = sW + NSTW * WSLOT;                        // [NSTX][XROWS*BK]

  // 8 waves: 0-3 compute, 4-7 only issue the DMA (same producer/consumer split as igemm2.hip)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int wm = wave >> 1, wn = wave & 1;
  int tile_m, tile_n, tile_z;
  {
    const int nx = gridDim.x, ny = gridDim.y;
    const int nwg = nx * ny * gridDim.z;
    const int lin = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7, pos = lin >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    tile_n = swz % ny;
    tile_m = (swz / ny) % nx;
    tile_z = swz / (ny * nx);
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int ks_id = tile_z;                        // nbatch == 1 for convolutions
  const int W = a.W, H = a.H, HW = H * W, W2 = W + 2;
  const int NC = a.Cin / BK;                       // channel chunks
  const int G = 3 * NC;                            // groups = (dy, chunk)
  const int per = (G + a.ksplit - 1) / a.ksplit;
  const int g0 = ks_id * per, g1 = min(G, g0 + per);
  const int ns = (g1 - g0) * 3;
  const int n_img = m0 / HW;
  const int y0 = (m0 - n_img * HW) / W;            // first image row of this tile
  const int R = BM / W;                            // image rows in the tile

  // ---- loader state ----
  const int lrow = lane >> 3, pch = lane & 7;
  const half_t* xbase[XI];                         // source pointer for dy = 0, chunk 0 (or null when the LDS row is padding)
  int xyr[XI];                                     // image row of this lane's LDS row (for the dy bounds test); -1000 = never valid
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int j = (wave * XI + i) * 8 + lrow;      // LDS row of the activation slot
    const int r = j / W2, xx = j - r * W2 - 1;
    const bool real = r < R && xx >= 0 && xx < W;
    xyr[i] = real ? y0 + r : -1000;
    xbase[i] = a.X + (((long long)n_img * H + (y0 + r)) * W + xx) * a.ldx + (pch ^ (j & 7)) * 8;
  }
  const half_t* wbase[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int row = (wave * WI + i) * 8 + lrow;
    wbase[i] = a.Wt + (long long)(n0 + row) * a.ldw + (pch ^ (row & 7)) * 8;
  }

  auto issue = [&](int s) {                        // s = K-step relative to this split
    const int g = g0 + s / 3, dxi = s - (s / 3) * 3;
    const int dyi = g / NC, chunk = g - dyi * NC;
    const int tap = dyi * 3 + dxi;
    half_t* w = sW + (s % NSTW) * WSLOT;
#pragma unroll
    for (int i = 0; i < WI; ++i)
      __builtin_amdgcn_global_load_lds(wbase[i] + tap * a.Cin + chunk * BK, (lds_void3*)(w + (wave * WI + i) * 8 * BK), 16, 0, 0);
    if (dxi == 0) {
      half_t* x = sX + ((s / 3) % NSTX) * XSLOT;
      const int dy = dyi - 1;
      const long long goff = (long long)dy * W * a.ldx + chunk * BK;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int yy = xyr[i] + dy;
        const half_t* src = (yy >= 0 && yy < H) ? xbase[i] + goff : (const half_t*)g_zero_line3;
        __builtin_amdgcn_global_load_lds(src, (lds_void3*)(x + (wave * XI + i) * 8 * BK), 16, 0, 0);
      }
    }
  };

  // fragment coordinates: LDS row (for dx = 0) of this lane's pixel in each 16-pixel sub-tile
  int xj[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int pl = wm * TMW + j * 16 + (lane & 15);
    const int r = pl / W;
    xj[j] = r * W2 + (pl - r * W) + 1;
  }

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (loader) {
#pragma unroll
    for (int s = 0; s < DIST; ++s)
      if (s < ns) issue(s);
  }

  for (int s = 0; s < ns; ++s) {
    // step s needs W(s) and X(group(s)); issued after them: issue(s+1), issue(s+2) = WI (+XI at a group start) each
    const int ph = s - (s / 3) * 3;
    if (loader) {
      if (s + DIST <= ns) {
        if (ph == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * WI + starts_ahead(0, DIST) * XI) : "memory");
        else if (ph == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * WI + starts_ahead(1, DIST) * XI) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * WI + starts_ahead(2, DIST) * XI) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (loader) {
      if (s + DIST < ns) issue(s + DIST);
      continue;
    }
    const half_t* bw = sW + (s % NSTW) * WSLOT;
    const half_t* bx = sX + ((s / 3) % NSTX) * XSLOT;
    const int dx = ph - 1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      half8 xf[MT], wf[NT];
      const int ch = (lane >> 4) + 4 * kk;
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int row = xj[j] + dx;
        xf[j] = *reinterpret_cast<const half8*>(bx + row * BK + ((ch ^ (row & 7)) * 8));
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int row = wn * TNW + i * 16 + (lane & 15);
        wf[i] = *reinterpret_cast<const half8*>(bw + row * BK + ((ch ^ (row & 7)) * 8));
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
  }
  igemm_epilogue<MT, NT, TMW, TNW, BN, 512>(a, acc, m0, n0, wm, wn, lane, 0, ks_id, reinterpret_cast<float*>(smem_raw), !loader);
#endif
}

template <int BM, int BN, int NSTW, int NSTX>
static int launch3(const IgemmArgs& a, hipStream_t s) {
  constexpr int XI = (BM == 128) ? 5 : 3;
  constexpr size_t ring = (size_t)(NSTW * BN * 64 + NSTX * XI * 32 * 64) * sizeof(half_t);
  constexpr size_t epi = (size_t)BM * (BN + 4) * 4 + (size_t)BM * (BN + 8) * 2 + 16384;      // staged epilogue: fp32 tile + fp16 tile + partial sums
  constexpr size_t smem = ring > epi ? ring : epi;
  static_assert(smem <= 163840, "LDS");
  auto kern = igemm3_kernel<BM, BN, NSTW, NSTX>;
  ISHAP_TRY(ishap_set_max_lds((const void*)kern, (int)smem));
  dim3 grid(a.M / BM, ceil_div(a.N, BN), a.ksplit);
  if (g_igemm_prof_start) hipExtLaunchKernelGGL(kern, grid, dim3(512), smem, s, g_igemm_prof_start, g_igemm_prof_stop, 0, a);
  else hipLaunchKernelGGL(kern, grid, dim3(512), smem, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// true when the shape fits this kernel: 3x3, no on-the-fly upsampling, Cin % 64 == 0, tile = whole image rows
bool igemm3_applicable(const IgemmArgs& a, bool big) {
  const int BM = big ? 128 : 64;
  if (!a.conv3 || a.ups || a.nbatch != 1 || a.Cin % 64 != 0 || a.K2 != 0) return false;
  if (a.W < 16 || a.W % 16 != 0 || BM % a.W != 0 || (a.H * a.W) % BM != 0) return false;
  const int R = BM / a.W;
  return R * (a.W + 2) <= (big ? 160 : 96);
}

#ifndef IG3_SMALL_W
#define IG3_SMALL_W 8
#define IG3_SMALL_X 4
#endif
int igemm3_launch_main(const IgemmArgs& a, bool big, hipStream_t s) {
  return big ? launch3<128, 128, 6, 3>(a, s) : launch3<64, 64, IG3_SMALL_W, IG3_SMALL_X>(a, s);
}
