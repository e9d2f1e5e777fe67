// 3x3 convolutions on the small maps (8x8, 16x16, 32x32 at batch 1: M = 64 .. 1024 pixels) -- the weight-streaming
// layers of the UNet (guided_diffusion/unet.py:185,211 for input_blocks 7..14, middle_block, output_blocks 0..8 and the
// input gradients of the same).  They move 5-38 MB of fp16 weights for 1-5 GFLOP, so the job is to read every weight
// ONCE at HBM rate and never leave the chip in between.  With the tiled kernel that took split-K over workgroups, fp32
// partial slices in HBM and a reduce launch.  Here one launch does it:
//   * a workgroup owns 64 pixels (whole rows of one image) x 16 output channels and ALL of K;
//   * K is split over its 9 consumer waves, one per tap of the stencil; a wave's MFMA weight operand comes straight
//     from global memory into registers (512 contiguous bytes per weight row and 256-channel chunk), eight K-steps =
//     sixteen 1-KiB loads in flight per wave, so the streamed-once bytes never touch LDS;
//   * the activation tile is staged by 3 loader waves with LDS-DMA, one channel chunk at a time (the tile's rows plus
//     one row above and below; rows outside the image come from a zero line, columns outside it are a per-lane select
//     of a zero chunk), into a three-slot ring with ONE barrier per chunk; all nine taps of a chunk read the same staged
//     pixels at shifted offsets, so activation traffic from L2 is 1/9 of the im2col volume;
//   * the 9 partial accumulators meet in LDS once (fixed order: bitwise reproducible) and go through the usual
//     epilogue duties (bias, residual, fp16 store, GroupNorm statistics / GroupNorm-backward sums as 64-bit
//     fixed-point atomics), exactly like igemm_skinny.hip.
// A folded 1x1 skip convolution (K2 columns from X2, unet.py:222,256) rides along as extra one-tap chunks.
#include "common.h"
#include "gn_bwd_terms.h"
#include "igemm_epilogue.h"      // IG_STAMP (diagnostic builds of the harness only)

#include <algorithm>
#include <type_traits>

namespace {

constexpr int S3_CONS = 9;                   // consumer waves: one per tap of the 3x3 stencil
constexpr int S3_LOAD = 3;                   // loader waves
constexpr int S3_THREADS = 64 * (S3_CONS + S3_LOAD);   // 768 threads = 3 waves per SIMD -> a 168-VGPR budget
constexpr int S3_RING = 8;                   // weight K-steps in flight per consumer wave
constexpr int S3_SLOTS = 3;                  // activation ring
__device__ __attribute__((aligned(128))) half_t g_zero_line_s3[256];   // zero-initialised: what rows outside the image read
typedef __attribute__((address_space(3))) void lds_void_s3;

struct S3Geom {
  int TH;          // image rows per tile (TH * W == 64)
  int CH;          // channels per staged chunk (64, 128 or 256; divides Cin and K2)
  int HP;          // staged pixels per tile: (TH + 2) * W
  int slot_bytes;  // ring slot = HP * CH * 2 (a multiple of 1 KiB)
  int ninstr;      // DMA instructions per chunk (slot_bytes / 1024)
  int w_shift;     // log2(W)
  // K is cut into `nslice` runs of whole chunks, one run per blockIdx.z (9-tap chunks first, then the 1-tap chunks of a
  // folded 1x1 source): slice z owns chunks [cut[z], cut[z+1]).  nslice > 1: the workgroups of a tile write fp32 partial
  // tiles to a.ws (slice-major, like split-K) and the consumer adds them up.
  int nslice;
  int cut[17];
};

// physical 16-byte chunk of staged pixel hp <-> logical chunk (the same involution on the DMA source and on the read)
__device__ __forceinline__ int s3_swz(int chunk, int hp, int CH) {
  return CH == 64 ? (chunk ^ ((hp >> 1) & 7)) : (chunk ^ (hp & 15));
}

// SPC = 64-channel sub-blocks per staged chunk (CH / 64): the K-steps a consumer wave owns in one chunk
template <int SPC>
__global__ __launch_bounds__(S3_THREADS) void conv3_small_kernel(const void* hX, const void* hWt, int hH, int hW, int hCin, int hldx, int hldw, int hTH,
                                                                 int hslot_bytes, int hninstr, int hwshift, IgemmArgs a, S3Geom geo) {
  // h*: copies of the fields X, Wt, H, W, Cin, ldx, ldw of `a` and TH, slot_bytes, ninstr, w_shift of `geo` as leading scalar
  // parameters, preloaded into SGPRs at dispatch (common.h, IgemmHot): the index arithmetic runs under the s_load of the blocks
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MT = 4;
  constexpr int CH = SPC * 64;
  constexpr int UNR = S3_RING / SPC;              // chunks per trip of the unrolled consumer loop
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wave >= S3_CONS;
  const int g = lane >> 4, col = lane & 15;
  const int W = hW, H = hH, HW = H * W;
  // XCD-aware order: the pixel tiles that stream the same weight rows run next to each other on one XCD (shared L2)
  int tile_m, tile_n;
  {
    const int nx = gridDim.x, ny = gridDim.y, nwg = nx * ny;     // (per slice: blockIdx.z is left alone)
    const int lin = blockIdx.x + nx * blockIdx.y;
    const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7, pos = lin >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    tile_m = swz % nx;
    tile_n = swz / nx;
  }
  const int tiles_per_img = H / hTH;
  const int n_img = tile_m / tiles_per_img;
  const int row0 = (tile_m - n_img * tiles_per_img) * hTH;      // first image row of the tile
  const int m0 = tile_m * 64;                                      // first output row (NHWC pixel index)
  const int n0 = tile_n * 16;
  const int nch9 = hCin / CH;
  const int slice = blockIdx.z;
  const int c_lo = geo.cut[slice], c_hi = geo.cut[slice + 1];      // this workgroup's chunks (global chunk numbers)
  const int nl = c_hi - c_lo;                                      // ring slots are indexed by the local number c - c_lo
  const int c9_hi = c_hi < nch9 ? c_hi : nch9;                     // 9-tap chunks: [c_lo, c9_hi); 1-tap: [max(c_lo, nch9), c_hi)
  const int c1_lo = (c_lo > nch9 ? c_lo : nch9) - nch9, c1_hi = c_hi - nch9;
  char* ring = smem_raw;
  const int zero_off = S3_SLOTS * hslot_bytes;                  // 64 zero bytes behind the ring

  IG_STAMP(0, wave == 0);
  f32x4 acc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (threadIdx.x < 4) reinterpret_cast<f32x4*>(smem_raw + zero_off)[threadIdx.x] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (loader) {
    // ---- loader waves: stage chunk c into slot c % 3.  Instruction i of the chunk covers slot bytes [i*1024, +1024):
    //      lane l -> staged pixel hp = byte / (CH*2) = (tile row hy - 1, column x), physical chunk pc; it fetches the
    //      swizzled logical chunk of that pixel's channels, or zeros for a row outside the image.  No per-instruction
    //      tables: a table that spills turns every DMA into a scratch reload on the same vmcnt queue. ----
    const int lw = wave - S3_CONS;
    constexpr int row_bytes = CH * 2;
    constexpr int rb_shift = SPC == 4 ? 9 : (SPC == 2 ? 8 : 7);
    const int lane_byte = lane * 16;
    const int lane_pix = lane_byte >> rb_shift;                        // pixel within the instruction's 1 KiB
    const int pc = (lane_byte & (row_bytes - 1)) >> 4;                 // physical 16-byte chunk within the pixel row
    constexpr int ppi = 1024 >> rb_shift;                              // pixels per instruction
    const int mine = (hninstr - lw + S3_LOAD - 1) / S3_LOAD;        // instructions of this wave per chunk
    auto issue = [&](int l) {
      const int c = c_lo + l;
      char* slot = ring + (l % S3_SLOTS) * hslot_bytes;
      const bool one = c >= nch9;                      // a chunk of the folded 1x1 source
      const half_t* base = one ? a.X2 : reinterpret_cast<const half_t*>(hX);
      const int ld = one ? a.ldx2 : hldx;
      const int c0 = (one ? c - nch9 : c) * CH;
      const bool ups = !one && a.ups;
      for (int i = lw; i < hninstr; i += S3_LOAD) {
        const int hp = i * ppi + lane_pix;
        const int hy = hp >> hwshift, x = hp & (W - 1);
        const int y = row0 + hy - 1;
        const bool ok = y >= 0 && y < H;
        const int sc = s3_swz(pc, hp, CH);
        const long long prow = ups ? (long long)n_img * (HW >> 2) + (y >> 1) * (W >> 1) + (x >> 1)
                                   : (long long)n_img * HW + y * W + x;
        const half_t* src = ok ? base + prow * ld + c0 + sc * 8 : g_zero_line_s3 + (sc & 31) * 8;
#ifndef S3_ABL_NOX      // (harness ablation: no activation staging)
        __builtin_amdgcn_global_load_lds(src, (lds_void_s3*)(slot + i * 1024), 16, 0, 0);
#endif
      }
    };
    auto wait_landed = [&](bool younger_in_flight) {
      // chunk c must have landed; the chunk issued after it may stay in flight (vmcnt is in-order)
      if (younger_in_flight) {
        switch (mine) {
#define S3_W(n) case n: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); break;
          S3_W(1) S3_W(2) S3_W(3) S3_W(4) S3_W(5) S3_W(6) S3_W(7) S3_W(8) S3_W(9) S3_W(10) S3_W(11) S3_W(12) S3_W(13)
          S3_W(14) S3_W(15) S3_W(16) S3_W(17) S3_W(18) S3_W(19) S3_W(20) S3_W(21) S3_W(22)
#undef S3_W
          default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    issue(0);
    if (nl > 1) issue(1);
    for (int l = 0; l < nl; ++l) {
      wait_landed(l + 1 < nl);
      __builtin_amdgcn_s_barrier();                    // B(l): chunk l is visible; the consumers are done with chunk l-1
      asm volatile("" ::: "memory");
      if (l + 2 < nl) issue(l + 2);                    // into the slot chunk l-1 has just left
    }
  } else {
    // ---- consumer waves: wave t owns TAP t of every 9-tap chunk -- its SPC 64-channel sub-blocks are its K-steps -- and
    //      (waves 0 .. SPC-1) one sub-block of every 1-tap chunk.  A fixed tap per wave means fixed offsets into the
    //      staged tile (computed once).  Weight fragments (A operand: 16 output channels x 32 k) go straight from global
    //      memory to registers, EIGHT K-steps (sixteen 1-KiB loads per wave, ~144 KB per CU) ahead of their use: the
    //      loads are inline asm and the waits are counted by hand, because hipcc either drains the queue at the first use
    //      or clusters the refills behind the MFMAs (both measured: the loop then runs at one HBM latency per chunk).  The
    //      schedule is branch-free -- a step past the end loads from the zero line and adds zeros -- so every step issues
    //      exactly 2 loads and the vmcnt constant holds: when a step waits, the younger loads are those of 7 steps. ----
    const int tap = wave;
    struct WFrag { half8 w[2]; };
    WFrag ringw[S3_RING];
    const half_t* wbase = reinterpret_cast<const half_t*>(hWt) + (long long)(n0 + col) * hldw + 8 * g;
    const half_t* zbase = g_zero_line_s3 + 8 * g;
    // LDS byte offset (and swizzle key) of this lane's staged pixel for each 16-pixel sub-tile, at this wave's tap and at
    // the centre tap; a column outside the image reads the zero chunk instead
    int xoff[MT], xkey[MT], xoff1[MT], xkey1[MT];
    bool xz[MT];
    {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int p = j * 16 + col;
        const int ty = p >> hwshift, tx = p & (W - 1);
        const int hc = (ty + 1) * W + tx, hp = hc + dy * W + dx;
        xz[j] = tx + dx < 0 || tx + dx >= W;
        xoff[j] = hp * (CH * 2); xkey[j] = hp;
        xoff1[j] = hc * (CH * 2); xkey1[j] = hc;
      }
    }
    auto wload = [&](const half_t* q, WFrag& f) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(f.w[0]) : "v"(q) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=&v"(f.w[1]) : "v"(q) : "memory");
    };
    // wait until at most Y loads are outstanding, then hand the fragment's registers to the MFMAs: the "+v" operands
    // make the wait the producer of those values, so no use can be scheduled above it
    auto wwait = [&](WFrag& f, auto younger) {
      constexpr int Y = decltype(younger)::value;
      asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f.w[0]), "+v"(f.w[1]) : "n"(Y) : "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    auto step = [&](const char* slot, const int (&off)[MT], const int (&key)[MT], bool use_z, int sub, const WFrag& f) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        half8 xf[MT];
#pragma unroll
        for (int jj = 0; jj < MT; ++jj) {
          const int chunk = sub * 8 + g + 4 * kk;
          const char* p = (use_z && xz[jj]) ? smem_raw + zero_off : slot + off[jj] + s3_swz(chunk, key[jj], CH) * 16;
          xf[jj] = *reinterpret_cast<const half8*>(p);
        }
#pragma unroll
        for (int jj = 0; jj < MT; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.w[kk], xf[jj], acc[jj], 0, 0, 0);
      }
    };
    auto load9 = [&](int c, int sub, WFrag& f) {
#ifdef S3_ABL_NOW       // (harness ablation: no weight stream)
      wload(zbase, f);
#else
      wload(c < c9_hi ? wbase + (long long)tap * hCin + (long long)c * CH + sub * 64 : zbase, f);
#endif
    };
    auto chunk9 = [&](int c, auto first) {
      constexpr int R0 = decltype(first)::value;       // first ring register set of this chunk
      __builtin_amdgcn_s_barrier();                    // B(c)
      asm volatile("" ::: "memory");
      const char* slot = ring + ((c - c_lo) % S3_SLOTS) * hslot_bytes;
#pragma unroll
      for (int sub = 0; sub < SPC; ++sub) {
        wwait(ringw[R0 + sub], std::integral_constant<int, (S3_RING - 1) * 2>{});
        step(slot, xoff, xkey, true, sub, ringw[R0 + sub]);
        __builtin_amdgcn_sched_barrier(0);
        load9(c + UNR, sub, ringw[R0 + sub]);          // refill the same registers, eight K-steps ahead
      }
    };
    IG_STAMP(1, wave == 0);
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int sub = 0; sub < SPC; ++sub) load9(c_lo + u, sub, ringw[u * SPC + sub]);
    IG_STAMP(2, wave == 0);
    for (int c = c_lo; c < c9_hi; c += UNR) {
      chunk9(c, std::integral_constant<int, 0>{});
      if constexpr (UNR > 1) { if (c + 1 < c9_hi) chunk9(c + 1, std::integral_constant<int, SPC>{}); }
      if constexpr (UNR > 2) {
        if (c + 2 < c9_hi) chunk9(c + 2, std::integral_constant<int, 2 * SPC>{});
        if (c + 3 < c9_hi) chunk9(c + 3, std::integral_constant<int, 3 * SPC>{});
      }
      if constexpr (UNR > 4) {
        if (c + 4 < c9_hi) chunk9(c + 4, std::integral_constant<int, 4 * SPC>{});
        if (c + 5 < c9_hi) chunk9(c + 5, std::integral_constant<int, 5 * SPC>{});
        if (c + 6 < c9_hi) chunk9(c + 6, std::integral_constant<int, 6 * SPC>{});
        if (c + 7 < c9_hi) chunk9(c + 7, std::integral_constant<int, 7 * SPC>{});
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus (zero-line) refills
    if (c1_hi > c1_lo) {
      // folded 1x1 source: the centre tap only, SPC K-steps per chunk -> waves 0 .. SPC-1 own one each; two chunks ahead
      const bool own1 = wave < SPC;
      const int sub1 = own1 ? wave : 0;
      auto load1 = [&](int c1, WFrag& f) {
        wload((own1 && c1 < c1_hi) ? wbase + (long long)9 * hCin + (long long)c1 * CH + sub1 * 64 : zbase, f);
      };
      auto chunk1 = [&](int c1, WFrag& f) {
        __builtin_amdgcn_s_barrier();                  // B(nch9 + c1)
        asm volatile("" ::: "memory");
        const char* slot = ring + ((nch9 + c1 - c_lo) % S3_SLOTS) * hslot_bytes;
        wwait(f, std::integral_constant<int, 2>{});
        step(slot, xoff1, xkey1, false, sub1, f);
        __builtin_amdgcn_sched_barrier(0);
        load1(c1 + 2, f);
      };
      load1(c1_lo, ringw[0]);
      load1(c1_lo + 1, ringw[1]);
      for (int c1 = c1_lo; c1 < c1_hi; c1 += 2) {
        chunk1(c1, ringw[0]);
        if (c1 + 1 < c1_hi) chunk1(c1 + 1, ringw[1]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }

  // ---- the 9 partial tiles meet in LDS, over the ring: every wave must be past its last read of it first ----
  IG_STAMP(3, wave == 0);
  __syncthreads();
  IG_STAMP(4, wave == 0);
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                 // [S3_CONS][MT][64]
  if (!loader) {
#pragma unroll
    for (int j = 0; j < MT; ++j) red[(wave * MT + j) * 64 + lane] = acc[j];
  }
  __syncthreads();
  IG_STAMP(5, wave == 0);
  if (wave >= MT) return;

  // ---- wave j finishes sub-tile j: lane = (pixel m, channels n .. n+3) ----
  const int j = wave;
  f32x4 v = red[(0 * MT + j) * 64 + lane];
#pragma unroll
  for (int w = 1; w < S3_CONS; ++w) v += red[(w * MT + j) * 64 + lane];
  const int m = m0 + j * 16 + col;
  const int n = n0 + g * 4;
  const bool ok = n < a.N;
  if (geo.nslice > 1) {          // a partial tile: the consumer adds the slices (bias, residual, statistics happen there)
    if (ok) *reinterpret_cast<f32x4*>(a.ws + ((long long)slice * a.M + m) * a.N + n) = v;
    IG_STAMP(6, wave == 0);
    return;
  }
  const int p = m - n_img * HW;
  const int oy = p / W, ox = p - oy * W;
  float s1v[4] = {0.f, 0.f, 0.f, 0.f}, s2v[4] = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    v *= a.alpha;
    if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.bias2) v += *reinterpret_cast<const f32x4*>(a.bias2 + n);
    if (a.res) {
      const long long rrow = a.res_ups ? ((long long)n_img * (HW >> 2) + (oy >> 1) * (W >> 1) + (ox >> 1)) : m;
      const half4 r = *reinterpret_cast<const half4*>(a.res + rrow * a.ldr + n);
      v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
    }
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
  }
  if (a.stat_out || a.gb_x) {
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
#endif
}

bool s3_geometry(const IgemmArgs& a, S3Geom& geo) {
  if (!a.conv3 || a.nbatch != 1 || (a.W != 8 && a.W != 16 && a.W != 32)) return false;
  geo.TH = 64 / a.W;
  geo.w_shift = a.W == 8 ? 3 : (a.W == 16 ? 4 : 5);
  if (a.H % geo.TH != 0 || a.M % 64 != 0) return false;
  if (a.Cin % 64 != 0 || a.K != 9 * a.Cin + a.K2 || a.K2 % 64 != 0) return false;
  geo.HP = (geo.TH + 2) * a.W;
  // the widest chunk that divides both channel counts and leaves room for three slots (+ the zero chunk)
  int ch = 256;
  while (ch >= 64) {
    if (a.Cin % ch == 0 && a.K2 % ch == 0 && (size_t)S3_SLOTS * geo.HP * ch * 2 + 1024 <= 160 * 1024) break;
    ch >>= 1;
  }
  if (ch < 64) return false;
  geo.CH = ch;
  geo.slot_bytes = geo.HP * ch * 2;
  if (geo.slot_bytes % 1024 != 0) return false;
  geo.ninstr = geo.slot_bytes / 1024;
  if ((geo.ninstr + S3_LOAD - 1) / S3_LOAD > 22) return false;
  // slices of about equal work (a 9-tap chunk = 9 units, a 1-tap chunk = 1), whole chunks each
  const int n9 = a.Cin / ch, n1 = a.K2 / ch, nch = n9 + n1;
  int ns = a.ksplit < 1 ? 1 : a.ksplit;
  if (ns > nch) ns = nch;
  if (ns > 16) ns = 16;
  const int total = 9 * n9 + n1;
  geo.cut[0] = 0;
  int c = 0, done = 0;
  for (int z = 1; z < ns; ++z) {
    const int goal = (int)((long long)total * z / ns);
    while (c < nch - (ns - z) && done + (c < n9 ? 9 : 1) <= goal) { done += c < n9 ? 9 : 1; ++c; }
    if (c <= geo.cut[z - 1]) { done += c < n9 ? 9 : 1; ++c; }      // every slice owns at least one chunk
    geo.cut[z] = c;
  }
  geo.cut[ns] = nch;
  geo.nslice = ns;
  return true;
}

template <int SPC>
int launch_s3(const IgemmArgs& a, const S3Geom& geo, hipStream_t s) {
  const size_t red = (size_t)S3_CONS * 4 * 64 * sizeof(f32x4);
  const size_t smem = std::max<size_t>((size_t)S3_SLOTS * geo.slot_bytes + 1024, red);
  ISHAP_REQUIRE(smem <= 160 * 1024, "small-map conv: LDS");
  auto kern = conv3_small_kernel<SPC>;
  ISHAP_TRY(ishap_set_max_lds((const void*)kern, 160 * 1024));
  dim3 grid(a.M / 64, (a.N + 15) / 16, geo.nslice);
  if (g_igemm_prof_start) hipExtLaunchKernelGGL(kern, grid, dim3(S3_THREADS), smem, s, g_igemm_prof_start, g_igemm_prof_stop, 0,
                                                (const void*)a.X, (const void*)a.Wt, a.H, a.W, a.Cin, a.ldx, a.ldw, geo.TH, geo.slot_bytes, geo.ninstr, geo.w_shift, a, geo);
  else hipLaunchKernelGGL(kern, grid, dim3(S3_THREADS), smem, s, (const void*)a.X, (const void*)a.Wt, a.H, a.W, a.Cin, a.ldx, a.ldw, geo.TH,
                          geo.slot_bytes, geo.ninstr, geo.w_shift, a, geo);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// shapes this kernel takes: 3x3 (+ folded 1x1) on maps 8, 16 or 32 pixels wide, channel counts that are multiples of 64,
// dense fp16 output, weights padded to 128 rows like every packed operand
bool igemm_small3_applicable(const IgemmArgs& a) {
  S3Geom geo;
  if (a.out_mode != IG_OUT_F16 || a.N % 4 != 0 || (a.ksplit > 1 && !a.ws)) return false;
  if (a.ldx % 8 != 0 || (a.K2 && a.ldx2 % 8 != 0)) return false;
  return s3_geometry(a, geo);
}

// how many K slices (workgroups per output tile) fill the chip for this shape: about `target` workgroups in all
int igemm_small3_slices(const IgemmArgs& a) {
  static const int target = [] { const char* e = getenv("ISHAP_S3_WGS"); return e ? atoi(e) : 384; }();
  IgemmArgs t = a;
  t.ksplit = 1;
  S3Geom geo;
  if (!s3_geometry(t, geo)) return 1;
  const long long wg1 = (long long)(a.M / 64) * ((a.N + 15) / 16);
  const int nch = a.Cin / geo.CH + a.K2 / geo.CH;
  int ns = (int)((target + wg1 / 2) / wg1);
  if (ns < 1) ns = 1;
  if (ns > nch) ns = nch;
  if (ns > 16) ns = 16;
  return ns;
}

// a.ksplit = K slices (1: finished outputs; > 1: fp32 partial tiles in a.ws, slice-major like split-K)
int igemm_small3_launch(const IgemmArgs& a, int, hipStream_t s) {
  S3Geom geo;
  ISHAP_REQUIRE(igemm_small3_applicable(a) && s3_geometry(a, geo), "small-map conv: shape");
  ISHAP_REQUIRE(geo.nslice == a.ksplit, "small-map conv: more K slices than channel chunks");
  ISHAP_REQUIRE(!(a.stat_out || a.gb_x) || (a.H * a.W) % 16 == 0, "a 16-pixel sub-tile must not straddle images");
  switch (geo.CH) {
    case 256: return launch_s3<4>(a, geo, s);
    case 128: return launch_s3<2>(a, geo, s);
    default: return launch_s3<1>(a, geo, s);
  }
}
