// EXPERIMENT (round 3), measured and NOT adopted: 17.0 us per launch against 17.5 us for the two launches it replaces
// (tools/experiments/persist_chain.hip, profiles/round3_persistent_chain.txt).  The K loop of a workgroup that owns ALL of K for a
// 64-pixel map is bound by LDS fragment reads: every (tap, 32-deep k-step, 16-pixel block) needs its own 1 KiB activation
// fragment whatever the number of output channels, 1.2 MB per 1024 input channels = ~2100 cycles per 128-channel chunk,
// 8 chunks = 8 us -- the product's sliced kernel spreads exactly that over 4 workgroups per tile.  Kept as a record.
//
// conv8: convolution on an 8x8 map with the NEXT layer's GroupNorm applied in the epilogue -- one launch where the
// small-map path needed two (sliced conv, then a group-local GroupNorm pass that adds the K slices up).
// Reference arithmetic: guided_diffusion/unet.py:236-256 (ResBlock: conv -> GroupNorm32 -> FiLM -> SiLU -> conv),
// :299-305 (AttentionBlock: GroupNorm32 -> qkv conv1d, proj_out + residual), nn.py:16-18 (GroupNorm32 in fp32).
//
// Why this shape (round-3 measurements, profiles/round3_persistent_chain.txt, round3_launch_floor.txt): on the 8x8 level
// a layer is 64 pixels x 1024 channels against 2-19 MB of once-read weights; a dependent launch costs ~1.5 us of boundary
// plus its own ramp, a cross-workgroup hand-off through memory ~2.5 us, and a CU ingests ~64 GB/s by LDS-DMA.  Split-K
// needs a second pass over fp32 slices (the 7.5 us GroupNorm launch); a persistent chain needs two hand-offs per layer and
// measured 1.12x SLOWER than launches.  What removes time is fewer exchange points: here a workgroup owns ALL of K for
// 8 output channels of all 64 pixels (128 workgroups for 1024 channels), so its outputs are final when its K loop ends,
// and the only thing that crosses workgroups is the two GroupNorm partial sums of a channel group (4 workgroups share a
// group of 32 channels), as data-tagged granules -- the rendezvous the group-local GroupNorm kernels already use.
//
//   * weights: packed once at load in fragment order [8-channel tile][128-input-channel chunk][tap][32-deep k-step][8][32],
//     so a chunk of a tile is 18 KiB of contiguous memory = 18 linear LDS-DMA instructions (the row-major operand costs
//     16 cache lines per instruction when read as MFMA fragments);
//   * activations: the chunk's 64 pixels x 128 channels go into a zero-haloed 10 x 10 LDS tile by LDS-DMA (16-byte chunks
//     XOR-swizzled by pixel on the SOURCE side, rule 21); all nine taps read the same staged pixels at shifted offsets;
//   * 3-slot ring, one barrier per chunk, 4 loader waves + 8 MFMA waves (k-step = wave & 3, taps split in two halves);
//   * epilogue: 8-wave sum in LDS, bias (+ folded 1x1 skip bias) (+ residual), fp16 store of the conv output (skip
//     connections and the backward pass read it), then -- if the consumer is a GroupNorm -- statistics of the stored
//     values about a pivot, (hi, lo) granule exchange with the other workgroups of the group, normalise / FiLM / SiLU and a
//     second fp16 store: the activation the next conv8 launch stages directly.
#include "conv8.h"        // tools/experiments: NOT part of libishap_hip.so (measured, not adopted -- see the header below)

#include <cstdlib>
#include <type_traits>

#ifdef C8_STAMPS          // diagnostic build (tools/experiments/persist_chain.hip -DC8_STAMPS): s_memtime of wave 0 at the phase boundaries
extern __device__ unsigned long long* g_c8_stamps;      // [workgroup][8]
#define C8_STAMP(k)                                                                                    \
  do {                                                                                                 \
    if (threadIdx.x == 0) {                                                                            \
      unsigned long long t_;                                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
      g_c8_stamps[(size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 8 + (k)] = t_;                       \
    }                                                                                                  \
  } while (0)
#else
#define C8_STAMP(k) do {} while (0)
#endif

namespace {

constexpr int C8_NT = 4;                       // output channels per workgroup (256 workgroups for 1024 channels)
constexpr int C8_CH = 128;                     // input channels per staged chunk
constexpr int C8_CONS = 8, C8_LOAD = 4;        // MFMA waves; loader waves (2 for the weights, 2 for the activations)
constexpr int C8_THREADS = 64 * (C8_CONS + C8_LOAD);
constexpr int C8_W9 = 9 * 4 * C8_NT * 32 * 2;  // 9 216 B: weights of one 9-tap chunk of one tile (9 KiB-instructions)
constexpr int C8_W1 = 4 * C8_NT * 32 * 2;      // 1 024 B: 1-tap chunk
constexpr int C8_XP = 100;                     // padded 10 x 10 pixels
constexpr int C8_XB = C8_XP * C8_CH * 2;       // 25 600 B
constexpr int C8_WSLOTS = 8, C8_XSLOTS = 3;    // weights: up to 7 chunks (HBM-cold, ~2 us away) in flight; activations (L2): 2
constexpr int C8_XOFF = C8_WSLOTS * C8_W9;     // 73 728
constexpr int C8_ZERO = C8_XOFF + C8_XSLOTS * C8_XB;   // 150 528: 64 zero bytes behind the rings
constexpr int C8_LDS = C8_ZERO + 256;

typedef __attribute__((address_space(3))) void lds_v8;

__device__ __forceinline__ float rh8(float v) { return (float)(half_t)v; }
__device__ __forceinline__ float silu8(float v) { return v / (1.f + __expf(-v)); }

// (hi, lo) granule rendezvous of a channel group's partial sums -- same protocol as norm_local.hip's group_rendezvous
// (data-tagged 8-byte granules, one agent-scope store each, relaxed agent-scope polls, bounded, raises the status word)
__device__ __forceinline__ void c8_rendezvous(double& a, double& b, unsigned long long* rec, int part, int parts, double* scratch,
                                              unsigned* status, int spin_limit) {
  if (parts <= 1) return;
  if (threadIdx.x < GN_REC_PER_PART) {
    const double d = (threadIdx.x & 2) ? b : a;
    const float hi = (float)d;
    const float v = (threadIdx.x & 1) ? (float)(d - (double)hi) : hi;
    __hip_atomic_store(rec + part * GN_REC_PER_PART + threadIdx.x, (unsigned long long)__float_as_uint(v) | (1ull << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if ((int)threadIdx.x < GN_REC_PER_PART * parts) {
    unsigned long long gr = 0;
    int spins = 0;
    do {
      gr = __hip_atomic_load(rec + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } while ((gr >> 32) == 0ull && ++spins < spin_limit);
    double v = (double)__uint_as_float((unsigned)(gr & 0xffffffffull));
    if ((gr >> 32) == 0ull) {
      __hip_atomic_store(status, (unsigned)ISHAP_DEV_GN_RENDEZVOUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v = __builtin_nan("");
    }
    scratch[threadIdx.x] = v;
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
  for (int p = 0; p < parts; ++p) {
    ta += scratch[GN_REC_PER_PART * p] + scratch[GN_REC_PER_PART * p + 1];
    tb += scratch[GN_REC_PER_PART * p + 2] + scratch[GN_REC_PER_PART * p + 3];
  }
  a = ta;
  b = tb;
}

// s_waitcnt vmcnt(n) for a run-time n in [0, 35] (the immediate must be a constant)
__device__ __forceinline__ void c8_wait_vm(int n) {
  switch (n) {
#define C8_W(k) case k: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k) : "memory"); break;
    C8_W(1) C8_W(2) C8_W(3) C8_W(4) C8_W(5) C8_W(6) C8_W(7) C8_W(8) C8_W(9) C8_W(10) C8_W(11) C8_W(12) C8_W(13) C8_W(14) C8_W(15)
    C8_W(16) C8_W(17) C8_W(18) C8_W(19) C8_W(20) C8_W(21) C8_W(22) C8_W(23) C8_W(24) C8_W(25) C8_W(26) C8_W(27) C8_W(28) C8_W(29)
    C8_W(30) C8_W(31) C8_W(32) C8_W(33) C8_W(34) C8_W(35)
#undef C8_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

__global__ __launch_bounds__(C8_THREADS) void conv8_kernel(Conv8Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool loader = wave >= C8_CONS;
  const int tile = blockIdx.x, n_img = blockIdx.y;
  const int nch9 = a.taps == 9 ? a.Cin / C8_CH : 0;
  const int nch1 = (a.taps == 9 ? a.K2 : a.Cin) / C8_CH;
  const int nch = nch9 + nch1;
  const half_t* X9 = a.X + (size_t)n_img * 64 * a.ldx;
  const half_t* X1 = a.taps == 9 ? (a.X2 ? a.X2 + (size_t)n_img * 64 * a.ldx2 : nullptr) : X9;
  const int ld1 = a.taps == 9 ? a.ldx2 : a.ldx;
  const char* Wt = reinterpret_cast<const char*>(a.W8) + (size_t)tile * ((size_t)nch9 * C8_W9 + (size_t)nch1 * C8_W1);
  C8_STAMP(0);

  // epilogue operands of the 64 owner threads (wave 0: pixel = lane, this tile's 4 channels), fetched now so that their
  // round trips lie under the K loop
  const int t = threadIdx.x;
  const bool owner = t < 64;
  const int n = tile * C8_NT;
  const long long m = (long long)n_img * 64 + t;
  const int cn = a.norm_c0 + n;                          // channel index within the consumer's GroupNorm
  f32x4 e_bias = {0.f, 0.f, 0.f, 0.f}, e_gam = e_bias, e_bet = e_bias, e_sc = e_bias, e_sh = e_bias;
  half4 e_res = {0, 0, 0, 0};
  if (owner) {
    if (a.bias) e_bias = *reinterpret_cast<const f32x4*>(a.bias + n);
    if (a.bias2) e_bias += *reinterpret_cast<const f32x4*>(a.bias2 + n);
    if (a.res) e_res = *reinterpret_cast<const half4*>(a.res + m * a.ldr + n);
    if (a.gamma) {
      e_gam = *reinterpret_cast<const f32x4*>(a.gamma + cn);
      e_bet = *reinterpret_cast<const f32x4*>(a.beta + cn);
      if (a.film) {
        e_sc = *reinterpret_cast<const f32x4*>(a.emb + (long long)n_img * a.emb_ld + cn);
        e_sh = *reinterpret_cast<const f32x4*>(a.emb + (long long)n_img * a.emb_ld + a.norm_C + cn);
      }
    }
  }

  // zero the halo of the activation tiles once (the interior is rewritten by every chunk, the halo never) and the zero line
  for (int i = threadIdx.x; i < C8_XSLOTS * 36 * (C8_CH * 2 / 16); i += C8_THREADS) {
    const int s = i / (36 * 16), r = i - s * (36 * 16), hp = r >> 4, ch = r & 15;
    // border pixels of the 10 x 10 tile: rows 0 and 9 (20), columns 0 and 9 of rows 1..8 (16)
    const int pp = hp < 10 ? hp : (hp < 20 ? 90 + (hp - 10) : ((hp - 20) >> 1) * 10 + 10 + ((hp - 20) & 1) * 9);
    reinterpret_cast<f32x4*>(smem + C8_XOFF + s * C8_XB + pp * (C8_CH * 2))[ch] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (threadIdx.x < 16) reinterpret_cast<f32x4*>(smem + C8_ZERO)[threadIdx.x] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  C8_STAMP(1);

  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (loader) {
    const int lw = wave - C8_CONS;
    if (lw < 2) {
      // ---- weight waves: KiB i of chunk c goes to wave (i & 1); up to 7 chunks ahead (ring of 8) ----
      auto nins = [&](int c) { const int tot = c < nch9 ? 9 : 1; return (tot - lw + 1) >> 1; };
      auto issue = [&](int c) {
        char* slot = smem + (c % C8_WSLOTS) * C8_W9;
        const bool nine = c < nch9;
        const int tot = nine ? 9 : 1;
        const char* wsrc = nine ? Wt + (size_t)c * C8_W9 : Wt + (size_t)nch9 * C8_W9 + (size_t)(c - nch9) * C8_W1;
        for (int i = lw; i < tot; i += 2)
          __builtin_amdgcn_global_load_lds(wsrc + i * 1024 + lane * 16, (lds_v8*)(slot + i * 1024), 16, 0, 0);
      };
      int ahead = 0;                                       // this wave's instructions in flight beyond chunk c
      const int pre = nch < C8_WSLOTS - 1 ? nch : C8_WSLOTS - 1;
      for (int c = 0; c < pre; ++c) { issue(c); if (c > 0) ahead += nins(c); }
      for (int c = 0; c < nch; ++c) {
        c8_wait_vm(ahead);                                 // chunk c has landed
        __builtin_amdgcn_s_barrier();                      // B(c)
        asm volatile("" ::: "memory");
        if (c + 1 < nch) ahead -= nins(c + 1);             // chunk c + 1 becomes the one waited for next
        if (c + C8_WSLOTS - 1 < nch) { issue(c + C8_WSLOTS - 1); ahead += nins(c + C8_WSLOTS - 1); }
      }
    } else {
      // ---- activation waves: 16 KiB-instructions per chunk (image row r / 2, pixels (r & 1) * 4 .. +3), 8 per wave; 2 chunks ahead ----
      const int xw = lw - 2;
      auto issue = [&](int c) {
        char* slot = smem + C8_XOFF + (c % C8_XSLOTS) * C8_XB;
        const bool nine = c < nch9;
        const half_t* xs = nine ? X9 : X1;
        const int ld = nine ? a.ldx : ld1;
        const int c0 = (nine ? c : c - nch9) * C8_CH;
        for (int r = xw; r < 16; r += 2) {
          const int y = r >> 1, x = (r & 1) * 4 + (lane >> 4), pc = lane & 15;
          const int key = (((y + 1) & 1) << 3) | ((x + 1) & 7);
          const half_t* src = xs + (size_t)(y * 8 + x) * ld + c0 + ((pc ^ key) & 15) * 8;
          char* dst = slot + ((y + 1) * 10 + 1 + (r & 1) * 4) * (C8_CH * 2);
          __builtin_amdgcn_global_load_lds(src, (lds_v8*)dst, 16, 0, 0);
        }
      };
      issue(0);
      if (nch > 1) issue(1);
      for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // B(c)
        asm volatile("" ::: "memory");
        if (c + 2 < nch) issue(c + 2);                     // into the slot chunk c - 1 has just left
      }
    }
  } else {
    const int ks = wave & 3, hv = wave >> 2;
    const int col = lane & 15, g = lane >> 4;
    const int chunk16 = ks * 4 + g;                       // this lane's 16-byte chunk of a staged pixel row
    for (int c = 0; c < nch; ++c) {
      __builtin_amdgcn_s_barrier();                       // B(c)
      asm volatile("" ::: "memory");
      if (c == 0) C8_STAMP(2);
      const char* wslot = smem + (c % C8_WSLOTS) * C8_W9;
      const char* xs = smem + C8_XOFF + (c % C8_XSLOTS) * C8_XB;
      // every fragment read of the chunk is issued before the first MFMA (up to 25 ds_read_b128 in flight per wave): with a
      // read-then-use loop each MFMA waited out an LDS round trip
      auto xaddr = [&](int j, int dy, int dx) {
        const int pix = j * 16 + col, yy = (pix >> 3) + 1 + dy, xx = (pix & 7) + 1 + dx;
        const int key = ((yy & 1) << 3) | (xx & 7);
        return xs + (yy * 10 + xx) * (C8_CH * 2) + ((chunk16 ^ key) & 15) * 16;
      };
      if (c < nch9) {
        auto run = [&](auto first, auto count) {
          constexpr int T0 = decltype(first)::value, NTAP = decltype(count)::value;
          half8 wf[NTAP], xf[NTAP][4];
#pragma unroll
          for (int k = 0; k < NTAP; ++k) {
            const int tap = T0 + k, dy = tap / 3 - 1, dx = tap % 3 - 1;
            wf[k] = *reinterpret_cast<const half8*>(col < C8_NT ? wslot + ((tap * 4 + ks) * C8_NT + col) * 64 + g * 16 : smem + C8_ZERO);
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[k][j] = *reinterpret_cast<const half8*>(xaddr(j, dy, dx));
          }
#pragma unroll
          for (int k = 0; k < NTAP; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[k], xf[k][j], acc[j], 0, 0, 0);
        };
        if (hv) run(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{});
        else run(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{});
      } else if (hv == 0) {
        const half8 wf = *reinterpret_cast<const half8*>(col < C8_NT ? wslot + (ks * C8_NT + col) * 64 + g * 16 : smem + C8_ZERO);
        half8 xf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const half8*>(xaddr(j, 0, 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[j], acc[j], 0, 0, 0);
      }
    }
  }

  C8_STAMP(3);
  // ---- the 8 partial tiles meet in LDS (over the rings: every wave is past its last read of them) ----
  __syncthreads();
  f32x4* red = reinterpret_cast<f32x4*>(smem);                         // [8 waves][4 blocks][16 pixel lanes]: only lanes 0..15 hold real rows
  double* scratch = reinterpret_cast<double*>(smem + 8192);             // 64 doubles
  float* fsc = reinterpret_cast<float*>(smem + 8192 + 512);            // pivot + the owner wave's two sums
  if (!loader && lane < 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(wave * 4 + j) * 16 + lane] = acc[j];
  }
  __syncthreads();
  // owner thread t < 64: pixel t, channels n .. n + 3 (accumulator rows 0..3 = lanes 0..15 of block t >> 4)
  float yv[4] = {0.f, 0.f, 0.f, 0.f};
  if (owner) {
    const int j = t >> 4, ln = t & 15;
    f32x4 v = red[(0 * 4 + j) * 16 + ln];
#pragma unroll
    for (int w = 1; w < C8_CONS; ++w) v += red[(w * 4 + j) * 16 + ln];
    if (a.bias || a.bias2) v += e_bias;
    if (a.res) { v[0] += (float)e_res[0]; v[1] += (float)e_res[1]; v[2] += (float)e_res[2]; v[3] += (float)e_res[3]; }
    const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    *reinterpret_cast<half4*>(a.y + m * a.ldy + n) = o;
#pragma unroll
    for (int c = 0; c < 4; ++c) yv[c] = (float)o[c];
  }
  C8_STAMP(4);
  if (!a.gamma) return;                                  // no consumer GroupNorm: the conv output is all (block-uniform)
  // ---- statistics of this tile's 64 x 4 stored values about the pivot P = pixel 0's first value (wave 0 only) ----
  double s = 0.0, qq = 0.0;
  if (owner) {
    const float P = __shfl(yv[0], 0);
    float ls = 0.f, lq = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { const float d = yv[c] - P; ls += d; lq = fmaf(d, d, lq); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ls += __shfl_xor(ls, o); lq += __shfl_xor(lq, o); }
    if (lane == 0) { fsc[0] = P; fsc[1] = ls; fsc[2] = lq; }
  }
  __syncthreads();
  {
    // to pivot 0 in double: sum x = S + n P, sum x^2 = Q + P (2 S + n P), n = 256 values
    const double S = (double)fsc[1], Q = (double)fsc[2], Pd = (double)fsc[0];
    s = S + 256.0 * Pd;
    qq = Q + Pd * (2.0 * S + 256.0 * Pd);
  }
  const int parts = a.cpg / C8_NT;
  const int grp = a.group_base + (tile * C8_NT) / a.cpg, part = tile % parts;
  C8_STAMP(5);
  c8_rendezvous(s, qq, a.rec + ((long long)n_img * 32 + grp) * GN_REC_STRIDE, part, parts, scratch, a.status, a.spin_limit);
  C8_STAMP(6);
  const double cnt = 64.0 * (double)a.cpg;
  const double md = s / cnt;
  double vd = qq / cnt - md * md;
  vd = vd < 0.0 ? 0.0 : vd;
  const float mean = (float)md, rstd = (float)(1.0 / sqrt(vd + 1e-5));
  if (t == 0 && part == 0 && a.stats_out) {
    a.stats_out[(n_img * 32 + grp) * 2] = mean;
    a.stats_out[(n_img * 32 + grp) * 2 + 1] = rstd;
  }
  if (owner) {
    half4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float y = rh8((yv[c] - mean) * rstd * e_gam[c] + e_bet[c]);
      if (a.film) {
        const float sc = rh8(1.f + rh8(e_sc[c]));
        const float sh = rh8(e_sh[c]);
        y = rh8(rh8(y * sc) + sh);
      }
      if (a.act) y = rh8(silu8(y));
      o[c] = (half_t)y;
    }
    *reinterpret_cast<half4*>(a.a_out + m * a.lda + cn) = o;
  }
  C8_STAMP(7);
#endif
}

// src: row-major operand [rows_pad][ld] with k = tap * Cin + c (9-tap part), then 9 * Cin + c2 (1-tap part of K2 columns);
// dst: [tile][9-tap chunks: tap, k-step, row, 32][1-tap chunks: k-step, row, 32]
__global__ void conv8_pack_kernel(const half_t* __restrict__ src, int ld, int Cout, int Cin, int K2, int taps, half_t* __restrict__ dst) {
  const int nch9 = taps == 9 ? Cin / C8_CH : 0, nch1 = (taps == 9 ? K2 : Cin) / C8_CH;
  const long long per_tile = (long long)nch9 * (C8_W9 / 2) + (long long)nch1 * (C8_W1 / 2);
  const long long total = (long long)(Cout / C8_NT) * per_tile;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int tile = (int)(i / per_tile);
    long long r = i - (long long)tile * per_tile;
    int col;
    int row;
    if (r < (long long)nch9 * (C8_W9 / 2)) {
      const int c = (int)(r / (C8_W9 / 2));
      int e = (int)(r - (long long)c * (C8_W9 / 2));
      const int k = e & 31; e >>= 5;
      row = e % C8_NT; e /= C8_NT;
      const int kst = e & 3, tap = e >> 2;
      col = tap * Cin + c * C8_CH + kst * 32 + k;
    } else {
      r -= (long long)nch9 * (C8_W9 / 2);
      const int c = (int)(r / (C8_W1 / 2));
      int e = (int)(r - (long long)c * (C8_W1 / 2));
      const int k = e & 31; e >>= 5;
      row = e % C8_NT; e /= C8_NT;
      const int kst = e & 3;
      col = (taps == 9 ? 9 * Cin : 0) + c * C8_CH + kst * 32 + k;
    }
    dst[i] = src[(long long)(tile * C8_NT + row) * ld + col];
  }
}

}  // namespace

size_t conv8_packed_halfs(int Cout, int Cin, int K2, int taps) {
  const int nch9 = taps == 9 ? Cin / C8_CH : 0, nch1 = (taps == 9 ? K2 : Cin) / C8_CH;
  return (size_t)(Cout / C8_NT) * ((size_t)nch9 * (C8_W9 / 2) + (size_t)nch1 * (C8_W1 / 2));
}

bool conv8_shape_ok(int H, int W, int Cin, int K2, int Cout, int taps) {
  return H == 8 && W == 8 && (taps == 9 || taps == 1) && Cin % C8_CH == 0 && K2 % C8_CH == 0 && Cout % C8_NT == 0 && (taps == 9 || K2 == 0);
}

int conv8_pack(const half_t* src, int ld, int Cout, int Cin, int K2, int taps, half_t* dst, hipStream_t s) {
  ISHAP_REQUIRE(conv8_shape_ok(8, 8, Cin, K2, Cout, taps), "conv8 pack: shape");
  const size_t total = conv8_packed_halfs(Cout, Cin, K2, taps);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(conv8_pack_kernel, dim3(blocks), dim3(256), 0, s, src, ld, Cout, Cin, K2, taps, dst);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

int conv8_launch(const Conv8Args& a0, hipStream_t s) {
  Conv8Args a = a0;
  ISHAP_REQUIRE(conv8_shape_ok(8, 8, a.Cin, a.K2, a.Cout, a.taps), "conv8: 8x8 map, channels in multiples of 128 / 8");
  ISHAP_REQUIRE(a.X && a.W8 && a.y && a.N >= 1 && (a.K2 == 0 || a.X2), "conv8: null argument");
  ISHAP_REQUIRE(a.ldx % 8 == 0 && (a.K2 == 0 || a.ldx2 % 8 == 0) && a.ldy % 4 == 0 && (!a.res || a.ldr % 4 == 0), "conv8: row strides");
  ISHAP_REQUIRE(!a.gamma || (a.norm_c0 % 4 == 0 && a.norm_C % 4 == 0 && a.emb_ld % 4 == 0), "conv8: consumer channel offsets");
  if (a.gamma) {
    ISHAP_REQUIRE(a.beta && a.a_out && a.rec && a.cpg % C8_NT == 0 && a.cpg / C8_NT <= GN_REC_STRIDE / GN_REC_PER_PART && a.lda % 4 == 0 &&
                      (!a.film || a.emb),
                  "conv8: consumer GroupNorm operands (channels per group: a multiple of 4, at most 32)");
    ISHAP_REQUIRE((a.Cout / C8_NT) * a.N <= ishap_cu_count(), "conv8: the rendezvous needs the whole grid resident (one workgroup per CU)");
    a.status = ishap_status_word();
    ISHAP_REQUIRE(a.status != nullptr, "device status word");
    static const int lim = [] { const char* e = getenv("ISHAP_GN_SPIN_LIMIT"); const int v = e ? atoi(e) : 0; return v > 0 ? v : GN_SPIN_LIMIT; }();
    a.spin_limit = lim;
  }
  ISHAP_TRY(ishap_set_max_lds((const void*)conv8_kernel, C8_LDS));
  dim3 grid(a.Cout / C8_NT, a.N);
  if (g_igemm_prof_start) hipExtLaunchKernelGGL(conv8_kernel, grid, dim3(C8_THREADS), C8_LDS, s, g_igemm_prof_start, g_igemm_prof_stop, 0, a);
  else hipLaunchKernelGGL(conv8_kernel, grid, dim3(C8_THREADS), C8_LDS, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
