// Drag guidance: motion-supervision loss on the UNet decoder feature and its gradient.
// Reference: drag_utils.py:141-159 (resize_feat_align: [1,2c,H,W] -> [3, 2*(c//3), H, W]),
// :309-334 (lattices, planar grids, int16 texel ids, complement "mask" sets via Python sets),
// :355-382 (grid_sample bilinear/zeros/align_corners=True of origin at the source lattice and of the
// edited feature at the target lattice; L2 or L1 mean; mask regulariser on untouched texels).
// The reference back-propagates this through autograd; here the gradient w.r.t. the tap is
// produced directly, in the tap's own NHWC layout, ready for the UNet backward pass.
//
// Observation that shrinks the work 2r+1 times: plane xy only sees (x,y) of a lattice point, so the
// (2r+1) lattice points that differ in z are identical terms of the mean.  Each plane therefore has
// B*(2r+1)^2 distinct sample positions with multiplicity (2r+1).  A wave owns one position; lanes run
// over channels, which are contiguous in NHWC (coalesced 2-byte reads, fp32 atomics for the scatter).
#include "drag.h"
#include <cstdlib>

__device__ __forceinline__ void plane_axes(int p, int& a_col, int& a_row) {
  // grid[...,0] indexes W (columns), grid[...,1] indexes H (rows); drag_utils.py:318-321
  a_col = (p == 1) ? 1 : 0;
  a_row = (p == 0) ? 1 : 2;
}

struct Bilin { int x0, y0; float w[4]; };
__device__ __forceinline__ Bilin bilin_setup(float u, float v, int W) {
#pragma clang fp contract(off)      // the same (u, v) must give the same weights at every call site (zero loss at zero displacement)
  Bilin b;
  float ix = ((u + 1.f) / 2.f) * (float)(W - 1);
  float iy = ((v + 1.f) / 2.f) * (float)(W - 1);
  float fx = floorf(ix), fy = floorf(iy);
  b.x0 = (int)fx; b.y0 = (int)fy;
  float wx1 = ix - fx, wx0 = (fx + 1.f) - ix, wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
  b.w[0] = wx0 * wy0; b.w[1] = wx1 * wy0; b.w[2] = wx0 * wy1; b.w[3] = wx1 * wy1;
  return b;
}

// workgroup sum of the per-lane loss terms, then ONE atomic on the accumulator (thousands of same-address atomics
// -- one per wave -- were the whole run time of these kernels)
__device__ __forceinline__ void fx_add(long long* dst, float v, float scale) {
  atomicAdd(reinterpret_cast<unsigned long long*>(dst), (unsigned long long)__float2ll_rn(v * scale));
}
__device__ __forceinline__ void block_loss_add(float lsum, long long* dst) {
  __shared__ float red[4];
  for (int o = 32; o > 0; o >>= 1) lsum += __shfl_xor(lsum, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) fx_add(dst, red[0] + red[1] + red[2] + red[3], DRAG_ACC_SCALE);
}

constexpr int DSEG = 5;     // consecutive lattice positions a wave walks
// Motion term.  A wave owns DSEG consecutive positions of one lattice ROW of one (plane, handle) pair and one chunk of 64
// channels, and walks them in order.  The bilinear row pair (y0, y0+1) of the target is the same for the whole row and
// consecutive positions are a fraction of a texel apart, so the scatter of a row lands in a few texel columns: the
// contributions are summed in registers per column and leave as one fixed-point atomic per (column, row) instead of four
// per position (3x fewer 64-bit atomics).  Summation order is fixed, the final
// adds are integer, so edits stay bitwise repeatable.
__device__ __forceinline__ void drag_motion_body(const DragArgs& a, int blk, int nblk) {
  const int side = 2 * a.r + 1;
  const int nchunk = (a.Cc + 63) / 64;
  const int nseg = (side + DSEG - 1) / DSEG;
  const int nrows = 3 * a.B * side * nchunk * nseg;
  const int lane = threadIdx.x & 63;
  const int nwaves = (nblk * blockDim.x) >> 6;
  const float mult = (float)side;
  const float ntot = 3.f * (float)a.Cc * (float)a.B * (float)side * (float)side * (float)side;
  float lsum = 0.f;
  for (int wave = (blk * blockDim.x + threadIdx.x) >> 6; wave < nrows; wave += nwaves) {
    const int seg = wave % nseg;
    const int cchunk = (wave / nseg) % nchunk;
    const int j = (wave / (nseg * nchunk)) % side;           // lattice index along the row axis
    const int b = (wave / (nseg * nchunk * side)) % a.B;
    const int p = wave / (nseg * nchunk * side * a.B);
    int ac, ar;
    plane_axes(p, ac, ar);
    const int c = cchunk * 64 + lane;
    const bool live = c < a.Cc;
    const int ch = live ? a.chmap[p * a.Cc + c] : 0;
    const float oj = a.voxel * (float)(j - a.r);
    const float su = a.sources[b * 3 + ac], sv = a.sources[b * 3 + ar] + oj;
    const float tu = a.targets[b * 3 + ac], tv = a.targets[b * 3 + ar] + oj;
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;     // [row y0 / y0+1][column xcur / xcur+1]
    int xcur = 0, ycur = 0;
    bool open = false;
    auto flush_col = [&](int x, float v0, float v1) {
#ifdef DRAG_ABL_NOATOM      // timing ablation (tools/build_variant.sh): the scatter's atomics left out -- wrong gradients
      if (v0 != 12345.f) return;
#endif
      if (!live || x < 0 || x >= a.W) return;
      if (ycur >= 0 && ycur < a.W) fx_add(a.gfx + ((long long)ycur * a.W + x) * a.ld + ch, v0, DRAG_FX_SCALE);
      if (ycur + 1 >= 0 && ycur + 1 < a.W) fx_add(a.gfx + ((long long)(ycur + 1) * a.W + x) * a.ld + ch, v1, DRAG_FX_SCALE);
    };
    // the segment's DSEG positions: all their texel loads are issued before the first is used.  Round 6: really -- written as
    // `if (in bounds) patch = fma(w, texel[...], patch)` every one of the 40 loads sat in a conditional block of its own, and the
    // compiler closes such a block with s_waitcnt vmcnt(0): 41 of the kernel's 46 waits stood right behind a load (26 us per step).
    // Now every load is unconditional at a clamped coordinate and a sample outside the map (zeros padding, drag_utils.py:355)
    // enters with weight 0: fma(0, texel, acc) = acc, the same sums (the sign of a zero apart, which the fixed-point scatter drops).
    float dseg[DSEG];
    half_t tp[DSEG][4], te[DSEG][4];
    float wp[DSEG][4], we[DSEG][4];
#pragma unroll
    for (int ii = 0; ii < DSEG; ++ii) {
      const int i = seg * DSEG + ii;
      const float oi = a.voxel * (float)(i - a.r);
      const Bilin bs = bilin_setup(su + oi, sv, a.W);
      const Bilin bt = bilin_setup(tu + oi, tv, a.W);
      const bool on = live && i < side;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int xs = bs.x0 + (q & 1), ys = bs.y0 + (q >> 1);
        const int xt = bt.x0 + (q & 1), yt = bt.y0 + (q >> 1);
        const int xsc = min(max(xs, 0), a.W - 1), ysc = min(max(ys, 0), a.W - 1);
        const int xtc = min(max(xt, 0), a.W - 1), ytc = min(max(yt, 0), a.W - 1);
        tp[ii][q] = a.orig[((long long)ysc * a.W + xsc) * a.ld + ch];
        te[ii][q] = a.edit[((long long)ytc * a.W + xtc) * a.ld + ch];
        wp[ii][q] = (on && xs == xsc && ys == ysc) ? bs.w[q] : 0.f;
        we[ii][q] = (on && xt == xtc && yt == ytc) ? bt.w[q] : 0.f;
      }
    }
#pragma unroll
    for (int ii = 0; ii < DSEG; ++ii) {
      float patch = 0.f, shift = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // explicit fused multiply-adds on both sides: identical inputs must give shift == patch exactly
        patch = __fmaf_rn(wp[ii][q], (float)tp[ii][q], patch);
        shift = __fmaf_rn(we[ii][q], (float)te[ii][q], shift);
      }
      dseg[ii] = shift - patch;
    }
#pragma unroll
    for (int ii = 0; ii < DSEG; ++ii) {
      const int i = seg * DSEG + ii;
      if (i >= side) break;
      const float oi = a.voxel * (float)(i - a.r);
      const Bilin bt = bilin_setup(tu + oi, tv, a.W);
      const float d = dseg[ii];
      float g;
      if (a.l1) { if (live) lsum += mult * fabsf(d); g = -((d > 0.f) - (d < 0.f)) * mult / ntot; }
      else { if (live) lsum += mult * d * d; g = -2.f * d * mult / ntot; }
      if (!open || bt.x0 != xcur || bt.y0 != ycur) {         // wave-uniform: the coordinates do not depend on the lane
        if (open) {
          flush_col(xcur, a00, a10);
          if (bt.x0 == xcur + 1 && bt.y0 == ycur) { a00 = a01; a10 = a11; }
          else { flush_col(xcur + 1, a01, a11); a00 = 0.f; a10 = 0.f; }
          a01 = 0.f; a11 = 0.f;
        }
        xcur = bt.x0; ycur = bt.y0; open = true;
      }
      a00 += bt.w[0] * g; a01 += bt.w[1] * g; a10 += bt.w[2] * g; a11 += bt.w[3] * g;
    }
    if (open) { flush_col(xcur, a00, a10); flush_col(xcur + 1, a01, a11); }
  }
  block_loss_add(lsum, a.acc + 0);
}

// touched[p][row][col] = 1 where a rounded lattice texel of any source/target point lands (drag_utils.py:322-334)
__global__ void drag_touch_kernel(DragArgs a) {
  const int side = 2 * a.r + 1;
  const int total = 3 * a.B * 2 * side * side;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int j = idx % side;
  int i = (idx / side) % side;
  int st = (idx / (side * side)) % 2;
  int b = (idx / (side * side * 2)) % a.B;
  int p = idx / (side * side * 2 * a.B);
  int ac, ar;
  plane_axes(p, ac, ar);
  const float* pt = st ? a.targets : a.sources;
  float u = pt[b * 3 + ac] + a.voxel * (float)(i - a.r);
  float v = pt[b * 3 + ar] + a.voxel * (float)(j - a.r);
  // th.round((p + 1) * (W - 1) / 2).type(int16): round-half-even, then wrap to int16
  int col = (int)(short)rintf((u + 1.f) * (float)(a.W - 1) / 2.f);
  int row = (int)(short)rintf((v + 1.f) * (float)(a.W - 1) / 2.f);
  // bit 0: the reference's rounded-texel sets; bit 1: texels the motion scatter can reach (the four bilinear corners of a
  // target position) -- the gather pass reads the scatter buffer only there.  Byte-wide atomic OR via the 32-bit word.
  auto mark = [&](int r_, int c_, unsigned bit) {
    if (c_ < 0 || c_ >= a.W || r_ < 0 || r_ >= a.W) return;
    const long long o = ((long long)p * a.W + r_) * a.W + c_;
    atomicOr(reinterpret_cast<unsigned*>(a.touched + (o & ~3ll)), bit << (8 * (o & 3)));
  };
  mark(row, col, 1u);
  if (st) {
    const Bilin bt = bilin_setup(u, v, a.W);
    // one texel of slack around the 2x2 corners: the scatter recomputes u, v and a differently contracted
    // multiply-add must not put a corner outside the marked set
    for (int dy = -1; dy <= 2; ++dy)
      for (int dx = -1; dx <= 2; ++dx) mark(bt.y0 + dy, bt.x0 + dx, 2u);
  }
}

__global__ void drag_count_kernel(DragArgs a) {
  __shared__ int red[256];
  int cnt = 0;
  for (int i = threadIdx.x; i < 3 * a.W * a.W; i += 256) cnt += (a.touched[i] & 1) ? 0 : 1;
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) a.nmask[0] = red[0];
}

// chw[p][ch] = number of (plane p, c) pairs that resize_feat_align maps to tap channel ch (0 or 1; 2 where the nearest
// resize repeats a channel): lets the gather pass below add the mask term per tap element without a scatter
__global__ void drag_chan_weight_kernel(DragArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 3 * a.ld) return;
  const int p = idx / a.ld, ch = idx - p * a.ld;
  int cnt = 0;
  for (int c = 0; c < a.Cc; ++c) cnt += a.chmap[p * a.Cc + c] == ch ? 1 : 0;
  a.chw[idx] = (unsigned char)min(cnt, 255);
}

__global__ __launch_bounds__(256) void drag_terms_kernel(DragArgs a, unsigned* absmax_bits) {
  if (absmax_bits && blockIdx.x == 0 && threadIdx.x == 0) absmax_bits[0] = 0u;   // for the atomicMax of the pass that follows
  drag_motion_body(a, blockIdx.x, gridDim.x);
}

// loss from the two fixed-point sums; leaves them zero for the next call
__device__ __forceinline__ void drag_finish(const DragArgs& a) {
  const int side = 2 * a.r + 1;
  const float ntot = 3.f * (float)a.Cc * (float)a.B * (float)side * (float)side * (float)side;
  float loss = -((float)a.acc[0] * (1.f / DRAG_ACC_SCALE)) / ntot;
  if (a.cof > 0.f) loss -= a.cof * ((float)a.acc[1] * (1.f / DRAG_ACC_SCALE)) / ((float)a.Cc * (float)a.nmask[0]);
  a.loss[0] = loss;
  a.acc[0] = 0;
  a.acc[1] = 0;
}
__global__ void drag_finish_kernel(DragArgs a) { drag_finish(a); }

// Gather pass: fixed-point scatter buffer (motion term) -> the fp32 gradient the ABI returns, plus the MASK term
// (drag_utils.py:376-381: cof * mean over the untouched texels of |edit - orig|^2 or |.|), which is elementwise in the tap's
// own layout once the channel map is inverted (chw) -- it used to be a second 2-million-atomic scatter.  The buffer is
// left zero for the next call (no memset launch); with `absmax_bits` the pass also finds max|g| for the loss scale.
__global__ __launch_bounds__(256) void drag_gather_kernel(DragArgs a, unsigned* __restrict__ absmax_bits) {
  typedef long long ll2 __attribute__((ext_vector_type(2)));
  typedef unsigned char uc8 __attribute__((ext_vector_type(8)));
  const long long n = (long long)a.W * a.W * a.ld;
  const int WW = a.W * a.W;
  const bool mask = a.cof > 0.f;
  const float denom = (float)a.Cc * (float)a.nmask[0];
  float m = 0.f, lsum = 0.f;
  // a thread takes 8 consecutive channels of one texel (ld % 8 == 0): 16-byte accesses on every stream
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (long long)gridDim.x * blockDim.x * 8) {
    float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int tex = (int)(i / a.ld), ch = (int)(i - (long long)tex * a.ld);
    const unsigned char t0 = a.touched[tex], t1 = a.touched[WW + tex], t2 = a.touched[2 * WW + tex];
    // the mask term's operands, requested together with the footprint bytes (round 6: they used to be issued behind the footprint
    // test and, plane by plane, inside conditional blocks -- each closed by an s_waitcnt vmcnt(0))
    const half8 e8 = *reinterpret_cast<const half8*>(a.edit + i), o8 = *reinterpret_cast<const half8*>(a.orig + i);
    uc8 cw[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) cw[p] = *reinterpret_cast<const uc8*>(a.chw + p * a.ld + ch);
    if ((t0 | t1 | t2) & 2) {                       // a target footprint covers this texel on some plane
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const ll2 v = *reinterpret_cast<const ll2*>(a.gfx + i + 2 * q);
        *reinterpret_cast<ll2*>(a.gfx + i + 2 * q) = (ll2){0, 0};
        g[2 * q] = (float)v[0] * (1.f / DRAG_FX_SCALE);
        g[2 * q + 1] = (float)v[1] * (1.f / DRAG_FX_SCALE);
      }
    }
    if (mask) {
      const unsigned char tp[3] = {t0, t1, t2};
      int w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int p = 0; p < 3; ++p)
        if (!(tp[p] & 1)) {
#pragma unroll
          for (int k = 0; k < 8; ++k) w[k] += cw[p][k];
        }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = (float)e8[k] - (float)o8[k], wk = (float)w[k];
        if (a.l1) { lsum += wk * fabsf(d); g[k] += wk * (-a.cof * (float)((d > 0.f) - (d < 0.f)) / denom); }
        else { lsum += wk * d * d; g[k] += wk * (-a.cof * 2.f * d / denom); }
      }
    }
    *reinterpret_cast<f32x4*>(a.grad + i) = (f32x4){g[0], g[1], g[2], g[3]};
    *reinterpret_cast<f32x4*>(a.grad + i + 4) = (f32x4){g[4], g[5], g[6], g[7]};
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fabsf(g[k]));
  }
  if (mask) block_loss_add(lsum, a.acc + 1);
  if (absmax_bits) {
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(absmax_bits, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
  }
}

int drag_setup_launch(const DragArgs& a, hipStream_t s) {
  const int side = 2 * a.r + 1;
  ISHAP_REQUIRE((3 * a.W * a.W) % 4 == 0 && a.ld % 8 == 0, "drag: 3*W*W must be a multiple of 4 and the tap channels of 8");
  ISHAP_CHECK_HIP(hipMemsetAsync(a.touched, 0, (size_t)3 * a.W * a.W, s));
  // the scatter buffer and the loss sums start at zero here; every loss call leaves them zero again
  ISHAP_CHECK_HIP(hipMemsetAsync(a.gfx, 0, (size_t)a.W * a.W * a.ld * sizeof(long long), s));
  ISHAP_CHECK_HIP(hipMemsetAsync(a.acc, 0, 2 * sizeof(long long), s));
  int total = 3 * a.B * 2 * side * side;
  hipLaunchKernelGGL(drag_touch_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(drag_count_kernel, dim3(1), dim3(256), 0, s, a);
  hipLaunchKernelGGL(drag_chan_weight_kernel, dim3(ceil_div(3 * a.ld, 256)), dim3(256), 0, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// every workgroup of these passes ends with same-address atomics (loss sum, max|g|) that serialise at ~10 ns each:
// few, fat workgroups (the loops are grid-stride).  Measured: tools/experiments/drag_probe.sh.
static int drag_terms_launch(const DragArgs& a, unsigned* bits, hipStream_t s) {
  constexpr int cap = 1024;          // swept in round 2 (tools/experiments/drag_probe.sh)
  const int side = 2 * a.r + 1;
  const int nrows = 3 * a.B * side * ((a.Cc + 63) / 64) * ((side + DSEG - 1) / DSEG);
  hipLaunchKernelGGL(drag_terms_kernel, dim3(min(ceil_div(nrows * 64, 256), cap)), dim3(256), 0, s, a, bits);
  return 0;
}
static unsigned gather_blocks(long long n) {
  constexpr int cap = 512;
  return (unsigned)std::min<long long>((n / 8 + 255) / 256, cap);
}

// requires ishap_drag_setup on these buffers first (it zeroes the scratch this call leaves zero again)
int drag_loss_grad_launch(const DragArgs& a, hipStream_t s) {
  const long long n = (long long)a.W * a.W * a.ld;
  ISHAP_REQUIRE(a.ld % 8 == 0, "drag: tap channels must be a multiple of 8");
  ISHAP_TRY(drag_terms_launch(a, nullptr, s));
  hipLaunchKernelGGL(drag_gather_kernel, dim3(gather_blocks(n)), dim3(256), 0, s, a, (unsigned*)nullptr);
  hipLaunchKernelGGL(drag_finish_kernel, dim3(1), dim3(1), 0, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---- fp32 gradient -> scaled fp16 cotangent (power-of-two loss scale chosen from max|g|) ----
__global__ void absmax_kernel(const float* __restrict__ g, long long n, unsigned* __restrict__ out_bits) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(g[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out_bits, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));   // one per workgroup
}
__device__ __forceinline__ float pick_scale(float m);
__global__ void pick_scale_kernel(const unsigned* __restrict__ bits, float* __restrict__ scale2) {
  const float sc = pick_scale(__uint_as_float(bits[0]));
  scale2[0] = sc;
  scale2[1] = 1.f / sc;
}
__device__ __forceinline__ float pick_scale(float m) {
  float sc = 1.f;
  if (m > 0.f && isfinite(m)) sc = exp2f(floorf(log2f(256.f / m)));
  return fminf(fmaxf(sc, 1.f / 1048576.f), 1.0e30f);
}
__global__ void scale_to_f16_kernel(const float* __restrict__ g, half_t* __restrict__ o, const float* __restrict__ scale2,
                                    long long n) {
  const float sc = scale2[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    o[i] = (half_t)(g[i] * sc);
}
// the same with the scale picked by every thread from max|g| (published by thread 0) and the drag loss finished here:
// three launches fewer per guided step
__global__ void drag_scale_kernel(const float* __restrict__ g, half_t* __restrict__ o, const unsigned* __restrict__ bits,
                                  float* __restrict__ scale2, DragArgs a, long long n) {
  const float sc = pick_scale(__uint_as_float(bits[0]));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scale2[0] = sc;
    scale2[1] = 1.f / sc;
    drag_finish(a);
  }
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * blockDim.x * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + i);
    *reinterpret_cast<half4*>(o + i) = (half4){(half_t)(v[0] * sc), (half_t)(v[1] * sc), (half_t)(v[2] * sc), (half_t)(v[3] * sc)};
  }
}
// loss + gradient + scaled fp16 cotangent of one guided step in three launches (motion scatter, gather + mask, scale)
int drag_loss_cotangent_launch(const DragArgs& a, half_t* cot, unsigned* bits, float* scale2, hipStream_t s) {
  const long long n = (long long)a.W * a.W * a.ld;
  ISHAP_REQUIRE(a.ld % 8 == 0, "drag: tap channels must be a multiple of 8");
  ISHAP_TRY(drag_terms_launch(a, bits, s));
  hipLaunchKernelGGL(drag_gather_kernel, dim3(gather_blocks(n)), dim3(256), 0, s, a, bits);
  hipLaunchKernelGGL(drag_scale_kernel, dim3((unsigned)std::min<long long>((n / 4 + 255) / 256, 1024)), dim3(256), 0, s,
                     (const float*)a.grad, cot, (const unsigned*)bits, scale2, a, n);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
int grad_to_scaled_f16_launch(const float* g, half_t* o, unsigned* bits, float* scale2, long long n, hipStream_t s) {
  ISHAP_CHECK_HIP(hipMemsetAsync(bits, 0, sizeof(unsigned), s));
  int blocks = (int)std::min<long long>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(absmax_kernel, dim3(std::min(blocks, 512)), dim3(256), 0, s, g, n, bits);
  hipLaunchKernelGGL(pick_scale_kernel, dim3(1), dim3(1), 0, s, bits, scale2);
  hipLaunchKernelGGL(scale_to_f16_kernel, dim3(blocks), dim3(256), 0, s, g, o, scale2, n);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
