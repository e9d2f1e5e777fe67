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

__device__ __forceinline__ void plane_axes(int p, int& a_col, int& a_row) {
  // grid[...,0] indexes W (columns), grid[...,1] indexes H (rows); drag_utils.py:318-321
  a_col = (p == 1) ? 1 : 0;
  a_row = (p == 0) ? 1 : 2;
}

struct Bilin { int x0, y0; float w[4]; };
__device__ __forceinline__ Bilin bilin_setup(float u, float v, int W) {
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

__global__ __launch_bounds__(256) void drag_motion_kernel(DragArgs a) {
  const int side = 2 * a.r + 1;
  const int npos = 3 * a.B * side * side;
  const int lane = threadIdx.x & 63;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  float lsum = 0.f;
  for (int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave < npos; wave += nwaves) {
  int j = wave % side;            // lattice index along the row axis
  int i = (wave / side) % side;   // along the column axis
  int b = (wave / (side * side)) % a.B;
  int p = wave / (side * side * a.B);
  int ac, ar;
  plane_axes(p, ac, ar);
  const float oi = a.voxel * (float)(i - a.r), oj = a.voxel * (float)(j - a.r);
  Bilin bs = bilin_setup(a.sources[b * 3 + ac] + oi, a.sources[b * 3 + ar] + oj, a.W);
  Bilin bt = bilin_setup(a.targets[b * 3 + ac] + oi, a.targets[b * 3 + ar] + oj, a.W);
  const float mult = (float)side;
  const float ntot = 3.f * (float)a.Cc * (float)a.B * (float)side * (float)side * (float)side;
  for (int c = lane; c < a.Cc; c += 64) {
    const int ch = a.chmap[p * a.Cc + c];
    float patch = 0.f, shift = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int xs = bs.x0 + (q & 1), ys = bs.y0 + (q >> 1);
      if (xs >= 0 && xs < a.W && ys >= 0 && ys < a.W) patch += bs.w[q] * (float)a.orig[((long long)ys * a.W + xs) * a.ld + ch];
      int xt = bt.x0 + (q & 1), yt = bt.y0 + (q >> 1);
      if (xt >= 0 && xt < a.W && yt >= 0 && yt < a.W) shift += bt.w[q] * (float)a.edit[((long long)yt * a.W + xt) * a.ld + ch];
    }
    const float d = shift - patch;
    float g;
    if (a.l1) { lsum += mult * fabsf(d); g = -((d > 0.f) - (d < 0.f)) * mult / ntot; }
    else { lsum += mult * d * d; g = -2.f * d * mult / ntot; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int xt = bt.x0 + (q & 1), yt = bt.y0 + (q >> 1);
      if (xt >= 0 && xt < a.W && yt >= 0 && yt < a.W)
        fx_add(a.gfx + ((long long)yt * a.W + xt) * a.ld + ch, bt.w[q] * g, DRAG_FX_SCALE);
    }
  }
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
  if (col >= 0 && col < a.W && row >= 0 && row < a.W) a.touched[(p * a.W + row) * a.W + col] = 1;
}

__global__ void drag_count_kernel(DragArgs a) {
  __shared__ int red[256];
  int cnt = 0;
  for (int i = threadIdx.x; i < 3 * a.W * a.W; i += 256) cnt += a.touched[i] ? 0 : 1;
  red[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) a.nmask[0] = red[0];
}

// mask term on untouched texels; one wave per (plane, texel)
__global__ __launch_bounds__(256) void drag_mask_kernel(DragArgs a) {
  const int lane = threadIdx.x & 63;
  const int WW = a.W * a.W;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const float denom = (float)a.Cc * (float)a.nmask[0];
  float lsum = 0.f;
  for (int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; wave < 3 * WW; wave += nwaves) {
  const int p = wave / WW, tex = wave % WW;
  if (a.touched[wave]) continue;
  for (int c = lane; c < a.Cc; c += 64) {
    const int ch = a.chmap[p * a.Cc + c];
    const long long o = (long long)tex * a.ld + ch;
    float d = (float)a.edit[o] - (float)a.orig[o];
    float g;
    if (a.l1) { lsum += fabsf(d); g = -a.cof * (float)((d > 0.f) - (d < 0.f)) / denom; }
    else { lsum += d * d; g = -a.cof * 2.f * d / denom; }
    fx_add(a.gfx + o, g, DRAG_FX_SCALE);   // (plane, c) -> unique ch, but chmap may repeat a channel (nearest resize)
  }
  }
  block_loss_add(lsum, a.acc + 1);
}

__global__ void drag_finish_kernel(DragArgs a) {
  const int side = 2 * a.r + 1;
  const float ntot = 3.f * (float)a.Cc * (float)a.B * (float)side * (float)side * (float)side;
  float loss = -((float)a.acc[0] * (1.f / DRAG_ACC_SCALE)) / ntot;
  if (a.cof > 0.f) loss -= a.cof * ((float)a.acc[1] * (1.f / DRAG_ACC_SCALE)) / ((float)a.Cc * (float)a.nmask[0]);
  a.loss[0] = loss;
}

// fixed-point scatter buffer -> the fp32 gradient the ABI returns
__global__ void drag_grad_out_kernel(const long long* __restrict__ gfx, float* __restrict__ grad, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    grad[i] = (float)gfx[i] * (1.f / DRAG_FX_SCALE);
}

int drag_setup_launch(const DragArgs& a, hipStream_t s) {
  const int side = 2 * a.r + 1;
  ISHAP_CHECK_HIP(hipMemsetAsync(a.touched, 0, (size_t)3 * a.W * a.W, s));
  int total = 3 * a.B * 2 * side * side;
  hipLaunchKernelGGL(drag_touch_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(drag_count_kernel, dim3(1), dim3(256), 0, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

int drag_loss_grad_launch(const DragArgs& a, hipStream_t s) {
  const int side = 2 * a.r + 1;
  const long long n = (long long)a.W * a.W * a.ld;
  ISHAP_CHECK_HIP(hipMemsetAsync(a.gfx, 0, (size_t)n * sizeof(long long), s));
  ISHAP_CHECK_HIP(hipMemsetAsync(a.acc, 0, 2 * sizeof(long long), s));
  int npos = 3 * a.B * side * side;
  hipLaunchKernelGGL(drag_motion_kernel, dim3(min(ceil_div(npos * 64, 256), 1024)), dim3(256), 0, s, a);
  if (a.cof > 0.f) hipLaunchKernelGGL(drag_mask_kernel, dim3(min(ceil_div(3 * a.W * a.W * 64, 256), 1024)), dim3(256), 0, s, a);
  hipLaunchKernelGGL(drag_finish_kernel, dim3(1), dim3(1), 0, s, a);
  hipLaunchKernelGGL(drag_grad_out_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, s, a.gfx, a.grad, n);
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
__global__ void pick_scale_kernel(const unsigned* __restrict__ bits, float* __restrict__ scale2) {
  float m = __uint_as_float(bits[0]);
  float sc = 1.f;
  if (m > 0.f && isfinite(m)) sc = exp2f(floorf(log2f(256.f / m)));
  sc = fminf(fmaxf(sc, 1.f / 1048576.f), 1.0e30f);
  scale2[0] = sc;
  scale2[1] = 1.f / sc;
}
__global__ void scale_to_f16_kernel(const float* __restrict__ g, half_t* __restrict__ o, const float* __restrict__ scale2,
                                    long long n) {
  const float sc = scale2[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    o[i] = (half_t)(g[i] * sc);
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
