// Triplane -> occupancy-logit decode on the fp32 matrix cores.
// Reference: triplane_decoder/axisnetworks.py:537-562 (MultiTriplane.forward: three bilinear
// grid_samples, summed), :86-90 (Fourier features: [sin(2*pi*f@B) | cos(...)]), :526-535
// (Linear128-ReLU-Linear128-ReLU-Linear1); triplane_decoder/visualize.py:79-97 (dense grid,
// 50 000-point chunks with a host round trip each -- here the whole grid stays on the device).
//
// The reference decoder is fp32.  The Fourier projection (K = 32, feeds sin / cos of phases up to tens of radians) runs
// on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain).  The two 128 x 128 layers -- 93 % of the FLOPs -- run on the fp16
// matrix pipe at fp32 grade: weights and activations are split x = x_hi + x_lo (two fp16 each), the product is formed as
// w_hi x_hi + w_lo x_hi + w_hi x_lo with fp32 accumulation (the dropped w_lo x_lo term is 2^-22 relative), i.e. three
// v_mfma_f32_32x32x16_f16 (32 cycles each, K = 16) where the fp32 pipe needs sixteen 64-cycle MFMAs: 5.3x fewer matrix
// cycles for fp32-grade results (parity tests: 1e-4 abs + 1e-4 rel against the fp32 reference, unchanged).
// One wave owns 32 query points (lane&31 = point, lane>>5 = which half of the K slots it feeds).
// Weights are the MFMA A operand (rows = output neuron) read from LDS; activations are the B operand
// (column = point).  D[neuron][point] keeps the point on the lane, so a layer's accumulator registers
// ARE the next layer's B operands: register s of lane-half h holds neuron (s&3)+8*(s>>2)+4h; for the fp16 layers
// registers 8s..8s+7 (converted pairwise) are the fragment of K-step s, and the weights are stored in LDS in that same
// permuted K order (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand").  Activations never touch LDS.
#include "decode.h"

#define LDH 136   // padded row stride (halfs) of the split weight images: 272 B = 16 B mod 256 -> conflict-free ds_read_b128
#define LDB 36    // padded row stride (floats) of the Fourier matrix

__device__ __forceinline__ int kmap(int s, int h) { return (s & 3) + 8 * (s >> 2) + 4 * h; }

// sin/cos of an fp32 angle: 3-term Cody-Waite reduction by pi/2 (exact for |x| < ~1e4, far above the Fourier
// phases seen here) + the classic single-precision minimax polynomials on [-pi/4, pi/4]; ~1e-7 absolute.
// (ocml's sincosf inlines its huge-argument path 64 times per tile and blows the register budget.)
__device__ __forceinline__ void sincos_cw(float x, float& s, float& c) {
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(-k, 1.5703125f, x);
  r = fmaf(-k, 4.837512969970703125e-4f, r);
  r = fmaf(-k, 7.54978995489188e-8f, r);
  const float z = r * r;
  const float sp = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float cp = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                        fmaf(z, -0.5f, 1.f));
  const int q = (int)k & 3;
  const float s1 = (q & 1) ? cp : sp;
  const float c1 = (q & 1) ? sp : cp;
  s = (q & 2) ? -s1 : s1;
  c = ((q + 1) & 2) ? -c1 : c1;
}

// latent [96][S][S] (NCHW, one shape) * range + middle  ->  planes [3][S][S][32]   (drag_utils.py:295)
__global__ void planes_prepare_kernel(const float* __restrict__ latent, const float* __restrict__ rng,
                                      const float* __restrict__ mid, float* __restrict__ planes, int S) {
  __shared__ float tile[32][33];
  const int p = blockIdx.z;
  const int pix0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int SS = S * S;
  for (int j = ty; j < 32; j += 8) {      // j = channel, tx = pixel
    int c = p * 32 + j;
    float r = rng ? rng[c] : 1.f, m = mid ? mid[c] : 0.f;
    tile[j][tx] = latent[(long long)c * SS + pix0 + tx] * r + m;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8)        // j = pixel, tx = channel
    planes[((long long)p * SS + pix0 + j) * 32 + tx] = tile[tx][j];
}
int planes_prepare_launch(const float* latent, const float* rng, const float* mid, float* planes, int S, hipStream_t s) {
  ISHAP_REQUIRE((S * S) % 32 == 0, "plane size");
  hipLaunchKernelGGL(planes_prepare_kernel, dim3(S * S / 32, 1, 3), dim3(256), 0, s, latent, rng, mid, planes, S);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(512) void triplane_decode_kernel(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  half_t* sW1h = reinterpret_cast<half_t*>(lds);          // [128][LDH] hi parts, K in fragment order
  half_t* sW1l = sW1h + 128 * LDH;
  half_t* sW2h = sW1l + 128 * LDH;
  half_t* sW2l = sW2h + 128 * LDH;
  float* sBt = reinterpret_cast<float*>(sW2l + 128 * LDH);   // [64][LDB]   Bt[y][k] = _B[k][y]
  float* sb1 = sBt + 64 * LDB;
  float* sb2 = sb1 + 128;
  float* sw3 = sb2 + 128;
  const int tid = threadIdx.x;
  for (int i = tid; i < 128 * 128; i += 512) {
    // LDS position p of row r holds input neuron k = 32 qb + 16 s + 8 (j >> 2) + 4 hh + (j & 3),  p = ((qb*2 + s)*2 + hh)*8 + j
    const int r = i >> 7, p = i & 127;
    const int j = p & 7, hh = (p >> 3) & 1, st = (p >> 4) & 1, qb = p >> 5;
    const int k = 32 * qb + 16 * st + 8 * (j >> 2) + 4 * hh + (j & 3);
    const float w1 = a.W1[r * 128 + k], w2 = a.W2[r * 128 + k];
    const half_t h1 = (half_t)w1, h2 = (half_t)w2;
    sW1h[r * LDH + p] = h1; sW1l[r * LDH + p] = (half_t)(w1 - (float)h1);
    sW2h[r * LDH + p] = h2; sW2l[r * LDH + p] = (half_t)(w2 - (float)h2);
  }
  for (int i = tid; i < 64 * 32; i += 512) {
    int y = i >> 5, k = i & 31;
    sBt[y * LDB + k] = a.B[k * 64 + y];
  }
  if (tid < 128) { sb1[tid] = a.b1[tid]; sb2[tid] = a.b2[tid]; sw3[tid] = a.w3[tid]; }
  __syncthreads();

  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int S = a.S;
  const float b3 = a.b3[0];
  const long long ntiles = (a.npts + 31) / 32;
  const float two_pi = 6.2831855f;      // float32(2*np.pi), axisnetworks.py:89

  for (long long tile = (long long)blockIdx.x * 8 + wave; tile < ntiles; tile += (long long)gridDim.x * 8) {
    long long pt = tile * 32 + l31;
    const bool valid = pt < a.npts;
    if (!valid) pt = a.npts - 1;
    float cx, cy, cz;
    if (a.coords) {
      cx = a.coords[pt * 3 + 0]; cy = a.coords[pt * 3 + 1]; cz = a.coords[pt * 3 + 2];
    } else {   // dense grid, x slowest (visualize.py:84-86, indexing='ij')
      const int res = a.res;
      int k = (int)(pt % res);
      long long ij = pt / res;
      int j = (int)(ij % res), i = (int)(ij / res);
      cx = a.lin[i]; cy = a.lin[j]; cz = a.lin[k];
    }
    // ---- bilinear features: this lane's 16 channels {8g+4h+e}, summed over the three planes ----
    f32x4 f[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) f[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      // plane 0: (x,y)  plane 1: (y,z)  plane 2: (x,z); first coord indexes W, second H (axisnetworks.py:549-551)
      const float u = (p == 1) ? cy : cx;
      const float v = (p == 0) ? cy : cz;
      const float ix = ((u + 1.f) / 2.f) * (float)(S - 1);
      const float iy = ((v + 1.f) / 2.f) * (float)(S - 1);
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy;
      const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
      const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
      const float* pl = a.planes + (long long)p * S * S * 32;
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {     // nw, ne, sw, se
        const int xx = x0 + (q & 1), yy = y0 + (q >> 1);
        const float w = ((q & 1) ? wx1 : wx0) * ((q >> 1) ? wy1 : wy0);
        if (xx >= 0 && xx < S && yy >= 0 && yy < S) {
          const float* t = pl + ((long long)yy * S + xx) * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] += w * *reinterpret_cast<const f32x4*>(t + 8 * g);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) f[g] += acc[g];
    }
    // ---- layer 0: y = f @ B  (64 outputs = 2 blocks of 32) ----
    f32x16 d0[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int r = 0; r < 16; ++r) d0[q][r] = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 av = *reinterpret_cast<const f32x4*>(sBt + (32 * q + l31) * LDB + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) d0[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], f[g][e], d0[q], 0, 0, 0);
      }
    }
    // ---- Fourier features: inputs of layer 1: blocks 0,1 = sin, blocks 2,3 = cos ----
    f32x16 x1[4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // the reference's fp32 angle 2*pi*y (axisnetworks.py:89), then sin / cos on the hardware unit, which takes its
        // argument in TURNS: ang / 2pi, reduced exactly by v_fract (period 1), v_sin_f32 / v_cos_f32 (4 instructions in
        // place of a 35-instruction Cody-Waite + polynomial evaluation; the decode kernel is VALU-bound once the big layers
        // run on the fp16 pipe).  Absolute error ~3e-6 in the features: two orders below the 1e-4 logit tolerance.
        const float ang = two_pi * d0[q][r];
        const float turns = ang * 0.15915494309189535f;
        const float fr = __builtin_amdgcn_fractf(turns);
        const float sv = __builtin_amdgcn_sinf(fr), cv = __builtin_amdgcn_cosf(fr);
        x1[q][r] = sv;
        x1[2 + q][r] = cv;
      }
    // ---- layers 1 and 2: 128 -> 128 on the fp16 pipe with hi/lo-split operands, bias + ReLU on the accumulators ----
    half8 xh[4][2], xl[4][2];
    auto split = [&](const f32x16 (&x)[4]) {
#pragma unroll
      for (int qb = 0; qb < 4; ++qb)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v = x[qb][8 * st + j];
            const half_t hi = (half_t)v;
            xh[qb][st][j] = hi;
            xl[qb][st][j] = (half_t)(v - (float)hi);
          }
    };
    auto layer = [&](const half_t* wh, const half_t* wl, int q) {
      f32x16 d;
#pragma unroll
      for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
      for (int qb = 0; qb < 4; ++qb)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const int off = (32 * q + l31) * LDH + ((qb * 2 + st) * 2 + h) * 8;
          const half8 ah = *reinterpret_cast<const half8*>(wh + off);
          const half8 al = *reinterpret_cast<const half8*>(wl + off);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[qb][st], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[qb][st], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[qb][st], d, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);      // keep the weight reads next to their MFMAs (hoisted, they spill)
        }
      return d;
    };
    split(x1);
    f32x16 x2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      x2[q] = layer(sW1h, sW1l, q);
#pragma unroll
      for (int r = 0; r < 16; ++r) x2[q][r] = fmaxf(x2[q][r] + sb1[32 * q + kmap(r, h)], 0.f);
    }
    split(x2);
    float partial = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x16 d = layer(sW2h, sW2l, q);
      // ---- output layer folded in: logit = w3 . relu(h2) + b3 ----
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int nrn = 32 * q + kmap(r, h);
        partial += sw3[nrn] * fmaxf(d[r] + sb2[nrn], 0.f);
      }
    }
    partial += __shfl_xor(partial, 32);
    if (h == 0 && valid) a.out[pt] = partial + b3;
  }
}

int triplane_decode_launch(const DecodeArgs& a, hipStream_t s) {
  ISHAP_REQUIRE(a.npts > 0, "no points");
  ISHAP_REQUIRE(a.coords != nullptr || (a.lin != nullptr && a.res > 0), "either coords or a grid axis");
  const size_t smem = (size_t)4 * 128 * LDH * sizeof(half_t) + (size_t)(64 * LDB + 3 * 128) * sizeof(float);
  ISHAP_TRY(ishap_set_max_lds((const void*)triplane_decode_kernel, (int)smem));
  long long ntiles = (a.npts + 31) / 32;
  int blocks = (int)std::min<long long>((ntiles + 7) / 8, 256);
  hipLaunchKernelGGL(triplane_decode_kernel, dim3(blocks), dim3(512), smem, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
