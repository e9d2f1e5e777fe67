// Small HBM-bound kernels around the matrix work: layout changes at the C-ABI boundary
// (torch-visible tensors are NCHW like the reference's, internal maps are NHWC fp16),
// skip-connection concat (unet.py:663), attention glue (head transposes + fp32 softmax,
// unet.py:347-353), timestep embedding + embedding MLPs (nn.py:102-120, unet.py:470-475,199-205).
#include "misc.h"

// --------------------------- NCHW fp32 -> NHWC fp16 (channel-padded) ---------------------------
// tile transpose through LDS: reads coalesced along HW, writes coalesced along C
// Blocks with blockIdx.y >= ytiles do not convert: together they zero `zero_n16` 16-byte words at `zero` (the UNet forward's
// statistics arena, which used to be a hipMemsetAsync launch of its own in front of this kernel -- one launch and boundary fewer per step).
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, half_t* __restrict__ dst,
                                                           int C, int HW, int Cpad, int ytiles, uint4* __restrict__ zero, long long zero_n16) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  if ((int)blockIdx.y >= ytiles) {
    const long long nb = (long long)gridDim.x * (gridDim.y - ytiles) * gridDim.z;
    const long long b = blockIdx.x + (long long)gridDim.x * ((blockIdx.y - ytiles) + (long long)(gridDim.y - ytiles) * blockIdx.z);
    for (long long i = b * 256 + threadIdx.x; i < zero_n16; i += nb * 256) zero[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, p = p0 + tx;
    tile[j][tx] = (c < C && p < HW) ? src[((long long)n * C + c) * HW + p] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int p = p0 + j, c = c0 + tx;
    if (p < HW && c < Cpad) dst[((long long)n * HW + p) * Cpad + c] = (half_t)tile[tx][j];
  }
}
int nchw_f32_to_nhwc_f16(const float* src, half_t* dst, int N, int C, int HW, int Cpad, hipStream_t s, void* zero, size_t zero_bytes) {
  const int ytiles = ceil_div(Cpad, 32);
  ISHAP_REQUIRE(zero_bytes % 16 == 0 && (reinterpret_cast<uintptr_t>(zero) & 15) == 0, "zeroed range: 16-byte granules");
  // one zeroing block per ~16 KB (4 stores per thread), at most one extra row of tiles per converting row
  const long long n16 = (long long)(zero_bytes / 16);
  const int per_row = ceil_div(HW, 32) * N;
  int zrows = zero && n16 ? (int)((n16 + 1024LL * per_row - 1) / (1024LL * per_row)) : 0;
  if (zrows > ytiles) zrows = ytiles;
  dim3 g(ceil_div(HW, 32), ytiles + zrows, N);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, g, dim3(256), 0, s, src, dst, C, HW, Cpad, ytiles, reinterpret_cast<uint4*>(zero), zrows ? n16 : 0LL);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// --------------------------- NHWC fp16 -> NCHW (fp16 or fp32) ---------------------------
template <typename OT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const half_t* __restrict__ src, OT* __restrict__ dst, int C,
                                                           int HW, int ld) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int p = p0 + j, c = c0 + tx;
    tile[j][tx] = (c < C && p < HW) ? (float)src[((long long)n * HW + p) * ld + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, p = p0 + tx;
    if (c < C && p < HW) dst[((long long)n * C + c) * HW + p] = (OT)tile[tx][j];
  }
}
int nhwc_f16_to_nchw(const half_t* src, void* dst, int out_f32, int N, int C, int HW, int ld, hipStream_t s) {
  dim3 g(ceil_div(HW, 32), ceil_div(C, 32), N);
  if (out_f32) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, g, dim3(256), 0, s, src, (float*)dst, C, HW, ld);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<half_t>, g, dim3(256), 0, s, src, (half_t*)dst, C, HW, ld);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// NCHW (fp16 or fp32) -> NHWC fp16 with row stride ld (used for cotangents entering the backward pass)
template <typename IT>
__global__ __launch_bounds__(256) void nchw_any_to_nhwc_kernel(const IT* __restrict__ src, half_t* __restrict__ dst,
                                                               int C, int HW, int ld, float mul) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, p = p0 + tx;
    tile[j][tx] = (c < C && p < HW) ? (float)src[((long long)n * C + c) * HW + p] * mul : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int p = p0 + j, c = c0 + tx;
    if (p < HW && c < ld) dst[((long long)n * HW + p) * ld + c] = (half_t)tile[tx][j];
  }
}
int nchw_to_nhwc_f16_scaled(const void* src, int src_f32, half_t* dst, int N, int C, int HW, int ld, float mul,
                            hipStream_t s) {
  dim3 g(ceil_div(HW, 32), ceil_div(ld, 32), N);
  if (src_f32) hipLaunchKernelGGL(nchw_any_to_nhwc_kernel<float>, g, dim3(256), 0, s, (const float*)src, dst, C, HW, ld, mul);
  else hipLaunchKernelGGL(nchw_any_to_nhwc_kernel<half_t>, g, dim3(256), 0, s, (const half_t*)src, dst, C, HW, ld, mul);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// --------------------------- channel concat / split (16-byte vectors) ---------------------------
__global__ void concat2_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, half_t* __restrict__ o,
                               long long M, int Ca, int Cb, const long long* __restrict__ sa,
                               const long long* __restrict__ sb, long long* __restrict__ so, int N) {
  if (so && blockIdx.x == 0) {           // per-channel GroupNorm sums of the concatenation = concatenation of the sums
    const int Cc = Ca + Cb;
    for (int i = threadIdx.x; i < N * Cc * 2; i += blockDim.x) {
      const int k = i & 1, c = (i >> 1) % Cc, n = (i >> 1) / Cc;
      so[i] = c < Ca ? sa[(n * Ca + c) * 2 + k] : sb[(n * Cb + (c - Ca)) * 2 + k];
    }
  }
  const int CV = (Ca + Cb) >> 3, CVa = Ca >> 3;
  const long long total = M * CV;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    long long m = i / CV;
    int cv = (int)(i % CV);
    half8 v = cv < CVa ? *reinterpret_cast<const half8*>(a + m * Ca + cv * 8)
                       : *reinterpret_cast<const half8*>(b + m * Cb + (cv - CVa) * 8);
    *reinterpret_cast<half8*>(o + m * (Ca + Cb) + cv * 8) = v;
  }
}
int concat2(const half_t* a, const half_t* b, half_t* o, long long M, int Ca, int Cb, hipStream_t s, const long long* sa,
            const long long* sb, long long* so, int N) {
  long long total = M * ((Ca + Cb) / 8);
  int blocks = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(concat2_kernel, dim3(blocks), dim3(256), 0, s, a, b, o, M, Ca, Cb, sa, sb, so, N);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// o = a + b (fp16, fp32 add)
__global__ void add_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, half_t* __restrict__ o, long long nvec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    half8 x = reinterpret_cast<const half8*>(a)[i], y = reinterpret_cast<const half8*>(b)[i], r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (half_t)((float)x[k] + (float)y[k]);
    reinterpret_cast<half8*>(o)[i] = r;
  }
}
int add_f16(const half_t* a, const half_t* b, half_t* o, long long n, hipStream_t s) {
  long long nvec = n / 8;
  int blocks = (int)std::min<long long>((nvec + 255) / 256, 4096);
  hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, s, a, b, o, nvec);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}


// --------------------------- timestep embedding + GEMV ---------------------------
// nn.py:102-120: [cos(t*f) | sin(t*f)], f_j = exp(-ln(1e4) * j / half)
__global__ void timestep_embedding_kernel(TsArg t, float* __restrict__ out, int dim) {
  const int n = blockIdx.x;
  const int half = dim / 2;
  for (int j = threadIdx.x; j < half; j += blockDim.x) {
    float f = expf(-logf(10000.f) * (float)j / (float)half);
    float a = t.t[n] * f;
    out[n * dim + j] = cosf(a);
    out[n * dim + half + j] = sinf(a);
  }
  if ((dim & 1) && threadIdx.x == 0) out[n * dim + dim - 1] = 0.f;
}
int timestep_embedding(const TsArg& t, float* out, int N, int dim, hipStream_t s) {
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(N), dim3(128), 0, s, t, out, dim);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// out[n][j] = sum_k W[j][k] * f(in[n][k]) + b[j], f = SiLU or identity; one wave per output row, fp32
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                   const float* __restrict__ in, float* __restrict__ out, int rows, int K,
                                                   int N, int silu_in) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* w = W + (long long)row * K;
  for (int n = 0; n < N; ++n) {
    float acc = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      f32x4 wv = *reinterpret_cast<const f32x4*>(w + k);
      f32x4 xv = *reinterpret_cast<const f32x4*>(in + (long long)n * K + k);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = xv[i];
        if (silu_in) x = x / (1.f + __expf(-x));
        acc += wv[i] * x;
      }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[(long long)n * rows + row] = acc + b[row];
  }
}
int gemv_f32(const float* W, const float* b, const float* in, float* out, int rows, int K, int N, int silu_in,
             hipStream_t s) {
  ISHAP_REQUIRE(K % 4 == 0, "gemv K % 4");
  hipLaunchKernelGGL(gemv_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, W, b, in, out, rows, K, N, silu_in);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// --------------------------- weight packing (load time, not on the hot path) ---------------------------
// OIHW (fp32) -> Wt[Npad][taps*Cpad] fp16, k = tap*Cpad + c.  transpose_flip: build the input-gradient
// operand instead: rows = input channel, k = tap'*Opad + o with tap' = 8 - tap (spatially flipped).
__global__ void pack_conv_kernel(const float* __restrict__ w, half_t* __restrict__ dst, int O, int I, int taps,
                                 int rows_pad, int cpad, int transpose_flip, int dst_ld, int col_off) {
  const long long total = (long long)rows_pad * taps * cpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c = (int)(i % cpad);
    int tap = (int)((i / cpad) % taps);
    int r = (int)(i / ((long long)cpad * taps));
    float v = 0.f;
    if (!transpose_flip) {
      if (r < O && c < I) v = w[((long long)r * I + c) * taps + tap];
    } else {
      if (r < I && c < O) v = w[((long long)c * I + r) * taps + (taps - 1 - tap)];
    }
    dst[(long long)r * dst_ld + col_off + tap * cpad + c] = (half_t)v;
  }
}
int pack_conv_weight(const float* w, half_t* dst, int O, int I, int taps, int rows_pad, int cpad, int transpose_flip,
                     hipStream_t s, int dst_ld, int col_off) {
  long long total = (long long)rows_pad * taps * cpad;
  int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(pack_conv_kernel, dim3(blocks), dim3(256), 0, s, w, dst, O, I, taps, rows_pad, cpad, transpose_flip,
                     dst_ld ? dst_ld : taps * cpad, col_off);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// fp32 head weights -> [hi | hi | lo] fp16 blocks along the input-channel axis (pairs with gn_apply SPLIT's
// [hi | lo | hi] activations: a*w ~= a_hi*w_hi + a_lo*w_hi + a_hi*w_lo, fp32-grade product on the fp16 MFMA)
__global__ void pack_conv_split_kernel(const float* __restrict__ w, half_t* __restrict__ dst, int O, int I, int taps,
                                       int rows_pad) {
  const int cpad = 3 * I;
  const long long total = (long long)rows_pad * taps * cpad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int c3 = (int)(i % cpad);
    int tap = (int)((i / cpad) % taps);
    int r = (int)(i / ((long long)cpad * taps));
    int part = c3 / I, c = c3 % I;
    float v = 0.f;
    if (r < O) {
      float f = w[((long long)r * I + c) * taps + tap];
      half_t hi = (half_t)f;
      v = (part < 2) ? (float)hi : (f - (float)hi);
    }
    dst[i] = (half_t)v;
  }
}
int pack_conv_weight_split(const float* w, half_t* dst, int O, int I, int taps, int rows_pad, hipStream_t s) {
  long long total = (long long)rows_pad * taps * 3 * I;
  int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(pack_conv_split_kernel, dim3(blocks), dim3(256), 0, s, w, dst, O, I, taps, rows_pad);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ void round_f16_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = (float)(half_t)src[i];
}
int round_through_f16(const float* src, float* dst, long long n, hipStream_t s) {
  int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(round_f16_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// NHWC fp16 (row stride ld) -> NCHW fp32 times *mul_dev (gradient leaving the backward pass, loss scale removed)
__global__ __launch_bounds__(256) void nhwc_to_nchw_scaled_kernel(const half_t* __restrict__ src, float* __restrict__ dst,
                                                                  int C, int HW, int ld, const float* __restrict__ mul_dev) {
  __shared__ float tile[32][33];
  const float mul = mul_dev ? *mul_dev : 1.f;
  const int n = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int p = p0 + j, c = c0 + tx;
    tile[j][tx] = (c < C && p < HW) ? (float)src[((long long)n * HW + p) * ld + c] * mul : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, p = p0 + tx;
    if (c < C && p < HW) dst[((long long)n * C + c) * HW + p] = tile[tx][j];
  }
}
int nhwc_f16_to_nchw_f32_scaled(const half_t* src, float* dst, int N, int C, int HW, int ld, const float* mul_dev,
                                hipStream_t s) {
  dim3 g(ceil_div(HW, 32), ceil_div(C, 32), N);
  hipLaunchKernelGGL(nhwc_to_nchw_scaled_kernel, g, dim3(256), 0, s, src, dst, C, HW, ld, mul_dev);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
