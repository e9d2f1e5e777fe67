// Occupancy-supervised guidance for the real-shape path (reference: drag_utils.py:447-463 in train_triplane):
//   prediction = decoder(0, coord); loss = -BCEWithLogitsLoss()(prediction, gt); loss.backward()
// i.e. forward + backward of MultiTriplane (axisnetworks.py:546-562) on a batch of sampled points, with the
// gradient landing on the triplane features (which are the un-normalised pred_xstart of the diffusion step).
// 40 000 points x 70 kFLOP: tiny next to the UNet, so this is a plain fp32 VALU kernel -- one 128-thread block per
// point, thread t = hidden neuron t, vectors through LDS, weights L1/L2-resident; scatter by fp32 atomics.
#include "decode.h"

__global__ __launch_bounds__(128) void decode_points_bwd_kernel(DecBwdArgs a) {
  __shared__ float fp[3][32], f[32], y[64], ff[128], h1[128], h2[128], dv[128], dv2[128], red[2];
  __shared__ int tex[3][4];
  __shared__ float texw[3][4];
  const int t = threadIdx.x;
  const int S = a.S;
  const float two_pi = 6.2831855f;
  for (long long pt = blockIdx.x; pt < a.npts; pt += gridDim.x) {
    const float cx = a.coords[pt * 3], cy = a.coords[pt * 3 + 1], cz = a.coords[pt * 3 + 2];
    // ---- bilinear taps of the three planes (xy, yz, xz; axisnetworks.py:549-551) ----
    if (t < 12) {
      const int p = t >> 2, q = t & 3;
      const float u = (p == 1) ? cy : cx, v = (p == 0) ? cy : cz;
      const float ix = ((u + 1.f) / 2.f) * (float)(S - 1), iy = ((v + 1.f) / 2.f) * (float)(S - 1);
      const float fx = floorf(ix), fy = floorf(iy);
      const int xx = (int)fx + (q & 1), yy = (int)fy + (q >> 1);
      const float wx = (q & 1) ? ix - fx : (fx + 1.f) - ix, wy = (q >> 1) ? iy - fy : (fy + 1.f) - iy;
      const bool ok = xx >= 0 && xx < S && yy >= 0 && yy < S;
      tex[p][q] = ok ? (p * S + yy) * S + xx : -1;
      texw[p][q] = wx * wy;
    }
    __syncthreads();
    if (t < 96) {
      const int p = t >> 5, c = t & 31;
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (tex[p][q] >= 0) acc += texw[p][q] * a.planes[(long long)tex[p][q] * 32 + c];
      fp[p][c] = acc;
    }
    __syncthreads();
    if (t < 32) f[t] = (fp[0][t] + fp[1][t]) + fp[2][t];
    __syncthreads();
    if (t < 64) {
      float acc = 0.f;
      for (int k = 0; k < 32; ++k) acc += f[k] * a.B[k * 64 + t];
      y[t] = acc;
    }
    __syncthreads();
    {
      const float ang = two_pi * y[t & 63];
      ff[t] = t < 64 ? sinf(ang) : cosf(ang);
    }
    __syncthreads();
    {
      float acc = a.b1[t];
      for (int k = 0; k < 128; ++k) acc += a.W1T[k * 128 + t] * ff[k];
      h1[t] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    {
      float acc = a.b2[t];
      for (int k = 0; k < 128; ++k) acc += a.W2T[k * 128 + t] * h1[k];
      h2[t] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    // ---- logit, BCE-with-logits, d(-BCE_mean)/dz ----
    float part = a.w3[t] * h2[t];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
    if ((t & 63) == 0) red[t >> 6] = part;
    __syncthreads();
    const float z = red[0] + red[1] + a.b3[0];
    const float gt = a.gt[pt];
    const float sig = 1.f / (1.f + expf(-z));
    const float dz = -(sig - gt) / (float)a.npts;
    if (t == 0) {
      const float bce = fmaxf(z, 0.f) - z * gt + log1pf(expf(-fabsf(z)));
      atomicAdd(a.loss, -bce / (float)a.npts);
      if (a.logits) a.logits[pt] = z;
    }
    // ---- backward through the MLP ----
    dv[t] = h2[t] > 0.f ? a.w3[t] * dz : 0.f;                     // d h2 (pre-activation)
    __syncthreads();
    {
      float acc = 0.f;
      for (int j = 0; j < 128; ++j) acc += a.W2[j * 128 + t] * dv[j];
      dv2[t] = h1[t] > 0.f ? acc : 0.f;                            // d h1 (pre-activation)
    }
    __syncthreads();
    {
      float acc = 0.f;
      for (int j = 0; j < 128; ++j) acc += a.W1[j * 128 + t] * dv2[j];
      dv[t] = acc;                                                  // d ff
    }
    __syncthreads();
    if (t < 64) {
      const float ang = two_pi * y[t];
      y[t] = two_pi * (dv[t] * cosf(ang) - dv[64 + t] * sinf(ang));   // d y
    }
    __syncthreads();
    if (t < 32) {
      float acc = 0.f;
      for (int j = 0; j < 64; ++j) acc += a.B[t * 64 + j] * y[j];
      f[t] = acc;                                                   // d feature
    }
    __syncthreads();
    if (t < 96) {
      const int p = t >> 5, c = t & 31;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (tex[p][q] >= 0) atomicAdd(a.dplanes + (long long)tex[p][q] * 32 + c, texw[p][q] * f[c]);
    }
    __syncthreads();
  }
}

int decode_points_bwd_launch(const DecBwdArgs& a, hipStream_t s) {
  ISHAP_REQUIRE(a.npts > 0, "no points");
  ISHAP_CHECK_HIP(hipMemsetAsync(a.dplanes, 0, (size_t)3 * a.S * a.S * 32 * sizeof(float), s));
  ISHAP_CHECK_HIP(hipMemsetAsync(a.loss, 0, sizeof(float), s));
  int blocks = (int)std::min<long long>(a.npts, 8192);
  hipLaunchKernelGGL(decode_points_bwd_kernel, dim3(blocks), dim3(128), 0, s, a);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// d loss / d planes  ->  the two routes back to the latent x of this step (drag_utils.py:448-450 and
// gaussian_diffusion.py:312-314,333-338):  pred_xstart = clamp(sr*x - srm1*eps(x), -1, 1), planes = pred_xstart*range + middle
//   g_direct = sr   * gx0                 (explicit dependence on x)
//   cot_out  = -srm1 * gx0 on the eps half of the model output, 0 on the variance half (to the UNet backward)
// with gx0 = dplanes * range * 1[-1 <= x0_unclamped <= 1].
__global__ void x0_grad_kernel(const float* __restrict__ dplanes, const float* __restrict__ rng, const float* __restrict__ x,
                               const float* __restrict__ model_out, float sr, float srm1, int clip, int S,
                               float* __restrict__ g_direct, float* __restrict__ cot_out) {
  __shared__ float tile[32][33];
  const int p = blockIdx.z, pix0 = blockIdx.x * 32, SS = S * S;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8)        // j = pixel, tx = channel
    tile[j][tx] = dplanes[((long long)p * SS + pix0 + j) * 32 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {      // j = channel, tx = pixel
    const int c = p * 32 + j;
    const long long o = (long long)c * SS + pix0 + tx;
    const float x0u = sr * x[o] - srm1 * model_out[o];
    float g = tile[tx][j] * (rng ? rng[c] : 1.f);
    if (clip && (x0u < -1.f || x0u > 1.f)) g = 0.f;
    g_direct[o] = sr * g;
    cot_out[o] = -srm1 * g;
    cot_out[(long long)96 * SS + o] = 0.f;
  }
}
int x0_grad_launch(const float* dplanes, const float* rng, const float* x, const float* model_out, float sr, float srm1,
                   int clip, int S, float* g_direct, float* cot_out, hipStream_t s) {
  hipLaunchKernelGGL(x0_grad_kernel, dim3(S * S / 32, 1, 3), dim3(256), 0, s, dplanes, rng, x, model_out, sr, srm1, clip, S,
                     g_direct, cot_out);
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
