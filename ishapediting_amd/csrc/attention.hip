// Fused attention for the UNet's AttentionBlocks (reference: guided_diffusion/unet.py:337-354, QKVAttentionLegacy:
// per head  w = softmax_fp32((q*s)^T (k*s)), s = d^-1/4;  a = w v;  tokens T <= 1024, head width d = 64).
// Forward and backward are flash-style: the T x T score matrix never leaves the CU.
//   forward : one workgroup = 64 queries of one (image, head); online softmax over 64-key tiles; writes a and
//             lse = m + log(sum) per query.
//   backward: ONE launch holding two kinds of workgroup -- dQ (same tiling as the forward) and dK/dV (one workgroup =
//             64 keys, loops over query tiles); P is recomputed from q, k and the saved lse; D_q = sum_d dA*A is
//             computed by each side for the queries it handles, so the two sides share nothing.
// All products run on v_mfma_f32_16x16x32_f16.  Operand convention used throughout: for a row-major matrix R[idx][k]
// (k contiguous) lane l loads R[i0 + (l&15)][k0 + 8*(l>>4) .. +7] -- this is the A fragment when idx is the output row
// and the B fragment when idx is the output column.  The accumulator holds C[row=(l>>4)*4+r][col=l&15].
// Probabilities / dS never leave the registers: the score products are oriented so that the accumulator (4 consecutive
// rows of one column per lane) is already a piece of the next product's B operand; operands contracted over their row
// index (V, K, Q, dA as "transposed" A operands) are read from the row-major LDS tiles with ds_read_b64_tr_b16.
// qkv layout (legacy order): token row of 3C halfs, head h at [h*3d, (h+1)*3d): q | k | v.
#include "attention.h"

__device__ __forceinline__ half8 ld_frag(const half_t* tile, int stride, int i0, int k0, int lane) {
  return *reinterpret_cast<const half8*>(tile + (i0 + (lane & 15)) * stride + k0 + 8 * (lane >> 4));
}

// A 64 x D tile (row-major, global row stride ld) travels global -> registers -> LDS in two halves so the loads of
// the NEXT tile are in flight while the current one is being multiplied (software prefetch; 64*D/8/256 chunks a thread).
template <int D>
struct TileRegs { half8 v[(64 * (D / 8) + 255) / 256]; };

template <int D>
__device__ __forceinline__ void load_tile(const half_t* __restrict__ src, int ld, TileRegs<D>& t, int tid) {
  constexpr int CPR = D / 8;
#pragma unroll
  for (int i = 0; i < (64 * CPR + 255) / 256; ++i) {
    const int c = tid + i * 256;
    if (c < 64 * CPR) t.v[i] = *reinterpret_cast<const half8*>(src + (long long)(c / CPR) * ld + (c % CPR) * 8);
  }
}
// registers -> LDS, row-major (16-byte writes)
template <int D, int RSTRIDE = D + 8>
__device__ __forceinline__ void store_tile(const TileRegs<D>& t, half_t* rows, int tid) {
  constexpr int CPR = D / 8;
#pragma unroll
  for (int i = 0; i < (64 * CPR + 255) / 256; ++i) {
    const int c = tid + i * 256;
    if (c < 64 * CPR) {
      const int r = c / CPR, ch = c % CPR;
      *reinterpret_cast<half8*>(rows + r * RSTRIDE + ch * 8) = t.v[i];
    }
  }
}

// gfx950 transposed LDS read: within each 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a
// 4-row x 16-column block of halfs; lane i receives column i of the 4 rows.  On a row-major [key][d] tile this hands
// lane i four consecutive keys of column d0+i -- a piece of a V^T fragment without a transposed copy.  EXEC must be full.
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ half4 lds_read_tr4(const half_t* p) {
  return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}

// XCD-aware workgroup -> work mapping.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest), and every
// XCD has an L2 of its own: with the plain (tile, head) grid the 16 query tiles of a head at T = 1024 sat on all 8 XCDs and
// each L2 fetched the head's K and V (and Q / dA in the backward) for itself -- 17.9 MB fetched per forward launch for 3.1 MB of
// qkv (profiles/round4_pmc_FETCH_SIZE.txt).  Work item w of `total` goes to the workgroup with linear id (w % P) * 8 + w / P
// ... i.e. workgroup `lin` takes item (lin % 8) * P + lin / 8 with P = total / 8: XCD c owns the contiguous items [c P, (c+1) P),
// and the items are ordered so that everything that reads one head's tensors is contiguous.  ISHAP_ATTN_XCD=0: identity.
__device__ __forceinline__ int attn_xcd_item(int lin, int total, int on) {
  if (!on || (total & 7) != 0) return lin;
  const int P = total >> 3;
  return (lin & 7) * P + (lin >> 3);
}

// ------------------------------------------------------------------------------------------------------------
// Forward.  Scores are produced TRANSPOSED (S^T = K Q^T: accumulator row = key, column = query), so a lane owns ONE
// query (column l&15) and 16 of the tile's 64 keys: the running max / sum are per-lane scalars finished with two
// cross-lane steps (the 4 lanes sharing a column), and the probabilities already sit in the registers in the shape
// of a B operand (8 keys of one query per lane) for O^T = V^T P^T -- they never visit LDS.  The MFMA contraction
// slots (group g = l>>4, j < 8) stand for keys {sub_a*16 + 4g + j, sub_b*16 + 4g + j-4}; the V^T fragment is read
// with the same assignment by two transposed LDS reads of the row-major V tile (lds_read_tr4).
// Two TEAMS of four waves share a workgroup's 64 queries: team t takes the key tiles t, t+2, t+4 ... (its own K / V
// staging buffers, the same barriers) and the two online-softmax states are merged through LDS at the end.  With one
// four-wave workgroup per CU (T = 1024, 8 heads: 128 workgroups) every global -> LDS -> MFMA round trip of a tile was
// exposed; two waves per SIMD on interleaved tiles hide them (T = 1024: 19.9 -> ~11 us).
// TEAMS = 2 or 4 (round 3: four teams = 1024 threads when the sequence has >= 8 key tiles, so that a team walks T / 256 of them)
template <int D, int TEAMS>
__global__ __launch_bounds__(256 * TEAMS) void attn_fwd_kernel(const void* h_qkv, void* h_out, float* h_lse, int h_T, int h_C, AttnArgs a0) {
  // leading scalar parameters are preloaded into SGPRs at dispatch (common.h, IgemmHot); the block `a0` arrives by s_load
  AttnArgs a = a0;
  a.qkv = reinterpret_cast<const half_t*>(h_qkv); a.out = reinterpret_cast<half_t*>(h_out); a.lse = h_lse; a.T = h_T; a.C = h_C;
  constexpr int RS = D + 8, KK = D / 32, DS = D / 16;
  constexpr int VS = D + 16;                 // V row stride: 160-byte rows keep the transposed reads conflict-free
  constexpr int MRG = DS * 4 + 2;            // floats of one lane's online-softmax state
  // dynamic LDS: [TEAMS] K tiles, [TEAMS] V tiles; after the key loop the same memory holds the states of teams 1 .. TEAMS-1
  extern __shared__ __attribute__((aligned(16))) char attn_smem[];
  half_t* const sK2 = reinterpret_cast<half_t*>(attn_smem);
  half_t* const sV2 = sK2 + TEAMS * 64 * RS;
  float* const mrg = reinterpret_cast<float*>(attn_smem);
  static_assert((TEAMS - 1) * 256 * MRG * 4 <= TEAMS * 64 * (RS + VS) * 2, "merge states alias the tile buffers");
  const int team = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  half_t* const sK = sK2 + team * 64 * RS;
  half_t* const sV = sV2 + team * 64 * VS;
  // item = tile + ntile * (head + heads * image)
  const int item = attn_xcd_item(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z, a0.xcd_map);
  const int q0 = (item % (int)gridDim.x) * 64, h = (item / (int)gridDim.x) % (int)gridDim.y, n = item / (int)(gridDim.x * gridDim.y);
  const int ld = 3 * a.C;
  const int g = lane >> 4, col = lane & 15;
  const half_t* base = a.qkv + (long long)n * a.T * ld + h * 3 * D;
  half8 qf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
    qf[kk] = *reinterpret_cast<const half8*>(base + (long long)(q0 + wave * 16 + col) * ld + kk * 32 + 8 * g);
  float m = -1e30f, lsum = 0.f;
  f32x4 ot[DS];                              // O^T: row d = i*16 + 4g + r, column = this lane's query
#pragma unroll
  for (int i = 0; i < DS; ++i) ot[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  TileRegs<D> rk, rv;
  const int ntile = a.T / 64, niter = (ntile + TEAMS - 1) / TEAMS;       // the same barrier count for every team
  if (team < ntile) {
    load_tile<D>(base + (long long)(team * 64) * ld + D, ld, rk, tid);
    load_tile<D>(base + (long long)(team * 64) * ld + 2 * D, ld, rv, tid);
  }
  for (int it = 0; it < niter; ++it) {
    const int kt = (TEAMS * it + team) * 64;
    const bool live = kt < a.T;
    __syncthreads();
    if (live) {
      store_tile<D>(rk, sK, tid);
      store_tile<D, VS>(rv, sV, tid);
      if (kt + TEAMS * 64 < a.T) {
        load_tile<D>(base + (long long)(kt + TEAMS * 64) * ld + D, ld, rk, tid);
        load_tile<D>(base + (long long)(kt + TEAMS * 64) * ld + 2 * D, ld, rv, tid);
      }
    }
    __syncthreads();
    if (!live) continue;
    f32x4 st[4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      st[sub] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        st[sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_frag(sK, RS, sub * 16, kk * 32, lane), qf[kk], st[sub], 0, 0, 0);
    }
    float mx = -1e30f;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int r = 0; r < 4; ++r) { st[sub][r] *= a.alpha; mx = fmaxf(mx, st[sub][r]); }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float corr = __expf(m - mn);
    float rs = 0.f;
    half8 pb[2];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(st[sub][r] - mn);
        rs += pv;
        pb[sub >> 1][(sub & 1) * 4 + r] = (half_t)pv;
      }
    rs += __shfl_xor(rs, 16);
    rs += __shfl_xor(rs, 32);
    lsum = lsum * corr + rs;
    m = mn;
#pragma unroll
    for (int i = 0; i < DS; ++i) {
      ot[i] *= corr;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const half_t* vb = sV + (4 * g + (col >> 2)) * VS + i * 16 + 4 * (col & 3);   // this lane's address in the 4-key block
        const half4 lo = lds_read_tr4(vb + (2 * pr) * 16 * VS);
        const half4 hi = lds_read_tr4(vb + (2 * pr + 1) * 16 * VS);
        const half8 va = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        ot[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[pr], ot[i], 0, 0, 0);
      }
    }
  }
  // merge the teams' states (teams 1 .. -> LDS -> team 0, in team order): m' = max, sums rescaled by exp(m - m')
  __syncthreads();                           // every team is done with its tiles: the states go where the tiles were
  if (team > 0) {
    float* dst = mrg + ((team - 1) * 256 + tid) * MRG;
    dst[0] = m;
    dst[1] = lsum;
#pragma unroll
    for (int i = 0; i < DS; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[2 + i * 4 + r] = ot[i][r];
  }
  __syncthreads();
  if (team > 0) return;
#pragma unroll
  for (int t = 1; t < TEAMS; ++t) {
    const float* src = mrg + ((t - 1) * 256 + tid) * MRG;
    const float m1 = src[0], l1 = src[1];
    const float mn = fmaxf(m, m1);
    const float c0 = __expf(m - mn), c1 = __expf(m1 - mn);
    lsum = lsum * c0 + l1 * c1;
#pragma unroll
    for (int i = 0; i < DS; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[i][r] = ot[i][r] * c0 + src[2 + i * 4 + r] * c1;
    m = mn;
  }
  const int q = q0 + wave * 16 + col;
  const float inv = 1.f / lsum;
#pragma unroll
  for (int i = 0; i < DS; ++i) {
    const half4 o4 = {(half_t)(ot[i][0] * inv), (half_t)(ot[i][1] * inv), (half_t)(ot[i][2] * inv), (half_t)(ot[i][3] * inv)};
    *reinterpret_cast<half4*>(a.out + ((long long)n * a.T + q) * a.C + h * D + i * 16 + 4 * g) = o4;
  }
  if (g == 0) a.lse[((long long)n * a.heads + h) * a.T + q] = m + __logf(lsum);
}

// ------------------------------------------------------------------------------------------------------------
// dQ[q][:] = sum_key dS[q][key] K[key][:],  dS = alpha * P * (dP - D_q),  dP = dA V^T.   Also writes D_q.
// Same transposed arrangement as the forward: S^T = K Q^T and dP^T = V dA^T put one query in a lane (lse and D_q are
// per-lane scalars), dS^T is packed in registers as the B operand of dQ^T = K^T dS^T, and K^T comes from the row-major
// K tile through the transposed LDS read.  Nothing is staged transposed and nothing round-trips through LDS.
// team merge of the backward bodies: NA accumulator quads per thread, team 1 -> team 0 (fixed order: bitwise reproducible).
// mrg: NA * 256 quads.  Returns false for the team that is done (it has passed every barrier of the merge).
// (A four-team form -- 1 024 threads, the kernel held to 128 registers -- was built in round 4, measured slower in situ,
// 0.1772 -> 0.1779-0.1782 s/shape, and removed in round 6.)
template <int NA, int TEAMS>
__device__ __forceinline__ bool attn_bwd_merge(f32x4 (&acc)[NA], int team, int tid, float* mrg) {
  static_assert(TEAMS == 1 || TEAMS == 2, "one or two teams");
  if (blockDim.x == 256 || TEAMS == 1) return true;          // a single tile is launched with one team
  f32x4* m4 = reinterpret_cast<f32x4*>(mrg);   // [NA][256] 16-byte entries
  if (team == 1) {
#pragma unroll
    for (int i = 0; i < NA; ++i) m4[i * 256 + tid] = acc[i];
  }
  __syncthreads();
  if (team != 0) return false;
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] += m4[i * 256 + tid];
  return true;
}

template <int D, int TEAMS>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& a, int n, int tile, int h, int team, half_t* sK, half_t* sV, float* mrg) {
  constexpr int RS = D + 8, KK = D / 32, DS = D / 16;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int q0 = tile * 64;
  const int ld = 3 * a.C;
  const int g = lane >> 4, col = lane & 15;
  const half_t* base = a.qkv + (long long)n * a.T * ld + h * 3 * D;
  const long long bh = (long long)n * a.heads + h;
  const int q = q0 + wave * 16 + col;        // this lane's query
  half8 qf[KK], daf[KK];
  float Dq = 0.f;
  {
    const long long row = (long long)n * a.T + q;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      qf[kk] = *reinterpret_cast<const half8*>(base + (long long)q * ld + kk * 32 + 8 * g);
      daf[kk] = *reinterpret_cast<const half8*>(a.dout + row * a.C + h * D + kk * 32 + 8 * g);
      const half8 av = *reinterpret_cast<const half8*>(a.out + row * a.C + h * D + kk * 32 + 8 * g);
#pragma unroll
      for (int j = 0; j < 8; ++j) Dq += (float)daf[kk][j] * (float)av[j];
    }
  }
  Dq += __shfl_xor(Dq, 16);
  Dq += __shfl_xor(Dq, 32);                  // D of query `col`, in all four lanes of the column
  const float lse = a.lse[bh * a.T + q];
  f32x4 dqt[DS];                             // dQ^T: row d = i*16 + 4g + r, column = this lane's query
#pragma unroll
  for (int i = 0; i < DS; ++i) dqt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  TileRegs<D> rk, rv;
  const int ntile = a.T / 64, niter = (ntile + TEAMS - 1) / TEAMS;       // the teams take alternating key tiles, as in the forward
  if (team < ntile) {
    load_tile<D>(base + (long long)(team * 64) * ld + D, ld, rk, tid);
    load_tile<D>(base + (long long)(team * 64) * ld + 2 * D, ld, rv, tid);
  }
  for (int it = 0; it < niter; ++it) {
    const int kt = (TEAMS * it + team) * 64;
    const bool live = kt < a.T;
    __syncthreads();
    if (live) {
      store_tile<D>(rk, sK, tid);
      store_tile<D>(rv, sV, tid);
      if (kt + 64 * TEAMS < a.T) {
        load_tile<D>(base + (long long)(kt + 64 * TEAMS) * ld + D, ld, rk, tid);
        load_tile<D>(base + (long long)(kt + 64 * TEAMS) * ld + 2 * D, ld, rv, tid);
      }
    }
    __syncthreads();
    if (!live) continue;
    half8 sb[2];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        st = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_frag(sK, RS, sub * 16, kk * 32, lane), qf[kk], st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_frag(sV, RS, sub * 16, kk * 32, lane), daf[kk], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(st[r] * a.alpha - lse);
        sb[sub >> 1][(sub & 1) * 4 + r] = (half_t)(a.alpha * pv * (dp[r] - Dq));
      }
    }
#pragma unroll
    for (int i = 0; i < DS; ++i)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const half_t* kb = sK + (4 * g + (col >> 2)) * RS + i * 16 + 4 * (col & 3);
        const half4 lo = lds_read_tr4(kb + (2 * pr) * 16 * RS);
        const half4 hi = lds_read_tr4(kb + (2 * pr + 1) * 16 * RS);
        const half8 ka = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        dqt[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, sb[pr], dqt[i], 0, 0, 0);
      }
  }
  // the other teams hand their sums to team 0 (fixed order, attn_bwd_merge)
  if (!attn_bwd_merge<DS, TEAMS>(dqt, team, tid, mrg)) return;
#pragma unroll
  for (int i = 0; i < DS; ++i) {
    const half4 o4 = {(half_t)dqt[i][0], (half_t)dqt[i][1], (half_t)dqt[i][2], (half_t)dqt[i][3]};
    *reinterpret_cast<half4*>(a.dqkv + ((long long)n * a.T + q) * ld + h * 3 * D + i * 16 + 4 * g) = o4;
  }
}

// ------------------------------------------------------------------------------------------------------------
// One workgroup = 64 keys: dV[key][:] = sum_q P[q][key] dA[q][:],  dK[key][:] = sum_q dS[q][key] Q[q][:].
// Here the scores are produced UN-transposed (S = Q K^T, dP = dA V^T: accumulator row = query, column = key) so a lane
// owns one key and 16 queries of the tile: P and dS are packed in registers as the B operands of dV^T = dA^T P and
// dK^T = Q^T dS, whose A operands come from the row-major dA / Q tiles through the transposed LDS read.
template <int D, int TEAMS>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& a, int n, int tile, int h, int team, half_t* sQ, half_t* sA, float* sD,
                                                  float* mrg) {
  constexpr int RS = D + 8, KK = D / 32, DS = D / 16;
  constexpr int CPR = D / 8, NCH = (64 * CPR + 255) / 256;
  static_assert(CPR == 8 || CPR == 4, "row sums below reduce over CPR consecutive lanes");
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int k0 = tile * 64;
  const int ld = 3 * a.C;
  const int g = lane >> 4, col = lane & 15;
  const half_t* base = a.qkv + (long long)n * a.T * ld + h * 3 * D;
  const long long bh = (long long)n * a.heads + h;
  const int key = k0 + wave * 16 + col;      // this lane's key
  half8 kf[KK], vf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const half_t* row = base + (long long)key * ld + kk * 32 + 8 * g;
    kf[kk] = *reinterpret_cast<const half8*>(row + D);
    vf[kk] = *reinterpret_cast<const half8*>(row + 2 * D);
  }
  f32x4 acc[2 * DS];                         // dK^T (acc[i]) / dV^T (acc[DS + i]): row d = i*16 + 4g + r, column = this lane's key
#pragma unroll
  for (int i = 0; i < 2 * DS; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // D_q = sum_d dA[q][d] * A[q][d] of the tile's 64 queries is recomputed here from the dA tile this workgroup stages
  // anyway plus the matching rows of the forward output: the dQ and dK/dV passes then share no data and run as ONE
  // launch (256 instead of 128 workgroups at T = 1024, one launch floor instead of two).
  TileRegs<D> rq, ra, ro;
  float lse_n = 0.f;                         // lse of query `tid` of the prefetched tile (threads 0..63; staged in sD[64..127] with the tile)
  const half_t* abase = a.dout + (long long)n * a.T * a.C + h * D;
  const half_t* obase = a.out + (long long)n * a.T * a.C + h * D;
  const float* lsep = a.lse + bh * a.T + (tid & 63);
  const int ntile = a.T / 64, niter = (ntile + TEAMS - 1) / TEAMS;       // the teams take alternating query tiles
  if (team < ntile) {
    const long long q0r = (long long)team * 64;
    load_tile<D>(base + q0r * ld, ld, rq, tid);
    load_tile<D>(abase + q0r * a.C, a.C, ra, tid);
    load_tile<D>(obase + q0r * a.C, a.C, ro, tid);
    if (tid < 64) lse_n = lsep[q0r];
  }
  for (int it = 0; it < niter; ++it) {
    const int qt = (TEAMS * it + team) * 64;
    const bool live = qt < a.T;
    __syncthreads();
    if (live) {
    store_tile<D>(rq, sQ, tid);
    store_tile<D>(ra, sA, tid);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {          // chunk c = tid + 256 i holds 8 halfs of row c / CPR: CPR consecutive lanes share a row
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) dot += (float)ra.v[i][j] * (float)ro.v[i][j];
      dot += __shfl_xor(dot, 1);
      dot += __shfl_xor(dot, 2);
      if (CPR == 8) dot += __shfl_xor(dot, 4);
      const int c = tid + i * 256;
      if (c < 64 * CPR && c % CPR == 0) sD[c / CPR] = dot;
    }
    if (tid < 64) sD[64 + tid] = lse_n;
    }
    if (live && qt + 64 * TEAMS < a.T) {
      load_tile<D>(base + (long long)(qt + 64 * TEAMS) * ld, ld, rq, tid);
      load_tile<D>(abase + (long long)(qt + 64 * TEAMS) * a.C, a.C, ra, tid);
      load_tile<D>(obase + (long long)(qt + 64 * TEAMS) * a.C, a.C, ro, tid);
      if (tid < 64) lse_n = lsep[qt + 64 * TEAMS];
    }
    __syncthreads();
    if (!live) continue;
    half8 pb[2], sb[2];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_frag(sQ, RS, sub * 16, kk * 32, lane), kf[kk], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_frag(sA, RS, sub * 16, kk * 32, lane), vf[kk], dp, 0, 0, 0);
      }
      const f32x4 dq4 = *reinterpret_cast<const f32x4*>(sD + sub * 16 + 4 * g);         // D_q and lse of queries sub*16 + 4g + (0..3)
      const f32x4 ls4 = *reinterpret_cast<const f32x4*>(sD + 64 + sub * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(sc[r] * a.alpha - ls4[r]);
        pb[sub >> 1][(sub & 1) * 4 + r] = (half_t)pv;
        sb[sub >> 1][(sub & 1) * 4 + r] = (half_t)(a.alpha * pv * (dp[r] - dq4[r]));
      }
    }
#pragma unroll
    for (int i = 0; i < DS; ++i)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int off = (4 * g + (col >> 2)) * RS + i * 16 + 4 * (col & 3);
        const half4 alo = lds_read_tr4(sA + off + (2 * pr) * 16 * RS), ahi = lds_read_tr4(sA + off + (2 * pr + 1) * 16 * RS);
        const half4 qlo = lds_read_tr4(sQ + off + (2 * pr) * 16 * RS), qhi = lds_read_tr4(sQ + off + (2 * pr + 1) * 16 * RS);
        const half8 aa = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        const half8 qa = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
        acc[DS + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aa, pb[pr], acc[DS + i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa, sb[pr], acc[i], 0, 0, 0);
      }
  }
  if (!attn_bwd_merge<2 * DS, TEAMS>(acc, team, tid, mrg)) return;
  half_t* row = a.dqkv + ((long long)n * a.T + key) * ld + h * 3 * D;
#pragma unroll
  for (int i = 0; i < DS; ++i) {
    const half4 k4 = {(half_t)acc[i][0], (half_t)acc[i][1], (half_t)acc[i][2], (half_t)acc[i][3]};
    const half4 v4 = {(half_t)acc[DS + i][0], (half_t)acc[DS + i][1], (half_t)acc[DS + i][2], (half_t)acc[DS + i][3]};
    *reinterpret_cast<half4*>(row + D + i * 16 + 4 * g) = k4;
    *reinterpret_cast<half4*>(row + 2 * D + i * 16 + 4 * g) = v4;
  }
}

// role 0 = dQ of 64 queries, role 1 = dK/dV of 64 keys (independent of each other, see above); grid (tiles, heads, 2 * images)
// Each workgroup is two teams of four waves on alternating tiles (own staging buffers, shared barriers), merged at the end.
// dynamic LDS: [TEAMS] tile A, [TEAMS] tile B, [TEAMS][64 D_q | 64 lse], merge area (TEAMS / 2) * 2 * DS * 4 * 256 floats
template <int D, int TEAMS>
__global__ __launch_bounds__(256 * TEAMS) void attn_bwd_kernel(const void* h_qkv, const void* h_out, const void* h_dout, void* h_dqkv, float* h_lse,
                                                               float* h_Dbuf, int h_T, int h_C, AttnArgs a0) {
  AttnArgs a = a0;                           // preloaded leading parameters, as in attn_fwd_kernel
  a.qkv = reinterpret_cast<const half_t*>(h_qkv); a.out = const_cast<half_t*>(reinterpret_cast<const half_t*>(h_out));
  a.dout = reinterpret_cast<const half_t*>(h_dout); a.dqkv = reinterpret_cast<half_t*>(h_dqkv); a.lse = h_lse; a.Dbuf = h_Dbuf;
  a.T = h_T; a.C = h_C;
  constexpr int RS = D + 8;
  extern __shared__ __attribute__((aligned(16))) char attn_bwd_smem[];
  half_t* s0 = reinterpret_cast<half_t*>(attn_bwd_smem);
  half_t* s1 = s0 + TEAMS * 64 * RS;
  float* sD = reinterpret_cast<float*>(s1 + TEAMS * 64 * RS);
  float* mrg = sD + TEAMS * 128;
  // item = tile + ntile * (role + 2 * (head + heads * image)): both roles of a head next to each other (they read the same q, k, v, dA)
  const int ntile = gridDim.x, heads = gridDim.y;
  const int item = attn_xcd_item(blockIdx.x + ntile * (blockIdx.y + heads * blockIdx.z), ntile * heads * (int)gridDim.z, a0.xcd_map);
  const int tile = item % ntile, role = (item / ntile) & 1, h = (item / (2 * ntile)) % heads, n = item / (2 * ntile * heads);
  const int team = threadIdx.x >> 8;
  if (role == 0) attn_bwd_dq_body<D, TEAMS>(a, n, tile, h, team, s0 + team * 64 * RS, s1 + team * 64 * RS, mrg);
  else attn_bwd_dkv_body<D, TEAMS>(a, n, tile, h, team, s0 + team * 64 * RS, s1 + team * 64 * RS, sD + team * 128, mrg);
}

static int check_attn(const AttnArgs& a) {
  ISHAP_REQUIRE(a.T % 64 == 0 && a.T >= 64, "attention: tokens must be a multiple of 64");
  ISHAP_REQUIRE(a.C == a.heads * a.d && (a.d == 64 || a.d == 32), "attention: head width 64 or 32");
  return 0;
}

// teams of the forward launch: four once a team of two would walk >= 4 key tiles (T = 1024: 13.5 -> 12.5 us, round 3)
static int attn_fwd_teams(int ntile) { return ntile >= 8 ? 4 : 2; }

static int attn_xcd_on() {
  static const int on = [] { const char* e = getenv("ISHAP_ATTN_XCD"); return e ? atoi(e) : 1; }();
  return on;
}

int attn_forward_launch(const AttnArgs& a_in, hipStream_t s) {
  AttnArgs a = a_in;
  a.xcd_map = attn_xcd_on();
  ISHAP_TRY(check_attn(a));
  dim3 g(a.T / 64, a.heads, a.N);
  const int ntile = a.T / 64;
  const bool four = attn_fwd_teams(ntile) == 4;
#define ATTN_FWD(Dv, TM)                                                                                          \
  do {                                                                                                            \
    auto kern = attn_fwd_kernel<Dv, TM>;                                                                          \
    const int smem = TM * 64 * ((Dv + 8) + (Dv + 16)) * (int)sizeof(half_t);                                      \
    ISHAP_TRY(ishap_set_max_lds((const void*)kern, smem));                                                        \
    hipLaunchKernelGGL(kern, g, dim3(256 * TM), smem, s, (const void*)a.qkv, (void*)a.out, a.lse, a.T, a.C, a);    \
  } while (0)
  if (a.d == 64) { if (four) ATTN_FWD(64, 4); else ATTN_FWD(64, 2); }
  else { if (four) ATTN_FWD(32, 4); else ATTN_FWD(32, 2); }
#undef ATTN_FWD
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

int attn_backward_launch(const AttnArgs& a_in, hipStream_t s) {
  AttnArgs a = a_in;
  a.xcd_map = attn_xcd_on();
  ISHAP_TRY(check_attn(a));
  dim3 g(a.T / 64, a.heads, a.N * 2);
  // one tile: a second team would have nothing to do
#define ATTN_BWD(Dv, TM, THREADS)                                                                                              \
  do {                                                                                                                         \
    auto kern = attn_bwd_kernel<Dv, TM>;                                                                                       \
    const int smem = TM * 2 * 64 * (Dv + 8) * (int)sizeof(half_t) + TM * 128 * 4 + (TM / 2) * 2 * (Dv / 16) * 4 * 256 * 4;       \
    ISHAP_TRY(ishap_set_max_lds((const void*)kern, smem));                                                                     \
    hipLaunchKernelGGL(kern, g, dim3(THREADS), smem, s, (const void*)a.qkv, (const void*)a.out, (const void*)a.dout,           \
                       (void*)a.dqkv, a.lse, a.Dbuf, a.T, a.C, a);                                                             \
  } while (0)
  if (a.d == 64) ATTN_BWD(64, 2, a.T > 64 ? 512 : 256);
  else ATTN_BWD(32, 2, a.T > 64 ? 512 : 256);
#undef ATTN_BWD
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Round 5: qkv GEMM + attention + proj_out of an 8x8-map AttentionBlock in one launch (Attn8Args, attention.h).
// A head is 192 qkv channels = A8_PARTS workgroups of 16 channels each (so that the 6.3 + 2.1 MB of weights stream through
// ~200 CUs, not 16).  Part j of head h:
//   1. qkv channels [h*192 + 16 j, +16) of all 64 tokens: 4 waves split K = C, fragments straight from global memory
//      (igemm_skinny.hip's scheme), partial tiles meet in LDS in wave order; + bias, fp16, written to the qkv tensor with
//      agent-scope (write-through) 8-byte stores; every wave drains its stores, then ONE lane raises the part's flag;
//   2. waits for the 12 flags of its head (bounded spin: a give-up raises the status word and poisons the outputs), reads the
//      head's q | k | v (24 KB) with agent-scope loads -- MI355X_MICROARCH.md hand-off table, first row: every store and every
//      load of the handed-off bytes is sc1, the flag follows the storing waves' vmcnt(0) -- and computes the head's attention
//      exactly as attn_fwd_kernel does (S^T = K Q^T, per-lane softmax state, O^T = V^T P^T through the transposed LDS read);
//      every part does this redundantly (2 x 0.5 MFLOP): it is cheaper than a second exchange;
//   3. its share of proj_out's K slice of this head: out_h[64][n] = a_h[64][64] Wproj[n][h*64 ..]^T for its 16-channel tiles
//      (64 tiles of C = 1024 dealt over the 12 parts), fp32, into slices[h]: the consumer adds the 16 slices up.
// Part 0 also writes a_h and lse for the backward pass.  All parts of all heads must be resident together: <= 256 workgroups
// (the launcher), on a sequence that holds the device's rendezvous tenancy (the caller).  A sequence WITHOUT the tenancy runs the
// same kernel as two launches (Attn8Args::phases = 1, then 2): step 1 alone, then steps 2-3 without the wait -- no workgroup
// waits for another, and the values are bitwise those of the one-launch form (round 6).
constexpr int A8_PARTS = 12;

__device__ __forceinline__ half4 ld8_agent(const half_t* p) {
  const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return __builtin_bit_cast(half4, b);
}
__device__ __forceinline__ void st8_agent(half_t* p, half4 v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef A8_NW
#define A8_NW 4          // waves per workgroup: 4 / 8 / 16 -> 14.2 / 15.4 / 27 us per launch (C = 1024)
#endif
constexpr int A8_WAVES = A8_NW;
constexpr int A8_SMEM = (A8_WAVES * 4 * 64 * 16 > 64 * (72 + 80 + 72) * 2 ? A8_WAVES * 4 * 64 * 16 : 64 * (72 + 80 + 72) * 2) + 16;
constexpr int A8_WAVES_UNUSED = 0;     // phase 1: one 64-deep K-step per wave at C = 1024, every fragment load of the workgroup in flight at once
__global__ __launch_bounds__(A8_WAVES * 64) void attn8_fused_kernel(Attn8Args a) {
  constexpr int D = 64, RS = D + 8, VS = D + 16, KK = D / 32, DS = D / 16;
  // one LDS block: phase 1's partial tiles red[wave][sub-tile][lane] (64 KB), then the K | V | A tiles (28 KB)
  extern __shared__ __attribute__((aligned(16))) char a8_smem[];         // A8_SMEM bytes: the tiles, then one int
  int& s_ok = *reinterpret_cast<int*>(a8_smem + A8_SMEM - 16);
  f32x4 (*red)[4][64] = reinterpret_cast<f32x4 (*)[4][64]>(a8_smem);
  half_t* const tK = reinterpret_cast<half_t*>(a8_smem);                  // [64][RS]
  half_t* const tV = tK + 64 * RS;                                        // [64][VS]
  half_t* const tA = tV + 64 * VS;                                        // [64][RS]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, col = lane & 15;
  const int h = blockIdx.x / A8_PARTS, part = blockIdx.x - h * A8_PARTS, n = blockIdx.y;
  const int C = a.C, ld3 = 3 * C;
  const half_t* const xn = a.xn + (long long)n * 64 * C;
  half_t* const qkv = a.qkv + (long long)n * 64 * ld3;
  // ---- phase 3 operands first: this part's proj_out weight fragments (read-only data, their latency hides under phase 1) ----
  const int ntile = C / 16;                                               // 16-channel output tiles of proj_out
  const int tper = ntile / A8_PARTS, trem = ntile - tper * A8_PARTS;      // parts < trem take one tile more
  const int t0 = part * tper + (part < trem ? part : trem), tcnt = tper + (part < trem ? 1 : 0);
  constexpr int TMAX = 6;                                                 // C <= 1152 (launcher)
  half8 wp[TMAX][KK];
  const bool do1 = (a.phases & 1) != 0, do23 = (a.phases & 2) != 0;        // block-uniform (Attn8Args::phases)
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
    if (t < tcnt && do23) {
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        wp[t][kk] = *reinterpret_cast<const half8*>(a.wproj + (long long)((t0 + t) * 16 + col) * C + h * D + kk * 32 + 8 * g);
    }
  // ---- phase 1: 64 tokens x 16 qkv channels, K = C split over the four waves ----
  const int n0 = h * 3 * D + part * 16;
  if (do1) {
    const int ks = C / 64, per = (ks + A8_WAVES - 1) / A8_WAVES, s0 = wave * per, s1 = min(ks, s0 + per);
    const half_t* wrow = a.wqkv + (long long)(n0 + col) * C + 8 * g;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    struct Frag { half8 w[2]; half8 x[4][2]; };
    auto load = [&](int s, Frag& f) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) f.w[kk] = *reinterpret_cast<const half8*>(wrow + s * 64 + kk * 32);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          f.x[j][kk] = *reinterpret_cast<const half8*>(xn + (long long)(j * 16 + col) * C + s * 64 + 8 * g + kk * 32);
    };
    // every fragment load of a batch of up to four K-steps is issued before the first MFMA (a wave has at most four steps at
    // C = 1024 with four waves: one round trip instead of a chain of them -- the two-deep ring took 6.5 us here)
    Frag ring[4];
    const int s1x = s1;
    for (int sb = s0; sb < s1x; sb += 4) {
#pragma unroll
      for (int dd = 0; dd < 4; ++dd)
        if (sb + dd < s1x) load(sb + dd, ring[dd]);
#pragma unroll
      for (int dd = 0; dd < 4; ++dd)
        if (sb + dd < s1x) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ring[dd].w[kk], ring[dd].x[j][kk], acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[wave][j][lane] = acc[j];
    __syncthreads();
    // wave j < 4 finishes token sub-tile j: lane = (token j*16 + col, channels n0 + 4g .. +3); partials add in wave order
    if (wave < 4) {
      const int j = wave;
      f32x4 v = red[0][j][lane];
#pragma unroll
      for (int w = 1; w < A8_WAVES; ++w) v += red[w][j][lane];
      v += *reinterpret_cast<const f32x4*>(a.bqkv + n0 + 4 * g);
      const half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      st8_agent(qkv + (long long)(j * 16 + col) * ld3 + n0 + 4 * g, o);
    }
  }
  if (tid == 0) s_ok = 1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's stores are acknowledged ...
  __syncthreads();                                                        // ... and so are every wave's (and red[] has been read)
  if (!do23) return;                                                      // two-launch form, first launch: the kernel boundary is the hand-off
  unsigned* const flags = a.flags + ((long long)n * a.heads + h) * 16;
  if (do1 && tid == 0) __hip_atomic_store(flags + part, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- phase 2: the head's q | k | v ----
  if (do1 && tid < A8_PARTS) {                                            // one-launch form only: wait for the head's other parts
    int spins = 0;
    unsigned f;
    do {
      f = __hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } while (f == 0u && ++spins < a.spin_limit);
    if (f == 0u) {                                                        // a part never arrived: not co-resident (or it faulted)
      __hip_atomic_store(a.status, (unsigned)ISHAP_DEV_CHAIN_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      s_ok = 0;
    }
  }
  __syncthreads();
  const bool ok = s_ok != 0;
  const half_t* const hb = qkv + h * 3 * D;                               // head block of a token row: q | k | v
  const int qw = wave & 3;                                                // waves 4 .. 15 mirror 0 .. 3 below and store nothing
  const bool mine = wave < 4;
  half8 qf[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const half_t* p = hb + (long long)(qw * 16 + col) * ld3 + kk * 32 + 8 * g;
    const half4 lo = ld8_agent(p), hi = ld8_agent(p + 4);
    qf[kk] = (half8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  {
    // K and V tiles: 64 rows x 16 eight-byte pieces each; ALL of a thread's pieces are requested before the first goes to LDS
    constexpr int NP = (64 * 16 + A8_WAVES * 64 - 1) / (A8_WAVES * 64);
    half4 kr[NP], vr[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = tid + i * A8_WAVES * 64;
      if (c < 64 * 16) {
        const int r = c >> 4, pc = c & 15;
        kr[i] = ld8_agent(hb + (long long)r * ld3 + D + pc * 4);
        vr[i] = ld8_agent(hb + (long long)r * ld3 + 2 * D + pc * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = tid + i * A8_WAVES * 64;
      if (c < 64 * 16) {
        const int r = c >> 4, pc = c & 15;
        *reinterpret_cast<half4*>(tK + r * RS + pc * 4) = kr[i];
        *reinterpret_cast<half4*>(tV + r * VS + pc * 4) = vr[i];
      }
    }
  }
  __syncthreads();
  if (!mine) return;                                                      // waves 4 .. only helped with phase 1 and the staging (an ended wave no longer counts at s_barrier)
  // S^T = K Q^T (row = key, column = this lane's query), softmax over the 64 keys, O^T = V^T P^T: attn_fwd_kernel's arithmetic
  f32x4 st[4];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub) {
    st[sub] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      st[sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_frag(tK, RS, sub * 16, kk * 32, lane), qf[kk], st[sub], 0, 0, 0);
  }
  float mx = -1e30f;
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st[sub][r] *= a.alpha; mx = fmaxf(mx, st[sub][r]); }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float rs = 0.f;
  half8 pb[2];
#pragma unroll
  for (int sub = 0; sub < 4; ++sub)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float pv = __expf(st[sub][r] - mx);
      rs += pv;
      pb[sub >> 1][(sub & 1) * 4 + r] = (half_t)pv;
    }
  rs += __shfl_xor(rs, 16);
  rs += __shfl_xor(rs, 32);
  f32x4 ot[DS];
#pragma unroll
  for (int i = 0; i < DS; ++i) {
    ot[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const half_t* vb = tV + (4 * g + (col >> 2)) * VS + i * 16 + 4 * (col & 3);
      const half4 lo = lds_read_tr4(vb + (2 * pr) * 16 * VS);
      const half4 hi = lds_read_tr4(vb + (2 * pr + 1) * 16 * VS);
      const half8 va = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      ot[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[pr], ot[i], 0, 0, 0);
    }
  }
  const int q = qw * 16 + col;
  const float inv = ok ? 1.f / rs : __builtin_nanf("");
  if (mine) {
#pragma unroll
    for (int i = 0; i < DS; ++i) {
      const half4 o4 = {(half_t)(ot[i][0] * inv), (half_t)(ot[i][1] * inv), (half_t)(ot[i][2] * inv), (half_t)(ot[i][3] * inv)};
      *reinterpret_cast<half4*>(tA + q * RS + i * 16 + 4 * g) = o4;         // a_h[q][d]: operand of the proj_out slice
      if (part == 0) *reinterpret_cast<half4*>(a.aout + ((long long)n * 64 + q) * C + h * D + i * 16 + 4 * g) = o4;
    }
    if (part == 0 && g == 0) a.lse[((long long)n * a.heads + h) * 64 + q] = mx + __logf(rs);
  }
  __syncthreads();
  if (!mine) return;
  // ---- phase 3: out_h[token][n] for this part's 16-channel tiles; wave = token sub-tile ----
  half8 af[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) af[kk] = ld_frag(tA, RS, qw * 16, kk * 32, lane);
  float* const slice = a.slices + ((long long)h * a.N + n) * 64 * C;
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
    if (t < tcnt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wp[t][kk], af[kk], acc, 0, 0, 0);
      // accumulator: row = channel (t0 + t) * 16 + 4g + r, column = token wave * 16 + col
      *reinterpret_cast<f32x4*>(slice + (long long)(qw * 16 + col) * C + (t0 + t) * 16 + 4 * g) = acc;
    }
}

bool attn8_applicable(int N, int T, int C, int d) {
  static const int on = [] { const char* e = getenv("ISHAP_ATTN8"); return e ? atoi(e) : 1; }();      // -1.0 % per edit (profiles/round5_ab_attn8_fused.txt)
  return on && T == 64 && d == 64 && C % 64 == 0 && C / 16 <= 6 * A8_PARTS && N * (C / 64) * A8_PARTS <= ishap_cu_count();
}

int attn8_fused_launch(const Attn8Args& a, hipStream_t s, bool one_launch) {
  ISHAP_REQUIRE(a.xn && a.wqkv && a.bqkv && a.wproj && a.qkv && a.aout && a.lse && a.slices && a.flags && a.status, "attn8: null argument");
  ISHAP_REQUIRE(a.heads * 64 == a.C && a.C / 16 <= 6 * A8_PARTS && a.N * a.heads * A8_PARTS <= ishap_cu_count(), "attn8: shape / co-residency");
  ISHAP_TRY(ishap_set_max_lds((const void*)attn8_fused_kernel, A8_SMEM));
  Attn8Args b = a;
  // polls before the wait for the head's other parts gives up (ISHAP_GN_SPIN_LIMIT: the test hook of the GroupNorm rendezvous)
  static const int spin = [] { const char* e = getenv("ISHAP_GN_SPIN_LIMIT"); const int n = e ? atoi(e) : 0; return n > 0 ? n : (1 << 22); }();
  b.spin_limit = spin;
  if (one_launch) {
    b.phases = 3;
    hipLaunchKernelGGL(attn8_fused_kernel, dim3(a.heads * A8_PARTS, a.N), dim3(A8_WAVES * 64), A8_SMEM, s, b);
  } else {
    // a sequence without the device's rendezvous tenancy: the SAME kernel twice -- qkv, then attention + proj_out slices -- with the
    // kernel boundary as the hand-off instead of the flags; same instructions on the same operands, so the same bits (round 6)
    b.phases = 1;
    hipLaunchKernelGGL(attn8_fused_kernel, dim3(a.heads * A8_PARTS, a.N), dim3(A8_WAVES * 64), A8_SMEM, s, b);
    b.phases = 2;
    hipLaunchKernelGGL(attn8_fused_kernel, dim3(a.heads * A8_PARTS, a.N), dim3(A8_WAVES * 64), A8_SMEM, s, b);
  }
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

