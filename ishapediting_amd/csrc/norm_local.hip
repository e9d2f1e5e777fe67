// Group-local GroupNorm(32) passes for the small maps (H*W <= 1024: the 32x32, 16x16 and 8x8 levels of the UNet).
// Reference arithmetic: guided_diffusion/nn.py:16-18,92-99 (GroupNorm32 on x.float()), unet.py:236-252 (ResBlock:
// SiLU, FiLM `gn(h)*(1+scale)+shift`, AvgPool2d / nearest-x2 on both branches), and autograd's input gradient of the same
// (drag_utils.py:383).
//
// One workgroup owns one (image, group): H*W pixels x C/32 channels.  That makes every reduction of the layer local --
// the statistics (forward) and the two group means of the input gradient (backward) -- so a GroupNorm is ONE launch with
// no atomics and no finalise prologue, and its input may be a convolution result that is still split over K: the kernel
// adds the fp32 slices (slice order: bitwise reproducible), bias and residual, rounds to fp16 like a stored activation,
// writes that tensor for its other consumers (skip connections, the backward pass) and carries on from the registers.
// The group's values are staged once in LDS (<= 96 KB) between the statistics pass and the apply pass.
// Rounding points are those of norm.hip / norm_bwd.hip, so both routes give the same values.
#include "norm.h"
#include "gn_bwd_terms.h"
#include "gn_act.h"

#ifdef GN_STAMPS          // diagnostic build (tools/experiments/persist_chain.hip -DGN_STAMPS): s_memtime of thread 0 at the phase boundaries
extern __device__ unsigned long long* g_gn_stamps;      // [workgroup][8]
#define GN_STAMP(k)                                                                                    \
  do {                                                                                                 \
    if (threadIdx.x == 0) {                                                                            \
      unsigned long long t_;                                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
      g_gn_stamps[(size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 8 + (k)] = t_;                       \
    }                                                                                                  \
  } while (0)
#else
#define GN_STAMP(k) do {} while (0)
#endif

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int VEC>
__device__ __forceinline__ void ld_half(const half_t* p, float* o) {
  if constexpr (VEC == 8) {
    const half8 v = *reinterpret_cast<const half8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)v[i];
  } else if constexpr (VEC == 4) {
    const half4 v = *reinterpret_cast<const half4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (float)v[i];
  } else if constexpr (VEC == 2) {
    const half2v v = *reinterpret_cast<const half2v*>(p);
    o[0] = (float)v[0]; o[1] = (float)v[1];
  } else {
    o[0] = (float)p[0];
  }
}
template <int VEC>
__device__ __forceinline__ void st_half(half_t* p, const float* v) {
  if constexpr (VEC == 8) {
    half8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (half_t)v[i];
    *reinterpret_cast<half8*>(p) = h;
  } else if constexpr (VEC == 4) {
    half4 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = (half_t)v[i];
    *reinterpret_cast<half4*>(p) = h;
  } else if constexpr (VEC == 2) {
    half2v h;
    h[0] = (half_t)v[0]; h[1] = (half_t)v[1];
    *reinterpret_cast<half2v*>(p) = h;
  } else {
    p[0] = (half_t)v[0];
  }
}
// Loads the COMPILER does not see as loads (inline asm): it inserts no s_waitcnt for them -- not at a use it hoisted, not at a block
// boundary -- so they stay in flight across whatever is issued next; the caller waits with pf_wait() before the first use.
// (vmcnt counts in issue order: any later compiler-inserted wait for a younger load also covers these.)
// RULE: the wait must lie on EVERY path, with the destinations as its operands, before the registers can be re-allocated -- the
// compiler believes the values arrived with the asm statement and frees the registers at their last use on each path.
template <int VEC> struct PfVec;        // VEC consecutive floats / halfs in one register tuple
template <> struct PfVec<2> {
  typedef f32x2 f_t; typedef unsigned h_t;
  static __device__ __forceinline__ f_t ldf(const float* p) { f_t v; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
  static __device__ __forceinline__ h_t ldh(const half_t* p) { h_t v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
  static __device__ __forceinline__ float hget(const h_t& v, int i) { return (float)__builtin_bit_cast(half2v, v)[i]; }
};
template <> struct PfVec<4> {
  typedef f32x4 f_t; typedef f32x2 h_t;
  static __device__ __forceinline__ f_t ldf(const float* p) { f_t v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
  static __device__ __forceinline__ h_t ldh(const half_t* p) { h_t v; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
  static __device__ __forceinline__ float hget(const h_t& v, int i) { return (float)__builtin_bit_cast(half4, v)[i]; }
};
template <int VEC>
__device__ __forceinline__ void add_f32(const float* p, float* acc) {
  if constexpr (VEC == 8) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] += a[i]; acc[4 + i] += b[i]; }
  } else if constexpr (VEC == 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += a[i];
  } else if constexpr (VEC == 2) {
    const f32x2 a = *reinterpret_cast<const f32x2*>(p);
    acc[0] += a[0]; acc[1] += a[1];
  } else {
    acc[0] += p[0];
  }
}

// sum of the pending slices at (row, c .. c+VEC-1) in slice order.  Every slice load of a batch is issued before the first add
// (NB loads in flight per lane: these kernels run on 32 CUs per image and are bound by load latency, not bandwidth);
// NB = the slice count rounded up to a power of two, the surplus loads re-read the last slice and are not added.
// At most 64 floats of slices are held at once (VEC = 8: batches of 8 slices -- sixteen at once spilled 300-420 bytes of
// scratch per lane in the batch-8 generate path, profiles/round4_c2_trace_before.txt); the adds stay in slice order.
template <int VEC, int NB>
__device__ __forceinline__ void slab_sum_nb(const float* p, long long zstride, int count, float* v) {
  float t[NB][VEC];
#pragma unroll
  for (int z = 0; z < NB; ++z) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) t[z][i] = 0.f;
    const int zz = z < count ? z : count - 1;
    add_f32<VEC>(p + zz * zstride, t[z]);
  }
#pragma unroll
  for (int z = 0; z < NB; ++z) {
    if (z < count) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] += t[z][i];
    }
  }
}
template <int VEC>
__device__ __forceinline__ void slab_sum(const SlabSrc& s, long long row, int ld, int c, float* v) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = 0.f;
  const float* p = s.ws + row * ld + c;
  constexpr int MAXNB = VEC == 8 ? 8 : 16;
  if (s.nslab <= 2) slab_sum_nb<VEC, 2>(p, s.zstride, s.nslab, v);
  else if (s.nslab <= 4) slab_sum_nb<VEC, 4>(p, s.zstride, s.nslab, v);
  else if (s.nslab <= 8) slab_sum_nb<VEC, 8>(p, s.zstride, s.nslab, v);
  else {
    for (int z0 = 0; z0 < s.nslab; z0 += MAXNB) {
      const int cnt = s.nslab - z0 < MAXNB ? s.nslab - z0 : MAXNB;
      slab_sum_nb<VEC, MAXNB>(p + z0 * s.zstride, s.zstride, cnt, v);
    }
  }
}

__device__ __forceinline__ float rh(float v) { return (float)(half_t)v; }

constexpr int GN_SCRATCH_BYTES = 512;      // 64 doubles in front of the staging area: block_sum2's 2 x 16 wave partials, then the rendezvous' 4 x parts values
// block-wide sums of two doubles (every thread gets them); `scratch` = 2 * 16 doubles of LDS
__device__ __forceinline__ void block_sum2(double& a, double& b, double* scratch) {
  a = wave_sum_f64(a);                   // DPP + v_readlane (common.h): the butterfly of 64-bit __shfl_xor cost ~400 cycles more
  b = wave_sum_f64(b);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // (no barrier in front: `scratch` is written here for the first time in both kernels)
  if ((threadIdx.x & 63) == 0) { scratch[wave] = a; scratch[16 + wave] = b; }
  __syncthreads();
  a = 0.0; b = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {         // fixed order; all 32 reads in flight (slots >= nw are stale and not added)
    const double pa = scratch[w], pb = scratch[16 + w];
    if (w < nw) { a += pa; b += pb; }
  }
}

// Several workgroups share one (image, group): each publishes its two partial sums as data-tagged 8-byte granules
// {fp32 value, tag} -- ONE naturally aligned agent-scope (sc1, write-through) store each, so a granule is valid by itself and
// needs no flag, fence or ordering -- and every workgroup polls all GN_REC_PER_PART * parts granules of its group with
// agent-scope loads until the tags are there (MI355X_MICROARCH.md, handoff-1to1: ~1 us per hop).
// A sum travels as TWO granules, hi = (float)s and lo = (float)(s - hi): the pair carries 48 significant bits, so the
// variance q/cnt - mean^2 formed from the exchanged totals is as cancellation-safe as the one-workgroup route, which keeps
// its sums in double (a group with |mean| = 100 std: one fp32 granule per sum lost ~10 % of the variance).
// (Alternative measured in round 3: sums about a common pivot every part derives from the group's first element, one fp32
// granule per sum -- the extra uniform loads of the pivot cost more than the two granules they save: 13 100 vs 11 850 cycles.
// Also measured: block sum + rendezvous merged into one routine where wave 0 alone adds the wave partials, publishes, polls
// and adds the parts by v_readlane (two barriers instead of the five of that time): 204 + 5 676 cycles against 1 456 + 3 332 -- slower.
// Round 6 got to two barriers the other way: every thread still adds the wave partials and the parts, but the poll results live in a
// scratch area of their own -- see the comment in the routine.)
// The totals are formed from the parts in part order in double: bitwise reproducible, no atomics.
// rec = [parts][4] granules, zeroed before the launch (tag 0 = not yet written).  The launcher only uses parts > 1 when the
// whole grid is resident at once; the spin is bounded all the same, and a give-up is an ERROR: it raises the process-wide
// device status word (common.h) and poisons this group's sums with NaN, so nothing downstream looks plausible.
// Round 5, XCD-local copy (GN_XCD_LOCAL): the launcher deals the parts of one (image, group) to workgroups with equal
// `linear id & 7`, which the hardware places on ONE XCD (observed round-robin placement: speed only, never relied on for
// correctness).  Every granule is then stored TWICE: with a plain store into the record's second half -- it stays in that XCD's
// L2, where the other parts' L1-bypassing polls find it after an L2 round trip (~200 cycles) instead of a trip to the memory side
// (~550-900) -- and with the agent-scope store as before.  A poller checks the local copy `GN_LOCAL_POLLS` times per look at the
// agent-scope copy, so a placement that is NOT co-located (or an L2 that does not show the plain store) costs time, never
// correctness: the agent-scope path is complete by itself.
#ifndef GN_XCD_LOCAL
#define GN_XCD_LOCAL 1
#endif
#ifndef GN_LOCAL_POLLS
#define GN_LOCAL_POLLS 4
#endif
__device__ __forceinline__ void group_rendezvous(double& a, double& b, unsigned long long* rec, int part, int parts,
                                                 double* scratch, unsigned* status, int spin_limit, bool xcd_local) {
  if (parts <= 1) return;
  // (round 6: no barrier in front and none behind -- the poll results go to the SECOND 32 doubles of `scratch`, which nothing else in
  // these kernels touches, so block_sum2's reads of the first 32 cannot collide with them; five barriers per launch became two)
  double* const mine = scratch + 32;
  if (threadIdx.x < GN_REC_PER_PART) {
    const double d = (threadIdx.x & 2) ? b : a;
    const float hi = (float)d;
    const float v = (threadIdx.x & 1) ? (float)(d - (double)hi) : hi;
    const unsigned long long granule = (unsigned long long)__float_as_uint(v) | (1ull << 32);
    if (GN_XCD_LOCAL && xcd_local) rec[GN_REC_HALF + part * GN_REC_PER_PART + threadIdx.x] = granule;      // plain: stays in this XCD's L2
    __hip_atomic_store(rec + part * GN_REC_PER_PART + threadIdx.x, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if ((int)threadIdx.x < GN_REC_PER_PART * parts) {
    unsigned long long gr = 0;
    int spins = 0;
    if (GN_XCD_LOCAL && xcd_local) {
      int k = 0;                         // polls GN_LOCAL_POLLS times at the local copy (sc1: past L1, served by L2), then once at the agent-scope one
      do {
        gr = __hip_atomic_load(rec + (k < GN_LOCAL_POLLS ? GN_REC_HALF : 0) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        k = k == GN_LOCAL_POLLS ? 0 : k + 1;
      } while ((gr >> 32) == 0ull && ++spins < spin_limit);
    } else {
      do {
        gr = __hip_atomic_load(rec + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } while ((gr >> 32) == 0ull && ++spins < spin_limit);
    }
    double v = (double)__uint_as_float((unsigned)(gr & 0xffffffffull));
    if ((gr >> 32) == 0ull) {          // a part never arrived: the grid was not co-resident (or a part faulted)
      __hip_atomic_store(status, (unsigned)ISHAP_DEV_GN_RENDEZVOUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      v = __builtin_nan("");
    }
    mine[threadIdx.x] = v;
  }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
  for (int p = 0; p < parts; ++p) {    // (hi + lo) is exact in double; parts add in part order
    ta += mine[GN_REC_PER_PART * p] + mine[GN_REC_PER_PART * p + 1];
    tb += mine[GN_REC_PER_PART * p + 2] + mine[GN_REC_PER_PART * p + 3];
  }
  a = ta;
  b = tb;
}

// workgroup -> (group, part).  xcd_local: the parts of a group get linear ids 8 apart (equal `id & 7` = one XCD under round-robin
// placement; gridDim.x is exactly 32 * parts, so the image index blockIdx.y adds a multiple of 8 to the linear id -- round 6: the
// prefetch workgroups that used to ride at the end of the grid and broke this for batches > 1 are gone); else parts are consecutive ids
__device__ __forceinline__ void group_of_block(int parts, bool xcd_local, int& g, int& part) {
  if (xcd_local) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    part = j % parts;
    g = (j / parts) * 8 + xcd;
  } else {
    g = blockIdx.x / parts;
    part = blockIdx.x - g * parts;
  }
}

// ------------------------------------------------------------------------------------------------ forward
template <int VEC, bool FILM, bool ACT, bool POOL>
__global__ __launch_bounds__(1024) void gn_local_kernel(int h_parts, int h_C, int h_HW, int h_Ca, GnLocalArgs a) {
  // h_*: copies of a.parts, a.C, a.H * a.W, a.Ca as leading scalar parameters -- preloaded into SGPRs at dispatch (common.h,
  // IgemmHot), so that the index arithmetic (three integer divisions) runs UNDER the s_load of the argument block, not after it
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* scratch = reinterpret_cast<double*>(smem_raw);                 // 64 doubles (block_sum2: 2 x 16; rendezvous: 4 x parts in the second half)
  half_t* st = reinterpret_cast<half_t*>(smem_raw + GN_SCRATCH_BYTES);   // [HW][cpg]
  const bool xcd_local = h_parts < 0;                                    // the launcher passes -(parts [+ 64]) for the XCD-local dealing
  const int parts = xcd_local ? (-h_parts & 63) : h_parts;
  int g, part;
  group_of_block(parts, xcd_local, g, part);
  const int n = blockIdx.y;
  // +64: touch the local copy of this group's record now -- its lines are then resident in this XCD's L2 when the parts' plain
  // stores and polls arrive (the first poll otherwise waits for the line's fill from the memory side, ~1 us)
  unsigned long long warm = 0;
  if (xcd_local && (-h_parts & 64) && (int)threadIdx.x < GN_REC_PER_PART * parts)
    warm = __hip_atomic_load(a.rec + ((long long)n * 32 + g) * GN_REC_STRIDE + GN_REC_HALF + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int C = h_C, cpg = C / 32, VPP = cpg / VEC, HW = h_HW;
  const int PP = HW / parts, p0 = part * PP;                             // this workgroup's pixels [p0, p0 + PP)
  const int nunits = PP * VPP, c0g = g * cpg;
  const int Cb = C - h_Ca;
  GN_STAMP(0);
  const bool pend = a.slab.pending();
  // affine / FiLM operands of this thread's FIRST unit, fetched now: their round trip then overlaps the statistics pass and
  // the rendezvous instead of standing between them and the output (a thread rarely has a second unit on these maps)
  const int c_first = c0g + ((int)threadIdx.x % VPP) * VEC;
  float pg[VEC], pb[VEC], psc[VEC], psh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    pg[i] = a.gamma[c_first + i];
    pb[i] = a.beta[c_first + i];
    psc[i] = FILM ? a.emb[(long long)n * a.emb_ld + c_first + i] : 0.f;
    psh[i] = FILM ? a.emb[(long long)n * a.emb_ld + C + c_first + i] : 0.f;
  }
  // Round 6: bias / bias2 / residual of a pending input's FIRST unit as inline-asm loads (PfVec above), requested here.  The source
  // below asks for them "before the slices are waited for", but the ISA had an s_waitcnt vmcnt(0) at the end of the conditional
  // blocks that hold those loads -- a full memory round trip in front of the slice loads of ~55 launches per guided step.
  // Unconditional loads (a missing operand reads gamma, a valid address, and is not used), waited for behind the slices.
  constexpr bool PF = VEC == 2;
  typedef PfVec<2> pf;
  typename pf::f_t r_b1{}, r_b2{};
  typename pf::h_t r_res{};
  if constexpr (PF) {
    const int u_f = (int)threadIdx.x < nunits ? (int)threadIdx.x : nunits - 1;
    const int pl_f = u_f / VPP, c_f = c0g + (u_f - pl_f * VPP) * VEC, p_f = p0 + pl_f;
    const bool on = pend && c_f < a.Ca;
    long long rrow = (long long)n * HW + p_f;
    if (a.slab.res_ups) {
      const int py = p_f / a.W, px = p_f - py * a.W;
      rrow = (long long)n * (HW >> 2) + (py >> 1) * (a.W >> 1) + (px >> 1);
    }
    r_b1 = pf::ldf((on && a.slab.bias) ? a.slab.bias + c_f : a.gamma + c_first);
    r_b2 = pf::ldf((on && a.slab.bias2) ? a.slab.bias2 + c_f : a.gamma + c_first);
    r_res = pf::ldh((on && a.slab.res) ? a.slab.res + rrow * a.slab.ldr + c_f : reinterpret_cast<const half_t*>(a.gamma + c_first));
  }
  auto pf_wait = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(r_b1), "+v"(r_b2), "+v"(r_res)::"memory"); };
  double s = 0.0, q = 0.0;
  for (int u = threadIdx.x; u < nunits; u += blockDim.x) {
    const int pl = u / VPP, cv = u - pl * VPP, c = c0g + cv * VEC;
    const int p = p0 + pl;
    const long long row = (long long)n * HW + p;
    float v[VEC];
    if (c < a.Ca) {
      if (pend) {
        // bias / residual are fetched BEFORE the slices are waited for: one memory round trip instead of three in a row
        // (the kernel is latency-bound, tools/experiments/persist_chain.hip -DGN_STAMPS)
        const bool pre = PF && u == (int)threadIdx.x;                    // requested above
        float bv[VEC], b2v[VEC], r[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) { bv[i] = 0.f; b2v[i] = 0.f; r[i] = 0.f; }
        if (!pre) {
          if (a.slab.bias) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) bv[i] = a.slab.bias[c + i];
          }
          if (a.slab.bias2) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) b2v[i] = a.slab.bias2[c + i];
          }
          if (a.slab.res) {
            long long rrow = row;
            if (a.slab.res_ups) {
              const int py = p / a.W, px = p - py * a.W;
              rrow = (long long)n * (HW >> 2) + (py >> 1) * (a.W >> 1) + (px >> 1);
            }
            ld_half<VEC>(a.slab.res + rrow * a.slab.ldr + c, r);
          }
        }
        slab_sum<VEC>(a.slab, row, a.Ca, c, v);
        if (pre) {
          pf_wait();                                                     // issued before the slices: landed with them
#pragma unroll
          for (int i = 0; i < VEC; ++i) { bv[i] = r_b1[i]; b2v[i] = r_b2[i]; r[i] = pf::hget(r_res, i); }
        }
        if (a.slab.bias) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[i] += bv[i];
        }
        if (a.slab.bias2) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[i] += b2v[i];
        }
        if (a.slab.res) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[i] += r[i];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = rh(v[i]);                  // the stored activation is fp16
        if (a.ya) st_half<VEC>(a.ya + row * a.Ca + c, v);
      } else {
        ld_half<VEC>(a.xa + row * a.Ca + c, v);
      }
    } else {
      ld_half<VEC>(a.xb + row * Cb + (c - a.Ca), v);
    }
    if (a.xcopy) st_half<VEC>(a.xcopy + row * C + c, v);
    st_half<VEC>(st + pl * cpg + cv * VEC, v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { s += (double)v[i]; q += (double)v[i] * (double)v[i]; }
  }
  if constexpr (PF) pf_wait();       // on EVERY path before the destinations can be re-allocated (PfVec, RULE)
  GN_STAMP(1);
  block_sum2(s, q, scratch);
  GN_STAMP(2);
  group_rendezvous(s, q, a.rec + ((long long)n * 32 + g) * GN_REC_STRIDE, part, parts, scratch, a.status, a.spin_limit, xcd_local);
  asm volatile("" ::"v"(warm));
  GN_STAMP(3);
  const double cnt = (double)HW * (double)cpg;
  const double md = s / cnt;
  double vd = q / cnt - md * md;
  vd = vd < 0.0 ? 0.0 : vd;
  const float mean = (float)md, rstd = (float)(1.0 / sqrt(vd + 1e-5));
  if (threadIdx.x == 0 && part == 0 && a.stats_out) {
    a.stats_out[(n * 32 + g) * 2] = mean;
    a.stats_out[(n * 32 + g) * 2 + 1] = rstd;
  }
  // (no barrier here: the staging writes all precede block_sum2's barrier)

  auto activate = [&](const float* x, int c, float* o) {
    const bool first = c == c_first;               // the prefetched operands (else: this thread's later units, from memory)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float gm = first ? pg[i] : a.gamma[c + i], bt = first ? pb[i] : a.beta[c + i];
      half_t yh = gn_affine(x[i], mean, rstd, gm, bt);
      if (FILM) {
        const half_t sc = (half_t)(1.f + rh(first ? psc[i] : a.emb[(long long)n * a.emb_ld + c + i]));
        const half_t sh = (half_t)(first ? psh[i] : a.emb[(long long)n * a.emb_ld + C + c + i]);
        yh = gn_film(yh, sc, sh);
      }
      float y = (float)yh;
      if (ACT) y = rh(gn_silu(y));
      o[i] = y;
    }
  };
  if (!POOL) {
    for (int u = threadIdx.x; u < nunits; u += blockDim.x) {
      const int pl = u / VPP, cv = u - pl * VPP, c = c0g + cv * VEC;
      float x[VEC], o[VEC];
      ld_half<VEC>(st + pl * cpg + cv * VEC, x);
      activate(x, c, o);
      st_half<VEC>(a.out + ((long long)n * HW + p0 + pl) * C + c, o);
    }
    GN_STAMP(4);
  } else {
    // a part is a whole number of row pairs (launcher), so its 2x2 cells are its own
    const int Wo = a.W >> 1, HWo = HW >> 2, PPo = PP >> 2, po0 = p0 >> 2;
    for (int u = threadIdx.x; u < PPo * VPP; u += blockDim.x) {
      const int pol = u / VPP, cv = u - pol * VPP, c = c0g + cv * VEC;
      const int po = po0 + pol;
      const int yo = po / Wo, xo = po - yo * Wo;
      float acc[VEC], xacc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) { acc[i] = 0.f; xacc[i] = 0.f; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int p = (2 * yo + (k >> 1)) * a.W + 2 * xo + (k & 1);
        float x[VEC], o[VEC];
        ld_half<VEC>(st + (p - p0) * cpg + cv * VEC, x);
        activate(x, c, o);
#pragma unroll
        for (int i = 0; i < VEC; ++i) { acc[i] += o[i]; xacc[i] += x[i]; }
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) { acc[i] *= 0.25f; xacc[i] *= 0.25f; }
      st_half<VEC>(a.out + ((long long)n * HWo + po) * C + c, acc);
      if (a.xpool) st_half<VEC>(a.xpool + ((long long)n * HWo + po) * C + c, xacc);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
// upstream gradient of pixel p (GN-input resolution), channels c..c+VEC-1, from a dense fp16 map or pending slices
template <int VEC>
__device__ __forceinline__ void upstream(const half_t* g, const SlabSrc& slab, int gmode, int n, int p, int H, int W, int C,
                                         int c, float* o) {
  auto at = [&](long long row, float* v) {
    if (slab.pending()) {
      slab_sum<VEC>(slab, row, C, c, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = rh(v[i]);                    // a stored gradient map is fp16
    } else {
      ld_half<VEC>(g + row * C + c, v);
    }
  };
  const int y = p / W, x = p - y * W;
  if (gmode == GB_SAME) {
    at((long long)n * H * W + p, o);
  } else if (gmode == GB_UNPOOL) {
    at((long long)n * (H >> 1) * (W >> 1) + (y >> 1) * (W >> 1) + (x >> 1), o);
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] *= 0.25f;
  } else {
    const int W2 = W << 1;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float t[VEC];
      at((long long)n * (H << 1) * W2 + (2 * y + (k >> 1)) * W2 + 2 * x + (k & 1), t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] += t[i];
    }
  }
}
template <int VEC>
__device__ __forceinline__ void addend(const half_t* g, int gmode, int n, int p, int H, int W, int C, int c, float* o) {
  SlabSrc none;
  upstream<VEC>(g, none, gmode, n, p, H, W, C, c, o);
}

// STAGE32: the staged upstream values are fp32 (GB_UNPOOL / GB_SUM4: a quarter or a sum of four fp16 values is not an fp16 value)
template <int VEC, bool FILM, bool ACT, bool STAGE32>
__global__ __launch_bounds__(1024) void gn_bwd_local_kernel(int h_parts, int h_C, int h_HW, GnBwdLocalArgs a) {
  // h_*: preloaded copies of a.parts, a.C, a.H * a.W (see gn_local_kernel)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* scratch = reinterpret_cast<double*>(smem_raw);
  half_t* st16 = reinterpret_cast<half_t*>(smem_raw + GN_SCRATCH_BYTES);
  float* st32 = reinterpret_cast<float*>(smem_raw + GN_SCRATCH_BYTES);
  const bool xcd_local = h_parts < 0;
  const int parts = xcd_local ? (-h_parts & 63) : h_parts;
  int g, part;
  group_of_block(parts, xcd_local, g, part);
  const int n = blockIdx.y;
  unsigned long long warm = 0;           // see gn_local_kernel
  if (xcd_local && (-h_parts & 64) && (int)threadIdx.x < GN_REC_PER_PART * parts)
    warm = __hip_atomic_load(a.rec + ((long long)n * 32 + g) * GN_REC_STRIDE + GN_REC_HALF + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int C = h_C, cpg = C / 32, VPP = cpg / VEC, HW = h_HW;
  const int PP = HW / parts, p0 = part * PP;
  const int nunits = PP * VPP, c0g = g * cpg;
  const float mu = a.stats[(n * 32 + g) * 2], rs = a.stats[(n * 32 + g) * 2 + 1];
  double s1 = 0.0, s2 = 0.0;
  // A thread's FIRST unit (its only one on these maps, with rare exceptions) keeps dyh / xh in registers across the
  // reduction and fetches its addends before it: the pass after the rendezvous then needs no memory round trip of its own.
  float f_dyh[VEC], f_xh[VEC], f_ad[VEC], f_a2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { f_dyh[i] = 0.f; f_xh[i] = 0.f; f_ad[i] = 0.f; f_a2[i] = 0.f; }
  // Round 6: everything the first unit needs besides the upstream gradient -- the saved GroupNorm input, gamma / beta / FiLM rows, the
  // addends -- is requested BEFORE the pending slices are waited for.  The ISA had them behind the slice loads' s_waitcnt vmcnt(0)
  // (loads return in order): two dependent memory round trips in a kernel that is nothing but latency (60 launches per guided step).
  // The compiler cannot be talked into it (it hoists the loop-invariant products of gamma / FiLM above the loop and waits for them,
  // and waits at the end of every conditional block that holds a load), so at VEC = 2 -- 58 of the 60 launches per guided step -- these
  // loads are inline asm (PfVec above), unconditional (a thread without a unit reads the last unit's operands) and first touched
  // behind pf_wait below, after the slices have been requested AND waited for.
  constexpr bool PF = VEC == 2;      // VEC = 4 (2 launches per step) is at its 128-register cap: the extra live values spilled, and a spill of an in-flight destination is a wrong value
  typedef PfVec<PF ? VEC : 2> pf;
  const int u_first = (int)threadIdx.x < nunits ? (int)threadIdx.x : nunits - 1;
  const int pl_f = u_first / VPP, c_f = c0g + (u_first - pl_f * VPP) * VEC, p_f = p0 + pl_f;
  const bool add_same = a.gmode == GB_SAME;
  typename pf::h_t r_x{}, r_ad{}, r_a2{};
  typename pf::f_t r_gam{}, r_bet{}, r_esc{}, r_esh{};
  if constexpr (PF) {
    r_x = pf::ldh(a.x + ((long long)n * HW + p_f) * C + c_f);
    r_gam = pf::ldf(a.gamma + c_f);
    if (FILM || ACT) r_bet = pf::ldf(a.beta + c_f);
    if (FILM) { r_esc = pf::ldf(a.emb + (long long)n * a.emb_ld + c_f); r_esh = pf::ldf(a.emb + (long long)n * a.emb_ld + C + c_f); }
    // addends at the gradient's own resolution are one vector each; the pooled / upsampled forms (up and down blocks) add several
    // values up and stay behind the loop
    // (always loaded, from the saved input when there is no such addend: a conditional asm load would leave a register merge -- a
    // copy of a destination that has not landed yet -- to the allocator's mercy)
    const long long o_f = ((long long)n * HW + p_f) * C + c_f;
    r_ad = pf::ldh((a.add && add_same) ? a.add + o_f : a.x + o_f);
    r_a2 = pf::ldh(a.add2 ? a.add2 + o_f : a.x + o_f);
  }
  auto pf_wait = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r_x), "+v"(r_ad), "+v"(r_a2), "+v"(r_gam), "+v"(r_bet), "+v"(r_esc), "+v"(r_esh)::"memory");
  };
  for (int u = threadIdx.x; u < nunits; u += blockDim.x) {
    const int pl = u / VPP, cv = u - pl * VPP, c = c0g + cv * VEC;
    const int p = p0 + pl;
    const bool first = u == (int)threadIdx.x;
    float up[VEC], xv[VEC];
    upstream<VEC>(a.g, a.slab, a.gmode, n, p, a.H, a.W, C, c, up);
    const bool pre = PF && first;                  // this unit's operands were requested above
    float gm[VEC], bt[VEC], es[VEC], eh[VEC];
    if constexpr (PF) {
      if (pre) {
        pf_wait();                                 // everything above has landed long ago (the slices were issued later): free
#pragma unroll
        for (int i = 0; i < VEC; ++i) { xv[i] = pf::hget(r_x, i); gm[i] = r_gam[i]; bt[i] = r_bet[i]; es[i] = r_esc[i]; eh[i] = r_esh[i]; }
      } else {
        ld_half<VEC>(a.x + ((long long)n * HW + p) * C + c, xv);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          gm[i] = a.gamma[c + i];
          bt[i] = (FILM || ACT) ? a.beta[c + i] : 0.f;
          es[i] = FILM ? a.emb[(long long)n * a.emb_ld + c + i] : 0.f;
          eh[i] = FILM ? a.emb[(long long)n * a.emb_ld + C + c + i] : 0.f;
        }
      }
    } else {
      ld_half<VEC>(a.x + ((long long)n * HW + p) * C + c, xv);
    }
    if (STAGE32) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) st32[pl * cpg + cv * VEC + i] = up[i];
    } else {
      st_half<VEC>(st16 + pl * cpg + cv * VEC, up);                   // exact: these values are fp16
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float dyh, xh;
      if constexpr (PF)
        gn_bwd_term(up[i], xv[i], mu, rs, gm[i], bt[i], es[i], eh[i], FILM, ACT, dyh, xh);
      else      // the operands straight from memory, as before round 6 (the VEC = 4 / 8 forms sit at their register cap)
        gn_bwd_term(up[i], xv[i], mu, rs, a.gamma[c + i], (FILM || ACT) ? a.beta[c + i] : 0.f,
                    FILM ? a.emb[(long long)n * a.emb_ld + c + i] : 0.f, FILM ? a.emb[(long long)n * a.emb_ld + C + c + i] : 0.f,
                    FILM, ACT, dyh, xh);
      s1 += (double)dyh;
      s2 += (double)dyh * (double)xh;
      if (first) { f_dyh[i] = dyh; f_xh[i] = xh; }
    }
  }
  // EVERY thread passes a wait for its asm loads before their destination registers may be given to anything else: a thread
  // without a unit never ran the loop's wait, and on its path the compiler considers those registers dead from here on -- a load
  // that lands later would overwrite whatever was allocated there (seen: the lane index of block_sum2, i.e. garbage group sums).
  if constexpr (PF) pf_wait();
  if ((int)threadIdx.x < nunits) {
    if (a.add) {
      if (PF && add_same) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) f_ad[i] = pf::hget(r_ad, i);
      } else {
        addend<VEC>(a.add, a.gmode, n, p_f, a.H, a.W, C, c_f, f_ad);
      }
    }
    if (a.add2) {
      if (PF) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) f_a2[i] = pf::hget(r_a2, i);
      } else {
        ld_half<VEC>(a.add2 + ((long long)n * HW + p_f) * C + c_f, f_a2);
      }
    }
  }
  block_sum2(s1, s2, scratch);
  group_rendezvous(s1, s2, a.rec + ((long long)n * 32 + g) * GN_REC_STRIDE, part, parts, scratch, a.status, a.spin_limit, xcd_local);
  asm volatile("" ::"v"(warm));
  const double cnt = (double)HW * (double)cpg;
  const float m1 = (float)(s1 / cnt), m2 = (float)(s2 / cnt);
  // (no barrier here: the staging writes all precede block_sum2's barrier, and a thread re-reads what it staged itself)
  for (int u = threadIdx.x; u < nunits; u += blockDim.x) {
    const int pl = u / VPP, cv = u - pl * VPP, c = c0g + cv * VEC;
    const int p = p0 + pl;
    const long long pix = (long long)n * HW + p;
    float up[VEC], xv[VEC], ad[VEC], a2[VEC], o[VEC];
    if (u == (int)threadIdx.x) {                   // the first unit: everything is in registers already
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float v = gn_bwd_dx(rs, f_dyh[i], f_xh[i], m1, m2);
        if (a.add) v += f_ad[i];
        if (a.add2) v = rh(v) + f_a2[i];                                 // same rounding as a separate fp16 add of the two maps
        o[i] = v;
      }
      if (a.csplit == 0) st_half<VEC>(a.dx + pix * C + c, o);
      else if (c < a.csplit) st_half<VEC>(a.dx + pix * a.csplit + c, o);
      else st_half<VEC>(a.dx2 + pix * (C - a.csplit) + (c - a.csplit), o);
      continue;
    }
    if (STAGE32) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) up[i] = st32[pl * cpg + cv * VEC + i];
    } else {
      ld_half<VEC>(st16 + pl * cpg + cv * VEC, up);
    }
    ld_half<VEC>(a.x + pix * C + c, xv);
    if (a.add) addend<VEC>(a.add, a.gmode, n, p, a.H, a.W, C, c, ad);
    if (a.add2) ld_half<VEC>(a.add2 + pix * C + c, a2);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float dyh, xh;
      gn_bwd_term(up[i], xv[i], mu, rs, a.gamma[c + i], (FILM || ACT) ? a.beta[c + i] : 0.f,
                  FILM ? a.emb[(long long)n * a.emb_ld + c + i] : 0.f, FILM ? a.emb[(long long)n * a.emb_ld + C + c + i] : 0.f,
                  FILM, ACT, dyh, xh);
      float v = gn_bwd_dx(rs, dyh, xh, m1, m2);
      if (a.add) v += ad[i];
      if (a.add2) v = rh(v) + a2[i];                                   // same rounding as a separate fp16 add of the two maps
      o[i] = v;
    }
    if (a.csplit == 0) st_half<VEC>(a.dx + pix * C + c, o);
    else if (c < a.csplit) st_half<VEC>(a.dx + pix * a.csplit + c, o);
    else st_half<VEC>(a.dx2 + pix * (C - a.csplit) + (c - a.csplit), o);
  }
}

// the widest vector that still gives every one of 1024 threads a unit (latency-bound kernels: lanes in flight matter more than
// bytes per lane); pending sources read fp32 slices, 16 B per lane per slice at width 4
int pick_vec(int cpg, int HW, bool pending) {
  const int cand[4] = {8, 4, 2, 1};
  int best = 1;
  for (int k = 0; k < 4; ++k) {
    const int v = cand[k];
    if (cpg % v) continue;
    if (pending && v == 8 && cpg % 4 == 0) continue;
    best = v;
    if ((long long)HW * (cpg / v) >= 1024 || v <= 2) break;
  }
  return best;
}
int pick_threads(int nunits) { return nunits >= 1024 ? 1024 : (nunits <= 256 ? 256 : (nunits + 63) / 64 * 64); }
// workgroups per (image, group): as many as keep the whole grid resident at once (the rendezvous spins) and leave a part
// at least `min_pixels` pixels made of whole units of `unit` pixels (a row pair when the kernel pools)
int pick_parts(int N, int HW, int cpg, int unit, bool have_rec) {
  static const int maxp = [] { const char* e = getenv("ISHAP_GN_PARTS"); return e ? atoi(e) : 8; }();
  constexpr int min_el = 128;       // elements per part at least (in-situ sweeps below)
  if (!have_rec) return 1;
  // the rendezvous costs ~3 atomic round trips (3-4 us), yet more, smaller parts still win down to a few hundred elements
  // per workgroup (in-situ sweep: 0.2445 / 0.2464 / 0.2500 / 0.2574 s per edit at >= 256 / 1024 / 2048 / 4096 elements;
  // with the tagged-granule rendezvous, tools/experiments/gn_parts_probe.sh: local-GN kernel time 193 / 198 / 216 / 243 at >= 128 / 256 /
  // 1024 / 2048 elements, 226 / 271 with at most 4 / 2 parts, 203-214 with up to 16 parts)
  int p = 1;
  // co-residency: a workgroup of these kernels (<= 1024 threads, <= 160 KB of LDS) always fits a compute unit by itself, so
  // a grid of at most one workgroup per CU OF THIS DEVICE (a partitioned MI355X exposes fewer than 256) becomes resident
  // whatever else is draining
  const int max_wgs = ishap_cu_count();
  while (p * 2 <= maxp && p * 2 <= GN_REC_HALF / GN_REC_PER_PART && 32 * N * (p * 2) <= max_wgs && HW % (p * 2) == 0 &&
         (HW / (p * 2)) % unit == 0 && (long long)(HW / (p * 2)) * cpg >= min_el)
    p *= 2;
  return p;
}
constexpr size_t LOCAL_LDS_CAP = 160 * 1024 - GN_SCRATCH_BYTES;
// polls before a rendezvous gives up; ISHAP_GN_SPIN_LIMIT exists so that a test can force the give-up path (1 poll)
int spin_limit() {
  static const int v = [] { const char* e = getenv("ISHAP_GN_SPIN_LIMIT"); const int n = e ? atoi(e) : 0; return n > 0 ? n : GN_SPIN_LIMIT; }();
  return v;
}

// the parts of a group on one XCD (group_of_block) and the XCD-local copy of the record: ISHAP_GN_XCD=0 switches both off
int xcd_deal(int parts) {          // 0: off; 1: XCD-local dealing + local copy; 2: also touch the local copy's lines at kernel start
  static const int on = [] { const char* e = getenv("ISHAP_GN_XCD"); return e ? atoi(e) : 1; }();      // in situ: 1 and 2 both -0.4 % against 0, no difference between them (profiles/round5_ab_gn_xcd_local.txt)
  return parts > 1 ? on : 0;
}

template <typename K>
int set_lds(K kern, size_t smem) {
  (void)smem;
  return ishap_set_max_lds((const void*)kern, (int)LOCAL_LDS_CAP + GN_SCRATCH_BYTES);
}

}  // namespace

bool gn_local_fits(int HW, int C) {
  return C % 32 == 0 && (size_t)HW * (C / 32) * sizeof(half_t) <= LOCAL_LDS_CAP;
}
int gn_local_parts(int N, int HW, int C) { return C % 32 ? 1 : pick_parts(N, HW, C / 32, 1, true); }
bool gn_bwd_local_fits(int HW, int C, int gmode) {
  return C % 32 == 0 && (size_t)HW * (C / 32) * (gmode != GB_SAME ? sizeof(float) : sizeof(half_t)) <= LOCAL_LDS_CAP;
}

int gn_local_launch(const GnLocalArgs& a, hipStream_t s) {
  ISHAP_REQUIRE(a.C % 32 == 0 && a.Ca > 0 && a.Ca <= a.C && a.Ca % 32 == 0, "GroupNorm channels: multiples of 32");
  ISHAP_REQUIRE((a.Ca == a.C) == (a.xb == nullptr), "second source exactly when the input is a concatenation");
  ISHAP_REQUIRE(a.slab.pending() || a.xa, "source A: a tensor or pending slices");
  ISHAP_REQUIRE(!a.pool || (a.H % 2 == 0 && a.W % 2 == 0 && !a.film), "pool variant");
  const int HW = a.H * a.W, cpg = a.C / 32;
  ISHAP_REQUIRE(gn_local_fits(HW, a.C), "group does not fit in LDS");
  GnLocalArgs b = a;
  b.spin_limit = spin_limit();
  b.status = ishap_status_word();
  ISHAP_REQUIRE(b.status != nullptr, "device status word");
  b.parts = pick_parts(a.N, HW, cpg, a.pool ? 2 * a.W : 1, a.rec != nullptr);
  const int PP = HW / b.parts;
  const int VEC = pick_vec(cpg, PP, a.slab.pending());
  const int T = pick_threads(PP * (cpg / VEC));
  const size_t smem = GN_SCRATCH_BYTES + (size_t)PP * cpg * sizeof(half_t);
  dim3 grid(32 * b.parts, a.N), blk(T);
#define GL_LAUNCH(V, F, A, P)                                                            \
  do {                                                                                   \
    auto kern = gn_local_kernel<V, F, A, P>;                                             \
    ISHAP_TRY(set_lds(kern, smem));                                                      \
    hipLaunchKernelGGL(kern, grid, blk, smem, s, xcd_deal(b.parts) ? -(b.parts + (xcd_deal(b.parts) == 2 ? 64 : 0)) : b.parts, b.C, b.H * b.W, b.Ca, b);      \
  } while (0)
#define GL_VARIANT(V)                                                                    \
  do {                                                                                   \
    if (a.pool) GL_LAUNCH(V, false, true, true);                                         \
    else if (a.film) GL_LAUNCH(V, true, true, false);                                    \
    else if (a.act) GL_LAUNCH(V, false, true, false);                                    \
    else GL_LAUNCH(V, false, false, false);                                              \
  } while (0)
  ISHAP_REQUIRE(!a.pool || a.act, "the pooled variant carries SiLU (ResBlock in_layers)");
  ISHAP_REQUIRE(!a.film || a.act, "FiLM is followed by SiLU (ResBlock out_layers)");
  switch (VEC) {
    case 8: GL_VARIANT(8); break;
    case 4: GL_VARIANT(4); break;
    case 2: GL_VARIANT(2); break;
    default: GL_VARIANT(1); break;
  }
#undef GL_VARIANT
#undef GL_LAUNCH
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}

int gn_bwd_local_launch(const GnBwdLocalArgs& a, hipStream_t s) {
  ISHAP_REQUIRE(a.C % 32 == 0, "GroupNorm channels");
  ISHAP_REQUIRE(a.csplit == 0 || (a.dx2 && a.csplit % 32 == 0 && a.csplit < a.C), "split output");
  ISHAP_REQUIRE(a.slab.pending() || a.g, "upstream gradient: a tensor or pending slices");
  ISHAP_REQUIRE(!a.film || a.act, "FiLM is followed by SiLU");
  const int HW = a.H * a.W, cpg = a.C / 32;
  const bool s32 = a.gmode != GB_SAME;      // 0.25 * fp16 and sums of four fp16 values are kept in fp32 between the passes
  GnBwdLocalArgs b = a;
  b.spin_limit = spin_limit();
  b.status = ishap_status_word();
  ISHAP_REQUIRE(b.status != nullptr, "device status word");
  b.parts = pick_parts(a.N, HW, cpg, 1, a.rec != nullptr);
  const int PP = HW / b.parts;
  const size_t smem = GN_SCRATCH_BYTES + (size_t)PP * cpg * (s32 ? sizeof(float) : sizeof(half_t));
  ISHAP_REQUIRE(smem <= LOCAL_LDS_CAP + GN_SCRATCH_BYTES, "group does not fit in LDS");
  const int VEC = pick_vec(cpg, PP, a.slab.pending());
  const int T = pick_threads(PP * (cpg / VEC));
  dim3 grid(32 * b.parts, a.N), blk(T);
#define GB_LAUNCH(V, F, A, S32)                                                          \
  do {                                                                                   \
    auto kern = gn_bwd_local_kernel<V, F, A, S32>;                                       \
    ISHAP_TRY(set_lds(kern, smem));                                                      \
    hipLaunchKernelGGL(kern, grid, blk, smem, s, xcd_deal(b.parts) ? -(b.parts + (xcd_deal(b.parts) == 2 ? 64 : 0)) : b.parts, b.C, b.H * b.W, b);            \
  } while (0)
#define GB_VARIANT(V)                                                                    \
  do {                                                                                   \
    if (s32) {                                                                           \
      if (a.film) GB_LAUNCH(V, true, true, true);                                        \
      else if (a.act) GB_LAUNCH(V, false, true, true);                                   \
      else GB_LAUNCH(V, false, false, true);                                             \
    } else {                                                                             \
      if (a.film) GB_LAUNCH(V, true, true, false);                                       \
      else if (a.act) GB_LAUNCH(V, false, true, false);                                  \
      else GB_LAUNCH(V, false, false, false);                                            \
    }                                                                                    \
  } while (0)
  switch (VEC) {
    case 8: GB_VARIANT(8); break;
    case 4: GB_VARIANT(4); break;
    case 2: GB_VARIANT(2); break;
    default: GB_VARIANT(1); break;
  }
#undef GB_VARIANT
#undef GB_LAUNCH
  ISHAP_CHECK_HIP(hipGetLastError());
  return 0;
}
