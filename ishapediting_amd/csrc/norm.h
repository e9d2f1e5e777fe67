#pragma once
#include "common.h"

static inline int gn_rows_per_block(int HW) {
  int r = HW / 256;
  if (r < 8) r = 8;
  if (r > HW) r = HW;
  return r;
}
static inline size_t gn_partial_floats(int N, int HW, int C) {
  return (size_t)N * (HW / gn_rows_per_block(HW)) * C * 2 * 2;    // (sum, sum of squares) as doubles = 2 floats each
}
// stats[n][32][2] = (mean, rstd); partial is scratch of gn_partial_floats()
int gn_stats_launch(const half_t* x, float* partial, float* stats, int N, int HW, int C, hipStream_t s);

struct GnApplyArgs {
  const half_t* x = nullptr;     // [N][H*W][C]
  half_t* out = nullptr;         // [N][HWo][C]  (or [N][HW][3C] when split)
  half_t* xpool = nullptr;       // pool: pooled raw x (may be null)
  const float* stats = nullptr;  // [N][32][2] (mean, rstd) -- used when `sums` is null
  const long long* sums = nullptr;  // [N][C][2] per-channel (sum, sum of squares), 64-bit fixed point, from the producer's epilogue
  float* stats_out = nullptr;    // with `sums`: block 0 stores the finalised (mean, rstd) here for the backward pass
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* emb = nullptr;    // film: per image (scale[C] | shift[C]) fp32, images emb_ld floats apart
  int emb_ld = 0;
  int N = 1, H = 0, W = 0, C = 0;
  int film = 0, act = 1, pool = 0, split = 0;
  // the input is the channel concatenation [x | x2] of two dense tensors (a skip connection): channels < csplit come
  // from x (row stride csplit, sums `sums`), the rest from x2 (row stride C - csplit, sums `sums2`); the raw
  // concatenation is also written to xcopy [N][H*W][C] for the consumers that need it.  Plain variant only (no pool/split).
  const half_t* x2 = nullptr;
  const long long* sums2 = nullptr;
  half_t* xcopy = nullptr;
  int csplit = 0;
  int main_blocks = 0;           // set by the launcher: the grid size (a preloaded scalar in the kernel)
};
int gn_apply_launch(const GnApplyArgs& a, hipStream_t s);

// ---- backward of  y = act(film(gn(x)))  w.r.t. x (no parameter gradients: the path never needs them) ----
// g is the gradient arriving at y (possibly at another resolution, see gmode); result
//   dx = rstd * (dyh - mean_g(dyh) - xh * mean_g(dyh * xh)) + add,   dyh = g * act'(.) * (1+scale) * gamma
enum { GB_SAME = 0, GB_UNPOOL = 1, GB_SUM4 = 2 };
struct GnBwdArgs {
  const half_t* g = nullptr;      // upstream gradient: [N][H*W][C] (SAME), [N][H/2*W/2][C] (UNPOOL: each /4), [N][2H*2W][C] (SUM4)
  const half_t* x = nullptr;      // GN input [N][H*W][C]
  const half_t* add = nullptr;    // optional extra gradient added to dx, same indexing mode as g
  const half_t* add2 = nullptr;   // optional second addend at the OUTPUT resolution, dense [N][H*W][C] (a skip-connection gradient)
  half_t* dx = nullptr;           // [N][H*W][C], or the first `csplit` channels when the output is split
  half_t* dx2 = nullptr;          // channels csplit..C-1 as their own dense tensor (the two halves of a skip concatenation)
  int csplit = 0;                 // 0: one dense output; else a multiple of 8
  const float* stats = nullptr;   // forward (mean, rstd) [N][32][2]
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* emb = nullptr;     // film
  int emb_ld = 0;
  long long* csums = nullptr;     // zeroed scratch [N][C][2]: per-channel (sum dyh, sum dyh*xh) in 64-bit fixed point
  int N = 1, H = 0, W = 0, C = 0;
  int film = 0, act = 1, gmode = GB_SAME;
  int sums_ready = 0;             // csums were already accumulated by the producing implicit-GEMM epilogue: apply pass only
  int main_blocks = 0;
};
int gn_backward_launch(const GnBwdArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------------
// Group-local GroupNorm kernels (norm_local.hip) for maps of at most 32 x 32 pixels: one workgroup owns one
// (image, group) -- H*W pixels x C/32 channels, staged in LDS -- so statistics, normalisation, FiLM, SiLU, pooling and
// (backward) the group means are ONE launch with no atomics, and the input may be a pending split-K result (SlabSrc).
// ---------------------------------------------------------------------------------------------------------------
constexpr int GN_REC_HALF = 32;       // granules of one copy of a group's record (up to 8 parts x 4 granules)
constexpr int GN_REC_STRIDE = 64;     // 8-byte granules per (image, group): the agent-scope copy, then the XCD-local copy (norm_local.hip)
constexpr int GN_REC_PER_PART = 4;    // (hi, lo) fp32 pair for each of a part's two sums: up to 8 parts
constexpr int GN_SPIN_LIMIT = 1 << 22; // polls before a rendezvous gives up (an error, see ISHAP_DEV_GN_RENDEZVOUS)

struct GnLocalArgs {
  // source A: channels [0, Ca) -- a dense fp16 tensor [N][H*W][Ca] or pending fp32 slices (then `ya` receives the fp16 tensor)
  const half_t* xa = nullptr;
  SlabSrc slab;
  half_t* ya = nullptr;
  int Ca = 0;
  // source B: channels [Ca, C) of a skip concatenation (dense fp16 [N][H*W][C-Ca]) or null when Ca == C
  const half_t* xb = nullptr;
  half_t* xcopy = nullptr;       // the raw concatenation [N][H*W][C] (written when non-null)
  half_t* out = nullptr;         // act(film(gn(x))) [N][HWo][C]
  half_t* xpool = nullptr;       // pool: 2x2 mean of the raw input (may be null)
  float* stats_out = nullptr;    // [N][32][2] (mean, rstd) for the backward pass
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* emb = nullptr;
  int emb_ld = 0;
  int N = 1, H = 0, W = 0, C = 0;
  int film = 0, act = 1, pool = 0;
  // optional zeroed scratch [N][32][16] x 64 bit: lets up to 8 workgroups share one (image, group) (they exchange their
  // partial sums as data-tagged granules); null = one workgroup per group.  `parts` is set by the launcher.
  unsigned long long* rec = nullptr;
  int parts = 1;
  unsigned* status = nullptr;    // device status word (set by the launcher)
  int spin_limit = GN_SPIN_LIMIT;
};
bool gn_local_fits(int HW, int C);              // LDS budget of the forward / backward staging
int gn_local_parts(int N, int HW, int C);       // workgroups per (image, group) the launcher picks when a rendezvous record is given
int gn_local_launch(const GnLocalArgs& a, hipStream_t s);

struct GnBwdLocalArgs {
  const half_t* g = nullptr;     // upstream gradient (dense fp16) or pending slices, at the resolution `gmode` says
  SlabSrc slab;                  // bias / residual unused: an input gradient has neither
  const half_t* x = nullptr;
  const half_t* add = nullptr;   // same indexing mode as g
  const half_t* add2 = nullptr;  // output resolution
  half_t* dx = nullptr;
  half_t* dx2 = nullptr;
  int csplit = 0;
  const float* stats = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* emb = nullptr;
  int emb_ld = 0;
  int N = 1, H = 0, W = 0, C = 0;
  int film = 0, act = 1, gmode = GB_SAME;
  unsigned long long* rec = nullptr;   // as in GnLocalArgs
  int parts = 1;
  unsigned* status = nullptr;
  int spin_limit = GN_SPIN_LIMIT;
};
bool gn_bwd_local_fits(int HW, int C, int gmode);
int gn_bwd_local_launch(const GnBwdLocalArgs& a, hipStream_t s);
