// Shared declarations for libishap_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <cstdlib>
#include <string>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// thread-local last error, never throw across the C ABI
void ishap_set_error(const std::string& msg);

#define ISHAP_CHECK_HIP(expr)                                                              \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      ishap_set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " +         \
                      __FILE__ + ":" + std::to_string(__LINE__));                          \
      return -1;                                                                           \
    }                                                                                      \
  } while (0)

#define ISHAP_REQUIRE(cond, msg)                                                           \
  do {                                                                                     \
    if (!(cond)) {                                                                         \
      ishap_set_error(std::string("requirement failed: ") + #cond + " -- " + (msg) +       \
                      " at " + __FILE__ + ":" + std::to_string(__LINE__));                 \
      return -2;                                                                           \
    }                                                                                      \
  } while (0)

// Device status word: ONE process-wide 32-bit word in pinned, device-mapped host memory.  A kernel that cannot go on
// correctly (a bounded spin that gave up) stores a non-zero ISHAP_DEV_* code there with a system-scope store and poisons
// its own outputs; every ABI entry point that enqueues work on a model context reads the word first (a plain host load,
// no synchronisation) and fails with that code's message, so an earlier launch's failure surfaces at the next call --
// the asynchronous-error contract of the HIP runtime itself.  ishap_device_status() reads (and clears) it on demand.
enum { ISHAP_DEV_OK = 0, ISHAP_DEV_GN_RENDEZVOUS = 1, ISHAP_DEV_CHAIN_TIMEOUT = 2 };
// Flags of the events that order the library's own streams / sequences among each other on ONE device (fork and join of the
// overlapped forward tail, the rendezvous tenancy's closing event): no timing, and no system-scope fence when the event completes --
// nothing on the host reads memory behind these events, and the default fence is a cache write-back + invalidate in front of
// whatever the stream runs next (hip_runtime_api.h, hipEventDisableSystemFence).  ISHAP_EVENT_FENCE=1 restores the default.
unsigned ishap_event_flags();
unsigned* ishap_status_word();        // null only if the pinned allocation failed
int ishap_check_status();             // 0, or -3 with the error string set (the word is cleared once reported)
int ishap_cu_count();                 // compute units of the current device (cached per device)

// Rendezvous tenancy (process-wide, per device).  A kernel whose workgroups wait for each other inside one launch (the
// group-local GroupNorm kernels with parts > 1) is only safe while no OTHER grid of that kind can be half-resident beside it:
// two such grids could each hold compute units the other needs.  One (owner, stream) pair per device may therefore launch
// rendezvous grids at a time: a call sequence (a UNet forward / backward, a stand-alone GroupNorm call) asks at its start
// and is told `false` -- it then runs the same kernels with ONE workgroup per group, which wait for nobody -- while another
// context, thread or stream of this process has such a sequence open or still in flight on the device (its closing event
// has not completed).  Launches that never wait (everything else in the library) need no tenancy and cannot deadlock a
// tenant: they finish by themselves and free their compute units.  What this cannot see: other PROCESSES on the same GPU
// and compute-unit-masked streams; there the bounded spin + status word (above) turn a starved rendezvous into an error
// (include/ishap.h, "Tenancy").
bool ishap_rendezvous_begin(const void* owner, hipStream_t s);   // true: this sequence may use in-launch rendezvous
void ishap_rendezvous_end(const void* owner, hipStream_t s, bool granted);
bool ishap_rendezvous_contended(const void* owner, hipStream_t s);   // reads and clears "someone else asked while I held the device"   // closes the sequence (records its event on s)

#define ISHAP_TRY(expr)        \
  do {                         \
    int _r = (expr);           \
    if (_r != 0) return _r;    \
  } while (0)

// Raise a kernel's dynamic-LDS limit once per (thread, device, kernel symbol).  The cache is thread-local (no shared
// mutable state, so concurrent callers need no lock) and keyed by the CURRENT device: a process that drives a second GPU,
// or a worker thread that switches devices, sets the attribute there too instead of launching with the 64 KB default.
static inline int ishap_set_max_lds(const void* kern, int bytes) {
  struct Done { const void* k; int dev; int bytes; };
  static thread_local Done done[192];
  static thread_local int ndone = 0;
  int dev = 0;
  ISHAP_CHECK_HIP(hipGetDevice(&dev));
  for (int i = 0; i < ndone; ++i)
    if (done[i].k == kern && done[i].dev == dev && done[i].bytes >= bytes) return 0;
  ISHAP_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (ndone < 192) done[ndone++] = Done{kern, dev, bytes};
  return 0;
}

// 16-byte output store of a streaming kernel's result.  ISHAP_WT_STORES=1 (compile time): write-through (`sc1`), so that the bytes
// leave for memory while the kernel runs instead of sitting dirty in the XCD's L2 until the end-of-kernel write-back (a
// dependent boundary costs + B / 6 TB/s behind B dirty bytes, MI355X_MICROARCH.md price list, row `boundary`; the next kernel
// reads them from the memory side either way: its L2 starts invalidated).
#ifndef ISHAP_WT_STORES
#define ISHAP_WT_STORES 0
#endif
#if defined(__HIPCC__)
__device__ __forceinline__ void store_out16(half_t* p, const half8& v) {
#if ISHAP_WT_STORES
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<half8*>(p) = v;
#endif
}
#endif

#define STAT_SCALE_SUM 16777216.f      /* 2^24 */
#define STAT_SCALE_SQ 1048576.f        /* 2^20 */
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------
// implicit GEMM (igemm.hip):  out[m][n] = alpha * sum_k X(m,k) * Wt[n][k]  (+bias[n]) (+res[m][n])
//   X is either a row-major matrix (MODE_GEMM) or an NHWC feature map gathered as 3x3 patches
//   with zero padding (MODE_CONV3, k = tap*Cin + c).  fp16 operands, fp32 accumulation on MFMA.
// ---------------------------------------------------------------------------------------
enum { IG_OUT_F16 = 0, IG_OUT_F32 = 1, IG_OUT_NCHW_F32 = 2 };

// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: four v_add_f32 with DPP operands
// (quad swaps, half-row mirror, row mirror) instead of four ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);    // row_half_mirror
  v += dpp_move<0x140>(v);    // row_mirror
  return v;
}

// The same for a double (its two 32-bit halves move together), then the four row sums meet through v_readlane: the sum over
// the 64 lanes of a wave, bitwise the same value in every lane, in ~20 instructions -- a __shfl_xor butterfly on doubles is
// 24 ds_bpermute round trips in a dependent chain (~900 cycles of the latency-bound GroupNorm kernels).
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_move_f64<0xB1>(v);
  v += dpp_move_f64<0x4E>(v);
  v += dpp_move_f64<0x141>(v);
  v += dpp_move_f64<0x140>(v);
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// sum over aligned groups of 8 lanes (quad swaps + half-row mirror), 64-bit integer and float forms
template <int CTRL>
__device__ __forceinline__ long long dpp_move_i64(long long v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(v & 0xffffffffll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, 0xF, 0xF, true);
  return ((long long)hi << 32) | (unsigned int)lo;
}
__device__ __forceinline__ long long lanes8_sum(long long v) {
  v += dpp_move_i64<0xB1>(v);
  v += dpp_move_i64<0x4E>(v);
  v += dpp_move_i64<0x141>(v);
  return v;
}
__device__ __forceinline__ float lanes8_sum(float v) {
  v += dpp_move<0xB1>(v);
  v += dpp_move<0x4E>(v);
  v += dpp_move<0x141>(v);
  return v;
}

__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v));
  v = fmaxf(v, dpp_move<0x4E>(v));
  v = fmaxf(v, dpp_move<0x141>(v));
  v = fmaxf(v, dpp_move<0x140>(v));
  return v;
}

// A convolution result that has not been materialised yet: the split-K slices of an implicit-GEMM launch left as fp32
// partial sums [nslab][rows][ld].  The next kernel on the tensor (a GroupNorm pass, forward or backward -- it has to read
// the whole tensor anyway) adds the slices in slice order, applies bias / residual, rounds to fp16 and goes on from
// there, so the separate reduce launch and the fp32 round trip it ends with disappear.
struct SlabSrc {
  const float* ws = nullptr;
  int nslab = 0;                 // 0: not pending
  long long zstride = 0;         // floats between slices (= rows * ld)
  const float* bias = nullptr;
  const float* bias2 = nullptr;
  const half_t* res = nullptr;   // residual [rows][ldr], optionally at half resolution (res_ups)
  int ldr = 0, res_ups = 0;
  __host__ __device__ bool pending() const { return nslab > 0; }
};

// The arguments an LDS-DMA convolution kernel needs before it can issue its first load, as its FIRST kernel parameter: 14
// dwords of plain scalars, which the build's -amdgpu-kernarg-preload-count=14 has the command processor place in SGPRs at
// dispatch.  A kernel otherwise starts by waiting ~430 cycles (0.19 us, tools/experiments/kernarg_probe.hip) for the s_load of its
// argument block; the rest of IgemmArgs (epilogue operands, second source, batch strides) arrives under the ring fill.
struct IgemmHot {
  const half_t* X;
  const half_t* Wt;
  int K, Cin, ldx, ldw, H, W, ksplit, nwg;
  unsigned packed;             // w_shift | hw_shift << 6 | nx_shift << 12 | ny_shift << 18 (6-bit two's complement, -1 = none) | ups << 24
  __host__ __device__ static int sx6(unsigned v) { return (int)(v & 0x3f) - (int)((v & 0x20) << 1); }
  __host__ __device__ int w_shift() const { return sx6(packed); }
  __host__ __device__ int hw_shift() const { return sx6(packed >> 6); }
  __host__ __device__ int nx_shift() const { return sx6(packed >> 12); }
  __host__ __device__ int ny_shift() const { return sx6(packed >> 18); }
  __host__ __device__ bool ups() const { return (packed >> 24) & 1u; }
};

struct IgemmArgs {
  const half_t* X = nullptr;   // activations
  const half_t* Wt = nullptr;  // [Npad][K] (K contiguous), rows >= N are zero
  void* out = nullptr;
  const float* bias = nullptr;   // [N] or null
  const float* bias2 = nullptr;  // second bias added like the first (a 1x1 skip convolution folded into this launch)
  const half_t* res = nullptr;   // residual [M][ldr] or null (may alias out)
  float* ws = nullptr;           // split-K workspace [ksplit][nbatch][M][N] fp32
  int M = 0, N = 0, K = 0;
  int conv3 = 0;                 // 0: plain rows, 1: 3x3 gather
  // 3x3 launches may carry a second activation source: after the 9*Cin gathered columns, K2 more columns are read
  // from X2 (plain rows, stride ldx2) against weight columns [9*Cin, 9*Cin + K2) -- y = conv3x3(X) + conv1x1(X2) in
  // one K loop (ResBlock: out_layers conv + skip_connection, unet.py:255-256).  K = 9*Cin + K2.
  const half_t* X2 = nullptr;
  int ldx2 = 0, K2 = 0;
  int Cin = 0;                   // channels per tap (conv3) -- K = 9*Cin
  int ldx = 0, ldw = 0, ldo = 0, ldr = 0;
  long long bsx = 0, bsw = 0, bso = 0;   // per-batch strides (elements)
  int nbatch = 1;
  int H = 0, W = 0;              // OUTPUT spatial size (conv3)
  int nx_shift = -1, ny_shift = -1;  // log2 of the grid's x / y extent when both are powers of two (set by the LDS-DMA launcher): tile remap by shifts
  int w_shift = -1, hw_shift = -1;   // log2(W), log2(H*W) when both are powers of two (set by the launcher): the kernels then
                                     // split a pixel index with shifts instead of ~50-instruction integer divisions
  int ups = 0;                   // conv3 source map is (H/2, W/2): nearest-neighbour upsample on the fly
  int res_ups = 0;               // residual map is (H/2, W/2)
  int ksplit = 1;
  // LAYOUT MATTERS HERE (round 6, tools/kernel_meta.py, tests/test_host_cpu.py::test_hot_kernels_use_no_scratch): these three dwords
  // (once lite / force_small / chunk_tiles_big) keep {alpha, out_mode, stat_out} OFF a 16-byte boundary of the kernarg block.
  // Without them the compiler fetched those four dwords with one s_load_dwordx4 into a PRIVATE copy of the argument block --
  // s_load + s_waitcnt + scratch_store at kernel entry, scratch_load in the epilogue, 32 bytes of scratch per lane -- and every
  // LDS-DMA convolution launch was 0.5-1.1 us slower (+3 % per edit, profiles/round6_ab_prune_scratch.txt)
  int reserved_[3] = {0, 0, 0};
  int chunk_tiles = 0;           // > 0 (a multiple of 8): igemm4 runs the layer as several launches of at most that many tiles (igemm4.hip, launch4)
  int defer_reduce = 0;          // ksplit > 1: leave the fp32 slices in `ws` (no reduce launch); the caller hands a SlabSrc to the consumer
  float alpha = 1.f;
  int out_mode = IG_OUT_F16;
  long long* stat_out = nullptr; // optional [N_img][N][2]: += per-channel (sum, sum of squares) of the fp16 outputs, as 64-bit
                                 // fixed point (STAT_SCALE_*): integer atomics commute, so the statistics are bitwise reproducible
  float flops_scale = 1.f;       // algorithmic / executed FLOPs (1/3 for the hi-lo split head conv)
  // Fused GroupNorm-backward sums (input-gradient launches whose output is the gradient arriving at act(film(GN(x)))):
  // the epilogue adds per-channel sum(dyh), sum(dyh*xhat) of its fp16 outputs to gb_csums [N_img][N][2] (fixed point,
  // STAT_SCALE_SUM both), so norm_bwd.hip's partial pass is not needed.  Exclusive with stat_out.
  const half_t* gb_x = nullptr;      // GN input [M][N] (row stride N)
  const float* gb_stats = nullptr;   // forward (mean, rstd) [N_img][32][2]
  const float* gb_gamma = nullptr;
  const float* gb_beta = nullptr;
  const float* gb_emb = nullptr;     // FiLM rows (scale at c, shift at N + c), per image stride gb_emb_ld
  int gb_emb_ld = 0;
  int gb_film = 0, gb_act = 0;
  long long* gb_csums = nullptr;
};
// set (non-null) by igemm.hip around a launch while ishap_profile_begin/end is active: the kernel launchers then attach
// these events to the dispatch itself (hipExtLaunchKernelGGL), so their elapsed time is the kernel's own duration
extern hipEvent_t g_igemm_prof_start, g_igemm_prof_stop;
int igemm_launch(const IgemmArgs& a, hipStream_t s);
int igemm_reduce_launch(const IgemmArgs& a, hipStream_t s);
int igemm4_small_map_slices(const IgemmArgs& a);  // igemm4.hip: K slices of its sliced launch on an 8x8 map (consumer adds them up), 0 = not taken
// picks a split so the grid fills the chip; returns workspace floats needed
int igemm_pick_ksplit(int M, int N, int K, int nbatch, bool pending = false);   // pending: the consumer adds the slices up (no reduce launch)
