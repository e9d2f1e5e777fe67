// UNet graph + executor (host side).  Structure mirrors guided_diffusion/unet.py:427-616.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/ishap.h"
#include "common.h"

struct Tensor {
  half_t* p = nullptr;
  int N = 0, H = 0, W = 0, C = 0;
  long long* sums = nullptr;   // [N][C][2] per-channel (sum, sum of squares), fixed point, gathered by the producer; or null
  // a skip concatenation that has not been materialised yet: p is the buffer the first consumer (the ResBlock's GroupNorm
  // pass) fills while it reads the two halves (cat_a: first cat_ca channels with sums cat_sa; cat_b: the rest, sums cat_sb)
  const half_t* cat_a = nullptr; const half_t* cat_b = nullptr; const long long* cat_sa = nullptr; const long long* cat_sb = nullptr;
  int cat_ca = 0;
  // the tensor is still the split-K slices of its producing convolution (common.h SlabSrc): `p` is where the fp16 values go
  // once the next GroupNorm pass (or slab_materialize) has added them up; cat_pend: the same for cat_a of a lazy concatenation
  SlabSrc pend, cat_pend;
  long long rows() const { return (long long)N * H * W; }
  long long numel() const { return rows() * C; }
};

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, high = 0;
  bool dry = false;
  void reset() { off = 0; }
  void* alloc(size_t bytes) {
    size_t o = align_up(off, 256);
    off = o + bytes;
    if (off > high) high = off;
    if (dry) return (void*)(uintptr_t)(0x1000 + o);
    if (off > cap) return nullptr;
    return base + o;
  }
};

struct ConvW {
  std::string path;
  int cin = 0, cout = 0, taps = 9;
  int kpad = 0;        // channels per tap in the packed forward operand
  half_t* w = nullptr;   // [rows_pad(cout)][taps*kpad]
  half_t* wT = nullptr;  // [rows_pad(cin)][taps*cout_pad]  (input-gradient operand)
  int cout_pad = 0;
  float* bias = nullptr;
  bool split = false;    // fp32 head: [hi|hi|lo] packing, kpad = 3*cin
  // K-concatenated forward operand shared with another conv of the same ResBlock ([c2 3x3 | skip 1x1]): this weight is
  // also packed into cat at column cat_off (row stride cat_ld)
  half_t* cat = nullptr;
  int cat_ld = 0, cat_off = 0;
};
struct NormW {
  float* gamma = nullptr;
  float* beta = nullptr;
  int C = 0;
};

struct ResSaved {   // what the backward pass re-reads
  Tensor x, h1, xs;     // block input, conv1 output, (pooled / plain) skip input
  float* stats1 = nullptr;
  float* stats2 = nullptr;
};
struct AttnSaved {
  Tensor x, qkv, a;
  float* stats = nullptr;
  float* lse = nullptr;
  half_t* P = nullptr;    // softmax probabilities [N*heads][T][T] kept for the backward (null: recompute)
};

struct ResL {
  std::string path;
  int cin = 0, cout = 0;
  bool up = false, down = false;
  NormW n1, n2;
  ConvW c1, c2, skip;
  bool has_skip = false;
  int emb_off = 0;     // row offset into the concatenated emb_layers matrix (2*cout rows)
  ResSaved sv;
};
struct AttnL {
  std::string path;
  int C = 0, heads = 0;
  NormW n;
  ConvW qkv, proj;
  AttnSaved sv;
};
struct LayerRef { int kind; int idx; };   // 0 conv(stem) 1 res 2 attn
struct BlockL {
  std::string name;
  std::vector<LayerRef> layers;
  int cin = 0, cout = 0, res_in = 0, res_out = 0, skip_ch = 0;
  Tensor out;       // block output of the last forward
  Tensor cat;       // concat input (output blocks)
};

struct ParamSlot {
  std::string name;
  int ndim = 0;
  long long shape[4] = {0, 0, 0, 0};
  int kind = 0;     // 0 conv weight, 1 conv bias, 2 plain fp32 vector/matrix copy, 3 emb weight, 4 emb bias
  ConvW* conv = nullptr;
  float* dst = nullptr;
  long long dst_off = 0;
  bool loaded = false;
  long long numel() const {
    long long n = 1;
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
  }
};

struct ishap_unet {
  ishap_unet_config cfg;
  int device = 0;
  int ted = 0;                 // time_embed_dim
  int in_pad = 0;              // padded input channels of the stem
  ConvW stem, head;
  NormW head_norm;
  std::vector<ResL> res;
  std::vector<AttnL> attn;
  std::vector<BlockL> in_blocks, out_blocks;
  BlockL mid;
  int final_ch = 0;
  // embeddings
  float *te_w0 = nullptr, *te_b0 = nullptr, *te_w2 = nullptr, *te_b2 = nullptr;
  float *emb_w = nullptr, *emb_b = nullptr;   // concatenated emb_layers [film_rows][ted]
  int film_rows = 0;
  float *d_temb = nullptr, *d_e1 = nullptr, *d_emb = nullptr, *d_film = nullptr;
  // FiLM rows of timesteps prepared ahead of a sampling loop (ishap_unet_prepare_timesteps): a forward whose timesteps
  // all equal one prepared value reads its row (stride 0 over the batch) and skips the four embedding launches, among
  // them the GEMV that streams the 168 MB of emb_layers weights
  float* film_cache = nullptr;
  size_t film_cache_rows = 0;
  std::vector<float> film_cache_ts;
  float *pc_temb = nullptr, *pc_e1 = nullptr, *pc_emb = nullptr;     // 16-row scratch of the prepare call
  const float* film_cur = nullptr;   // what the current forward / backward reads: d_film (stride film_rows) or a cache row (stride 0)
  int film_cur_ld = 0;
  // parameter table
  std::vector<ParamSlot> params;
  std::map<std::string, int> param_index;
  int n_loaded = 0;
  // memory
  Arena arena;
  float* ws = nullptr;          size_t ws_floats = 0;       // split-K partials
  float* gn_partial = nullptr;  size_t gn_partial_floats = 0;
  // last forward
  int last_N = 0, last_feat = -1;
  bool have_saved = false;
  Tensor tap;
  Tensor x0, h_final;
  float* head_stats = nullptr;
  long long* stat_base = nullptr;   // arena of per-channel GroupNorm sums, zeroed once per forward
  size_t stat_cap = 0, stat_off = 0, stat_high = 0, stat_fwd_mark = 0;
  int bwd_since_fwd = 0;
  size_t fwd_mark = 0;          // arena offset after the forward (backward scratch goes above it)
  // Overlapped forward tail (keep_for_backward bit 1): the output blocks after the tap and the head are enqueued on a
  // stream of the context's own, forked from the caller's stream at the tap, so that they run beside the loss and the
  // backward pass the caller enqueues next (those need nothing after the tap).  ishap_unet_join_tail orders a stream
  // behind them; the next forward / a full-depth backward / a block read-out joins by itself.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_tail = nullptr, ev_mid = nullptr;
  // diagnostic time marks (ISHAP_BWD_MARKS=1; include/ishap.h, ishap_unet_marks): timing events after every block of the backward
  // pass and around the deferred forward tail on the side stream -- un-traced per-segment durations of the overlapped step
  bool marks_on = false;
  std::vector<hipEvent_t> marks;
  std::vector<int> mark_tags;
  int marks_n = 0;
  hipEvent_t mark_tail_begin = nullptr, mark_tail_end = nullptr;
  bool mark_tail_set = false;
  bool tail_pending = false;
  // deferred tail (ISHAP_TAIL_DEFER): planned by the forward, enqueued by ishap_unet_run_tail behind ev_mid (recorded by the
  // backward after its first output blocks)
  struct TailState { size_t split; Tensor h; std::vector<Tensor> hs; int N; float* out; size_t arena_off, stat_off; bool keep; };
  TailState tail{};
  bool tail_deferred = false, mid_recorded = false;
  float* ws_side = nullptr;          // the side stream's own copies of the shared scratch buffers
  float* gn_partial_side = nullptr;
  float* attn_D = nullptr;      // backward attention row sums
  size_t attn_D_floats = 0;
};

struct Exec {
  ishap_unet* u;
  hipStream_t s;
  bool dry;
  bool keep = false;   // the forward keeps what a following backward re-reads
  float* ws = nullptr;          // split-K / GroupNorm-statistics scratch of this launch sequence (null: the context's)
  float* gn_partial = nullptr;
  int chunk_tiles = 0;          // > 0: the dx-reuse convolutions of this sequence run as launches of at most that many tiles (the overlapped forward tail)
  bool tenant = true;           // this sequence holds the device's rendezvous tenancy (common.h ishap_rendezvous_begin)
};
// RAII around a launch sequence: asks for the tenancy at construction, closes it (event on the stream) at scope exit
struct TenancyScope {
  const void* owner; hipStream_t s; bool granted;
  TenancyScope(const void* o, hipStream_t st, bool dry) : owner(o), s(st), granted(dry ? true : ishap_rendezvous_begin(o, st)), dry_(dry) {}
  ~TenancyScope() { if (!dry_) ishap_rendezvous_end(owner, s, granted); }
  TenancyScope(const TenancyScope&) = delete;
  TenancyScope& operator=(const TenancyScope&) = delete;
 private:
  bool dry_;
};
// Arena allocation that FAILS THE CALL when the arena sized by the create-time dry runs is exceeded (a null pointer
// must never reach a kernel): ISHAP_ALLOC(ptr, e, count) inside any function returning an int status.
template <typename T>
static inline int aalloc_checked(Exec& e, size_t count, T** out) {
  *out = (T*)e.u->arena.alloc(count * sizeof(T));
  ISHAP_REQUIRE(*out != nullptr, "activation arena exhausted (shape beyond what ishap_unet_create sized for)");
  return 0;
}
#define ISHAP_ALLOC(ptr, e, count) ISHAP_TRY(aalloc_checked((e), (size_t)(count), &(ptr)))
#define ISHAP_SALLOC(ptr, e, count)                                                                   \
  do {                                                                                                \
    (ptr) = salloc((e), (size_t)(count));                                                             \
    ISHAP_REQUIRE((ptr) != nullptr, "GroupNorm statistics arena exhausted");                          \
  } while (0)
// X [N,H,W,ldx] (*) Wt -> out; taps 9 (3x3, pad 1) or 1; picks split-K and uses the context workspace
// pend_out: the caller's consumer can read split-K slices (a group-local GroupNorm pass): when the launch splits K, the
// slices stay in an arena buffer described by *pend_out and no reduce kernel runs (fp16 dense outputs only)
int conv_op(Exec& e, const half_t* X, int N, int H, int W, int ldx, const half_t* Wt, int kpad, int taps, int cout,
            const float* bias, const half_t* res, int ldr, void* out, int ldo, int out_mode, int ups, int res_ups,
            long long* stat_out = nullptr, const struct GnBwdArgs* gb = nullptr, const half_t* X2 = nullptr, int ldx2 = 0,
            int K2 = 0, const float* bias2 = nullptr, int ldw = 0, SlabSrc* pend_out = nullptr);
// small maps (<= 32 x 32): GroupNorm passes run group-local (norm_local.hip), producers gather no statistics
int unet_join_tail(ishap_unet* u, hipStream_t s);
bool small_map(int HW);
bool exec_is_solo(const Exec& e);   // no other stream of this context has work in flight (in-launch rendezvous allowed)
bool local_gn(int HW, int C);
int slab_materialize(Exec& e, Tensor& t);    // add up a pending tensor with the stand-alone reduce kernel (consumers that cannot)
long long* salloc(Exec& e, size_t count);   // from the stats arena
int gn_stats_op(Exec& e, const Tensor& x, float* stats);

int unet_build(ishap_unet* u);
int unet_forward_impl(ishap_unet* u, const float* x, const float* ts, int N, int feat_layer, float* out,
                      void* inter_feat, int keep, hipStream_t s, bool dry);
int unet_backward_impl(ishap_unet* u, const half_t* cot_tap, const void* cot_out, int cot_out_f16, const float* scale2,
                       float* dx, hipStream_t s, bool dry);
