// UNet input-gradient pass: d(sum(tap * cot)) / dx, hand-derived, layer by layer in reverse.
// Reference: loss.backward() at drag_utils.py:383 (autograd through guided_diffusion/unet.py:634-671 from
// the tapped output block back to x) and :458 (full depth, from the model output).  Only what the path
// needs is computed: gradients w.r.t. activations.  Weight gradients (which the reference computes and
// throws away, its parameters keep requires_grad=True) and the attention checkpoint's second forward
// (unet.py:297) have no counterpart here: activations needed for derivatives are recomputed inside the
// elementwise kernels from the saved GN inputs and statistics.
// Every convolution's input gradient is the same implicit-GEMM kernel run on the gradient map with the
// tap-flipped, transposed weight operand packed at load time (ConvW::wT).
#include "unet.h"

#include "attention.h"
#include "misc.h"
#include "norm.h"

static int gn_bwd_op(Exec& e, GnBwdArgs g) {
  if (!g.sums_ready) ISHAP_SALLOC(g.csums, e, (size_t)g.N * g.C * 2);     // zeroed with the rest of the stats arena at the start of the forward
  if (e.dry) return 0;
  ISHAP_REQUIRE(g.csums != nullptr, "stats arena exhausted");
  return gn_backward_launch(g, e.s);
}

// dY [N,H,W,cout] -> dX [N,H,W,rows of wT] through the transposed / flipped operand
// `gb`: the GroupNorm whose activation this gradient arrives at, at the same resolution (GB_SAME) -- its per-channel
// backward sums are then accumulated in this launch's epilogue (gb->csums allocated here, gb->sums_ready set).
// `may_pend`: the only reader of dx_out is a group-local GroupNorm-backward pass, which adds up split-K slices itself.
static int dgrad_op(Exec& e, const ConvW& c, const Tensor& dy, Tensor& dx_out, int n_out, GnBwdArgs* gb = nullptr,
                    bool may_pend = false) {
  dx_out = Tensor{nullptr, dy.N, dy.H, dy.W, n_out};
  ISHAP_ALLOC(dx_out.p, e, dx_out.numel());
  if (gb) {
    ISHAP_SALLOC(gb->csums, e, (size_t)gb->N * gb->C * 2);
    gb->sums_ready = 1;
    ISHAP_REQUIRE(gb->C == n_out && gb->gmode == GB_SAME, "fused GroupNorm-backward sums need the gradient at the GN resolution");
  }
  return conv_op(e, dy.p, dy.N, dy.H, dy.W, dy.C, c.wT, c.cout_pad, c.taps, n_out, nullptr, nullptr, 0, dx_out.p, n_out,
                 IG_OUT_F16, 0, 0, nullptr, gb, nullptr, 0, 0, nullptr, 0, may_pend ? &dx_out.pend : nullptr);
}

// group-local input gradient of act(film(GN(x))) on a small map; `up` may still be pending (norm_local.hip)
static int gn_bwd_local_op(Exec& e, const GnBwdArgs& g, Tensor& up) {
  GnBwdLocalArgs a;
  a.g = up.p; a.slab = up.pend; a.x = g.x; a.add = g.add; a.add2 = g.add2; a.dx = g.dx; a.dx2 = g.dx2; a.csplit = g.csplit;
  a.stats = g.stats; a.gamma = g.gamma; a.beta = g.beta; a.emb = g.emb; a.emb_ld = g.emb_ld;
  a.N = g.N; a.H = g.H; a.W = g.W; a.C = g.C; a.film = g.film; a.act = g.act; a.gmode = g.gmode;
  long long* rec = nullptr;
  ISHAP_SALLOC(rec, e, (size_t)g.N * 32 * GN_REC_STRIDE);       // zeroed with the rest of the statistics arena
  a.rec = exec_is_solo(e) ? reinterpret_cast<unsigned long long*>(rec) : nullptr;     // see gn_local_op
  up.pend = SlabSrc{};
  if (e.dry) return 0;
  return gn_bwd_local_launch(a, e.s);
}
static bool local_gn_bwd(int HW, int C, int gmode) { return small_map(HW) && gn_bwd_local_fits(HW, C, gmode); }

// `split` > 0: the block input was a skip concatenation [h | skip]; its gradient is written as two dense tensors
// (dx = first `split` channels, *dx2 = the rest) so no slicing pass is needed afterwards.
// `add2`: a gradient map of the block input's shape (the skip-connection gradient of an input block) added to the result.
static int res_backward(Exec& e, ResL& L, const Tensor& dy, Tensor& dx, int split = 0, Tensor* dx2 = nullptr,
                        const half_t* add2 = nullptr) {
  ishap_unet* u = e.u;
  const ResSaved& sv = L.sv;
  const Tensor& x = sv.x;
  const Tensor& h1 = sv.h1;
  ISHAP_REQUIRE(dy.C == L.cout && dy.H == h1.H && dy.N == x.N, "ResBlock gradient shape");
  // out_layers: conv2 <- SiLU <- FiLM <- GN2
  Tensor dc;
  Tensor dh1 = h1;
  {
    GnBwdArgs g;
    g.x = h1.p; g.stats = sv.stats2; g.gamma = L.n2.gamma; g.beta = L.n2.beta;
    g.emb = u->film_cur + L.emb_off; g.emb_ld = u->film_cur_ld;
    g.N = h1.N; g.H = h1.H; g.W = h1.W; g.C = L.cout; g.film = 1; g.act = 1; g.gmode = GB_SAME;
    const bool loc = local_gn_bwd(h1.H * h1.W, L.cout, GB_SAME);
    ISHAP_TRY(dgrad_op(e, L.c2, dy, dc, L.cout, loc ? nullptr : &g, loc));
    ISHAP_ALLOC(dh1.p, e, h1.numel());
    g.g = dc.p; g.dx = dh1.p;
    if (loc) ISHAP_TRY(gn_bwd_local_op(e, g, dc));
    else ISHAP_TRY(gn_bwd_op(e, g));
  }
  // in_layers: conv1 <- (up/down sample) <- SiLU <- GN1
  GnBwdArgs g1;
  g1.x = x.p; g1.stats = sv.stats1; g1.gamma = L.n1.gamma; g1.beta = L.n1.beta;
  g1.N = x.N; g1.H = x.H; g1.W = x.W; g1.C = L.cin; g1.film = 0; g1.act = 1;
  g1.gmode = L.down ? GB_UNPOOL : (L.up ? GB_SUM4 : GB_SAME);
  const bool loc1 = local_gn_bwd(x.H * x.W, L.cin, g1.gmode);
  Tensor da;
  ISHAP_TRY(dgrad_op(e, L.c1, dh1, da, L.cin, (!loc1 && g1.gmode == GB_SAME) ? &g1 : nullptr, loc1));
  const half_t* add = dy.p;     // identity skip: gradient of `x_upd(x)` (unet.py:241,256)
  if (L.has_skip) {
    Tensor dxs;
    ISHAP_TRY(dgrad_op(e, L.skip, dy, dxs, L.cin));
    add = dxs.p;
  }
  dx = x;
  if (split > 0) {
    dx.C = split;
    *dx2 = x;
    dx2->C = x.C - split;
    dx2->sums = nullptr;
    ISHAP_ALLOC(dx.p, e, dx.numel());
    ISHAP_ALLOC(dx2->p, e, dx2->numel());
    g1.dx2 = dx2->p; g1.csplit = split;
  } else {
    ISHAP_ALLOC(dx.p, e, x.numel());
  }
  dx.sums = nullptr;
  g1.g = da.p; g1.add = add; g1.dx = dx.p; g1.add2 = add2;
  if (loc1) ISHAP_TRY(gn_bwd_local_op(e, g1, da));
  else ISHAP_TRY(gn_bwd_op(e, g1));
  return 0;
}

static int attn_backward(Exec& e, AttnL& L, const Tensor& dy, Tensor& dx) {
  ishap_unet* u = e.u;
  const AttnSaved& sv = L.sv;
  const Tensor& x = sv.x;
  const int N = x.N, T = x.H * x.W, C = L.C, heads = L.heads, d = C / heads;
  const float alpha = 1.f / sqrtf((float)d);
  Tensor dA;
  ISHAP_TRY(dgrad_op(e, L.proj, dy, dA, C));
  Tensor dqkv{nullptr, N, x.H, x.W, 3 * C};
  ISHAP_ALLOC(dqkv.p, e, dqkv.numel());
  const size_t d_need = (size_t)N * heads * T;
  if (e.dry) {
    if (d_need > u->attn_D_floats) u->attn_D_floats = d_need;
  } else {
    AttnArgs g;
    g.qkv = sv.qkv.p; g.out = sv.a.p; g.dout = dA.p; g.dqkv = dqkv.p; g.lse = sv.lse; g.Dbuf = u->attn_D;
    g.N = N; g.T = T; g.C = C; g.heads = heads; g.d = d; g.alpha = alpha;
    ISHAP_TRY(attn_backward_launch(g, e.s));
  }
  Tensor dn;
  dx = x;
  {
    GnBwdArgs g;
    g.x = x.p; g.stats = sv.stats; g.gamma = L.n.gamma; g.beta = L.n.beta;
    g.N = N; g.H = x.H; g.W = x.W; g.C = C; g.film = 0; g.act = 0; g.gmode = GB_SAME;
    const bool loc = local_gn_bwd(T, C, GB_SAME);
    ISHAP_TRY(dgrad_op(e, L.qkv, dqkv, dn, C, loc ? nullptr : &g, loc));
    ISHAP_ALLOC(dx.p, e, x.numel());
    g.g = dn.p; g.add = dy.p; g.dx = dx.p;
    if (loc) ISHAP_TRY(gn_bwd_local_op(e, g, dn));
    else ISHAP_TRY(gn_bwd_op(e, g));
  }
  return 0;
}

static int block_backward(Exec& e, BlockL& b, Tensor g, Tensor& out, int split = 0, Tensor* out2 = nullptr,
                          const half_t* add2 = nullptr, bool* add2_done = nullptr) {
  ishap_unet* u = e.u;
  for (int i = (int)b.layers.size() - 1; i >= 0; --i) {
    const LayerRef& l = b.layers[i];
    Tensor dx;
    if (l.kind == 0) {
      ISHAP_TRY(dgrad_op(e, u->stem, g, dx, u->in_pad));
    } else if (l.kind == 1) {
      if (i == 0 && split > 0) ISHAP_TRY(res_backward(e, u->res[l.idx], g, dx, split, out2, nullptr));
      else if (i == 0 && add2) { ISHAP_TRY(res_backward(e, u->res[l.idx], g, dx, 0, nullptr, add2)); *add2_done = true; }
      else ISHAP_TRY(res_backward(e, u->res[l.idx], g, dx, 0, nullptr, nullptr));
    } else {
      ISHAP_TRY(attn_backward(e, u->attn[l.idx], g, dx));
    }
    g = dx;
  }
  out = g;
  return 0;
}

// ISHAP_BWD_MARKS=1: a timing event on the backward's stream, tagged (0 start, 100 + i after output block i, 200 after the middle
// block, 300 + i after input block i, 999 end); read back by ishap_unet_marks
static int bwd_mark(ishap_unet* u, hipStream_t s, int tag, bool dry) {
  if (dry || !u->marks_on) return 0;
  if (tag == 0) { u->marks_n = 0; u->mark_tail_set = false; }
  if ((size_t)u->marks_n == u->marks.size()) {
    hipEvent_t ev;
    ISHAP_CHECK_HIP(hipEventCreate(&ev));
    u->marks.push_back(ev);
    u->mark_tags.push_back(0);
  }
  u->mark_tags[u->marks_n] = tag;
  ISHAP_CHECK_HIP(hipEventRecord(u->marks[u->marks_n++], s));
  return 0;
}

int unet_backward_impl(ishap_unet* u, const half_t* cot_tap, const void* cot_out, int cot_out_f16, const float* scale2,
                       float* dx, hipStream_t s, bool dry) {
  Exec e{u, s, dry};
  TenancyScope tenancy(u, s, dry);
  e.tenant = tenancy.granted;
  u->arena.off = u->fwd_mark;
  u->stat_off = u->stat_fwd_mark;
  if (!dry) {
    ISHAP_REQUIRE(u->have_saved, "backward needs a preceding forward with keep_for_backward=1");
    if (u->bwd_since_fwd++ > 0 && u->stat_cap > u->stat_fwd_mark)   // a second backward on the same forward: fresh zeros
      ISHAP_CHECK_HIP(hipMemsetAsync(u->stat_base + u->stat_fwd_mark, 0, (u->stat_cap - u->stat_fwd_mark) * sizeof(long long), s));
  }
  ISHAP_TRY(bwd_mark(u, s, 0, dry));
  const ishap_unet_config& cfg = u->cfg;
  const int N = u->last_N;            // dry runs follow a dry forward of the same batch size
  const int n_in = (int)u->in_blocks.size(), n_out = (int)u->out_blocks.size();
  int F;
  Tensor g;
  if (cot_out) {
    if (!dry) ISHAP_TRY(unet_join_tail(u, s));    // full depth: needs the blocks an overlapped forward put on the side stream
    // head backward: out = conv3x3(silu(gn(h)))  (unet.py:612-616,667-669)
    F = n_out - 1;
    const int S = cfg.image_size, opad = u->head.cout_pad;
    Tensor dout{nullptr, N, S, S, opad};
    ISHAP_ALLOC(dout.p, e, dout.numel());
    if (!dry) ISHAP_TRY(nchw_to_nhwc_f16_scaled(cot_out, cot_out_f16 ? 0 : 1, dout.p, N, cfg.out_channels, S * S, opad, 1.f, s));
    Tensor dact;
    GnBwdArgs a;
    a.x = u->h_final.p; a.stats = u->head_stats; a.gamma = u->head_norm.gamma;
    a.beta = u->head_norm.beta; a.N = N; a.H = S; a.W = S; a.C = u->final_ch; a.act = 1;
    const bool loc = local_gn_bwd(S * S, u->final_ch, GB_SAME);
    ISHAP_TRY(dgrad_op(e, u->head, dout, dact, u->final_ch, loc ? nullptr : &a, loc));
    g = u->h_final;
    if (dry) g = Tensor{nullptr, N, S, S, u->final_ch};
    Tensor gh = g;
    ISHAP_ALLOC(gh.p, e, g.numel());
    a.g = dact.p; a.dx = gh.p;
    if (loc) ISHAP_TRY(gn_bwd_local_op(e, a, dact));
    else ISHAP_TRY(gn_bwd_op(e, a));
    g = gh;
  } else {
    F = dry ? n_out - 1 : u->last_feat;
    ISHAP_REQUIRE(F >= 0, "the last forward had no tap (feat_layer < 0)");
    g = u->out_blocks[F].out;
    g.p = const_cast<half_t*>(cot_tap);
  }
  std::vector<Tensor> skipgrad(n_in);
  for (int i = F; i >= 0; --i) {
    BlockL& b = u->out_blocks[i];
    const int Cs = b.skip_ch, Ch = b.cin - Cs;
    ISHAP_REQUIRE(b.layers[0].kind == 1 && Ch % 8 == 0, "an output block starts with a ResBlock on the concatenation");
    Tensor gh, gs;
    ISHAP_TRY(block_backward(e, b, g, gh, Ch, &gs));       // the first ResBlock writes d/d[h | skip] as two tensors
    skipgrad[n_in - 1 - i] = gs;     // hs.pop() order (unet.py:663)
    g = gh;
    ISHAP_TRY(bwd_mark(u, s, 100 + i, dry));
    // a deferred forward tail starts behind this point: the backward's first (chip-filling) blocks have the GPU to themselves.
    // Default: once the first output block on a map of at most 16 x 16 pixels has been differentiated (the real model: after
    // out8 ... out5, i.e. four blocks -- in-situ sweep of the count, profiles/round5_overlap_tail_ab.txt 7); ISHAP_TAIL_MID=k: after
    // k blocks, 0: never (the tail then starts at the fork)
    static const int mid_k = [] { const char* v = getenv("ISHAP_TAIL_MID"); return v ? atoi(v) : -1; }();
    const bool here = mid_k > 0 ? F - i + 1 == mid_k : (mid_k < 0 && b.res_in <= 16);
    if (!dry && u->tail_deferred && !u->mid_recorded && here) {
      ISHAP_CHECK_HIP(hipEventRecord(u->ev_mid, s));
      u->mid_recorded = true;
    }
  }
  // The gradient entering input block i is (gradient from the block after it) + (its skip-connection gradient).  When the
  // block after it ends its backward with a ResBlock (always, except the stem), that ResBlock's last kernel adds the skip
  // gradient itself; `pending` is the addend still owed when that was not possible.
  auto run = [&](BlockL& blk, const half_t* owed_next, const half_t*& pending) -> int {
    bool done = false;
    Tensor o;
    ISHAP_TRY(block_backward(e, blk, g, o, 0, nullptr, owed_next, &done));
    g = o;
    pending = (owed_next && !done) ? owed_next : nullptr;
    return 0;
  };
  const half_t* pending = nullptr;
  ISHAP_TRY(run(u->mid, n_in > 0 ? skipgrad[n_in - 1].p : nullptr, pending));
  ISHAP_TRY(bwd_mark(u, s, 200, dry));
  for (int i = n_in - 1; i >= 0; --i) {
    if (pending) {                       // fall-back: a separate add
      Tensor sum = g;
      ISHAP_ALLOC(sum.p, e, g.numel());
      if (!dry) ISHAP_TRY(add_f16(g.p, pending, sum.p, g.numel(), s));
      g = sum;
    }
    ISHAP_TRY(run(u->in_blocks[i], i > 0 ? skipgrad[i - 1].p : nullptr, pending));
    ISHAP_TRY(bwd_mark(u, s, 300 + i, dry));
  }
  if (!dry)
    ISHAP_TRY(nhwc_f16_to_nchw_f32_scaled(g.p, dx, N, cfg.in_channels, cfg.image_size * cfg.image_size, u->in_pad,
                                          scale2 ? scale2 + 1 : nullptr, s));
  ISHAP_TRY(bwd_mark(u, s, 999, dry));
  return 0;
}

// elapsed milliseconds from the start mark of the last backward to each of its marks, and to the begin / end of the forward tail
// that ran beside it on the side stream (-1: no deferred tail ran).  Synchronises with the device.
extern "C" int ishap_unet_marks(ishap_unet* u, int* tags, float* ms, int cap, float* tail_begin_ms, float* tail_end_ms) {
  ISHAP_REQUIRE(u && tags && ms && cap > 0, "null argument");
  ISHAP_CHECK_HIP(hipSetDevice(u->device));
  if (tail_begin_ms) *tail_begin_ms = -1.f;
  if (tail_end_ms) *tail_end_ms = -1.f;
  if (u->marks_n == 0) return 0;
  ISHAP_CHECK_HIP(hipDeviceSynchronize());
  const int n = u->marks_n < cap ? u->marks_n : cap;
  for (int k = 0; k < n; ++k) {
    tags[k] = u->mark_tags[k];
    ISHAP_CHECK_HIP(hipEventElapsedTime(&ms[k], u->marks[0], u->marks[k]));
  }
  if (u->mark_tail_set) {
    if (tail_begin_ms) ISHAP_CHECK_HIP(hipEventElapsedTime(tail_begin_ms, u->marks[0], u->mark_tail_begin));
    if (tail_end_ms) ISHAP_CHECK_HIP(hipEventElapsedTime(tail_end_ms, u->marks[0], u->mark_tail_end));
  }
  return n;
}

extern "C" int ishap_unet_backward_input(ishap_unet* u, const void* cot, const float* scale2, float* dx, void* stream) {
  ISHAP_REQUIRE(u && cot && dx, "null argument");
  ISHAP_TRY(ishap_check_status());
  ISHAP_CHECK_HIP(hipSetDevice(u->device));
  return unet_backward_impl(u, (const half_t*)cot, nullptr, 0, scale2, dx, (hipStream_t)stream, false);
}
extern "C" int ishap_unet_backward_from_output(ishap_unet* u, const void* cot_out, int cot_is_f16, const float* scale2,
                                               float* dx, void* stream) {
  ISHAP_REQUIRE(u && cot_out && dx, "null argument");
  ISHAP_TRY(ishap_check_status());
  ISHAP_CHECK_HIP(hipSetDevice(u->device));
  return unet_backward_impl(u, nullptr, cot_out, cot_is_f16, scale2, dx, (hipStream_t)stream, false);
}
