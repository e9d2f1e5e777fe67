// UNet graph construction, parameter loading and the forward executor.
// Reference structure: guided_diffusion/unet.py:427-616 (constructor), :634-671 (forward),
// :236-256 (ResBlock), :299-305 + :337-354 (AttentionBlock, legacy qkv order).
// One C call runs the whole network: ~500 kernel launches on the caller's stream, no host sync.
#include "unet.h"

#include <cstring>

#include "attention.h"
#include "decode.h"
#include "misc.h"
#include "norm.h"

static int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------------
// graph
// ------------------------------------------------------------------------------------------------
static void init_conv(ConvW& c, const std::string& path, int cin, int cout, int taps) {
  c.path = path; c.cin = cin; c.cout = cout; c.taps = taps;
  c.kpad = round_up(cin, 32);
  c.cout_pad = round_up(cout, 32);
}

int unet_build(ishap_unet* u) {
  const ishap_unet_config& cfg = u->cfg;
  const int mc = cfg.model_channels;
  u->ted = mc * 4;
  // stem input channels padded with zeros: to 64-wide K-steps when that is cheap (96 -> 128: the LDS-DMA kernel), else to 32
  u->in_pad = cfg.in_channels >= 64 ? round_up(cfg.in_channels, 64) : round_up(cfg.in_channels, 32);
  auto has_att = [&](int ds) {
    for (int i = 0; i < cfg.n_att; ++i) if (cfg.attention_ds[i] == ds) return true;
    return false;
  };
  auto heads_of = [&](int ch) { return ch / cfg.num_head_channels; };
  // count layers first so the vectors never reallocate (ParamSlot keeps pointers into them)
  u->res.reserve(256);
  u->attn.reserve(64);
  int film = 0;
  auto add_res = [&](const std::string& path, int cin, int cout, bool up, bool down) {
    ResL r;
    r.path = path; r.cin = cin; r.cout = cout; r.up = up; r.down = down;
    r.n1.C = cin; r.n2.C = cout;
    init_conv(r.c1, path + ".in_layers.2", cin, cout, 9);
    init_conv(r.c2, path + ".out_layers.3", cout, cout, 9);
    r.has_skip = cin != cout;
    if (r.has_skip) init_conv(r.skip, path + ".skip_connection", cin, cout, 1);
    r.emb_off = film;
    film += 2 * cout;
    u->res.push_back(r);
    return (int)u->res.size() - 1;
  };
  auto add_attn = [&](const std::string& path, int ch) {
    AttnL a;
    a.path = path; a.C = ch; a.heads = heads_of(ch);
    a.n.C = ch;
    init_conv(a.qkv, path + ".qkv", ch, 3 * ch, 1);
    init_conv(a.proj, path + ".proj_out", ch, ch, 1);
    u->attn.push_back(a);
    return (int)u->attn.size() - 1;
  };

  int ch = cfg.channel_mult[0] * mc;
  int res = cfg.image_size;
  init_conv(u->stem, "input_blocks.0.0", cfg.in_channels, ch, 9);
  u->stem.kpad = u->in_pad;
  {
    BlockL b;
    b.name = "input_blocks.0";
    b.layers.push_back({0, 0});
    b.cin = cfg.in_channels; b.cout = ch; b.res_in = b.res_out = res;
    u->in_blocks.push_back(b);
  }
  std::vector<int> chans{ch};
  int ds = 1;
  for (int level = 0; level < cfg.n_mult; ++level) {
    for (int k = 0; k < cfg.num_res_blocks; ++k) {
      BlockL b;
      b.name = "input_blocks." + std::to_string(u->in_blocks.size());
      int cout = cfg.channel_mult[level] * mc;
      b.layers.push_back({1, add_res(b.name + ".0", ch, cout, false, false)});
      b.cin = ch; ch = cout; b.cout = ch; b.res_in = b.res_out = res;
      if (has_att(ds)) b.layers.push_back({2, add_attn(b.name + ".1", ch)});
      u->in_blocks.push_back(b);
      chans.push_back(ch);
    }
    if (level != cfg.n_mult - 1) {
      BlockL b;
      b.name = "input_blocks." + std::to_string(u->in_blocks.size());
      b.layers.push_back({1, add_res(b.name + ".0", ch, ch, false, true)});
      b.cin = b.cout = ch; b.res_in = res; b.res_out = res / 2;
      u->in_blocks.push_back(b);
      chans.push_back(ch);
      ds *= 2;
      res /= 2;
    }
  }
  u->mid.name = "middle_block";
  u->mid.layers.push_back({1, add_res("middle_block.0", ch, ch, false, false)});
  u->mid.layers.push_back({2, add_attn("middle_block.1", ch)});
  u->mid.layers.push_back({1, add_res("middle_block.2", ch, ch, false, false)});
  u->mid.cin = u->mid.cout = ch; u->mid.res_in = u->mid.res_out = res;
  for (int level = cfg.n_mult - 1; level >= 0; --level) {
    for (int i = 0; i < cfg.num_res_blocks + 1; ++i) {
      int ich = chans.back();
      chans.pop_back();
      BlockL b;
      b.name = "output_blocks." + std::to_string(u->out_blocks.size());
      int cout = mc * cfg.channel_mult[level];
      b.layers.push_back({1, add_res(b.name + ".0", ch + ich, cout, false, false)});
      b.cin = ch + ich; b.skip_ch = ich; ch = cout; b.cout = ch; b.res_in = res;
      if (has_att(ds)) b.layers.push_back({2, add_attn(b.name + "." + std::to_string(b.layers.size()), ch)});
      if (level && i == cfg.num_res_blocks) {
        b.layers.push_back({1, add_res(b.name + "." + std::to_string(b.layers.size()), ch, ch, true, false)});
        ds /= 2;
        res *= 2;
      }
      b.res_out = res;
      u->out_blocks.push_back(b);
    }
  }
  u->final_ch = ch;
  u->film_rows = film;
  u->head_norm.C = ch;
  init_conv(u->head, "out.2", ch, cfg.out_channels, 9);
  u->head.split = true;
  u->head.kpad = 3 * ch;

  // ---- parameter table (torch state_dict names, gd/unet.py module tree) ----
  auto reg = [&](const std::string& name, std::initializer_list<long long> shape, int kind, ConvW* conv, float** dst,
                 long long off) {
    ParamSlot p;
    p.name = name; p.ndim = (int)shape.size();
    int i = 0;
    for (long long v : shape) p.shape[i++] = v;
    p.kind = kind; p.conv = conv; p.dst_off = off;
    p.dst = dst ? *dst : nullptr;
    u->param_index[name] = (int)u->params.size();
    u->params.push_back(p);
  };
  // device buffers for plain fp32 parameters
  auto dalloc = [&](float** p, size_t n) -> int {
    ISHAP_CHECK_HIP(hipMalloc((void**)p, n * sizeof(float)));
    ISHAP_CHECK_HIP(hipMemset(*p, 0, n * sizeof(float)));
    return 0;
  };
  auto conv_alloc = [&](ConvW& c) -> int {
    size_t wn = (size_t)round_up(c.cout, 128) * c.taps * c.kpad;
    ISHAP_CHECK_HIP(hipMalloc((void**)&c.w, wn * sizeof(half_t)));
    ISHAP_CHECK_HIP(hipMemset(c.w, 0, wn * sizeof(half_t)));
    size_t tn = (size_t)round_up(round_up(c.cin, 32), 128) * c.taps * c.cout_pad;
    ISHAP_CHECK_HIP(hipMalloc((void**)&c.wT, tn * sizeof(half_t)));
    ISHAP_CHECK_HIP(hipMemset(c.wT, 0, tn * sizeof(half_t)));
    ISHAP_TRY(dalloc(&c.bias, round_up(c.cout, 4)));
    return 0;
  };
  auto reg_conv = [&](ConvW& c) -> int {
    ISHAP_TRY(conv_alloc(c));
    if (c.taps == 9) reg(c.path + ".weight", {c.cout, c.cin, 3, 3}, 0, &c, nullptr, 0);
    else reg(c.path + ".weight", {c.cout, c.cin, 1}, 0, &c, nullptr, 0);
    reg(c.path + ".bias", {c.cout}, 1, &c, nullptr, 0);
    return 0;
  };
  auto reg_norm = [&](NormW& n, const std::string& path) -> int {
    ISHAP_TRY(dalloc(&n.gamma, n.C));
    ISHAP_TRY(dalloc(&n.beta, n.C));
    reg(path + ".weight", {n.C}, 2, nullptr, &n.gamma, 0);
    reg(path + ".bias", {n.C}, 2, nullptr, &n.beta, 0);
    return 0;
  };
  const int ted = u->ted;
  ISHAP_TRY(dalloc(&u->te_w0, (size_t)ted * mc));
  ISHAP_TRY(dalloc(&u->te_b0, ted));
  ISHAP_TRY(dalloc(&u->te_w2, (size_t)ted * ted));
  ISHAP_TRY(dalloc(&u->te_b2, ted));
  ISHAP_TRY(dalloc(&u->emb_w, (size_t)film * ted));
  ISHAP_TRY(dalloc(&u->emb_b, film));
  reg("time_embed.0.weight", {ted, mc}, 2, nullptr, &u->te_w0, 0);
  reg("time_embed.0.bias", {ted}, 2, nullptr, &u->te_b0, 0);
  reg("time_embed.2.weight", {ted, ted}, 2, nullptr, &u->te_w2, 0);
  reg("time_embed.2.bias", {ted}, 2, nullptr, &u->te_b2, 0);
  // stem is the only stem conv in the skip-less plain form; the skip 1x1 convs are Conv2d [O,I,1,1]
  ISHAP_TRY(conv_alloc(u->stem));
  reg("input_blocks.0.0.weight", {u->stem.cout, u->stem.cin, 3, 3}, 0, &u->stem, nullptr, 0);
  reg("input_blocks.0.0.bias", {u->stem.cout}, 1, &u->stem, nullptr, 0);
  for (auto& r : u->res) {
    ISHAP_TRY(reg_norm(r.n1, r.path + ".in_layers.0"));
    ISHAP_TRY(reg_conv(r.c1));
    reg(r.path + ".emb_layers.1.weight", {2 * r.cout, ted}, 2, nullptr, &u->emb_w, (long long)r.emb_off * ted);
    reg(r.path + ".emb_layers.1.bias", {2 * r.cout}, 2, nullptr, &u->emb_b, r.emb_off);
    ISHAP_TRY(reg_norm(r.n2, r.path + ".out_layers.0"));
    ISHAP_TRY(reg_conv(r.c2));
    if (r.has_skip) {
      ISHAP_TRY(conv_alloc(r.skip));
      if (!r.down && !r.up && r.c2.kpad % 64 == 0 && r.skip.kpad % 64 == 0) {
        // forward operand [c2 (9 taps) | skip (1x1)] concatenated along K: both convolutions in one launch
        const int ld = 9 * r.c2.kpad + r.skip.kpad;
        const size_t n = (size_t)round_up(r.cout, 128) * ld;
        ISHAP_CHECK_HIP(hipMalloc((void**)&r.c2.cat, n * sizeof(half_t)));
        ISHAP_CHECK_HIP(hipMemset(r.c2.cat, 0, n * sizeof(half_t)));
        r.c2.cat_ld = ld; r.c2.cat_off = 0;
        r.skip.cat = r.c2.cat; r.skip.cat_ld = ld; r.skip.cat_off = 9 * r.c2.kpad;
      }
      reg(r.skip.path + ".weight", {r.cout, r.cin, 1, 1}, 0, &r.skip, nullptr, 0);
      reg(r.skip.path + ".bias", {r.cout}, 1, &r.skip, nullptr, 0);
    }
  }
  for (auto& a : u->attn) {
    ISHAP_TRY(reg_norm(a.n, a.path + ".norm"));
    ISHAP_TRY(reg_conv(a.qkv));
    ISHAP_TRY(reg_conv(a.proj));
  }
  ISHAP_TRY(reg_norm(u->head_norm, "out.0"));
  ISHAP_TRY(conv_alloc(u->head));
  reg("out.2.weight", {u->head.cout, u->head.cin, 3, 3}, 0, &u->head, nullptr, 0);
  reg("out.2.bias", {u->head.cout}, 1, &u->head, nullptr, 0);

  const int NB = cfg.max_batch;
  ISHAP_TRY(dalloc(&u->d_temb, (size_t)NB * mc));
  ISHAP_TRY(dalloc(&u->d_e1, (size_t)NB * ted));
  ISHAP_TRY(dalloc(&u->d_emb, (size_t)NB * ted));
  ISHAP_TRY(dalloc(&u->d_film, (size_t)NB * film));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// parameter loading
// ------------------------------------------------------------------------------------------------
static int load_param(ishap_unet* u, ParamSlot& p, const float* data, hipStream_t s) {
  const long long n = p.numel();
  if (p.kind == 0) {
    ConvW& c = *p.conv;
    if (c.split) {
      ISHAP_TRY(pack_conv_weight_split(data, c.w, c.cout, c.cin, c.taps, round_up(c.cout, 128), s));
    } else {
      ISHAP_TRY(pack_conv_weight(data, c.w, c.cout, c.cin, c.taps, round_up(c.cout, 128), c.kpad, 0, s));
      if (c.cat) ISHAP_TRY(pack_conv_weight(data, c.cat, c.cout, c.cin, c.taps, round_up(c.cout, 128), c.kpad, 0, s, c.cat_ld, c.cat_off));
    }
    ISHAP_TRY(pack_conv_weight(data, c.wT, c.cout, c.cin, c.taps, round_up(round_up(c.cin, 32), 128), c.cout_pad, 1, s));
  } else if (p.kind == 1) {
    ConvW& c = *p.conv;
    if (c.split) ISHAP_CHECK_HIP(hipMemcpyAsync(c.bias, data, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    else ISHAP_TRY(round_through_f16(data, c.bias, n, s));   // torso conv bias is half in the reference (fp16_util.py:19-21)
  } else {
    ISHAP_CHECK_HIP(hipMemcpyAsync(p.dst + p.dst_off, data, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// op helpers
// ------------------------------------------------------------------------------------------------
long long* salloc(Exec& e, size_t count) {
  ishap_unet* u = e.u;
  size_t o = (u->stat_off + 63) / 64 * 64;
  u->stat_off = o + count;
  if (u->stat_off > u->stat_high) u->stat_high = u->stat_off;
  if (e.dry) return (long long*)(uintptr_t)(0x1000 + o * 8);
  return u->stat_off <= u->stat_cap ? u->stat_base + o : nullptr;
}

// In-launch rendezvous (several workgroups per GroupNorm group) is allowed on a sequence that holds the device's tenancy and is
// not the overlapped forward tail itself.  A tail still running on the side stream does not forbid it (round 4; it used to):
// the tail never launches rendezvous grids (exec_is_solo is false for it), and kernels that wait for nobody cannot starve a
// grid that does -- they finish and free their compute units (the same argument as for tenants, common.h).
bool exec_is_solo(const Exec& e) {
  return e.dry || (e.tenant && (e.u->side == nullptr || e.s != e.u->side));
}

bool small_map(int HW) {
  // ISHAP_LOCAL_GN=0: the two-pass GroupNorm route everywhere (A/B); the 64x64 maps group-local as well measured slower (round 5)
  static const int on = [] { const char* v = getenv("ISHAP_LOCAL_GN"); return v ? atoi(v) : 1; }();
  return on && HW <= 1024;
}
bool local_gn(int HW, int C) { return small_map(HW) && gn_local_fits(HW, C); }

int conv_op(Exec& e, const half_t* X, int N, int H, int W, int ldx, const half_t* Wt, int kpad, int taps,
                   int cout, const float* bias, const half_t* res, int ldr, void* out, int ldo, int out_mode, int ups,
                   int res_ups, long long* stat_out, const GnBwdArgs* gb, const half_t* X2, int ldx2, int K2,
                   const float* bias2, int ldw, SlabSrc* pend_out) {
  IgemmArgs a;
  a.stat_out = stat_out;
  if (gb) {       // this launch produces the gradient arriving at act(film(GN(x))): accumulate the GN-backward sums in its epilogue
    a.gb_x = gb->x; a.gb_stats = gb->stats; a.gb_gamma = gb->gamma; a.gb_beta = gb->beta; a.gb_emb = gb->emb;
    a.gb_emb_ld = gb->emb_ld; a.gb_film = gb->film; a.gb_act = gb->act; a.gb_csums = gb->csums;
  }
  a.X = X; a.Wt = Wt; a.out = out; a.bias = bias; a.res = res;
  a.M = N * H * W; a.N = cout; a.K = taps * kpad + K2;
  a.X2 = X2; a.ldx2 = ldx2; a.K2 = K2; a.bias2 = bias2;
  a.conv3 = taps == 9; a.Cin = kpad;
  a.ldx = ldx; a.ldw = ldw ? ldw : taps * kpad; a.ldo = ldo; a.ldr = ldr;
  a.H = H; a.W = W; a.ups = ups; a.res_ups = res_ups;
  a.out_mode = out_mode;
  a.chunk_tiles = e.chunk_tiles;
  a.flops_scale = (Wt == e.u->head.w) ? 1.f / 3.f : 1.f;
  a.ksplit = igemm_pick_ksplit(a.M, a.N, a.K, 1, pend_out != nullptr);
  if (!a.conv3 && a.M <= 64 && a.K % 64 == 0) {
    // 1x1 GEMMs on the 8x8 maps: the one-launch small-map kernel (10.7 vs 16.2 us at K = 3072) -- unless the consumer adds K
    // slices up and K is long: then the tiled kernel with 8-16 slices left pending (harness, K = 3072 -> 1024: 5.0 us against
    // the skinny kernel's 8.1; at K = 1024 the two tie, profiles/round4_gemm1x1_slices_probe.txt).  ISHAP_G1_SLICES=0: off
    static const int sliced = [] { const char* v = getenv("ISHAP_G1_SLICES"); return v ? atoi(v) : 1; }();
    a.ksplit = (sliced && pend_out && a.K >= 2048) ? (a.K >= 3072 ? 16 : 8) : 1;
  }
  if (a.conv3 && H * W == 64 && pend_out) {
    // 3x3 on the 8x8 maps whose consumer adds K slices up: igemm4's 64x64 tiles (= one image) with ~16 slices
    const int s4 = igemm4_small_map_slices(a);
    if (s4 > 1) a.ksplit = s4;
  }
  if (pend_out) *pend_out = SlabSrc{};
  if (pend_out && a.ksplit > 1) {
    // the consumer adds the slices up itself: they live in the arena until it has run
    ISHAP_REQUIRE(out_mode == IG_OUT_F16 && ldo == cout && !stat_out && !gb, "deferred reduce: dense fp16 output, no epilogue sums");
    float* slab = nullptr;
    ISHAP_ALLOC(slab, e, (size_t)a.ksplit * a.M * a.N);
    pend_out->ws = slab; pend_out->nslab = a.ksplit; pend_out->zstride = (long long)a.M * a.N;
    pend_out->bias = bias; pend_out->bias2 = bias2; pend_out->res = res; pend_out->ldr = ldr; pend_out->res_ups = res_ups;
    if (e.dry) return 0;
    a.ws = slab;
    a.defer_reduce = 1;
    return igemm_launch(a, e.s);
  }
  size_t need = a.ksplit > 1 ? (size_t)a.ksplit * a.M * a.N : 0;
  if (e.dry) {
    if (need > e.u->ws_floats) e.u->ws_floats = need;
    return 0;
  }
  ISHAP_REQUIRE(need <= e.u->ws_floats, "split-K workspace too small");
  a.ws = e.ws ? e.ws : e.u->ws;
  return igemm_launch(a, e.s);
}

// A pending tensor whose next consumer cannot add the slices up itself: the stand-alone reduce (bias, residual, fp16).
int slab_materialize(Exec& e, Tensor& t) {
  if (!t.pend.pending()) return 0;
  const SlabSrc p = t.pend;
  t.pend = SlabSrc{};
  if (e.dry) return 0;
  IgemmArgs a;
  a.ws = const_cast<float*>(p.ws); a.ksplit = p.nslab; a.M = (int)t.rows(); a.N = t.C; a.K = 64;
  a.bias = p.bias; a.bias2 = p.bias2; a.res = p.res; a.ldr = p.ldr; a.res_ups = p.res_ups;
  a.H = t.H; a.W = t.W; a.out = t.p; a.ldo = t.C; a.out_mode = IG_OUT_F16;
  return igemm_reduce_launch(a, e.s);
}

int gn_stats_op(Exec& e, const Tensor& x, float* stats) {
  size_t need = gn_partial_floats(x.N, x.H * x.W, x.C);
  if (e.dry) {
    if (need > e.u->gn_partial_floats) e.u->gn_partial_floats = need;
    return 0;
  }
  ISHAP_REQUIRE(need <= e.u->gn_partial_floats, "GroupNorm statistics scratch too small");
  return gn_stats_launch(x.p, e.gn_partial ? e.gn_partial : e.u->gn_partial, stats, x.N, x.H * x.W, x.C, e.s);
}

// GroupNorm (+FiLM) (+SiLU) (+2x2 pool) of `x` into `out` on a small map: one group-local launch that also adds up `x`
// when it is still pending (and its first half when x is a lazy skip concatenation), see norm_local.hip
static int gn_local_op(Exec& e, Tensor& x, const NormW& nw, half_t* out, half_t* xpool, float* stats_out, const float* emb,
                       int emb_ld, int film, int act, int pool) {
  GnLocalArgs g;
  if (x.cat_a) {
    g.xa = x.cat_a; g.slab = x.cat_pend; g.ya = x.cat_pend.pending() ? const_cast<half_t*>(x.cat_a) : nullptr;
    g.Ca = x.cat_ca; g.xb = x.cat_b; g.xcopy = x.p;
  } else {
    g.xa = x.p; g.slab = x.pend; g.ya = x.pend.pending() ? x.p : nullptr; g.Ca = x.C;
  }
  x.pend = SlabSrc{};
  x.cat_pend = SlabSrc{};
  g.out = out; g.xpool = xpool; g.stats_out = stats_out; g.gamma = nw.gamma; g.beta = nw.beta; g.emb = emb; g.emb_ld = emb_ld;
  g.N = x.N; g.H = x.H; g.W = x.W; g.C = x.C; g.film = film; g.act = act; g.pool = pool;
  long long* rec = nullptr;
  ISHAP_SALLOC(rec, e, (size_t)x.N * 32 * GN_REC_STRIDE);       // zeroed with the statistics arena at the start of the forward
  // several workgroups per group rendezvous inside the launch: only while this launch sequence is the context's only one
  // (beside an overlapped forward tail two half-resident rendezvous grids could wait on each other's compute units)
  g.rec = exec_is_solo(e) ? reinterpret_cast<unsigned long long*>(rec) : nullptr;
  if (e.dry) return 0;
  return gn_local_launch(g, e.s);
}

static int res_forward(Exec& e, ResL& L, Tensor x, Tensor& y) {
  ishap_unet* u = e.u;
  const int N = x.N, H = x.H, W = x.W;
  const int Ho = L.down ? H / 2 : (L.up ? H * 2 : H), Wo = L.down ? W / 2 : (L.up ? W * 2 : W);
  ISHAP_REQUIRE(x.C == L.cin, "ResBlock input channels");
  float* st1 = nullptr; ISHAP_ALLOC(st1, e, (size_t)N * 64);
  float* st2 = nullptr; ISHAP_ALLOC(st2, e, (size_t)N * 64);
  const bool lazy_cat = x.cat_a != nullptr;
  ISHAP_REQUIRE(!lazy_cat || (!L.down && !L.up), "a skip concatenation feeds a plain ResBlock");
  const bool loc_in = local_gn(H * W, L.cin), loc_out = local_gn(Ho * Wo, L.cout);
  const bool nosum_out = small_map(Ho * Wo);      // the consumers of an activation on a small map gather their own statistics
  Tensor a{nullptr, N, L.down ? Ho : H, L.down ? Wo : W, L.cin};
  ISHAP_ALLOC(a.p, e, a.numel());
  Tensor xs = x;
  xs.pend = SlabSrc{}; xs.cat_pend = SlabSrc{};
  if (L.down) { xs = a; ISHAP_ALLOC(xs.p, e, a.numel()); }
  if (loc_in) {
    ISHAP_TRY(gn_local_op(e, x, L.n1, a.p, L.down ? xs.p : nullptr, st1, nullptr, 0, 0, 1, L.down));
  } else {
    ISHAP_REQUIRE(!lazy_cat || (x.cat_sa && x.cat_sb), "a lazy concatenation on a large map carries the producers' sums");
    ISHAP_TRY(slab_materialize(e, x));
    if (!x.sums && !lazy_cat) ISHAP_TRY(gn_stats_op(e, x, st1));
    if (!e.dry) {
      GnApplyArgs g;
      g.x = x.p; g.out = a.p; g.xpool = L.down ? xs.p : nullptr;
      g.stats = st1; g.sums = x.sums; g.stats_out = x.sums ? st1 : nullptr; g.gamma = L.n1.gamma; g.beta = L.n1.beta;
      g.N = N; g.H = H; g.W = W; g.C = L.cin; g.act = 1; g.pool = L.down;
      if (lazy_cat) {
        g.x = x.cat_a; g.x2 = x.cat_b; g.sums = x.cat_sa; g.sums2 = x.cat_sb; g.csplit = x.cat_ca; g.xcopy = x.p;
        g.stats_out = st1;
      }
      ISHAP_TRY(gn_apply_launch(g, e.s));
    }
  }
  Tensor h1{nullptr, N, Ho, Wo, L.cout};
  ISHAP_ALLOC(h1.p, e, h1.numel());
  if (!nosum_out) ISHAP_SALLOC(h1.sums, e, (size_t)N * L.cout * 2);
  Tensor c = h1;
  c.pend = SlabSrc{};
  ISHAP_ALLOC(c.p, e, h1.numel());
  ISHAP_TRY(conv_op(e, a.p, N, Ho, Wo, L.cin, L.c1.w, L.c1.kpad, 9, L.cout, L.c1.bias, nullptr, 0, h1.p, L.cout, IG_OUT_F16,
                    L.up, 0, h1.sums, nullptr, nullptr, 0, 0, nullptr, 0, loc_out ? &h1.pend : nullptr));
  if (loc_out) {
    ISHAP_TRY(gn_local_op(e, h1, L.n2, c.p, nullptr, st2, u->film_cur + L.emb_off, u->film_cur_ld, 1, 1, 0));
  } else {
    if (!h1.sums) ISHAP_TRY(gn_stats_op(e, h1, st2));
    if (!e.dry) {
      GnApplyArgs g;
      g.x = h1.p; g.out = c.p; g.stats = st2; g.sums = h1.sums; g.stats_out = h1.sums ? st2 : nullptr; g.gamma = L.n2.gamma; g.beta = L.n2.beta;
      g.emb = u->film_cur + L.emb_off; g.emb_ld = u->film_cur_ld;
      g.N = N; g.H = Ho; g.W = Wo; g.C = L.cout; g.film = 1; g.act = 1;
      ISHAP_TRY(gn_apply_launch(g, e.s));
    }
  }
  y = h1;
  y.pend = SlabSrc{};
  ISHAP_ALLOC(y.p, e, h1.numel());
  y.sums = nullptr;
  if (!nosum_out) ISHAP_SALLOC(y.sums, e, (size_t)N * L.cout * 2);
  // the block output's next reader is always a GroupNorm pass (the next ResBlock's, an attention block's, the head's):
  // on a small map it may stay pending
  SlabSrc* ypend = small_map(Ho * Wo) ? &y.pend : nullptr;
  if (L.has_skip && L.c2.cat) {
    // y = conv2(c) + skip(x) as ONE launch: the 1x1 skip convolution is L.cin more K columns read from x
    ISHAP_TRY(conv_op(e, c.p, N, Ho, Wo, L.cout, L.c2.cat, L.c2.kpad, 9, L.cout, L.c2.bias, nullptr, 0, y.p, L.cout,
                      IG_OUT_F16, 0, 0, y.sums, nullptr, xs.p, L.cin, L.skip.kpad, L.skip.bias, L.c2.cat_ld, ypend));
  } else if (L.has_skip) {
    ISHAP_TRY(conv_op(e, xs.p, N, Ho, Wo, L.cin, L.skip.w, L.skip.kpad, 1, L.cout, L.skip.bias, nullptr, 0, y.p, L.cout,
                      IG_OUT_F16, 0, 0));
    ISHAP_TRY(conv_op(e, c.p, N, Ho, Wo, L.cout, L.c2.w, L.c2.kpad, 9, L.cout, L.c2.bias, y.p, L.cout, y.p, L.cout,
                      IG_OUT_F16, 0, 0, y.sums));
  } else {
    ISHAP_TRY(conv_op(e, c.p, N, Ho, Wo, L.cout, L.c2.w, L.c2.kpad, 9, L.cout, L.c2.bias, xs.p, L.cin, y.p, L.cout,
                      IG_OUT_F16, 0, L.up, y.sums, nullptr, nullptr, 0, 0, nullptr, 0, ypend));
  }
  h1.pend = SlabSrc{};
  x.pend = SlabSrc{}; x.cat_pend = SlabSrc{};
  L.sv.x = x; L.sv.h1 = h1; L.sv.xs = xs; L.sv.stats1 = st1; L.sv.stats2 = st2;
  return 0;
}

static int attn_forward(Exec& e, AttnL& L, Tensor x, Tensor& y) {
  ishap_unet* u = e.u;
  (void)u;
  const int N = x.N, T = x.H * x.W, C = L.C, heads = L.heads, d = C / heads;
  ISHAP_REQUIRE(x.C == C, "attention channels");
  ISHAP_REQUIRE(d % 32 == 0, "head width must be a multiple of 32");
  float* st = nullptr; ISHAP_ALLOC(st, e, (size_t)N * 64);
  float* lse = nullptr; ISHAP_ALLOC(lse, e, (size_t)N * heads * T);
  Tensor nrm = x;
  nrm.pend = SlabSrc{};
  ISHAP_ALLOC(nrm.p, e, x.numel());
  if (local_gn(T, C)) {
    ISHAP_TRY(gn_local_op(e, x, L.n, nrm.p, nullptr, st, nullptr, 0, 0, 0, 0));
  } else {
    ISHAP_TRY(slab_materialize(e, x));
    if (!x.sums) ISHAP_TRY(gn_stats_op(e, x, st));
    if (!e.dry) {
      GnApplyArgs g;
      g.x = x.p; g.out = nrm.p; g.stats = st; g.sums = x.sums; g.stats_out = x.sums ? st : nullptr; g.gamma = L.n.gamma; g.beta = L.n.beta;
      g.N = N; g.H = x.H; g.W = x.W; g.C = C; g.act = 0;
      ISHAP_TRY(gn_apply_launch(g, e.s));
    }
  }
  Tensor qkv{nullptr, N, x.H, x.W, 3 * C};
  ISHAP_ALLOC(qkv.p, e, qkv.numel());
  if (attn8_applicable(N, T, C, d) && L.qkv.kpad == C && L.proj.kpad == C && small_map(T)) {
    // 8x8 map: qkv GEMM + attention + proj_out in ONE launch (attention.hip, attn8_fused_kernel); proj_out leaves as per-head
    // K slices that the next GroupNorm pass adds up with its bias and the residual x (ISHAP_ATTN8=1).  The route depends on the
    // shape only: a sequence without the rendezvous tenancy (a second context, the deferred forward tail's replay) makes the same
    // allocations and runs the same kernel as two launches -- the same bits (round 6)
    Tensor a = x;
    ISHAP_ALLOC(a.p, e, x.numel());
    float* slices = nullptr;
    ISHAP_ALLOC(slices, e, (size_t)heads * N * T * C);
    long long* fl = nullptr;
    ISHAP_SALLOC(fl, e, (size_t)N * heads * 8);                  // 16 flags of 4 bytes per (image, head), zeroed with the statistics arena
    y = x;
    y.pend = SlabSrc{};
    ISHAP_ALLOC(y.p, e, x.numel());
    y.sums = nullptr;
    y.pend.ws = slices; y.pend.nslab = heads; y.pend.zstride = (long long)N * T * C;
    y.pend.bias = L.proj.bias; y.pend.res = x.p; y.pend.ldr = C;
    if (!e.dry) {
      Attn8Args g;
      g.xn = nrm.p; g.wqkv = L.qkv.w; g.bqkv = L.qkv.bias; g.wproj = L.proj.w; g.qkv = qkv.p; g.aout = a.p; g.lse = lse;
      g.slices = slices; g.flags = reinterpret_cast<unsigned*>(fl); g.status = ishap_status_word();
      ISHAP_REQUIRE(g.status != nullptr, "device status word");
      g.N = N; g.C = C; g.heads = heads; g.alpha = 1.f / sqrtf((float)d);
      ISHAP_TRY(attn8_fused_launch(g, e.s, exec_is_solo(e)));
    }
    L.sv.x = x; L.sv.qkv = qkv; L.sv.a = a; L.sv.stats = st; L.sv.lse = lse; L.sv.P = nullptr;
    return 0;
  }
  ISHAP_TRY(conv_op(e, nrm.p, N, x.H, x.W, C, L.qkv.w, L.qkv.kpad, 1, 3 * C, L.qkv.bias, nullptr, 0, qkv.p, 3 * C,
                    IG_OUT_F16, 0, 0));
  Tensor a = x;
  ISHAP_ALLOC(a.p, e, x.numel());
  if (!e.dry) {
    // fused flash-style attention: w = softmax((q*s)^T (k*s)), a = w v   (unet.py:347-353)
    AttnArgs g;
    g.qkv = qkv.p; g.out = a.p; g.lse = lse; g.N = N; g.T = T; g.C = C; g.heads = heads; g.d = d;
    g.alpha = 1.f / sqrtf((float)d);
    ISHAP_TRY(attn_forward_launch(g, e.s));
  }
  y = x;
  y.pend = SlabSrc{};
  ISHAP_ALLOC(y.p, e, x.numel());
  y.sums = nullptr;
  const bool nosum = small_map(T);
  if (!nosum) ISHAP_SALLOC(y.sums, e, (size_t)N * C * 2);
  ISHAP_TRY(conv_op(e, a.p, N, x.H, x.W, C, L.proj.w, L.proj.kpad, 1, C, L.proj.bias, x.p, C, y.p, C, IG_OUT_F16, 0, 0,
                    y.sums, nullptr, nullptr, 0, 0, nullptr, 0, nosum ? &y.pend : nullptr));
  L.sv.x = x; L.sv.qkv = qkv; L.sv.a = a; L.sv.stats = st; L.sv.lse = lse; L.sv.P = nullptr;
  return 0;
}

static int block_forward(Exec& e, BlockL& b, Tensor h, Tensor& out) {
  ishap_unet* u = e.u;
  for (auto& l : b.layers) {
    Tensor y;
    if (l.kind == 0) {
      y = Tensor{nullptr, h.N, h.H, h.W, u->stem.cout};
      ISHAP_ALLOC(y.p, e, y.numel());
      if (!small_map(h.H * h.W)) ISHAP_SALLOC(y.sums, e, (size_t)h.N * u->stem.cout * 2);
      ISHAP_TRY(conv_op(e, h.p, h.N, h.H, h.W, h.C, u->stem.w, u->stem.kpad, 9, u->stem.cout, u->stem.bias, nullptr, 0, y.p,
                        u->stem.cout, IG_OUT_F16, 0, 0, y.sums));
    } else if (l.kind == 1) {
      ISHAP_TRY(res_forward(e, u->res[l.idx], h, y));
    } else {
      ISHAP_TRY(attn_forward(e, u->attn[l.idx], h, y));
    }
    h = y;
  }
  out = h;
  b.out = h;                      // what later readers see: by then the next GroupNorm pass has added up a pending output
  b.out.pend = SlabSrc{};
  return 0;
}

// order `s` behind a forward tail that is still running on the context's side stream (no-op when there is none)
int unet_run_tail(ishap_unet* u);
int unet_join_tail(ishap_unet* u, hipStream_t s) {
  if (u->tail_deferred) ISHAP_TRY(unet_run_tail(u));     // planned but never enqueued: enqueue it now
  if (!u->tail_pending) return 0;
  ISHAP_CHECK_HIP(hipStreamWaitEvent(s, u->ev_tail, 0));
  u->tail_pending = false;
  return 0;
}

// the launch-sequence settings of the forward tail on the context's side stream
static void tail_exec(Exec& e, ishap_unet* u) {
  e.s = u->side;
  e.ws = u->ws_side;
  e.gn_partial = u->gn_partial_side;
  // the tail's 3x3 convolutions as launches of at most P tiles: it then holds the LDS of at most P compute units at a time and
  // the backward chain on the caller's stream keeps the rest.  In-situ sweep with the start point (profiles/round5_overlap_tail_ab.txt):
  // 96 / 112 / 128 / 144 / 160 / 192 / 256 tiles -> 0.1706 / 0.1707 / 0.1682 / 0.1707 / 0.1714 / 0.1727 / 0.1785 s per edit
  static const int wgs = [] { const char* v = getenv("ISHAP_TAIL_DEFER_WGS"); return v ? atoi(v) : 128; }();
  e.chunk_tiles = wgs;
}

// output blocks [i0, i1): skip concatenation + block (gd/unet.py:661-664); records the tap after block `feat_layer`
static int out_blocks_range(Exec& e, ishap_unet* u, size_t i0, size_t i1, Tensor& h, std::vector<Tensor>& hs, int feat_layer) {
  for (size_t i = i0; i < i1; ++i) {
    BlockL& b = u->out_blocks[i];
    Tensor skip = hs.back();
    hs.pop_back();
    Tensor cat{nullptr, h.N, h.H, h.W, h.C + skip.C};
    ISHAP_REQUIRE(skip.H == h.H && cat.C == b.cin, "skip connection shape");
    ISHAP_ALLOC(cat.p, e, cat.numel());
    const bool first_is_res = b.layers[0].kind == 1;
    if (first_is_res && local_gn(h.H * h.W, cat.C)) {
      // no copy pass: the ResBlock's first (group-local) GroupNorm reads both halves -- adding up h if it is still
      // pending -- and writes the concatenation as it goes
      cat.cat_a = h.p; cat.cat_b = skip.p; cat.cat_ca = h.C; cat.cat_pend = h.pend;
    } else if (first_is_res && h.sums && skip.sums && h.C % 8 == 0) {
      cat.cat_a = h.p; cat.cat_b = skip.p; cat.cat_sa = h.sums; cat.cat_sb = skip.sums; cat.cat_ca = h.C;
    } else {
      ISHAP_TRY(slab_materialize(e, h));
      cat.sums = (h.sums && skip.sums) ? salloc(e, (size_t)cat.N * cat.C * 2) : nullptr;
      if (!e.dry) ISHAP_TRY(concat2(h.p, skip.p, cat.p, cat.rows(), h.C, skip.C, e.s, h.sums, skip.sums, cat.sums, cat.N));
    }
    b.cat = cat;
    b.cat.cat_pend = SlabSrc{};
    Tensor y;
    ISHAP_TRY(block_forward(e, b, cat, y));
    h = y;
    if ((int)i == feat_layer) {
      u->tap = h;
      u->tap.pend = SlabSrc{};
    }
  }
  return 0;
}

// ---- head in fp32 (unet.py:667-669): GroupNorm, SiLU, 3x3 conv with fp32 weights.  The fp32 products are
//      formed on the fp16 MFMA from hi/lo splits of both operands (3 partial products, fp32 accumulate). ----
static int forward_head(Exec& e, ishap_unet* u, Tensor h, int N, float* out) {
  const ishap_unet_config& cfg = u->cfg;
  const int S = cfg.image_size;
  ISHAP_TRY(slab_materialize(e, h));      // the head's GroupNorm runs on the full-size map (never group-local in the real model)
  u->h_final = h;
  ISHAP_ALLOC(u->head_stats, e, (size_t)N * 64);
  if (!h.sums) ISHAP_TRY(gn_stats_op(e, h, u->head_stats));
  half_t* hsplit = nullptr; ISHAP_ALLOC(hsplit, e, (size_t)h.numel() * 3);
  if (!e.dry) {
    GnApplyArgs g;
    g.x = h.p; g.out = hsplit; g.stats = u->head_stats; g.sums = h.sums; g.stats_out = h.sums ? u->head_stats : nullptr;
    g.gamma = u->head_norm.gamma; g.beta = u->head_norm.beta;
    g.N = N; g.H = S; g.W = S; g.C = h.C; g.act = 1; g.split = 1;
    ISHAP_TRY(gn_apply_launch(g, e.s));
  }
  ISHAP_TRY(conv_op(e, hsplit, N, S, S, 3 * h.C, u->head.w, u->head.kpad, 9, cfg.out_channels, u->head.bias, nullptr, 0, out,
                    0, IG_OUT_NCHW_F32, 0, 0));
  return 0;
}

// the deferred forward tail (ISHAP_TAIL_DEFER): the launches unet_forward_impl only planned, on the side stream, behind the event
// the backward recorded after its first blocks (or behind the fork when it recorded none)
static int unet_run_tail_body(ishap_unet* u) {
  ISHAP_CHECK_HIP(hipStreamWaitEvent(u->side, u->mid_recorded ? u->ev_mid : u->ev_fork, 0));
  if (u->marks_on && u->marks_n > 0) {           // diagnostic marks (unet.h): the tail's span on the side stream
    if (!u->mark_tail_begin) { ISHAP_CHECK_HIP(hipEventCreate(&u->mark_tail_begin)); ISHAP_CHECK_HIP(hipEventCreate(&u->mark_tail_end)); }
    ISHAP_CHECK_HIP(hipEventRecord(u->mark_tail_begin, u->side));
  }
  Exec e{u, u->side, false};
  e.tenant = false;
  e.keep = u->tail.keep;
  tail_exec(e, u);
  Tensor h = u->tail.h;
  std::vector<Tensor> hs = u->tail.hs;
  ISHAP_TRY(out_blocks_range(e, u, u->tail.split, u->out_blocks.size(), h, hs, u->last_feat));
  ISHAP_TRY(forward_head(e, u, h, u->tail.N, u->tail.out));
  if (u->marks_on && u->marks_n > 0 && u->mark_tail_begin) {
    ISHAP_CHECK_HIP(hipEventRecord(u->mark_tail_end, u->side));
    u->mark_tail_set = true;
  }
  ISHAP_REQUIRE(u->arena.off == u->fwd_mark && u->stat_off == u->stat_fwd_mark, "the deferred tail allocated differently from its plan");
  return 0;
}

int unet_run_tail(ishap_unet* u) {
  if (!u->tail_deferred) return 0;
  u->tail_deferred = false;
  const size_t keep_arena = u->arena.off, keep_stat = u->stat_off;
  u->arena.off = u->tail.arena_off;
  u->stat_off = u->tail.stat_off;
  const int r = unet_run_tail_body(u);
  // on EVERY exit path: the caller's arena offsets come back, and whatever was enqueued on the side stream before a failure is
  // closed by ev_tail, so that the next join (forward, full-depth backward, read-out) still orders behind it before the arena
  // or `out` is reused
  u->arena.off = keep_arena;
  u->stat_off = keep_stat;
  if (hipEventRecord(u->ev_tail, u->side) == hipSuccess) u->tail_pending = true;
  else if (r == 0) { ishap_set_error("hipEventRecord(ev_tail) failed after the deferred forward tail"); return -1; }
  return r;
}

int unet_forward_impl(ishap_unet* u, const float* x, const float* ts, int N, int feat_layer, float* out,
                      void* inter_feat, int keep, hipStream_t s, bool dry) {
  const ishap_unet_config& cfg = u->cfg;
  ISHAP_REQUIRE(N >= 1 && N <= cfg.max_batch && N <= 16, "batch size outside [1, max_batch]");
  ISHAP_REQUIRE(feat_layer < (int)u->out_blocks.size(), "feat_layer out of range");
  Exec e{u, s, dry};
  TenancyScope tenancy(u, s, dry);       // in-launch rendezvous only while no other context / stream of the process runs such grids here
  e.tenant = tenancy.granted;
  e.keep = (keep & 1) != 0;
  // the overlapped tail fills what ONE edit leaves idle; when another context / stream of the process is running sequences on the
  // device (no tenancy: common.h) the chip is shared already and a second stream per edit only oversubscribes the hardware
  // queues (three interleaved edits: 0.120 s/shape in the plain sequence, 0.248 with three overlapped tails, 0.143 with the
  // tenant's alone): also not while anyone else has asked for the device since this context's last sequence
  const bool overlap = (keep & 2) != 0 && !dry && feat_layer >= 0 && feat_layer + 1 < (int)u->out_blocks.size() && tenancy.granted &&
                       !ishap_rendezvous_contended(u, s);
  if (!dry) ISHAP_TRY(unet_join_tail(u, s));       // the previous forward's tail still owns the arena it is about to reuse
  if (overlap && !u->side) {
    // lowest priority: the tail is throughput work that should fill what the latency-bound chain on the caller's stream
    // leaves idle, not compete with it for compute units
    int prio_least = 0, prio_greatest = 0;
    ISHAP_CHECK_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    ISHAP_CHECK_HIP(hipStreamCreateWithPriority(&u->side, hipStreamNonBlocking, prio_least));
    ISHAP_CHECK_HIP(hipEventCreateWithFlags(&u->ev_fork, ishap_event_flags()));
    ISHAP_CHECK_HIP(hipEventCreateWithFlags(&u->ev_tail, ishap_event_flags()));
    ISHAP_CHECK_HIP(hipEventCreateWithFlags(&u->ev_mid, ishap_event_flags()));
    if (u->ws_floats) ISHAP_CHECK_HIP(hipMalloc((void**)&u->ws_side, u->ws_floats * sizeof(float)));
    ISHAP_CHECK_HIP(hipMalloc((void**)&u->gn_partial_side, std::max<size_t>(u->gn_partial_floats, 64) * sizeof(float)));
  }
  u->arena.reset();
  u->stat_off = 0;
  // the statistics arena is zeroed by the layout conversion below (its first use comes after it)
  const bool zero_in_convert = (u->stat_cap % 2 == 0) && (reinterpret_cast<uintptr_t>(u->stat_base) & 15) == 0;
  if (!dry && u->stat_cap && !zero_in_convert) ISHAP_CHECK_HIP(hipMemsetAsync(u->stat_base, 0, u->stat_cap * sizeof(long long), s));
  u->have_saved = false;
  const int S = cfg.image_size, HW = S * S;
  // ---- timestep embedding -> emb -> every ResBlock's (scale | shift)   (unet.py:651, :245-250) ----
  u->film_cur = u->d_film;
  u->film_cur_ld = u->film_rows;
  bool prepared = false;
  if (!dry && !u->film_cache_ts.empty()) {
    bool same = true;
    for (int i = 1; i < N; ++i) same = same && ts[i] == ts[0];
    if (same)
      for (size_t k = 0; k < u->film_cache_ts.size() && !prepared; ++k)
        if (u->film_cache_ts[k] == ts[0]) {
          u->film_cur = u->film_cache + k * (size_t)u->film_rows;
          u->film_cur_ld = 0;
          prepared = true;
        }
  }
  if (!dry && !prepared) {
    TsArg ta;
    for (int i = 0; i < 16; ++i) ta.t[i] = i < N ? ts[i] : 0.f;
    ISHAP_TRY(timestep_embedding(ta, u->d_temb, N, cfg.model_channels, s));
    ISHAP_TRY(gemv_f32(u->te_w0, u->te_b0, u->d_temb, u->d_e1, u->ted, cfg.model_channels, N, 0, s));
    ISHAP_TRY(gemv_f32(u->te_w2, u->te_b2, u->d_e1, u->d_emb, u->ted, u->ted, N, 1, s));
    ISHAP_TRY(gemv_f32(u->emb_w, u->emb_b, u->d_emb, u->d_film, u->film_rows, u->ted, N, 1, s));
  }
  // ---- x: NCHW fp32 -> NHWC fp16 (h = x.type(self.dtype), unet.py:657) ----
  Tensor h{nullptr, N, S, S, u->in_pad};
  ISHAP_ALLOC(h.p, e, h.numel());
  if (!dry) ISHAP_TRY(nchw_f32_to_nhwc_f16(x, h.p, N, cfg.in_channels, HW, u->in_pad, s, zero_in_convert ? u->stat_base : nullptr,
                                           zero_in_convert ? u->stat_cap * sizeof(long long) : 0));
  u->x0 = h;
  std::vector<Tensor> hs;
  for (auto& b : u->in_blocks) {
    Tensor y;
    ISHAP_TRY(block_forward(e, b, h, y));
    h = y;
    hs.push_back(h);
    hs.back().pend = SlabSrc{};     // read again only after the next block's first GroupNorm pass has added it up
  }
  {
    Tensor y;
    ISHAP_TRY(block_forward(e, u->mid, h, y));
    h = y;
  }
  u->tap = Tensor{};
  const size_t n_out = u->out_blocks.size();
  const size_t split = overlap ? (size_t)feat_layer + 1 : n_out;
  ISHAP_TRY(out_blocks_range(e, u, 0, split, h, hs, feat_layer));
  if (overlap) {
    // everything after the tap needs only what exists now; whatever the caller enqueues on `s` after this call (loss,
    // backward) needs nothing of what follows: fork
    ISHAP_TRY(slab_materialize(e, h));             // the tap is complete on the caller's stream
    u->tap = h;
    ISHAP_CHECK_HIP(hipEventRecord(u->ev_fork, s));
    // DEFERRED tail (round 5: -2.2 % against a tail enqueued here, at the tap; that form was removed in round 6): only PLAN the
    // rest -- allocate what it will allocate, launch nothing -- so that the caller's backward gets its scratch above the whole
    // forward; ishap_unet_run_tail enqueues the launches later, behind an event the backward records after its first blocks
    // (the tail then runs beside the backward's latency-bound middle, not beside its chip-filling first and last launches).
    // The plan and the replay make the same allocations by construction: no route below depends on the tenancy, only launch
    // forms do (gn_local_op's parts, attn8_fused_launch's one- or two-launch form)
    u->tail = ishap_unet::TailState{split, h, hs, N, out, u->arena.off, u->stat_off, (keep & 1) != 0};
    Exec plan = e;
    plan.dry = true;
    Tensor hp = h;
    std::vector<Tensor> hsp = hs;
    ISHAP_TRY(out_blocks_range(plan, u, split, n_out, hp, hsp, feat_layer));
    ISHAP_TRY(forward_head(plan, u, hp, N, out));
    u->tail_deferred = true;
    u->mid_recorded = false;
  } else {
    ISHAP_TRY(forward_head(e, u, h, N, out));
  }
  if (feat_layer >= 0 && inter_feat && !dry)
    ISHAP_TRY(nhwc_f16_to_nchw(u->tap.p, inter_feat, 0, N, u->tap.C, u->tap.H * u->tap.W, u->tap.C, s));
  u->last_N = N;
  u->last_feat = feat_layer;
  u->have_saved = (keep & 1) != 0 && !dry;
  u->fwd_mark = u->arena.off;
  u->stat_fwd_mark = u->stat_off;
  u->bwd_since_fwd = 0;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int ishap_version(void) { return 3; }   // 2: ishap_mesh_smooth takes the scratch size; 3: ishap_step_coefs carries the rng fields

int ishap_unet_create(const ishap_unet_config* cfg, int device, ishap_unet** out) {
  ISHAP_REQUIRE(cfg && out, "null argument");
  ISHAP_REQUIRE(cfg->n_mult >= 1 && cfg->n_mult <= 8 && cfg->n_att <= 8, "config arrays");
  ISHAP_REQUIRE(cfg->model_channels % 32 == 0, "model_channels must be a multiple of 32 (GroupNorm32)");
  ISHAP_REQUIRE(cfg->num_head_channels > 0 && cfg->num_head_channels % 32 == 0, "num_head_channels % 32");
  ISHAP_REQUIRE(cfg->out_channels % 4 == 0, "out_channels % 4");
  ISHAP_REQUIRE(cfg->max_batch >= 1 && cfg->max_batch <= 16, "max_batch in [1,16]");
  int minres = cfg->image_size >> (cfg->n_mult - 1);
  ISHAP_REQUIRE(minres * minres >= 64 && (cfg->image_size & (cfg->image_size - 1)) == 0,
                "image_size: power of two with at least 8x8 at the deepest level");
  ISHAP_CHECK_HIP(hipSetDevice(device));
  ishap_unet* u = new ishap_unet();
  u->cfg = *cfg;
  { const char* v = getenv("ISHAP_BWD_MARKS"); u->marks_on = v && atoi(v) != 0; }
  u->device = device;
  int r = unet_build(u);
  if (r) { delete u; return r; }
  // size the arena / workspaces with dry runs of forward + backward at EVERY batch size a call may use: the split-K
  // policy depends on M = N*H*W, so a smaller batch can need a larger fp32 partial workspace than max_batch does
  u->arena.dry = true;
  std::vector<float> ts(cfg->max_batch, 0.f);
  for (int nb = 1; nb <= cfg->max_batch && !r; ++nb) {
    r = unet_forward_impl(u, nullptr, ts.data(), nb, (int)u->out_blocks.size() - 1, nullptr, nullptr, 1, 0, true);
    if (!r) r = unet_backward_impl(u, nullptr, (const void*)0x1000, 0, nullptr, nullptr, 0, true);   // on top of the kept forward state
  }
  if (r) { delete u; return r; }
  u->arena.dry = false;
  u->arena.cap = align_up(u->arena.high + (1 << 20), 1 << 20);
  ISHAP_CHECK_HIP(hipMalloc((void**)&u->arena.base, u->arena.cap));
  if (u->ws_floats) ISHAP_CHECK_HIP(hipMalloc((void**)&u->ws, u->ws_floats * sizeof(float)));
  ISHAP_CHECK_HIP(hipMalloc((void**)&u->gn_partial, std::max<size_t>(u->gn_partial_floats, 64) * sizeof(float)));
  u->stat_cap = (u->stat_high + 1023) / 1024 * 1024;
  ISHAP_CHECK_HIP(hipMalloc((void**)&u->stat_base, std::max<size_t>(u->stat_cap, 1024) * sizeof(long long)));
  if (u->attn_D_floats) ISHAP_CHECK_HIP(hipMalloc((void**)&u->attn_D, u->attn_D_floats * sizeof(float)));
  u->have_saved = false;
  *out = u;
  return 0;
}

void ishap_unet_destroy(ishap_unet* u) {
  if (!u) return;
  auto fr = [](void* p) { if (p) (void)hipFree(p); };
  auto frc = [&](ConvW& c) { fr(c.w); fr(c.wT); fr(c.bias); if (c.cat && c.cat_off == 0) fr(c.cat); };
  frc(u->stem); frc(u->head);
  fr(u->head_norm.gamma); fr(u->head_norm.beta);
  for (auto& r : u->res) {
    frc(r.c1); frc(r.c2); if (r.has_skip) frc(r.skip);
    fr(r.n1.gamma); fr(r.n1.beta); fr(r.n2.gamma); fr(r.n2.beta);
  }
  for (auto& a : u->attn) { frc(a.qkv); frc(a.proj); fr(a.n.gamma); fr(a.n.beta); }
  fr(u->te_w0); fr(u->te_b0); fr(u->te_w2); fr(u->te_b2); fr(u->emb_w); fr(u->emb_b);
  fr(u->d_temb); fr(u->d_e1); fr(u->d_emb); fr(u->d_film);
  fr(u->film_cache); fr(u->pc_temb); fr(u->pc_e1); fr(u->pc_emb);
  fr(u->ws_side); fr(u->gn_partial_side);
  if (u->side) (void)hipStreamDestroy(u->side);
  if (u->ev_fork) (void)hipEventDestroy(u->ev_fork);
  if (u->ev_tail) (void)hipEventDestroy(u->ev_tail);
  if (u->ev_mid) (void)hipEventDestroy(u->ev_mid);
  for (hipEvent_t ev : u->marks) (void)hipEventDestroy(ev);
  if (u->mark_tail_begin) (void)hipEventDestroy(u->mark_tail_begin);
  if (u->mark_tail_end) (void)hipEventDestroy(u->mark_tail_end);
  fr(u->arena.base); fr(u->ws); fr(u->gn_partial); fr(u->attn_D); fr(u->stat_base);
  delete u;
}

int ishap_unet_num_params(const ishap_unet* u) { return u ? (int)u->params.size() : 0; }

int ishap_unet_param_info(const ishap_unet* u, int index, char* name, int name_cap, int* ndim, long long* shape) {
  ISHAP_REQUIRE(u && index >= 0 && index < (int)u->params.size(), "param index");
  const ParamSlot& p = u->params[index];
  if (name && name_cap > 0) {
    std::strncpy(name, p.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (ndim) *ndim = p.ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  return 0;
}

int ishap_unet_load_param(ishap_unet* u, const char* name, const float* data, long long numel, void* stream) {
  ISHAP_REQUIRE(u && name && data, "null argument");
  auto it = u->param_index.find(name);
  ISHAP_REQUIRE(it != u->param_index.end(), std::string("unexpected key in state_dict: ") + name);
  ParamSlot& p = u->params[it->second];
  ISHAP_REQUIRE(p.numel() == numel, std::string("size mismatch for ") + name);
  ISHAP_CHECK_HIP(hipSetDevice(u->device));
  ISHAP_TRY(load_param(u, p, data, (hipStream_t)stream));
  if (!p.loaded) { p.loaded = true; u->n_loaded++; }
  u->film_cache_ts.clear();            // rows prepared from the previous weights are stale
  return 0;
}

// (scale | shift) rows of all ResBlocks for n timesteps, computed once ahead of a sampling loop: forwards at these
// timesteps then skip timestep_embedding + time_embed + the emb_layers GEMV (gd/unet.py:651 and :245-250 do not depend on
// x).  n = 0 drops the prepared rows.  Same kernels as the in-forward path, so the values are bit-identical.
int ishap_unet_prepare_timesteps(ishap_unet* u, const float* ts, int n, void* stream) {
  ISHAP_REQUIRE(u && (ts || n == 0) && n >= 0 && n <= 4096, "prepare_timesteps arguments");
  ISHAP_CHECK_HIP(hipSetDevice(u->device));
  hipStream_t s = (hipStream_t)stream;
  u->film_cache_ts.clear();
  // a kept forward may be reading a prepared row (its backward needs the FiLM values again): the rows are about to be
  // rewritten, so that forward can no longer be differentiated
  if (u->film_cur != u->d_film) { u->have_saved = false; u->film_cur = u->d_film; u->film_cur_ld = u->film_rows; }
  if (n == 0) return 0;
  ISHAP_REQUIRE(u->n_loaded == (int)u->params.size(), "prepare_timesteps before all parameters are loaded");
  const int mc = u->cfg.model_channels;
  if ((size_t)n > u->film_cache_rows) {
    if (u->film_cache) ISHAP_CHECK_HIP(hipFree(u->film_cache));      // hipFree waits for work that may still read it
    u->film_cache = nullptr;
    u->film_cache_rows = 0;
    ISHAP_CHECK_HIP(hipMalloc((void**)&u->film_cache, (size_t)n * u->film_rows * sizeof(float)));
    u->film_cache_rows = (size_t)n;
  }
  if (!u->pc_temb) {
    ISHAP_CHECK_HIP(hipMalloc((void**)&u->pc_temb, (size_t)16 * mc * sizeof(float)));
    ISHAP_CHECK_HIP(hipMalloc((void**)&u->pc_e1, (size_t)16 * u->ted * sizeof(float)));
    ISHAP_CHECK_HIP(hipMalloc((void**)&u->pc_emb, (size_t)16 * u->ted * sizeof(float)));
  }
  for (int c0 = 0; c0 < n; c0 += 16) {
    const int m = std::min(16, n - c0);
    TsArg ta;
    for (int i = 0; i < 16; ++i) ta.t[i] = i < m ? ts[c0 + i] : 0.f;
    ISHAP_TRY(timestep_embedding(ta, u->pc_temb, m, mc, s));
    ISHAP_TRY(gemv_f32(u->te_w0, u->te_b0, u->pc_temb, u->pc_e1, u->ted, mc, m, 0, s));
    ISHAP_TRY(gemv_f32(u->te_w2, u->te_b2, u->pc_e1, u->pc_emb, u->ted, u->ted, m, 1, s));
    ISHAP_TRY(gemv_f32(u->emb_w, u->emb_b, u->pc_emb, u->film_cache + (size_t)c0 * u->film_rows, u->film_rows, u->ted, m, 1, s));
  }
  u->film_cache_ts.assign(ts, ts + n);
  return 0;
}

int ishap_unet_params_loaded(const ishap_unet* u) { return u ? u->n_loaded : 0; }

int ishap_unet_join_tail(ishap_unet* u, void* stream) {
  ISHAP_REQUIRE(u, "null argument");
  ISHAP_CHECK_HIP(hipSetDevice(u->device));
  return unet_join_tail(u, (hipStream_t)stream);
}

int ishap_unet_run_tail(ishap_unet* u) {
  ISHAP_REQUIRE(u, "null argument");
  ISHAP_CHECK_HIP(hipSetDevice(u->device));
  return unet_run_tail(u);
}

int ishap_unet_forward(ishap_unet* u, const float* x, const float* timesteps, int N, int feat_layer, float* out,
                       void* inter_feat, int keep_for_backward, void* stream) {
  ISHAP_REQUIRE(u && x && timesteps && out, "null argument");
  ISHAP_REQUIRE(u->n_loaded == (int)u->params.size(), "missing keys: not every state_dict tensor has been loaded");
  ISHAP_TRY(ishap_check_status());          // an earlier launch's device-side failure surfaces here
  ISHAP_CHECK_HIP(hipSetDevice(u->device));
  return unet_forward_impl(u, x, timesteps, N, feat_layer, out, inter_feat, keep_for_backward, (hipStream_t)stream, false);
}

int ishap_unet_tap_shape(const ishap_unet* u, int feat_layer, int* channels, int* size) {
  ISHAP_REQUIRE(u && feat_layer >= 0 && feat_layer < (int)u->out_blocks.size(), "feat_layer out of range");
  if (channels) *channels = u->out_blocks[feat_layer].cout;
  if (size) *size = u->out_blocks[feat_layer].res_out;
  return 0;
}

const void* ishap_unet_tap_ptr(const ishap_unet* u) { return u ? u->tap.p : nullptr; }

long long ishap_unet_workspace_bytes(const ishap_unet* u) {
  if (!u) return 0;
  return (long long)(u->arena.cap + u->ws_floats * sizeof(float) + u->gn_partial_floats * sizeof(float) +
                     u->stat_cap * sizeof(long long) + u->attn_D_floats * sizeof(float));
}

int ishap_unet_block_output(const ishap_unet* u, int group, int index, int* channels, int* size, void* dst_nchw_f16,
                            void* stream) {
  ISHAP_REQUIRE(u && group >= 0 && group <= 2, "group: 0 input_blocks, 1 middle_block, 2 output_blocks");
  const BlockL* b = nullptr;
  if (group == 0) { ISHAP_REQUIRE(index >= 0 && index < (int)u->in_blocks.size(), "input block index"); b = &u->in_blocks[index]; }
  else if (group == 1) b = &u->mid;
  else { ISHAP_REQUIRE(index >= 0 && index < (int)u->out_blocks.size(), "output block index"); b = &u->out_blocks[index]; }
  if (channels) *channels = b->cout;
  if (size) *size = b->res_out;
  if (!dst_nchw_f16) return 0;
  ISHAP_REQUIRE(u->have_saved && b->out.p, "block outputs stay resident only after a forward with keep_for_backward=1");
  ISHAP_TRY(ishap_check_status());
  ISHAP_CHECK_HIP(hipSetDevice(u->device));
  ISHAP_TRY(unet_join_tail(const_cast<ishap_unet*>(u), (hipStream_t)stream));
  return nhwc_f16_to_nchw(b->out.p, dst_nchw_f16, 0, b->out.N, b->out.C, b->out.H * b->out.W, b->out.C, (hipStream_t)stream);
}

int ishap_unet_copy_tap(const ishap_unet* u, void* dst, void* stream) {
  ISHAP_REQUIRE(u && dst && u->tap.p, "no resident tap (run a forward with feat_layer >= 0 first)");
  ISHAP_CHECK_HIP(hipMemcpyAsync(dst, u->tap.p, (size_t)u->tap.numel() * sizeof(half_t), hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
  return 0;
}

}  // extern "C"
