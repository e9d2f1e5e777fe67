// C-ABI wrappers for the step / drag / decode kernels (declarations and reference citations: include/ishap.h).
#include "../../include/ishap.h"
#include "common.h"
#include "ddpm.h"
#include "decode.h"
#include "drag.h"

static thread_local std::string g_err;
void ishap_set_error(const std::string& msg) { g_err = msg; }

#include <atomic>
#include <mutex>
static std::atomic<unsigned*> g_status{nullptr};
static std::mutex g_status_mu;
unsigned* ishap_status_word() {
  unsigned* p = g_status.load(std::memory_order_acquire);
  if (p) return p;
  std::lock_guard<std::mutex> lk(g_status_mu);
  p = g_status.load(std::memory_order_relaxed);
  if (p) return p;
  void* h = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) return nullptr;
  *(volatile unsigned*)h = 0u;
  g_status.store((unsigned*)h, std::memory_order_release);
  return (unsigned*)h;
}
static const char* status_text(unsigned code) {
  switch (code) {
    case ISHAP_DEV_GN_RENDEZVOUS:
      return "device-side failure: a group-local GroupNorm rendezvous timed out (its workgroups were not co-resident); the "
             "affected launch wrote NaN -- results since then are invalid";
    case ISHAP_DEV_CHAIN_TIMEOUT:
      return "device-side failure: a small-map chain kernel gave up waiting for another workgroup; the affected launch "
             "wrote NaN -- results since then are invalid";
    default: return "device-side failure: unknown status code";
  }
}
int ishap_check_status() {
  unsigned* p = g_status.load(std::memory_order_acquire);
  if (!p) return 0;
  const unsigned code = *(volatile unsigned*)p;
  if (code == 0u) return 0;
  *(volatile unsigned*)p = 0u;
  ishap_set_error(std::string(status_text(code)) + " (code " + std::to_string(code) + ")");
  return -3;
}
int ishap_cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  int v = cached[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cached[dev].store(n, std::memory_order_relaxed);
  return n;
}

namespace {
struct Tenant {
  std::mutex mu;
  const void* owner = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;      // recorded when the tenant's last sequence was enqueued completely
  bool have = false, open = false, recorded = false;
  bool denied = false;            // another context / stream asked while this tenant held the device (read and cleared by the tenant)
};
Tenant g_tenant[64];
}  // namespace

unsigned ishap_event_flags() {
  static const unsigned f = [] {
    const char* e = getenv("ISHAP_EVENT_FENCE");
    return (e && atoi(e)) ? (unsigned)hipEventDisableTiming : (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence);
  }();
  return f;
}

bool ishap_rendezvous_begin(const void* owner, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  Tenant& t = g_tenant[dev];
  std::lock_guard<std::mutex> lk(t.mu);
  if (t.have && t.owner == owner && t.stream == s) {
    if (t.open) return false;                   // re-entered from another thread on the same (owner, stream): do not share
    t.open = true;                              // same in-order stream: the earlier sequence precedes this one on the device
    return true;
  }
  if (t.have) {
    if (t.open) { t.denied = true; return false; }                   // another sequence is being enqueued right now
    if (t.recorded && hipEventQuery(t.done) != hipSuccess) { t.denied = true; return false; }   // ... or is still running
  }
  if (!t.done && hipEventCreateWithFlags(&t.done, ishap_event_flags()) != hipSuccess) { t.done = nullptr; return false; }
  t.owner = owner; t.stream = s; t.have = true; t.open = true; t.recorded = false; t.denied = false;
  return true;
}

// true once after another context / stream was refused the tenancy while `owner` held it: the device is being shared, and work
// that only pays when ONE sequence has the chip to itself (the overlapped forward tail) should not be started
bool ishap_rendezvous_contended(const void* owner, hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  Tenant& t = g_tenant[dev];
  std::lock_guard<std::mutex> lk(t.mu);
  if (!t.have || t.owner != owner || t.stream != s) return true;
  const bool d = t.denied;
  t.denied = false;
  return d;
}

void ishap_rendezvous_end(const void* owner, hipStream_t s, bool granted) {
  if (!granted) return;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
  Tenant& t = g_tenant[dev];
  std::lock_guard<std::mutex> lk(t.mu);
  if (!t.have || t.owner != owner || t.stream != s) return;
  t.recorded = hipEventRecord(t.done, s) == hipSuccess;
  t.open = false;
  if (!t.recorded) t.have = false;
}

extern "C" int ishap_rendezvous_would_grant(const void* owner, void* stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  Tenant& t = g_tenant[dev];
  std::lock_guard<std::mutex> lk(t.mu);
  if (!t.have) return 1;
  if (t.owner == owner && t.stream == (hipStream_t)stream) return t.open ? 0 : 1;
  if (t.open) return 0;
  return (!t.recorded || hipEventQuery(t.done) == hipSuccess) ? 1 : 0;
}

extern "C" {

const char* ishap_last_error(void) { return g_err.c_str(); }

int ishap_device_status(void) { return ishap_check_status(); }

int ishap_ddpm_step(const float* x, const float* model_out, const float* noise, const float* variance_in,
                    const ishap_step_coefs* k, int N, int C, int HW, float* sample, float* pred_xstart,
                    float* variance, float* mean, void* stream) {
  ISHAP_REQUIRE(x && model_out && k, "null argument");
  ISHAP_REQUIRE(k->mode >= 0 && k->mode <= 3, "mode");
  ISHAP_REQUIRE(k->mode != 2 || noise, "mode 2 needs variance_noise in `noise`");
  DdpmStepArgs a;
  a.x = x; a.model_out = model_out; a.noise = noise; a.variance_in = variance_in;
  a.sample = sample; a.pred_xstart = pred_xstart; a.variance = variance; a.mean = mean;
  a.N = N; a.C = C; a.HW = HW;
  a.min_log = k->min_log; a.max_log = k->max_log; a.sqrt_recip = k->sqrt_recip; a.sqrt_recipm1 = k->sqrt_recipm1;
  a.coef1 = k->coef1; a.coef2 = k->coef2; a.nonzero = k->nonzero; a.clip = k->clip_denoised; a.mode = k->mode;
  a.ddim_a = k->ddim_a; a.ddim_b = k->ddim_b; a.ddim_sigma = k->ddim_sigma;
  a.rng = k->rng; a.rng_seed = k->rng_seed; a.rng_offset = k->rng_offset; a.noise_out = k->noise_out;
  return ddpm_step_launch(a, (hipStream_t)stream);
}

int ishap_ddpm_step_guided(const float* x, const float* model_out, const float* noise, const float* variance_in,
                           const ishap_step_coefs* k, int N, int C, int HW, const float* grad, float scale,
                           const float* grad_mul_dev, float* guided, float* sample, float* variance, void* stream) {
  ISHAP_REQUIRE(x && model_out && k && grad && guided, "null argument");
  ISHAP_REQUIRE(k->mode == 0, "the guided step is p_sample_guidance's sqrt(variance) form (mode 0)");
  DdpmStepArgs a;
  a.x = x; a.model_out = model_out; a.noise = noise; a.variance_in = variance_in;
  a.sample = sample; a.variance = variance;
  a.N = N; a.C = C; a.HW = HW;
  a.min_log = k->min_log; a.max_log = k->max_log; a.sqrt_recip = k->sqrt_recip; a.sqrt_recipm1 = k->sqrt_recipm1;
  a.coef1 = k->coef1; a.coef2 = k->coef2; a.nonzero = k->nonzero; a.clip = k->clip_denoised; a.mode = k->mode;
  a.guide_grad = grad; a.guide_scale = scale; a.guide_mul = grad_mul_dev; a.guided = guided;
  a.rng = k->rng; a.rng_seed = k->rng_seed; a.rng_offset = k->rng_offset; a.noise_out = k->noise_out;
  return ddpm_step_launch(a, (hipStream_t)stream);
}

int ishap_guided_update(const float* sample, const float* variance, const float* grad, float scale,
                        const float* grad_mul_dev, long long numel, float* out, void* stream) {
  ISHAP_REQUIRE(sample && variance && grad && out, "null argument");
  return guided_update_launch(sample, variance, grad, out, scale, grad_mul_dev, numel, (hipStream_t)stream);
}

int ishap_axpby(const float* x, const float* y, float a, float b, long long numel, float* out, void* stream) {
  ISHAP_REQUIRE(x && y && out, "null argument");
  return axpby_launch(x, y, out, a, b, numel, (hipStream_t)stream);
}

static int fill_drag(const ishap_drag_args* a, DragArgs& d) {
  ISHAP_REQUIRE(a && a->chmap && a->sources && a->targets && a->touched && a->nmask && a->acc && a->grad_fx && a->chan_weight,
                "null argument");
  ISHAP_REQUIRE(a->W > 1 && a->B >= 1 && a->r >= 0 && a->Cc >= 1 && a->ld >= 1, "drag dims");
  d.W = a->W; d.ld = a->ld; d.Cc = a->Cc; d.chmap = a->chmap; d.sources = a->sources; d.targets = a->targets;
  d.B = a->B; d.r = a->r; d.voxel = a->voxel; d.cof = a->cof; d.l1 = a->l1;
  d.touched = a->touched; d.nmask = a->nmask; d.acc = (long long*)a->acc; d.gfx = (long long*)a->grad_fx;
  d.chw = a->chan_weight;
  return 0;
}

int ishap_drag_setup(const ishap_drag_args* a, void* stream) {
  DragArgs d;
  ISHAP_TRY(fill_drag(a, d));
  return drag_setup_launch(d, (hipStream_t)stream);
}

int ishap_drag_loss_grad(const ishap_drag_args* a, const void* edit, const void* orig, float* grad, float* loss,
                         void* stream) {
  DragArgs d;
  ISHAP_TRY(fill_drag(a, d));
  ISHAP_REQUIRE(edit && orig && grad && loss, "null argument");
  d.edit = (const half_t*)edit; d.orig = (const half_t*)orig; d.grad = grad; d.loss = loss;
  return drag_loss_grad_launch(d, (hipStream_t)stream);
}

int ishap_drag_loss_cotangent(const ishap_drag_args* a, const void* edit, const void* orig, float* grad, float* loss,
                              void* cot_f16, unsigned* bits, float* scale2, void* stream) {
  DragArgs d;
  ISHAP_TRY(fill_drag(a, d));
  ISHAP_REQUIRE(edit && orig && grad && loss && cot_f16 && bits && scale2, "null argument");
  d.edit = (const half_t*)edit; d.orig = (const half_t*)orig; d.grad = grad; d.loss = loss;
  return drag_loss_cotangent_launch(d, (half_t*)cot_f16, bits, scale2, (hipStream_t)stream);
}

int ishap_grad_to_scaled_f16(const float* grad, void* out_f16, unsigned* bits, float* scale2, long long numel,
                             void* stream) {
  ISHAP_REQUIRE(grad && out_f16 && bits && scale2, "null argument");
  return grad_to_scaled_f16_launch(grad, (half_t*)out_f16, bits, scale2, numel, (hipStream_t)stream);
}

int ishap_planes_prepare(const float* latent, const float* range, const float* middle, int S, float* planes,
                         void* stream) {
  ISHAP_REQUIRE(latent && planes, "null argument");
  return planes_prepare_launch(latent, range, middle, planes, S, (hipStream_t)stream);
}

static int fill_dec(const ishap_decoder_weights* w, DecodeArgs& d) {
  ISHAP_REQUIRE(w && w->B && w->W1 && w->b1 && w->W2 && w->b2 && w->w3 && w->b3, "null decoder weights");
  d.B = w->B; d.W1 = w->W1; d.b1 = w->b1; d.W2 = w->W2; d.b2 = w->b2; d.w3 = w->w3; d.b3 = w->b3;
  return 0;
}

int ishap_triplane_decode_points(const float* planes, int S, const ishap_decoder_weights* w, const float* coords,
                                 long long npts, float* logits, void* stream) {
  DecodeArgs d;
  ISHAP_TRY(fill_dec(w, d));
  ISHAP_REQUIRE(planes && coords && logits, "null argument");
  d.planes = planes; d.S = S; d.coords = coords; d.npts = npts; d.out = logits;
  return triplane_decode_launch(d, (hipStream_t)stream);
}

int ishap_triplane_points_loss_grad(const float* planes, int S, const ishap_decoder_weights* w, const float* W1T,
                                    const float* W2T, const float* coords, const float* gt, long long npts,
                                    float* dplanes, float* loss, float* logits, void* stream) {
  DecodeArgs d;
  ISHAP_TRY(fill_dec(w, d));
  ISHAP_REQUIRE(planes && W1T && W2T && coords && gt && dplanes && loss, "null argument");
  DecBwdArgs b;
  b.planes = planes; b.S = S; b.B = d.B; b.W1 = d.W1; b.b1 = d.b1; b.W2 = d.W2; b.b2 = d.b2; b.w3 = d.w3; b.b3 = d.b3;
  b.W1T = W1T; b.W2T = W2T; b.coords = coords; b.gt = gt; b.npts = npts; b.dplanes = dplanes; b.loss = loss; b.logits = logits;
  return decode_points_bwd_launch(b, (hipStream_t)stream);
}

int ishap_x0_grad_to_cotangent(const float* dplanes, const float* range, const float* x, const float* model_out,
                               float sqrt_recip, float sqrt_recipm1, int clip_denoised, int S, float* g_direct,
                               float* cot_out, void* stream) {
  ISHAP_REQUIRE(dplanes && x && model_out && g_direct && cot_out && (S * S) % 32 == 0, "null argument");
  return x0_grad_launch(dplanes, range, x, model_out, sqrt_recip, sqrt_recipm1, clip_denoised, S, g_direct, cot_out,
                        (hipStream_t)stream);
}

int ishap_triplane_decode_grid(const float* planes, int S, const ishap_decoder_weights* w, const float* axis, int res,
                               float* volume, void* stream) {
  DecodeArgs d;
  ISHAP_TRY(fill_dec(w, d));
  ISHAP_REQUIRE(planes && axis && volume && res > 0, "null argument");
  d.planes = planes; d.S = S; d.lin = axis; d.res = res; d.npts = (long long)res * res * res; d.out = volume;
  return triplane_decode_launch(d, (hipStream_t)stream);
}

}  // extern "C"
