// UNet input-gradient pass (filled in below).
#include "unet.h"
int unet_backward_impl(ishap_unet* u, const half_t* cot_tap, const float* cot_out, const float* scale2, float* dx,
                       hipStream_t s, bool dry) {
  if (dry) return 0;
  ISHAP_REQUIRE(false, "backward not built yet");
}
extern "C" int ishap_unet_backward_input(ishap_unet* u, const void* cot, const float* scale2, float* dx, void* stream) {
  ISHAP_REQUIRE(u && cot && dx, "null argument");
  return unet_backward_impl(u, (const half_t*)cot, nullptr, scale2, dx, (hipStream_t)stream, false);
}
extern "C" int ishap_unet_backward_from_output(ishap_unet* u, const float* cot_out, float* dx, void* stream) {
  ISHAP_REQUIRE(u && cot_out && dx, "null argument");
  return unet_backward_impl(u, nullptr, cot_out, nullptr, dx, (hipStream_t)stream, false);
}
