// css_encoder.hip -- MPNet encoder entry points (TEMPORARY: kernels land next commit).
#include "css_common.h"
extern "C" {
#define CSS_TODO(name) css::set_error(name ": encoder not built yet"); return CSS_ERR_STATE
int css_encoder_create(const css_encoder_cfg*, int, css_encoder**) { CSS_TODO("css_encoder_create"); }
int css_encoder_free(css_encoder*) { return CSS_OK; }
int css_encoder_load_weights(css_encoder*, const css_tensor*, int) { CSS_TODO("css_encoder_load_weights"); }
int css_encoder_init_synthetic(css_encoder*, uint64_t) { CSS_TODO("css_encoder_init_synthetic"); }
int css_encoder_export_weight(const css_encoder*, const char*, float*, int64_t) { CSS_TODO("css_encoder_export_weight"); }
int css_encoder_forward(css_encoder*, const int32_t*, const int32_t*, int, int, float*) { CSS_TODO("css_encoder_forward"); }
int css_encoder_forward_dev(css_encoder*, const int32_t*, const int32_t*, int, int, int, int, float*, void*) { CSS_TODO("css_encoder_forward_dev"); }
}
