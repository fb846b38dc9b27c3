// css_common.h -- internal helpers shared by the translation units of libcss_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/css_hip.h"

// largest k ONE pass of the scan kernels serves (LDS lists, wave_insert: two slots per lane); css_index_search serves
// k up to CSS_MAX_K by passes of this size (css_index.hip, search_any_k)
#define CSS_KERNEL_MAX_K 128

namespace css {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

// Brackets the launches of one named kernel with HIP events on `stream` when
// profiling is enabled (css_prof_enable); see css_core.hip.
struct ProfScope {
    ProfScope(const char* name, hipStream_t stream);
    ~ProfScope();
    const char* name_;
    hipStream_t stream_;
    hipEvent_t start_ = nullptr;
    bool active_ = false;
};

struct DeviceGuard {
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        if (prev_ != dev) (void)hipSetDevice(dev);
        else prev_ = -1;
    }
    ~DeviceGuard() {
        if (prev_ >= 0) (void)hipSetDevice(prev_);
    }
    int prev_ = -1;
};

int check_device(int device);  // CSS_OK or CSS_ERR_NO_DEVICE / CSS_ERR_INVALID
// hipFuncAttributeMaxDynamicSharedMemorySize for (kernel, device), set once (thread safe)
int ensure_dynamic_lds(const void* kernel, size_t bytes, int device);
bool prof_enabled();

}  // namespace css

#define CSS_HIP_TRY(expr)                                                        \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) return css::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define CSS_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            css::set_error(__VA_ARGS__); \
            return CSS_ERR_INVALID;     \
        }                               \
    } while (0)

#define CSS_LAUNCH_CHECK() CSS_HIP_TRY(hipGetLastError())
