// css_core.hip -- error reporting, device discovery and in-library kernel timing.
//
// Device discovery replaces the reference's faiss.get_num_gpus() /
// StandardGpuResources probe (src/gpu_utils.py:108-139) and
// torch.cuda.get_device_properties() use (src/gpu_utils.py:49-63) for the
// storage half of the path: the ROCm-aware policy in
// claude_semantic_search_amd/gpu_utils.py is fed from css_device_info().
#include "css_common.h"

#include <map>
#include <mutex>
#include <vector>

namespace css {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) at %s:%d in `%s`", (int)e, hipGetErrorString(e), file, line, what);
    if (e == hipErrorOutOfMemory) return CSS_ERR_OOM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return CSS_ERR_NO_DEVICE;
    return CSS_ERR_HIP;
}

int check_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (hipGetDeviceCount: %s); libcss_hip has no CPU fallback",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        (void)hipGetLastError();
        return CSS_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range [0, %d)", device, n);
        return CSS_ERR_INVALID;
    }
    return CSS_OK;
}

int ensure_dynamic_lds(const void* kernel, size_t bytes, int device) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(kernel, device);
    auto it = done.find(key);
    if (it != done.end() && it->second >= bytes) return CSS_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)", __FILE__, __LINE__);
    done[key] = bytes;
    return CSS_OK;
}

// ---------------------------------------------------------------- profiling
struct ProfRec {
    hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::map<std::string, std::vector<ProfRec>> g_prof;

bool prof_enabled() { return g_prof_on; }

ProfScope::ProfScope(const char* name, hipStream_t stream) : name_(name), stream_(stream) {
    if (!g_prof_on) return;
    if (hipEventCreate(&start_) != hipSuccess) return;
    if (hipEventRecord(start_, stream_) != hipSuccess) {
        (void)hipEventDestroy(start_);
        return;
    }
    active_ = true;
}

ProfScope::~ProfScope() {
    if (!active_) return;
    hipEvent_t stop;
    if (hipEventCreate(&stop) != hipSuccess) return;
    (void)hipEventRecord(stop, stream_);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[name_].push_back({start_, stop});
}

}  // namespace css

extern "C" {

const char* css_version(void) { return "css_hip 0.1.0 (gfx950)"; }

const char* css_last_error(void) { return css::g_err.c_str(); }

int css_device_count(int* n) {
    CSS_REQUIRE(n != nullptr, "css_device_count: n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return CSS_OK;
}

int css_device_info(int device, css_devinfo* out) {
    CSS_REQUIRE(out != nullptr, "css_device_info: out is NULL");
    int rc = css::check_device(device);
    if (rc != CSS_OK) return rc;
    hipDeviceProp_t p;
    CSS_HIP_TRY(hipGetDeviceProperties(&p, device));
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", p.name);
    snprintf(out->gcn_arch, sizeof(out->gcn_arch), "%s", p.gcnArchName);
    out->compute_units = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->hbm_total_bytes = (int64_t)p.totalGlobalMem;
    out->lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    out->clock_mhz = p.clockRate / 1000;
    css::DeviceGuard g(device);
    size_t fr = 0, tot = 0;
    CSS_HIP_TRY(hipMemGetInfo(&fr, &tot));
    out->hbm_free_bytes = (int64_t)fr;
    return CSS_OK;
}

int css_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(css::g_prof_mu);
    css::g_prof_on = on != 0;
    return CSS_OK;
}

int css_prof_reset(void) {
    std::lock_guard<std::mutex> lk(css::g_prof_mu);
    for (auto& kv : css::g_prof)
        for (auto& r : kv.second) {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
    css::g_prof.clear();
    return CSS_OK;
}

int css_prof_read(const char* kernel, double* total_ms, int64_t* launches) {
    CSS_REQUIRE(kernel && total_ms && launches, "css_prof_read: NULL argument");
    std::lock_guard<std::mutex> lk(css::g_prof_mu);
    double tot = 0.0;
    int64_t n = 0;
    auto it = css::g_prof.find(kernel);
    if (it != css::g_prof.end()) {
        for (auto& r : it->second) {
            CSS_HIP_TRY(hipEventSynchronize(r.b));
            float ms = 0.f;
            CSS_HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
            tot += ms;
            ++n;
        }
    }
    *total_ms = tot;
    *launches = n;
    return CSS_OK;
}

}  // extern "C"
