// Error reporting, ABI version and the per-device zero page of libfgt_hip.so.
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include "common.h"

static thread_local char g_err[512] = "";

void fgt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fgt_last_error(void) { return g_err; }
extern "C" int fgt_abi_version(void) { return 3; }

// One 256-byte zero-filled allocation PER DEVICE (the target of out-of-image im2col gathers: a kernel on device d must not be
// handed memory of device 0), created under a mutex.  fgt_init(device) creates it eagerly so that the lazy path below never runs
// inside a stream capture (hipMalloc / hipMemset are not capturable).
namespace {
constexpr int MAX_DEV = 64;
std::mutex g_zp_mutex;
std::atomic<float*> g_zp[MAX_DEV] = {};

const float* zero_page_of(int dev) {
    if (dev < 0 || dev >= MAX_DEV) return nullptr;
    if (float* p = g_zp[dev].load(std::memory_order_acquire)) return p;      // per-launch fast path: no lock
    std::lock_guard<std::mutex> lock(g_zp_mutex);
    if (!g_zp[dev].load(std::memory_order_relaxed)) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return nullptr;
        if (cur != dev && hipSetDevice(dev) != hipSuccess) return nullptr;
        void* q = nullptr;
        const bool ok = hipMalloc(&q, 256) == hipSuccess && hipMemset(q, 0, 256) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
        if (cur != dev) (void)hipSetDevice(cur);
        if (!ok) return nullptr;
        g_zp[dev].store(static_cast<float*>(q), std::memory_order_release);
    }
    return g_zp[dev].load(std::memory_order_relaxed);
}
}  // namespace

const float* fgt_zero_page() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    return zero_page_of(dev);
}

extern "C" int fgt_init(int device) {
    if (device < 0 && hipGetDevice(&device) != hipSuccess) {
        fgt_set_error("fgt_init: no current HIP device");
        return FGT_ELAUNCH;
    }
    if (!zero_page_of(device)) {
        fgt_set_error("fgt_init: could not allocate the zero page on device %d", device);
        return FGT_ELAUNCH;
    }
    return FGT_OK;
}
