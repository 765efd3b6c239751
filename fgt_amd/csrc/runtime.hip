// Error reporting, ABI version and the per-device zero page of libfgt_hip.so.
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include <vector>
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

// ---- optional per-launch timing with HIP events on the launch stream (bench.py's roofline blocks) --------------------------------
// A launcher brackets its kernel with fgt_prof_begin / fgt_prof_end; records carry a kind (FGT_PROF_*) and the launch's ALGORITHMIC
// flop count and its unique-byte floor (every input / output byte once).  fgt_prof_collect_kind synchronises the events of one kind and returns its totals.  Single host thread (the bench).
namespace {
struct ProfRec { hipEvent_t a, b; double flops, bytes; int kind; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t get_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
}  // namespace

bool fgt_prof_on() { return g_prof_on; }

int fgt_prof_begin(int kind, double flops, double bytes, hipStream_t s) {
    if (!g_prof_on) return -1;
    ProfRec r{get_event(), get_event(), flops, bytes, kind};
    hipEventRecord(r.a, s);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}

void fgt_prof_end(int idx, hipStream_t s) {
    if (idx >= 0 && idx < (int)g_prof.size()) hipEventRecord(g_prof[idx].b, s);
}

extern "C" void fgt_prof_enable(int on) { g_prof_on = on != 0; }

extern "C" int fgt_prof_collect_kind(int kind, double* total_ms, double* total_flops, double* total_bytes, long* launches) {
    double ms = 0, fl = 0, by = 0;
    long n = 0;
    std::vector<ProfRec> keep;
    for (auto& r : g_prof) {
        if (kind >= 0 && r.kind != kind) { keep.push_back(r); continue; }
        hipEventSynchronize(r.b);
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) { fgt_set_error("hipEventElapsedTime failed"); return FGT_ELAUNCH; }
        ms += t; fl += r.flops; by += r.bytes; ++n;
        g_event_pool.push_back(r.a); g_event_pool.push_back(r.b);
    }
    g_prof.swap(keep);
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (total_bytes) *total_bytes = by;
    if (launches) *launches = n;
    return FGT_OK;
}

extern "C" int fgt_prof_collect(double* total_ms, double* total_flops, long* launches) {
    return fgt_prof_collect_kind(FGT_PROF_CONV, total_ms, total_flops, nullptr, launches);
}
