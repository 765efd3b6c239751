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
extern "C" int fgt_abi_version(void) { return 9; }

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
unsigned g_prof_mask = 0;      // bit k: launches of kind k record events
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_event_pool;

hipEvent_t get_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
}  // namespace

bool fgt_prof_on() { return g_prof_mask != 0; }

int fgt_prof_begin(int kind, double flops, double bytes, hipStream_t s) {
    if (kind < 0 || kind > 31 || !((g_prof_mask >> kind) & 1u)) return -1;
    ProfRec r{get_event(), get_event(), flops, bytes, kind};
    hipEventRecord(r.a, s);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}

void fgt_prof_end(int idx, hipStream_t s) {
    if (idx >= 0 && idx < (int)g_prof.size()) hipEventRecord(g_prof[idx].b, s);
}

extern "C" void fgt_prof_enable(int on) { g_prof_mask = on ? ~0u : 0u; }
extern "C" void fgt_prof_enable_kinds(unsigned mask) { g_prof_mask = mask; }

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

// ---- sustained matrix-core rate of THIS chip under load (bench.py reports it beside the nominal peak) ------------------------------
// 256 workgroups x 8 wavefronts issue nothing but independent MFMAs on random operands held in registers; every workgroup measures the
// shader clock it ran at as d(s_memtime) / d(s_memrealtime) (s_memrealtime: constant 100 MHz).  The chip clocks to its power budget:
// on random data the bf16 pipe sustains ~1.8 PF at ~1.75 GHz, not the nominal 2.5 PF at 2.4 GHz (tools/micro/clock_probe.hip).
namespace {
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned probe_hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <bool F32>
__global__ void __launch_bounds__(512) mfma_probe_kernel(float* out, int iters, unsigned long long* cyc, unsigned long long* real) {
    probe_bf16x8 a[2], b[2];
    float fa[2], fb[2];
    for (int j = 0; j < 2; ++j) {
        for (int i = 0; i < 8; ++i) {
            const unsigned h = probe_hash(threadIdx.x * 64 + blockIdx.x * 4096 + i * 2 + j);
            a[j][i] = (__bf16)((float)(h & 0xffff) / 65536.f - 0.5f);
            b[j][i] = (__bf16)((float)(h >> 16) / 65536.f - 0.5f);
        }
        const unsigned h = probe_hash(threadIdx.x * 2 + blockIdx.x * 1024 + j + 77);
        fa[j] = (float)(h & 0xffffff) / 16777216.f - 0.5f;
        fb[j] = (float)(h >> 8) / 16777216.f - 0.5f;
    }
    probe_f32x16 acc[4];
    for (int n = 0; n < 4; ++n)
        for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if constexpr (F32) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[n & 1], fb[r], acc[n], 0, 0, 0);
                else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n & 1], b[r], acc[n], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < 4; ++n)
        for (int e = 0; e < 16; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; real[blockIdx.x] = r1 - r0; }
}
}  // namespace

extern "C" long fgt_mfma_probe_workspace(void) { return 256L * 512 * 4 + 2 * 256 * 8; }

extern "C" int fgt_mfma_probe(int f32, int iters, void* workspace, double* tflops, double* ghz, void* stream) {
    if (!workspace || iters <= 0) { fgt_set_error("fgt_mfma_probe: workspace / iters"); return FGT_EINVAL; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* out = static_cast<float*>(workspace);
    unsigned long long* cyc = reinterpret_cast<unsigned long long*>(out + 256 * 512);
    unsigned long long* real = cyc + 256;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { fgt_set_error("fgt_mfma_probe: hipEventCreate"); return FGT_ELAUNCH; }
    // warm the clocks into their loaded state, then the timed launch
    for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) hipEventRecord(e0, s);
        if (f32) hipLaunchKernelGGL(mfma_probe_kernel<true>, dim3(256), dim3(512), 0, s, out, iters, cyc, real);
        else hipLaunchKernelGGL(mfma_probe_kernel<false>, dim3(256), dim3(512), 0, s, out, iters, cyc, real);
    }
    hipEventRecord(e1, s);
    int rc = fgt_check_launch("mfma_probe");
    if (rc == FGT_OK && hipEventSynchronize(e1) != hipSuccess) { fgt_set_error("fgt_mfma_probe: sync"); rc = FGT_ELAUNCH; }
    if (rc == FGT_OK) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(512);
        if (hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost) != hipSuccess) { fgt_set_error("fgt_mfma_probe: copy"); rc = FGT_ELAUNCH; }
        else {
            double g = 0;
            for (int i = 0; i < 256; ++i) g += (double)h[i] / ((double)h[256 + i] * 10.0);
            const double flops_per_mfma = f32 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
            if (tflops) *tflops = (double)iters * 8 * flops_per_mfma * 256 * 8 / (ms * 1e-3) / 1e12;
            if (ghz) *ghz = g / 256;
        }
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}
