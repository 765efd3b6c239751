// Error reporting and ABI version of libfgt_hip.so.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void fgt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* fgt_last_error(void) { return g_err; }
extern "C" int fgt_abi_version(void) { return 2; }

const float* fgt_zero_page() {
    static float* zp = nullptr;
    if (!zp) {
        void* q = nullptr;
        if (hipMalloc(&q, 256) != hipSuccess) return nullptr;
        if (hipMemset(q, 0, 256) != hipSuccess) return nullptr;
        zp = static_cast<float*>(q);
    }
    return zp;
}
