// Shared helpers for the gfx950 kernels of libfgt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/fgt_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void fgt_set_error(const char* fmt, ...);

// runtime.hip: per-launch event timing (no-ops returning -1 unless fgt_prof_enable(1))
bool fgt_prof_on();
int fgt_prof_begin(int kind, double flops, double bytes, hipStream_t s);
void fgt_prof_end(int idx, hipStream_t s);

// brackets the launches of one C-ABI call with an event pair (no-op unless fgt_prof_enable(1)); `bytes` = algorithmic bytes of the call
struct FgtProfScope {
    int idx;
    hipStream_t s;
    FgtProfScope(int kind, double flops, double bytes, void* stream) : idx(fgt_prof_begin(kind, flops, bytes, (hipStream_t)stream)), s((hipStream_t)stream) {}
    ~FgtProfScope() { fgt_prof_end(idx, s); }
    FgtProfScope(const FgtProfScope&) = delete;
    FgtProfScope& operator=(const FgtProfScope&) = delete;
};

#define FGT_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            fgt_set_error(__VA_ARGS__); \
            return FGT_EINVAL;          \
        }                               \
    } while (0)

static inline int fgt_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fgt_set_error("%s: %s", what, hipGetErrorString(e));
        return FGT_ELAUNCH;
    }
    return FGT_OK;
}

__device__ __forceinline__ float fgt_act(float v, int act, float slope) {
    switch (act) {
        case FGT_ACT_LRELU: return v > 0.f ? v : v * slope;
        case FGT_ACT_RELU: return fmaxf(v, 0.f);
        case FGT_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case FGT_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// One-time, per-device raise of a kernel's dynamic-LDS limit (function attributes are per device; two racing threads both
// set the same value, which is harmless).  `done` is a per-kernel-instance bit mask of devices already configured.
static inline int fgt_set_max_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done, const char* what) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    if ((done.load(std::memory_order_acquire) >> dev) & 1ull) return FGT_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        fgt_set_error("hipFuncSetAttribute(%s, %d bytes): %s", what, bytes, hipGetErrorString(e));
        return FGT_ELAUNCH;
    }
    done.fetch_or(1ull << dev, std::memory_order_release);
    return FGT_OK;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hi/lo split of a float4 run: packed bf16 {hi0,hi1},{hi2,hi3} and {lo0,lo1},{lo2,lo3} with hi = bf16_rne(x),
// lo = bf16_rne(x - hi) (one v_cvt_pk_bf16_f32 per pair).  THE definition of the split format: every producer uses it.
typedef __bf16 fgt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float fgt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fgt_split4(const float4 v, uint2& hi, uint2& lo) {
    const fgt_f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    const unsigned ha = __builtin_bit_cast(unsigned, __builtin_convertvector(a, fgt_bf16x2));
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(b, fgt_bf16x2));
    const fgt_f32x2 la = {v.x - __builtin_bit_cast(float, ha << 16), v.y - __builtin_bit_cast(float, ha & 0xFFFF0000u)};
    const fgt_f32x2 lb = {v.z - __builtin_bit_cast(float, hb << 16), v.w - __builtin_bit_cast(float, hb & 0xFFFF0000u)};
    hi = make_uint2(ha, hb);
    lo = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(la, fgt_bf16x2)),
                    __builtin_bit_cast(unsigned, __builtin_convertvector(lb, fgt_bf16x2)));
}

// The fp16 activation format (fgt_conv_desc.in_split = 3, FGT_PREC_F16): ONE plane, h = f16_rne(clamp(x, +-65504)).  The clamp keeps a
// stray out-of-range activation finite (fp16 has no room above 65504: without it the value would turn into inf and the next GEMM
// into NaN).  THE definition of the format: every producer uses it (plane stride argument -1).
typedef _Float16 fgt_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 fgt_half4(const float4 v) {
    const fgt_f32x2 a = {__builtin_amdgcn_fmed3f(v.x, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.y, -65504.f, 65504.f)};
    const fgt_f32x2 b = {__builtin_amdgcn_fmed3f(v.z, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v.w, -65504.f, 65504.f)};
    return make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, fgt_f16x2)),
                      __builtin_bit_cast(unsigned, __builtin_convertvector(b, fgt_f16x2)));
}
