// Shared helpers for the gfx950 kernels of libfgt_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/fgt_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void fgt_set_error(const char* fmt, ...);

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

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
