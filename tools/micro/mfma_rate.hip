// Micro-benchmark: issue rate of v_mfma_f32_32x32x16_bf16 / 16x16x32 as a function of the number of independent accumulators per
// wavefront and of the wavefronts per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + threadIdx.x * 0.001f + i); b[i] = (__bf16)(seed * 0.5f + i); }
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) acc[n][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int e = 0; e < 4; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int nacc, int waves_per_simd, double flops_per_mfma) {
    float* out; hipMalloc(&out, 256 * 16 * 64 * 4 * sizeof(float));
    const int iters = 2000;
    dim3 grid(256), block(64 * 4 * waves_per_simd);     // one workgroup per CU, waves_per_simd wavefronts on each of the 4 SIMDs
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 10, 1.0f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 8 * nacc * 256 * 4 * waves_per_simd;
    const double per_simd = (double)iters * 8 * nacc * waves_per_simd;
    printf("%-10s acc=%d waves/SIMD=%d: %.3f ms, %.0f TFLOP/s, %.1f ns per MFMA per SIMD\n", name, nacc, waves_per_simd, ms,
           mfmas * flops_per_mfma / ms / 1e9, ms * 1e6 / per_simd);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        run("32x32x16", k32<1>, 1, w, 32768); run("32x32x16", k32<2>, 2, w, 32768); run("32x32x16", k32<4>, 4, w, 32768);
        run("32x32x16", k32<8>, 8, w, 32768);
    }
    for (int w : {1, 2, 4}) {
        run("16x16x32", k16<1>, 1, w, 16384); run("16x16x32", k16<2>, 2, w, 16384); run("16x16x32", k16<4>, 4, w, 16384);
        run("16x16x32", k16<8>, 8, w, 16384);
    }
    return 0;
}
