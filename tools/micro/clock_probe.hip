// Micro-benchmark: the shader clock the chip sustains under matrix-core load, measured inside the kernel as d(s_memtime) / d(s_memrealtime)
// (s_memrealtime ticks at a constant 100 MHz), and the resulting bf16 MFMA rate — for constant operands, random operands, and random
// operands re-read from LDS every step (the power the fragment traffic adds).  The nominal 2.5 PF peak assumes 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_probe.hip -o tools/micro/bin/clock_probe && tools/micro/bin/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>   // 0: constant operands, 1: random operands in registers, 2: random operands read from LDS before every 8 MFMAs
__global__ void __launch_bounds__(512) probe(float* out, int iters, unsigned long long* cyc, unsigned long long* real) {
    __shared__ __attribute__((aligned(16))) bf16x8 tile[8 * 64 * 4];
    bf16x8 a[2], b[2];
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 8; ++i) {
            const unsigned h = hash(threadIdx.x * 64 + blockIdx.x * 4096 + i * 2 + j);
            a[j][i] = MODE == 0 ? (__bf16)1.0f : (__bf16)((float)(h & 0xffff) / 65536.f - 0.5f);
            b[j][i] = MODE == 0 ? (__bf16)1.0f : (__bf16)((float)(h >> 16) / 65536.f - 0.5f);
        }
    for (int j = 0; j < 4; ++j) tile[threadIdx.x * 4 + j] = j & 1 ? b[j >> 1] : a[j >> 1];
    __syncthreads();
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 2) {
            a[0] = tile[threadIdx.x * 4 + 0]; b[0] = tile[threadIdx.x * 4 + 1];
            a[1] = tile[threadIdx.x * 4 + 2]; b[1] = tile[threadIdx.x * 4 + 3];
            asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n & 1], b[r], acc[n], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int n = 0; n < 4; ++n) for (int e = 0; e < 16; ++e) s += acc[n][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; real[blockIdx.x] = r1 - r0; }
}

template <int MODE>
void run(const char* what, int waves_per_simd) {
    const int grid = 256, nt = 64 * 4 * waves_per_simd, iters = 40000;
    float* out; unsigned long long *cyc, *real;
    hipMalloc(&out, grid * nt * sizeof(float)); hipMalloc(&cyc, grid * 8); hipMalloc(&real, grid * 8);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(nt), 0, 0, out, 2000, cyc, real);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(nt), 0, 0, out, iters, cyc, real);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(grid), r(grid);
    hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost); hipMemcpy(r.data(), real, grid * 8, hipMemcpyDeviceToHost);
    double ghz = 0; for (int i = 0; i < grid; ++i) ghz += (double)c[i] / ((double)r[i] * 10.0); ghz /= grid;
    const double fl = (double)iters * 8 * 32768.0 * grid * 4 * waves_per_simd;
    printf("%-44s %d waves/SIMD: %7.2f ms  %6.0f TFLOP/s  shader clock %.2f GHz  (%.1f cycles per MFMA per SIMD)\n", what, waves_per_simd, ms, fl / ms / 1e9, ghz,
           (double)c[0] / ((double)iters * 8 * waves_per_simd));
    hipFree(out); hipFree(cyc); hipFree(real);
}

int main() {
    for (int w : {1, 2}) {
        run<0>("constant operands (1.0)", w);
        run<1>("random operands in registers", w);
        run<2>("random operands, re-read from LDS per 8 MFMAs", w);
    }
    return 0;
}
