// Micro-benchmark: sustained global -> LDS rate per CU of global_load_lds_dwordx4 (and, for comparison, global_load_dwordx4 into
// registers) as a function of the access shape (bytes a row contributes to one 1-KB wave instruction: 64 / 128 / 1024), of the
// footprint (L1-sized / L2-sized / beyond L2) and of the wavefronts per CU.  Every CU gets the same number of workgroups in one round.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_dma_rate.hip -o tools/micro/bin/lds_dma_rate && tools/micro/bin/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One workgroup = NW wavefronts.  Per iteration every wavefront issues PIECES 1-KB instructions, then waits until at most PIECES are
// outstanding (two iterations in flight).  The workgroup walks a [rows x row_stride] byte matrix like an implicit-GEMM A tile: its
// NW * PIECES * (1024 / seg) rows are fixed, the k offset advances by `seg` bytes per iteration and wraps at `kwrap`.
template <int MODE, int PIECES>   // MODE 0: LDS-DMA, 1: loads into registers
__global__ void __launch_bounds__(1024) rate_kernel(const char* src, long row_stride, int seg, int kwrap, int iters, long wg_rows_stride, int row_mod,
                                                   unsigned long long* cycles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lanes_per_row = seg / 16, rows_per_piece = 64 / lanes_per_row;
    const int r = lane / lanes_per_row, c = lane % lanes_per_row;
    const char* base[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        long row = (long)blockIdx.x * wg_rows_stride + (long)(wave * PIECES + p) * rows_per_piece + r;
        if (row_mod) row %= row_mod;
        base[p] = src + row * row_stride + c * 16;
    }
    char* dst = lds + wave * PIECES * 1024;
    f32x4 regs[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) regs[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            if constexpr (MODE == 0) {
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(base[p] + k), (lds_ptr_t)(dst + p * 1024), 16, 0, 0);
            } else {
                // "+v": the destination stays live across the loop, so the allocator cannot hand these registers to an address while a load
                // is still in flight (the compiler does not see the asynchronous write)
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(regs[p]) : "v"(base[p] + k));
            }
        }
        k += seg;
        if (k >= kwrap) k = 0;
        if constexpr (PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if constexpr (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) acc += regs[p][0];
    if (sink && lane == 0) sink[blockIdx.x * nw + wave] = acc + lds[threadIdx.x];
}

// The conv's A (im2col) stream of a 3x3 layer, to compare K orders: every workgroup owns 128 consecutive pixels (4 workgroups of the
// same XCD share them, like the 4 N tiles of one M tile), a step fetches rows pix+tap of one 32-channel chunk of both planes
// (16 instructions per workgroup) plus a 16-KB weight slab that every workgroup shares.  ORDER 0: tap-major (ky, kx, chunk) — the
// library's K order: the kx re-read of a row comes CHUNKS steps later; ORDER 1: (ky, chunk, kx) — it comes in the very next step.
template <int ORDER>
__global__ void __launch_bounds__(512) conv_like_kernel(const char* act, const char* wts, int chunks, int width, int mtiles, int lds_reads, float* sink,
                                                        unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;     // 8 wavefronts: A pieces (plane, 16-row group) = wave + 8*i, i < 2; B 2 pieces
    const int w = blockIdx.x;
    // mtiles == 0: 128 M tiles, one round of workgroups (44-MB footprint).  mtiles > 0: the library's XCD-aware order over `mtiles` M tiles
    // x 4 N tiles (csrc/conv_tile.h) — the whole layer's input is streamed, round after round of workgroups, like the real launch
    long m_tile = (long)(w >> 5) * 8 + (w & 7);
    long rows_total = (long)gridDim.x / 4 * 128;
    if (mtiles > 0) {
        const int mchunk = (mtiles + 7) / 8, xcd = w & 7, i = w >> 3;
        m_tile = (long)xcd * mchunk + i / 4;
        if (m_tile >= mtiles) return;
        rows_total = (long)mtiles * 128;
    }
    const long row_bytes = (long)chunks * 64;                       // one plane of a pixel: chunks * 32 channels * 2 B
    const long plane_bytes = row_bytes * (rows_total + 4 * width + 256);
    const int r = lane >> 2, c = lane & 3;
    char* dst = lds + wave * 4 * 1024;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int step = 0;
    float facc = 0.f;
    for (int ky = 0; ky < 3; ++ky)
        for (int a = 0; a < (ORDER ? chunks : 3); ++a)
            for (int b = 0; b < (ORDER ? 3 : chunks); ++b, ++step) {
                const int kx = ORDER ? b : a, ch = ORDER ? a : b;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int piece = wave + 8 * i, plane = piece >> 3, grp = piece & 7;
                    const long pix = m_tile * 128 + grp * 16 + r + (long)ky * width + kx;
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(act + plane * plane_bytes + pix * row_bytes + ch * 64 + c * 16), (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    __builtin_amdgcn_global_load_lds((glb_ptr_t)(wts + ((long)step * 16 + wave * 2 + i) * 1024 + lane * 16), (lds_ptr_t)(dst + (2 + i) * 1024), 16, 0, 0);
                // the conv's fragment traffic: lds_reads conflict-free ds_read_b128 per wavefront and step (12 in the 128x128 8-wavefront tile)
                for (int q = 0; q < lds_reads; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(lds + ((wave * 12 + q) * 1024 + lane * 16) % (64 * 1024));
                    facc += v[0];
                }
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) cycles[blockIdx.x] = __builtin_readcyclecounter() - t0;
    if (sink) sink[blockIdx.x * 512 + threadIdx.x] = facc;
}

template <int ORDER>
void run_conv_like(const char* src, int chunks, int width, int mtiles = 0, int lds_reads = 0) {
    const int grid = mtiles ? 8 * ((mtiles + 7) / 8) * 4 : 512;
    unsigned long long* cyc;
    hipMalloc(&cyc, grid * sizeof(unsigned long long));
    hipMemset(cyc, 0, grid * sizeof(unsigned long long));
    const size_t smem = 79 * 1024;                                   // two workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_like_kernel<ORDER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const char* wts = src + (768u << 20);                            // weight slabs in the last quarter of the buffer
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((conv_like_kernel<ORDER>), dim3(grid), dim3(512), smem, 0, src, wts, chunks, width, mtiles, lds_reads, (float*)nullptr, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    int n = 0;
    for (auto v : h) if (v) { mean += (double)v; ++n; }
    mean /= n;
    const int steps = 9 * chunks;
    printf("conv-like A+B stream + %2d ds_read_b128 per wavefront and step, %2d chunks of 32 channels, order %s, %s: %7.0f cycles per step (2 workgroups per CU, 32 KB per step each) = %5.1f B/clk/CU\n",
           lds_reads, chunks, ORDER ? "(ky, chunk, kx)" : "(ky, kx, chunk)", mtiles ? "whole layer (20 frames of 60x108, 4 N tiles)" : "one round of workgroups", mean / steps,
           2.0 * 32768 * steps / mean);
    hipFree(cyc);
}

template <int MODE, int PIECES>
void run(const char* src, size_t bytes, int nw, int wgs_per_cu, int seg, long row_stride, int kwrap, const char* label, int row_mod = 0) {
    const int grid = 256 * wgs_per_cu, iters = 400;
    unsigned long long* cyc;
    hipMalloc(&cyc, grid * sizeof(unsigned long long));
    const int rows_per_wg = nw * PIECES * (64 / (seg / 16));
    long wg_rows_stride = rows_per_wg;
    if ((long)grid * rows_per_wg * row_stride > (long)bytes) wg_rows_stride = 0;   // small footprint: every workgroup reads the same rows
    const size_t smem = 160 * 1024 / wgs_per_cu - 1024;                           // forces exactly wgs_per_cu workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&rate_kernel<MODE, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL((rate_kernel<MODE, PIECES>), dim3(grid), dim3(nw * 64), smem, 0, src, row_stride, seg, kwrap, iters, wg_rows_stride, row_mod, cyc, nullptr);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<MODE, PIECES>), dim3(grid), dim3(nw * 64), smem, 0, src, row_stride, seg, kwrap, iters, wg_rows_stride, row_mod, cyc, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= grid;
    const double bytes_wg = (double)iters * nw * PIECES * 1024;
    const double foot = row_mod ? (double)row_mod * kwrap : wg_rows_stride ? (double)grid * rows_per_wg * kwrap : (double)rows_per_wg * kwrap;
    printf("%-4s seg %4d B  %2d waves x %d WG/CU  pieces/iter %d  footprint %8.2f MB (%s): %6.1f B/clk/CU (%5.0f cycles per 1-KB instr per CU), %6.2f TB/s chip\n",
           MODE == 0 ? "DMA" : "REG", seg, nw, wgs_per_cu, PIECES, foot / 1e6, label, bytes_wg * wgs_per_cu / mean, mean / (iters * nw * PIECES * wgs_per_cu),
           bytes_wg * grid / (ms * 1e-3) / 1e12);
    hipFree(cyc);
}

int main() {
    const size_t bytes = 1ull << 30;
    char* src;
    hipMalloc(&src, bytes);
    hipMemset(src, 1, bytes);
    // conv-like: rows 1280 B apart (640 channels of bf16); footprints: 8 KB (L1), 2.6 MB shared by every workgroup (fits each XCD's L2),
    // 84 MB (Infinity Cache), 1 GB streamed once (HBM)
    for (int seg : {64, 128, 1024}) {
        const long stride = seg == 1024 ? 1024 : 1280;
        const int wrap = seg == 1024 ? 1024 : 1280;
        run<0, 4>(src, bytes, 8, 2, seg, stride, 128 < seg ? seg : 128, "L1", 64);
        run<1, 4>(src, bytes, 8, 2, seg, stride, 128 < seg ? seg : 128, "L1", 64);
        run<0, 4>(src, bytes, 8, 2, seg, stride, wrap, "L2", 2048);
        run<1, 4>(src, bytes, 8, 2, seg, stride, wrap, "L2", 2048);
        run<0, 4>(src, bytes, 8, 2, seg, stride, wrap, "Infinity Cache", 65536);
    }
    // wavefront count / depth at the conv's shape (64-B segments), L2-resident
    for (int nw : {4, 8, 16}) run<0, 4>(src, bytes, nw, 1, 64, 1280, 1280, "L2", 2048);
    run<0, 8>(src, bytes, 8, 2, 64, 1280, 1280, "L2, 16 KB/wave in flight", 2048);
    run<0, 2>(src, bytes, 8, 2, 64, 1280, 1280, "L2, 4 KB/wave in flight", 2048);
    run<0, 8>(src, bytes, 8, 1, 128, 1280, 1280, "L2", 2048);
    run<0, 8>(src, bytes, 16, 1, 128, 1280, 1280, "L2", 2048);
    run<0, 4>(src, bytes, 16, 2, 64, 1280, 1280, "L2, 32 waves/CU", 2048);
    // streaming (no reuse): every row read once
    run<0, 4>(src, bytes, 8, 2, 128, 8192, 8192, "streaming: HBM");
    run<1, 4>(src, bytes, 8, 2, 128, 8192, 8192, "streaming: HBM");
    // K order of a 3x3 layer's im2col stream (see conv_like_kernel)
    for (int chunks : {4, 8, 10, 20}) {
        run_conv_like<0>(src, chunks, 108);
        run_conv_like<1>(src, chunks, 108);
    }
    run_conv_like<0>(src, 10, 108, 1013);      // the e20 enc10 launch per group: 129 600 output pixels, 320 channels per group
    run_conv_like<0>(src, 20, 108, 1013);
    for (int r : {6, 12, 24}) run_conv_like<0>(src, 10, 108, 0, r);      // + the fragment reads of the conv (12 per wavefront and step)
    run_conv_like<0>(src, 10, 108, 1013, 12);
    hipFree(src);
    return 0;
}
