// Micro-benchmark: what does one wavefront pay to get a 1-KB LDS-DMA instruction ACCEPTED, and what rate does a CU sustain — by instruction
// form (global_load_lds_dwordx4 with a 64-bit address per lane vs buffer_load_dwordx4 ... offen lds with a 32-bit offset per lane into a
// buffer resource), by piece shape (16 rows x 64 B: half cache lines, the conv kernels' "narrow" pieces / 8 rows x 128 B: full lines) and by
// the number of wavefronts issuing at once.  Footprint: a [rows x 1280 B] matrix walked like an implicit-GEMM A tile (L2-resident).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_issue.hip -o tools/micro/bin/dma_issue && tools/micro/bin/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int FORM, int PIECES>          // FORM 0: global_load_lds, 1: buffer_load ... lds
__global__ void __launch_bounds__(1024) issue_kernel(const char* src, long bytes, int seg, int iters, int rows_total, unsigned long long* issue_cyc,
                                                    unsigned long long* total_cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lpr = seg / 16, rpp = 64 / lpr;                  // lanes per row, rows per piece
    const int r = lane / lpr, c = lane % lpr;
    unsigned off[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        long row = ((long)blockIdx.x * nw * PIECES + wave * PIECES + p) * rpp + r;
        row %= rows_total;
        off[p] = (unsigned)(row * 1280 + c * 16);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)bytes, 0x00020000);
#endif
    char* dst = lds + wave * PIECES * 1024;
    int k = 0;
    unsigned long long issue = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned long long a = __builtin_readcyclecounter();
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (FORM == 0) __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + off[p] + k), (lds_ptr_t)(dst + p * 1024), 16, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(dst + p * 1024), 16, off[p] + k, 0, 0, 0);
#endif
        }
        issue += __builtin_readcyclecounter() - a;
        k += seg;
        if (k >= 1280) k = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // one burst at a time: the issue cost of a burst of PIECES against an idle queue of this wave
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) issue_cyc[blockIdx.x * nw + wave] = issue;
    if (threadIdx.x == 0) total_cyc[blockIdx.x] = t1 - t0;
}

template <int FORM, int PIECES>
void run(const char* src, long bytes, int nw, int seg, const char* what) {
    const int wgs = 256, iters = 400, rows_total = 2048;
    unsigned long long *ic, *tc;
    hipMalloc(&ic, sizeof(unsigned long long) * wgs * nw);
    hipMalloc(&tc, sizeof(unsigned long long) * wgs);
    const size_t smem = (size_t)nw * PIECES * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&issue_kernel<FORM, PIECES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((issue_kernel<FORM, PIECES>), dim3(wgs), dim3(nw * 64), smem, 0, src, bytes, seg, iters, rows_total, ic, tc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hi(wgs * nw), ht(wgs);
    hipMemcpy(hi.data(), ic, sizeof(unsigned long long) * wgs * nw, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), tc, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
    std::sort(hi.begin(), hi.end());
    std::sort(ht.begin(), ht.end());
    const double issue_per_piece = (double)hi[hi.size() / 2] / iters / PIECES;
    const double cyc_per_piece_cu = (double)ht[ht.size() / 2] / iters / (nw * PIECES);
    printf("%-26s %2d waves x %d pieces of %4d-B rows: issue %6.1f cycles per piece per wavefront | burst round trip %7.1f cycles | %5.1f cycles per piece per CU = %5.1f B/clk/CU  (%s)\n",
           FORM ? "buffer_load..offen lds" : "global_load_lds", nw, PIECES, seg, issue_per_piece, (double)ht[ht.size() / 2] / iters, cyc_per_piece_cu, 1024.0 / cyc_per_piece_cu, what);
    hipFree(ic);
    hipFree(tc);
}

int main() {
    const long bytes = 256l << 20;
    char* src;
    hipMalloc(&src, bytes);
    hipMemset(src, 1, bytes);
    for (int seg : {64, 128}) {
        for (int nw : {1, 4, 8, 16}) {
            run<0, 4>(src, bytes, nw, seg, "L2");
            run<1, 4>(src, bytes, nw, seg, "L2");
        }
        run<0, 8>(src, bytes, 4, seg, "L2");
        run<1, 8>(src, bytes, 4, seg, "L2");
        run<0, 2>(src, bytes, 8, seg, "L2");
        run<1, 2>(src, bytes, 8, seg, "L2");
    }
    hipFree(src);
    return 0;
}
