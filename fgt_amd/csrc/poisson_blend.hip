// Poisson blending of the propagated gradients into the frames: tool/utils/Poisson_blend_img.py:19-244 as called per
// frame by tool/video_inpainting.py:644-682, for all frames and colour channels of a clip at once.
//
// The reference assembles, per frame, an over-determined sparse system with up to four equations per hole pixel p (one per 4-neighbour q
// whose connecting gradient is valid: inside the image and outside the gradient mask):
//     q known:  x_p       = T_q + r_pq            r_pq = -gx[p] (right), +gx[left of p], -gy[p] (down), +gy[above p]
//     q hole:   x_p - x_q = r_pq
// and solves it in the least-squares sense with scipy's LSQR in float64, channel by channel (:37-44).  Every interior edge appears twice
// (once from each end, the same equation negated), so the normal equations are a weighted graph Laplacian:
//     d_p x_p - sum_{q hole} 2 x_q = sum_{q hole} 2 r_pq + sum_{q known} (r_pq + T_q),    d_p = 2 #hole-edges + #known-edges,
// symmetric positive (semi-)definite: conjugate gradients from x = 0 converges to the least-squares solution LSQR iterates towards
// (components with no known neighbour are singular; CG from 0 gives their minimum-norm solution like LSQR, and those pixels are the
// ones `UnfilledMask` reports, which the tool overwrites afterwards).
//
// GPU shape (as csrc/laplace_fill.hip): one problem per (frame, channel) on blockIdx.y, 1024 pixels per workgroup, threads at non-hole
// pixels exit; ordered two-stage double-precision dot products (no atomics: bit-reproducible); a problem whose residual fell below
// tol * |r0| freezes; fixed iteration count, nothing is read back.  HBM-bound stencil work, ~20 float accesses per hole pixel and
// iteration.  `UnfilledMask` (:143-168) is a raster-order reachability recurrence: one workgroup per frame walks the rows, each row a
// parallel scan over the columns of the boolean maps c -> a | (b & c).
#include "common.h"

namespace {

constexpr int PPB = 1024;

struct BlendP {
    const float* trg;              // [N, H, W, 3] target frames (0..1)
    const float *gx, *gy;          // [N, H, W, 3] propagated gradients (gx[..., x] = I[x+1] - I[x]; last column / row unused)
    const unsigned char *hole, *gmask;   // [N, H, W]
    unsigned char* ecode;          // [N, H, W]: 2 bits per direction (0 right, 1 down, 2 left, 3 up): 0 none, 1 known neighbour, 2 hole neighbour
    float *x, *r, *p0, *p1, *q;    // [N*3, H, W] planar per (frame, channel)
    double *prr0, *prr, *ppq;
    int N, H, W, nblk;
    float tol2;
};

__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ double total(const double* part, int nblk, double* sh) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) v += part[i];
    return block_sum(v, sh);
}

// equation set of every hole pixel (Poisson_blend_img.py:171-199: validNeighbor * HaveGrad * Boundary / NonBoundary; `edge` is all zero)
__global__ void __launch_bounds__(256) blend_codes(const BlendP P) {
    const long HW = (long)P.H * P.W, total_px = (long)P.N * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_px; i += (long)gridDim.x * blockDim.x) {
        unsigned code = 0;
        if (P.hole[i]) {
            const long f = i / HW, pix = i - f * HW;
            const int y = (int)(pix / P.W), x = (int)(pix - (long)y * P.W);
            const bool g0 = P.gmask[i] == 0;                                       // gradient at p itself (right / down differences)
            if (x + 1 < P.W && g0) code |= (P.hole[i + 1] ? 2u : 1u);
            if (y + 1 < P.H && g0) code |= (P.hole[i + P.W] ? 2u : 1u) << 2;
            if (x > 0 && P.gmask[i - 1] == 0) code |= (P.hole[i - 1] ? 2u : 1u) << 4;
            if (y > 0 && P.gmask[i - P.W] == 0) code |= (P.hole[i - P.W] ? 2u : 1u) << 6;
        }
        P.ecode[i] = (unsigned char)code;
    }
}

__global__ void __launch_bounds__(256) blend_init(const BlendP P) {
    __shared__ double sh[4];
    const int b = blockIdx.y, f = b / 3, c = b - 3 * f;
    const long HW = (long)P.H * P.W, base = (long)b * HW, fb = (long)f * HW;
    double acc = 0.0;
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * PPB + k * 256 + threadIdx.x;
        if (i >= HW) continue;
        const unsigned code = P.ecode[fb + i];
        if (!P.hole[fb + i]) { P.x[base + i] = P.trg[(fb + i) * 3 + c]; continue; }   // known pixels carry the target (read as neighbours never: see apply)
        const float rr[4] = {-P.gx[(fb + i) * 3 + c], -P.gy[(fb + i) * 3 + c],
                             (code >> 4) & 3 ? P.gx[(fb + i - 1) * 3 + c] : 0.f, (code >> 6) & 3 ? P.gy[(fb + i - P.W) * 3 + c] : 0.f};
        const int off[4] = {1, P.W, -1, -P.W};
        float rhs = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const unsigned e = (code >> (2 * n)) & 3u;
            if (e == 2) rhs += 2.f * rr[n];
            else if (e == 1) rhs += rr[n] + P.trg[(fb + i + off[n]) * 3 + c];
        }
        P.x[base + i] = 0.f;
        P.r[base + i] = rhs;
        P.p1[base + i] = 0.f;
        acc += (double)rhs * rhs;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) {
        P.prr0[(long)b * P.nblk + blockIdx.x] = s;
        P.prr[(long)b * P.nblk + blockIdx.x] = s;
    }
}

__global__ void __launch_bounds__(256) blend_apply(const BlendP P, int k) {
    __shared__ double sh[4];
    const int b = blockIdx.y, f = b / 3;
    const long HW = (long)P.H * P.W, base = (long)b * HW, fb = (long)f * HW;
    const long pb = (long)b * P.nblk, B = (long)P.N * 3;
    const int parity = k & 1;
    const double rr0 = total(P.prr0 + pb, P.nblk, sh);
    const double rr_new = total(P.prr + (long)parity * B * P.nblk + pb, P.nblk, sh);
    const double rr_old = k > 0 ? total(P.prr + (long)(parity ^ 1) * B * P.nblk + pb, P.nblk, sh) : 0.0;
    const bool frozen = !(rr_new > (double)P.tol2 * rr0) || !(rr_old > 0.0);
    const float beta = frozen ? 0.f : (float)(rr_new / rr_old);
    const float* pprev = (parity ? P.p0 : P.p1) + base;
    float* pcur = (parity ? P.p1 : P.p0) + base;
    const float* r = P.r + base;
    double acc = 0.0;
    for (int kk = 0; kk < 4; ++kk) {
        const int i = blockIdx.x * PPB + kk * 256 + threadIdx.x;
        if (i >= HW || !P.hole[fb + i]) continue;
        const unsigned code = P.ecode[fb + i];
        const int off[4] = {1, P.W, -1, -P.W};
        const float pc = r[i] + beta * pprev[i];
        float diag = 0.f, v = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const unsigned e = (code >> (2 * n)) & 3u;
            if (e == 2) { diag += 2.f; v -= 2.f * (r[i + off[n]] + beta * pprev[i + off[n]]); }
            else if (e == 1) diag += 1.f;
        }
        v += diag * pc;
        pcur[i] = pc;
        P.q[base + i] = v;
        acc += (double)v * pc;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) P.ppq[pb + blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) blend_update(const BlendP P, int k) {
    __shared__ double sh[4];
    const int b = blockIdx.y, f = b / 3;
    const long HW = (long)P.H * P.W, base = (long)b * HW, fb = (long)f * HW;
    const long pb = (long)b * P.nblk, B = (long)P.N * 3;
    const int parity = k & 1;
    const double rr0 = total(P.prr0 + pb, P.nblk, sh);
    const double rr = total(P.prr + (long)parity * B * P.nblk + pb, P.nblk, sh);
    const double pq = total(P.ppq + pb, P.nblk, sh);
    const bool frozen = !(rr > (double)P.tol2 * rr0) || !(pq > 0.0);
    const float alpha = frozen ? 0.f : (float)(rr / pq);
    const float* pcur = (parity ? P.p1 : P.p0) + base;
    double acc = 0.0;
    for (int kk = 0; kk < 4; ++kk) {
        const int i = blockIdx.x * PPB + kk * 256 + threadIdx.x;
        if (i >= HW || !P.hole[fb + i]) continue;
        const float rn = P.r[base + i] - alpha * P.q[base + i];
        P.x[base + i] += alpha * pcur[i];
        P.r[base + i] = rn;
        acc += (double)rn * rn;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) P.prr[(long)(parity ^ 1) * B * P.nblk + pb + blockIdx.x] = s;
}

// planar solution -> channels-last blend: imgBlend = hole * recon + (1 - hole) * target (:47-48)
__global__ void __launch_bounds__(256) blend_finish(const BlendP P, float* out) {
    const long HW = (long)P.H * P.W, total_px = (long)P.N * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_px; i += (long)gridDim.x * blockDim.x) {
        const long f = i / HW, pix = i - f * HW;
        const bool h = P.hole[i] != 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[i * 3 + c] = h ? P.x[(f * 3 + c) * HW + pix] : P.trg[i * 3 + c];
    }
}

// UnfilledMask (:143-168).  Cleared maps C (top-left sweep) and C2 (bottom-right sweep):
//   C [i][j] = !hole | (C[i-1][j] & gm[i-1][j] == 0) | (C[i][j-1] & gm[i][j-1] == 0)
//   C2[i][j] = !hole | (gm[i][j] == 0 & (C2[i+1][j] | C2[i][j+1]))
// unfilled = hole & !C & !C2.  Row recurrence c_j = a_j | (b_j & c_{j-1}): inclusive scan of the maps (a, b) under composition.
constexpr int SCAN_T = 1024;
__device__ __forceinline__ void row_scan(bool a, bool b, bool carry_in, int j, int n, unsigned char* sa, unsigned char* sb, bool& out) {
    // Hillis-Steele over n <= SCAN_T columns held one per thread (thread j); element = (a, b); (a2,b2) o (a1,b1) = (a2 | (b2 & a1), b2 & b1)
    sa[j] = a; sb[j] = b;
    __syncthreads();
    for (int d = 1; d < n; d <<= 1) {
        bool na = sa[j], nb = sb[j];
        if (j >= d && j < n) { na = sa[j] | (sb[j] & sa[j - d]); nb = sb[j] & sb[j - d]; }
        __syncthreads();
        sa[j] = na; sb[j] = nb;
        __syncthreads();
    }
    out = sa[j] | (sb[j] & carry_in);
}

__global__ void __launch_bounds__(SCAN_T) unfilled_kernel(const unsigned char* hole, const unsigned char* gmask, int H, int W, unsigned char* work /* [N,H,W] */,
                                                          unsigned char* unfilled) {
    __shared__ unsigned char sa[SCAN_T], sb[SCAN_T];
    __shared__ unsigned char carry;
    const long HW = (long)H * W, fb = (long)blockIdx.x * HW;
    const unsigned char* ho = hole + fb;
    const unsigned char* gm = gmask + fb;
    unsigned char* C = work + fb;
    const int j = threadIdx.x;
    // ---- top-left sweep: rows ascending, columns ascending (chunks of SCAN_T columns with a carry)
    for (int i = 0; i < H; ++i) {
        if (j == 0) carry = 0;
        __syncthreads();
        for (int c0 = 0; c0 < W; c0 += SCAN_T) {
            const int x = c0 + j, n = min(SCAN_T, W - c0);
            bool a = false, b = false;
            if (x < W) {
                a = !ho[(long)i * W + x] || (i > 0 && C[(long)(i - 1) * W + x] && gm[(long)(i - 1) * W + x] == 0);
                b = x > 0 && gm[(long)i * W + x - 1] == 0;
            }
            bool out;
            const bool cin = carry != 0;
            __syncthreads();
            row_scan(a, b, cin, j, n, sa, sb, out);
            if (x < W) C[(long)i * W + x] = out;
            __syncthreads();
            if (j == n - 1) carry = out;
            __syncthreads();
        }
    }
    // unfilled_tl = hole & !C -> keep in `unfilled`, then reuse C for the bottom-right sweep
    for (long p = j; p < HW; p += SCAN_T) unfilled[fb + p] = ho[p] && !C[p];
    __syncthreads();
    for (int i = H - 1; i >= 0; --i) {
        if (j == 0) carry = 0;
        __syncthreads();
        for (int c1 = W; c1 > 0; c1 -= SCAN_T) {                       // chunks from the right; thread j handles column c1 - 1 - j
            const int n = min(SCAN_T, c1), x = c1 - 1 - j;
            bool a = false, b = false;
            if (j < n) {
                const bool g0 = gm[(long)i * W + x] == 0;
                a = !ho[(long)i * W + x] || (g0 && i + 1 < H && C[(long)(i + 1) * W + x]);
                b = g0 && x + 1 < W;
            }
            bool out;
            const bool cin = carry != 0;
            __syncthreads();
            row_scan(a, b, cin, j, n, sa, sb, out);
            if (j < n) C[(long)i * W + x] = out;
            __syncthreads();
            if (j == n - 1) carry = out;
            __syncthreads();
        }
    }
    for (long p = j; p < HW; p += SCAN_T) unfilled[fb + p] = unfilled[fb + p] && !C[p];
}

inline long al256(long b) { return (b + 255) / 256 * 256; }

}  // namespace

// solve_onchip.hip: the CG iterations of all N * 3 problems inside ONE launch (one workgroup per problem)
int fgt_blend_onchip(const float* trg, const float* gx, const float* gy, const unsigned char* hole, const unsigned char* ecode, const int* bbox,
                     float* x, int* status, int N, int H, int W, int max_rows, int max_cols, int iters, float tol, hipStream_t s);

extern "C" long fgt_poisson_blend_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const long n = (long)N * H * W, nblk = ((long)H * W + PPB - 1) / PPB;
    return al256(4l * N * 3 * nblk * 8) + 5 * al256(3 * n * 4) + 2 * al256(n);
}

static int poisson_blend_impl(const float* target, const float* gx, const float* gy, const unsigned char* hole, const unsigned char* gmask,
                              int N, int H, int W, int iters, float tol, float* blend, unsigned char* unfilled, void* workspace, void* stream,
                              const int* bbox, int max_rows, int max_cols, int* status) {
    FGT_REQUIRE(target && gx && gy && hole && gmask && blend && unfilled && workspace, "fgt_poisson_blend: null pointer");
    FGT_REQUIRE(N > 0 && H > 1 && W > 1 && iters >= 0 && tol >= 0.f, "fgt_poisson_blend: bad sizes");
    FGT_REQUIRE((long)H * W < (1l << 30) && (long)N * 3 <= 65535, "fgt_poisson_blend: clip too large for one call (N * 3 problems on grid.y)");
    FGT_REQUIRE(((uintptr_t)workspace & 7) == 0, "fgt_poisson_blend: workspace must be 8-byte aligned");
    BlendP P;
    P.trg = target; P.gx = gx; P.gy = gy; P.hole = hole; P.gmask = gmask; P.N = N; P.H = H; P.W = W;
    P.nblk = (int)(((long)H * W + PPB - 1) / PPB);
    P.tol2 = tol * tol;
    const long n = (long)N * H * W, np = (long)N * 3 * P.nblk;
    char* w = static_cast<char*>(workspace);
    double* d = reinterpret_cast<double*>(w); w += al256(4 * np * 8);
    P.prr0 = d; P.prr = d + np; P.ppq = d + 3 * np;
    float* fl[5];
    for (int i = 0; i < 5; ++i) { fl[i] = reinterpret_cast<float*>(w); w += al256(3 * n * 4); }
    P.x = fl[0]; P.r = fl[1]; P.p0 = fl[2]; P.p1 = fl[3]; P.q = fl[4];
    P.ecode = reinterpret_cast<unsigned char*>(w); w += al256(n);
    unsigned char* scan_work = reinterpret_cast<unsigned char*>(w);
    hipStream_t s = (hipStream_t)stream;
    const int gpx = cdiv(n, 256) > 16384 ? 16384 : cdiv(n, 256);
    hipLaunchKernelGGL(blend_codes, dim3(gpx), dim3(256), 0, s, P);
    dim3 grid(P.nblk, N * 3), block(256);
    hipLaunchKernelGGL(blend_init, grid, block, 0, s, P);
    if (bbox) {
        // every problem on chip, all iterations in one launch (the known pixels of P.x carry the target: blend_init wrote them)
        if (int rc = fgt_blend_onchip(target, gx, gy, hole, P.ecode, bbox, P.x, status, N, H, W, max_rows, max_cols, iters, tol, s)) return rc;
    } else {
        for (int k = 0; k < iters; ++k) {
            hipLaunchKernelGGL(blend_apply, grid, block, 0, s, P, k);
            hipLaunchKernelGGL(blend_update, grid, block, 0, s, P, k);
        }
    }
    hipLaunchKernelGGL(blend_finish, dim3(gpx), dim3(256), 0, s, P, blend);
    hipLaunchKernelGGL(unfilled_kernel, dim3(N), dim3(SCAN_T), 0, s, hole, gmask, H, W, scan_work, unfilled);
    return fgt_check_launch("poisson_blend");
}

extern "C" int fgt_poisson_blend(const float* target, const float* gx, const float* gy, const unsigned char* hole, const unsigned char* gmask,
                                 int N, int H, int W, int iters, float tol, float* blend, unsigned char* unfilled, void* workspace, void* stream) {
    return poisson_blend_impl(target, gx, gy, hole, gmask, N, H, W, iters, tol, blend, unfilled, workspace, stream, nullptr, 0, 0, nullptr);
}

extern "C" int fgt_poisson_blend_onchip(const float* target, const float* gx, const float* gy, const unsigned char* hole, const unsigned char* gmask,
                                        const int* bbox, int N, int H, int W, int max_rows, int max_cols, int iters, float tol, float* blend,
                                        unsigned char* unfilled, int* status, void* workspace, void* stream) {
    FGT_REQUIRE(bbox && status && max_rows >= 0 && max_cols >= 0, "fgt_poisson_blend_onchip: bbox / status / bounds");
    return poisson_blend_impl(target, gx, gy, hole, gmask, N, H, W, iters, tol, blend, unfilled, workspace, stream, bbox, max_rows, max_cols, status);
}
