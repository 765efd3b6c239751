// Laplace ("diffusion") fill of the masked region of a batch of H x W scalar maps on gfx950.
//
// Reference: tool/utils/region_fill.py:7-63 (regionfill at factor 1) as called per flow channel by diffusion(),
// tool/video_inpainting.py:42-51: for every masked pixel p
//     n(p) * x(p) - sum_{q in N4(p), q masked} x(q) = sum_{q in N4(p), q in the image, q not masked} I(q),
// n(p) = number of 4-neighbours inside the image (4 / 3 / 2), unmasked pixels keep I.  The reference assembles the sparse
// matrix and calls scipy's direct solver in float64 once per map (158 flows x 2 channels per 80-frame clip, ~17 k unknowns each).
//
// Here all maps are solved together by conjugate gradients on the masked 5-point stencil (the matrix is symmetric positive
// definite: a graph Laplacian plus the Dirichlet diagonal).  HBM-bound and tiny per iteration, so the design goals are no host
// synchronisation and bit-reproducibility:
//   * one problem per blockIdx.y, 1024 pixels per workgroup, threads at unmasked pixels exit (traffic ~ hole area);
//   * dot products are two-stage and ordered: every workgroup writes a double partial, every consumer workgroup re-reduces the
//     (<= a few hundred) partials of its problem in a fixed order: no atomics, no mutable scalars, so every launch is a pure function
//     of the previous launches' outputs and results are bit-identical run to run;
//   * a problem whose residual has dropped below tol^2 * |r0|^2 freezes itself (alpha = beta = 0) - the iteration count is fixed, so
//     the call never reads anything back.
// Per iteration two launches: K_apply (p_k = r + beta p_{k-1} formed on the fly at the pixel and its four neighbours from the
// previous direction buffer, written to the other one; q = A p_k; partial p.q) and K_update (x += alpha p, r -= alpha q,
// partial r.r).
#include "common.h"

namespace {

constexpr int PPB = 1024;   // pixels per workgroup (256 threads x 4)

struct FillP {
    const float* I;            // [B, H, W]
    const unsigned char* mask; // [n_masks, H, W], problem b uses mask b % n_masks
    float *x, *r, *p0, *p1, *q; // [B, H, W]; the search direction is double buffered (p_k is built from p_{k-1} by its readers)
    double *prr0, *prr, *ppq;  // partials: prr0 [B, nblk], prr [2, B, nblk], ppq [B, nblk]
    int B, H, W, n_masks, nblk;
    float tol2;
};

__device__ __forceinline__ double block_sum(double v, double* sh) {
    // fixed-order tree: lanes by xor shuffle, then the 4 wavefronts in order
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// sum of the nblk partials of problem b (every workgroup computes the same value in the same order)
__device__ __forceinline__ double total(const double* part, int nblk, double* sh) {
    double v = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) v += part[i];
    return block_sum(v, sh);
}

__global__ void __launch_bounds__(256) fill_init(const FillP P) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const long base = (long)b * P.H * P.W;
    const unsigned char* m = P.mask + (long)(b % P.n_masks) * P.H * P.W;
    double acc = 0.0;
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.x * PPB + k * 256 + threadIdx.x;
        if (i >= P.H * P.W) continue;
        const float v = P.I[base + i];
        if (!m[i]) { P.x[base + i] = v; continue; }
        const int y = i / P.W, x = i - y * P.W;
        float rhs = 0.f;                                   // region_fill.py:66-101 (formRightSide)
        if (y > 0 && !m[i - P.W]) rhs += P.I[base + i - P.W];
        if (y < P.H - 1 && !m[i + P.W]) rhs += P.I[base + i + P.W];
        if (x > 0 && !m[i - 1]) rhs += P.I[base + i - 1];
        if (x < P.W - 1 && !m[i + 1]) rhs += P.I[base + i + 1];
        P.x[base + i] = 0.f;                               // x0 = 0 inside the hole: r0 = rhs
        P.r[base + i] = rhs;
        P.p1[base + i] = 0.f;                              // "p_{-1}": p_0 = r_0 + 0 * p_{-1}
        acc += (double)rhs * rhs;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) {
        P.prr0[(long)b * P.nblk + blockIdx.x] = s;
        P.prr[(long)b * P.nblk + blockIdx.x] = s;          // parity 0
    }
}

__global__ void __launch_bounds__(256) fill_apply(const FillP P, int k) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const long base = (long)b * P.H * P.W;
    const long pb = (long)b * P.nblk;
    const int parity = k & 1;
    // beta_k = |r_k|^2 / |r_{k-1}|^2 (0 for k = 0 and for a frozen problem): recomputed by every workgroup from the ordered partials
    const double rr0 = total(P.prr0 + pb, P.nblk, sh);
    const double rr_new = total(P.prr + (long)parity * P.B * P.nblk + pb, P.nblk, sh);
    const double rr_old = k > 0 ? total(P.prr + (long)(parity ^ 1) * P.B * P.nblk + pb, P.nblk, sh) : 0.0;
    const bool frozen = !(rr_new > (double)P.tol2 * rr0) || !(rr_old > 0.0);
    const float beta = frozen ? 0.f : (float)(rr_new / rr_old);
    const float* pprev = (parity ? P.p0 : P.p1) + base;     // p_{k-1}
    float* pcur = (parity ? P.p1 : P.p0) + base;            // p_k
    const float* r = P.r + base;
    const unsigned char* m = P.mask + (long)(b % P.n_masks) * P.H * P.W;
    double acc = 0.0;
    for (int kk = 0; kk < 4; ++kk) {
        const int i = blockIdx.x * PPB + kk * 256 + threadIdx.x;
        if (i >= P.H * P.W || !m[i]) continue;
        const int y = i / P.W, x = i - y * P.W;
        const float pc = r[i] + beta * pprev[i];
        const int nn = (y > 0) + (y < P.H - 1) + (x > 0) + (x < P.W - 1);      // region_fill.py:104-117
        float v = (float)nn * pc;
        if (y > 0 && m[i - P.W]) v -= r[i - P.W] + beta * pprev[i - P.W];
        if (y < P.H - 1 && m[i + P.W]) v -= r[i + P.W] + beta * pprev[i + P.W];
        if (x > 0 && m[i - 1]) v -= r[i - 1] + beta * pprev[i - 1];
        if (x < P.W - 1 && m[i + 1]) v -= r[i + 1] + beta * pprev[i + 1];
        pcur[i] = pc;
        P.q[base + i] = v;
        acc += (double)v * pc;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) P.ppq[pb + blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) fill_update(const FillP P, int k) {
    __shared__ double sh[4];
    const int b = blockIdx.y;
    const long base = (long)b * P.H * P.W;
    const long pb = (long)b * P.nblk;
    const int parity = k & 1;
    const double rr0 = total(P.prr0 + pb, P.nblk, sh);
    const double rr = total(P.prr + (long)parity * P.B * P.nblk + pb, P.nblk, sh);
    const double pq = total(P.ppq + pb, P.nblk, sh);
    const bool frozen = !(rr > (double)P.tol2 * rr0) || !(pq > 0.0);
    const float alpha = frozen ? 0.f : (float)(rr / pq);
    const float* pcur = (parity ? P.p1 : P.p0) + base;
    const unsigned char* m = P.mask + (long)(b % P.n_masks) * P.H * P.W;
    double acc = 0.0;
    for (int kk = 0; kk < 4; ++kk) {
        const int i = blockIdx.x * PPB + kk * 256 + threadIdx.x;
        if (i >= P.H * P.W || !m[i]) continue;
        const float rn = P.r[base + i] - alpha * P.q[base + i];
        P.x[base + i] += alpha * pcur[i];
        P.r[base + i] = rn;
        acc += (double)rn * rn;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) P.prr[(long)(parity ^ 1) * P.B * P.nblk + pb + blockIdx.x] = s;
}

}  // namespace

extern "C" long fgt_laplace_fill_workspace(int B, int H, int W) {
    const long nblk = ((long)H * W + PPB - 1) / PPB;
    return 4l * B * H * W * sizeof(float) + 4l * B * nblk * sizeof(double);
}

extern "C" int fgt_laplace_fill(const float* I, const unsigned char* mask, int B, int n_masks, int H, int W, float* out,
                                void* workspace, int iters, float tol, void* stream) {
    FGT_REQUIRE(I && mask && out && workspace, "fgt_laplace_fill: null pointer");
    FGT_REQUIRE(B > 0 && n_masks > 0 && H > 0 && W > 0 && iters >= 0 && tol >= 0.f, "fgt_laplace_fill: bad sizes");
    FGT_REQUIRE((long)H * W < (1l << 30), "fgt_laplace_fill: map too large");
    FGT_REQUIRE(((uintptr_t)workspace & 7) == 0, "fgt_laplace_fill: workspace must be 8-byte aligned");
    FillP P;
    P.I = I; P.mask = mask; P.B = B; P.H = H; P.W = W; P.n_masks = n_masks;
    P.nblk = (int)(((long)H * W + PPB - 1) / PPB);
    P.tol2 = tol * tol;
    const long n = (long)B * H * W, np = (long)B * P.nblk;
    double* d = static_cast<double*>(workspace);
    P.prr0 = d; P.prr = d + np; P.ppq = d + 3 * np;
    float* f = reinterpret_cast<float*>(d + 4 * np);
    P.x = out; P.r = f; P.p0 = f + n; P.p1 = f + 2 * n; P.q = f + 3 * n;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(P.nblk, B), block(256);
    hipLaunchKernelGGL(fill_init, grid, block, 0, s, P);
    for (int k = 0; k < iters; ++k) {
        hipLaunchKernelGGL(fill_apply, grid, block, 0, s, P, k);
        hipLaunchKernelGGL(fill_update, grid, block, 0, s, P, k);
    }
    return fgt_check_launch("laplace_fill");
}
