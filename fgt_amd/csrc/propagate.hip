// Flow-guided gradient propagation on the GPU: tool/get_flowNN_gradient.py:11-534 (Nonlocal = False) with
// tool/utils/common_utils.py:149-254 (interp, BFconsistCheck / FBconsistCheck, consistCheck) folded in.
//
// The reference walks the clip frame by frame on the CPU, per hole pixel: follow the completed backward (forward) flow to the
// previous (next) frame, check that the opposite flow brings the pixel back (round-trip error < consistencyThres), and record the
// "flow neighbour" — the landing position if it is a known pixel, or the landing pixel's own neighbour (plus the sub-pixel
// remainder) if that pixel is a hole that already has one.  Then every hole pixel takes the gradient found at its neighbour
// (bilinear), visiting frames in temporal order so chained pixels read propagated values, and the backward- and forward-pass
// candidates are fused with weights exp(-round-trip error / alpha).
//
// GPU shape: the dependence is frame -> frame only, so a sweep is one launch per frame with a thread per pixel; ~320 small launches
// per clip in total, all enqueued by ONE C-ABI call (no host round trip).  It is latency-bound index / gather work: no LDS, no MFMA.
// Arithmetic widths follow numpy's promotions in the reference (float32 positions, float64 neighbour chains / weights) and the
// products and sums are kept un-fused (see the note on FP contraction below): the reference's numpy has no FMAs.
// The one primitive that is NOT numpy in the reference, cv2.remap(INTER_LINEAR), follows the specification in oracle/prop_oracle.py
// (`remap_bilinear`: 1/32-pixel coordinate table, fp32 weights, zero border; tab = 0: plain float bilinear).
#include "common.h"

// hipcc contracts a*b + c into an FMA by default (-ffp-contract=fast); numpy never fuses.  HIP's __fmul_rn / __fadd_rn do not help: on
// this toolchain they are header-inline plain operators compiled WITH the contract flag, so after inlining the pair is fused anyway
// (seen in the ISA as v_pk_fma_f32 in the bilinear sum, and on the GPU as 1-ulp differences in the round-trip error of 0.5 % of the
// pixels, which the exp(-error / alpha) weights amplified).  The products / sums below are therefore this file's own functions,
// defined under `fp contract(off)`.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float mulf(float a, float b) { return a * b; }
__device__ __forceinline__ float addf(float a, float b) { return a + b; }
__device__ __forceinline__ double muld(double a, double b) { return a * b; }
__device__ __forceinline__ double addd(double a, double b) { return a + b; }

struct PropP {
    const unsigned char* mask;     // [N, HW]
    const float *flow_f, *flow_b;  // [N-1, HW, 2] (u, v)
    int N, H, W, tab;
    double thres, alpha;
    // workspace
    double *nn_y, *nn_x;           // [2][N*HW]
    int* nn_t;                     // [2][N*HW]   source frame, -1 = no flow neighbour
    float* cuv;                    // [2][N*HW][2]
    float *cand[2][2];             // [pass][x|y] gradients [N, HW, 3]
};

struct Tap { int ix, iy; float w00, w01, w10, w11; };

// oracle/prop_oracle.py remap_bilinear: coordinates -> integer cell + fp32 weights
__device__ __forceinline__ Tap make_tap(float x, float y, int tab) {
    Tap t;
    float fx, fy;
    if (tab) {
        const long sx = (long)rint((double)x * tab), sy = (long)rint((double)y * tab);     // cvRound: half to even
        const long ix = tab == 32 ? (sx >> 5) : (sx >= 0 ? sx / tab : -((-sx + tab - 1) / tab));
        const long iy = tab == 32 ? (sy >> 5) : (sy >= 0 ? sy / tab : -((-sy + tab - 1) / tab));
        fx = (float)(sx - ix * tab) / (float)tab;
        fy = (float)(sy - iy * tab) / (float)tab;
        t.ix = (int)max(min(ix, 1l << 30), -(1l << 30));
        t.iy = (int)max(min(iy, 1l << 30), -(1l << 30));
    } else {
        const float flx = floorf(x), fly = floorf(y);
        fx = x - flx;
        fy = y - fly;
        t.ix = (int)fmaxf(fminf(flx, 1e9f), -1e9f);
        t.iy = (int)fmaxf(fminf(fly, 1e9f), -1e9f);
    }
    const float gx = 1.f - fx, gy = 1.f - fy;
    t.w00 = mulf(gx, gy); t.w01 = mulf(fx, gy); t.w10 = mulf(gx, fy); t.w11 = mulf(fx, fy);
    return t;
}

// value = ((v00*w00 + v01*w01) + v10*w10) + v11*w11, taps outside the image are 0; img has `nc` interleaved channels, channel c
__device__ __forceinline__ float sample(const float* img, int H, int W, int nc, int c, const Tap& t) {
    auto at = [&](int yy, int xx) { return ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? img[((long)yy * W + xx) * nc + c] : 0.f; };
    const float a = mulf(at(t.iy, t.ix), t.w00), b = mulf(at(t.iy, t.ix + 1), t.w01);
    const float c2 = mulf(at(t.iy + 1, t.ix), t.w10), d = mulf(at(t.iy + 1, t.ix + 1), t.w11);
    return addf(addf(addf(a, b), c2), d);
}

// One frame of a sweep (get_flowNN_gradient.py:72-243 with dir = +1 / :245-364 with dir = -1).  k = 0: backward-flow neighbours
// (forward sweep), k = 1: forward-flow neighbours (backward sweep).
__global__ void __launch_bounds__(256) prop_pass_kernel(const PropP p, int t, int dir) {
    const long HW = (long)p.H * p.W;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    const int k = dir > 0 ? 0 : 1;
    const int s = dir > 0 ? t - 1 : t + 1, fi = dir > 0 ? t - 1 : t;
    const long base = (long)k * p.N * HW;
    const long o = base + (long)t * HW + pix;
    if (!p.mask[(long)t * HW + pix]) return;
    const int y = (int)(pix / p.W), x = (int)(pix - (long)y * p.W);
    const float* to = (dir > 0 ? p.flow_b : p.flow_f) + (long)fi * HW * 2;
    const float* back = (dir > 0 ? p.flow_f : p.flow_b) + (long)fi * HW * 2;
    const float ny = addf((float)y, to[pix * 2 + 1]), nx = addf((float)x, to[pix * 2]);       // :92-101
    const int iy = (int)rintf(ny), ix = (int)rintf(nx);                                                   // :104 np.round: half to even
    const Tap tp = make_tap(nx, ny, p.tab);
    const float by = addf(ny, sample(back, p.H, p.W, 2, 1, tp)), bx = addf(nx, sample(back, p.H, p.W, 2, 0, tp));
    const double dy = (double)by - y, dx = (double)bx - x;                                                // common_utils.py:201-204 (float64)
    const bool consist = sqrt(addd(muld(dy, dy), muld(dx, dx))) < p.thres;
    const float au = fabsf(addf(bx, -(float)x)), av = fabsf(addf(by, -(float)y));             // |consistCheck(...)[y, x]| (:118-119, fp32)
    const bool inb = iy >= 0 && iy < p.H - 1 && ix >= 0 && ix < p.W - 1;                                  // :123-127
    if (!inb || !consist) return;
    const long q = (long)s * HW + (long)iy * p.W + ix;
    if (!p.mask[q]) {                                                                                     // case 1 (:139-166)
        p.nn_y[o] = (double)ny; p.nn_x[o] = (double)nx; p.nn_t[o] = s;
        p.cuv[o * 2] = au; p.cuv[o * 2 + 1] = av;
        return;
    }
    const long oq = base + q;                                                                              // case 2 (:168-237)
    if (p.nn_t[oq] < 0) return;
    const double cy = p.nn_y[oq] + ((double)ny - iy), cx = p.nn_x[oq] + ((double)nx - ix);
    const double ry = rint(cy), rx = rint(cx);
    if (!(ry >= 0 && ry < p.H - 1 && rx >= 0 && rx < p.W - 1)) return;                                    // :196-200, :212
    p.nn_y[o] = cy; p.nn_x[o] = cx; p.nn_t[o] = p.nn_t[oq];
    p.cuv[o * 2] = fmaxf(au, fabsf(p.cuv[oq * 2])); p.cuv[o * 2 + 1] = fmaxf(av, fabsf(p.cuv[oq * 2 + 1]));
}

// get_flowNN_gradient.py:366-425 for the hole pixels of ONE target frame t (frames are finalised in sweep order, so the source frame
// of every pixel of frame t is already final): cand[k][x|y][t, pix, :] = bilinear sample of cand[k][x|y][source frame]
__global__ void __launch_bounds__(256) prop_gather_kernel(const PropP p, int t, int k) {
    const long HW = (long)p.H * p.W;
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    const long o = (long)k * p.N * HW + (long)t * HW + pix;
    const int s = p.nn_t[o];
    if (s < 0 || !p.mask[(long)t * HW + pix]) return;
    const Tap tp = make_tap((float)p.nn_x[o], (float)p.nn_y[o], p.tab);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        float* a = p.cand[k][g];
        const float* src = a + (long)s * HW * 3;
        float* dst = a + ((long)t * HW + pix) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[c] = sample(src, p.H, p.W, 3, c, tp);
    }
}

// :427-532: weights exp(-|round trip| / alpha) of the candidates a pixel has, normalised; fused gradient (float64 -> float32)
__global__ void __launch_bounds__(256) prop_fuse_kernel(const PropP p, const float* gx, const float* gy, float* out_gx, float* out_gy,
                                                        unsigned char* tofill) {
    const long HW = (long)p.H * p.W, total = (long)p.N * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const bool m = p.mask[i] != 0;
        double w[2];
        bool have[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const long o = (long)k * total + i;
            have[k] = m && p.nn_t[o] >= 0;
            const double cu = p.cuv[o * 2], cv = p.cuv[o * 2 + 1];
            w[k] = have[k] ? exp(-sqrt(addd(muld(cu, cu), muld(cv, cv))) / p.alpha) : 0.0;
        }
        const bool any = have[0] || have[1];
        tofill[i] = m && !any;
        if (!any) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { out_gx[i * 3 + c] = gx[i * 3 + c]; out_gy[i * 3 + c] = gy[i * 3 + c]; }
            continue;
        }
        const double den = w[0] + w[1];
        const double n = (double)((int)have[0] + (int)have[1]);
        const double w0 = den == 0.0 ? (have[0] ? 1.0 / n : 0.0) : w[0] / den, w1 = den == 0.0 ? (have[1] ? 1.0 / n : 0.0) : w[1] / den;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            out_gx[i * 3 + c] = (float)addd(muld((double)p.cand[0][0][i * 3 + c], w0), muld((double)p.cand[1][0][i * 3 + c], w1));
            out_gy[i * 3 + c] = (float)addd(muld((double)p.cand[0][1][i * 3 + c], w0), muld((double)p.cand[1][1][i * 3 + c], w1));
        }
    }
}

__global__ void __launch_bounds__(256) prop_init_kernel(int* nn_t, float* cuv, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        nn_t[i] = -1;
        cuv[i * 2] = 0.f;
        cuv[i * 2 + 1] = 0.f;
    }
}

inline long align256(long b) { return (b + 255) / 256 * 256; }

}  // namespace

extern "C" long fgt_flow_propagate_workspace(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const long n = (long)N * H * W;
    return 2 * align256(2 * n * 8) + align256(2 * n * 4) + align256(2 * n * 2 * 4) + 4 * align256(n * 3 * 4);
}

extern "C" int fgt_flow_propagate(const float* gx, const float* gy, const unsigned char* mask, const float* flow_f, const float* flow_b,
                                  int N, int H, int W, double consistency_thres, double alpha, int tab, float* out_gx, float* out_gy,
                                  unsigned char* mask_tofill, void* workspace, void* stream) {
    FGT_REQUIRE(gx && gy && mask && out_gx && out_gy && mask_tofill && workspace, "fgt_flow_propagate: null pointer");
    FGT_REQUIRE(N >= 1 && H >= 2 && W >= 2 && (N == 1 || (flow_f && flow_b)), "fgt_flow_propagate: bad sizes");
    FGT_REQUIRE(tab == 0 || (tab > 0 && tab <= 1024), "fgt_flow_propagate: tab must be 0 (float bilinear) or the coordinate table size (32)");
    FGT_REQUIRE(alpha > 0, "fgt_flow_propagate: alpha must be positive");
    FGT_REQUIRE(((uintptr_t)workspace & 7) == 0, "fgt_flow_propagate: workspace must be 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const long HW = (long)H * W, n = (long)N * HW;
    PropP p;
    p.mask = mask; p.flow_f = flow_f; p.flow_b = flow_b; p.N = N; p.H = H; p.W = W; p.tab = tab; p.thres = consistency_thres; p.alpha = alpha;
    char* w = static_cast<char*>(workspace);
    p.nn_y = reinterpret_cast<double*>(w); w += align256(2 * n * 8);
    p.nn_x = reinterpret_cast<double*>(w); w += align256(2 * n * 8);
    p.nn_t = reinterpret_cast<int*>(w); w += align256(2 * n * 4);
    p.cuv = reinterpret_cast<float*>(w); w += align256(2 * n * 2 * 4);
    for (int k = 0; k < 2; ++k)
        for (int g = 0; g < 2; ++g) {
            p.cand[k][g] = reinterpret_cast<float*>(w); w += align256(n * 3 * 4);
            if (hipMemcpyAsync(p.cand[k][g], g == 0 ? gx : gy, n * 3 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) {
                fgt_set_error("fgt_flow_propagate: copy of the gradients failed");
                return FGT_ELAUNCH;
            }
        }
    const int blocks = cdiv(HW, 256);
    hipLaunchKernelGGL(prop_init_kernel, dim3(cdiv(2 * n, 256) > 16384 ? 16384 : cdiv(2 * n, 256)), dim3(256), 0, s, p.nn_t, p.cuv, 2 * n);
    for (int t = 1; t < N; ++t) hipLaunchKernelGGL(prop_pass_kernel, dim3(blocks), dim3(256), 0, s, p, t, +1);       // forward sweep
    for (int t = N - 2; t >= 0; --t) hipLaunchKernelGGL(prop_pass_kernel, dim3(blocks), dim3(256), 0, s, p, t, -1);  // backward sweep
    for (int t = 0; t < N; ++t) hipLaunchKernelGGL(prop_gather_kernel, dim3(blocks), dim3(256), 0, s, p, t, 0);
    for (int t = N - 1; t >= 0; --t) hipLaunchKernelGGL(prop_gather_kernel, dim3(blocks), dim3(256), 0, s, p, t, 1);
    hipLaunchKernelGGL(prop_fuse_kernel, dim3(cdiv(n, 256) > 16384 ? 16384 : cdiv(n, 256)), dim3(256), 0, s, p, gx, gy, out_gx, out_gy, mask_tofill);
    return fgt_check_launch("flow_propagate");
}
