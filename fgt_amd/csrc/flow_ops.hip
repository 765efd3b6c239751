// Flow-side bandwidth kernels: bilinear warps (image_warp / bilinear_sampler), forward-backward
// consistency, RAFT correlation pyramid pooling + lookup, convex upsampling and instance norm.
// All gather/HBM-bound; channels-last so that a warp's 4 corner reads are contiguous channel runs.
#include "common.h"
#include "conv_params.h"      // FgtFastDiv

namespace {

inline int grid_for(long total, int block = 256) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// XCD-contiguous work order for the gather kernels whose neighbouring work items read the SAME source rows (warp, forward-backward check).
// Workgroup i is dispatched to XCD i % 8, each XCD has a private L2: with the plain order the three workgroups that touch an image row sit on
// three XCDs and each fetches it from the fabric — rocprofv3 counted 2.66 x the algorithmic bytes for the clip's warps (round 5).  Here XCD x walks
// a CONTIGUOUS eighth of every grid-stride pass, so a row's re-reads hit that XCD's L2.  (grids are multiples of 8: grid_for8)
__device__ __forceinline__ long xcd_first_item() {
    const int per = gridDim.x >> 3;
    return ((long)(blockIdx.x & 7) * per + (blockIdx.x >> 3)) * blockDim.x + threadIdx.x;
}
inline int grid_for8(long total, int block = 256) {
    long g = (total + block - 1) / block;
    g = (g + 7) / 8 * 8;
    return (int)(g < 8 ? 8 : (g > 16384 ? 16384 : g));
}

// Source pixel coordinate of output pixel (x, y), following the reference's arithmetic in fp32.
//  align_corners = 0 / relative flow : LAFC/models/utils/fbConsistencyCheck.py:15-25 then grid_sample's
//      unnormalise ((g + 1) * size - 1) / 2                      (linspace base grid is float64 -> float32)
//  align_corners = 1 / absolute coords: RAFT/utils/utils.py:60-65 then ((g + 1) / 2) * (size - 1)
__device__ __forceinline__ void sample_coord(float fx, float fy, int x, int y, int W, int H, int align_corners, int absolute,
                                             float& ix, float& iy) {
    if (!absolute) {
        const float bx = (float)(-1.0 + (double)x * (2.0 / (double)(W - 1)));
        const float by = (float)(-1.0 + (double)y * (2.0 / (double)(H - 1)));
        const float gx = bx + fx / (float)((W - 1.0) / 2.0);
        const float gy = by + fy / (float)((H - 1.0) / 2.0);
        if (align_corners) { ix = ((gx + 1.f) / 2.f) * (float)(W - 1); iy = ((gy + 1.f) / 2.f) * (float)(H - 1); }
        else { ix = ((gx + 1.f) * (float)W - 1.f) / 2.f; iy = ((gy + 1.f) * (float)H - 1.f) / 2.f; }
    } else {
        const float gx = 2.f * fx / (float)(W - 1) - 1.f;
        const float gy = 2.f * fy / (float)(H - 1) - 1.f;
        if (align_corners) { ix = ((gx + 1.f) / 2.f) * (float)(W - 1); iy = ((gy + 1.f) / 2.f) * (float)(H - 1); }
        else { ix = ((gx + 1.f) * (float)W - 1.f) / 2.f; iy = ((gy + 1.f) * (float)H - 1.f) / 2.f; }
    }
}

// The launch-invariant parts of the relative-flow branch above, evaluated ONCE on the host in the same IEEE arithmetic (two fp64 divisions and two fp64 -> fp32
// conversions that every work item repeated), and the pixel decode as a multiply-shift: the warp kernels spent more instructions here than on their taps.
struct WarpK { double sx, sy; float hx, hy; FgtFastDiv dW; };
inline WarpK warp_consts(int W, int H) {
    WarpK k;
    k.sx = 2.0 / (double)(W - 1); k.sy = 2.0 / (double)(H - 1);
    k.hx = (float)((W - 1.0) / 2.0); k.hy = (float)((H - 1.0) / 2.0);
    k.dW = fgt_fastdiv_make((unsigned)W);
    return k;
}
__device__ __forceinline__ void sample_coord_k(float fx, float fy, int x, int y, int W, int H, int align_corners, int absolute, const WarpK& k,
                                               float& ix, float& iy) {
    if (absolute) { sample_coord(fx, fy, x, y, W, H, align_corners, 1, ix, iy); return; }
    const float bx = (float)(-1.0 + (double)x * k.sx);
    const float by = (float)(-1.0 + (double)y * k.sy);
    const float gx = bx + fx / k.hx;
    const float gy = by + fy / k.hy;
    if (align_corners) { ix = ((gx + 1.f) / 2.f) * (float)(W - 1); iy = ((gy + 1.f) / 2.f) * (float)(H - 1); }
    else { ix = ((gx + 1.f) * (float)W - 1.f) / 2.f; iy = ((gy + 1.f) * (float)H - 1.f) / 2.f; }
}
__device__ __forceinline__ int fdivw(int n, const FgtFastDiv f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.sh); }

struct Bilin { int x0, y0; float wnw, wne, wsw, wse; };
__device__ __forceinline__ Bilin bilin(float ix, float iy) {
    Bilin b;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    b.x0 = (int)fx0; b.y0 = (int)fy0;
    const float x1 = fx0 + 1.f, y1 = fy0 + 1.f;
    b.wnw = (x1 - ix) * (y1 - iy);
    b.wne = (ix - fx0) * (y1 - iy);
    b.wsw = (x1 - ix) * (iy - fy0);
    b.wse = (ix - fx0) * (iy - fy0);
    return b;
}

// One work item = V consecutive channels of one output pixel (V = 4 / 2: 16- / 8-byte loads and stores; V = 1: any channel count).  The
// four taps are loaded UNCONDITIONALLY from clamped addresses and zeroed by a select (grid_sample's zeros padding) so that all of a
// lane's loads are in flight together — the round-3 kernel walked the channels of a pixel with four predicated scalar loads each
// (0.12 of the HBM roof on 3-channel frames).  Same arithmetic, same order: v = ((0 + nw*wnw) + ne*wne) + sw*wsw) + se*wse.
template <int V, bool FLOW8 = true>
__global__ void __launch_bounds__(256) warp_kernel(const float* __restrict__ img, int ldi, const float* __restrict__ flow, int B, int H, int W, int C,
                                                   int align_corners, int absolute, float* __restrict__ out, int ldo, WarpK wk, FgtFastDiv dcpv) {
    typedef float vec __attribute__((ext_vector_type(V)));
    const int cpv = C / V;
    const int per_img = H * W * cpv;                     // grid: (items of one image / 256 — XCD-contiguous, blockIdx.y = image)
    const long b = blockIdx.y;
    for (int item = (int)xcd_first_item(); item < per_img; item += gridDim.x * blockDim.x) {
        const int pl = fdivw(item, dcpv);
        const int c = (item - pl * cpv) * V;
        const int y = fdivw(pl, wk.dW), x = pl - y * W;
        const long pix = b * H * W + pl;
        float2 fl;                                       // (FLOW8 = false: a flow pointer at an odd float offset — two 4-byte loads)
        if constexpr (FLOW8) fl = *reinterpret_cast<const float2*>(flow + pix * 2);
        else fl = make_float2(flow[pix * 2], flow[pix * 2 + 1]);
        float ix, iy;
        sample_coord_k(fl.x, fl.y, x, y, W, H, align_corners, absolute, wk, ix, iy);
        const Bilin bl = bilin(ix, iy);
        const bool vx0 = bl.x0 >= 0 && bl.x0 < W, vx1 = bl.x0 + 1 >= 0 && bl.x0 + 1 < W;
        const bool vy0 = bl.y0 >= 0 && bl.y0 < H, vy1 = bl.y0 + 1 >= 0 && bl.y0 + 1 < H;
        const int cx0 = min(max(bl.x0, 0), W - 1), cx1 = min(max(bl.x0 + 1, 0), W - 1);
        const int cy0 = min(max(bl.y0, 0), H - 1), cy1 = min(max(bl.y0 + 1, 0), H - 1);
        const float* base = img + b * H * W * ldi + c;
        const vec nw = *reinterpret_cast<const vec*>(base + ((long)cy0 * W + cx0) * ldi);
        const vec ne = *reinterpret_cast<const vec*>(base + ((long)cy0 * W + cx1) * ldi);
        const vec sw = *reinterpret_cast<const vec*>(base + ((long)cy1 * W + cx0) * ldi);
        const vec se = *reinterpret_cast<const vec*>(base + ((long)cy1 * W + cx1) * ldi);
        vec v;
#pragma unroll
        for (int u = 0; u < V; ++u) {
            // (explicit fused multiply-adds: every access-width variant of this kernel rounds identically)
            float a = __builtin_fmaf((vx0 && vy0) ? nw[u] : 0.f, bl.wnw, 0.f);
            a = __builtin_fmaf((vx1 && vy0) ? ne[u] : 0.f, bl.wne, a);
            a = __builtin_fmaf((vx0 && vy1) ? sw[u] : 0.f, bl.wsw, a);
            a = __builtin_fmaf((vx1 && vy1) ? se[u] : 0.f, bl.wse, a);
            v[u] = a;
        }
        __builtin_nontemporal_store(v, reinterpret_cast<vec*>(out + pix * ldo + c));
    }
}

// C = 2 (flow fields, the forward-backward check's operands): TWO adjacent pixels per work item — one 16-byte flow load, eight 8-byte tap
// loads in flight together, one 16-byte store.  Same arithmetic per pixel as warp_kernel<2>.
__global__ void __launch_bounds__(256) warp_c2x2_kernel(const float* __restrict__ img, const float* __restrict__ flow, int B, int H, int W,
                                                        int align_corners, int absolute, float* __restrict__ out, WarpK wk) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int per_img = H * W / 2;                       // (W even: a pixel pair never straddles two rows)
    const long b = blockIdx.y;
    for (int item = (int)xcd_first_item(); item < per_img; item += gridDim.x * blockDim.x) {
        const int pl = item * 2;
        const int y = fdivw(pl, wk.dW), x = pl - y * W;
        const long pix = b * H * W + pl;
        const f4 fl = __builtin_nontemporal_load(reinterpret_cast<const f4*>(flow + pix * 2));
        const float* base = img + b * H * W * 2;
        float2 tap[2][4];
        float wgt[2][4];
        bool val[2][4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float ix, iy;
            sample_coord_k(q ? fl.z : fl.x, q ? fl.w : fl.y, x + q, y, W, H, align_corners, absolute, wk, ix, iy);
            const Bilin bl = bilin(ix, iy);
            const bool vx0 = bl.x0 >= 0 && bl.x0 < W, vx1 = bl.x0 + 1 >= 0 && bl.x0 + 1 < W;
            const bool vy0 = bl.y0 >= 0 && bl.y0 < H, vy1 = bl.y0 + 1 >= 0 && bl.y0 + 1 < H;
            const int cx0 = min(max(bl.x0, 0), W - 1), cx1 = min(max(bl.x0 + 1, 0), W - 1);
            const int cy0 = min(max(bl.y0, 0), H - 1), cy1 = min(max(bl.y0 + 1, 0), H - 1);
            tap[q][0] = *reinterpret_cast<const float2*>(base + ((long)cy0 * W + cx0) * 2);
            tap[q][1] = *reinterpret_cast<const float2*>(base + ((long)cy0 * W + cx1) * 2);
            tap[q][2] = *reinterpret_cast<const float2*>(base + ((long)cy1 * W + cx0) * 2);
            tap[q][3] = *reinterpret_cast<const float2*>(base + ((long)cy1 * W + cx1) * 2);
            wgt[q][0] = bl.wnw; wgt[q][1] = bl.wne; wgt[q][2] = bl.wsw; wgt[q][3] = bl.wse;
            val[q][0] = vx0 && vy0; val[q][1] = vx1 && vy0; val[q][2] = vx0 && vy1; val[q][3] = vx1 && vy1;
        }
        float o[4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) a = __builtin_fmaf(val[q][k] ? (u ? tap[q][k].y : tap[q][k].x) : 0.f, wgt[q][k], a);
                o[q * 2 + u] = a;
            }
        __builtin_nontemporal_store(f4{o[0], o[1], o[2], o[3]}, reinterpret_cast<f4*>(out + pix * 2));
    }
}

__device__ __forceinline__ void warp2(const float* src, const float* flw, long b, int x, int y, int H, int W, const WarpK& wk, float& ox, float& oy) {
    const long pix = (b * H + y) * W + x;
    float ix, iy;
    sample_coord_k(flw[pix * 2], flw[pix * 2 + 1], x, y, W, H, 0, 0, wk, ix, iy);
    const Bilin bl = bilin(ix, iy);
    ox = oy = 0.f;
    const float* base = src + b * H * W * 2;
    const int xs[4] = {bl.x0, bl.x0 + 1, bl.x0, bl.x0 + 1}, ys[4] = {bl.y0, bl.y0, bl.y0 + 1, bl.y0 + 1};
    const float ws[4] = {bl.wnw, bl.wne, bl.wsw, bl.wse};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (xs[k] >= 0 && xs[k] < W && ys[k] >= 0 && ys[k] < H) {
            ox += base[((long)ys[k] * W + xs[k]) * 2] * ws[k];
            oy += base[((long)ys[k] * W + xs[k]) * 2 + 1] * ws[k];
        }
}

// fbConsistencyCheck.py:33-47
__global__ void __launch_bounds__(256) fb_kernel(const float* ffw, const float* fbw, int B, int H, int W, float a1, float a2,
                                                 float* occ_fw, float* occ_bw, WarpK wk) {
    const int per_img = H * W;
    const long b = blockIdx.y;
    for (int pl = (int)xcd_first_item(); pl < per_img; pl += gridDim.x * blockDim.x) {
        const int y = fdivw(pl, wk.dW), x = pl - y * W;
        const long pix = b * per_img + pl;
        float bwx, bwy, fwx, fwy;
        warp2(fbw, ffw, b, x, y, H, W, wk, bwx, bwy);  // flow_bw warped by flow_fw
        warp2(ffw, fbw, b, x, y, H, W, wk, fwx, fwy);  // flow_fw warped by flow_bw
        const float fx = ffw[pix * 2], fy = ffw[pix * 2 + 1], gx = fbw[pix * 2], gy = fbw[pix * 2 + 1];
        const float dfx = fx + bwx, dfy = fy + bwy, dbx = gx + fwx, dby = gy + fwy;
        const float mag_fw = (fx * fx + fy * fy) + (bwx * bwx + bwy * bwy);
        const float mag_bw = (gx * gx + gy * gy) + (fwx * fwx + fwy * fwy);
        occ_fw[pix] = (dfx * dfx + dfy * dfy) > (a1 * mag_fw + a2) ? 1.f : 0.f;
        occ_bw[pix] = (dbx * dbx + dby * dby) > (a1 * mag_bw + a2) ? 1.f : 0.f;
    }
}

__global__ void __launch_bounds__(256) avgpool2_kernel(const float* src, long rows, int H, int W, float* dst) {
    const int Ho = H / 2, Wo = W / 2;
    const long total = rows * Ho * Wo;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % Wo); const long r = idx / Wo;
        const int y = (int)(r % Ho); const long row = r / Ho;
        const float* s = src + row * H * W + (long)(2 * y) * W + 2 * x;
        dst[idx] = (((s[0] + s[1]) + s[W]) + s[W + 1]) * 0.25f;
    }
}

struct PyrPtrs { const float* p[4]; };

// RAFT/corr.py:29-50.  Output channel = lvl*(2r+1)^2 + a*(2r+1) + b samples level lvl at
// (x/2^lvl + (a - r), y/2^lvl + (b - r))  -- the reference adds meshgrid(dy, dx) to (x, y).
// One tap, exactly the reference's arithmetic (per-tap coordinate, normalise / un-normalise round trip, zeros outside).
__device__ __forceinline__ float corr_tap_global(const float* vol, int Hl, int Wl, const Bilin& bl) {
    float v = 0.f;
    const bool vx0 = bl.x0 >= 0 && bl.x0 < Wl, vx1 = bl.x0 + 1 >= 0 && bl.x0 + 1 < Wl;
    const bool vy0 = bl.y0 >= 0 && bl.y0 < Hl, vy1 = bl.y0 + 1 >= 0 && bl.y0 + 1 < Hl;
    if (vx0 && vy0) v += vol[(long)bl.y0 * Wl + bl.x0] * bl.wnw;
    if (vx1 && vy0) v += vol[(long)bl.y0 * Wl + bl.x0 + 1] * bl.wne;
    if (vx0 && vy1) v += vol[(long)(bl.y0 + 1) * Wl + bl.x0] * bl.wsw;
    if (vx1 && vy1) v += vol[(long)(bl.y0 + 1) * Wl + bl.x0 + 1] * bl.wse;
    return v;
}

// All (2r+1)^2 taps of a (pixel, level) read the same (2r+2)^2 integer-aligned window of that pixel's correlation map (the taps differ by
// whole pixels).  A block stages the windows of CL_PB pixels x 4 levels in LDS — 10 rows x 12 values each for r = 4: one margin COLUMN on
// either side because floor() of the per-tap coordinate may land one off floor(x) + (a - r) after the reference's normalise / un-normalise
// round trip (it costs no extra cache line), no margin rows (each row is its own line of a 5.4 GB volume: the kernel is bound by those
// fetches) — with row-contiguous loads, zeros outside the map; then every thread computes 4 consecutive output channels from LDS with the
// per-tap weights (a value * weight sum in the same order as corr_tap_global: adding 0 * w for an outside corner is exact, so the two paths
// are bit-identical; a tap whose corner falls outside the staged window takes the global path).  The one-thread-per-tap kernel this
// replaces issued 4 scattered 4-byte loads per output (adjacent lanes = adjacent map ROWS): 0.43 ms per call at 864x480 x 32 pairs.
constexpr int CL_PB = 8, CL_WIN = 12, CL_ROWS = 10, CL_PITCH = 13, CL_WSZ = CL_ROWS * CL_PITCH;
// RT = the radius as a compile-time constant (4: RAFT's), 0 = run-time radius: channel -> (level, a, b) are two integer divisions per tap, ~30 instructions
// each when the divisor is a kernel argument
template <int RT>
__global__ void __launch_bounds__(256) corr_lookup_kernel(PyrPtrs pyr, int levels, long npix, int H1, int W1, int radius_rt, const float* coords,
                                                          float* out, int ldo, __bf16* out_s, int ld_s, long ps, int nch_pad, int deep, int table) {
    __shared__ float win[CL_PB * 4 * CL_WSZ];
    __shared__ int wbase[CL_PB * 4][2];
    __shared__ float ctab[CL_PB * 4][2][CL_ROWS];       // un-normalised sample coordinates per (pixel, level): [x | y][tap index 0..2r]
    const int tid = threadIdx.x;
    const long q0 = (long)blockIdx.x * CL_PB;
    const int npx = (int)min((long)CL_PB, npix - q0);
    const int radius = RT > 0 ? RT : radius_rt;
    const int side = 2 * radius + 1, per_lvl = side * side, nch = levels * per_lvl;
    if (tid < npx * levels) {
        const int pl = tid / levels, lvl = tid - pl * levels;
        const float scale = (float)(1 << lvl);
        wbase[tid][0] = (int)floorf(coords[(q0 + pl) * 2] / scale) - radius - 1;
        wbase[tid][1] = (int)floorf(coords[(q0 + pl) * 2 + 1] / scale) - radius;
    }
    // The sample coordinate of tap (a, b) is separable: ix depends on (pixel, level, a), iy on (pixel, level, b) — 2 x 9 values per (pixel, level) where
    // the tap loop evaluated 2 x 81 (each: two fp32 divisions through the reference's normalise / un-normalise round trip; ~40 % of the kernel's time
    // was this arithmetic).  The same expressions, evaluated once: bit-identical taps (tests/test_flow_gpu.py: FGT_LOOKUP_TABLE=0 is the old loop).
    if (table) {
        for (int i = tid; i < npx * levels * side; i += 256) {
            const int w = i / side, k = i - w * side;
            const int pl = w / levels, lvl = w - pl * levels;
            const float scale = (float)(1 << lvl);
            float ix, iy;
            sample_coord(coords[(q0 + pl) * 2] / scale + (float)(k - radius), coords[(q0 + pl) * 2 + 1] / scale + (float)(k - radius), 0, 0, W1 >> lvl, H1 >> lvl, 1, 1, ix, iy);
            ctab[w][0][k] = ix;
            ctab[w][1][k] = iy;
        }
    }
    __syncthreads();
    auto stage1 = [&](int idx) -> float {
        const int w = idx / (CL_ROWS * CL_WIN), e = idx - w * (CL_ROWS * CL_WIN);
        const int j = e / CL_WIN, i = e - j * CL_WIN;
        const int pl = w / levels, lvl = w - pl * levels;
        const int Hl = H1 >> lvl, Wl = W1 >> lvl;
        const int gx = wbase[w][0] + i, gy = wbase[w][1] + j;
        float v = 0.f;
        if (gx >= 0 && gx < Wl && gy >= 0 && gy < Hl) v = pyr.p[lvl][(q0 + pl) * Hl * Wl + (long)gy * Wl + gx];
        return v;
    };
    auto put1 = [&](int idx, float v) {
        const int w = idx / (CL_ROWS * CL_WIN), e = idx - w * (CL_ROWS * CL_WIN);
        const int j = e / CL_WIN, i = e - j * CL_WIN;
        win[w * CL_WSZ + j * CL_PITCH + i] = v;
    };
    if (deep && npx == CL_PB && levels == 4) {
        // A full block (8 pixels x 4 levels): a thread takes GROUPS of 4 consecutive window values of one row — one row address per group instead of
        // one per value (the per-value index arithmetic of the rolled loop was as many instructions as the taps themselves) — and requests all of its
        // groups before using any: the kernel is latency-bound (269 MB of counter traffic in 0.43 ms), the rolled loop kept 2-4 loads per thread in flight.
        constexpr int GPR = CL_WIN / 4, NGRP = CL_PB * 4 * CL_ROWS * GPR, NGT = (NGRP + 255) / 256;      // 3 groups per row, 960 groups, 4 per thread
        float v[NGT][4];
#pragma unroll
        for (int l = 0; l < NGT; ++l) {
            const int gi = tid + l * 256;
#pragma unroll
            for (int u = 0; u < 4; ++u) v[l][u] = 0.f;
            if (gi < NGRP) {
                const int w = gi / (CL_ROWS * GPR), e = gi - w * (CL_ROWS * GPR);
                const int j = e / GPR, i0 = (e - j * GPR) * 4;
                const int pl = w >> 2, lvl = w & 3;
                const int Hl = H1 >> lvl, Wl = W1 >> lvl;
                const int gx0 = wbase[w][0] + i0, gy = wbase[w][1] + j;
                if (gy >= 0 && gy < Hl) {
                    const float* row = pyr.p[lvl] + (q0 + pl) * Hl * Wl + (long)gy * Wl;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (gx0 + u >= 0 && gx0 + u < Wl) v[l][u] = row[gx0 + u];
                }
            }
        }
#pragma unroll
        for (int l = 0; l < NGT; ++l) {
            const int gi = tid + l * 256;
            if (gi < NGRP) {
                const int w = gi / (CL_ROWS * GPR), e = gi - w * (CL_ROWS * GPR);
                const int j = e / GPR, i0 = (e - j * GPR) * 4;
#pragma unroll
                for (int u = 0; u < 4; ++u) win[w * CL_WSZ + j * CL_PITCH + i0 + u] = v[l][u];
            }
        }
    } else {
        for (int idx = tid; idx < npx * levels * CL_ROWS * CL_WIN; idx += 256) put1(idx, stage1(idx));
    }
    __syncthreads();
    const int quads = (out_s ? nch_pad : nch) / 4;         // 4 consecutive channels per thread (nch % 4 == 0)
    for (int idx = tid; idx < npx * quads; idx += 256) {
        const int pl = idx / quads, c0 = (idx - pl * quads) * 4;
        const long q = q0 + pl;
        const float X = coords[q * 2], Y = coords[q * 2 + 1];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ch = c0 + u;
            v[u] = 0.f;
            if (ch >= nch) continue;                          // zero padding of the split output
            const int lvl = ch / per_lvl, t = ch - lvl * per_lvl;
            const int a = t / side, b = t - a * side;
            const int Hl = H1 >> lvl, Wl = W1 >> lvl;
            const float scale = (float)(1 << lvl);
            const int w = pl * levels + lvl;
            float ix, iy;
            if (table) {
                ix = ctab[w][0][a];
                iy = ctab[w][1][b];
            } else {
                const float cx = X / scale + (float)(a - radius);
                const float cy = Y / scale + (float)(b - radius);
                sample_coord(cx, cy, 0, 0, Wl, Hl, 1, 1, ix, iy);
            }
            // (the bilinear factors stay per tap: tabulating them per axis as well measured no faster and lost bit-equality with the per-tap loop —
            //  hipcc contracts `ix - floor(ix)` with the multiplication that produced ix in one context and not in the other)
            const Bilin bl = bilin(ix, iy);
            const int rx = bl.x0 - wbase[w][0], ry = bl.y0 - wbase[w][1];
            if ((unsigned)rx < (unsigned)(CL_WIN - 1) && (unsigned)ry < (unsigned)(CL_ROWS - 1)) {
                const float* wv = win + w * CL_WSZ + ry * CL_PITCH + rx;
                float s = 0.f;
                s += wv[0] * bl.wnw;
                s += wv[1] * bl.wne;
                s += wv[CL_PITCH] * bl.wsw;
                s += wv[CL_PITCH + 1] * bl.wse;
                v[u] = s;
            } else {
                v[u] = corr_tap_global(pyr.p[lvl] + q * Hl * Wl, Hl, Wl, bl);
            }
        }
        if (out && c0 < nch) *reinterpret_cast<float4*>(out + q * ldo + c0) = make_float4(v[0], v[1], v[2], v[3]);
        if (out_s) {
            uint2 hi, lo;
            fgt_split4(make_float4(v[0], v[1], v[2], v[3]), hi, lo);
            *reinterpret_cast<uint2*>(out_s + q * ld_s + c0) = hi;
            *reinterpret_cast<uint2*>(out_s + q * ld_s + c0 + ps) = lo;
        }
    }
}

// RAFT/raft.py:73-84
__global__ void __launch_bounds__(256) convex_up_kernel(const float* flow, int ldf, const float* mask, int ldm, int B, int H,
                                                        int W, float* out) {
    const long total = (long)B * H * W * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int ij = (int)(idx & 63); const long pix = idx >> 6;
        const int i = ij >> 3, j = ij & 7;
        const int x = (int)(pix % W); const long r = pix / W;
        const int y = (int)(r % H); const long b = r / H;
        float mk[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) { mk[k] = mask[pix * ldm + k * 64 + ij]; mx = fmaxf(mx, mk[k]); }
        float den = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { mk[k] = expf(mk[k] - mx); den += mk[k]; }
        float ux = 0.f, uy = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const float* f = flow + ((b * H + yy) * W + xx) * ldf;
                const float wgt = mk[k] / den;
                ux += wgt * (8.f * f[0]);
                uy += wgt * (8.f * f[1]);
            }
        }
        const long HW8 = (long)(8 * H) * (8 * W);
        const long o = (long)(8 * y + i) * (8 * W) + 8 * x + j;
        out[(b * 2 + 0) * HW8 + o] = ux;
        out[(b * 2 + 1) * HW8 + o] = uy;
    }
}

}  // namespace

extern "C" int fgt_warp(const float* img, int ldi, const float* flow, int B, int H, int W, int C, int align_corners,
                        int absolute_coords, float* out, int ldo, void* stream) {
    FGT_REQUIRE(img && flow && out && B > 0 && B <= 65535 && H > 1 && W > 1 && C > 0 && (long)H * W * C < (1l << 31), "fgt_warp: bad arguments");
    const WarpK wk = warp_consts(W, H);
    FgtProfScope prof(FGT_PROF_WARP, 0.0, 4.0 * (double)B * H * W * (2.0 * C + 2.0), stream);
    const auto al = [](const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    FGT_REQUIRE(al(flow, 4) && al(img, 4) && al(out, 4), "fgt_warp: pointers must be 4-byte aligned");
    const bool flow8 = al(flow, 8);                    // (a contiguous view at an odd float offset is legal: scalar flow loads then)
    if (C == 2 && ldi == 2 && ldo == 2 && W % 2 == 0 && al(img, 8) && al(flow, 16) && al(out, 16)) {
        hipLaunchKernelGGL(warp_c2x2_kernel, dim3(grid_for8((long)H * W / 2), B), dim3(256), 0, (hipStream_t)stream, img, flow, B, H, W, align_corners,
                           absolute_coords, out, wk);
        return fgt_check_launch("warp");
    }
    const int V = (C % 4 == 0 && ldi % 4 == 0 && ldo % 4 == 0 && al(img, 16) && al(out, 16)) ? 4
                : (C % 2 == 0 && ldi % 2 == 0 && ldo % 2 == 0 && al(img, 8) && al(out, 8)) ? 2 : 1;
    const long items = (long)H * W * (C / V);            // per image (blockIdx.y)
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid_for8(items), B), dim3(256), 0, (hipStream_t)stream, img, ldi, flow, B, H, W, C, align_corners, absolute_coords, out, ldo,
                           wk, fgt_fastdiv_make((unsigned)(C / V)));
    };
    if (!flow8) { if (V == 4) go(warp_kernel<4, false>); else if (V == 2) go(warp_kernel<2, false>); else go(warp_kernel<1, false>); }
    else if (V == 4) go(warp_kernel<4>); else if (V == 2) go(warp_kernel<2>); else go(warp_kernel<1>);
    return fgt_check_launch("warp");
}

extern "C" int fgt_fb_consistency(const float* flow_fw, const float* flow_bw, int B, int H, int W, float alpha1, float alpha2,
                                  float* occ_fw, float* occ_bw, void* stream) {
    FGT_REQUIRE(flow_fw && flow_bw && occ_fw && occ_bw && H > 1 && W > 1 && B > 0 && B <= 65535, "fgt_fb_consistency: bad arguments");
    hipLaunchKernelGGL(fb_kernel, dim3(grid_for8((long)H * W), B), dim3(256), 0, (hipStream_t)stream, flow_fw, flow_bw, B, H, W,
                       alpha1, alpha2, occ_fw, occ_bw, warp_consts(W, H));
    return fgt_check_launch("fb_consistency");
}

extern "C" int fgt_avgpool2(const float* src, long rows, int H, int W, float* dst, void* stream) {
    FGT_REQUIRE(src && dst && rows > 0 && H >= 2 && W >= 2, "fgt_avgpool2: bad arguments");
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for(rows * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream, src, rows, H, W, dst);
    return fgt_check_launch("avgpool2");
}

extern "C" int fgt_corr_lookup_split(const float* const* pyr, int levels, int B, int H1, int W1, int radius, const float* coords,
                                     float* out, int ldo, void* out_s, int ld_s, long ps, int nch_pad, void* stream) {
    FGT_REQUIRE(pyr && coords && (out || out_s) && levels >= 1 && levels <= 4, "fgt_corr_lookup: bad arguments");
    const int side = 2 * radius + 1, nch = levels * side * side;
    FGT_REQUIRE(radius >= 1 && radius <= (CL_ROWS - 2) / 2, "fgt_corr_lookup: radius %d (the staged window holds radius <= %d)", radius, (CL_ROWS - 2) / 2);
    FGT_REQUIRE(nch % 4 == 0, "fgt_corr_lookup: levels * (2 radius + 1)^2 = %d is not a multiple of 4", nch);
    FGT_REQUIRE(!out || (ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0), "fgt_corr_lookup: out must be 16-byte aligned with a row stride that is a multiple of 4 floats");
    FGT_REQUIRE(!out_s || (nch_pad >= nch && nch_pad % 4 == 0 && ld_s >= nch_pad && ld_s % 4 == 0 && ps % 4 == 0 && (reinterpret_cast<uintptr_t>(out_s) & 7) == 0),
                "fgt_corr_lookup: split output: nch_pad %d, ld_s %d, ps %ld", nch_pad, ld_s, ps);
    PyrPtrs pp{};
    for (int l = 0; l < levels; ++l) { FGT_REQUIRE(pyr[l], "fgt_corr_lookup: null level"); pp.p[l] = pyr[l]; }
    FGT_REQUIRE((H1 >> (levels - 1)) >= 2 && (W1 >> (levels - 1)) >= 2, "fgt_corr_lookup: coarsest level smaller than 2x2");
    const long npix = (long)B * H1 * W1;
    const long nblk = (npix + CL_PB - 1) / CL_PB;
    FGT_REQUIRE(nblk <= 0x7fffffffl, "fgt_corr_lookup: too many query pixels");
    // unique bytes: per (pixel, level) the (2r+2)^2 window, every output form once, the coordinates
    FgtProfScope prof(FGT_PROF_CORR_LOOKUP, 0.0, (double)npix * (4.0 * levels * (2 * radius + 2) * (2 * radius + 2) + 4.0 * nch * ((out ? 1 : 0) + (out_s ? 1 : 0)) + 8.0), stream);
    static const int deep = [] { const char* e = getenv("FGT_LOOKUP_DEEP"); return e ? atoi(e) : 1; }();      // (A/B: 0 = the rolled staging loop)
    static const int table = [] { const char* e = getenv("FGT_LOOKUP_TABLE"); return e ? atoi(e) : 1; }();    // (A/B: 0 = per-tap coordinate arithmetic)
    if (radius == 4)
        hipLaunchKernelGGL(corr_lookup_kernel<4>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, pp, levels, npix, H1, W1, radius,
                           coords, out, ldo, static_cast<__bf16*>(out_s), ld_s, ps, nch_pad, deep, table);
    else
        hipLaunchKernelGGL(corr_lookup_kernel<0>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, pp, levels, npix, H1, W1, radius,
                           coords, out, ldo, static_cast<__bf16*>(out_s), ld_s, ps, nch_pad, deep, table);
    return fgt_check_launch("corr_lookup");
}

extern "C" int fgt_corr_lookup(const float* const* pyr, int levels, int B, int H1, int W1, int radius, const float* coords,
                               float* out, int ldo, void* stream) {
    return fgt_corr_lookup_split(pyr, levels, B, H1, W1, radius, coords, out, ldo, nullptr, 0, 0, 0, stream);
}

extern "C" int fgt_convex_upsample(const float* flow, int ldf, const float* mask, int ldm, int B, int H, int W, float* out,
                                   void* stream) {
    FGT_REQUIRE(flow && mask && out, "fgt_convex_upsample: bad arguments");
    hipLaunchKernelGGL(convex_up_kernel, dim3(grid_for((long)B * H * W * 64)), dim3(256), 0, (hipStream_t)stream, flow, ldf, mask,
                       ldm, B, H, W, out);
    return fgt_check_launch("convex_upsample");
}

