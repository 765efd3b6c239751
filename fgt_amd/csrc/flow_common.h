// Coordinate / bilinear-tap arithmetic shared by the flow-side kernels (flow_ops.hip, corr_motion.hip): the reference's fp32 arithmetic step by step.
#pragma once
#include "common.h"

namespace {

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

}  // namespace
