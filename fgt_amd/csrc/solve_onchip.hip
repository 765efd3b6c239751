// On-chip conjugate gradients for the two sparse solves of the tool: the diffusion fill of the flows
// (tool/utils/region_fill.py:7-63) and the Poisson blend of the propagated gradients (tool/utils/Poisson_blend_img.py:19-75).
//
// laplace_fill.hip / poisson_blend.hip run one CG iteration as two launches over the whole frame: ~57 us per launch whatever the hole, 2 000 -
// 3 000 launches per call (115 ms per direction / 269 ms per clip: each of the two stages cost 2-3x the whole FGT stage).  The problems
// are small — one per (flow map) or (frame, channel), ~17-33 k unknowns inside the hole's bounding box — and there are 158-240 of them:
// exactly one per CU.  Here ONE WORKGROUP solves ONE problem entirely on chip, all iterations inside one launch:
//   * the search direction p lives in LDS as a dense image of the hole's bounding box (+ a one-cell halo, cells outside the hole stay 0, so
//     the 5-point stencil needs no index lists and no predicates on the neighbour reads: p = 0 there is exactly "not an unknown");
//   * r and q = A p live in registers: a thread owns SPT horizontal strips of 4 cells (16-byte LDS reads of the rows above / below); the
//     solution x is accumulated in place in the output map (one aligned float4 read-modify-write per strip and iteration through L2:
//     nobody else touches it) — three values per cell in registers would cap a workgroup at ~20 k cells, two leave room for ~27 k;
//   * the two dot products of an iteration are fixed-order double-precision block reductions (wave shuffles, then the wavefront partials in
//     order): no atomics — results are bit-reproducible run to run — and a problem simply LEAVES its loop when |r| <= tol |r0|.
// Same iteration as the multi-launch kernels (same alpha / beta in fp32, same stencil order), 3 barriers per iteration instead of 2
// launches.  Capacity: NT * SPT strips and (rows + 2) x (4 * strips per row + 8) floats of LDS; the caller passes host-known upper
// bounds of the bounding boxes (fgt_mask_bbox + one read-back in the Python wrapper) and gets FGT_EINVAL when a problem cannot fit, in
// which case it takes the multi-launch path (any hole shape, any size).  A device-side check guards the bounds all the same: a problem
// that does not fit its instantiation is filled with NaN and reported in `status`, never written out of bounds.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ bounding boxes of the masks
__global__ void bbox_init(int* bbox, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { bbox[4 * i] = 1 << 30; bbox[4 * i + 1] = 1 << 30; bbox[4 * i + 2] = -1; bbox[4 * i + 3] = -1; }
}

__global__ void __launch_bounds__(256) bbox_kernel(const unsigned char* mask, int H, int W, int* bbox) {
    const int m = blockIdx.y;
    const unsigned char* mk = mask + (long)m * H * W;
    int y0 = 1 << 30, x0 = 1 << 30, y1 = -1, x1 = -1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        if (mk[i]) {
            const int y = i / W, x = i - y * W;
            y0 = min(y0, y); y1 = max(y1, y); x0 = min(x0, x); x1 = max(x1, x);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        y0 = min(y0, __shfl_xor(y0, o, 64)); x0 = min(x0, __shfl_xor(x0, o, 64));
        y1 = max(y1, __shfl_xor(y1, o, 64)); x1 = max(x1, __shfl_xor(x1, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && y1 >= 0) {
        atomicMin(bbox + 4 * m, y0); atomicMin(bbox + 4 * m + 1, x0); atomicMax(bbox + 4 * m + 2, y1); atomicMax(bbox + 4 * m + 3, x1);
    }
}

// ------------------------------------------------------------------------------------------------ shared pieces
// fixed-order sum over the workgroup: xor shuffles inside a wavefront, then the NT / 64 wavefront partials in order.  `sh` alternates
// between two scratch rows (parity) so that one barrier per reduction is enough.
template <int NT>
__device__ __forceinline__ double block_sum(double v, double* sh, int& parity) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    double* row = sh + parity * (NT / 64);
    if ((threadIdx.x & 63) == 0) row[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) s += row[w];
    parity ^= 1;
    return s;
}

struct Box {
    int y0, x0, Hb, Wq, Wp;     // origin, rows, strips per row, LDS row stride (floats)
};

// geometry of problem's bounding box; false = empty mask
__device__ __forceinline__ bool load_box(const int* bb, Box& B) {
    const int y0 = bb[0], x0 = bb[1], y1 = bb[2], x1 = bb[3];
    if (y1 < y0 || x1 < x0) return false;
    const int xa = x0 & ~3;                       // strips start at absolute multiples of 4 columns: a strip of x is one aligned float4
    B.y0 = y0; B.x0 = xa; B.Hb = y1 - y0 + 1; B.Wq = (x1 - xa + 4) / 4; B.Wp = B.Wq * 4 + 8;
    return true;
}

constexpr int RED_DOUBLES = 2 * 16;       // reduction scratch at the start of the dynamic LDS (two rows of <= 16 wavefront partials)

// ------------------------------------------------------------------------------------------------ the solver (both problems)
struct SolveP {
    // diffusion fill (BLEND = false): problem b = map b, mask b % n_masks
    const float* I;              // [B, H, W]
    // Poisson blend (BLEND = true): problem b = (frame b / 3, channel b % 3)
    const float *trg, *gx, *gy;  // [N, H, W, 3]
    const unsigned char* ecode;  // [N, H, W]: 2 bits per direction (0 right, 1 down, 2 left, 3 up): 0 none, 1 known neighbour, 2 hole neighbour
    // common
    const unsigned char* mask;   // [n_masks, H, W] (blend: the hole masks, n_masks = N)
    const int* bbox;             // [n_masks, 4]
    float* x;                    // [problems, H, W]: holds I / the target outside the hole on entry; the solution is accumulated in place
    int* status;                 // [problems]: 2 * iterations used; 1 = did not fit (NaN written)
    int problems, H, W, n_masks, iters, lds_floats;
    float tol2;
};

template <bool BLEND, int NT, int SPT>
__global__ void __launch_bounds__(NT) solve_onchip_kernel(const SolveP P) {
    extern __shared__ __attribute__((aligned(16))) double smem_d[];
    double* sh = smem_d;
    float* Ps = reinterpret_cast<float*>(smem_d + RED_DOUBLES);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int mi = BLEND ? b / 3 : b % P.n_masks;
    const int ch = BLEND ? b - 3 * mi : 0;
    const long HW = (long)P.H * P.W;
    const unsigned char* mk = P.mask + (long)mi * HW;
    const unsigned char* ec = BLEND ? P.ecode + (long)mi * HW : nullptr;
    const long fb = (long)mi * HW;                       // blend: pixel base of the frame in the channels-last inputs
    float* xg = P.x + (long)b * HW;
    Box X;
    if (!load_box(P.bbox + 4 * mi, X)) { if (tid == 0) P.status[b] = 0; return; }
    const int nstr = X.Hb * X.Wq;
    if (nstr > NT * SPT || (X.Hb + 2) * X.Wp > P.lds_floats) {          // does not fit this instantiation: loud, never out of bounds
        for (int i = tid; i < nstr * 4; i += NT) {
            const int y = X.y0 + i / (X.Wq * 4), xx = X.x0 + i % (X.Wq * 4);
            if (mk[(long)y * P.W + xx]) xg[(long)y * P.W + xx] = __builtin_nanf("");
        }
        if (tid == 0) P.status[b] = 1;
        return;
    }
    for (int i = tid; i < (X.Hb + 2) * X.Wp; i += NT) Ps[i] = 0.f;

    float r[SPT][4], q[SPT][4];
    unsigned mb[SPT];           // bit j: cell j of the strip is an unknown; bits 4+4j..7+4j: the diagonal of cell j (fill: n(p); blend: d_p);
                                // blend: cd[s] bit 4j+n = direction n (0 right, 1 down, 2 left, 3 up) of cell j couples to an unknown (weight 2)
    unsigned cd[BLEND ? SPT : 1];
    int lo[SPT];                // LDS offset of the strip's first cell
    int go[SPT];                // offset of the strip's first cell in the problem's map (an aligned float4)
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        const int st = tid + s * NT;
        mb[s] = 0; lo[s] = 0; go[s] = 0;
        if constexpr (BLEND) cd[s] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[s][j] = 0.f; q[s][j] = 0.f; }
        if (st < nstr) {
            const int ys = st / X.Wq, xs = (st - ys * X.Wq) * 4;
            lo[s] = (ys + 1) * X.Wp + 4 + xs;
            const int Y = X.y0 + ys;
            go[s] = Y * P.W + X.x0 + xs;
            float4 xi = *reinterpret_cast<const float4*>(xg + go[s]);       // x0 = 0 at the unknowns, everything else stays what it is
            float xv[4] = {xi.x, xi.y, xi.z, xi.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int Xc = X.x0 + xs + j;
                const long i = (long)Y * P.W + Xc;
                if (!mk[i]) continue;
                xv[j] = 0.f;
                mb[s] |= 1u << j;
                float rhs = 0.f;
                if constexpr (!BLEND) {
                    mb[s] |= (unsigned)((Y > 0) + (Y < P.H - 1) + (Xc > 0) + (Xc < P.W - 1)) << (4 + 4 * j);     // region_fill.py:104-117
                    const float* I = P.I + (long)b * HW;                // region_fill.py:66-101 (formRightSide), the order of fill_init
                    if (Y > 0 && !mk[i - P.W]) rhs += I[i - P.W];
                    if (Y < P.H - 1 && !mk[i + P.W]) rhs += I[i + P.W];
                    if (Xc > 0 && !mk[i - 1]) rhs += I[i - 1];
                    if (Xc < P.W - 1 && !mk[i + 1]) rhs += I[i + 1];
                } else {
                    const unsigned code = ec[i];
                    {   // per cell: 4 flags "direction n couples to an unknown" (weight 2) and the diagonal d_p = 2 #hole-edges + #known-edges
                        unsigned fl = 0, dg = 0;
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            const unsigned e = (code >> (2 * n)) & 3u;
                            fl |= (e == 2 ? 1u : 0u) << n;
                            dg += e == 2 ? 2u : (e == 1 ? 1u : 0u);
                        }
                        cd[s] |= fl << (4 * j);
                        mb[s] |= dg << (4 + 4 * j);
                    }
                    // right-hand side exactly as blend_init (Poisson_blend_img.py:171-244 folded into the normal equations)
                    const float rr4[4] = {-P.gx[(fb + i) * 3 + ch], -P.gy[(fb + i) * 3 + ch],
                                          (code >> 4) & 3 ? P.gx[(fb + i - 1) * 3 + ch] : 0.f, (code >> 6) & 3 ? P.gy[(fb + i - P.W) * 3 + ch] : 0.f};
                    const int off[4] = {1, P.W, -1, -P.W};
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const unsigned e = (code >> (2 * n)) & 3u;
                        if (e == 2) rhs += 2.f * rr4[n];
                        else if (e == 1) rhs += rr4[n] + P.trg[(fb + i + off[n]) * 3 + ch];
                    }
                }
                r[s][j] = rhs;
                acc += (double)rhs * rhs;
            }
            if (mb[s]) *reinterpret_cast<float4*>(xg + go[s]) = make_float4(xv[0], xv[1], xv[2], xv[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    int parity = 0;
    __syncthreads();                                                    // the zeroed image is complete before anyone writes p into it
    const double rr0 = block_sum<NT>(acc, sh, parity);
    double rr = rr0, rr_old = 0.0;
    int k = 0;
    for (; k < P.iters; ++k) {
        if (!(rr > (double)P.tol2 * rr0)) break;                        // converged (or rr0 == 0): every thread takes the same branch
        const float beta = (k > 0 && rr_old > 0.0) ? (float)(rr / rr_old) : 0.f;
        // p_k = r + beta p_{k-1} at this thread's own cells
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            if (mb[s]) {
                float4 pv = *reinterpret_cast<const float4*>(Ps + lo[s]);
                pv.x = (mb[s] & 1u) ? r[s][0] + beta * pv.x : 0.f;
                pv.y = (mb[s] & 2u) ? r[s][1] + beta * pv.y : 0.f;
                pv.z = (mb[s] & 4u) ? r[s][2] + beta * pv.z : 0.f;
                pv.w = (mb[s] & 8u) ? r[s][3] + beta * pv.w : 0.f;
                *reinterpret_cast<float4*>(Ps + lo[s]) = pv;
            }
            __builtin_amdgcn_sched_barrier(0);      // one strip at a time: hoisting every strip's LDS loads costs 14 registers per strip
        }
        __syncthreads();
        // q = A p (cells outside the hole hold 0: Dirichlet neighbours drop out), partial p.q
        double apq = 0.0;
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            if (mb[s]) {
                const float4 c = *reinterpret_cast<const float4*>(Ps + lo[s]);
                const float4 u = *reinterpret_cast<const float4*>(Ps + lo[s] - X.Wp);
                const float4 d = *reinterpret_cast<const float4*>(Ps + lo[s] + X.Wp);
                const float l = Ps[lo[s] - 1], rt = Ps[lo[s] + 4];
                const float pc[4] = {c.x, c.y, c.z, c.w};
                // neighbours in the edge codes' direction order: 0 right, 1 down, 2 left, 3 up
                const float nb[4][4] = {{c.y, d.x, l, u.x}, {c.z, d.y, c.x, u.y}, {c.w, d.z, c.y, u.z}, {rt, d.w, c.z, u.w}};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v;
                    if constexpr (!BLEND) {
                        v = (float)((mb[s] >> (4 + 4 * j)) & 15u) * pc[j];
                        v -= nb[j][3]; v -= nb[j][1]; v -= nb[j][2]; v -= nb[j][0];          // up, down, left, right: the order of fill_apply
                    } else {
                        const unsigned fl = cd[s] >> (4 * j);
                        v = 0.f;
#pragma unroll
                        for (int n = 0; n < 4; ++n) v -= ((fl >> n) & 1u) ? 2.f * nb[j][n] : 0.f;
                        v += (float)((mb[s] >> (4 + 4 * j)) & 15u) * pc[j];
                    }
                    v = ((mb[s] >> j) & 1u) ? v : 0.f;
                    q[s][j] = v;
                    apq += (double)v * pc[j];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const double pq = block_sum<NT>(apq, sh, parity);
        if (!(pq > 0.0)) break;
        const float alpha = (float)(rr / pq);
        double arr = 0.0;
        // x += alpha p in place in the output map (this workgroup is the only one that touches it), r -= alpha q, partial r.r
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            if (mb[s]) {
                const float4 c = *reinterpret_cast<const float4*>(Ps + lo[s]);
                float4 xv = *reinterpret_cast<const float4*>(xg + go[s]);
                xv.x += alpha * c.x; xv.y += alpha * c.y; xv.z += alpha * c.z; xv.w += alpha * c.w;      // (p = 0 outside the hole: x + 0 = x)
                *reinterpret_cast<float4*>(xg + go[s]) = xv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float rn = r[s][j] - alpha * q[s][j];
                    r[s][j] = rn;
                    arr += (double)rn * rn;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        rr_old = rr;
        rr = block_sum<NT>(arr, sh, parity);
    }
    if (tid == 0) P.status[b] = k << 1;
}

// instantiations: (threads, strips per thread) -> capacity NT * SPT strips of 4 cells
struct Inst { int nt, spt; };
constexpr Inst INSTS[] = {{1024, 1}, {1024, 2}, {1024, 4}, {512, 10}, {512, 12}};
constexpr int LDS_BYTES_MAX = 160 * 1024;

// smallest instantiation that holds max_rows x max_cols boxes; -1 = none
inline int pick_inst(int max_rows, int max_cols, int& lds_floats) {
    const long wq = (max_cols + 6) / 4;                 // strips start at the multiple of 4 below the box's first column: up to 3 extra cells
    const long nstr = (long)max_rows * wq;
    const long fl = (long)(max_rows + 2) * (wq * 4 + 8);
    if (fl * 4 + RED_DOUBLES * 8 > LDS_BYTES_MAX) return -1;
    lds_floats = (int)fl;
    for (int i = 0; i < (int)(sizeof(INSTS) / sizeof(INSTS[0])); ++i)
        if ((long)INSTS[i].nt * INSTS[i].spt >= nstr) return i;
    return -1;
}

template <bool BLEND, int NT, int SPT>
int launch_one(const SolveP& P, hipStream_t s) {
    static std::atomic<unsigned long long> done{0};
    const int bytes = P.lds_floats * 4 + RED_DOUBLES * 8;
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&solve_onchip_kernel<BLEND, NT, SPT>), LDS_BYTES_MAX, done, "solve_onchip")) return rc;
    hipLaunchKernelGGL((solve_onchip_kernel<BLEND, NT, SPT>), dim3(P.problems), dim3(NT), bytes, s, P);
    return fgt_check_launch("solve_onchip");
}

template <bool BLEND>
int launch_inst(int inst, const SolveP& P, hipStream_t s) {
    switch (inst) {
        case 0: return launch_one<BLEND, 1024, 1>(P, s);
        case 1: return launch_one<BLEND, 1024, 2>(P, s);
        case 2: return launch_one<BLEND, 1024, 4>(P, s);
        case 3: return launch_one<BLEND, 512, 10>(P, s);
        default: return launch_one<BLEND, 512, 12>(P, s);
    }
}

__global__ void __launch_bounds__(256) copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

}  // namespace

extern "C" int fgt_mask_bbox(const unsigned char* mask, int n, int H, int W, int* bbox, void* stream) {
    FGT_REQUIRE(mask && bbox && n > 0 && H > 0 && W > 0 && (long)H * W < (1l << 30), "fgt_mask_bbox: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bbox_init, dim3(cdiv(n, 256)), dim3(256), 0, s, bbox, n);
    const int gx = cdiv((long)H * W, 256 * 8) > 64 ? 64 : cdiv((long)H * W, 256 * 8);
    hipLaunchKernelGGL(bbox_kernel, dim3(gx, n), dim3(256), 0, s, mask, H, W, bbox);
    return fgt_check_launch("mask_bbox");
}

extern "C" int fgt_laplace_fill_onchip(const float* I, const unsigned char* mask, const int* bbox, int B, int n_masks, int H, int W, float* out,
                                       int max_rows, int max_cols, int iters, float tol, int* status, void* stream) {
    FGT_REQUIRE(I && mask && bbox && out && status, "fgt_laplace_fill_onchip: null pointer");
    FGT_REQUIRE(B > 0 && n_masks > 0 && H > 0 && W > 0 && iters >= 0 && tol >= 0.f && max_rows >= 0 && max_cols >= 0, "fgt_laplace_fill_onchip: bad sizes");
    FGT_REQUIRE((((uintptr_t)I | (uintptr_t)out) & 15) == 0 && W % 4 == 0, "fgt_laplace_fill_onchip: maps must be 16-byte aligned with W %% 4 == 0 (use fgt_laplace_fill)");
    SolveP P{};
    P.I = I; P.mask = mask; P.bbox = bbox; P.x = out; P.status = status; P.problems = B; P.H = H; P.W = W; P.n_masks = n_masks; P.iters = iters;
    P.tol2 = tol * tol;
    const int inst = pick_inst(max_rows, max_cols, P.lds_floats);
    FGT_REQUIRE(inst >= 0, "fgt_laplace_fill_onchip: a %d x %d bounding box does not fit one workgroup (use fgt_laplace_fill)", max_rows, max_cols);
    hipStream_t s = (hipStream_t)stream;
    const long n4 = (long)B * H * W / 4;
    if (out != I) hipLaunchKernelGGL(copy_kernel, dim3(n4 / 256 + 1 > 8192 ? 8192 : (int)(n4 / 256 + 1)), dim3(256), 0, s, reinterpret_cast<const float4*>(I), reinterpret_cast<float4*>(out), n4);
    return launch_inst<false>(inst, P, s);
}

// Internal entry of fgt_poisson_blend_onchip (poisson_blend.hip): the CG part on chip.  x ([N*3, H, W] planar) must already hold the target at
// the known pixels (blend_init wrote it).
int fgt_blend_onchip(const float* trg, const float* gx, const float* gy, const unsigned char* hole, const unsigned char* ecode, const int* bbox,
                     float* x, int* status, int N, int H, int W, int max_rows, int max_cols, int iters, float tol, hipStream_t s) {
    if (W % 4 != 0 || ((uintptr_t)x & 15) != 0) { fgt_set_error("fgt_poisson_blend_onchip: needs W %% 4 == 0 (use fgt_poisson_blend)"); return FGT_EINVAL; }
    SolveP P{};
    P.trg = trg; P.gx = gx; P.gy = gy; P.mask = hole; P.ecode = ecode; P.bbox = bbox; P.x = x; P.status = status;
    P.problems = N * 3; P.n_masks = N; P.H = H; P.W = W; P.iters = iters; P.tol2 = tol * tol;
    const int inst = pick_inst(max_rows, max_cols, P.lds_floats);
    if (inst < 0) { fgt_set_error("fgt_poisson_blend_onchip: a %d x %d bounding box does not fit one workgroup (use fgt_poisson_blend)", max_rows, max_cols); return FGT_EINVAL; }
    return launch_inst<true>(inst, P, s);
}
