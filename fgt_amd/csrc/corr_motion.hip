// Round 6.  RAFT's correlation lookup FUSED with the 1 x 1 convolution that consumes it (gfx950).
//
//   RAFT/corr.py:29-50   corr = cat_l bilinear_sampler(pyramid[l], centroid_l + delta)          4 levels x 9 x 9 taps = 324 channels per pixel
//   RAFT/update.py:64,73 cor  = relu(convc1(corr))                                              324 -> 256, 1 x 1
//
// As two launches (fgt_corr_lookup_split + fgt_conv2d) the 324 taps of every pixel went out to HBM as a 352-channel split tensor (292 MB per call at
// 32 pairs of 864x480) and came back into a K = 352 GEMM that is all prologue and epilogue (175 TFLOP/s where the 3 x 3 layers reach 330): 0.43 + 0.21 ms
// per refinement iteration, 15 % of it.  Here a workgroup owns 64 pixels and all 256 output channels:
//   per level l = 0..3:   the pixels' 10 x 12 windows of pyramid level l -> LDS (row-contiguous loads; two workgroups per CU: one's scattered fetches
//                         fly under the other's matrix work) | barrier |
//                         81 bilinear taps per pixel from LDS, exactly the arithmetic of corr_lookup_kernel (same coordinate round trip, same
//                         sum order: the tap VALUES are bit-identical), split to bf16 hi / lo and written straight into the A tile
//                         [3 K-steps][64 rows][hi 32 | lo 32] (81 taps padded to 96 = three 32-channel K-steps per level) | barrier |
//                         3 K-steps of MFMAs (lo*hi, hi*lo, hi*hi per k-half as in every bf16x3 kernel here): A fragments from LDS, B fragments — the
//                         weights, re-packed per level into MFMA FRAGMENT order at pack time, 384 KB, L2-resident — straight from global memory
//   epilogue:             conv_epilogue (bias, ReLU, split output in the planes or the interleaved layout) — the shared code of the conv kernels.
// K order is (level, tap) with zero rows at taps 81..95 of a level instead of channel 0..351: the same products, another fp32 summation order than
// the two-launch path (differences at fp32 rounding, tests/test_flow_gpu.py); routed by an explicit call (fgt_corr_motion), never by tuning.
#include "conv_tile.h"
#include "flow_common.h"

namespace {

constexpr int CM_BM = 64, CM_BN = 256, CM_NW = 4;
constexpr int CM_TAPS_PAD = 96;                  // taps per level in the K walk (81 real for radius 4)
constexpr int CM_WIN = 12, CM_ROWS = 10, CM_PITCH = 13, CM_WSZ = CM_ROWS * CM_PITCH + 1;      // staged window per (pixel, level); odd size: the pixels' windows start in different banks
constexpr int CM_NLD = (CM_BM * CM_ROWS * CM_WIN + 255) / 256;                                 // window values per thread and level (30)
constexpr int CM_STAGE = 8192;                   // floats: conv_epilogue's view of the LDS (fast path: 4 wavefronts x 32 x 64 floats = 32 KB from the base)
constexpr size_t CM_SMEM = 64 * 1024;            // windows 33.5 KB + A tile 24 KB; 64 KB so that the epilogue's general path would fit too

struct CorrMotionP {
    PyrPtrs pyr;
    long npix;
    int H1, W1, radius;
    const float* coords;
    const __bf16* wfrag;         // [12 K-steps][4 column blocks of 64][2 x 32 columns][hi | lo][2 k-halves][64 lanes][8] bf16
    ConvP ep;                    // what conv_epilogue reads: bias, activation, split output
};

__global__ void __launch_bounds__(256, 2) corr_motion_kernel(const CorrMotionP a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int wb[4][CM_BM][2];              // window origin (x, y) per level and pixel
    float* const win = smem;                     // [64][CM_WSZ]
    char* const At = reinterpret_cast<char*>(smem + CM_BM * CM_WSZ);      // [3][64 rows][128 bytes], 16-byte slots swizzled by (row >> 1) & 7 (conv_wide.hip's image)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const long q0 = (long)blockIdx.x * CM_BM;
    const int npx = (int)min((long)CM_BM, a.npix - q0);
    const int radius = a.radius, side = 2 * radius + 1, per_lvl = side * side;

    for (int i = tid; i < 4 * CM_BM; i += 256) {
        const int lvl = i / CM_BM, pl = i - lvl * CM_BM;
        int bx = 0, by = 0;
        if (pl < npx) {
            const float scale = (float)(1 << lvl);
            bx = (int)floorf(a.coords[(q0 + pl) * 2] / scale) - radius - 1;
            by = (int)floorf(a.coords[(q0 + pl) * 2 + 1] / scale) - radius;
        }
        wb[lvl][pl][0] = bx;
        wb[lvl][pl][1] = by;
    }
    // this thread's pixel in the tap phase
    const int pp = tid & 63, g0 = tid >> 6;
    const bool pvalid = pp < npx;
    const long qp = q0 + (pvalid ? pp : 0);
    const float X = a.coords[qp * 2], Y = a.coords[qp * 2 + 1];
    __syncthreads();

    // (a run-time index into the kernel-argument struct made hipcc copy ALL of it — 450 bytes with the epilogue's ConvP — to scratch and read every
    //  field back from there: 1 665 scratch loads in the first build.  The level's volume is selected with compares instead.)
    auto level_ptr = [&](int lvl) __attribute__((always_inline)) -> const float* {
        return lvl == 0 ? a.pyr.p[0] : lvl == 1 ? a.pyr.p[1] : lvl == 2 ? a.pyr.p[2] : a.pyr.p[3];
    };
    // ---- the 64 pixels' 10 x 12 windows of one level -> LDS (zeros outside the map / past the last pixel), six loads in flight per thread and batch.
    // (Requesting a whole level — 30 values per thread — into registers ahead of the previous level's matrix work spilled 70-160 registers beside the
    //  64 accumulators; the second workgroup of the CU is what runs under this one's fetches.)
    auto stage = [&](int lvl) __attribute__((always_inline)) {
        const int Hl = a.H1 >> lvl, Wl = a.W1 >> lvl;
        const float* vol = level_ptr(lvl);
#pragma unroll 6
        for (int l = 0; l < CM_NLD; ++l) {
            const int idx = tid + l * 256;
            const int w = idx / (CM_ROWS * CM_WIN), e = idx - w * (CM_ROWS * CM_WIN);
            const int j = e / CM_WIN, i = e - j * CM_WIN;
            float v = 0.f;
            if (w < npx) {
                const int gx = wb[lvl][w][0] + i, gy = wb[lvl][w][1] + j;
                if (gx >= 0 && gx < Wl && gy >= 0 && gy < Hl) v = vol[(q0 + w) * Hl * Wl + (long)gy * Wl + gx];
            }
            win[w * CM_WSZ + j * CM_PITCH + i] = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int rsw = (l31 >> 1) & 7;
    const __bf16* const wl = a.wfrag + (long)wave * (2 * 2 * 2 * 512) + lane * 8;      // this wavefront's 64 columns, this lane's 8 values of every fragment

#pragma unroll 1
    for (int lvl = 0; lvl < 4; ++lvl) {
        stage(lvl);                                  // (every wavefront is past the previous level's tap phase: two barriers ago)
        __syncthreads();
        // ---- taps: thread (pixel pp, 8-tap groups g0, g0 + 4, g0 + 8) -> bf16 hi / lo -> A tile
        {
            const int Hl = a.H1 >> lvl, Wl = a.W1 >> lvl;
            const float scale = (float)(1 << lvl);
            const int bx = wb[lvl][pp][0], by = wb[lvl][pp][1];
            const float* wp = win + pp * CM_WSZ;
#pragma unroll 1
            for (int gh = 0; gh < 6; ++gh) {         // (one half group = 4 taps at a time: 24 taps' worth of coordinate arithmetic in flight at once spilled 159 registers)
                const int g = g0 + 4 * (gh >> 1), hf = gh & 1;       // K-step g >> 2 of this level, 16-byte slot g & 3, its first / second 8 bytes
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = g * 8 + hf * 4 + u;
                    v[u] = 0.f;
                    if (t < per_lvl && pvalid) {
                        const int ta = t / side, tb = t - ta * side;
                        const float cx = X / scale + (float)(ta - radius);
                        const float cy = Y / scale + (float)(tb - radius);
                        float ix, iy;
                        sample_coord(cx, cy, 0, 0, Wl, Hl, 1, 1, ix, iy);
                        const Bilin bl = bilin(ix, iy);
                        const int rx = bl.x0 - bx, ry = bl.y0 - by;
                        if ((unsigned)rx < (unsigned)(CM_WIN - 1) && (unsigned)ry < (unsigned)(CM_ROWS - 1)) {
                            const float* wv = wp + ry * CM_PITCH + rx;
                            float s = 0.f;
                            s += wv[0] * bl.wnw;
                            s += wv[1] * bl.wne;
                            s += wv[CM_PITCH] * bl.wsw;
                            s += wv[CM_PITCH + 1] * bl.wse;
                            v[u] = s;
                        } else {
                            v[u] = corr_tap_global(level_ptr(lvl) + qp * Hl * Wl, Hl, Wl, bl);
                        }
                    }
                }
                uint2 h0, l0;
                fgt_split4(make_float4(v[0], v[1], v[2], v[3]), h0, l0);
                char* row = At + ((g >> 2) * CM_BM + pp) * 128;
                const int sl = ((g & 3) ^ ((pp >> 1) & 7)) * 16 + hf * 8;
                *reinterpret_cast<uint2*>(row + sl) = h0;
                *reinterpret_cast<uint2*>(row + (sl ^ 64)) = l0;
            }
        }
        __syncthreads();
        // ---- 3 K-steps: A from LDS, B from global (fragment order: one 1-KB run per wave-instruction)
#pragma unroll 1
        for (int kk = 0; kk < 3; ++kk) {
            const int s = lvl * 3 + kk;
            const __bf16* wf = wl + (long)s * (4 * 2 * 2 * 2 * 512);
            const char* base = At + kk * CM_BM * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 bh[2], bl_[2], ah[2], al[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8*>(wf + ((j * 2 + 0) * 2 + ks) * 512);
                    bl_[j] = *reinterpret_cast<const bf16x8*>(wf + ((j * 2 + 1) * 2 + ks) * 512);
                }
                const int so = ((ks * 2 + lh) ^ rsw) * 16;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const char* A = base + (i * 32 + l31) * 128 + so;
                    ah[i] = *reinterpret_cast<const bf16x8*>(A);
                    al[i] = *reinterpret_cast<const bf16x8*>(A + ((so ^ 64) - so));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl_[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                 // the epilogue stages through the LDS the last level's fragments were read from
    conv_epilogue<CM_BM, CM_BN, 1, CM_NW, CM_STAGE, 2, 2, false, false>(a.ep, acc, smem, (int)q0, 0, 0);
}

}  // namespace

extern "C" int fgt_corr_motion(const float* const* pyr, int levels, int B, int H1, int W1, int radius, const float* coords, const void* w_frag,
                               const float* bias, void* out_s, int ld_s, long long ps, void* stream) {
    FGT_REQUIRE(pyr && coords && w_frag && out_s, "fgt_corr_motion: null pointer");
    FGT_REQUIRE(levels == 4 && radius >= 1 && radius <= (CM_ROWS - 2) / 2 && (2 * radius + 1) * (2 * radius + 1) <= CM_TAPS_PAD,
                "fgt_corr_motion: built for 4 levels and radius <= %d (got %d levels, radius %d)", (CM_ROWS - 2) / 2, levels, radius);
    FGT_REQUIRE(B > 0 && H1 > 0 && W1 > 0 && (H1 >> 3) >= 2 && (W1 >> 3) >= 2, "fgt_corr_motion: coarsest level smaller than 2x2");
    const long npix = (long)B * H1 * W1;
    FGT_REQUIRE(npix < (1l << 31), "fgt_corr_motion: too many query pixels");
    FGT_REQUIRE(((uintptr_t)w_frag & 15) == 0 && ((uintptr_t)out_s & 7) == 0 && (!bias || ((uintptr_t)bias & 15) == 0), "fgt_corr_motion: pointer alignment");
    FGT_REQUIRE(ld_s % 4 == 0 && ((ps == 32 && ld_s >= 2 * CM_BN && ld_s % 64 == 0) || (ps > 0 && ps != 32 && ps % 4 == 0 && ld_s >= CM_BN)),
                "fgt_corr_motion: split output: ld_s %d, ps %lld", ld_s, ps);
    CorrMotionP a{};
    for (int l = 0; l < 4; ++l) { FGT_REQUIRE(pyr[l], "fgt_corr_motion: null level"); a.pyr.p[l] = pyr[l]; }
    a.npix = npix; a.H1 = H1; a.W1 = W1; a.radius = radius; a.coords = coords;
    a.wfrag = static_cast<const __bf16*>(w_frag);
    ConvP& p = a.ep;
    p.d.N = 1; p.d.H = 1; p.d.W = (int)npix; p.d.Ho = 1; p.d.Wo = (int)npix;
    p.d.Cout = CM_BN; p.d.groups = 1;
    p.d.act = FGT_ACT_RELU; p.d.slope = 0.f; p.d.out_scale = 1.f; p.d.epi = FGT_EPI_NONE;
    p.d.out_split = 1; p.d.ldo_s = ld_s; p.d.ooff_s = 0; p.d.pso = ps;
    p.M = (int)npix; p.HoWo = (int)npix; p.Cout_g = CM_BN;
    p.cbias = bias; p.out_s = static_cast<__bf16*>(out_s); p.pso = ps;
    p.zero_page = fgt_zero_page();
    FGT_REQUIRE(p.zero_page != nullptr, "fgt_corr_motion: could not allocate the zero page");
    static const int nt_env = [] { const char* e = getenv("FGT_CONV_NT"); return e ? atoi(e) : 1; }();
    p.nt_store = nt_env;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&corr_motion_kernel), (int)CM_SMEM, lds_set, "corr_motion")) return rc;
    const int side = 2 * radius + 1, nch = 4 * side * side;
    // credited as the two launches it replaces: the lookup's window bytes + the split output, and the GEMM's 2 * rows * 324 * 256 flops
    FgtProfScope prof(FGT_PROF_CONV, 2.0 * (double)npix * nch * CM_BN, (double)npix * (4.0 * 4 * (2 * radius + 2) * (2 * radius + 2) + 4.0 * CM_BN + 8.0), stream);
    hipLaunchKernelGGL(corr_motion_kernel, dim3((unsigned)((npix + CM_BM - 1) / CM_BM)), dim3(256), CM_SMEM, (hipStream_t)stream, a);
    return fgt_check_launch("corr_motion");
}
