// DIAGNOSTIC BUILDS ONLY (fgt_amd.build.build(variant="diag")): measured in round 3 and NOT adopted — correct (tests/test_taps_gpu.py: 20 geometries,
// errors vs fp64 equal to conv_split's), 31-39 % fewer LDS-DMA instructions per flop, and 4-12 % SLOWER than the early-release kernel on every
// layer but one (profiles/r03_run6_split_sweep_taps_vs_early_release.txt): the K loop is not bound by the LDS-DMA instruction count alone.
//
// bf16x3 implicit-GEMM convolution for STRIDE-1 "SAME" convolutions with kw >= 3 on pre-split (planes) operands: the im2col rows of a
// (ky, 32-channel chunk) stay in LDS for ALL kx taps (gfx950).
//
// conv_split.hip streams, per K-step, a 16 KB A (im2col) tile and a 16 KB B (weight) tile; its K loop runs at the rate the vector-memory
// pipe delivers 1-KB LDS-DMA instructions to a CU (NOTEBOOK.md: the fp16 kernel — same instruction count per step, a third of the MFMAs —
// takes the same time per step).  For a stride-1 convolution whose output map has the size of its input map, the A tile of tap (ky, kx) is
// the A tile of tap (ky, 0) shifted by kx * dw PIXELS in the flattened (n, y, x) index: output pixel m reads input pixel
// m + (ky*dh - ph) * W + (kx*dw - pw) whenever that pixel is in the same image row.  So this kernel walks K in the order
// (ky, chunk, kx) and loads, per (ky, chunk), BM + 16 consecutive input rows ONCE (9 pieces per plane instead of 8 per tap): for a 3x3
// layer 18 + 3 * 16 = 66 LDS-DMA instructions per (ky, chunk) instead of 96 (-31 %), for the 1x5 GRU convs of RAFT 18 + 80 instead of
// 160 (-39 %).  Tap kx reads its MFMA fragments from LDS rows r + kx*dw; a pixel whose tap leaves the image row (x + kx*dw - pw outside
// [0, W)) reads a zero row instead (per-lane address select); rows whose input row y + ky*dh - ph is outside the image were never
// fetched (the DMA read the zero page).
//
// Numerics: the same products as conv_split.hip, accumulated in the order (ky, chunk, kx) instead of (ky, kx, chunk) — NOT bit-identical
// to the other kernels (fp32 accumulation order), identical in error (tests/test_taps_gpu.py: both within 2e-5 of fp64 on the same split
// operands).  Which kernel a layer runs on is therefore decided by its GEOMETRY alone (fgt_conv_taps_eligible), never by the autotuner:
// an eligible split-input layer always runs here, so results do not depend on tuning.  Tiles of this kernel are bit-identical to each other.
//
// LDS: two A buffers [hi: BM+16 rows | lo: BM+16 rows] of 64-byte rows (the four 16-byte slots XOR-swizzled with (row >> 2) & 3 as in
// conv_tile.h: any 16 consecutive rows are conflict free, so the shifted reads are too), two B stages [hi BN | lo BN], one zero row.
// 128x128: 2 * 18 KB + 2 * 16 KB = 68 KB: two workgroups per CU.  Schedule: per step the B tile of the next step and a third of the
// next (ky, chunk)'s A rows are requested at the top, fragments read, MFMAs, vmcnt(0), barrier.
#include "../conv_tile.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

constexpr int HALO = 16;          // extra A rows per (ky, chunk): (kw - 1) * dw <= 16

template <int BM, int BN, int WM, int WN, int MINW>
__global__ void __launch_bounds__(WM* WN * 64, MINW) conv_taps_kernel(const ConvP p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int AR = BM + HALO;                        // A rows per plane
    constexpr int GA = AR / 16, GB = BN / 16;            // 16-row DMA groups per plane
    constexpr int NPA = 2 * GA;                          // A pieces per (ky, chunk): piece j -> plane j / GA, group j % GA
    constexpr int B_IT = 2 * GB / NW;                    // B pieces per wavefront and step
    constexpr int A_BYTES = 2 * AR * 64, B_BYTES = 2 * BN * 64;
    constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES + 64;
    constexpr int STAGE = LDS_BYTES / 8;                 // floats in half of the LDS (the epilogue's view of its scratch)
    constexpr int APW = (NPA + NW - 1) / NW;             // A pieces a wavefront owns per (ky, chunk)
    static_assert(AR % 16 == 0 && (2 * GB) % NW == 0 && B_IT >= 1 && TM >= 1 && TN >= 1 && APW <= 5, "tile / wavefront geometry (APW < 2 kw for kw >= 3)");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    char* const lds = reinterpret_cast<char*>(smem);
    char* const Abuf = lds;                              // [2][hi AR rows | lo AR rows]
    char* const Bbuf = lds + 2 * A_BYTES;                // [2][hi BN rows | lo BN rows]
    char* const zrow = lds + 2 * A_BYTES + 2 * B_BYTES;  // 64 zero bytes
    if (tid < 16) reinterpret_cast<float*>(zrow)[tid] = 0.f;

    const __bf16* const x0 = reinterpret_cast<const __bf16*>(p.x0);
    const __bf16* const x1 = reinterpret_cast<const __bf16*>(p.x1);
    const int ld0 = d.ld0, ld1 = d.ld1;
    const int chb0 = d.off0 + g * p.Cg0, chb1 = d.off1 + g * p.Cg1 - p.Cg0;     // (interleaved: off* are logical channels, multiples of 32)
    const long ps0 = p.ps0, ps1 = p.ps1;
    const bool il = d.in_split == 2;
    const int Cg0 = p.Cg0, nchunk = p.Cg / 32;
    const int W = d.W, H = d.H, kw = d.kw, kh = d.kh, dwx = d.dw;
    const long HW = (long)H * W, NHW = (long)d.N * HW;

    // ---- this lane's A rows: row (lane >> 2) of each of its pieces j = wave + it * NW (plane j / GA, group j % GA), chunk column kc
    const int lrow = lane >> 2;
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);       // swizzle on the source side
    long a_off[APW];                                     // element offset of the row's pixel for ky*dh - ph = 0 (center row), or -1
    int a_y[APW];                                        // its y coordinate
#pragma unroll
    for (int it = 0; it < APW; ++it) {
        const int j = wave + it * NW;
        const int grp = j % GA;
        const long q = (long)bm0 - d.pw + grp * 16 + lrow;     // flattened input pixel of LDS row grp*16 + lrow for the center row
        if (j < NPA && q >= 0 && q < NHW) {
            const long rem = q % HW;
            a_y[it] = (int)(rem / W);
            a_off[it] = q;
        } else {
            a_y[it] = 0; a_off[it] = -1;
        }
    }
    // weights: interleaved rows [Kpad/32][hi 32 | lo 32]; piece (plane, group) of B: row bn0 + grp*16 + lrow
    const __bf16* wbase[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int piece = wave + it * NW, plane = piece / GB, grp = piece % GB;
        const int brow = bn0 + grp * 16 + lrow;          // rows past Npad: zeros
        wbase[it] = brow < d.Npad ? reinterpret_cast<const __bf16*>(p.w) + ((long)g * d.Npad + brow) * (2 * d.Kpad) + plane * 32 + kc * 8 : nullptr;
    }
    const unsigned long zpi = reinterpret_cast<unsigned long>(p.zero_page);
    auto sel = [&](const __bf16* ptr, bool ok) {
        const unsigned long a = reinterpret_cast<unsigned long>(ptr);
        return reinterpret_cast<const void*>(zpi + ((a - zpi) & (ok ? ~0ul : 0ul)));
    };
    // the A rows of super-step (ky, chunk) -> A buffer `ab`; `it`: which of this wavefront's pieces
    auto issue_A = [&](int it, int ky, int chunk, int ab) {
        const int j = wave + it * NW;
        if (j >= NPA) return;                             // (wave-uniform)
        const int plane = j / GA, grp = j % GA;
        const int ci = chunk * 32;
        const bool in0 = ci < Cg0;
        const __bf16* src = in0 ? x0 : x1;
        const int ld = in0 ? ld0 : ld1;
        // planes: channel c of a pixel at element c, the lo plane ps further; interleaved (in_split = 2): (c / 32) * 64 + c % 32, lo 32 further
        const int cch = (in0 ? chb0 : chb1) + ci;                   // first channel of the chunk (a multiple of 32 when interleaved)
        const long eo = il ? 2 * (long)cch + kc * 8 + plane * 32 : (long)cch + kc * 8 + (plane ? (in0 ? ps0 : ps1) : 0);
        const int dy = ky * d.dh - d.ph;
        const bool ok = a_off[it] >= 0 && (unsigned)(a_y[it] + dy) < (unsigned)H;
        const __bf16* ptr = src + (a_off[it] + (long)dy * W) * ld + eo;
        glds16(sel(ptr, ok), Abuf + ab * A_BYTES + plane * AR * 64 + grp * 1024);
    };
    // the B tile of step (ky, chunk, kx) -> B stage `bs`
    auto issue_B = [&](int ky, int chunk, int kx, int bs) {
        const long kstep = ((long)(ky * kw + kx) * p.Cg + chunk * 32) / 32;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int piece = wave + it * NW, plane = piece / GB, grp = piece % GB;
            const bool bok = BN <= 128 || wbase[it] != nullptr;
            glds16(sel(wbase[it] + kstep * 64, bok), Bbuf + bs * B_BYTES + plane * BN * 64 + grp * 1024);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    // x coordinate of this lane's output pixels (one per 32-row block): the taps that leave the image row read the zero row
    int oxv[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) oxv[i] = (bm0 + wm * WTM + i * 32 + l31) % W;

    // ---- prologue: A rows of (0, 0) and the B tile of step 0
    const int nss = kh * nchunk, nsteps = nss * kw;
#pragma unroll
    for (int it = 0; it < APW; ++it) issue_A(it, 0, 0, 0);
    issue_B(0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int ky = 0, chunk = 0, kx = 0, ss = 0;                // this step (wave-uniform scalar state)
    int nky = 0, nchk = 0;                                // the next super-step
    if (nss > 1) { nchk = 1; if (nchk == nchunk) { nchk = 0; nky = 1; } }
    for (int step = 0; step < nsteps; ++step) {
        // ---- requests at the top of the step: the next step's B tile, and this step's share of the next super-step's A rows
        {
            int bkx = kx + 1, bch = chunk, bky = ky;
            if (bkx == kw) { bkx = 0; bch = nchk; bky = nky; }
            if (step + 1 < nsteps) issue_B(bky, bch, bkx, (step + 1) & 1);
            if (ss + 1 < nss) {
                // this wavefront's APW pieces spread over the kw steps of the super-step: piece `it` goes out in step it mod kw (APW < 2 kw)
#pragma unroll
                for (int it = 0; it < APW; ++it) {
                    const int b = it >= kw ? it - kw : it;
                    if (b == kx) issue_A(it, nky, nchk, (ss + 1) & 1);
                }
            }
        }
        // ---- fragments of this step: A rows shifted by kx*dw (zero row where the tap leaves the image row), B stage step & 1
        bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
        {
            const char* Ab = Abuf + (ss & 1) * A_BYTES;
            const char* Bb = Bbuf + (step & 1) * B_BYTES;
            const int sh = kx * dwx;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int R = wm * WTM + i * 32 + l31 + sh;
                const bool xin = (unsigned)(oxv[i] + sh - d.pw) < (unsigned)W;
                const int rs = (R >> 2) & 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int so = ((ks * 2 + lh) ^ rs) * 16;
                    const char* a_hi = xin ? Ab + R * 64 + so : zrow;
                    const char* a_lo = xin ? Ab + AR * 64 + R * 64 + so : zrow;
                    ah[ks][i] = *reinterpret_cast<const bf16x8*>(a_hi);
                    al[ks][i] = *reinterpret_cast<const bf16x8*>(a_lo);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int so = swz(l31, ks * 2 + lh);    // B rows: wave-tile base (multiple of 32) + l31
                const __bf16* Bhi = reinterpret_cast<const __bf16*>(Bb) + (wn * WTN + l31) * LDB + so;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[ks][j] = *reinterpret_cast<const bf16x8*>(Bhi + j * 32 * LDB);
                    bl[ks][j] = *reinterpret_cast<const bf16x8*>(Bhi + BN * LDB + j * 32 * LDB);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);               // keep the fragment reads ahead of the MFMAs
        // same products as conv_split.hip (lo*hi, hi*lo, hi*hi per k-half)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bh[ks][j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- advance (ky, chunk, kx)
        if (++kx == kw) {
            kx = 0; chunk = nchk; ky = nky; ++ss;
            if (++nchk == nchunk) { nchk = 0; ++nky; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN>(p, acc, smem, bm0, bn0, g);
}

template <int BM, int BN, int WM, int WN, int MINW>
int launch(const ConvP& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t smem = (size_t)2 * (2 * (BM + HALO) * 64) + (size_t)2 * (2 * BN * 64) + 64;
    static_assert(smem <= 160 * 1024, "LDS buffers do not fit");
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_taps_kernel<BM, BN, WM, WN, MINW>), (int)smem, lds_set, "conv_taps")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_taps_kernel<BM, BN, WM, WN, MINW>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_taps");
}

}  // namespace

// Geometry this kernel serves (decided by the layer alone, never by tuning): bf16x3 on split inputs (planes or interleaved) with interleaved weights,
// stride 1, no upsample, zero padding, output map = input map ("same"), 3 <= kw, (kw - 1) * dw <= 16, Cin/groups a multiple of 32 per source.
bool fgt_conv_taps_eligible(const ConvP& p) {
    const fgt_conv_desc& d = p.d;
    return d.precision == FGT_PREC_BF16X3 && (d.in_split == 1 || d.in_split == 2) && d.w_il == 1 && d.sh == 1 && d.sw == 1 && !d.upsample && d.pad_mode == 0 &&
           d.in_relu == 0 && d.Ho == d.H && d.Wo == d.W && d.kw >= 3 && (d.kw - 1) * d.dw <= HALO && p.Cg0 % 32 == 0 && p.Cg1 % 32 == 0 &&
           d.Kpad == p.K && p.Cout_g > 4;
}

int fgt_conv_taps_launch(int tile, const ConvP& p, hipStream_t s) {
    switch (tile) {
        case FGT_TILE_128x128x8: return launch<128, 128, 2, 4, 4>(p, s);
        case FGT_TILE_128x128: return launch<128, 128, 2, 2, 2>(p, s);
        case FGT_TILE_128x64: return launch<128, 64, 2, 2, 2>(p, s);
        case FGT_TILE_64x64: return launch<64, 64, 2, 2, 2>(p, s);
        default: fgt_set_error("fgt_conv2d: tile %d is not built for the tap-reusing kernel", tile + FGT_TILE_TAPS); return FGT_EINVAL;
    }
}
