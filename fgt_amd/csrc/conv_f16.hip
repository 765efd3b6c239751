// fp16 implicit-GEMM convolution / GEMM on operands that were rounded ONCE by their producer, fed by LDS-DMA (gfx950).  FGT_PREC_F16.
//
// The third arithmetic mode of fgt_conv2d.  bf16x3 (conv_split.hip) keeps 16 significant bits of every operand and pays three MFMAs and
// two bf16 planes of traffic per product; its K loop is bound by the rate at which the memory hierarchy delivers LDS-DMA instructions to
// a CU and by the chip's power-limited matrix clock.  Here an activation is ONE fp16 plane (11 significant bits,
// h = f16_rne(x): half the bytes of fp32 in HBM, in the LDS-DMA stream and in the fragment reads) and a product is ONE
// v_mfma_f32_32x32x16_f16 with fp32 accumulation.  End to end the FGT forward stays 5-10x inside the 1e-3 bar of the north star
// (tests/test_f16_gpu.py; bf16 operands — 8 bits — sit AT the bar: BASELINE.md §3).
//
// Data movement is conv_split.hip's: every wavefront copies 16-row x 64-byte pieces of the im2col (A) and weight (B) tiles global -> LDS
// with global_load_lds_dwordx4, swizzle applied on the source side, out-of-image taps and the K tail read the zero page, one DMA
// instruction per piece whatever the predicates.  The difference is what the second half of a stage holds: not the lo plane of the same
// 32 channels but the NEXT 32 channels — a K-step is 64 channels, so a pixel's 128 bytes are one full cache line (the planes layout
// reads two half-used lines), a stage has the same bytes and the same DMA instruction count as a bf16x3 stage, and there are half as
// many barriers per unit of K.  LDS image of a stage: [A k0-31 | A k32-63 | B k0-31 | B k32-63], rows of 32 fp16 (64 bytes), the four
// 16-byte slots of row r XOR-swizzled with (r >> 2) & 3 (conflict-free ds_read_b128, as conv_tile.h).
// The two k-halves of a lane's chunk column can sit in different taps (Cin/groups = 40 in the FFN's 7x7 conv): each half walks the
// (tap, source, channel) sequence on its own.
//
// Schedules: the plain double buffer and early stage release (EA: two tiles in flight on two stages), as conv_split.hip.
#include "conv_tile.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate (6 bits on gfx9)");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int BK2 = 64;   // channels per K-step (two 32-channel halves)

template <int BM, int BN, int WM, int WN, int MINW, int EA>
__global__ void __launch_bounds__(WM* WN * 64, MINW) conv_f16_kernel(const ConvP p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int STAGE = (BM + BN) * LDB;              // floats per stage = (BM + BN) rows x 2 halves x 64 bytes
    constexpr int GA = BM / 16, GB = BN / 16;           // 16-row DMA groups per half
    constexpr int A_IT = GA / NW;                       // A groups per wavefront (both halves of the same rows)
    constexpr int B_IT = 2 * GB / NW;                   // B (group, half) pieces per wavefront: piece j = wave + it*NW -> half j / GB, group j % GB
    constexpr int DPT = 2 * A_IT + B_IT;                // DMA instructions per tile and wavefront
    static_assert(GA % NW == 0 && A_IT >= 1 && (2 * GB) % NW == 0 && B_IT >= 1 && TM >= 1 && TN >= 1, "tile / wavefront geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    const _Float16* const x0 = reinterpret_cast<const _Float16*>(p.x0);
    const _Float16* const x1 = reinterpret_cast<const _Float16*>(p.x1);
    // (copied out of the kernel-argument struct: see conv_split.hip)
    const int ld0 = d.ld0, ld1 = d.ld1;
    const int chb0 = d.off0 + g * p.Cg0, chb1 = d.off1 + g * p.Cg1 - p.Cg0;
    const int Cg0 = p.Cg0, Cg = p.Cg;

    // ---- this lane's DMA rows: row (lane >> 2) of each of its 16-row groups, k-chunk kc of both halves of every K-step
    const int lrow = lane >> 2;
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);      // swizzle on the source side (all groups start at multiples of 16 rows)
    int a_iy0[A_IT], a_ix0[A_IT], a_nb[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = bm0 + (wave + it * NW) * 16 + lrow;
        if (m < p.M) {
            const int n_img = m / p.HoWo, rem = m - n_img * p.HoWo;
            const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
            a_iy0[it] = oy * d.sh - d.ph;
            a_ix0[it] = ox * d.sw - d.pw;
            a_nb[it] = n_img * d.H * d.W;
        } else {
            a_iy0[it] = 0; a_ix0[it] = 0; a_nb[it] = -1;
        }
    }
    // per-half position in the K sequence k = (ky*kw + kx)*Cg + ci and the per-row gather bases of its current (tap, source)
    int k_cur[2], ci[2], ky[2], kx[2], seg_end[2];
    unsigned a_okmask[2];
    const _Float16* a_base[2][A_IT];
    auto retap = [&](int h) {
        const bool in0 = ci[h] < Cg0;
        const _Float16* src = in0 ? x0 : x1;
        const int ld = in0 ? ld0 : ld1;
        const int chb = in0 ? chb0 : chb1;               // channel = chb + ci
        seg_end[h] = in0 ? Cg0 : Cg;
        const int dy = ky[h] * d.dh, dx = kx[h] * d.dw;
        const int ush = d.upsample ? 1 : 0;
        const bool rep = d.pad_mode != 0;
        unsigned okm = 0;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            int iy = a_iy0[it] + dy, ix = a_ix0[it] + dx;
            const int cy = min(max(iy, 0), p.Hin - 1), cx = min(max(ix, 0), p.Win - 1);
            iy = rep ? cy : iy;
            ix = rep ? cx : ix;
            const bool ok = a_nb[it] >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            okm |= (ok ? 1u : 0u) << it;
            a_base[h][it] = src + ((long)(a_nb[it] + (iy >> ush) * d.W + (ix >> ush)) * ld + chb);
        }
        a_okmask[h] = okm;
    };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        k_cur[h] = h * 32 + kc * 8;
        const int tap = k_cur[h] / p.Cg;
        ci[h] = k_cur[h] - tap * p.Cg;
        ky[h] = tap / d.kw;
        kx[h] = tap - ky[h] * d.kw;
        retap(h);
    }
    auto advance_A = [&](int h) {
        k_cur[h] += BK2;
        ci[h] += BK2;
        if (ci[h] >= seg_end[h]) {
            while (ci[h] >= Cg) {
                ci[h] -= Cg;
                if (++kx[h] == d.kw) { kx[h] = 0; ++ky[h]; }
            }
            retap(h);
        }
    };

    // weights: [groups][Npad][Kpad] fp16, Kpad % 64 == 0; a K-step's 64 values of a row are one 128-byte line
    const _Float16* wrow[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int piece = wave + it * NW, half = piece / GB, grp = piece % GB;
        const int brow = bn0 + grp * 16 + lrow;          // rows past Npad (tiles wider than the 128-row padding): zeros
        wrow[it] = brow < d.Npad ? reinterpret_cast<const _Float16*>(p.w) + ((long)g * d.Npad + brow) * d.Kpad + half * 32 + kc * 8 : nullptr;
    }

    char* const lds = reinterpret_cast<char*>(smem);
    constexpr int STAGE_B = STAGE * 4;
    // one DMA instruction per piece whatever the predicates: the zero-page select is arithmetic on the address
    const unsigned long zpi = reinterpret_cast<unsigned long>(p.zero_page);
    auto sel = [&](const _Float16* ptr, bool ok) {
        const unsigned long a = reinterpret_cast<unsigned long>(ptr);
        return reinterpret_cast<const void*>(zpi + ((a - zpi) & (ok ? ~0ul : 0ul)));
    };
    auto issue_tile = [&](int slot) {
        char* st = lds + slot * STAGE_B;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool kval = k_cur[h] < p.K;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const bool ok = kval && ((a_okmask[h] >> it) & 1u);
                glds16(sel(a_base[h][it] + ci[h], ok), st + h * BM * 64 + (wave + it * NW) * 1024);
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int piece = wave + it * NW, half = piece / GB, grp = piece % GB;    // wave-uniform
            const bool bok = BN <= 128 || wrow[it] != nullptr;
            glds16(sel(wrow[it], bok), st + 2 * BM * 64 + half * BN * 64 + grp * 1024);
            if (bok) wrow[it] += BK2;
        }
        advance_A(0);
        advance_A(1);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;

    // ---- prologue: tile 0 landed (EA: tiles 0 and 1 in flight)
    constexpr int AHEAD = EA ? 2 : 1;
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
        if (t < p.nk) issue_tile(t);
    if (p.nk >= AHEAD) wait_vmcnt<DPT * (AHEAD - 1)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int slot = 0;

    // fragments of one k-half (32 channels = two MFMA k-steps of 16): operand rows = wave-tile base (multiple of 32) + l31
    auto read_half = [&](int h, f16x8 (&a)[2][TM], f16x8 (&b)[2][TN]) {
        const _Float16* base = reinterpret_cast<const _Float16*>(smem + slot * STAGE);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = swz(l31, ks * 2 + lh);
            const _Float16* A = base + h * BM * LDB + (wm * WTM + l31) * LDB + so;
            const _Float16* B = base + 2 * BM * LDB + h * BN * LDB + (wn * WTN + l31) * LDB + so;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[ks][i] = *reinterpret_cast<const f16x8*>(A + i * 32 * LDB);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[ks][j] = *reinterpret_cast<const f16x8*>(B + j * 32 * LDB);
        }
    };
    auto mfmas = [&](f16x8 (&a)[2][TM], f16x8 (&b)[2][TN]) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
    };

    if constexpr (EA) {
        //   step kt: read tile kt (stage kt&1) | lgkmcnt(0) | barrier | issue tile kt+2 -> stage kt&1 | MFMAs | vmcnt(DPT): tile kt+1 landed,
        //            tile kt+2 may fly | barrier
        for (int kt = 0; kt < p.nk; ++kt) {
            f16x8 a0[2][TM], b0[2][TN], a1[2][TM], b1[2][TN];
            read_half(0, a0, b0);
            read_half(1, a1, b1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                   // every wavefront holds its fragments of tile kt: the stage can be refilled
            const bool more = kt + 2 < p.nk;
            if (more) issue_tile(slot);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a0, b0);
            mfmas(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            if (more) wait_vmcnt<DPT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
    } else {
        for (int kt = 0; kt < p.nk; ++kt) {
            if (kt + 1 < p.nk) issue_tile(slot ^ 1);
            f16x8 a0[2][TM], b0[2][TN], a1[2][TM], b1[2][TN];
            read_half(0, a0, b0);
            read_half(1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);              // keep all fragment reads of the step ahead of its MFMAs
            mfmas(a0, b0);
            mfmas(a1, b1);
            // the wait + barrier stay BEHIND the MFMAs (the DMA latency runs underneath this wavefront's matrix work)
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN, false, false>(conv_epilogue_args(p), acc, smem, bm0, bn0, g);      // (no bias maps in the f16 mode: fgt_conv2d rejects them)
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// "Wide" LDS image (tile codes + 100): a stage row is the pixel's whole 128-byte line [64 channels] instead of two 64-byte half rows in two
// sub-tiles, and an LDS-DMA instruction copies 8 rows x 128 bytes (8 FULL cache lines, 8 lanes per line) instead of 16 rows x 64 bytes
// (16 half lines: the other half of every line is fetched by another instruction).  tools/micro/lds_dma_rate.hip measured the global -> LDS
// rate from L2 at 37 B/clk/CU for 64-byte segments and 70 for 128-byte segments; the K loop is bound by exactly that stream.
// The eight 16-byte slots of row r are XOR-swizzled with (r >> 1) & 7 (two rows per 256-byte bank line: the 16 rows of a ds_read_b128 lane
// group cover all 16 slots), applied on the source side as before.  A lane owns ONE chunk column of the 64-channel step (one tap state
// instead of two); a wavefront's pieces all have the parity of its index (even wavefront counts), so the column is the same for all of them.
// Same MFMA sequence per accumulator (k ascending): bit-identical to the narrow layout.
template <int BM, int BN, int WM, int WN, int MINW, int EA>
__global__ void __launch_bounds__(WM* WN * 64, MINW) conv_f16w_kernel(const ConvP p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int STAGE = (BM + BN) * LDB;              // floats per stage = (BM + BN) rows x 128 bytes
    constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;   // 8-row DMA pieces per wavefront and tile
    constexpr int DPT = PA + PB;
    static_assert(NW % 2 == 0 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0 && PA >= 1 && PB >= 1 && TM >= 1 && TN >= 1, "tile / wavefront geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    const _Float16* const x0 = reinterpret_cast<const _Float16*>(p.x0);
    const _Float16* const x1 = reinterpret_cast<const _Float16*>(p.x1);
    const int ld0 = d.ld0, ld1 = d.ld1;
    const int chb0 = d.off0 + g * p.Cg0, chb1 = d.off1 + g * p.Cg1 - p.Cg0;
    const int Cg0 = p.Cg0, Cg = p.Cg;

    // ---- this lane's DMA rows: row (lane >> 3) of each of its 8-row pieces, chunk column kc (8 channels) of every 64-channel K-step
    const int lrow = lane >> 3;
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);     // slot (lane & 7) of row r holds chunk slot ^ ((r >> 1) & 7); piece parity = wave parity
    int a_iy0[PA], a_ix0[PA], a_nb[PA];
#pragma unroll
    for (int it = 0; it < PA; ++it) {
        const int m = bm0 + (wave + it * NW) * 8 + lrow;
        if (m < p.M) {
            const int n_img = m / p.HoWo, rem = m - n_img * p.HoWo;
            const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
            a_iy0[it] = oy * d.sh - d.ph;
            a_ix0[it] = ox * d.sw - d.pw;
            a_nb[it] = n_img * d.H * d.W;
        } else {
            a_iy0[it] = 0; a_ix0[it] = 0; a_nb[it] = -1;
        }
    }
    int k_cur = kc * 8;
    int ky, kx, ci, seg_end = 0;
    {
        const int tap = k_cur / p.Cg;
        ci = k_cur - tap * p.Cg;
        ky = tap / d.kw;
        kx = tap - ky * d.kw;
    }
    unsigned a_okmask = 0;
    const _Float16* a_base[PA];
    auto retap = [&]() {
        const bool in0 = ci < Cg0;
        const _Float16* src = in0 ? x0 : x1;
        const int ld = in0 ? ld0 : ld1;
        const int chb = in0 ? chb0 : chb1;
        seg_end = in0 ? Cg0 : Cg;
        const int dy = ky * d.dh, dx = kx * d.dw;
        const int ush = d.upsample ? 1 : 0;
        const bool rep = d.pad_mode != 0;
        a_okmask = 0;
#pragma unroll
        for (int it = 0; it < PA; ++it) {
            int iy = a_iy0[it] + dy, ix = a_ix0[it] + dx;
            const int cy = min(max(iy, 0), p.Hin - 1), cx = min(max(ix, 0), p.Win - 1);
            iy = rep ? cy : iy;
            ix = rep ? cx : ix;
            const bool ok = a_nb[it] >= 0 && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            a_okmask |= (ok ? 1u : 0u) << it;
            a_base[it] = src + ((long)(a_nb[it] + (iy >> ush) * d.W + (ix >> ush)) * ld + chb);
        }
    };
    retap();

    const _Float16* wrow[PB];
#pragma unroll
    for (int it = 0; it < PB; ++it) {
        const int brow = bn0 + (wave + it * NW) * 8 + lrow;          // rows past Npad (tiles wider than the 128-row padding): zeros
        wrow[it] = brow < d.Npad ? reinterpret_cast<const _Float16*>(p.w) + ((long)g * d.Npad + brow) * d.Kpad + kc * 8 : nullptr;
    }

    char* const lds = reinterpret_cast<char*>(smem);
    constexpr int STAGE_B = STAGE * 4;
    const unsigned long zpi = reinterpret_cast<unsigned long>(p.zero_page);
    auto sel = [&](const _Float16* ptr, bool ok) {
        const unsigned long a = reinterpret_cast<unsigned long>(ptr);
        return reinterpret_cast<const void*>(zpi + ((a - zpi) & (ok ? ~0ul : 0ul)));
    };
    auto issue_tile = [&](int slot) {
        char* st = lds + slot * STAGE_B;
        const bool kval = k_cur < p.K;
#pragma unroll
        for (int it = 0; it < PA; ++it) {
            const bool ok = kval && ((a_okmask >> it) & 1u);
            glds16(sel(a_base[it] + ci, ok), st + (wave + it * NW) * 1024);
        }
#pragma unroll
        for (int it = 0; it < PB; ++it) {
            const bool bok = BN <= 128 || wrow[it] != nullptr;
            glds16(sel(wrow[it], bok), st + BM * 128 + (wave + it * NW) * 1024);
            if (bok) wrow[it] += BK2;
        }
        k_cur += BK2;
        ci += BK2;
        if (ci >= seg_end) {
            while (ci >= Cg) {
                ci -= Cg;
                if (++kx == d.kw) { kx = 0; ++ky; }
            }
            retap();
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
    constexpr int AHEAD = EA ? 2 : 1;
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
        if (t < p.nk) issue_tile(t);
    if (p.nk >= AHEAD) wait_vmcnt<DPT * (AHEAD - 1)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int slot = 0;

    // fragments of the whole step: four MFMA k-steps of 16 channels; operand rows = wave-tile base (multiple of 32) + l31
    const int rsw = (l31 >> 1) & 7;
    auto read_frags = [&](f16x8 (&a)[4][TM], f16x8 (&b)[4][TN]) {
        const char* base = reinterpret_cast<const char*>(smem + slot * STAGE);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int so = ((ks * 2 + lh) ^ rsw) * 16;
            const char* A = base + (wm * WTM + l31) * 128 + so;
            const char* B = base + BM * 128 + (wn * WTN + l31) * 128 + so;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[ks][i] = *reinterpret_cast<const f16x8*>(A + i * 32 * 128);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[ks][j] = *reinterpret_cast<const f16x8*>(B + j * 32 * 128);
        }
    };
    auto mfmas = [&](f16x8 (&a)[4][TM], f16x8 (&b)[4][TN]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
    };

    if constexpr (EA) {
        for (int kt = 0; kt < p.nk; ++kt) {
            f16x8 a[4][TM], b[4][TN];
            read_frags(a, b);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                   // every wavefront holds its fragments of tile kt: the stage can be refilled
            const bool more = kt + 2 < p.nk;
            if (more) issue_tile(slot);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a, b);
            __builtin_amdgcn_sched_barrier(0);
            if (more) wait_vmcnt<DPT>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
    } else {
        for (int kt = 0; kt < p.nk; ++kt) {
            if (kt + 1 < p.nk) issue_tile(slot ^ 1);
            f16x8 a[4][TM], b[4][TN];
            read_frags(a, b);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a, b);
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN, false, false>(conv_epilogue_args(p), acc, smem, bm0, bn0, g);      // (no bias maps in the f16 mode: fgt_conv2d rejects them)
}

template <int BM, int BN, int WM, int WN, int MINW = 2, int EA = 0>
int launch_w(const ConvP& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t smem = (size_t)2 * (BM + BN) * LDB * sizeof(float);
    static_assert(smem <= 160 * 1024, "LDS stages do not fit");
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_f16w_kernel<BM, BN, WM, WN, MINW, EA>), (int)smem, lds_set, "conv_f16w")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_f16w_kernel<BM, BN, WM, WN, MINW, EA>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_f16w");
}

template <int BM, int BN, int WM, int WN, int MINW = 2, int EA = 0>
int launch(const ConvP& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t smem = (size_t)2 * (BM + BN) * LDB * sizeof(float);
    static_assert(smem <= 160 * 1024, "LDS stages do not fit");
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_f16_kernel<BM, BN, WM, WN, MINW, EA>), (int)smem, lds_set, "conv_f16")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_f16_kernel<BM, BN, WM, WN, MINW, EA>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_f16");
}

}  // namespace

// called by fgt_conv2d (conv_igemm.hip) for desc.in_split == 3
int fgt_conv_f16_launch(int tile, const ConvP& p, hipStream_t s) {
    switch (tile) {
        case FGT_TILE_128x128: return launch<128, 128, 2, 2>(p, s);
        case FGT_TILE_128x64: return launch<128, 64, 2, 2>(p, s);
        case FGT_TILE_64x64: return launch<64, 64, 2, 2>(p, s);
        case FGT_TILE_128x32: return launch<128, 32, 4, 1>(p, s);
        case FGT_TILE_256x128: return launch<256, 128, 4, 2>(p, s);
        case FGT_TILE_128x128x8: return launch<128, 128, 2, 4, 4>(p, s);
        case FGT_TILE_256x128x16: return launch<256, 128, 4, 4, 4>(p, s);
        case FGT_TILE_256x64x8: return launch<256, 64, 4, 2, 2>(p, s);
        case FGT_TILE_128x128_EA: return launch<128, 128, 2, 2, 2, 1>(p, s);
        case FGT_TILE_128x64_EA: return launch<128, 64, 2, 2, 2, 1>(p, s);
        case FGT_TILE_64x64_EA: return launch<64, 64, 2, 2, 2, 1>(p, s);
        case FGT_TILE_256x128_EA: return launch<256, 128, 4, 2, 2, 1>(p, s);
        case FGT_TILE_128x128x8_EA: return launch<128, 128, 2, 4, 4, 1>(p, s);
        case FGT_TILE_256x128x16_EA: return launch<256, 128, 4, 4, 4, 1>(p, s);
        case FGT_TILE_256x64x8_EA: return launch<256, 64, 4, 2, 2, 1>(p, s);
        // + 100: the wide LDS image (128-byte rows, full-line LDS-DMA pieces); bit-identical results
        case 100 + FGT_TILE_128x128: return launch_w<128, 128, 2, 2>(p, s);
        case 100 + FGT_TILE_128x64: return launch_w<128, 64, 2, 2>(p, s);
        case 100 + FGT_TILE_64x64: return launch_w<64, 64, 2, 2>(p, s);
        case 100 + FGT_TILE_128x32: return launch_w<128, 32, 4, 1>(p, s);
        case 100 + FGT_TILE_256x128: return launch_w<256, 128, 4, 2>(p, s);
        case 100 + FGT_TILE_128x128x8: return launch_w<128, 128, 2, 4, 4>(p, s);
        case 100 + FGT_TILE_256x128x16: return launch_w<256, 128, 4, 4, 4>(p, s);
        case 100 + FGT_TILE_256x64x8: return launch_w<256, 64, 4, 2, 2>(p, s);
        case 100 + FGT_TILE_128x128_EA: return launch_w<128, 128, 2, 2, 2, 1>(p, s);
        case 100 + FGT_TILE_128x64_EA: return launch_w<128, 64, 2, 2, 2, 1>(p, s);
        case 100 + FGT_TILE_64x64_EA: return launch_w<64, 64, 2, 2, 2, 1>(p, s);
        case 100 + FGT_TILE_256x128_EA: return launch_w<256, 128, 4, 2, 2, 1>(p, s);
        case 100 + FGT_TILE_128x128x8_EA: return launch_w<128, 128, 2, 4, 4, 1>(p, s);
        case 100 + FGT_TILE_256x128x16_EA: return launch_w<256, 128, 4, 4, 4, 1>(p, s);
        case 100 + FGT_TILE_256x64x8_EA: return launch_w<256, 64, 4, 2, 2, 1>(p, s);
        default: fgt_set_error("fgt_conv2d: tile %d is not built for FGT_PREC_F16", tile); return FGT_EINVAL;
    }
}
