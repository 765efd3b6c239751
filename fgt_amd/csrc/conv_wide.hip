// bf16x3 implicit-GEMM convolution / GEMM on INTERLEAVED pre-split operands with full-line LDS-DMA pieces (gfx950).
//
// Same arithmetic as conv_split.hip (hi/lo bf16 operands, per accumulator and k-half the products lo*hi, hi*lo, hi*hi in that order,
// fp32 accumulate: results are bit-identical to it and to conv_igemm.hip's bf16x3 path), different bytes on the wire.  The K loop of
// conv_split.hip is bound by the rate at which the memory hierarchy delivers 1-KB LDS-DMA instructions to a CU: its
// pieces are 16 rows x 64 bytes — 16 HALF cache lines, the other half of every line is fetched by another instruction of another plane.
// tools/micro/lds_dma_rate.hip measured the global -> LDS rate from L2 at 37 B/clk/CU for 64-byte segments and 70 for 128-byte ones.
// Here an activation row is stored interleaved per 32 channels, [hi 32 | lo 32] bf16 = ONE 128-byte line per pixel and K-step
// (fgt_conv_desc.in_split = 2; the weights have had that form since round 1, w_il = 1), an LDS row IS that line, and an LDS-DMA
// instruction copies 8 rows x 128 bytes = 8 FULL lines (8 lanes per line).
//
// LDS image of a stage: [A: BM rows x 128 B | B: BN rows x 128 B]; a row holds the eight 16-byte chunks (hi k0-7, hi k8-15, hi k16-23,
// hi k24-31, lo k0-7, ...) of one pixel / output channel, chunk c of row r in slot c ^ ((r >> 1) & 7): two rows share a 256-byte bank
// line, so the 16 rows of a ds_read_b128 lane group ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of every 32) cover all 16 slots of it
// (conflict free; the lo chunk of a fragment is its hi chunk's slot ^ 4 = 64 bytes further in the same row).  The swizzle is applied
// on the SOURCE side of the DMA (the destination is wave-uniform base + 16 * lane): lane l of piece P fetches chunk
// (l & 7) ^ ((4 * (P & 1) + (l >> 4)) & 7) of row l >> 3; a wavefront's pieces all have its parity (even wavefront counts), so a
// lane's chunk column is ONE value.  With Cin/groups a multiple of 32 per source (required by in_split = 2) a K-step never straddles a
// tap or a source: the (tap, source, channel) walk is wave-uniform scalar code, a lane only keeps one row pointer per piece.
//
// Schedules (SCHED): 0 = plain double buffer, 1 = early stage release (two tiles in flight on two stages), both as in conv_split.hip;
// 2 = the 8-phase schedule of cdna_hip_programming.md's 256^2 GEMM template on a 256-row tile (8 wavefronts as two staggered
// groups, one half tile staged per phase, counted vmcnt) — with interleaved rows the stage IS that template's [128][64]-bf16 half tile,
// which the planes layout of round 2's P8 tiles (conv_split.hip) could not offer.
#include "conv_tile.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate (6 bits on gfx9)");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int WM, int WN, int MINW, int SCHED>
__global__ void __launch_bounds__(WM* WN * 64, MINW) conv_wide_kernel(const ConvP p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int STAGE = (BM + BN) * LDB;              // floats per stage = (BM + BN) rows x 128 bytes
    constexpr int STAGE_B = STAGE * 4;
    constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;   // 8-row DMA pieces per wavefront and tile
    constexpr int DPT = PA + PB;
    static_assert(NW % 2 == 0 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0 && PA >= 1 && PB >= 1 && TM >= 1 && TN >= 1, "tile / wavefront geometry");
    static_assert(SCHED != 2 || (NW == 8 && (WN == 2 || WN == 4) && BM == 256 && TM % 2 == 0 && TN % 2 == 0 && PA == 4 && (PB == 2 || PB == 4)),
                  "8-phase schedule: 256-row tile on 8 wavefronts (2 x 4 or 4 x 2: wavefronts 0-3 own A rows 0-127)");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    // ABI 8, batched GEMM (desc.gb_o != 0): group g has its own rows, its own weight rows and its own output block — the channel offsets of the
    // grouped-convolution form (g * Cin/groups, g * Npad weight rows, g * Cout/groups output channels) are replaced by three element strides
    const bool gb = d.gb_o != 0;
    const __bf16* const x0 = reinterpret_cast<const __bf16*>(p.x0) + (gb ? (long)g * d.gb_x0 : 0l);
    const __bf16* const x1 = reinterpret_cast<const __bf16*>(p.x1);
    // (copied out of the kernel-argument struct: see conv_split.hip)
    const int ld0 = d.ld0, ld1 = d.ld1;
    const int Cg0 = p.Cg0, Cg = p.Cg;
    // element offset of logical channel c of a pixel's interleaved row: (c / 32) * 64 + c % 32 (hi), + 32 (lo); a K-step starts at a
    // multiple of 32, so chunk column kc (0-3 hi, 4-7 lo) of the step that starts at channel c sits at 2 * c + kc * 8
    const int chb0 = 2 * (d.off0 + (gb ? 0 : g * p.Cg0)), chb1 = 2 * (d.off1 + g * p.Cg1 - p.Cg0);

    // ---- this lane's DMA rows: row (lane >> 3) of each of its 8-row pieces, chunk column kc of every K-step
    const int lrow = lane >> 3;
    const int kc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
    int a_iy0[PA], a_ix0[PA], a_nb[PA];
    // ABI 9 (desc.ps_phase_pad): this workgroup's columns lie in one sub-pixel (a, b) of the x2 output: padding (1 - a, 1 - b) (see conv_split.hip)
    int ph_e = d.ph, pw_e = d.pw;
    if (d.ps_phase_pad) {
        const int q = bn0 / d.ps_c;
        ph_e -= q >> 1;
        pw_e -= q & 1;
    }
#pragma unroll
    for (int it = 0; it < PA; ++it) {
        const int m = bm0 + (wave + it * NW) * 8 + lrow;
        if (m < p.M) {
            const int n_img = m / p.HoWo, rem = m - n_img * p.HoWo;
            const int oy = rem / d.Wo, ox = rem - oy * d.Wo;
            a_iy0[it] = oy * d.sh - ph_e;
            a_ix0[it] = ox * d.sw - pw_e;
            a_nb[it] = n_img * d.H * d.W;
        } else {
            a_iy0[it] = 0; a_ix0[it] = 0; a_nb[it] = -1;
        }
    }
    // wave-uniform position in the K sequence k = (ky*kw + kx)*Cg + ci (ci a multiple of 32) and the per-row bases of its (tap, source)
    int ci = 0, ky = 0, kx = 0, seg_end = 0;
    unsigned a_okmask = 0;
    const __bf16* a_base[PA];
    auto retap = [&]() {
        const bool in0 = ci < Cg0;
        const __bf16* src = in0 ? x0 : x1;
        const int ld = in0 ? ld0 : ld1;
        const int chb = (in0 ? chb0 : chb1) + kc * 8;    // element offset of this lane's chunk for ci = 0 of the source
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
    auto advance_A = [&]() {
        ci += BK;
        if (ci >= seg_end) {
            if (ci >= Cg) {
                ci = 0;
                if (++kx == d.kw) { kx = 0; ++ky; }
            }
            retap();
        }
    };

    // weights: [groups][Npad][Kpad/32][hi 32 | lo 32] bf16: a K-step's 64 values of a row are one 128-byte line
    const __bf16* wrow[PB];
#pragma unroll
    for (int it = 0; it < PB; ++it) {
        const int brow = bn0 + (wave + it * NW) * 8 + lrow;          // rows past Npad (tiles wider than the 128-row padding): zeros
        // (batched: a group's weight rows are the Cout/groups rows of its operand tensor — a multiple of 8, so a piece is all in or all out)
        const __bf16* const wg = reinterpret_cast<const __bf16*>(p.w) + (gb ? (long)g * d.gb_w : (long)g * d.Npad * (2 * d.Kpad));
        wrow[it] = brow < (gb ? p.Cout_g : d.Npad) ? wg + (long)brow * (2 * d.Kpad) + kc * 8 : nullptr;
    }

    char* const lds = reinterpret_cast<char*>(smem);
    // one DMA instruction per piece whatever the predicates: the zero-page select is arithmetic on the address
    const unsigned long zpi = reinterpret_cast<unsigned long>(p.zero_page);
    auto sel = [&](const __bf16* ptr, bool ok) {
        const unsigned long a = reinterpret_cast<unsigned long>(ptr);
        return reinterpret_cast<const void*>(zpi + ((a - zpi) & (ok ? ~0ul : 0ul)));
    };
    auto issue_A = [&](int it, int slot) {
        const bool ok = (a_okmask >> it) & 1u;
        glds16(sel(a_base[it] + 2 * ci, ok), lds + slot * STAGE_B + (wave + it * NW) * 1024);
    };
    auto issue_B = [&](int it, int slot) {
        const bool bok = wrow[it] != nullptr;
        glds16(sel(wrow[it], bok), lds + slot * STAGE_B + BM * 128 + (wave + it * NW) * 1024);
        if (bok) wrow[it] += 2 * BK;
    };
    auto issue_tile = [&](int slot) {
#pragma unroll
        for (int it = 0; it < PA; ++it) issue_A(it, slot);
#pragma unroll
        for (int it = 0; it < PB; ++it) issue_B(it, slot);
        advance_A();
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    const int rsw = (l31 >> 1) & 7;                     // operand rows: wave-tile base (multiple of 32) + l31
    int slot = 0;

    if constexpr (SCHED != 2) {
        // ---- prologue: tile 0 landed (early release: tiles 0 and 1 in flight)
        constexpr int AHEAD = SCHED == 1 ? 2 : 1;
#pragma unroll
        for (int t = 0; t < AHEAD; ++t)
            if (t < p.nk) issue_tile(t);
        if (p.nk >= AHEAD) wait_vmcnt<DPT * (AHEAD - 1)>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();

        auto read_frags = [&](bf16x8 (&ah)[2][TM], bf16x8 (&al)[2][TM], bf16x8 (&bh)[2][TN], bf16x8 (&bl)[2][TN]) {
            const char* base = reinterpret_cast<const char*>(smem + slot * STAGE);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int so = ((ks * 2 + lh) ^ rsw) * 16;
                const char* A = base + (wm * WTM + l31) * 128 + so;
                const char* B = base + BM * 128 + (wn * WTN + l31) * 128 + so;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[ks][i] = *reinterpret_cast<const bf16x8*>(A + i * 32 * 128);
                    al[ks][i] = *reinterpret_cast<const bf16x8*>(A + i * 32 * 128 + ((so ^ 64) - so));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[ks][j] = *reinterpret_cast<const bf16x8*>(B + j * 32 * 128);
                    bl[ks][j] = *reinterpret_cast<const bf16x8*>(B + j * 32 * 128 + ((so ^ 64) - so));
                }
            }
        };
        // same product order as conv_igemm.hip / conv_split.hip (lo*hi, hi*lo, hi*hi per k-half): bit-identical accumulators
        auto mfmas = [&](bf16x8 (&ah)[2][TM], bf16x8 (&al)[2][TM], bf16x8 (&bh)[2][TN], bf16x8 (&bl)[2][TN]) {
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
        };

        if constexpr (SCHED == 1) {
            //   step kt: read tile kt (stage kt&1) | lgkmcnt(0) | barrier | issue tile kt+2 -> stage kt&1 | MFMAs | vmcnt(DPT): tile kt+1
            //            landed, tile kt+2 may fly | barrier
            for (int kt = 0; kt < p.nk; ++kt) {
                bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
                read_frags(ah, al, bh, bl);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                   // every wavefront holds its fragments of tile kt: the stage can be refilled
                const bool more = kt + 2 < p.nk;
                if (more) issue_tile(slot);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);            // (the MFMA block ahead of the other wavefronts' address arithmetic: +1...2 %, same-box A/B)
                mfmas(ah, al, bh, bl);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) wait_vmcnt<DPT>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                slot ^= 1;
            }
        } else {
            for (int kt = 0; kt < p.nk; ++kt) {
                if (kt + 1 < p.nk) issue_tile(slot ^ 1);
                bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
                read_frags(ah, al, bh, bl);
                __builtin_amdgcn_sched_barrier(0);              // keep all fragment reads of the step ahead of its MFMAs
                mfmas(ah, al, bh, bl);
                // the wait + barrier stay BEHIND the MFMAs (the DMA latency runs underneath this wavefront's matrix work)
                __builtin_amdgcn_sched_barrier(0);
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                slot ^= 1;
            }
        }
    } else {
        // ---- 8-phase staggered schedule (the phase / hazard analysis is conv_split.hip's P8 block: same half tiles, same readers, same
        // counts — HA0 / HA1 = A rows 0-127 / 128-255 read by group G0 (waves 0-3, wm = 0) / G1 (waves 4-7, wm = 1), HB0 / HB1 = B rows
        // 0-127 / 128-255 read by everyone; a wavefront contributes 2 pieces to every half tile).  Per K tile and wavefront:
        //     L(q): fragment reads of quadrant q + 2 LDS-DMA pieces of tile kt+1 (q = 0: HB0 [or all of B when BN = 128], 1: HB1, 2: HA0,
        //           3: HA1)   | s_barrier |   C(q): (TM/2)*(TN/2)*6 MFMAs under s_setprio 1   | s_barrier |
        // G1 runs ONE barrier behind G0.  Quadrant order (0,0) (0,1) (1,1) (1,0).  Counted waits in FRONT of the barriers that end
        // intervals 8kt+7 and 8kt+8: see conv_split.hip.
        constexpr int HM = TM / 2, HN = TN / 2;
        constexpr int N0 = 2, N3 = 2;
        issue_tile(0);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        const bool g1 = wave >= 4;
        bf16x8 ah[HM][2], al[HM][2], bh[TN][2], bl[TN][2];  // [block][k half]
        auto readA8 = [&](int ih) {
            const char* base = reinterpret_cast<const char*>(smem + slot * STAGE);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int so = ((ks * 2 + lh) ^ rsw) * 16;
                const char* A = base + (wm * WTM + ih * HM * 32 + l31) * 128 + so;
#pragma unroll
                for (int i = 0; i < HM; ++i) {
                    ah[i][ks] = *reinterpret_cast<const bf16x8*>(A + i * 32 * 128);
                    al[i][ks] = *reinterpret_cast<const bf16x8*>(A + i * 32 * 128 + ((so ^ 64) - so));
                }
            }
        };
        auto readB8 = [&](int jh) {
            const char* base = reinterpret_cast<const char*>(smem + slot * STAGE);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int so = ((ks * 2 + lh) ^ rsw) * 16;
                const char* B = base + BM * 128 + (wn * WTN + jh * HN * 32 + l31) * 128 + so;
#pragma unroll
                for (int j = 0; j < HN; ++j) {
                    bh[jh * HN + j][ks] = *reinterpret_cast<const bf16x8*>(B + j * 32 * 128);
                    bl[jh * HN + j][ks] = *reinterpret_cast<const bf16x8*>(B + j * 32 * 128 + ((so ^ 64) - so));
                }
            }
        };
        auto mm8 = [&](int ih, int jh) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < HM; ++i)
#pragma unroll
                    for (int j = 0; j < HN; ++j)
                        acc[ih * HM + i][jh * HN + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i][ks], bh[jh * HN + j][ks], acc[ih * HM + i][jh * HN + j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < HM; ++i)
#pragma unroll
                    for (int j = 0; j < HN; ++j)
                        acc[ih * HM + i][jh * HN + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][ks], bl[jh * HN + j][ks], acc[ih * HM + i][jh * HN + j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < HM; ++i)
#pragma unroll
                    for (int j = 0; j < HN; ++j)
                        acc[ih * HM + i][jh * HN + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i][ks], bh[jh * HN + j][ks], acc[ih * HM + i][jh * HN + j], 0, 0, 0);
            }
        };
        auto stage8 = [&](int q) {
            const int si = slot ^ 1;
            if (q == 0) {
                issue_B(0, si);
                issue_B(1, si);
            } else if (q == 1) {
                if constexpr (PB == 4) { issue_B(2, si); issue_B(3, si); }
            } else if (q == 2) {
                issue_A(0, si); issue_A(1, si);
            } else {
                issue_A(2, si); issue_A(3, si);
                advance_A();
            }
        };
        auto compute8 = [&](int ih, int jh) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mm8(ih, jh);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (g1) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < p.nk; ++kt) {
            const bool st = kt + 1 < p.nk;                  // tile kt+1 exists: stage it during this tile
            // ---- phase 0: quadrant (0,0)
            if (st) stage8(0);
            readA8(0);
            readB8(0);
            __builtin_amdgcn_sched_barrier(0);
            if (!g1) { if (st) wait_vmcnt<N0>(); else wait_vmcnt<0>(); }
            __builtin_amdgcn_s_barrier();
            compute8(0, 0);
            __builtin_amdgcn_s_barrier();
            // ---- phase 1: quadrant (0,1)
            if (st) stage8(1);
            readB8(1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            compute8(0, 1);
            __builtin_amdgcn_s_barrier();
            // ---- phase 2: quadrant (1,1)
            if (st) stage8(2);
            readA8(1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            compute8(1, 1);
            __builtin_amdgcn_s_barrier();
            // ---- phase 3: quadrant (1,0)
            if (st) stage8(3);
            __builtin_amdgcn_sched_barrier(0);
            if (g1) { if (st) wait_vmcnt<N3>(); else wait_vmcnt<0>(); }
            __builtin_amdgcn_s_barrier();
            compute8(1, 0);
            if (g1) wait_vmcnt<0>();
            else if (st) wait_vmcnt<N3>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
        if (!g1) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // (no bias-map / two-headed instances: interleaved-input layers with those operands are tap-routed or run conv_split.hip; the launcher declines)
    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN, false, false, true>(conv_epilogue_args(p), acc, smem, bm0, bn0, gb ? 0 : g, gb ? (long)g * d.gb_o : 0l);
}

template <int BM, int BN, int WM, int WN, int MINW = 2, int SCHED = 0>
int launch(const ConvP& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t smem = (size_t)2 * (BM + BN) * LDB * sizeof(float);
    static_assert(smem <= 160 * 1024, "LDS stages do not fit");
    if (p.d.ps_phase_pad && p.d.ps_c % BN != 0) { fgt_set_error("fgt_conv2d: ps_phase_pad needs a tile whose N width (%d) divides ps_c (%d)", BN, p.d.ps_c); return FGT_EINVAL; }
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_wide_kernel<BM, BN, WM, WN, MINW, SCHED>), (int)smem, lds_set, "conv_wide")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_wide_kernel<BM, BN, WM, WN, MINW, SCHED>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_wide");
}

}  // namespace

// called by fgt_conv2d (conv_igemm.hip) for desc.in_split == 2 with a tile code >= 100 (`tile` = code - 100)
int fgt_conv_wide_launch(int tile, const ConvP& p, hipStream_t s) {
    if (p.d.ld_bias > 0 || p.d.dual_n0 > 0) { fgt_set_error("fgt_conv2d: the wide bf16x3 tiles are built without bias-map / two-headed epilogues"); return FGT_EINVAL; }
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
#ifdef FGT_DIAG      // the 8-phase schedule: measured = the early-release tiles on long-K layers, slower on K = 512 GEMMs; never selected by the autotuner
        case FGT_TILE_256x256_P8: return launch<256, 256, 2, 4, 2, 2>(p, s);       // one workgroup per CU (128 KB of stages)
        case FGT_TILE_256x128_P8: return launch<256, 128, 4, 2, 2, 2>(p, s);
#endif
        default: fgt_set_error("fgt_conv2d: tile %d is not built for the wide (interleaved) bf16x3 kernel", tile + 100); return FGT_EINVAL;
    }
}
