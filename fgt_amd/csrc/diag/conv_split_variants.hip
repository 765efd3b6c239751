// DIAGNOSTIC BUILDS ONLY (fgt_amd.build.build(variant="diag"), -DFGT_DIAG): every schedule variant of the pre-split bf16x3 kernel that was measured
// and NOT adopted (rings, ping-pong, interleaved, 8-phase, loader wavefronts, XY, the s_memtime trace build).  The product kernel is csrc/conv_split.hip.
//
// bf16x3 implicit-GEMM convolution / GEMM with PRE-SPLIT operands, fed by LDS-DMA (gfx950).
//
// Same math as conv_igemm.hip's FGT_PREC_BF16X3 path (hi/lo bf16 operands, three v_mfma_f32_32x32x16_bf16 per product in the
// same order, fp32 accumulate: results are bit-identical), different data movement.  The activations arrive already split
// (desc.in_split: two bf16 planes, written once by their producer), the weights are pre-split at pack time, so a K-step's
// tiles are plain 16-byte copies.  Every wavefront moves them global -> LDS with global_load_lds_dwordx4: no staging
// registers, no conversion VALU work, no ds_write pass, and (for a 3x3 conv with 4 N tiles) a value that used to be split 36
// times is split once.  What the register-staged kernel spends on the loader (16 staging VGPRs + ~40 VALU per K-step per
// wavefront at the 128-VGPR occupancy step) goes to holding both k-halves' MFMA fragments at once instead.
//
// LDS image of one stage (identical to conv_igemm.hip): [A_hi | A_lo | B_hi | B_lo], rows of 32 bf16 (64 bytes), the four
// 16-byte slots of row r XOR-swizzled with (r >> 2) & 3.  An LDS-DMA instruction writes lane l's 16 bytes at
// M0 + 16*l, i.e. one instruction fills 16 consecutive rows of one plane (row = l >> 2, slot = l & 3); the swizzle is applied
// on the SOURCE side: lane l fetches the k-chunk (l & 3) ^ ((l >> 4) & 3) of its row.  Out-of-image taps and the K tail read
// the library's zero page (select on the address).
//
// Pipeline: a ring of NS LDS stages.  At the top of step kt the DMAs of tile kt+NS-1 are issued into the stage whose reads ended
// before the previous barrier; the step's fragment reads and MFMAs run on stage kt % NS; then `s_waitcnt vmcnt(DPT*(NS-2))`
// (DPT = DMA instructions per tile and wavefront, a compile-time constant: every wavefront issues the same number, each as ONE
// instruction whatever the lanes' in-image predicates) retires tile kt+1 and leaves the younger tiles in flight across the
// barrier that publishes it (raw s_barrier: __syncthreads() would drain vmcnt to 0).  The reads of a stage happen strictly
// after the barrier that follows the wait, as the LDS-DMA ordering rule requires.  NS = 2 is the plain double buffer
// (two workgroups per CU cover each other); NS = 3 with a 256x128 tile is one workgroup per CU with 96 KB of tiles in flight.
#include "../conv_tile.h"

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

#ifdef FGT_CONV_TRACE
// Diagnostic build only (python tools/conv_trace.py): every wavefront stamps s_memtime at the phase boundaries of its first TR_STEPS K-steps
// into spare LDS and the workgroup dumps them (plus HW_ID / XCC_ID) to a global buffer before the epilogue.  Not part of the product library.
constexpr int TR_STEPS = 32, TR_NST = 8, TR_HDR = 12;
__device__ unsigned* g_conv_trace = nullptr;
__device__ long g_conv_trace_words = 0;
#define TR_STAMP(i) ts[i] = __builtin_readcyclecounter()
#define TR_STORE(kt)                                                                                              \
    if ((kt) < TR_STEPS && lane == 0) {                                                                           \
        unsigned* tr_ = trace_lds + (wave * TR_STEPS + (kt)) * TR_NST;                                            \
        for (int i_ = 0; i_ < TR_NST; ++i_) tr_[i_] = (unsigned)ts[i_];                                           \
    }
#else
#define TR_STAMP(i)
#define TR_STORE(kt)
#endif

template <int BM, int BN, int WM, int WN, int MINW, int NS, bool PP, bool IL = false, int P8 = 0, int EA = 0, bool XY = false>
__global__ void __launch_bounds__(WM* WN * 64, MINW) conv_split_kernel(const ConvP p) {
    constexpr int NW = WM * WN;
    // XY (with EA): only the upper half of the wavefronts (one per SIMD) issues LDS-DMAs — all of them; the lower half goes straight from
    // the stage-release barrier into its MFMAs, so the matrix pipe works while the loading half is parked in front of the memory pipe
    constexpr int NL = XY ? NW / 2 : NW;                // wavefronts that load
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int STAGE = (BM + BN) * LDB;              // floats per stage (= (BM+BN) * 64 bf16 = hi + lo planes)
    constexpr int GA = BM / 16, GB = BN / 16;           // 16-row DMA groups per plane
    constexpr int A_IT = GA / NL;                       // A groups per loading wavefront (hi and lo plane of the same rows)
    constexpr int B_IT = 2 * GB / NL;                   // B (group, plane) pieces per loading wavefront: piece j = lwave + it*NL -> plane j / GB, group j % GB
    constexpr int DPT = 2 * A_IT + B_IT;                // DMA instructions per tile and loading wavefront
    static_assert(GA % NL == 0 && A_IT >= 1 && (2 * GB) % NL == 0 && B_IT >= 1 && TM >= 1 && TN >= 1, "tile / wavefront geometry");
    static_assert(!XY || (EA && NW % 2 == 0), "XY: early-release tiles with an even wavefront count");
    static_assert(NS >= 2 && NS <= 4, "LDS ring depth");
    static_assert(!PP || (NS >= 3 && NW % 2 == 0), "ping-pong needs a ring of >= 3 stages and an even wavefront count");
    static_assert(!IL || (NS == 2 && !PP && TM >= 2), "interleaved schedule: double buffer, >= 2 row blocks per wavefront");
    static_assert(!EA || (NS == 2 && !PP && !IL && !P8), "early stage release: plain double buffer");
    static_assert(!P8 || (NS == 2 && !PP && !IL && NW == 8 && BM == 256 && TM % 2 == 0 && TN % 2 == 0 && A_IT == 2 && (B_IT == 2 || B_IT == 4)),
                  "8-phase schedule: 256-row tile on 8 wavefronts, double buffer");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool loads = !XY || wave >= NW / 2;           // (wave-uniform) this wavefront issues DMAs
    const int lwave = XY ? (wave & (NL - 1)) : wave;     // its index among the loading wavefronts (the others never issue: any valid index)
    static_assert(!XY || (NL & (NL - 1)) == 0, "XY: power-of-two loader count");
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    const __bf16* const x0 = reinterpret_cast<const __bf16*>(p.x0);
    const __bf16* const x1 = reinterpret_cast<const __bf16*>(p.x1);
    const __bf16* const zp = reinterpret_cast<const __bf16*>(p.zero_page);
    // (copied out of the kernel-argument struct: a select between two struct fields would otherwise be compiled as a
    //  select between their ADDRESSES followed by a vector load from the argument segment, i.e. a vmcnt(0) in the loop)
    // il: hi/lo interleaved per 32 channels (in_split = 2): logical channel c -> element (c/32)*64 + c%32, lo 32 further.  A lane's
    // chunk column kc*8 is the same in every K-step, so the element offset of logical channel chb + ci is 2*(chb + ci) - kc*8.
    const bool il = d.in_split == 2;
    const int ld0 = d.ld0, ld1 = d.ld1;
    const int chb0 = (d.off0 + g * p.Cg0) << (il ? 1 : 0), chb1 = (d.off1 + g * p.Cg1 - p.Cg0) << (il ? 1 : 0);
    const long ps0 = il ? 32 : p.ps0, ps1 = il ? 32 : p.ps1;
    const int Cg0 = p.Cg0, Cg = p.Cg;

    // ---- this lane's DMA rows: row (lane >> 2) of each of its 16-row groups, k-chunk kc of every K-step
    const int lrow = lane >> 2;
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);      // swizzle on the source side (all groups start at multiples of 16 rows)
    const int il_sh = il ? 1 : 0, il_sub = il ? kc * 8 : 0;
    int a_iy0[A_IT], a_ix0[A_IT], a_nb[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = bm0 + (lwave + it * NL) * 16 + lrow;
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
    int tap = k_cur / p.Cg;
    int ci = k_cur - tap * p.Cg;
    int ky = tap / d.kw, kx = tap - ky * d.kw;

    // per-row gather bases for the current (tap, source); recomputed only when the chunk moves to another tap / source
    const __bf16* a_base[A_IT];
    unsigned a_okmask = 0;
    int seg_end = 0;
    long a_ps = 0;
    auto retap = [&]() {
        const bool in0 = ci < Cg0;
        const __bf16* src = in0 ? x0 : x1;
        const int ld = in0 ? ld0 : ld1;
        const int chb = in0 ? chb0 : chb1;   // channel = chb + ci
        a_ps = in0 ? ps0 : ps1;
        seg_end = in0 ? Cg0 : Cg;
        const int dy = ky * d.dh, dx = kx * d.dw;
        const int ush = d.upsample ? 1 : 0;
        const bool rep = d.pad_mode != 0;
        a_okmask = 0;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
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

    // weights: planes [2][groups][Npad][Kpad] bf16, or interleaved rows of 2*Kpad (w_il)
    const __bf16* wrow[B_IT];
    const bool wil = d.w_il != 0;                       // interleaved weights: [hi 32 | lo 32] per K-step
    const long w_ps = wil ? 32 : (long)d.groups * d.Npad * d.Kpad;
    const int w_adv = wil ? 2 * BK : BK;
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int piece = lwave + it * NL, plane = piece / GB, grp = piece % GB;
        const int brow = bn0 + grp * 16 + lrow;          // rows past Npad (tiles wider than the 128-row padding): zeros
        wrow[it] = brow < d.Npad ? reinterpret_cast<const __bf16*>(p.w) + ((long)g * d.Npad + brow) * (wil ? 2 * d.Kpad : d.Kpad) + kc * 8 + plane * w_ps
                                 : nullptr;
    }

    char* const lds = reinterpret_cast<char*>(smem);
    constexpr int STAGE_B = STAGE * 4;
    // one DMA instruction per piece whatever the predicates: the zero-page select is arithmetic on the address (a select between
    // a uniform and a per-lane pointer gets compiled into two exec-masked instructions, which would make the vmcnt count vary)
    const unsigned long zpi = reinterpret_cast<unsigned long>(zp);
    auto sel = [&](const __bf16* ptr, bool ok) {
        const unsigned long a = reinterpret_cast<unsigned long>(ptr);
        return reinterpret_cast<const void*>(zpi + ((a - zpi) & (ok ? ~0ul : 0ul)));
    };
    auto advance_A = [&]() {
        k_cur += BK;
        ci += BK;
        if (ci >= seg_end) {
            while (ci >= Cg) {
                ci -= Cg;
                if (++kx == d.kw) { kx = 0; ++ky; }
            }
            retap();
        }
    };
    // single DMA pieces (interleaved schedule): piece j < 2*A_IT is (A group j/2, plane j%2), the rest are the B pieces
    auto issue_piece = [&](int j, int slot) {
        char* st = lds + slot * STAGE_B;
        if (j < 2 * A_IT) {
            const int it = j >> 1, plane = j & 1;
            const bool ok = (k_cur < p.K) && ((a_okmask >> it) & 1u);
            const __bf16* src = a_base[it] + ((ci << il_sh) - il_sub) + (plane ? a_ps : 0);
            glds16(sel(src, ok), st + (lwave + it * NL) * 1024 + plane * BM * 64);
        } else {
            const int it = j - 2 * A_IT;
            const int piece = lwave + it * NL, plane = piece / GB, grp = piece % GB;    // wave-uniform
            const bool bok = BN <= 128 || wrow[it] != nullptr;
            glds16(sel(wrow[it], bok), st + 2 * BM * 64 + plane * BN * 64 + grp * 1024);
            if (bok) wrow[it] += w_adv;
        }
    };
    auto issue_tile = [&](int slot) {
        char* st = lds + slot * STAGE_B;
        const bool kval = k_cur < p.K;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const bool ok = kval && ((a_okmask >> it) & 1u);
            const __bf16* src = a_base[it] + ((ci << il_sh) - il_sub);
            char* dst = st + (lwave + it * NL) * 1024;
            glds16(sel(src, ok), dst);                           // A_hi rows
            glds16(sel(src + a_ps, ok), dst + BM * 64);          // A_lo rows
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int piece = lwave + it * NL, plane = piece / GB, grp = piece % GB;    // wave-uniform
            char* dst = st + 2 * BM * 64 + plane * BN * 64 + grp * 1024;
            const bool bok = BN <= 128 || wrow[it] != nullptr;
            glds16(sel(wrow[it], bok), dst);
            if (bok) wrow[it] += w_adv;
        }
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
#ifdef FGT_CONV_TRACE
    unsigned long long ts[TR_NST] = {};
    unsigned* const trace_lds = reinterpret_cast<unsigned*>(smem + NS * STAGE);
    const unsigned long long tr_t0 = __builtin_readcyclecounter();
    const unsigned long long tr_r0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: shader clock = d(memtime) / d(memrealtime)
    for (int i = tid; i < NW * TR_STEPS * TR_NST; i += NW * 64) trace_lds[i] = 0;
#endif

    // ---- prologue: tiles 0 .. NS-2 in flight, tile 0 landed (EA: tiles 0 and 1 in flight)
    constexpr int AHEAD = EA ? 2 : NS - 1;
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
        if (t < p.nk && loads) issue_tile(t);
    if (p.nk >= AHEAD) wait_vmcnt<DPT * (AHEAD - 1)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int slot = 0, slot_in = AHEAD % NS;
#ifdef FGT_CONV_TRACE
    const unsigned long long tr_t1 = __builtin_readcyclecounter();   // prologue over (tile 0 landed)
#endif

    auto read_frags = [&](bf16x8 (&ah)[2][TM], bf16x8 (&al)[2][TM], bf16x8 (&bh)[2][TN], bf16x8 (&bl)[2][TN]) {
        const __bf16* base = reinterpret_cast<const __bf16*>(smem + slot * STAGE);
        // operand rows: wave-tile base (multiple of 32) + l31, so (row >> 2) & 3 == (l31 >> 2) & 3 for every fragment
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = swz(l31, ks * 2 + lh);
            const __bf16* Ahi = base + (wm * WTM + l31) * LDB + so;
            const __bf16* Bhi = base + 2 * BM * LDB + (wn * WTN + l31) * LDB + so;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[ks][i] = *reinterpret_cast<const bf16x8*>(Ahi + i * 32 * LDB);
                al[ks][i] = *reinterpret_cast<const bf16x8*>(Ahi + BM * LDB + i * 32 * LDB);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[ks][j] = *reinterpret_cast<const bf16x8*>(Bhi + j * 32 * LDB);
                bl[ks][j] = *reinterpret_cast<const bf16x8*>(Bhi + BN * LDB + j * 32 * LDB);
            }
        }
    };
    // same product order as conv_igemm.hip (lo*hi, hi*lo, hi*hi per k-half): bit-identical accumulators
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

    if constexpr (IL) {
        // Interleaved schedule (one workgroup per CU, large tile): the K-step's DMA pieces and fragment reads are spread through
        // its MFMAs instead of being issued as bursts.  A-fragments live in ONE register set that is refilled row block by row block
        // for the next k-half as soon as the row block's MFMAs of the current half are issued; B-fragments are double buffered.
        // Per (row block i, k-half ks): 3*TN MFMAs in the product order of conv_igemm.hip, then one fragment refill and one DMA.
        bf16x8 ah[TM], al[TM], bh[2][TN], bl[2][TN];
        auto readA = [&](int i, int ks) {
            const __bf16* base = reinterpret_cast<const __bf16*>(smem + slot * STAGE);
            const __bf16* Ahi = base + (wm * WTM + i * 32 + l31) * LDB + swz(l31, ks * 2 + lh);
            ah[i] = *reinterpret_cast<const bf16x8*>(Ahi);
            al[i] = *reinterpret_cast<const bf16x8*>(Ahi + BM * LDB);
        };
        auto readB = [&](int ks) {
            const __bf16* base = reinterpret_cast<const __bf16*>(smem + slot * STAGE);
            const __bf16* Bhi = base + 2 * BM * LDB + (wn * WTN + l31) * LDB + swz(l31, ks * 2 + lh);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[ks][j] = *reinterpret_cast<const bf16x8*>(Bhi + j * 32 * LDB);
                bl[ks][j] = *reinterpret_cast<const bf16x8*>(Bhi + BN * LDB + j * 32 * LDB);
            }
        };
        auto mm = [&](int i, int ks) {
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[ks][j], acc[i][j], 0, 0, 0);
        };
        constexpr int GROUPS = 2 * TM;                       // (ks, i) groups per K-step
        constexpr int PPG = (DPT + GROUPS - 1) / GROUPS;     // DMA pieces issued behind each MFMA group
        for (int kt = 0; kt < p.nk; ++kt) {
            const bool more = kt + 1 < p.nk;
            readB(0);
            readA(0, 0);
            readA(1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gi = 0; gi < GROUPS; ++gi) {
                const int ks = gi / TM, i = gi % TM;
                mm(i, ks);
                __builtin_amdgcn_sched_barrier(0);
                // refill: the A row block two groups ahead (same k-half, or row block 0/1 of the next half), B of the next half
                const int gn = gi + 2;
                if (gn < GROUPS) {
                    if (gn == TM) readB(1);
                    readA(gn % TM, gn / TM);
                }
                if (more) {
#pragma unroll
                    for (int j = gi * PPG; j < (gi + 1) * PPG && j < DPT; ++j) {
                        issue_piece(j, slot_in);
                        if (j == 2 * A_IT - 1) advance_A();
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
            slot_in ^= 1;
        }
    } else if constexpr (P8 != 0) {
        // ---- 8-phase staggered schedule (two K tiles = 8 phases; cdna_hip_programming.md "256^2 8-phase template", re-derived for the
        // hi/lo-split implicit GEMM).  One workgroup per CU, 8 wavefronts = two groups G0 (waves 0-3) / G1 (waves 4-7), one wavefront of
        // each group per SIMD.  A K tile is 4 phases, one per quadrant of the wavefront's output tile (row half ih x column half jh):
        //     L(q): fragment reads for the quadrant + 2 LDS-DMA pieces of tile kt+1   | s_barrier |
        //     C(q): (TM/2)*(TN/2)*6 MFMAs under s_setprio 1                           | s_barrier |
        // G1 runs ONE barrier behind G0, so in every barrier interval one wavefront of each SIMD issues loads / DMAs while its partner
        // owns the matrix pipe (the DMA issue cost, ~100-185 cycles per piece, and the LDS reads hide under the partner's MFMAs instead
        // of idling the pipe for all 8 wavefronts at once).  Quadrant order (0,0) (0,1) (1,1) (1,0): A half ih stays in registers for two
        // phases, both B halves stay for the tile -> reads per phase 12 / 4 / 8 / 0 (256x256).
        // Staging of tile kt+1 (other LDS stage) during tile kt, by "half tiles" (HB0, HB1: B rows 0-127 / 128-255; HA0, HA1: A rows):
        //     L(0): HB0   L(1): HB1   L(2): HA0   L(3): HA1        (every wavefront contributes its pieces of the same half tile)
        // Intervals are numbered by barrier count, tile kt starts at 8kt: G0 has L(q) in 8kt+2q, C(q) in 8kt+2q+1; G1 one later.
        //   reads of tile kt:   HB in 8kt+0..3, HA0 (G0 only: its rows) in 8kt+0 and +4, HA1 (G1 only) in 8kt+1 and +5; a read issued in
        //                       interval n is complete (lgkmcnt(0) at the head of C) before its wavefront leaves interval n+1;
        //   WAR: stage of tile kt+1 held tile kt-1, last read in interval 8kt-3  ->  every DMA below (>= 8kt) is safe;
        //   RAW: G0 reads HB, HA0 of tile kt+1 in 8kt+8, G1 reads HB, HA1 in 8kt+9.  Counted waits in FRONT of the barriers that end
        //        intervals 8kt+7 and 8kt+8 (never in the phase that reads):
        //            end of 8kt+7:  G0 (end of C(3)) vmcnt(n3)   G1 (end of L(3)) vmcnt(n3)    -> all but the HA1 pieces have landed
        //            end of 8kt+8:  G0 (end of L(0) of tile kt+1) vmcnt(n0) [0 when nothing was staged]   G1 (end of C(3)) vmcnt(0)
        //        (n_q = pieces a wavefront issues in L(q); G1's vmcnt(0) waits for pieces issued a whole interval earlier).
        // Measured (profiles/r02_run3_split_sweep_p8*.txt): bit-identical, and NOT faster than the 128x128 8-wavefront tile — 318 vs 310 TF
        // algorithmic on the 640->512 g2 3x3 layer, slower on K = 512 GEMMs (16 K tiles: prologue / epilogue weigh more).  Timing-only
        // ablations of this loop on that layer: no DMA 439 TF (the ceiling of the phase structure, 53 % of the nominal bf16 peak), DMA +
        // fragment reads without MFMAs 508, reads + barriers alone 1101; staggered = lock-step within 3 %.  The DMA stream alone needs 86 %
        // of the MFMA-only time and runs at 64 KB in flight per CU / ~2 us = 31 GB/s per CU (8 TB/s chip-wide): it is LATENCY-bound by
        // the one-tile prefetch depth that 160 KB of LDS allows at this tile size, and overlaps only about half with the MFMA phases.
        // Kept as explicit tiles (parity-tested); the autotuner does not consider them.
        constexpr int HM = TM / 2, HN = TN / 2;
        constexpr int N0 = 2, N3 = 2;                       // pieces issued in L(0) (B hi + lo of HB0, or of the whole B tile) and L(3) (HA1 hi + lo)
        constexpr bool LOCK = (P8 & 4) != 0;                // A/B variant: both groups in lock step (same phases, no stagger)
        constexpr bool NO_DMA = (P8 & 8) != 0;              // timing-only ablations (wrong results): no staging inside the loop,
        constexpr bool NO_MMA = (P8 & 16) != 0;             //   no MFMAs (fragment reads kept alive),
        const bool g1 = !LOCK && wave >= 4;
        bf16x8 ah[HM][2], al[HM][2], bh[TN][2], bl[TN][2];  // [block][k half]
        auto readA8 = [&](int ih) {
            const __bf16* base = reinterpret_cast<const __bf16*>(smem + slot * STAGE);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const __bf16* Ahi = base + (wm * WTM + ih * HM * 32 + l31) * LDB + swz(l31, ks * 2 + lh);
#pragma unroll
                for (int i = 0; i < HM; ++i) {
                    ah[i][ks] = *reinterpret_cast<const bf16x8*>(Ahi + i * 32 * LDB);
                    al[i][ks] = *reinterpret_cast<const bf16x8*>(Ahi + BM * LDB + i * 32 * LDB);
                }
            }
        };
        auto readB8 = [&](int jh) {
            const __bf16* base = reinterpret_cast<const __bf16*>(smem + slot * STAGE);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const __bf16* Bhi = base + 2 * BM * LDB + (wn * WTN + jh * HN * 32 + l31) * LDB + swz(l31, ks * 2 + lh);
#pragma unroll
                for (int j = 0; j < HN; ++j) {
                    bh[jh * HN + j][ks] = *reinterpret_cast<const bf16x8*>(Bhi + j * 32 * LDB);
                    bl[jh * HN + j][ks] = *reinterpret_cast<const bf16x8*>(Bhi + BN * LDB + j * 32 * LDB);
                }
            }
        };
        // same per-accumulator product order as conv_igemm.hip (per k half: lo*hi, hi*lo, hi*hi): bit-identical results
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
        // DMA pieces of issue_piece(): j = 2*it + plane for the A row group it (0: HA0, 1: HA1); j = 4 + it for the B pieces, where
        // B_IT = 4 (BN = 256): it = 0 / 2 -> hi / lo of HB0, it = 1 / 3 -> hi / lo of HB1;  B_IT = 2 (BN = 128): it = 0 / 1 -> hi / lo.
        auto stage8 = [&](int q) {
            if (q == 0) {
                issue_piece(4, slot_in);
                issue_piece(B_IT == 4 ? 6 : 5, slot_in);
            } else if (q == 1) {
                if constexpr (B_IT == 4) { issue_piece(5, slot_in); issue_piece(7, slot_in); }
            } else if (q == 2) {
                issue_piece(0, slot_in); issue_piece(1, slot_in);
            } else {
                issue_piece(2, slot_in); issue_piece(3, slot_in);
                advance_A();
            }
        };
        auto compute8 = [&](int ih, int jh) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (P8 & 2) __builtin_amdgcn_s_setprio(1);
            if constexpr (!NO_MMA) mm8(ih, jh);
            else {
#pragma unroll
                for (int i = 0; i < HM; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) asm volatile("" ::"v"(ah[i][ks]), "v"(al[i][ks]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) asm volatile("" ::"v"(bh[j][ks]), "v"(bl[j][ks]));
            }
            if constexpr (P8 & 2) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (g1) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < p.nk; ++kt) {
            const bool st = !NO_DMA && kt + 1 < p.nk;       // tile kt+1 exists: stage it during this tile
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
            if (g1 || LOCK) wait_vmcnt<0>();                 // (lock step: the HA1 pieces are read by rows 128-255 in the very next interval)
            else if (st) wait_vmcnt<N3>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
            slot_in ^= 1;
        }
        if (!g1) __builtin_amdgcn_s_barrier();
    } else if constexpr (EA) {
        // Early stage release.  The K loop of the plain double buffer is LATENCY-bound: the DMAs of tile kt+1 are issued at the top of
        // step kt and must have landed at its end, so a step cannot be shorter than the L2 / HBM -> LDS latency (~1.9 us under load:
        // PMC on the bench, profiles/r02_run5_pmc_clock.txt: matrix pipe 39 % busy, wavefronts 39 % parked at s_waitcnt / barrier at
        // 2.1 GHz), whatever the compute.  But a step reads ALL its fragments into registers before its first MFMA, so the stage is dead
        // as soon as every wavefront has read it: one extra barrier after the fragment reads frees it for tile kt+2 a whole step
        // early — two tiles (2 x DPT pieces per wavefront) in flight with the same two stages, a copy now has two steps to land.
        //   step kt: read tile kt (stage kt&1) | lgkmcnt(0) | barrier | issue tile kt+2 -> stage kt&1 | MFMAs | vmcnt(DPT): tile kt+1
        //            landed, tile kt+2 may fly | barrier.   Same products in the same order: bit-identical.
        // Timeline of this loop (tools/conv_trace.py, profiles/r02_run7_conv_trace_enc10.txt; 1.84 GHz measured in the kernel): step 2 816
        // cycles = reads 416 | barrier 220 | DMA issue 896 | 12 MFMAs 352 | vmcnt 60 | barrier 424.  With two tiles in flight the latency IS
        // hidden (vmcnt never waits); what is left is the rate of the vector-memory pipe: 44 cycles per 1-KB instruction per CU.
        for (int kt = 0; kt < p.nk; ++kt) {
            bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
            TR_STAMP(0);
            read_frags(ah, al, bh, bl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            TR_STAMP(1);
            __builtin_amdgcn_s_barrier();                   // every wavefront holds its fragments of tile kt: the stage can be refilled
            TR_STAMP(2);
            const bool more = kt + 2 < p.nk;
            // (DMAs BEFORE the MFMAs on purpose: the vector-memory pipe is the critical resource and has to be fed as early as the stage
            //  is free.  The opposite order — MFMAs first, DMA issue underneath them — measured 6 % slower on the 3x3 layers:
            //  profiles/r02_run9_split_sweep_mfma_first.txt; moving the DMA issue of HALF of the wavefronts to the top of the next step, so that
            //  only half of it stands in front of MFMAs: 326 vs 339 TF, profiles/r02_run9_split_sweep_late_half.txt — the 32 instructions of a
            //  tile take the same time wherever they are issued)
            if (more && loads) issue_tile(slot);
            __builtin_amdgcn_sched_barrier(0);
            TR_STAMP(3);
            mfmas(ah, al, bh, bl);
            __builtin_amdgcn_sched_barrier(0);
            TR_STAMP(4);
            if (more) wait_vmcnt<DPT>(); else wait_vmcnt<0>();
            TR_STAMP(5);
            __builtin_amdgcn_s_barrier();
            TR_STAMP(6);
            TR_STORE(kt);
            slot ^= 1;
        }
    } else if constexpr (!PP) {
        for (int kt = 0; kt < p.nk; ++kt) {
            TR_STAMP(0);
            if (kt + AHEAD < p.nk) issue_tile(slot_in);
            TR_STAMP(2);
            bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
            read_frags(ah, al, bh, bl);
            __builtin_amdgcn_sched_barrier(0);   // keep all fragment reads of the step ahead of its MFMAs
#ifdef FGT_CONV_TRACE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            TR_STAMP(3);
            mfmas(ah, al, bh, bl);
            // The wait + barrier stay BEHIND the MFMAs (hoisted above them, the DMA latency would be exposed in front of this
            // wavefront's matrix work instead of running underneath it).  Tile kt+1 must have landed; while NS-2 younger tiles
            // exist they stay in flight (the last NS-2 steps drain everything: a constant immediate needs a constant count).
            __builtin_amdgcn_sched_barrier(0);
            TR_STAMP(4);
            if (kt + AHEAD < p.nk) wait_vmcnt<DPT * (AHEAD - 1)>(); else wait_vmcnt<0>();
            TR_STAMP(5);
            __builtin_amdgcn_s_barrier();
            TR_STAMP(6);
            TR_STORE(kt);
            slot = slot + 1 == NS ? 0 : slot + 1;
            slot_in = slot_in + 1 == NS ? 0 : slot_in + 1;
        }
    } else {
        // Ping-pong: the wavefronts form two groups (one wavefront of each group per SIMD).  A step is two barrier-delimited
        // phases, R = {issue the DMAs of tile kt+NS-1, read tile kt's fragments, wait until this wavefront's pieces of tile kt+1
        // have landed} and M = {the MFMAs}.  Group 1 runs ONE barrier behind group 0, so in every interval one group is in R
        // (LDS + address work) while the other is in M (matrix pipe): the two halves of a K-step that a lock-stepped workgroup
        // serialises overlap across the groups.  Ordering (intervals numbered by barrier count, G0: R(k) in 2k, M(k) in 2k+1;
        // G1 one later): tile k+1 is read from interval 2k+2 on, every wavefront's wait for its pieces sits in its R(k) (<= 2k+1);
        // the stage of tile k-1 is overwritten from interval 2k on, its last reads are in G1's R(k-1) (2k-1).
        const bool g1 = wave >= NW / 2;
        if (g1) __builtin_amdgcn_s_barrier();
        for (int kt = 0; kt < p.nk; ++kt) {
            if (kt + AHEAD < p.nk) issue_tile(slot_in);
            bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
            read_frags(ah, al, bh, bl);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + AHEAD < p.nk) wait_vmcnt<DPT * (AHEAD - 1)>(); else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            mfmas(ah, al, bh, bl);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            slot = slot + 1 == NS ? 0 : slot + 1;
            slot_in = slot_in + 1 == NS ? 0 : slot_in + 1;
        }
        if (!g1) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef FGT_CONV_TRACE
    unsigned* tr_hdr = nullptr;
    {
        const unsigned long long tr_t2 = __builtin_readcyclecounter();   // K loop over
        __syncthreads();
        constexpr int PER_WG = NW * (TR_HDR + TR_STEPS * TR_NST);
        const long wg = (long)blockIdx.y * gridDim.x + blockIdx.x;
        unsigned* out = g_conv_trace;
        if (out && (wg + 1) * PER_WG <= g_conv_trace_words) {
            out += wg * PER_WG;
            if (lane == 0) {
                unsigned* h = out + wave * TR_HDR;
                h[0] = __builtin_amdgcn_s_getreg(63492);    // HW_ID
                h[1] = __builtin_amdgcn_s_getreg(63508);    // XCC_ID
                h[2] = (unsigned)tr_t0;
                h[3] = (unsigned)p.nk;
                h[4] = (unsigned)tr_t1;
                h[5] = (unsigned)tr_t2;
                tr_hdr = h;
            }
            for (int i = tid; i < NW * TR_STEPS * TR_NST; i += NW * 64) out[NW * TR_HDR + i] = trace_lds[i];
        }
        __syncthreads();
    }
#endif

    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN>(p, acc, smem, bm0, bn0, g);
#ifdef FGT_CONV_TRACE
    if (tr_hdr) {
        tr_hdr[8] = (unsigned)__builtin_readcyclecounter();              // epilogue instructions issued (stores may be in flight)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr_hdr[6] = (unsigned)__builtin_readcyclecounter();              // epilogue over (stores acknowledged)
        tr_hdr[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - tr_r0);
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Loader-wavefront variant ("lw" tiles).  The timeline of the early-release loop above (tools/conv_trace.py) shows every wavefront parked
// ~900 cycles per K step in front of the shared vector-memory pipe while it issues its own 4 LDS-DMAs — in FRONT of its MFMAs; issuing the
// MFMAs first instead starves the pipe (measured, 6 % slower).  Here the two jobs belong to different wavefronts: NW consumer wavefronts
// (fragment reads + MFMAs + epilogue, no address arithmetic at all) and two loader wavefronts per workgroup — one owns the whole A (im2col)
// tile, one the B (weight) tile — that do nothing but address arithmetic and LDS-DMA issue.  Same two stages, two barriers per step and
// two tiles in flight as `EA`:
//     consumers:  read tile kt (stage kt&1) | lgkmcnt(0) | barrier A | MFMAs                                   | barrier B
//     loaders:                                             barrier A | issue tile kt+2 -> stage kt&1, vmcnt: tile kt+1 landed | barrier B
// so between A and B the matrix pipe and the vector-memory pipe both start at once.  Same products in the same order: bit-identical.
// Measured (profiles/r02_run9_split_sweep_loader_waves.txt, _conv_trace_loader_waves.txt): SLOWER — 262 vs 337 TF algorithmic on the
// 640->512 3x3 layer.  One wavefront issues a 1-KB LDS-DMA every 70 (weights, 8 cache lines) to 133 cycles (im2col rows, 16 lines): the A
// loader needs 2 100 cycles for its 16 instructions, the B loader 1 100, and the consumers wait at barrier B.  Saturating the pipe takes
// the issue parallelism of >= 8 wavefronts, which the register file does not offer on top of 8 consumers (20 wavefronts per CU: 96
// registers).  `XY` (half of the wavefronts issue everything, the other half starts its MFMAs at once) ties with `EA` (338 vs 342): the
// 32 instructions of a tile take 800-1 900 cycles whoever issues them — the pipe, not the issue order, is the limit.  Explicit tiles only.
template <int BM, int BN, int WM, int WN, int MINW>
__global__ void __launch_bounds__((WM * WN + 2) * 64, MINW) conv_split_lw_kernel(const ConvP p) {
    constexpr int NW = WM * WN;                         // consumer wavefronts; wavefront NW loads A, NW + 1 loads B
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int STAGE = (BM + BN) * LDB;              // floats per stage
    constexpr int STAGE_B = STAGE * 4;
    constexpr int GA = BM / 16, GB = BN / 16;           // 16-row DMA groups per plane
    constexpr int PA = 2 * GA, PB = 2 * GB;             // DMA instructions per tile of the A / the B loader
    static_assert(TM >= 1 && TN >= 1 && PA <= 32 && PB <= 32, "tile / wavefront geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;
    const int nk = p.nk;
#ifdef FGT_CONV_TRACE
    unsigned long long ts[TR_NST] = {};
    unsigned* const trace_lds = reinterpret_cast<unsigned*>(smem + 2 * STAGE);
    const unsigned long long tr_t0 = __builtin_readcyclecounter();
    const unsigned long long tr_r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = tid; i < (NW + 2) * TR_STEPS * TR_NST; i += (NW + 2) * 64) trace_lds[i] = 0;
    __syncthreads();
#endif

    if (wave >= NW) {
        // ================================================================ loaders
        char* const lds = reinterpret_cast<char*>(smem);
        const __bf16* const zp = reinterpret_cast<const __bf16*>(p.zero_page);
        const unsigned long zpi = reinterpret_cast<unsigned long>(zp);
        auto sel = [&](const __bf16* ptr, bool ok) {      // arithmetic select (one DMA instruction whatever the predicate)
            const unsigned long a = reinterpret_cast<unsigned long>(ptr);
            return reinterpret_cast<const void*>(zpi + ((a - zpi) & (ok ? ~0ul : 0ul)));
        };
        const int lrow = lane >> 2;
        const int kc = (lane & 3) ^ ((lane >> 4) & 3);  // swizzle on the source side
        if (wave == NW) {
            // ---- A loader: row (lane >> 2) of each of the GA 16-row groups, both planes
            const __bf16* const x0 = reinterpret_cast<const __bf16*>(p.x0);
            const __bf16* const x1 = reinterpret_cast<const __bf16*>(p.x1);
            const bool il = d.in_split == 2;
            const int ld0 = d.ld0, ld1 = d.ld1;
            const int chb0 = (d.off0 + g * p.Cg0) << (il ? 1 : 0), chb1 = (d.off1 + g * p.Cg1 - p.Cg0) << (il ? 1 : 0);
            const long ps0 = il ? 32 : p.ps0, ps1 = il ? 32 : p.ps1;
            const int Cg0 = p.Cg0, Cg = p.Cg;
            const int il_sh = il ? 1 : 0, il_sub = il ? kc * 8 : 0;
            int a_iy0[GA], a_ix0[GA], a_nb[GA];
#pragma unroll
            for (int it = 0; it < GA; ++it) {
                const int m = bm0 + it * 16 + lrow;
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
            int tap = k_cur / p.Cg;
            int ci = k_cur - tap * p.Cg;
            int ky = tap / d.kw, kx = tap - ky * d.kw;
            const __bf16* a_base[GA];
            unsigned a_okmask = 0;
            int seg_end = 0;
            long a_ps = 0;
            auto retap = [&]() {
                const bool in0 = ci < Cg0;
                const __bf16* src = in0 ? x0 : x1;
                const int ld = in0 ? ld0 : ld1;
                const int chb = in0 ? chb0 : chb1;
                a_ps = in0 ? ps0 : ps1;
                seg_end = in0 ? Cg0 : Cg;
                const int dy = ky * d.dh, dx = kx * d.dw;
                const int ush = d.upsample ? 1 : 0;
                const bool rep = d.pad_mode != 0;
                a_okmask = 0;
#pragma unroll
                for (int it = 0; it < GA; ++it) {
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
            auto issue_a = [&](int slot) {
                char* st = lds + slot * STAGE_B;
                const bool kval = k_cur < p.K;
#pragma unroll
                for (int it = 0; it < GA; ++it) {
                    const bool ok = kval && ((a_okmask >> it) & 1u);
                    const __bf16* src = a_base[it] + ((ci << il_sh) - il_sub);
                    char* dst = st + it * 1024;
                    glds16(sel(src, ok), dst);                           // A_hi rows
                    glds16(sel(src + a_ps, ok), dst + BM * 64);          // A_lo rows
                }
                k_cur += BK;
                ci += BK;
                if (ci >= seg_end) {
                    while (ci >= Cg) {
                        ci -= Cg;
                        if (++kx == d.kw) { kx = 0; ++ky; }
                    }
                    retap();
                }
            };
            issue_a(0);
            if (nk > 1) { issue_a(1); wait_vmcnt<PA>(); } else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                                // P: tile 0 landed
            for (int kt = 0; kt < nk; ++kt) {
                __builtin_amdgcn_s_barrier();                            // A: every consumer holds tile kt in registers
                TR_STAMP(2);
                const bool more = kt + 2 < nk;
                if (more) issue_a(kt & 1);
                TR_STAMP(3);
                if (more) wait_vmcnt<PA>(); else wait_vmcnt<0>();        // tile kt+1 landed (tile kt+2 may fly)
                TR_STAMP(5);
                __builtin_amdgcn_s_barrier();                            // B
                TR_STAMP(6);
                TR_STORE(kt);
            }
        } else {
            // ---- B loader: row (lane >> 2) of each of the GB 16-row groups of both weight planes
            const bool wil = d.w_il != 0;
            const long w_ps = wil ? 32 : (long)d.groups * d.Npad * d.Kpad;
            const int w_adv = wil ? 2 * BK : BK;
            const __bf16* wrow[PB];
#pragma unroll
            for (int it = 0; it < PB; ++it) {
                const int plane = it / GB, grp = it % GB;
                const int brow = bn0 + grp * 16 + lrow;          // rows past Npad (tiles wider than the 128-row padding): zeros
                wrow[it] = brow < d.Npad ? reinterpret_cast<const __bf16*>(p.w) + ((long)g * d.Npad + brow) * (wil ? 2 * d.Kpad : d.Kpad) + kc * 8 + plane * w_ps
                                         : nullptr;
            }
            auto issue_b = [&](int slot) {
                char* st = lds + slot * STAGE_B + 2 * BM * 64;
#pragma unroll
                for (int it = 0; it < PB; ++it) {
                    const int plane = it / GB, grp = it % GB;
                    const bool bok = BN <= 128 || wrow[it] != nullptr;
                    glds16(sel(wrow[it], bok), st + plane * BN * 64 + grp * 1024);
                    if (bok) wrow[it] += w_adv;
                }
            };
            issue_b(0);
            if (nk > 1) { issue_b(1); wait_vmcnt<PB>(); } else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();                                // P
            for (int kt = 0; kt < nk; ++kt) {
                __builtin_amdgcn_s_barrier();                            // A
                TR_STAMP(2);
                const bool more = kt + 2 < nk;
                if (more) issue_b(kt & 1);
                TR_STAMP(3);
                if (more) wait_vmcnt<PB>(); else wait_vmcnt<0>();
                TR_STAMP(5);
                __builtin_amdgcn_s_barrier();                            // B
                TR_STAMP(6);
                TR_STORE(kt);
            }
        }
        return;
    }

    // ==================================================================== consumers
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
#ifdef FGT_CONV_TRACE
    const unsigned long long tr_t1 = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier();                                        // P: tile 0 landed
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
        TR_STAMP(0);
        {
            const __bf16* base = reinterpret_cast<const __bf16*>(smem + slot * STAGE);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int so = swz(l31, ks * 2 + lh);
                const __bf16* Ahi = base + (wm * WTM + l31) * LDB + so;
                const __bf16* Bhi = base + 2 * BM * LDB + (wn * WTN + l31) * LDB + so;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[ks][i] = *reinterpret_cast<const bf16x8*>(Ahi + i * 32 * LDB);
                    al[ks][i] = *reinterpret_cast<const bf16x8*>(Ahi + BM * LDB + i * 32 * LDB);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[ks][j] = *reinterpret_cast<const bf16x8*>(Bhi + j * 32 * LDB);
                    bl[ks][j] = *reinterpret_cast<const bf16x8*>(Bhi + BN * LDB + j * 32 * LDB);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        TR_STAMP(1);
        __builtin_amdgcn_s_barrier();                                    // A: the stage may be refilled
        TR_STAMP(2);
        TR_STAMP(3);
        // same product order as conv_igemm.hip (lo*hi, hi*lo, hi*hi per k-half): bit-identical accumulators
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
        TR_STAMP(4);
        TR_STAMP(5);
        __builtin_amdgcn_s_barrier();                                    // B: tile kt+1 is in its stage
        TR_STAMP(6);
        TR_STORE(kt);
        slot ^= 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef FGT_CONV_TRACE
    unsigned* tr_hdr = nullptr;
    {
        const unsigned long long tr_t2 = __builtin_readcyclecounter();
        __syncthreads();                                                 // (the loaders have left: only the consumers count)
        constexpr int PER_WG = (NW + 2) * (TR_HDR + TR_STEPS * TR_NST);
        const long wg = (long)blockIdx.y * gridDim.x + blockIdx.x;
        unsigned* out = g_conv_trace;
        if (out && (wg + 1) * PER_WG <= g_conv_trace_words) {
            out += wg * PER_WG;
            if (lane == 0) {
                unsigned* h = out + wave * TR_HDR;
                h[0] = __builtin_amdgcn_s_getreg(63492);
                h[1] = __builtin_amdgcn_s_getreg(63508);
                h[2] = (unsigned)tr_t0;
                h[3] = (unsigned)p.nk;
                h[4] = (unsigned)tr_t1;
                h[5] = (unsigned)tr_t2;
                tr_hdr = h;
            }
            for (int i = tid; i < (NW + 2) * TR_STEPS * TR_NST; i += NW * 64) out[(NW + 2) * TR_HDR + i] = trace_lds[i];
        }
        __syncthreads();
    }
#endif
    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN>(p, acc, smem, bm0, bn0, g);
#ifdef FGT_CONV_TRACE
    if (tr_hdr) {
        tr_hdr[8] = (unsigned)__builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tr_hdr[6] = (unsigned)__builtin_readcyclecounter();
        tr_hdr[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - tr_r0);
    }
#endif
}

template <int BM, int BN, int WM, int WN, int MINW>
int launch_lw(const ConvP& p, hipStream_t s) {
    constexpr int NT = (WM * WN + 2) * 64;
#ifdef FGT_CONV_TRACE
    constexpr size_t smem = (size_t)2 * (BM + BN) * LDB * sizeof(float) + (size_t)(WM * WN + 2) * TR_STEPS * TR_NST * 4;
#else
    constexpr size_t smem = (size_t)2 * (BM + BN) * LDB * sizeof(float);
#endif
    static_assert(smem <= 160 * 1024, "LDS stages do not fit");
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_split_lw_kernel<BM, BN, WM, WN, MINW>), (int)smem, lds_set, "conv_split_lw")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_split_lw_kernel<BM, BN, WM, WN, MINW>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_split_lw");
}

template <int BM, int BN, int WM, int WN, int MINW = 2, int NS = 2, bool PP = false, bool IL = false, int P8 = 0, int EA = 0, bool XY = false>
int launch(const ConvP& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
#ifdef FGT_CONV_TRACE
    constexpr size_t smem = (size_t)NS * (BM + BN) * LDB * sizeof(float) + (size_t)WM * WN * TR_STEPS * TR_NST * 4;
#else
    constexpr size_t smem = (size_t)NS * (BM + BN) * LDB * sizeof(float);
#endif
    static_assert(smem <= 160 * 1024, "LDS ring does not fit");
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_split_kernel<BM, BN, WM, WN, MINW, NS, PP, IL, P8, EA, XY>), (int)smem, lds_set, "conv_split")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_split_kernel<BM, BN, WM, WN, MINW, NS, PP, IL, P8, EA, XY>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_split");
}

}  // namespace

#ifdef FGT_CONV_TRACE
extern "C" int fgt_debug_conv_trace(void* buf, long words) {
    unsigned* b = static_cast<unsigned*>(buf);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_trace), &b, sizeof(b)) != hipSuccess) return FGT_ELAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_trace_words), &words, sizeof(words)) != hipSuccess) return FGT_ELAUNCH;
    return FGT_OK;
}
#endif

int fgt_conv_split_diag_launch(int tile, const ConvP& p, hipStream_t s) {
    switch (tile) {
        case FGT_TILE_128x128: return launch<128, 128, 2, 2>(p, s);
        case FGT_TILE_128x64: return launch<128, 64, 2, 2>(p, s);
        case FGT_TILE_64x64: return launch<64, 64, 2, 2>(p, s);
        case FGT_TILE_128x32: return launch<128, 32, 4, 1>(p, s);
        case FGT_TILE_256x128: return launch<256, 128, 4, 2>(p, s);
        case FGT_TILE_128x128x8: return launch<128, 128, 2, 4, 4>(p, s);
        case FGT_TILE_256x128x16: return launch<256, 128, 4, 4, 4>(p, s);
        case FGT_TILE_256x64x8: return launch<256, 64, 4, 2, 2>(p, s);
        case FGT_TILE_256x128x8_S3: return launch<256, 128, 4, 2, 2, 3>(p, s);       // one workgroup per CU, 3-stage ring (144 KB)
        case FGT_TILE_256x128x16_S3: return launch<256, 128, 4, 4, 4, 3>(p, s);
        case FGT_TILE_128x128x8_S4: return launch<128, 128, 2, 4, 4, 4>(p, s);        // one workgroup per CU, 4-stage ring (128 KB)
        case FGT_TILE_256x128x8_PP: return launch<256, 128, 4, 2, 2, 3, true>(p, s);  // + ping-pong wavefront groups
        case FGT_TILE_128x128x8_PP: return launch<128, 128, 2, 4, 4, 4, true>(p, s);
        case FGT_TILE_256x128x8_IL: return launch<256, 128, 4, 2, 2, 2, false, true>(p, s);
        // early stage release (two tiles in flight on two stages): the production tiles again, bit-identical results
        case FGT_TILE_128x128_EA: return launch<128, 128, 2, 2, 2, 2, false, false, 0, 1>(p, s);
        case FGT_TILE_128x64_EA: return launch<128, 64, 2, 2, 2, 2, false, false, 0, 1>(p, s);
        case FGT_TILE_64x64_EA: return launch<64, 64, 2, 2, 2, 2, false, false, 0, 1>(p, s);
        case FGT_TILE_128x128x8_EA: return launch<128, 128, 2, 4, 4, 2, false, false, 0, 1>(p, s);
        case FGT_TILE_256x128x16_EA: return launch<256, 128, 4, 4, 4, 2, false, false, 0, 1>(p, s);
        case FGT_TILE_256x64x8_EA: return launch<256, 64, 4, 2, 2, 2, false, false, 0, 1>(p, s);
        // early release with the DMA issue on half of the wavefronts (one per SIMD)
        case FGT_TILE_128x128x8_XY: return launch<128, 128, 2, 4, 4, 2, false, false, 0, 1, true>(p, s);
        // loader wavefronts: 8 (or 4) consumer wavefronts + an A loader + a B loader per workgroup, two workgroups per CU
        case FGT_TILE_128x128x8_LW: return launch_lw<128, 128, 2, 4, 5>(p, s);      // 20 wavefronts per CU: <= 96 registers
        case FGT_TILE_128x128_LW: return launch_lw<128, 128, 2, 2, 3>(p, s);        // 12 wavefronts per CU
        case FGT_TILE_128x64_LW: return launch_lw<128, 64, 2, 2, 3>(p, s);
        case FGT_TILE_256x256_P8: return launch<256, 256, 2, 4, 2, 2, false, false, 3>(p, s);     // 8-phase staggered schedule, setprio around the MFMAs
        case FGT_TILE_256x128_P8: return launch<256, 128, 4, 2, 2, 2, false, false, 3>(p, s);
#ifdef FGT_P8_ABLATIONS   // A/B and timing-only instances behind profiles/r02_run3_split_sweep_p8*.txt (build with -DFGT_P8_ABLATIONS to reproduce)
        case FGT_TILE_256x256_P8N: return launch<256, 256, 2, 4, 2, 2, false, false, 1>(p, s);    // without s_setprio
        case FGT_TILE_256x256_P8L: return launch<256, 256, 2, 4, 2, 2, false, false, 7>(p, s);    // the same phases in lock step
        case 22: return launch<256, 256, 2, 4, 2, 2, false, false, 3 + 8>(p, s);                   // timing only (wrong results): no DMA in the loop
        case 23: return launch<256, 256, 2, 4, 2, 2, false, false, 3 + 16>(p, s);                  // timing only: no MFMAs
        case 24: return launch<256, 256, 2, 4, 2, 2, false, false, 3 + 8 + 16>(p, s);              // timing only: fragment reads + barriers
        case 25: return launch<256, 256, 2, 4, 2, 2, false, false, 7 + 8>(p, s);                   // timing only: lock step, no DMA
#endif
        default: fgt_set_error("fgt_conv2d: unknown tile %d", tile); return FGT_EINVAL;
    }
}
