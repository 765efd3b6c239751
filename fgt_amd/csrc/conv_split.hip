// bf16x3 implicit-GEMM convolution / GEMM with PRE-SPLIT operands, fed by LDS-DMA (gfx950).
//
// Same math as conv_igemm.hip's FGT_PREC_BF16X3 path (hi/lo bf16 operands, three v_mfma_f32_32x32x16_bf16 per product in the
// same order, fp32 accumulate: results are bit-identical), different data movement.  The activations arrive already split
// (desc.in_split: two bf16 planes, written once by their producer), the weights are pre-split at pack time, so a K-step's
// tiles are plain 16-byte copies.  Every wavefront moves them global -> LDS with global_load_lds_dwordx4: no staging
// registers, no conversion VALU work, no ds_write pass, and (for a 3x3 conv with 4 N tiles) a value that used to be split 36
// times is split once.
//
// LDS image of one stage (identical to conv_igemm.hip): [A_hi | A_lo | B_hi | B_lo], rows of 32 bf16 (64 bytes), the four
// 16-byte slots of row r XOR-swizzled with (r >> 2) & 3.  An LDS-DMA instruction writes lane l's 16 bytes at
// M0 + 16*l, i.e. one instruction fills 16 consecutive rows of one plane (row = l >> 2, slot = l & 3); the swizzle is applied
// on the SOURCE side: lane l fetches the k-chunk (l & 3) ^ ((l >> 4) & 3) of its row.  Out-of-image taps and the K tail read
// the library's zero page (select on the address).  Inputs in the interleaved layout (in_split = 2) are served here with the same
// 16-row x 64-byte pieces; their full-line form (8 rows x 128 bytes) is csrc/conv_wide.hip.
//
// Two schedules on two LDS stages (EA):
//   0  plain double buffer: the DMAs of tile kt+1 are issued at the top of step kt, one barrier per step;
//   1  early stage release: a step reads ALL its fragments into registers before its first MFMA, so one extra barrier after the reads
//      frees the stage for tile kt+2 a whole step early — two tiles in flight, +5...11 % on the long-K convs (picked per shape by the
//      autotuner; bit-identical).
// Everything else that was tried on this loop — 3 / 4-stage rings, ping-pong wavefront groups, the 8-phase 256^2 schedule, dedicated
// loader wavefronts, MFMA-first / late-half issue orders, the s_memtime trace build — measured within +-5 % or slower (NOTEBOOK.md) and
// lives in csrc/diag/conv_split_variants.hip, which only diagnostic builds (`fgt_amd.build.build(variant="diag")`) compile.
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

template <int BM, int BN, int WM, int WN, int MINW, int EA>
__global__ void __launch_bounds__(WM* WN * 64, MINW) conv_split_kernel(const ConvP p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int STAGE = (BM + BN) * LDB;              // floats per stage (= (BM+BN) * 64 bf16 = hi + lo planes)
    constexpr int GA = BM / 16, GB = BN / 16;           // 16-row DMA groups per plane
    constexpr int A_IT = GA / NW;                       // A groups per wavefront (hi and lo plane of the same rows)
    constexpr int B_IT = 2 * GB / NW;                   // B (group, plane) pieces per wavefront: piece j = wave + it*NW -> plane j / GB, group j % GB
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
    // ABI 9 (desc.ps_phase_pad): the workgroup's columns lie in ONE sub-pixel (a, b) of the x2 output (the launcher checks BN | ps_c): its padding is (1 - a, 1 - b)
    int ph_e = d.ph, pw_e = d.pw;
    if (d.ps_phase_pad) {
        const int q = bn0 / d.ps_c;
        ph_e -= q >> 1;
        pw_e -= q & 1;
    }
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = bm0 + (wave + it * NW) * 16 + lrow;
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
        const int piece = wave + it * NW, plane = piece / GB, grp = piece % GB;
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
    auto issue_tile = [&](int slot) {
        char* st = lds + slot * STAGE_B;
        const bool kval = k_cur < p.K;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const bool ok = kval && ((a_okmask >> it) & 1u);
            const __bf16* src = a_base[it] + ((ci << il_sh) - il_sub);
            char* dst = st + (wave + it * NW) * 1024;
            glds16(sel(src, ok), dst);                           // A_hi rows
            glds16(sel(src + a_ps, ok), dst + BM * 64);          // A_lo rows
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int piece = wave + it * NW, plane = piece / GB, grp = piece % GB;    // wave-uniform
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

    // ---- prologue: tile 0 landed (early release: tiles 0 and 1 in flight)
    constexpr int AHEAD = EA ? 2 : 1;
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
        if (t < p.nk) issue_tile(t);
    if (p.nk >= AHEAD) wait_vmcnt<DPT * (AHEAD - 1)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    int slot = 0;

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

    if constexpr (EA) {
        //   step kt: read tile kt (stage kt&1) | lgkmcnt(0) | barrier | issue tile kt+2 -> stage kt&1 | MFMAs | vmcnt(DPT): tile kt+1
        //            landed, tile kt+2 may fly | barrier.   Same products in the same order: bit-identical.
        // Timeline of this loop (NOTEBOOK.md, profiles/r02_run7_conv_trace_enc10.txt; 1.84 GHz measured in the kernel): step 2 816
        // cycles = reads 416 | barrier 220 | DMA issue 896 | 12 MFMAs 352 | vmcnt 60 | barrier 424.  With two tiles in flight the latency IS
        // hidden (vmcnt never waits); what is left is the rate of the vector-memory pipe: 44 cycles per 1-KB instruction per CU.
        // (DMAs BEFORE the MFMAs on purpose: the opposite order measured 6 % slower, profiles/r02_run9_split_sweep_mfma_first.txt)
        for (int kt = 0; kt < p.nk; ++kt) {
            bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
            read_frags(ah, al, bh, bl);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                   // every wavefront holds its fragments of tile kt: the stage can be refilled
            const bool more = kt + 2 < p.nk;
            if (more) issue_tile(slot);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);                // (the MFMA block ahead of the other wavefronts' address arithmetic: +1...2 %, same-box A/B)
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
            // The wait + barrier stay BEHIND the MFMAs (hoisted above them, the DMA latency would be exposed in front of this
            // wavefront's matrix work instead of running underneath it).
            __builtin_amdgcn_sched_barrier(0);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            slot ^= 1;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // (epilogue diet, round 6: bias-map / two-headed bodies only in the two tiles the static fallback routes to when the tap kernels are switched
    //  off — 128x128 on 4 wavefronts and 64x64, plain schedule; the launcher declines such layers on the other tiles)
    constexpr bool OPS = EA == 0 && WM * WN == 4 && ((BM == 128 && BN == 128) || (BM == 64 && BN == 64));
    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN, OPS, OPS, true>(conv_epilogue_args(p), acc, smem, bm0, bn0, g);
}

template <int BM, int BN, int WM, int WN, int MINW = 2, int EA = 0>
int launch(const ConvP& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t smem = (size_t)2 * (BM + BN) * LDB * sizeof(float);
    static_assert(smem <= 160 * 1024, "LDS stages do not fit");
    constexpr bool OPS = EA == 0 && WM * WN == 4 && ((BM == 128 && BN == 128) || (BM == 64 && BN == 64));
    if (!OPS && (p.d.ld_bias > 0 || p.d.dual_n0 > 0)) { fgt_set_error("fgt_conv2d: this conv_split tile is built without bias-map / two-headed epilogues (use 128x128 or 64x64)"); return FGT_EINVAL; }
    if (p.d.ps_phase_pad && p.d.ps_c % BN != 0) { fgt_set_error("fgt_conv2d: ps_phase_pad needs a tile whose N width (%d) divides ps_c (%d)", BN, p.d.ps_c); return FGT_EINVAL; }
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_split_kernel<BM, BN, WM, WN, MINW, EA>), (int)smem, lds_set, "conv_split")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_split_kernel<BM, BN, WM, WN, MINW, EA>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_split");
}

}  // namespace

#ifdef FGT_DIAG
int fgt_conv_split_diag_launch(int tile, const ConvP& p, hipStream_t s);   // csrc/diag/conv_split_variants.hip (diagnostic builds only)
#endif

int fgt_conv_split_launch(int tile, const ConvP& p, hipStream_t s) {
#ifdef FGT_DIAG
    // diagnostic builds: FGT_DIAG_TILES_ALL=1 sends the product tiles to their instrumented twins as well (tools/conv_trace.py)
    static const bool all_diag = [] { const char* e = getenv("FGT_DIAG_TILES_ALL"); return e && atoi(e) != 0; }();
    if (all_diag) return fgt_conv_split_diag_launch(tile, p, s);
#endif
    switch (tile) {
        case FGT_TILE_128x128: return launch<128, 128, 2, 2>(p, s);
        case FGT_TILE_128x64: return launch<128, 64, 2, 2>(p, s);
        case FGT_TILE_64x64: return launch<64, 64, 2, 2>(p, s);
        case FGT_TILE_128x32: return launch<128, 32, 4, 1>(p, s);
        case FGT_TILE_256x128: return launch<256, 128, 4, 2>(p, s);
        case FGT_TILE_128x128x8: return launch<128, 128, 2, 4, 4>(p, s);
        case FGT_TILE_256x128x16: return launch<256, 128, 4, 4, 4>(p, s);
        case FGT_TILE_256x64x8: return launch<256, 64, 4, 2, 2>(p, s);
        // early stage release (two tiles in flight on two stages): the same tiles again, bit-identical results
        case FGT_TILE_128x128_EA: return launch<128, 128, 2, 2, 2, 1>(p, s);
        case FGT_TILE_128x64_EA: return launch<128, 64, 2, 2, 2, 1>(p, s);
        case FGT_TILE_64x64_EA: return launch<64, 64, 2, 2, 2, 1>(p, s);
        case FGT_TILE_128x128x8_EA: return launch<128, 128, 2, 4, 4, 1>(p, s);
        case FGT_TILE_256x128x16_EA: return launch<256, 128, 4, 4, 4, 1>(p, s);
        case FGT_TILE_256x64x8_EA: return launch<256, 64, 4, 2, 2, 1>(p, s);
        default:
#ifdef FGT_DIAG
            return fgt_conv_split_diag_launch(tile, p, s);
#else
            fgt_set_error("fgt_conv2d: unknown tile %d (the ring / ping-pong / 8-phase / loader-wavefront variants are in diagnostic builds only)", tile);
            return FGT_EINVAL;
#endif
    }
}
