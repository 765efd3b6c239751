// Round 4.  The tap-reusing bf16x3 convolution (conv_taps.hip) with the requests of a step INTERLEAVED into its matrix work (gfx950).
//
// conv_taps.hip's step is a serial chain per wavefront — fragment reads | barrier | LDS-DMA issue | MFMAs | wait | barrier — that only OTHER
// wavefronts fill (4 per SIMD at <= 128 registers: the matrix pipe ends 52-58 % busy).  What this round measured about the links of that chain
// (NOTEBOOK §11.3; tools/pp_trace.py, tools/micro/dma_issue.hip):
//   * hipcc, left to itself, hoists ALL fragment reads of a step above its first MFMA and waits lgkmcnt(0): 700 cycles of LDS round trips in
//     front of 768 cycles of MFMAs.  Reads as inline assembly with hand-counted lgkmcnt, one sub-step (12 MFMAs) ahead, run UNDER the MFMAs:
//     36 cycles per MFMA (32 = the pipe's rate).
//   * an LDS-DMA instruction is accepted in ~24 cycles while <= 32 are outstanding on the CU and in 200-600 cycles when every wavefront
//     issues its share at once behind a barrier (the queue is full; the CU retires one 16-row x 64-byte piece per ~37 cycles from L2).  Spread
//     over the step — one piece per ~6-9 MFMAs of a wavefront — the queue never fills and the issue sits in the shadow of the MFMAs.
//   * with the requests of step s+1 issued DURING step s (behind its first sub-steps) and waited for at its end, ONE barrier per step is
//     enough on two stages: what a step reads was published by the barrier before it, what it requests is read after the barrier behind it.
//   * what remains per step is the ~500-600 cycles from the barrier to the first MFMA (every wavefront's first 8 ds_read_b128 at once: 64 KB
//     through the LDS) and ~300 cycles of tail + barrier.  Two workgroups per CU (the 128-row tile) fill these gaps for each other; the
//     256-row tiles (one workgroup per CU, half the bytes per flop) do not.
// The first design of this file split the 8 wavefronts of a 256-row tile into two ping-pong groups (one computes a whole step while the other
// requests): six versions, all within -50 ... +3 % of conv_taps.hip — a wavefront that requests 10 pieces at once waits 2 000-3 000 cycles for
// the queue whoever its partner is (profiles/r04_run5...12_pp_trace*.txt, git history).
//
// Tiles: BM x BN on BM / 32 wavefronts of 64 x 64 (BN = 128) or 128 x 64 (BN = 256):
//   128 x 128, 4 wavefronts, 68 KB of LDS: two workgroups per CU        ("128x128it")
//   256 x 128, 8 wavefronts, 100 KB; 256 x 256, 8 wavefronts, 132 KB: one workgroup per CU  ("256x128it", "256x256it")
// Step (kx tap of a (32-channel chunk, ky) super-step) of every wavefront:
//   (before the barrier: the first A fragments — their A buffer was published at least a barrier ago) | barrier | 4 B reads | 24 (48) MFMAs with
//   AT MOST ONE filler behind each — a ds_read_b128 of the next sub-step, or one block of the next step's fragment addresses — and ONE LDS-DMA
//   request behind every fourth (the B(s+1) pieces first, then A pieces of the next (ky, chunk)) | wait for this step's B requests | barrier
// What the per-wavefront timeline (tools/pp_trace.py) said along the way, cycles per step of the 128 x 128 tile at two workgroups per CU:
//   requests all behind the first 12 MFMAs 3 260 -> one per four MFMAs 2 950 -> wide image 2 980 (the DMA engine was not the bound) -> first A
//   reads before the barrier + first MFMA group waiting for 2 reads instead of 8 (barrier -> first MFMA 560 -> 235) 2 850 -> fillers one per
//   MFMA instead of a burst behind each group (a lone wavefront hides ~5 issue slots per MFMA, a burst leaves the pipe idle) 2 570; no
//   compare-and-branch chain in front of the vmcnt wait: another 2 % in the sweeps.  A wavefront ALONE on its SIMD (one workgroup per CU, no
//   DMA) still needs 1 000-1 100 cycles for its 24 MFMAs (768) + ~900 for tail / barrier / first reads: the two wavefronts of a SIMD fall into
//   lock step (period ~ 2 x MFMAs + the rest) rather than filling each other's gaps; s_setprio for the second half of a step did not break it.
// Normal mode only (the reused taps are the kx taps of a stride-1 "same" convolution): the transposed k x 1 mode and the nearest-x2 upsampling
// stay on conv_taps.hip's tiles (fgt_conv_taps_il_launch declines them).  The request path is written for few live scalars: its first version
// kept conv_taps.hip's generality and executed ~200 v_readlane reloads of spilled SGPRs per step.
//
// Numerics: the products and the accumulation order of conv_taps.hip ((chunk, ky, kx) since round 5; per accumulator and k-half lo*hi, hi*lo, hi*hi):
// BIT-IDENTICAL to its tiles (tests/test_taps_gpu.py), so the autotuner chooses among all of them (routing stays by geometry).
// LDS: B stages [2][hi BN | lo BN] x 64-byte rows, A buffers [2][hi: BM + 16 rows + zero row | lo: ...].
//
// WIDE image (interleaved split inputs, in_split = 2 — what the bf16x3 pipeline produces for C % 32 == 0; weights are per-step interleaved lines
// anyway): every LDS row is the full 128-byte line [hi 64 | lo 64] of a pixel's / an output channel's 32-channel chunk, requested as 8-row x
// 128-byte pieces — the CU retires such a piece in 19 cycles against 37 for a 16-row x 64-byte one.  16-byte slots are XOR-swizzled by
// (row >> 1) & 7 on the source side (a lane fetches the slot that belongs at its linear LDS position), which makes every ds_read_b128 lane
// group hit 16 distinct slots of the 256-byte bank window for ANY tap shift.  Same values, same products: bit-identical to the plane image
// (tests/test_taps_gpu.py runs both).  Measured +1 ... +13 % over the plane image (profiles/r04_run27_taps_il_wide_sweep.txt).  Off in round 5
// (run-to-run differences on a shared GPU), on again since round 6: the cause was a compiler-placed register copy, see retire_pre_reads.
// A GEMM mode of this kernel (KW = 1: 1 x 1 layers, A rows requested with the B tile of every step) was built, measured on the K = 512 / 768
// linear layers of the transformer and dropped: 237-280 TFLOP/s against 253-309 of conv_split / conv_wide — with 16-24 steps per tile those
// layers are bound by the tile's prologue and its 64 KB output, not by the step (NOTEBOOK §11).
#include "conv_tile.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int MAXA> __device__ __forceinline__ void wait_vmcnt_upto(int na) {     // vmcnt(na), na wave-uniform in [0, MAXA]
    if constexpr (MAXA == 0) wait_vmcnt<0>();
    else {
        if (na == MAXA) wait_vmcnt<MAXA>();
        else wait_vmcnt_upto<MAXA - 1>(na);
    }
}

#ifndef FGT_IL_A_EARLY
#define FGT_IL_A_EARLY 2          // A request schedule: 0 spread / 1 all in tap 0; 2: the default per instance (below)
#endif
constexpr int HALO = 16;          // extra A rows per (ky, chunk): (kw - 1) * dw <= 16

#ifdef FGT_PP_TRACE
// Diagnostic builds only (fgt_amd.build.build(variant="pptrace", extra_flags=["-DFGT_PP_TRACE"]), tools/pp_trace.py): s_memtime stamps of one
// workgroup's wavefronts over the first PP_TR_STEPS steps: [wave][step][0: step top, 5: first MFMA about to issue, 6: last MFMA issued,
// 3: requests waited for (before the barrier), 4: behind the barrier].  FGT_CONV_PIPE=2: no LDS-DMA in the loop (timing only).
constexpr int PP_TR_STEPS = 24;
__device__ unsigned long long fgt_pp_trace_buf[8 * PP_TR_STEPS * 10];
#define PP_STAMP(slot)                                                                                       \
    do {                                                                                                     \
        if (blockIdx.x == 8 && blockIdx.y == 0 && tr_step < PP_TR_STEPS) {                                   \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                      \
            if (lane == 0) fgt_pp_trace_buf[(wave * PP_TR_STEPS + tr_step) * 10 + (slot)] = t_;               \
        }                                                                                                    \
    } while (0)
#else
#define PP_STAMP(slot) do {} while (0)
#endif

template <int BM, int BN, int KW, bool WIDE>
__global__ void __launch_bounds__(BM * 2, 2) conv_taps_il_kernel(const ConvP p) {
    constexpr int NW = BM / 32;                          // 8 (BM = 256) or 4 (BM = 128) wavefronts
    constexpr int WN = BN / 64, WM = NW / WN;            // 256x256: 2 x 4 wavefronts of 128x64; 256x128: 4 x 2 of 64x64; 128x128: 2 x 2 of 64x64
    constexpr int WTM = BM / WM, WTN = 64, TM = WTM / 32, TN = 2;
    constexpr int NIH = TM / 2, NU = 2 * NIH;            // sub-steps per step: (k-half, pair of 32-row blocks)
    constexpr int HPS = BN == 256 ? 1 : 3;               // request slots per sub-step: behind every 4 MFMAs, or (256x256: registers) every 12
    constexpr int AR = BM + HALO;                        // A rows per plane filled by DMA; row AR is the zero row
    constexpr int APL = (AR + 1) * 64;                   // bytes per A plane
    constexpr int GB = BN / 16;                          // 16-row DMA groups per B plane
    constexpr int A_BYTES = 2 * APL, B_BYTES = 2 * BN * 64;
    constexpr int LDS_BYTES = 2 * B_BYTES + 2 * A_BYTES;
    constexpr int STAGE = LDS_BYTES / 8;                 // floats in half of the LDS (the epilogue's view of its scratch)
    constexpr int NGA = BM / 16 / NW;                    // A row groups (of both planes) a wavefront owns: 2
    constexpr int APW = 2 * NGA + 1;                     // A pieces a wavefront may own per (ky, chunk): 2 groups x 2 planes + the halo group (last two waves);
                                                         // WIDE: 4 groups of 8 full rows + a halo group (last two waves)
    // Schedule of a wavefront's A requests for the next (ky, chunk), ASCHED:
    //   0  piece it in tap it % (KW-1); a tap's pieces may still fly at its end, the last land by the end of tap KW-1 — so the FIRST step of a
    //      super-step reads its A fragments behind the barrier, the others before it (PRE);
    //   1  all in tap 0, landed by the end of tap 1 and published by ITS barrier: every step reads its first A fragments before the barrier
    //      in front of it.  (9 requests per wavefront in one step overrun the CU's DMA queue for a while; a third schedule — spread as in 0,
    //      the pieces of tap KW-2 first in their step and waited for at its end — was measured 3 % slower than this one (run 38), the A
    //      requests behind the step's last MFMA instead of in its request slots 1-3 % slower (run 41).)
    // Default: 0 on the 64-byte image (narrow pieces: more requests in a step cost more than the early reads buy), 1 on the wide image.
    // (Round 5 saw schedule 1 differ from run to run when three processes shared the GPU — 1-5 of 150 launches on the 128-row tile, 1 of 450 passes
    // of the step on the 256 x 128 one — and switched it and the wide image off without a cause.  The cause was not in the schedule: see
    // retire_pre_reads below.)
    constexpr int ASCHED = FGT_IL_A_EARLY != 2 ? FGT_IL_A_EARLY : (WIDE ? 1 : 0);
    constexpr bool A_EARLY = ASCHED != 0;                 // every step's first A fragments are read before the barrier in front of it
    constexpr int ASTEPS = ASCHED == 1 ? 1 : KW - 1;
    constexpr int MAXA = (APW + ASTEPS - 1) / ASTEPS;    // most A pieces a wavefront requests in one step
    constexpr int BPP = GB / NW;                         // B pieces per plane and wavefront (256x256: 2, 256x128: 1, 128x128: 2)
    static_assert((HPS == 3 || NU == 4) && (BM == 256 || BM == 128) && (BN == 128 || BN == 256) && NGA == 2 && BPP >= 1 && GB % NW == 0 && KW >= 3 && TM % 2 == 0 && WM * WN == NW,
                  "tile / wavefront geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    char* const lds = reinterpret_cast<char*>(smem);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)lds;     // LDS byte address of the dynamic segment
    char* const Bst = lds;                               // [2][hi BN rows | lo BN rows]
    char* const Abuf = lds + 2 * B_BYTES;                // [2][hi AR rows, zero row | lo AR rows, zero row]
    if constexpr (WIDE) {                                // [2][AR rows, zero row] x 128 bytes (hi 64 | lo 64, 16-byte slots swizzled by (row >> 1) & 7)
        if (tid < 64) reinterpret_cast<float*>(Abuf + (tid >> 5) * A_BYTES + AR * 128)[tid & 31] = 0.f;
    } else {
        if (tid < 64) reinterpret_cast<float*>(Abuf + (tid >> 5) * A_BYTES + ((tid >> 4) & 1) * APL + AR * 64)[tid & 15] = 0.f;
    }

    const int W = d.W, H = d.H, HW = H * W;
    const int dwx = d.dw, p_i = d.pw;
    const bool il = d.in_split == 2;
    const int nch0 = p.Cg0 / 32, nch1 = p.Cg1 / 32, nchunk = nch0 + nch1;
    // ABI 7 (desc.ky_skip_n0): the weights of this tile's columns are all zero for ky = 0 — its K walk starts at ky = 1 (fold as a convolution:
    // the sub-pixel rows ry >= 1 of a token cell receive nothing from the token row above)
    const int ky0 = (d.ky_skip_n0 > 0 && bn0 >= d.ky_skip_n0) ? 1 : 0;
    const int n_o = d.kh - ky0;                          // ky taps walked
    const int nss = n_o * nchunk;
    const int cstride = il ? 128 : 64;                   // bytes from one 32-channel chunk of a pixel to the next

    // ---- im2col source iterator (wave-uniform): the super-step whose A rows are requested next (conv_taps.hip, normal mode)
    auto src_hi = [&](int s) {
        const __bf16* x = reinterpret_cast<const __bf16*>(s ? p.x1 : p.x0);
        const long c0 = s ? (long)d.off1 + (long)g * p.Cg1 : (long)d.off0 + (long)g * p.Cg0;
        return reinterpret_cast<const char*>(x + (il ? 2 * c0 : c0));
    };
    auto src_lo_off = [&](int s) { return il ? 64l : 2 * (s ? p.ps1 : p.ps0); };
    const char* a_hi = src_hi(0);
    const char* a_lo = a_hi + src_lo_off(0);
    int a_ld2 = 2 * d.ld0, a_left = nch0, a_src = 0;      // (a_ld2: bytes from one pixel of the source to the next)
    const int a_dy0 = ky0 * d.dh - d.ph;
    int a_dy = a_dy0, a_dyW = a_dy * W;                   // ky tap shift: in rows / in pixels
    int a_o = 0;                                          // ky tap of the super-step the A stream is in
    auto a_advance = [&]() {                              // next ky of the same chunk; behind the last one: the next chunk (never wraps)
        a_dy += d.dh; a_dyW += d.dh * W;
        if (++a_o < n_o) return;
        a_o = 0; a_dy = a_dy0; a_dyW = a_dy0 * W;
        a_hi += cstride; a_lo += cstride;
        if (--a_left == 0 && a_src == 0 && nch1 > 0) {
            a_src = 1; a_left = nch1; a_ld2 = 2 * d.ld1;
            a_hi = src_hi(1);
            a_lo = a_hi + src_lo_off(1);
        }
    };

    // ---- A pieces (16 rows x 64 bytes of one plane): wavefront w owns, of BOTH planes, the row groups w and w + NW (pieces it = 0..3: plane
    // it >> 1, group w + NW * (it & 1)) and — the last two wavefronts — the halo group BM / 16 of plane 0 / 1 (piece 4).  A lane fetches row
    // (lane >> 2) of a group, 16-byte column kc (swizzled on the source side): three (pixel, image row) pairs per lane describe all five pieces.
    // WIDE (interleaved inputs: a pixel's 32-channel chunk is ONE 128-byte line [hi | lo]): pieces of 8 rows x 128 bytes — the CU retires
    // such a piece in about half the time of a 16-row x 64-byte one (tools/micro/dma_issue.hip).  Wavefront w owns the 8-row groups w, w + NW,
    // w + 2 NW, w + 3 NW (pieces 0..3) and — the last two wavefronts — the halo groups BM / 8 and BM / 8 + 1 (piece 4; the second one only when
    // (kw - 1) * dw > 8).  A lane fetches row (lane >> 3), 16-byte slot (lane & 7) ^ ((LDS row >> 1) & 7): every group of one wavefront has
    // the same parity, so the source-side swizzle is one per-lane constant.
    constexpr int NPA = WIDE ? 5 : 3;                     // (pixel, image row) pairs per lane
    const int lrow = WIDE ? lane >> 3 : lane >> 2;
    const int kc16 = WIDE ? ((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) * 16 : ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    const bool halo2 = (KW - 1) * dwx > 8;
    int a_q[NPA], a_y[NPA];
#pragma unroll
    for (int c = 0; c < NPA; ++c) {
        const int grp = WIDE ? (c < 4 ? wave + NW * c : BM / 8 + (wave - (NW - 2))) : (c < 2 ? wave + NW * c : BM / 16);
        const long q = (long)bm0 - p_i + grp * (WIDE ? 8 : 16) + lrow;           // flattened (n, y, x) index of the LDS row for the centre taps
        const bool valid = q >= 0 && q < (long)d.N * HW;
        const int rem = valid ? (int)(q % HW) : 0;
        a_q[c] = valid ? (int)q : 0;
        a_y[c] = valid ? rem / W : -(1 << 30);
    }
    const char* const zp = reinterpret_cast<const char*>(p.zero_page);
#ifdef FGT_PP_TRACE
    const int abl = p.pipe;
    int tr_step = 0;
#else
    constexpr int abl = 1;
#endif
    auto issue_A = [&](auto IT, int ab) __attribute__((always_inline)) {
        constexpr int it = decltype(IT)::value, c = WIDE ? it : (it < 4 ? (it & 1) : 2);
        if (abl == 2) return 0;
        if constexpr (it == 4) { if (wave < NW - 2) return 0; }  // (wave-uniform)
        if constexpr (WIDE) {
            if constexpr (it == 4) { if (wave == NW - 1 && !halo2) return 0; }
            const int grp = it < 4 ? wave + NW * it : BM / 8 + (wave - (NW - 2));
            const char* ptr = a_hi + ((long)(a_q[c] + a_dyW) * a_ld2 + kc16);
            const bool ok = (unsigned)(a_y[c] + a_dy) < (unsigned)H;
            glds16(ok ? ptr : zp, Abuf + ab * A_BYTES + grp * 1024);
            return 1;
        }
        const int plane = it < 4 ? it >> 1 : wave & 1;
        const int grp = it < 4 ? wave + NW * (it & 1) : BM / 16;
        const char* ptr = (plane ? a_lo : a_hi) + ((long)(a_q[c] + a_dyW) * a_ld2 + kc16);
        const bool ok = (unsigned)(a_y[c] + a_dy) < (unsigned)H;
        glds16(ok ? ptr : zp, Abuf + ab * A_BYTES + plane * APL + grp * 1024);
        return 1;
    };

    // ---- weights: interleaved rows [Kpad/32][hi 32 | lo 32], K-step order as in conv_taps.hip: kstep(ky, c, kx) = (ky*KW + kx) * nchunk + c, walked (chunk, ky, kx).
    // One per-lane 64-bit base + a wave-uniform byte offset per piece; the K position is a scalar running offset (32 bits: it stays inside a row).
    // A wavefront requests the 16-row groups BPP * wave .. BPP * wave + BPP - 1 of both planes.
    // WIDE: 8-row groups of full 128-byte lines; wavefront w requests the groups 2 BPP w .. 2 BPP w + 2 BPP - 1 (request r: parity r & 1, which
    // flips bit 2 of the swizzled slot = 64 bytes)
    const int kcB = WIDE ? ((lane & 7) ^ (lane >> 4)) * 16 : kc16;
    const char* const w_lane = reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(p.w) + ((long)g * d.Npad + bn0 + lrow) * (2 * d.Kpad)) + kcB;
    const int w_row16 = (WIDE ? 32 : 64) * d.Kpad;        // bytes from one 16-row (WIDE: 8-row) group of the weight image to the next
    const int dkx = nchunk * 128;                         // to the next kx of a (chunk, ky) and from its last kx to the next ky of the chunk
    const int dchunk = 128 - (n_o * KW - 1) * dkx;        // from the last step of a chunk to the first step of the next chunk
    int w_k = ky0 * KW * dkx;                             // byte offset of the K-step the B stream is at
    int b_c = 0;                                          // ky tap (within its chunk) of the super-step the B stream is in
    const int npad_rows = d.Npad - bn0;                   // weight rows of this tile that exist (the rest reads the zero page)
    auto issue_B = [&](auto RQ, int bs) __attribute__((always_inline)) {           // request r of 2 BPP: (plane, group) / WIDE: 8-row group
        constexpr int r = decltype(RQ)::value;
        if (abl == 2) return;
        if constexpr (WIDE) {
            const int grp = wave * (2 * BPP) + r;
            const bool ok = grp * 8 < npad_rows;          // (wave-uniform)
            const unsigned long src = (unsigned long)(w_lane + ((long)grp * w_row16 + w_k)) ^ (unsigned long)((r & 1) * 64);     // (lines are 128-byte aligned)
            glds16(ok ? reinterpret_cast<const char*>(src) : zp, Bst + bs * B_BYTES + grp * 1024);
        } else {
            constexpr int plane = r / BPP, i = r % BPP;
            const int grp = wave * BPP + i;
            const bool ok = grp * 16 < npad_rows;         // (wave-uniform)
            glds16(ok ? w_lane + ((long)grp * w_row16 + plane * 64 + w_k) : zp, Bst + bs * B_BYTES + plane * BN * 64 + grp * 1024);
        }
    };
    auto advance_B = [&](bool last_kx) {                  // behind the B tile of a step with kx = KW-1 (last_kx) or kx < KW-1
        int dlt = dkx;
        if (last_kx && ++b_c == n_o) { b_c = 0; dlt = dchunk; }
        w_k += dlt;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    int Rb[TM], oxp[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        Rb[i] = wm * WTM + i * 32 + l31;
        oxp[i] = (bm0 + Rb[i]) % W - p_i;
    }
    // B fragment rows: wave-tile base (multiple of 32) + l31; WIDE: 128-byte rows, slot (plane*4 + khalf*2 + lh) ^ ((l31 >> 1) & 7)
    const unsigned b_lane = WIDE ? (unsigned)((wn * WTN + l31) * 128 + ((lh ^ ((l31 >> 1) & 7)) * 16)) : (unsigned)((wn * WTN + l31) * 64);
    const unsigned so0 = (unsigned)swz(l31, lh) * 2u, so1 = (unsigned)swz(l31, 2 + lh) * 2u;

    // ---- prologue: A rows of super-step 0 and the B tile of step 0, landed and published
    static_for<APW>([&](auto IT) __attribute__((always_inline)) { issue_A(IT, 0); });
    a_advance();
    static_for<2 * BPP>([&](auto RQ) __attribute__((always_inline)) { issue_B(RQ, 0); });
    advance_B(false);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- fragment addresses of a step (tap kx of A buffer Ab, B stage bs): computed for the NEXT step behind the current step's last MFMAs
    unsigned fa_a0[TM], fa_bb = 0;
    int fa_sh = 0;                                        // the tap shift the parts below work with
    auto frag_addr_part = [&](auto PARTc, int kx, unsigned Ab, int bs) __attribute__((always_inline)) {    // part i: A block i (part 0 also: shift, B)
        constexpr int i = decltype(PARTc)::value;
        if constexpr (i == 0) {
            // (opaque: the per-tap fragment addresses are recomputed per step, not hoisted and kept live across the loop)
            int shv = kx * dwx;
            asm volatile("" : "+v"(shv));
            fa_sh = __builtin_amdgcn_readfirstlane(shv);
            fa_bb = lds0 + (unsigned)(bs * B_BYTES) + b_lane;
        }
        const bool xin = (unsigned)(oxp[i] + fa_sh) < (unsigned)W;
        const int R = xin ? Rb[i] + fa_sh : AR;
        if constexpr (WIDE) fa_a0[i] = lds0 + Ab + (unsigned)R * 128u + (unsigned)(((R >> 1) & 7) ^ lh) * 16u;   // k-half 1: ^ 32 bytes, lo: ^ 64 bytes
        else fa_a0[i] = lds0 + Ab + (unsigned)R * 64u + (unsigned)(((R >> 2) & 3) ^ lh) * 16u;  // k-half 0: slot lh; k-half 1: slot 2 + lh (^ 32 bytes)
    };
    auto frag_addr = [&](int kx, unsigned Ab, int bs) __attribute__((always_inline)) {
        static_for<TM>([&](auto I) __attribute__((always_inline)) { frag_addr_part(I, kx, Ab, bs); });
    };
    auto rd = [&](bf16x8& dst, unsigned addr) __attribute__((always_inline)) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
        dst = __builtin_bit_cast(bf16x8, v);
    };
    // ---- one step of one wavefront's matrix work: TM x TN x 6 MFMAs in sub-steps of 12 (k-half, pair of 32-row blocks), three groups of four
    // (lo*hi, hi*lo, hi*hi).  Fragment reads are inline assembly with hand-counted waits (hipcc alone hoists every read of a step above its
    // first MFMA and waits for all of them): LDS returns in order, so lgkmcnt(n) with n = the reads requested after the ones needed retires
    // exactly those; the wait carries the fragments as "+v" operands, so no MFMA can be scheduled above it.  Order of a step:
    //   (before the barrier, in the previous step)  A fragments of sub-step 0 — their A buffer was published a barrier ago (PRE)
    //   barrier | B fragments of k-half 0: hi, hi, lo, lo | wait hi | 4 MFMAs lo*hi | request slot | reads of the NEXT sub-step (B of the
    //   next k-half first, then A lo, lo, hi, hi) | wait lo | 4 + 4 MFMAs, a request slot behind each | next sub-step: wait for its A lo
    //   (all but the 2 youngest reads) ... behind the first group of the LAST sub-step: the next step's fragment addresses and its first A reads.
    // So the first MFMA of a step waits for two ds_read_b128 behind the barrier instead of eight (+ eight more queued in front of the wait).
    bf16x8 Bh[2][TN], Bl[2][TN];                          // [k half][32-column block]
    bf16x8 Ah[2][2], Al[2][2];                            // [register set][block of the pair]
    // read q of a B k-half (hi, hi, lo, lo) / of an A sub-step (lo, lo, hi, hi)
    auto readB = [&](auto KS, auto Qc, unsigned bb) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value, j = decltype(Qc)::value % TN, lo = decltype(Qc)::value / TN;
        if constexpr (WIDE) rd(lo ? Bl[ks][j] : Bh[ks][j], (bb ^ (unsigned)((ks ? 32 : 0) + (lo ? 64 : 0))) + j * 32 * 128);
        else rd(lo ? Bl[ks][j] : Bh[ks][j], bb + (ks ? so1 : so0) + (lo ? BN * 64 : 0) + j * 32 * 64);
    };
    auto readA = [&](auto U, auto Qc, const unsigned (&a0)[TM]) __attribute__((always_inline)) {
        constexpr int u = decltype(U)::value, ks = u / NIH, ih = u % NIH, set = u & 1;
        constexpr int ii = decltype(Qc)::value % 2, hi = decltype(Qc)::value / 2;
        const unsigned a = a0[2 * ih + ii] ^ (ks ? 32u : 0u);
        rd(hi ? Ah[set][ii] : Al[set][ii], hi ? a : (WIDE ? a ^ 64u : a + APL));
    };
    auto loadB = [&](auto KS, unsigned bb) __attribute__((always_inline)) {
        static_for<2 * TN>([&](auto Q) __attribute__((always_inline)) { readB(KS, Q, bb); });
    };
    auto loadA = [&](auto U, const unsigned (&a0)[TM]) __attribute__((always_inline)) {
        static_for<4>([&](auto Q) __attribute__((always_inline)) { readA(U, Q, a0); });
    };
    // PRE: this step's first A fragments were requested before the barrier (by the previous step / the prologue); nxt(part): the next step's
    // fragment addresses, one A block per part.  Every MFMA is followed by AT MOST one ds_read_b128 or one address part (a wavefront that is
    // alone on its SIMD hides ~5 issue slots per MFMA; a burst behind a group of MFMAs leaves the pipe idle), a request slot behind every fourth.
    auto compute = [&](auto PREc, auto PRENc, auto&& hook, auto&& nxt) __attribute__((always_inline)) {
        constexpr bool PRE = decltype(PREc)::value, PRE_NEXT = decltype(PRENc)::value;
        unsigned a0[TM];
        static_for<TM>([&](auto I) __attribute__((always_inline)) { a0[decltype(I)::value] = fa_a0[decltype(I)::value]; });
        const unsigned bb = fa_bb;
        if constexpr (!PRE) loadA(std::integral_constant<int, 0>{}, a0);
        loadB(std::integral_constant<int, 0>{}, bb);
        static_for<NU>([&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, ks = u / NIH, ih = u % NIH, set = u & 1;
            constexpr bool more = u + 1 < NU, nextB = more && (u + 1) % NIH == 0;
            constexpr int NB = nextB ? 2 * TN : 0;                                      // reads behind MFMA m: m < NB: B of the next k-half; then 4 A reads
            constexpr int A0 = more ? NB : 4;                                           // (last sub-step: address parts behind MFMAs 0..TM-1, reads behind 4..7)
            constexpr int NRD = more ? NB + 4 : (PRE_NEXT ? 8 : 0);
            static_assert(TM <= 4 && NRD <= 8 && 12 == 3 * 2 * TN, "filler slots of a sub-step");
            // operands of the first group: A lo of this sub-step and B hi of its k-half — everything but the 2 youngest reads
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(Al[set][0]), "+v"(Al[set][1]), "+v"(Bh[ks][0]), "+v"(Bh[ks][1]));
            if constexpr (u == 0) PP_STAMP(5);
            static_for<12>([&](auto Mc) __attribute__((always_inline)) {
                constexpr int m = decltype(Mc)::value, prod = m / 4, ii = (m % 4) / TN, j = m % TN;
                if constexpr (m == 4) {                                                 // hi*lo needs A hi and B lo: all but the reads behind MFMAs 0..3
                    constexpr int pending = more ? 4 : 0;
                    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(Ah[set][0]), "+v"(Ah[set][1]), "+v"(Bl[ks][0]), "+v"(Bl[ks][1]) : "n"(pending));
                }
                // same products as conv_taps.hip / conv_split.hip (lo*hi, hi*lo, hi*hi per accumulator and k-half)
                const bf16x8 av = prod == 0 ? Al[set][ii] : Ah[set][ii];
                const bf16x8 bv = prod == 1 ? Bl[ks][j] : Bh[ks][j];
                acc[2 * ih + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[2 * ih + ii][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                                      // the filler goes BEHIND this MFMA
                if constexpr (more) {
                    if constexpr (m < NB) readB(std::integral_constant<int, (u + 1) / NIH>{}, Mc, bb);
                    else if constexpr (m < NB + 4) readA(std::integral_constant<int, u + 1>{}, std::integral_constant<int, m - NB>{}, a0);
                } else {
                    if constexpr (m < TM) nxt(Mc);
                    if constexpr (PRE_NEXT && m >= 4 && m < 8) readA(std::integral_constant<int, 0>{}, std::integral_constant<int, m - 4>{}, fa_a0);
                }
                if constexpr (m % 4 == 3 && (HPS == 3 || m == 11)) hook(std::integral_constant<int, HPS == 3 ? u * 3 + prod : u>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    constexpr int NHOOK = NU * HPS;                       // request slots per step
    constexpr int NREQ = 2 * BPP + MAXA;                  // requests a wavefront may make per step: B pieces of both planes, then A pieces

    int bs = 0;                                           // B stage of this step
    frag_addr(0, (unsigned)(2 * B_BYTES), 0);
    // Fragment reads in flight must never cross a loop edge.  hipcc counts an asm statement's VGPR destination as written at ;;#ASMEND: where the
    // registers of A-fragment set 0 are loop-carried it places PHI copies (v_mov_b64 of the ds_read_b128 destinations) in the preheader and on the
    // back-edge of the super-step loop — AHEAD of the s_waitcnt that covers the reads.  A copy takes whatever the register holds: the fragment when
    // the LDS answered within the ~60 instructions in between (a lone process: every bit-equality test of two rounds), stale bits when other work
    // shares the CU's LDS (1-5 of 150 launches with three processes on the GPU).  That was round 5's "not reproducible on a shared GPU" of the early
    // schedule (NOTEBOOK §13.1; tools/asm_hazard_audit.py finds the copies in the ISA and tests/test_build_resources.py runs it over every object
    // with asm reads).  So: the pre-loop reads and the pre-barrier reads of a super-step's LAST step are retired before the edge — they were
    // issued four MFMAs and the request wait ago; the pre-barrier reads of the other steps cross straight-line code only.
    auto retire_pre_reads = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Al[0][0]), "+v"(Al[0][1]), "+v"(Ah[0][0]), "+v"(Ah[0][1]));
    };
    if constexpr (A_EARLY) { loadA(std::integral_constant<int, 0>{}, fa_a0); retire_pre_reads(); }
    for (int ss = 0; ss < nss; ++ss) {
        const bool last = ss + 1 == nss;
        const unsigned Ab = (unsigned)(2 * B_BYTES + (ss & 1) * A_BYTES);          // byte offset of this super-step's A buffer in the LDS
        static_for<KW>([&](auto KX) __attribute__((always_inline)) {
            constexpr int kx = decltype(KX)::value;
            const bool moreB = !(last && kx == KW - 1);
            int na = 0;
            // request slot h of this step: the B pieces of step s+1 first (they are read right behind the barrier), one per slot, then this
            // wavefront's A pieces of super-step ss+1 that belong to tap kx (pieces it = kx, kx + ASTEPS, ...)
            auto hook = [&](auto Hh) __attribute__((always_inline)) {
                constexpr int h = decltype(Hh)::value;
                // HPS == 3: requests r with r * NHOOK / NREQ == h; HPS == 1 (four slots): B plane 0 | B plane 1 | the A pieces | addresses only
                constexpr int r0 = HPS == 3 ? (h * NREQ + NHOOK - 1) / NHOOK : (h < 2 ? h * BPP : h == 2 ? 2 * BPP : NREQ);
                constexpr int r1 = HPS == 3 ? ((h + 1) * NREQ + NHOOK - 1) / NHOOK : (h < 2 ? (h + 1) * BPP : NREQ);
                static_for<r1 - r0>([&](auto RR) __attribute__((always_inline)) {
                    constexpr int r = r0 + decltype(RR)::value;
                    if constexpr (r < 2 * BPP) {
                        if (moreB) issue_B(std::integral_constant<int, r>{}, bs ^ 1);
                    } else {
                        constexpr int it = kx + (r - 2 * BPP) * ASTEPS;
                        if constexpr (kx < ASTEPS && it < APW) {
                            if (!last) na += issue_A(std::integral_constant<int, it>{}, (ss + 1) & 1);
                        }
                    }
                });
            };
            auto nxt = [&](auto PARTc) __attribute__((always_inline)) {  // the next step's fragment addresses, one A block per part
                constexpr int nkx = (kx + 1) % KW;
                const unsigned nAb = nkx == 0 ? (unsigned)(2 * B_BYTES + ((ss + 1) & 1) * A_BYTES) : Ab;
                frag_addr_part(PARTc, nkx, nAb, bs ^ 1);
            };
            PP_STAMP(0);
            compute(std::integral_constant<bool, (A_EARLY || kx > 0)>{}, std::integral_constant<bool, (A_EARLY || kx + 1 < KW)>{}, hook, nxt);
            PP_STAMP(6);
            if (moreB) advance_B((kx + 1) % KW == KW - 1);
            __builtin_amdgcn_sched_barrier(0);
            // the next step's B tile (and every older A piece) has landed; this step's A pieces may fly on.  (Taps without A requests: no
            // compare-and-branch chain — five taken branches cost ~150 cycles per step.)
            if constexpr (kx >= ASTEPS) wait_vmcnt<0>();
            else wait_vmcnt_upto<MAXA>(na);
            PP_STAMP(3);
            if constexpr (A_EARLY && kx == KW - 1) retire_pre_reads();                  // (the loop edge is behind this step: see above)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP(4);
            bs ^= 1;
#ifdef FGT_PP_TRACE
            ++tr_step;
#endif
        });
        a_advance();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN, KW == 5, KW == 5>(conv_epilogue_args(p), acc, smem, bm0, bn0, g);      // (bias maps / two heads: the 5-tap instances only, see conv_taps.hip)
}

template <int BM, int BN, int KW, bool WIDE>
int launch_kw_img(const ConvP& p, hipStream_t s) {
    constexpr size_t smem0 = (size_t)2 * (2 * BN * 64) + (size_t)2 * (2 * (BM + HALO + 1) * 64);
    static_assert(smem0 <= 160 * 1024, "LDS buffers do not fit");
#ifdef FGT_PP_TRACE
    static const size_t pad = [] { const char* e = getenv("FGT_IL_LDS_PAD"); return e ? (size_t)atoi(e) : 0; }();      // (occupancy experiments: one workgroup per CU)
    const size_t smem = smem0 + pad;
#else
    constexpr size_t smem = smem0;
#endif
    if (KW != 5 && (p.d.dual_n0 > 0 || p.d.ld_bias > 0)) { fgt_set_error("fgt_conv2d: bias maps / two-headed epilogues are built for layers with 5 reused taps (got %d)", KW); return FGT_EINVAL; }
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_taps_il_kernel<BM, BN, KW, WIDE>), (int)smem, lds_set, "conv_taps_il")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_taps_il_kernel<BM, BN, KW, WIDE>), grid, dim3(BM * 2), smem, s, q);
    return fgt_check_launch("conv_taps_il");
}

// interleaved split inputs (in_split = 2; weights are always per-step interleaved lines): the wide LDS image; plane inputs: 64-byte rows
template <int BM, int BN, int KW>
int launch_kw(const ConvP& p, hipStream_t s) {
    // FGT_TAPS_WIDE=0 selects the 64-byte image for interleaved inputs too (A/B measurements).  Round 5 shipped with the wide image off: its early
    // request schedule was not reproducible on a shared GPU; round 6 found the cause in the compiler's placement of PHI copies (retire_pre_reads
    // above), fixed it, and tools/asm_hazard_audit.py now proves the absence of that hazard class per build.
    static const bool wide = [] { const char* e = getenv("FGT_TAPS_WIDE"); return !(e && e[0] == '0'); }();
    return p.d.in_split == 2 && wide ? launch_kw_img<BM, BN, KW, true>(p, s) : launch_kw_img<BM, BN, KW, false>(p, s);
}

template <int BM, int BN>
int launch(const ConvP& p, hipStream_t s) {
    ConvP q = p;
    q.tr_li = 0;
    if (p.d.kw == 1 || p.d.upsample) {                    // k x 1 (transposed tile order) and nearest-x2 upsampling: conv_taps.hip's tiles
        fgt_set_error("fgt_conv2d: the interleaved-request tap tiles do not serve k x 1 or upsampling layers");
        return FGT_EINVAL;
    }
    if constexpr (BN == 256) {                           // (kw = 5, 7 on the 256x256 tile need more than 256 registers: not built)
        if (p.d.kw != 3) {
            fgt_set_error("fgt_conv2d: the 256x256 interleaved-request tap tile does not serve kw = %d layers", p.d.kw);
            return FGT_EINVAL;
        }
        return launch_kw<BM, BN, 3>(q, s);
    } else {
        switch (p.d.kw) {
            case 3: return launch_kw<BM, BN, 3>(q, s);
            case 5: return launch_kw<BM, BN, 5>(q, s);
            case 7: return launch_kw<BM, BN, 7>(q, s);
            default: fgt_set_error("fgt_conv2d: the tap-reusing kernel is built for 3, 5, 7 reused taps (got %d x %d)", p.d.kh, p.d.kw); return FGT_EINVAL;
        }
    }
}

}  // namespace

#if defined(FGT_PP_TRACE) && !defined(FGT_IL_PART)
extern "C" int fgt_debug_pp_trace(unsigned long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(fgt_pp_trace_buf), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#endif

// called by fgt_conv_taps_launch (conv_taps.hip) for the "...it" tile codes; same eligibility as the other tap tiles minus k x 1 / upsampling.
// Build time: the three tiles are separate translation units (this file = the 128 x 128 tile and the dispatcher; conv_taps_il_256x128.hip and
// conv_taps_il_256x256.hip include it with FGT_IL_PART = 1 / 2) — one unit took 11 minutes of hipcc with the epilogue's instances.  Trace builds
// (-DFGT_PP_TRACE: one stamp buffer) keep everything in this unit.
int fgt_conv_taps_il_256x128(const ConvP& p, hipStream_t s);
int fgt_conv_taps_il_256x256(const ConvP& p, hipStream_t s);
#if defined(FGT_PP_TRACE)
#if !defined(FGT_IL_PART)
int fgt_conv_taps_il_256x128(const ConvP& p, hipStream_t s) { return launch<256, 128>(p, s); }
int fgt_conv_taps_il_256x256(const ConvP& p, hipStream_t s) { return launch<256, 256>(p, s); }
#endif
#elif defined(FGT_IL_PART) && FGT_IL_PART == 1
int fgt_conv_taps_il_256x128(const ConvP& p, hipStream_t s) { return launch<256, 128>(p, s); }
#elif defined(FGT_IL_PART) && FGT_IL_PART == 2
int fgt_conv_taps_il_256x256(const ConvP& p, hipStream_t s) { return launch<256, 256>(p, s); }
#endif
#if !defined(FGT_IL_PART)
int fgt_conv_taps_il_launch(int bm, int bn, const ConvP& p, hipStream_t s) {
    if (bm == 128) return launch<128, 128>(p, s);
    return bn == 256 ? fgt_conv_taps_il_256x256(p, s) : fgt_conv_taps_il_256x128(p, s);
}
#endif
