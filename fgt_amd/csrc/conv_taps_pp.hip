// Round 4.  The tap-reusing bf16x3 convolution (conv_taps.hip) on 256-ROW tiles with two PING-PONG wavefront groups (gfx950).
//
// Why.  conv_taps.hip's 128x128 / 128x64 tiles are bound by the rate at which LDS-DMA delivers 1-KB pieces to a CU (NOTEBOOK "what bounds
// the conv kernel": 23 B/clk/CU in the kernel; a 128x128 step moves 22.5 KB for 768 SIMD-cycles of MFMAs = 30 B/clk at full matrix rate).
// A 256x256 tile moves 43 KB per step for 3 072 SIMD-cycles (14 B/clk) — but it is ONE workgroup per CU, and the step of the round-3
// kernels is a serial chain per wavefront (fragment reads | barrier | DMA issue | MFMAs | wait | barrier) that only OTHER workgroups
// fill: the 256-row instances of round 2 / 3 (8-phase, 16 wavefronts) lost what the smaller byte stream gained.
// Here the chain is cut differently:
//   * 8 wavefronts = two groups, G0 = waves 0-3 (tile rows 0-127), G1 = waves 4-7 (rows 128-255); each SIMD hosts one wavefront of each.
//   * (first version; the second dropped the mid-step barrier, see the main loop) a step (kx tap of a (ky, 32-channel chunk) super-step) is
//     TWO barrier intervals.  In the first G0 computes the WHOLE step — 48 MFMAs
//     (BN = 256: 1 536 cycles of its SIMD's matrix pipe) with its fragment reads software-pipelined underneath (sub-steps of 12 MFMAs, the
//     next sub-step's ds_read_b128 in flight: two register sets) — while G1 requests its share of the NEXT step's B tile and of the next
//     super-step's A rows; in the second interval the roles swap.  The matrix pipe of every SIMD always has exactly one wavefront feeding
//     it, the DMA issue (address arithmetic + back-pressure of the vector-memory pipe) sits in the partner's interval, and there are 2
//     barriers per 48 MFMAs per wavefront instead of 2 per 12.
//   * hazards by interval (step s: G0 computes in interval 2s, G1 in 2s+1; B(s) lives in stage s & 1, A(ss) in buffer ss & 1):
//       B stage s&1 is last read in interval 2s+1.  B(s+1) -> stage (s+1)&1 (last read in 2s-1) is requested by G1 — and only G1 — in interval 2s
//       and retired by its vmcnt(0) at the end of 2s+1 (two intervals to land); the barrier that ends 2s+1 publishes it; first read in 2s+2.
//       A(ss+1) is requested by every wavefront (its share) in its load intervals of taps kx <= KW-2 of super-step ss (the buffer was last read
//       in super-step ss-1); G1 retires its pieces with the vmcnt(0) of every compute interval, G0 with one vmcnt(0) at the end of the
//       super-step's last load interval.  No request has less than two intervals to land.
//     Every read of a staged buffer is at least one barrier after the wait that retired it (cdna_hip_programming.md, 8-phase template rule).
//
// Numerics: the products and the accumulation order of conv_taps.hip ((ky, chunk, kx); per accumulator and k-half lo*hi, hi*lo, hi*hi):
// BIT-IDENTICAL to its tiles (tests/test_taps_gpu.py), so the autotuner may choose among them (routing is still by geometry).
// LDS: B stages [2][hi BN | lo BN] x 64-byte rows, A buffers [2][hi: 272 rows + zero row | lo: ...] — 132 KB (BN = 256) / 100 KB (BN = 128).
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

constexpr int HALO = 16;          // extra A rows per (ky, chunk): (kw - 1) * dw <= 16

#ifdef FGT_PP_TRACE
// Diagnostic builds only (fgt_amd.build.build(variant="pptrace", extra_flags=["-DFGT_PP_TRACE"]), tools/pp_trace.py): s_memtime stamps of
// workgroup 0's wavefronts at the interval boundaries of the first PP_TR_STEPS steps: [wave][step][0: step top, 1: first half done (before the
// barrier), 2: after the barrier, 3: second half done incl. its waits, 4: after the closing barrier, 5: first MFMA about to issue, 6: last MFMA
// issued, 7: B pieces requested, 8: A pieces requested].
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

// SCHED 0: the ping-pong schedule (header).  SCHED 1 ("...it" tiles): NO roles — every wavefront runs  | compute(s) with its requests
// INTERLEAVED |: its share of B(s+1) goes out behind the first sub-steps' MFMAs (landing under the rest of the step), its A pieces behind a
// later one, `vmcnt(A pieces of this step)` + ONE barrier end the step.  The requests of a step are spread over its length (one LDS-DMA
// instruction per ~9 MFMAs and wavefront: the CU's 32-deep DMA queue never fills, an instruction is accepted in ~24 cycles instead of
// 200-600, tools/micro/dma_issue.hip) and sit in the shadow of the partner wavefront's MFMAs on the same SIMD.
template <int BN, int KW, int SCHED>
__global__ void __launch_bounds__(512, 2) conv_taps_pp_kernel(const ConvP p) {
    constexpr int BM = 256, NW = 8;
    constexpr int WN = BN / 64, WM = NW / WN;            // BN = 256: 2 x 4 wavefronts of 128x64; BN = 128: 4 x 2 wavefronts of 64x64
    constexpr int WTM = BM / WM, WTN = 64, TM = WTM / 32, TN = 2;
    constexpr int NIH = TM / 2, NU = 2 * NIH;            // sub-steps per step: (k-half, pair of 32-row blocks)
    constexpr int AR = BM + HALO;                        // A rows per plane filled by DMA; row AR is the zero row
    constexpr int APL = (AR + 1) * 64;                   // bytes per A plane
    constexpr int GA = AR / 16, GB = BN / 16;            // 16-row DMA groups per plane
    constexpr int B_IT = 2 * GB / NW;                    // B pieces per wavefront and step
    constexpr int A_BYTES = 2 * APL, B_BYTES = 2 * BN * 64;
    constexpr int LDS_BYTES = 2 * B_BYTES + 2 * A_BYTES;
    constexpr int STAGE = LDS_BYTES / 8;                 // floats in half of the LDS (the epilogue's view of its scratch)
    constexpr int APW_ = 5;                              // A pieces a wavefront owns per (ky, chunk): 2 groups x 2 planes (+ the halo group: waves 6, 7)
    constexpr int ASTEPS = KW - 1;                       // they go out in the load intervals of taps 0 .. KW-2 (piece it in tap it % ASTEPS)
    constexpr int MAXA = (APW_ + ASTEPS - 1) / ASTEPS;   // most A pieces a wavefront requests in one load interval
    static_assert(AR == 272 && GA == 17 && (2 * GB) % NW == 0 && B_IT >= 1 && (BN == 128 || BN == 256) && KW >= 3 && TM % 2 == 0, "tile / wavefront geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
#ifdef FGT_PP_TRACE
    const bool g1 = (wave >= 4) != (p.pipe == 6);        // (FGT_CONV_PIPE=6: the older wavefronts take the load-first role)
#else
    const bool g1 = wave >= 4;                           // (waves 0-3 own tile rows 0-127 in both geometries; the roles do not depend on the rows)
#endif
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    char* const lds = reinterpret_cast<char*>(smem);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)lds;     // LDS byte address of the dynamic segment
    char* const Bst = lds;                               // [2][hi BN rows | lo BN rows]
    char* const Abuf = lds + 2 * B_BYTES;                // [2][hi AR rows, zero row | lo AR rows, zero row]
    if (tid < 64) reinterpret_cast<float*>(Abuf + (tid >> 5) * A_BYTES + ((tid >> 4) & 1) * APL + AR * 64)[tid & 15] = 0.f;

    // Normal mode only (the reused taps are the kx taps of a stride-1 "same" convolution; the transposed k x 1 mode and the nearest-x2 upsampling
    // of conv_taps.hip stay on its tiles: fgt_conv_taps_pp_launch declines them).  The load path is written for few live scalars: the first
    // version of this file kept conv_taps.hip's generality and spent ~1 000 cycles per wavefront and step on 200 v_readlane reloads of spilled
    // SGPRs around 5 LDS-DMA instructions (profiles/r04_run11_pp_isa_mix.txt).
    const int W = d.W, H = d.H, HW = H * W;
    const int dwx = d.dw, p_i = d.pw;
    const bool il = d.in_split == 2;
    const int nch0 = p.Cg0 / 32, nch1 = p.Cg1 / 32, nchunk = nch0 + nch1;
    const int nss = d.kh * nchunk;
    const int cstride = il ? 128 : 64;                   // bytes from one 32-channel chunk of a pixel to the next

    auto src_hi = [&](int s) {
        const __bf16* x = reinterpret_cast<const __bf16*>(s ? p.x1 : p.x0);
        const long c0 = s ? (long)d.off1 + (long)g * p.Cg1 : (long)d.off0 + (long)g * p.Cg0;
        return reinterpret_cast<const char*>(x + (il ? 2 * c0 : c0));
    };
    auto src_lo_off = [&](int s) { return il ? 64l : 2 * (s ? p.ps1 : p.ps0); };
    const char* a_hi = src_hi(0);
    const char* a_lo = a_hi + src_lo_off(0);
    int a_ld2 = 2 * d.ld0, a_left = nch0, a_src = 0;      // (a_ld2: bytes from one pixel of the source to the next)
    int a_dy = -d.ph, a_dyW = -d.ph * W;                  // ky tap shift: in rows / in pixels
    auto a_advance = [&]() {
        a_hi += cstride; a_lo += cstride;
        if (--a_left == 0) {
            if (a_src == 0 && nch1 > 0) {
                a_src = 1; a_left = nch1; a_ld2 = 2 * d.ld1;
            } else {
                a_src = 0; a_left = nch0; a_ld2 = 2 * d.ld0;
                a_dy += d.dh; a_dyW += d.dh * W;
            }
            a_hi = src_hi(a_src);
            a_lo = a_hi + src_lo_off(a_src);
        }
    };

    // ---- A pieces (16 rows x 64 bytes of one plane): wavefront w owns, of BOTH planes, the row groups w and w + 8 (pieces it = 0..3: plane
    // it >> 1, group w + 8 * (it & 1)) and — waves 6 and 7 — the halo group 16 of plane 0 / 1 (piece 4).  A lane fetches row (lane >> 2) of a
    // group, 16-byte column kc (swizzled on the source side): three (pixel, image row) pairs per lane describe all five pieces.
    const int lrow = lane >> 2;
    const int kc16 = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;
    int a_q[3], a_y[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int grp = c < 2 ? wave + 8 * c : 16;
        const long q = (long)bm0 - p_i + grp * 16 + lrow;           // flattened (n, y, x) index of the LDS row for the centre taps
        const bool valid = q >= 0 && q < (long)d.N * HW;
        const int rem = valid ? (int)(q % HW) : 0;
        a_q[c] = valid ? (int)q : 0;
        a_y[c] = valid ? rem / W : -(1 << 30);
    }
    const char* const zp = reinterpret_cast<const char*>(p.zero_page);
#ifdef FGT_PP_TRACE
    const int abl = p.pipe;                               // FGT_CONV_PIPE: 1 normal, 2 no LDS-DMA in the loop
#else
    constexpr int abl = 1;
#endif
    constexpr int APW = 5, NPA = 34;
    auto issue_A = [&](auto IT, int ab) __attribute__((always_inline)) {
        constexpr int it = decltype(IT)::value, c = it < 4 ? (it & 1) : 2;
        if (abl == 2) return 0;
        if constexpr (it == 4) { if (wave < 6) return 0; }       // (wave-uniform)
        const int plane = it < 4 ? it >> 1 : wave & 1;
        const int grp = it < 4 ? wave + 8 * (it & 1) : 16;
        const char* ptr = (plane ? a_lo : a_hi) + ((long)(a_q[c] + a_dyW) * a_ld2 + kc16);
        const bool ok = (unsigned)(a_y[c] + a_dy) < (unsigned)H;
        glds16(ok ? ptr : zp, Abuf + ab * A_BYTES + plane * APL + grp * 1024);
        return 1;
    };

    // ---- weights: interleaved rows [Kpad/32][hi 32 | lo 32], K-step order as in conv_taps.hip: kstep(ky, c, kx) = (ky*KW + kx) * nchunk + c.
    // One per-lane 64-bit base + a wave-uniform byte offset per piece; the K position is a scalar running offset (32 bits: it stays inside a row).
    // Every wavefront requests its share of a B tile: the 16-row groups BPP * wave .. BPP * wave + BPP - 1 of both planes (BN = 256: 2 + 2
    // pieces, BN = 128: 1 + 1).  (Measured on the way here, profiles/r04_run10_pp_trace.txt: with G1 alone requesting all 32 pieces of a
    // 256-wide tile its load phase took 3 000 cycles — a wavefront gets an LDS-DMA instruction accepted every ~190 cycles (weights, L2) to
    // ~500 cycles (im2col rows) next to a computing partner, whoever issues and whether or not M0 changes between them.)
    const char* const w_lane = reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(p.w) + ((long)g * d.Npad + bn0 + lrow) * (2 * d.Kpad)) + kc16;
    const int w_row16 = 64 * d.Kpad;                      // bytes from one 16-row group of the weight image to the next
    int w_k = 0;                                          // byte offset of the K-step the B stream is at
    const int dkx = nchunk * 128, dss = 128 - (KW - 1) * dkx;      // to the next kx of a (ky, chunk) / from its last kx to the next chunk; to the next ky: + 128
    int b_c = 0;                                          // chunk (within its ky) of the super-step the B stream is in
    constexpr int BPP = GB / NW;                          // pieces per plane and wavefront
    static_assert(BPP >= 1, "B pieces per wavefront");
    const int npad_rows = d.Npad - bn0;                   // weight rows of this tile that exist (the rest reads the zero page)
    auto issue_B_plane = [&](int plane, int bs) __attribute__((always_inline)) {
        if (abl == 2) return;
#pragma unroll
        for (int i = 0; i < BPP; ++i) {
            const int grp = wave * BPP + i;
            const bool ok = grp * 16 < npad_rows;         // (wave-uniform)
            glds16(ok ? w_lane + ((long)grp * w_row16 + plane * 64 + w_k) : zp, Bst + bs * B_BYTES + plane * BN * 64 + grp * 1024);
        }
    };
    auto issue_B = [&](int bs) __attribute__((always_inline)) {
        issue_B_plane(0, bs);
        issue_B_plane(1, bs);
    };
    auto advance_B = [&](bool last_kx) {                  // behind the B tile of a step with kx = KW-1 (last_kx) or kx < KW-1
        int dlt = dkx;
        if (last_kx) {
            dlt = dss;
            if (++b_c == nchunk) { b_c = 0; dlt = 128; }
        }
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
    const unsigned b_lane = (unsigned)((wn * WTN + l31) * 64);     // B fragment rows: wave-tile base (multiple of 32) + l31
    const unsigned so0 = (unsigned)swz(l31, lh) * 2u, so1 = (unsigned)swz(l31, 2 + lh) * 2u;

    // ---- prologue: A rows of super-step 0 and the B tile of step 0, landed and published
    static_for<APW>([&](auto IT) __attribute__((always_inline)) { issue_A(IT, 0); });
    a_advance();
    issue_B(0);
    advance_B(false);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- one step of one wavefront's matrix work: TM x TN x 6 MFMAs in sub-steps of 12 (k-half, pair of 32-row blocks), the NEXT sub-step's
    // fragment reads in flight underneath (two register sets).  The reads are inline assembly with hand-counted lgkmcnt waits: left to
    // itself hipcc hoisted all 16-24 ds_read_b128 of a step above its first MFMA and waited lgkmcnt(0) — 700 cycles of LDS round trips in
    // front of 768 cycles of MFMAs, nothing overlapped (profiles/r04_run5_pp_trace_ablations.txt).  The wait carries the sub-step's
    // fragments as "+v" operands, so no MFMA can be scheduled above it; LDS returns in order, so lgkmcnt(n) with n = the reads requested
    // after them retires exactly this sub-step's (anything else in that queue only makes the wait stricter).
#ifdef FGT_PP_TRACE
    int tr_step = 0;
#endif
    auto rd = [&](bf16x8& dst, unsigned addr) __attribute__((always_inline)) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
        dst = __builtin_bit_cast(bf16x8, v);
    };
    auto compute = [&](auto KX, unsigned Ab, int bs, auto&& hook) __attribute__((always_inline)) {
        constexpr int kx = decltype(KX)::value;
        // (opaque: the per-tap fragment addresses are recomputed per step, not hoisted and kept live across the loop)
        int shv = kx * dwx;
        asm volatile("" : "+v"(shv));
        const int sh = __builtin_amdgcn_readfirstlane(shv);
        unsigned a0[TM];
        static_for<TM>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const bool xin = (unsigned)(oxp[i] + sh) < (unsigned)W;
            const int R = xin ? Rb[i] + sh : AR;
            a0[i] = lds0 + Ab + (unsigned)R * 64u + (unsigned)(((R >> 2) & 3) ^ lh) * 16u;     // k-half 0: slot lh; k-half 1: slot 2 + lh (^ 32 bytes)
        });
        const unsigned bb = lds0 + (unsigned)(bs * B_BYTES) + b_lane;
        bf16x8 Bh[2][TN], Bl[2][TN];                      // [k half][32-column block]
        bf16x8 Ah[2][2], Al[2][2];                        // [register set][block of the pair]
        auto loadB = [&](auto KS) __attribute__((always_inline)) {
            constexpr int ks = decltype(KS)::value;
            const unsigned so = ks ? so1 : so0;
            static_for<TN>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                rd(Bh[ks][j], bb + so + j * 32 * 64);
                rd(Bl[ks][j], bb + so + BN * 64 + j * 32 * 64);
            });
        };
        auto loadA = [&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, ks = u / NIH, ih = u % NIH, set = u & 1;
            static_for<2>([&](auto II) __attribute__((always_inline)) {
                constexpr int ii = decltype(II)::value;
                const unsigned a = a0[2 * ih + ii] ^ (ks ? 32u : 0u);
                rd(Ah[set][ii], a);
                rd(Al[set][ii], a + APL);
            });
        };
        loadB(std::integral_constant<int, 0>{});
        loadA(std::integral_constant<int, 0>{});
        static_for<NU>([&](auto U) __attribute__((always_inline)) {
            constexpr int u = decltype(U)::value, ks = u / NIH, ih = u % NIH, set = u & 1;
            constexpr bool more = u + 1 < NU, nextB = more && (u + 1) % NIH == 0;
            if constexpr (nextB) loadB(std::integral_constant<int, (u + 1) / NIH>{});
            if constexpr (more) loadA(std::integral_constant<int, u + 1>{});
            constexpr int pending = more ? 4 + (nextB ? 2 * TN : 0) : 0;          // reads requested after this sub-step's
            asm volatile("s_waitcnt lgkmcnt(%8)"
                         : "+v"(Ah[set][0]), "+v"(Al[set][0]), "+v"(Ah[set][1]), "+v"(Al[set][1]), "+v"(Bh[ks][0]), "+v"(Bl[ks][0]), "+v"(Bh[ks][1]), "+v"(Bl[ks][1])
                         : "n"(pending));
            if constexpr (u == 0) PP_STAMP(5);
            // same products as conv_taps.hip / conv_split.hip (lo*hi, hi*lo, hi*hi per accumulator and k-half)
            static_for<3>([&](auto P) __attribute__((always_inline)) {
                constexpr int prod = decltype(P)::value;
                static_for<2>([&](auto II) __attribute__((always_inline)) {
                    constexpr int ii = decltype(II)::value;
                    static_for<TN>([&](auto J) __attribute__((always_inline)) {
                        constexpr int j = decltype(J)::value;
                        const bf16x8 av = prod == 0 ? Al[set][ii] : Ah[set][ii];
                        const bf16x8 bv = prod == 1 ? Bl[ks][j] : Bh[ks][j];
                        acc[2 * ih + ii][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[2 * ih + ii][j], 0, 0, 0);
                    });
                });
            });
            __builtin_amdgcn_sched_barrier(0);            // the hook's address arithmetic / LDS-DMA issue goes BEHIND this sub-step's MFMAs
            hook(U);
        });
    };
    auto no_hook = [](auto) {};

    // ---- the requests of step (lss, lkx): this wavefront's share of the NEXT step's B tile, then of the next super-step's A rows.
    // Every wavefront keeps its own load cursor: G1 runs it one step ahead of its matrix work (below).
    int lss = 0, lkx = 0, lbs = 1;                        // load cursor, and the B stage the next request goes to
    auto load = [&]() __attribute__((always_inline)) {
        const bool lastss = lss + 1 == nss;
        if (lss < nss && !(lastss && lkx == KW - 1)) {
            issue_B(lbs);
            advance_B((lkx + 1) % KW == KW - 1);
        }
        PP_STAMP(7);
        int na = 0;
        if (lss < nss && lkx < ASTEPS && !lastss) {
            static_for<APW>([&](auto IT) __attribute__((always_inline)) {
                if (decltype(IT)::value % ASTEPS == lkx) na += issue_A(IT, (lss + 1) & 1);
            });
        }
        PP_STAMP(8);
        lbs ^= 1;
        if (++lkx == KW) { lkx = 0; ++lss; a_advance(); }
        return na;
    };

    if constexpr (SCHED == 1) {
        int bs = 0;
        for (int ss = 0; ss < nss; ++ss) {
            const bool last = ss + 1 == nss;
            const unsigned Ab = (unsigned)(2 * B_BYTES + (ss & 1) * A_BYTES);
            static_for<KW>([&](auto KX) __attribute__((always_inline)) {
                constexpr int kx = decltype(KX)::value;
                const bool moreB = !(last && kx == KW - 1);
                int na = 0;
                auto a_pieces = [&]() __attribute__((always_inline)) {
                    if constexpr (kx < ASTEPS) {
                        if (!last) {
                            static_for<APW>([&](auto IT) __attribute__((always_inline)) {
                                if constexpr (decltype(IT)::value % ASTEPS == kx) na += issue_A(IT, (ss + 1) & 1);
                            });
                        }
                    }
                };
                auto hook = [&](auto U) __attribute__((always_inline)) {
                    constexpr int u = decltype(U)::value;
                    if constexpr (NU >= 4) {
                        if constexpr (u == 0) { if (moreB) issue_B_plane(0, bs ^ 1); }
                        if constexpr (u == 1) { if (moreB) issue_B_plane(1, bs ^ 1); }
                        if constexpr (u == 2) a_pieces();
                    } else {
                        if constexpr (u == 0) { if (moreB) issue_B(bs ^ 1); a_pieces(); }
                    }
                };
                PP_STAMP(0);
                compute(KX, Ab, bs, hook);
                PP_STAMP(6);
                if (moreB) advance_B((kx + 1) % KW == KW - 1);
                __builtin_amdgcn_sched_barrier(0);
                wait_vmcnt_upto<MAXA>(na);                // the next step's B tile (and every older A piece) has landed; this step's A pieces may fly on
                PP_STAMP(1); PP_STAMP(2); PP_STAMP(7); PP_STAMP(8); PP_STAMP(3);
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
    } else {
    // ---- ONE barrier per step.  G0: C(s) L(s) | C(s+1) L(s+1) | ...      G1: L(s) C(s) | L(s+1) C(s+1) | ...   ("|" = workgroup barrier).
    // Within a step the two wavefronts of a SIMD are staggered by construction (G1 requests first, G0 computes first) and nothing orders
    // them: what a step reads was published by the barrier before it, what it requests is read after the barrier behind it (header table;
    // the mid-step barrier of the first version protected nothing and made every interval as long as one wavefront's whole compute phase:
    // 5 400 cycles per step for 3 072 cycles of MFMAs per SIMD, profiles/r04_run8_pp_trace.txt).
    if (g1) load();
    int bs = 0;                                           // B stage of this step
    for (int ss = 0; ss < nss; ++ss) {
        const unsigned Ab = (unsigned)(2 * B_BYTES + (ss & 1) * A_BYTES);          // byte offset of this super-step's A buffer in the LDS
        static_for<KW>([&](auto KX) __attribute__((always_inline)) {
            PP_STAMP(0);
            __builtin_amdgcn_s_setprio(1);
            compute(KX, Ab, bs, no_hook);
            PP_STAMP(6);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP(1);
            if (g1) {
                wait_vmcnt<0>();                          // requested at the top of this step (next step's B tile, A pieces), landed under the matrix work
                PP_STAMP(2); PP_STAMP(7); PP_STAMP(8);
            } else {
                PP_STAMP(2);
                const int na = load();                    // B pieces first (needed right behind the barrier), then A pieces (needed a super-step later)
                __builtin_amdgcn_sched_barrier(0);
                wait_vmcnt_upto<MAXA>(na);                // everything but the A pieces just requested has landed
            }
            PP_STAMP(3);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            PP_STAMP(4);
            if (g1) load();                               // next step's requests, right behind the barrier that freed their buffers
            bs ^= 1;
#ifdef FGT_PP_TRACE
            ++tr_step;
#endif
        });
    }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN>(p, acc, smem, bm0, bn0, g);
}

template <int BN, int KW, int SCHED>
int launch_kw(const ConvP& p, hipStream_t s) {
    constexpr int BM = 256;
    constexpr size_t smem = (size_t)2 * (2 * BN * 64) + (size_t)2 * (2 * (BM + HALO + 1) * 64);
    static_assert(smem <= 160 * 1024, "LDS buffers do not fit");
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_taps_pp_kernel<BN, KW, SCHED>), (int)smem, lds_set, "conv_taps_pp")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_taps_pp_kernel<BN, KW, SCHED>), grid, dim3(512), smem, s, q);
    return fgt_check_launch("conv_taps_pp");
}

template <int BN, int SCHED>
int launch(const ConvP& p, hipStream_t s) {
    ConvP q = p;
    q.tr_li = 0;
    if (p.d.kw == 1 || p.d.upsample) {                    // k x 1 (transposed tile order) and nearest-x2 upsampling: conv_taps.hip's tiles
        fgt_set_error("fgt_conv2d: the 256-row ping-pong tap tiles do not serve k x 1 or upsampling layers");
        return FGT_EINVAL;
    }
    switch (p.d.kw) {
        case 3: return launch_kw<BN, 3, SCHED>(q, s);
        case 5: return launch_kw<BN, 5, SCHED>(q, s);
        case 7: return launch_kw<BN, 7, SCHED>(q, s);
        default: fgt_set_error("fgt_conv2d: the tap-reusing kernel is built for 3, 5, 7 reused taps (got %d x %d)", p.d.kh, p.d.kw); return FGT_EINVAL;
    }
}

}  // namespace

#ifdef FGT_PP_TRACE
extern "C" int fgt_debug_pp_trace(unsigned long long* host_out, int n) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(fgt_pp_trace_buf), sizeof(unsigned long long) * n) == hipSuccess ? 0 : 1;
}
#endif

// called by fgt_conv_taps_launch (conv_taps.hip) for the 256-row tile codes; same eligibility as the other tap tiles
int fgt_conv_taps_pp_launch(int bn, int sched, const ConvP& p, hipStream_t s) {
    if (sched) return bn == 256 ? launch<256, 1>(p, s) : launch<128, 1>(p, s);
    return bn == 256 ? launch<256, 0>(p, s) : launch<128, 0>(p, s);
}
