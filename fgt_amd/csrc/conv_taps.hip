// Round 3.  Version 1 of this kernel (git history) was correct and 4-12 % SLOWER than the early-release tiles of conv_split.hip with 31 % fewer
// LDS-DMA instructions (profiles/r03_run6_split_sweep_taps_vs_early_release.txt): its K loop carried 118 VALU + 131 SALU instructions per step
// next to 12 MFMAs (the early-release kernel: 70 + 42), 40 of them v_readlane reloads of spilled scalars, and one B tile in flight instead of
// two.  This version: kw is a template parameter (the kx loop is unrolled: piece ownership, shifts and wait counts are compile-time), the
// weight pointers and the im2col source are iterators (one add per piece and step), the zero row is a ROW INDEX select ahead of the address
// arithmetic (20-40 VALU + 22 SALU per step), and the schedule is the early-release one (two B tiles in flight).  Measured
// (profiles/r03_run7_split_sweep_taps_v2.txt, r03_run9_*): +4...10 % over the early-release tiles on the Cout >= 256 layers, +15...27 % on the
// Cout <= 192 layers of RAFT / LAFC / the decoder, where the 128x64 tile fits three workgroups per CU.
//
// bf16x3 implicit-GEMM convolution for STRIDE-1 "SAME" convolutions with kw in {3, 5, 7} on pre-split operands: the im2col rows of a
// (ky, 32-channel chunk) stay in LDS for ALL kx taps (gfx950).
//
// For a stride-1 convolution whose output map has the size of its input map, the A tile of tap (ky, kx) is the A tile of tap (ky, 0)
// shifted by kx * dw PIXELS in the flattened (n, y, x) index: output pixel m reads input pixel m + (ky*dh - ph) * W + (kx*dw - pw)
// whenever that pixel is in the same image row.  So this kernel walks K in the order (chunk, ky, kx) and loads, per (ky, chunk), BM + 16
// consecutive input rows ONCE (9 pieces per plane instead of 8 per tap): for a 3x3 layer 18 + 3 * 16 = 66 LDS-DMA instructions per
// (ky, chunk) instead of 96 (-31 %), for the 1x5 GRU convs of RAFT 18 + 80 instead of 160 (-39 %) — and the pieces that remain are mostly
// WEIGHT rows (L2 hits), the im2col stream from the Infinity Cache / HBM shrinks by kw.  Tap kx reads its MFMA fragments from LDS rows
// r + kx*dw; a pixel whose tap leaves the image row (x + kx*dw - pw outside [0, W)) reads the plane's zero row instead; rows whose input
// row y + ky*dh - ph is outside the image were never fetched (the DMA read the zero page).
//
// K ORDER (round 5): the walk is (chunk, ky, kx) — the three ky taps of a 32-channel chunk in consecutive super-steps.  Their im2col rows are the
// same cache lines one image row apart (another tile's ky = 0 rows, or this tile's own for narrow maps): walked (ky, chunk, kx), a line came back
// nchunk super-steps later, long evicted from the XCD's 4 MB L2, and the DMA of two thirds of the A pieces paid the fabric's latency (300-600
// cycles an instruction against 66-190 for an L2 hit: NOTEBOOK §11.3).  Measured -4...-8 % on the encoder / decoder 3x3 layers, same box
// (profiles/r05_run9_*).  The weight image keeps its K-step order kstep(ky, c, kx) = (ky*KW + kx) * nchunk + c; only the iterators changed.
//
// Numerics: the same products as conv_split.hip, accumulated in the order (chunk, ky, kx) instead of (ky, kx, chunk) — NOT bit-identical
// to the other kernels (fp32 accumulation order), identical in error (tests/test_taps_gpu.py: both within 2e-5 of fp64 on the same split
// operands).  Which kernel a layer runs on is therefore decided by its GEOMETRY alone (fgt_conv_taps_eligible), never by the autotuner: an
// eligible split-input layer always runs here (the autotuner picks among THIS kernel's tiles, which are bit-identical to each other), so
// results do not depend on tuning.
//
// LDS: two B stages [hi BN | lo BN] of 64-byte rows, two A buffers [hi: BM+16 rows + zero row | lo: BM+16 rows + zero row] (the four
// 16-byte slots of a row XOR-swizzled with (row >> 2) & 3 as in conv_tile.h: any 16 consecutive rows are conflict free, so the shifted
// reads are too).  128x128: 2 * 16 KB + 2 * 18.1 KB = 68.3 KB: two workgroups per CU.
// Step (ss, kx): read fragments (A buffer ss & 1 shifted by kx*dw, B stage) | lgkmcnt(0) | barrier | request this step's share of
// super-step ss+1's A rows, then the B tile of step + 2 into the stage just read | MFMAs | vmcnt(B pieces of this step) | barrier.
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

template <int BASE, int MAXA> __device__ __forceinline__ void wait_vmcnt_plus(int na) {     // vmcnt(BASE + na), na wave-uniform in [0, MAXA]
    if constexpr (MAXA == 0) wait_vmcnt<BASE>();
    else {
        if (na == MAXA) wait_vmcnt<BASE + MAXA>();
        else wait_vmcnt_plus<BASE, MAXA - 1>(na);
    }
}

constexpr int HALO = 16;          // extra A rows per (ky, chunk): (kw - 1) * dw <= 16

template <int BM, int BN, int WM, int WN, int MINW, int KW>
__global__ void __launch_bounds__(WM* WN * 64, MINW) conv_taps_kernel(const ConvP p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int AR = BM + HALO;                        // A rows per plane filled by DMA; row AR is the zero row
    constexpr int APL = (AR + 1) * 64;                   // bytes per A plane
    constexpr int GA = AR / 16, GB = BN / 16;            // 16-row DMA groups per plane
    constexpr int NPA = 2 * GA;                          // A pieces per (ky, chunk): piece j -> plane j / GA, group j % GA
    constexpr int B_IT = 2 * GB / NW;                    // B pieces per wavefront and step
    constexpr int A_BYTES = 2 * APL, B_BYTES = 2 * BN * 64;
    constexpr int LDS_BYTES = 2 * B_BYTES + 2 * A_BYTES;
    constexpr int STAGE = LDS_BYTES / 8;                 // floats in half of the LDS (the epilogue's view of its scratch)
    constexpr int APW = (NPA + NW - 1) / NW;             // A pieces a wavefront owns per (ky, chunk)
    constexpr int ASTEPS = KW - 1;                       // they go out in steps 0 .. KW-2 of the previous super-step (piece it in step it % ASTEPS)
    constexpr bool A_LATE = NW >= 8;                     // A pieces behind the B pieces of their step, waited for one step later
    static_assert(AR % 16 == 0 && (2 * GB) % NW == 0 && B_IT >= 1 && TM >= 1 && TN >= 1 && BN <= 128 && KW >= 3, "tile / wavefront geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    char* const lds = reinterpret_cast<char*>(smem);
    char* const Bst = lds;                               // [2][hi BN rows | lo BN rows]
    char* const Abuf = lds + 2 * B_BYTES;                // [2][hi AR rows, zero row | lo AR rows, zero row]
    if (tid < 64) reinterpret_cast<float*>(Abuf + (tid >> 5) * A_BYTES + ((tid >> 4) & 1) * APL + AR * 64)[tid & 15] = 0.f;

    // Axes.  Normal mode: the reused taps are the kx taps, the tile's rows are consecutive pixels of the flattened (n, y, x) index.  TRANSPOSED
    // mode (p.tr_li != 0: k x 1 convolutions, RAFT's vertical GRU pass): the reused taps are the ky taps and the tile's rows walk the image
    // in (n, x, y) order — row m' = (n, x, y) is pixel n*H*W + y*W + x; every LDS-DMA lane carries its own address, so a column strip costs
    // what a row strip costs, and the epilogue maps m' back to the pixel (conv_out_row).  "inner" = the reused axis, "outer" = the other one.
    // NEAREST x2 UPSAMPLING (desc.upsample, normal mode only): the tile walks the 2H x 2W output grid, LDS row = output-grid pixel; its source
    // is input pixel ((y' + dy) >> 1, x' >> 1): adjacent LDS rows fetch the same 64 bytes twice, the taps shift in the output grid as always.
    const bool tr = p.tr_li != 0;
    const int ush = d.upsample ? 1 : 0;
    const int W = d.W << ush, H = d.H << ush;               // the grid the tile walks (= the output map)
    const int HW = H * W;
    const int Li = tr ? H : W, Lo = tr ? W : H;             // extents
    const int Si = tr ? W : 1, So = tr ? 1 : W;             // pixel strides
    const int dwx = tr ? d.dh : d.dw, d_o = tr ? d.dw : d.dh, p_i = tr ? d.ph : d.pw, p_o = tr ? d.pw : d.ph;
    const bool il = d.in_split == 2;
    const int nch0 = p.Cg0 / 32, nch1 = p.Cg1 / 32, nchunk = nch0 + nch1;
    // ABI 7 (desc.ky_skip_n0, normal mode): the weights of this tile's columns are all zero for ky = 0 — its K walk starts at ky = 1
    const int ky0 = (!tr && d.ky_skip_n0 > 0 && bn0 >= d.ky_skip_n0) ? 1 : 0;
    const int n_o = (tr ? d.kw : d.kh) - ky0;             // outer taps walked (the axis that is not reused)
    const int nss = n_o * nchunk;
    const long cstride = il ? 128 : 64;                  // bytes from one 32-channel chunk of a pixel to the next

    // ---- im2col source iterator (wave-uniform): the super-step whose A rows are requested next.  Byte pointers to chunk (ky, c) of pixel 0:
    // planes: channel c at element c, lo plane ps further; interleaved: element (c / 32) * 64 + c % 32, lo 32 elements further.
    auto src_hi = [&](int s) {
        const __bf16* x = reinterpret_cast<const __bf16*>(s ? p.x1 : p.x0);
        const long c0 = s ? (long)d.off1 + (long)g * p.Cg1 : (long)d.off0 + (long)g * p.Cg0;
        return reinterpret_cast<const char*>(x + (il ? 2 * c0 : c0));
    };
    auto src_lo_off = [&](int s) { return il ? 64l : 2 * (s ? p.ps1 : p.ps0); };
    const char* a_hi = src_hi(0);
    const char* a_lo = a_hi + src_lo_off(0);
    int a_ld = d.ld0, a_left = nch0, a_src = 0;
    const int a_dy0 = ky0 * d_o - p_o;
    int a_dy = a_dy0, a_dyW = a_dy * So;                  // outer tap shift: in outer coordinates / in pixels
    int a_o = 0;                                          // outer tap of the super-step the A stream is in
    auto a_advance = [&]() {                              // next outer tap of the same chunk; behind the last one: the next chunk (never wraps)
        a_dy += d_o; a_dyW += d_o * So;
        if (++a_o < n_o) return;
        a_o = 0; a_dy = a_dy0; a_dyW = a_dy0 * So;
        a_hi += cstride; a_lo += cstride;
        if (--a_left == 0 && a_src == 0 && nch1 > 0) {
            a_src = 1; a_left = nch1; a_ld = d.ld1;
            a_hi = src_hi(1);
            a_lo = a_hi + src_lo_off(1);
        }
    };

    // ---- this lane's A rows: row (lane >> 2) of each of its pieces j = wave + it * NW (plane j / GA, group j % GA), 16-byte column kc
    const int lrow = lane >> 2;
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);       // swizzle on the source side
    int a_q[APW], a_y[APW];                              // flattened input pixel of the row for ky*dh - ph = 0, and its y (far negative: never fetched)
    // (upsampling: a_q = n*Hin*Win + (x' >> 1), the source pixel of output-grid row 0 of that column; a_y is the OUTPUT-grid row)
#pragma unroll
    for (int it = 0; it < APW; ++it) {
        const int j = wave + it * NW;
        const long q = (long)bm0 - p_i + (j % GA) * 16 + lrow;      // tile-order index of the LDS row for the centre taps
        const bool valid = j < NPA && q >= 0 && q < (long)d.N * HW;
        const int rem = valid ? (int)(q % HW) : 0;
        const int co = rem / Li, ci = rem - co * Li;                // outer / inner coordinate
        a_q[it] = !valid ? 0 : ush ? (int)((q - rem) >> 2) + (ci >> 1) : (int)(q - rem) + co * So + ci * Si;   // its pixel
        a_y[it] = valid ? co : -(1 << 30);
    }
    const char* const zp = reinterpret_cast<const char*>(p.zero_page);
    auto issue_A = [&](int it, int ab) {
        const int j = wave + it * NW;
        if (j >= NPA) return 0;                           // (wave-uniform)
        const int plane = j / GA, grp = j % GA;
        const int pix = a_q[it] + (ush ? ((a_y[it] + a_dy) >> 1) * d.W : a_dyW);
        const char* ptr = (plane ? a_lo : a_hi) + 2 * ((long)pix * a_ld + kc * 8);
        const bool ok = (unsigned)(a_y[it] + a_dy) < (unsigned)Lo;
        glds16(ok ? ptr : zp, Abuf + ab * A_BYTES + plane * APL + grp * 1024);
        return 1;
    };

    // ---- weights: interleaved rows [Kpad/32][hi 32 | lo 32] (128 bytes per K-step of 32 channels); running pointers, K-step order
    // kstep(ky, c, kx) = (ky*KW + kx) * nchunk + c; walked (chunk, ky, kx): + nchunk to the next kx AND from a ky's last kx to the next ky, and
    // 1 - (n_o*KW - 1) * nchunk from a chunk's last step to the first step of the next chunk
    const char* wp[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int piece = wave + it * NW, plane = piece / GB, grp = piece % GB;
        const int brow = bn0 + grp * 16 + lrow;          // (< Npad: Npad is a multiple of 128 >= BN)
        wp[it] = reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(p.w) + ((long)g * d.Npad + brow) * (2 * d.Kpad) + plane * 32 + kc * 8);
    }
    // (transposed: the reused taps are ky: + kw * nchunk per tap, and the step to the next kx is the step to the next chunk)
    const long dkx = (long)nchunk * (tr ? d.kw : 1) * 128;
    const long dchunk = 128 - ((long)n_o * KW - 1) * dkx;  // from the last step of a chunk (last outer tap, last reused tap) to the first step of the next chunk
    if (ky0) {
#pragma unroll
        for (int it = 0; it < B_IT; ++it) wp[it] += (long)KW * dkx;      // the K-steps of ky = 0
    }
    int b_c = 0;                                          // outer tap (within its chunk) of the super-step the B stream is in
    auto issue_B = [&](int bs) {
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int piece = wave + it * NW, plane = piece / GB, grp = piece % GB;
            glds16(wp[it], Bst + bs * B_BYTES + plane * BN * 64 + grp * 1024);
        }
    };
    auto advance_B = [&](bool last_kx) {                  // behind the B tile of a step with kx = KW-1 (last_kx) or kx < KW-1
        long dlt = dkx;                                   // the next reused tap — and the next outer tap of the same chunk — is one tap's K-steps further
        if (last_kx && ++b_c == n_o) { b_c = 0; dlt = dchunk; }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) wp[it] += dlt;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    // LDS row of this lane's output pixels for kx = 0 (one per 32-row block), and x - pw of the pixel (the taps that leave the image row read the zero row)
    int Rb[TM], oxp[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        Rb[i] = wm * WTM + i * 32 + l31;
        oxp[i] = (bm0 + Rb[i]) % Li - p_i;
    }
    const unsigned b_lane = (unsigned)((wn * WTN + l31) * 64);     // B fragment rows: wave-tile base (multiple of 32) + l31

    // ---- prologue: A rows of super-step 0, the B tiles of steps 0 and 1 (KW >= 3: both in super-step 0)
#pragma unroll
    for (int it = 0; it < APW; ++it) issue_A(it, 0);
    a_advance();
    issue_B(0); advance_B(false);
    issue_B(1); advance_B(KW == 2);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    int bs = 0;                                           // B stage of this step
    for (int ss = 0; ss < nss; ++ss) {
        const bool last = ss + 1 == nss;
        const unsigned Ab = (unsigned)(2 * B_BYTES + (ss & 1) * A_BYTES);          // byte offset of this super-step's A buffer in the LDS
        static_for<KW>([&](auto KX) {
            constexpr int kx = decltype(KX)::value;
            bf16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
            {
                int sh = kx * dwx;
                // (kw = 3 on 4 wavefronts: hipcc hoists the fragment addresses of the three taps out of the loop; for kw = 5, 7 and in the
                //  128-register tiles they would spill: the shift is opaque there, so they are recomputed per step, 9 VALU instructions per block)
                if constexpr (KW > 3 || NW >= 8) asm volatile("" : "+s"(sh));
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const bool xin = (unsigned)(oxp[i] + sh) < (unsigned)Li;
                    const int R = xin ? Rb[i] + sh : AR;
                    const unsigned a0 = Ab + (unsigned)R * 64u + (unsigned)(((R >> 2) & 3) ^ lh) * 16u;     // k-half 0: slot lh; k-half 1: slot 2 + lh
                    const unsigned a1 = a0 ^ 32u;
                    ah[0][i] = *reinterpret_cast<const bf16x8*>(lds + a0);
                    al[0][i] = *reinterpret_cast<const bf16x8*>(lds + a0 + APL);
                    ah[1][i] = *reinterpret_cast<const bf16x8*>(lds + a1);
                    al[1][i] = *reinterpret_cast<const bf16x8*>(lds + a1 + APL);
                }
                const unsigned bb = (unsigned)(bs * B_BYTES) + b_lane;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const unsigned so = (unsigned)swz(l31, ks * 2 + lh) * 2u;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bh[ks][j] = *reinterpret_cast<const bf16x8*>(lds + bb + so + j * 32 * 64);
                        bl[ks][j] = *reinterpret_cast<const bf16x8*>(lds + bb + so + BN * 64 + j * 32 * 64);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                 // every wavefront holds its fragments: the B stage can be refilled
            constexpr bool crosses = kx + 2 >= KW;        // the step two ahead belongs to the next super-step
            const bool more = !crosses || !last;
            // Request order.  8-wavefront tiles: B first, then the A pieces, which may stay in flight across this step's end (two steps to
            // land: they come from the Infinity Cache / HBM, the weights from L2); 4-wavefront tiles: A pieces first, landed with this step
            // (same-box A/B: +3 % on the 128x128 tile on 8 wavefronts, -2 % on the 128x64 tile on 4).
            int na = 0;
            if constexpr (A_LATE)
                if (more) { issue_B(bs); advance_B((kx + 2) % KW == KW - 1); }
            if constexpr (kx < ASTEPS) {
                if (!last) {
#pragma unroll
                    for (int it = 0; it < APW; ++it)
                        if (it % ASTEPS == kx) na += issue_A(it, (ss + 1) & 1);
                }
            }
            if constexpr (!A_LATE)
                if (more) { issue_B(bs); advance_B((kx + 2) % KW == KW - 1); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);                // (the MFMA block ahead of the other wavefronts' address arithmetic: +1...2 %, same-box A/B)
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
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (A_LATE) {
                // everything older than this step's requests has landed (the A pieces of the step before, the next step's B tile)
                constexpr int MAXA = kx < ASTEPS ? (APW + ASTEPS - 1 - kx) / ASTEPS : 0;
                if (more) wait_vmcnt_plus<B_IT, MAXA>(na); else wait_vmcnt_plus<0, MAXA>(na);
            } else {
                if (more) wait_vmcnt<B_IT>(); else wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            bs ^= 1;
        });
        a_advance();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // Epilogue diet (round 6): the bias-map bodies (fgt_conv_desc.ld_bias) and the two-headed epilogue (dual_n0) are instantiated for the 5-tap
    // instances only — their one caller is RAFT's SepConvGRU (1 x 5 and 5 x 1 convolutions, RAFT/update.py:36-58); launch_kw declines other layers.
    // The bias-map bodies were half of every instance's code (the library went from 22 to 45 MB when round 5 added them to every kernel).
    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN, KW == 5, KW == 5>(conv_epilogue_args<MINW != 6>(p), acc, smem, bm0, bn0, g);
}

template <int BM, int BN, int WM, int WN, int MINW, int KW>
int launch_kw(const ConvP& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t smem = (size_t)2 * (2 * BN * 64) + (size_t)2 * (2 * (BM + HALO + 1) * 64);
    static_assert(smem <= 160 * 1024, "LDS buffers do not fit");
    if (KW != 5 && (p.d.dual_n0 > 0 || p.d.ld_bias > 0)) { fgt_set_error("fgt_conv2d: bias maps / two-headed epilogues are built for layers with 5 reused taps (got %d)", KW); return FGT_EINVAL; }
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_taps_kernel<BM, BN, WM, WN, MINW, KW>), (int)smem, lds_set, "conv_taps")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_taps_kernel<BM, BN, WM, WN, MINW, KW>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_taps");
}

template <int BM, int BN, int WM, int WN, int MINW>
int launch(const ConvP& p, hipStream_t s) {
    ConvP q = p;
    q.tr_li = p.d.kw == 1 ? p.d.H : 0;                    // k x 1: transposed mode (never with upsampling: fgt_conv_taps_eligible)
    switch (q.tr_li ? p.d.kh : p.d.kw) {
        case 3: return launch_kw<BM, BN, WM, WN, MINW, 3>(q, s);
        case 5: return launch_kw<BM, BN, WM, WN, MINW, 5>(q, s);
        case 7: return launch_kw<BM, BN, WM, WN, MINW, 7>(q, s);
        default: fgt_set_error("fgt_conv2d: the tap-reusing kernel is built for 3, 5, 7 reused taps (got %d x %d)", p.d.kh, p.d.kw); return FGT_EINVAL;
    }
}

}  // namespace

// Geometry this kernel serves (decided by the layer alone, never by tuning): bf16x3 on split inputs (planes or interleaved) with interleaved weights,
// stride 1, zero padding, output map = input map ("same") or its nearest x2 upsampling, kw in {3, 5, 7}, (kw - 1) * dw <= 16, Cin/groups a multiple of 32 per source.
bool fgt_conv_taps_eligible(const ConvP& p) {
    const fgt_conv_desc& d = p.d;
    return d.precision == FGT_PREC_BF16X3 && (d.in_split == 1 || d.in_split == 2) && (d.w_il == 1 || d.w_il == 2) && d.sh == 1 && d.sw == 1 && d.pad_mode == 0 &&
           d.in_relu == 0 && d.Ho == (d.H << (d.upsample ? 1 : 0)) && d.Wo == (d.W << (d.upsample ? 1 : 0)) && (!d.upsample || d.kw > 1) && p.Cg0 % 32 == 0 && p.Cg1 % 32 == 0 &&
           (((d.kw == 3 || d.kw == 5 || d.kw == 7) && (d.kw - 1) * d.dw <= HALO) ||                        // kx taps reused
            (d.kw == 1 && (d.kh == 3 || d.kh == 5 || d.kh == 7) && (d.kh - 1) * d.dh <= HALO)) &&           // k x 1: ky taps reused (transposed tile order)
           d.Kpad == p.K && p.Cout_g > 4;
}

// Layers routed to this kernel when the caller leaves the tile to the library (geometry only, like eligibility).  Not the k x 1 convolutions over
// a SHORT reused axis: LAFC's temporal 3 x 1 convs run over T = 3 — a third of the taps fall outside, the tile's rows are 3-pixel columns —
// and measure 12 % slower to 2 % faster than the early-release tiles (profiles/r03_run13_split_sweep_lafc_temporal.txt).
bool fgt_conv_taps_preferred(const ConvP& p) {
    return fgt_conv_taps_eligible(p) && (p.d.kw > 1 || p.d.H >= 16);
}

int fgt_conv_taps_launch(int tile, const ConvP& p, hipStream_t s) {
    switch (tile) {
        case FGT_TILE_128x128x8: return launch<128, 128, 2, 4, 4>(p, s);
        case FGT_TILE_128x128: return launch<128, 128, 2, 2, 2>(p, s);
        case FGT_TILE_128x64: return launch<128, 64, 2, 2, 2>(p, s);
        case FGT_TILE_64x64: return launch<64, 64, 2, 2, 2>(p, s);
        case FGT_TILE_128x32: return launch<128, 64, 4, 2, 6>(p, s);          // "128x64x8t": 128x64 on 8 wavefronts of 32x32, three workgroups per CU
        // "...it" tiles: the same arithmetic with the requests interleaved into the matrix work (conv_taps_il.hip; bit-identical to the tiles above)
        case FGT_TILE_256x128: return fgt_conv_taps_il_launch(256, 128, p, s);      // "256x128it"
        case FGT_TILE_256x256_P8: return fgt_conv_taps_il_launch(256, 256, p, s);   // "256x256it"
        case FGT_TILE_128x128_EA: return fgt_conv_taps_il_launch(128, 128, p, s);   // "128x128it"
        default: fgt_set_error("fgt_conv2d: tile %d is not built for the tap-reusing kernel", tile + FGT_TILE_TAPS); return FGT_EINVAL;
    }
}
