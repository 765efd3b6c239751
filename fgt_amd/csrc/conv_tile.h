// Pieces shared by the implicit-GEMM kernels (conv_igemm.hip: register-staged fp32 inputs; conv_split.hip: pre-split bf16
// inputs through LDS-DMA): LDS tile geometry, the hi/lo split, the XCD-aware tile order and the fused epilogue.
#pragma once
#include <utility>
#include "common.h"
#include "conv_params.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = 33;

constexpr int LDB = 32;  // bf16 elements per LDS row in bf16x3 mode: 64-byte rows, no padding.
// The four 16-byte slots of a row are XOR-swizzled with (row >> 2) & 3: the 16 rows of a ds_read_b128 lane group then
// cover all 16 slots of the 256-byte bank line (conflict free) and the 8-byte stores of two adjacent rows never share a
// bank either.  (The 80-byte padded layout it replaces had 2-way store conflicts: SQ_LDS_BANK_CONFLICT = 33 % of
// SQ_LDS_IDX_ACTIVE in profiles/r01_run4_pmc_*.json.)
__device__ __forceinline__ int swz(int row, int slot) { return (slot ^ ((row >> 2) & 3)) * 8; }

__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) { fgt_split4(v, hi, lo); }

// XCD-aware tile order.  Workgroup L is dispatched to XCD L % 8 (each XCD has a private 4 MiB L2).  Every XCD walks
// its own contiguous range of M tiles with the N tiles of one M tile adjacent in time, so the im2col re-reads
// (kh*kw taps x N tiles of the same input rows) hit that XCD's L2 instead of going back to HBM.  xcd_swizzle = 0
// keeps the plain (m fastest) order for A/B measurements.  Returns false for the padding blocks of the last chunk.
__device__ __forceinline__ bool conv_tile_index(const ConvP& p, int& m_idx, int& n_idx) {
    if (p.xcd_swizzle) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        if (p.d.tile_order == 1) {      // N-major inside the XCD's range of M tiles: co-resident workgroups share one N tile's weights in the L2
            n_idx = i / p.mchunk;
            m_idx = xcd * p.mchunk + i % p.mchunk;
        } else {
            n_idx = i % p.ntiles;
            m_idx = xcd * p.mchunk + i / p.ntiles;
        }
        return m_idx < p.mtiles;
    }
    m_idx = blockIdx.x % p.mtiles;
    n_idx = blockIdx.x / p.mtiles;
    return true;
}

// A loop over 0..N-1 whose index is a compile-time constant in the body.  `#pragma unroll` is a request hipcc may decline ("unrolled
// size is too large" for the row-block loop of the epilogue with 8 accumulator tiles per wavefront): the accumulator array was then
// indexed at run time and lived in scratch memory around the epilogue.
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// Output pixel of tile row m.  Every kernel but the transposed mode of conv_taps.hip walks the output in (n, y, x) order: m IS the pixel.
// Transposed (p.tr_li = H): m = (n, x, y) -> n*H*W + y*W + x.
__device__ __forceinline__ int conv_out_row(const ConvP& p, int m) {
    if (!p.tr_li) return m;
    const int n = m / p.HoWo, rem = m - n * p.HoWo;
    const int x = rem / p.tr_li, y = rem - x * p.tr_li;
    return n * p.HoWo + y * (p.HoWo / p.tr_li) + x;
}

__device__ __forceinline__ int fgt_fastdiv(int n, const FgtFastDiv f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.sh); }

// ABI 7 (fold as a convolution): output column n -> sub-pixel (ry, rx) and channel; false for the padding columns between the ry = 0 block and ps_g0.
template <bool PHASE = false>
__device__ __forceinline__ bool conv_ps_column(const fgt_conv_desc& d, int n, int& ry, int& rx, int& ch) {
    const int rc = d.ps_r * d.ps_c;
    bool ok = true;
    int q = n;
    if (n >= rc) { ok = n >= d.ps_g0; q = n - (d.ps_g0 - rc); }
    ry = q / rc;
    const int r2 = q - ry * rc;
    rx = r2 / d.ps_c;
    ch = r2 - rx * d.ps_c;
    if constexpr (PHASE) { if (d.ps_phase_pad) ok = ch < d.ps_phase_pad; }       // ABI 9: cv real channels per sub-pixel, the rest of ps_c is padding to the tile width
    return ok;
}

// Tile row mt -> row `mo` of the output tensor (sub-pixel output: the pixel of this column's sub-pixel; okr = it lies inside the ps_H x ps_W map)
// and row `rem` of the per-image aux tables.
__device__ __forceinline__ void conv_row_of(const ConvP& p, int mt, int ry, int rx, bool col_in, long& mo, int& rem, bool& okr) {
    const fgt_conv_desc& d = p.d;
    okr = true;
    rem = 0;
    if (!d.ps_r && !d.aux_per_image) { mo = conv_out_row(p, mt); return; }
    const int f = fgt_fastdiv(mt, p.div_howo);
    rem = mt - f * p.HoWo;
    if (!d.ps_r) { mo = mt; return; }
    const int I = fgt_fastdiv(rem, p.div_wo), J = rem - I * d.Wo;
    const int y = d.ps_r * I + ry, x = d.ps_r * J + rx;
    okr = col_in && y < d.ps_H && x < d.ps_W;
    mo = ((long)f * d.ps_H + y) * d.ps_W + x;
}

// The epilogue's view of the kernel arguments (round 6).  A conv kernel takes ONE by-value ConvP (112 dwords); what the epilogue needs of it — two dozen descriptor
// fields, six pointers — stayed live through the K loop, i.e. was spilled to VGPR lanes and read back with v_readlane (312 spilled SGPRs in the hottest tap instance, 830+
// in the ones with bias-map bodies).  Behind an opaque copy of the kernarg segment pointer the epilogue re-reads them with scalar loads when it starts instead: 312 -> 72
// and 832 -> 201 spilled SGPRs.  Every conv kernel's ONLY kernel argument is its ConvP (offset 0 of the segment); -DFGT_EPI_KERNARG=0 keeps the old form for A/B builds.
#ifndef FGT_EPI_KERNARG
#define FGT_EPI_KERNARG 1
#endif
template <bool REREAD = true>
__device__ __forceinline__ const ConvP& conv_epilogue_args(const ConvP& p) {
#if FGT_EPI_KERNARG
    if constexpr (!REREAD) return p;               // (the 80-register tap tile: the segment pointer and the reloads cost it two spilled VGPRs)
    typedef const __attribute__((address_space(4))) ConvP* kconvp_t;
    kconvp_t pe = (kconvp_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pe));
    return *(const ConvP*)pe;
#else
    return p;
#endif
}

// ---- epilogue.  The accumulators go through LDS (the tile buffers are free now) so that the global side is a compact,
// coalesced float4 loop shared by every epilogue flavour: bias / per-channel scale, activation, mul / add / GRU combine,
// then the fp32 store (NHWC slice or NCHW) and / or the pre-split bf16 store (desc.out_split) the next conv's LDS-DMA
// loader consumes.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
// STAGE = floats per LDS stage; the caller guarantees 2 stages are allocated and no longer in use.
// BIAS_MAP = false: a kernel family that never sees desc.ld_bias > 0 (the fp16 kernels: the host rejects the combination) skips those instances.
// out_goff: element offset added to `out` (ABI 8, the batched GEMM mode of conv_wide.hip: group g's output block; the caller passes g = 0 then).
// DUAL = false: a kernel family that never sees desc.dual_n0 > 0 (fp32 / fp16 inputs: the host requires split inputs for two heads).
// PHASE = true: the two kernel families that serve desc.ps_phase_pad (ABI 9: conv_split.hip, conv_wide.hip; the host rejects it elsewhere).
template <int BM, int BN, int WM, int WN, int STAGE, int TM, int TN, bool BIAS_MAP = true, bool DUAL = true, bool PHASE = false>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, f32x16 (&acc)[TM][TN], float* smem, int bm0, int bn0, int g, long out_goff = 0) {
    constexpr int NT = WM * WN * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    static_assert(TM == WTM / 32 && TN == WTN / 32, "accumulator shape");
    constexpr int EP_PASSES = (BM * BN > 2 * STAGE) ? ((BM * BN > 4 * STAGE) ? 4 : 2) : 1;
    constexpr int EP_BM = BM / EP_PASSES;
    constexpr int WM_PER_PASS = WM / EP_PASSES;
    static_assert(EP_BM * BN <= 2 * STAGE && WM % EP_PASSES == 0, "epilogue staging does not fit in the tile buffers");
    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    float* Cs = smem;
    const bool vec_ok = (p.Cout_g % 4 == 0) && !d.out_nchw && (d.ldo % 4 == 0) && (d.ooff % 4 == 0) &&
                        (d.epi == FGT_EPI_NONE || d.ld_aux1 % 4 == 0) && (d.epi < FGT_EPI_GRU || d.ld_aux2 % 4 == 0) && d.ld_bias % 4 == 0;
    const bool want_f32 = d.out_split != 1, want_split = d.out_split != 0;

    // ---- fast path (every layer of the hot path: float4-aligned channels-last output): WAVE-PRIVATE staging.  Each wavefront
    // transposes its own accumulators, one 32-row block at a time, through its own 32 x WTN patch of LDS and streams the rows out
    // as float4 (a row of the patch is WTN consecutive channels = one or two full 128-byte lines).  No workgroup barrier: the
    // wavefronts of a tile finish independently, and the next workgroup's K loop overlaps this one's stores.  Scale / bias are
    // per-lane constants (a lane keeps its 4 channels), the LDS reads and aux loads of a row block are issued before its stores.
    // (The workgroup-wide version of this epilogue took 271 us of an 802-us K = 512 GEMM.)
    constexpr int PATCH = 32 * WTN;                                    // floats per wavefront
    if (vec_ok && (WM * WN) * PATCH <= 2 * STAGE) {
        float* Ws = smem + wave * PATCH;
        constexpr int LPR = WTN / 4;                                   // lanes per row (8 or 16)
        constexpr int RPI = 64 / LPR;                                  // rows per wave-instruction (8 or 4)
        constexpr int ITER = 32 / RPI;                                 // 4 or 8
        const int c4 = lane % LPR, r0 = lane / LPR;
        const int n = bn0 + wn * WTN + c4 * 4;
        const bool col_ok = n < p.Cout_g;                              // Cout_g % 4 == 0: the whole float4 is in or out
        const int co = g * p.Cout_g + n;
        const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 sc = (p.cscale && col_ok) ? *reinterpret_cast<const float4*>(p.cscale + co) : one;
        const float4 bi = (p.cbias && col_ok && d.ld_bias == 0) ? *reinterpret_cast<const float4*>(p.cbias + co) : zero;
        // ABI 7, sub-pixel output: a lane keeps its 4 columns, so its sub-pixel and output channel are per-lane constants (ps_c % 4 == 0: a float4 never
        // straddles two sub-pixels); a row's (image, I, J) costs two multiply-shift divisions.
        int ps_ry = 0, ps_rx = 0, och = co;
        bool ps_col = true;
        if (d.ps_r) ps_col = conv_ps_column<PHASE>(d, n, ps_ry, ps_rx, och);
        // ABI 8, two heads (desc.dual_n0 > 0, FGT_EPI_MUL, groups = 1).  Head 0 = columns [0, n0): fp32 output, no combine;
        // head 1 = [n0, 2 n0): times aux1[m, n - n0], split output at channel n - n0.
        // (the host requires n0 % 64 == 0 and every wavefront's columns are 32 or 64 wide: which head a WAVEFRONT serves is a scalar)
        const bool dual = DUAL && d.dual_n0 > 0;
        const bool head1 = dual && __builtin_amdgcn_readfirstlane(bn0 + wn * WTN) >= d.dual_n0;
        if (head1) och = co - d.dual_n0;
        const bool st_f32 = want_f32 && !head1, st_split = want_split && (!dual || head1);
        const bool a1_och = dual || (PHASE && d.ps_phase_pad != 0);      // (ABI 9: with ps_phase_pad aux1 is shaped like the sub-pixel output: column = channel of the pixel)
        // The body is instantiated per number of aux operands (AUX = 0: no epilogue operand, 1: mul / add, 2: GRU) and dispatched on desc.epi.
        // With desc.epi tested at run time inside ONE body, hipcc put `s_waitcnt vmcnt(0)` in front of every row group (the join of the
        // paths with and without aux loads) — and on gfx9 that counter also holds the STORES until they are acknowledged: every group of
        // stores was waited for before the next one was issued, 13 k cycles for 64 KB per workgroup in the K = 512 GEMMs
        // (profiles/r02_run8_conv_trace_qkv_epilogue.txt).  AUX = 0 has no load behind its first store: the stores stream.
        auto body = [&](auto auxc) __attribute__((always_inline)) {
        constexpr int AUX = decltype(auxc)::value & 3;
        constexpr bool BMAP = (decltype(auxc)::value >> 2) != 0;       // desc.ld_bias > 0: the bias is an [M, Cout] map (one more loaded operand)
        constexpr bool LD = AUX >= 1 || BMAP;
        // loads of UN row groups in flight together (the 128-accumulator-register tiles have little room: 2 at a time)
        // (two aux operands, each held for two groups: 2 rows at a time as well)
        constexpr int UN = (TM * TN >= 8 || TM * TN == 1 || AUX == 2 || BMAP) ? 2 : ITER;      // (one accumulator tile per wavefront: the <= 85-register tiles)
        constexpr int GPB = ITER / UN, NG = TM * GPB;                  // row groups per 32-row block, per wavefront
        // Aux operands are requested ONE GROUP AHEAD, in front of the previous group's stores: the wait for a group's operands then covers
        // the loads older than those stores, not the stores (same counter, in order) — otherwise every group would wait for the write
        // acknowledgements of the group before it.
        // (not in the one-accumulator-tile kernels, which run at <= 85 registers: there a group's operands are requested right before use)
        constexpr bool AHEAD = TM * TN > 1;
        float4 ax1[AHEAD ? 2 : 1][UN], ax2[AHEAD ? 2 : 1][UN], axb[AHEAD ? 2 : 1][UN];
        auto load_aux = [&](auto gc, float4 (&a1)[UN], float4 (&a2)[UN], float4 (&ab_)[UN]) __attribute__((always_inline)) {
            constexpr int gi = decltype(gc)::value, i = gi / GPB, c0 = (gi % GPB) * UN;
#pragma unroll
            for (int u0 = 0; u0 < UN; ++u0) {
                const int mt = bm0 + wm * WTM + i * 32 + r0 + (c0 + u0) * RPI;
                const bool okk = col_ok && mt < p.M;
                long m;
                int rem;
                bool okr;
                conv_row_of(p, mt, ps_ry, ps_rx, ps_col, m, rem, okr);
                const long m1 = d.aux_per_image ? (long)rem : m;
                // (out-of-range lanes read the zero page: the select is on the address, the loads stay back to back)
                if constexpr (AUX >= 1) a1[u0] = *reinterpret_cast<const float4*>(okk && (!dual || head1) ? p.aux1 + m1 * d.ld_aux1 + (a1_och ? och : co) : p.zero_page);
                if constexpr (AUX >= 2) {
                    const float* q2 = d.epi == FGT_EPI_PS_ADD2 ? p.aux2 + m * d.ld_aux2 + och : p.aux2 + (d.epi == FGT_EPI_AFFINE ? m1 : m) * d.ld_aux2 + co;
                    a2[u0] = *reinterpret_cast<const float4*>(okk && okr ? q2 : p.zero_page);
                }
                if constexpr (BMAP) ab_[u0] = *reinterpret_cast<const float4*>(okk ? p.cbias + m * d.ld_bias + co : p.zero_page);
            }
        };
        if constexpr (LD && AHEAD) load_aux(std::integral_constant<int, 0>{}, ax1[0], ax2[0], axb[0]);
        static_for<NG>([&](auto gc) {
            constexpr int gi = decltype(gc)::value, i = gi / GPB, c0 = (gi % GPB) * UN;
            if constexpr (gi % GPB == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) Ws[((e & 3) + 8 * (e >> 2) + 4 * lh) * WTN + j * 32 + l31] = acc[i][j][e];
                __builtin_amdgcn_wave_barrier();                       // the patch is exchanged between lanes of this wavefront only
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if constexpr (LD && AHEAD && gi + 1 < NG) load_aux(std::integral_constant<int, gi + 1>{}, ax1[(gi + 1) & 1], ax2[(gi + 1) & 1], axb[(gi + 1) & 1]);
            if constexpr (LD && !AHEAD) load_aux(gc, ax1[0], ax2[0], axb[0]);
            constexpr int ab = AHEAD ? (gi & 1) : 0;
            const int mrow = bm0 + wm * WTM + i * 32 + r0;
            float4 cv[UN];
            bool ok[UN];
#pragma unroll
            for (int u0 = 0; u0 < UN; ++u0) {
                const int it = c0 + u0;
                ok[u0] = col_ok && mrow + it * RPI < p.M;
                cv[u0] = *reinterpret_cast<const float4*>(Ws + (r0 + it * RPI) * WTN + c4 * 4);
            }
#pragma unroll
            for (int u0 = 0; u0 < UN; ++u0) {
                long m;
                int rem_;
                bool okr;
                conv_row_of(p, mrow + (c0 + u0) * RPI, ps_ry, ps_rx, ps_col, m, rem_, okr);
                const float4 bv = BMAP ? axb[ab][u0] : bi;
                float v[4] = {cv[u0].x * sc.x + bv.x, cv[u0].y * sc.y + bv.y, cv[u0].z * sc.z + bv.z, cv[u0].w * sc.w + bv.w};
                float x1[4] = {0.f, 0.f, 0.f, 0.f}, x2[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (AUX >= 1) { const float4 t = ax1[ab][u0]; x1[0] = t.x; x1[1] = t.y; x1[2] = t.z; x1[3] = t.w; }
                if constexpr (AUX >= 2) { const float4 t = ax2[ab][u0]; x2[0] = t.x; x2[1] = t.y; x2[2] = t.z; x2[3] = t.w; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float pre = v[u];
                    if constexpr (AUX == 2) {           // ABI 7: the two kinds applied in front of the activation
                        if (d.epi == FGT_EPI_AFFINE) pre = fmaf(pre, x2[u], x1[u]);
                        else if (d.epi == FGT_EPI_PS_ADD2) pre = pre + x1[u] + x2[u];
                    }
                    float x = fgt_act(pre, d.act, d.slope) * d.out_scale;
                    if constexpr (AUX == 1) {
                        if (d.epi == FGT_EPI_MUL) x *= (dual && !head1) ? 1.f : x1[u];
                        else x = fgt_act(x + x1[u], d.act2, d.slope);
                    }
                    if constexpr (AUX == 2) {
                        if (d.epi == FGT_EPI_GRU) x = (1.f - x1[u]) * x2[u] + x1[u] * x;
                    }
                    v[u] = x;
                }
                if (ok[u0] && okr) {
                    typedef float nt_f4 __attribute__((ext_vector_type(4)));
                    typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
                    const bool nt = p.nt_store != 0;
                    if (st_f32) {
                        float* o = p.out + out_goff + m * d.ldo + d.ooff + och;
                        if (nt) __builtin_nontemporal_store(nt_f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<nt_f4*>(o));
                        else *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                    if (st_split) {
                        const int cs = d.ooff_s + och;      // pso == 32: interleaved layout, channel c -> (c/32)*64 + c%32, lo 32 further
                        if (p.pso < 0) {                    // pso == -1: one fp16 plane (in_split = 3 of the consumer)
                            const uint2 h = fgt_half4(make_float4(v[0], v[1], v[2], v[3]));
                            __bf16* o = p.out_s + m * d.ldo_s + cs;
                            if (nt) __builtin_nontemporal_store(nt_u2{h.x, h.y}, reinterpret_cast<nt_u2*>(o));
                            else *reinterpret_cast<uint2*>(o) = h;
                        } else {
                            uint2 hi, lo;
                            split4(make_float4(v[0], v[1], v[2], v[3]), hi, lo);
                            __bf16* o = p.out_s + m * d.ldo_s + (p.pso == 32 ? ((cs >> 5) << 6) + (cs & 31) : cs);
                            if (nt) {
                                __builtin_nontemporal_store(nt_u2{hi.x, hi.y}, reinterpret_cast<nt_u2*>(o));
                                __builtin_nontemporal_store(nt_u2{lo.x, lo.y}, reinterpret_cast<nt_u2*>(o + p.pso));
                            } else {
                                *reinterpret_cast<uint2*>(o) = hi;
                                *reinterpret_cast<uint2*>(o + p.pso) = lo;
                            }
                        }
                    }
                }
            }
            if constexpr (gi % GPB == GPB - 1) __builtin_amdgcn_wave_barrier();     // the patch is rewritten by the next row block
        });
        };
        if (BIAS_MAP && d.ld_bias > 0) {       // a bias MAP (RAFT's GRU convs: the iteration-invariant context term, update.py:45-58): its own instances
            if constexpr (BIAS_MAP) {
                if (d.epi == FGT_EPI_NONE) body(std::integral_constant<int, 4>{});
                else if (d.epi >= FGT_EPI_GRU) body(std::integral_constant<int, 6>{});
                else body(std::integral_constant<int, 5>{});
            }
        } else if (d.epi == FGT_EPI_NONE) body(std::integral_constant<int, 0>{});
        else if (d.epi >= FGT_EPI_GRU) body(std::integral_constant<int, 2>{});
        else body(std::integral_constant<int, 1>{});
        return;
    }

    // ---- general path (odd channel counts, NCHW output): workgroup-wide staging
#pragma unroll
    for (int ps = 0; ps < EP_PASSES; ++ps) {
        if (ps > 0) __syncthreads();
        if (wm / WM_PER_PASS == ps) {
            const int wml = wm % WM_PER_PASS;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        Cs[(wml * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * BN + wn * WTN + j * 32 + l31] = acc[i][j][e];
        }
        __syncthreads();
        const int mbase = bm0 + ps * EP_BM;
        for (int idx = tid; idx < EP_BM * (BN / 4); idx += NT) {
            const int row = idx / (BN / 4), c4 = idx - row * (BN / 4);
            const int n = bn0 + c4 * 4;
            if (mbase + row >= p.M || n >= p.Cout_g) continue;
            const int co = g * p.Cout_g + n;
            int ps_ry = 0, ps_rx = 0, och = co, rem;
            bool okr;
            long m;
            const bool ps_col = d.ps_r ? conv_ps_column<PHASE>(d, n, ps_ry, ps_rx, och) : true;     // (host: sub-pixel output implies vec_ok)
            const bool dual = DUAL && d.dual_n0 > 0, head1 = dual && n >= d.dual_n0;                 // (host: two heads imply vec_ok)
            if (head1) och = co - d.dual_n0;
            conv_row_of(p, mbase + row, ps_ry, ps_rx, ps_col, m, rem, okr);
            if (!okr) continue;
            const long m1 = d.aux_per_image ? (long)rem : m;
            const float4 cv = *reinterpret_cast<const float4*>(Cs + row * BN + c4 * 4);
            float v[4] = {cv.x, cv.y, cv.z, cv.w};
            const int nvalid = min(4, p.Cout_g - n);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u < nvalid) {
                    const float cs = p.cscale ? p.cscale[co + u] : 1.f;
                    const float cb = !p.cbias ? 0.f : d.ld_bias > 0 ? p.cbias[m * d.ld_bias + co + u] : p.cbias[co + u];
                    float pre = v[u] * cs + cb;
                    if (d.epi == FGT_EPI_AFFINE) pre = fmaf(pre, p.aux2[m1 * d.ld_aux2 + co + u], p.aux1[m1 * d.ld_aux1 + co + u]);
                    else if (d.epi == FGT_EPI_PS_ADD2) pre = pre + p.aux1[m1 * d.ld_aux1 + co + u] + p.aux2[m * d.ld_aux2 + och + u];
                    float x = fgt_act(pre, d.act, d.slope) * d.out_scale;
                    if (d.epi == FGT_EPI_MUL) {
                        if (!dual || head1) x *= p.aux1[m1 * d.ld_aux1 + ((dual || (PHASE && d.ps_phase_pad)) ? och : co) + u];
                    } else if (d.epi == FGT_EPI_ADD) {
                        x = fgt_act(x + p.aux1[m1 * d.ld_aux1 + ((PHASE && d.ps_phase_pad) ? och : co) + u], d.act2, d.slope);
                    } else if (d.epi == FGT_EPI_GRU) {
                        const float z = p.aux1[m * d.ld_aux1 + co + u];
                        const float hh = p.aux2[m * d.ld_aux2 + co + u];
                        x = (1.f - z) * hh + z * x;
                    }
                    v[u] = x;
                }
            }
            if (vec_ok) {
                if (want_f32 && !head1) *reinterpret_cast<float4*>(p.out + out_goff + m * d.ldo + d.ooff + och) = make_float4(v[0], v[1], v[2], v[3]);
                if (want_split && (!dual || head1)) {       // validated by the host: vec_ok holds whenever out_split is set
                    const int cs = d.ooff_s + och;      // pso == 32: interleaved layout, channel c -> (c/32)*64 + c%32, lo 32 further
                    if (p.pso < 0) {                    // pso == -1: one fp16 plane
                        *reinterpret_cast<uint2*>(p.out_s + m * d.ldo_s + cs) = fgt_half4(make_float4(v[0], v[1], v[2], v[3]));
                    } else {
                        uint2 hi, lo;
                        split4(make_float4(v[0], v[1], v[2], v[3]), hi, lo);
                        __bf16* o = p.out_s + m * d.ldo_s + (p.pso == 32 ? ((cs >> 5) << 6) + (cs & 31) : cs);
                        *reinterpret_cast<uint2*>(o) = hi;
                        *reinterpret_cast<uint2*>(o + p.pso) = lo;
                    }
                }
            } else if (d.out_nchw) {
                const int n_img = m / p.HoWo, rem = m - n_img * p.HoWo;
                for (int u = 0; u < nvalid; ++u) p.out[((long)n_img * d.Cout + co + u) * p.HoWo + rem] = v[u];
            } else {
                for (int u = 0; u < nvalid; ++u) p.out[out_goff + m * d.ldo + d.ooff + co + u] = v[u];
            }
        }
    }
}

}  // namespace
