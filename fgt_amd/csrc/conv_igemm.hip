// Implicit-GEMM convolution / GEMM for gfx950 on the matrix cores, register-staged loader (fp32 activations in HBM).
// (conv_split.hip is the sibling for activations that arrive pre-split into hi/lo bf16: LDS-DMA loader, same arithmetic.)
//
//   M = N*Ho*Wo output pixels, N = Cout per group, K = kh*kw*Cin_per_group (k = (ky*kw+kx)*Cg + ci).
//
// One workgroup computes a BM x BN tile with WM x WN wavefronts; each wavefront owns (BM/WM) x (BN/WN)
// outputs as TM x TN accumulators of v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 157 TF peak).
// Per K-step of 32: every thread gathers float4 runs of the channels-last input (im2col on the fly:
// stride, dilation, zero/replicate padding, nearest x2 upsample, two concatenated sources, optional
// ReLU on the gathered value) and float4 runs of the packed weights into registers while the previous
// step's LDS tiles feed the MFMAs; LDS is double buffered with one barrier per step.
// LDS tiles are [rows][33] floats: both the 4 x ds_write_b32 of a float4 run and the per-lane
// ds_read_b32 of an MFMA operand (row = lane&31, k = 2*kk + lane>>5) are bank-conflict free.
//
// Two arithmetic modes share the loader and the epilogue (desc.precision):
//   0  fp32      v_mfma_f32_32x32x2_f32 on fp32 LDS tiles ([rows][33] floats) — bit-exact fp32 FMA chains.
//   1  bf16x3    every fp32 operand is split on the way into LDS into hi = bf16(x) and lo = bf16(x - hi) (RNE, one
//                v_cvt_pk_bf16_f32 per pair) and each product is issued as three v_mfma_f32_32x32x16_bf16
//                (lo*hi + hi*lo + hi*hi, fp32 accumulate): ~2^-16 relative error per product at 16/3 = 5.3x the
//                fp32-MFMA rate.  LDS tiles are [rows][32] bf16 with XOR-swizzled 16-byte slots (conv_tile.h); the weights arrive
//                pre-split (planes, or interleaved per K-step: fgt_conv_desc.w_il).
#include <stdlib.h>
#include <vector>
#include "common.h"
#include "conv_params.h"

#include "conv_tile.h"

namespace {

template <int BM, int BN, int WM, int WN, int PREC, int MINW>
__global__ void __launch_bounds__(WM* WN * 64, MINW) conv_igemm_kernel(const ConvP p) {
    constexpr int NT = WM * WN * 64;
    constexpr int RPP = NT / 8;  // tile rows covered per pass of the loader
    constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    // stage size in floats: fp32 tiles, or hi+lo bf16 tiles (2 * LDB * 2 bytes = LDB floats per row)
    constexpr int STAGE = PREC == 0 ? (BM + BN) * LDS_LD : (BM + BN) * LDB;
    static_assert(A_IT >= 1 && B_IT >= 1 && TM >= 1 && TN >= 1, "tile too small for the thread count");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int q = tid & 7, r = tid >> 3;
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    // ---- per-thread im2col row state (fixed over the K loop)
    int a_iy0[A_IT], a_ix0[A_IT];
    int a_nb[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = bm0 + r + it * RPP;
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
    // k-decomposition of this thread's float4 column, advanced incrementally by BK per step
    int k_cur = q * 4;
    int tap = k_cur / p.Cg;
    int ci = k_cur - tap * p.Cg;
    int ky = tap / d.kw, kx = tap - ky * d.kw;

    // Out-of-range gathers read a zero page instead of being predicated: the select is on the ADDRESS, never on the
    // loaded value, so the tile fetch is straight-line code and the loads of a K-step are issued back to back
    // (a branch or select that consumes a loaded value makes the compiler wait for every load separately).
    const float* zp = p.zero_page;

    // fp32: float4 run q of row r.  bf16x3: the weights arrive pre-split ([2][groups][Npad][Kpad] bf16: hi plane, lo
    // plane); thread q < 4 fetches 8 hi values (16 bytes) of its row, q >= 4 the matching 8 lo values: no conversion.
    const float* wrow[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const long rowi = (long)g * d.Npad + bn0 + r + it * RPP;
        if constexpr (PREC == 0) wrow[it] = p.w + rowi * d.Kpad + q * 4;
        else if (d.w_il) wrow[it] = p.w + rowi * d.Kpad + (q >> 2) * 16 + (q & 3) * 4;   // interleaved rows of 2*Kpad bf16 = Kpad floats: [hi 32 | lo 32] per step
        else wrow[it] = p.w + ((q >> 2) * (long)d.groups * d.Npad * d.Kpad + rowi * d.Kpad) / 2 + (q & 3) * 4;   // in floats (2 bf16 each)
    }

    float4 va[A_IT], vb[B_IT];

    // Per-row gather bases for the current (tap, source) of this thread's k column: recomputed only when the column
    // moves to another tap or crosses from source 0 to source 1 (every Cin_g/32 K-steps for the heavy layers), so the
    // per-step address math is one add + one select per row.
    const float* a_base[A_IT];
    unsigned a_okmask = 0;
    int seg_end = 0;
    auto retap = [&]() {
        const bool in0 = ci < p.Cg0;
        const float* src = in0 ? p.x0 : p.x1;
        const int ld = in0 ? d.ld0 : d.ld1;
        const int chb = in0 ? d.off0 + g * p.Cg0 : d.off1 + g * p.Cg1 - p.Cg0;   // channel = chb + ci
        seg_end = in0 ? p.Cg0 : p.Cg;
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

    auto load_tiles = [&]() {
        // A: gather (straight-line: the select is on the address, never on the loaded value)
        const bool kval = k_cur < p.K;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const bool ok = kval && ((a_okmask >> it) & 1u);
            va[it] = *reinterpret_cast<const float4*>(ok ? a_base[it] + ci : zp);
        }
        // B: packed weights (zero padded to Npad x Kpad); prefetches past Kpad read the zero page
        const bool kb_ok = k_cur < d.Kpad;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            vb[it] = *reinterpret_cast<const float4*>(kb_ok ? wrow[it] : zp);
            wrow[it] += PREC == 0 ? BK : (d.w_il ? BK : BK / 2);
        }
        // advance the k decomposition
        k_cur += BK;
        ci += BK;
        if (ci >= seg_end) {
            while (ci >= p.Cg) {
                ci -= p.Cg;
                if (++kx == d.kw) { kx = 0; ++ky; }
            }
            retap();
        }
    };

    auto store_tiles = [&](int buf) {
        if (d.in_relu) {   // ReLU on the gathered values (ffn_base.py:40-45), applied where they are consumed anyway
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                va[it] = make_float4(fmaxf(va[it].x, 0.f), fmaxf(va[it].y, 0.f), fmaxf(va[it].z, 0.f), fmaxf(va[it].w, 0.f));
        }
        if constexpr (PREC == 0) {
            float* As = smem + buf * STAGE;
            float* Bs = As + BM * LDS_LD;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                float* dst = As + (r + it * RPP) * LDS_LD + q * 4;
                dst[0] = va[it].x; dst[1] = va[it].y; dst[2] = va[it].z; dst[3] = va[it].w;
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                float* dst = Bs + (r + it * RPP) * LDS_LD + q * 4;
                dst[0] = vb[it].x; dst[1] = vb[it].y; dst[2] = vb[it].z; dst[3] = vb[it].w;
            }
        } else {
            // [A_hi | A_lo | B_hi | B_lo], rows of LDB bf16; this thread's float4 -> 8-byte runs at column q*4
            __bf16* base = reinterpret_cast<__bf16*>(smem + buf * STAGE);
            __bf16* Ahi = base; __bf16* Alo = Ahi + BM * LDB; __bf16* Bhi = Alo + BM * LDB; __bf16* Blo = Bhi + BN * LDB;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                uint2 hi, lo;
                split4(va[it], hi, lo);
                const int row = r + it * RPP;
                const int o = row * LDB + swz(row, q >> 1) + (q & 1) * 4;
                *reinterpret_cast<uint2*>(Ahi + o) = hi;
                *reinterpret_cast<uint2*>(Alo + o) = lo;
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {      // already bf16: 16 bytes straight into the hi (q < 4) or lo plane
                const int row = r + it * RPP;
                __bf16* dst = (q < 4 ? Bhi : Blo) + row * LDB + swz(row, q & 3);
                // (built component-wise: a whole-struct float4 copy through this pointer keeps vb[] in scratch memory)
                const float4 t = vb[it];
                *reinterpret_cast<uint4*>(dst) = make_uint4(__builtin_bit_cast(unsigned, t.x), __builtin_bit_cast(unsigned, t.y),
                                                            __builtin_bit_cast(unsigned, t.z), __builtin_bit_cast(unsigned, t.w));
            }
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

    auto mfma_bf16 = [&](int buf, int ks) {
        const __bf16* base = reinterpret_cast<const __bf16*>(smem + buf * STAGE);
        // operand rows: wave-tile base (multiple of 32) + l31, so (row >> 2) & 3 == (l31 >> 2) & 3 for every fragment
        const int so = swz(l31, ks * 2 + lh);
        const __bf16* Ahi = base + (wm * WTM + l31) * LDB + so;
        const __bf16* Alo = Ahi + BM * LDB;
        const __bf16* Bhi = base + 2 * BM * LDB + (wn * WTN + l31) * LDB + so;
        const __bf16* Blo = Bhi + BN * LDB;
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            ah[i] = *reinterpret_cast<const bf16x8*>(Ahi + i * 32 * LDB);
            al[i] = *reinterpret_cast<const bf16x8*>(Alo + i * 32 * LDB);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bh[j] = *reinterpret_cast<const bf16x8*>(Bhi + j * 32 * LDB);
            bl[j] = *reinterpret_cast<const bf16x8*>(Blo + j * 32 * LDB);
        }
        // the three partial products of one accumulator are spread out: consecutive MFMAs never share an accumulator
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    };

    if (PREC == 1 && p.pipe) {
        // Software-pipelined schedule (one basic block per K-step, no branches): registers hold tile kt+1 while LDS holds
        // tile kt.  The split + LDS store of tile kt+1 and the global gathers of tile kt+2 sit between the two halves of
        // tile kt's MFMAs so that each wavefront overlaps its own VALU / LDS / VMEM work with its own matrix work.
        if constexpr (PREC == 1) {
            load_tiles();
            store_tiles(0);
            load_tiles();
            __syncthreads();
            for (int kt = 0; kt < p.nk; ++kt) {
                const int buf = kt & 1;
                mfma_bf16(buf, 0);
                store_tiles(buf ^ 1);
                load_tiles();
                mfma_bf16(buf, 1);
                __syncthreads();
            }
        }
    } else {
    load_tiles();
    store_tiles(0);
    __syncthreads();

    for (int kt = 0; kt < p.nk; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < p.nk;
        if (more) load_tiles();
        if constexpr (PREC == 0) {
            const float* Ab = smem + buf * STAGE + (wm * WTM + l31) * LDS_LD + lh;
            const float* Bb = smem + buf * STAGE + BM * LDS_LD + (wn * WTN + l31) * LDS_LD + lh;
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = Ab[i * 32 * LDS_LD + 2 * kk];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bb[j * 32 * LDS_LD + 2 * kk];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else {
            mfma_bf16(buf, 0);
            mfma_bf16(buf, 1);
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }
    }

    // (fp32 inputs: no two-headed layers, fgt_conv2d rejects them; bias maps reach this kernel in the exact-fp32 mode only — RAFT's GRU convs on
    //  fp32 maps — so the bf16x3-on-fp32-inputs instances are built without those bodies and launch() declines)
    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN, PREC == 0, false>(conv_epilogue_args(p), acc, smem, bm0, bn0, g);
}

template <int BM, int BN, int WM, int WN, int PREC, int MINW = 2>
int launch(const ConvP& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t smem = 2ul * (PREC == 0 ? (BM + BN) * LDS_LD : (BM + BN) * LDB) * sizeof(float);
    if (PREC != 0 && p.d.ld_bias > 0) { fgt_set_error("fgt_conv2d: bias maps on fp32 inputs are built for FGT_PREC_FP32 only (bf16x3 takes them on split inputs)"); return FGT_EINVAL; }
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, PREC, MINW>), (int)smem, lds_set, "conv_igemm")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, PREC, MINW>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_igemm");
}

template <int PREC>
int launch_tile(int tile, const ConvP& p, hipStream_t s) {
    switch (tile) {
        case FGT_TILE_128x128: return launch<128, 128, 2, 2, PREC>(p, s);
        case FGT_TILE_128x64: return launch<128, 64, 2, 2, PREC>(p, s);
        case FGT_TILE_64x64: return launch<64, 64, 2, 2, PREC>(p, s);
        case FGT_TILE_128x32: return launch<128, 32, 4, 1, PREC>(p, s);
        case FGT_TILE_256x128: return launch<256, 128, 4, 2, PREC>(p, s);
        case FGT_TILE_128x128x8: return launch<128, 128, 2, 4, PREC, 4>(p, s);   // 8 wavefronts of 64x32, <= 128 VGPRs: 4 waves/SIMD
        case FGT_TILE_256x128x16: return launch<256, 128, 4, 4, PREC, 4>(p, s); // 16 wavefronts of 64x32
        case FGT_TILE_256x64x8: return launch<256, 64, 4, 2, PREC, 2>(p, s);     // Cout = 64 layers: 8 wavefronts of 64x32
        default: fgt_set_error("fgt_conv2d: unknown tile %d", tile); return FGT_EINVAL;
    }
}

}  // namespace

// The tap-reusing kernel (conv_taps.hip) serves every layer fgt_conv_taps_eligible accepts: decided by the layer's geometry, never by tuning
// (its accumulation order differs from the other kernels').  FGT_CONV_TAPS=0 turns the routing off (A/B measurements).
static bool taps_enabled() {
    static const int taps_env = [] { const char* e = getenv("FGT_CONV_TAPS"); return e ? atoi(e) : 1; }();
    return taps_env != 0;
}

static void conv_params(const fgt_conv_desc& d, ConvP& p) {
    p.d = d;
    p.Cg0 = d.C0 / d.groups; p.Cg1 = d.C1 / d.groups; p.Cg = p.Cg0 + p.Cg1;
    p.K = d.kh * d.kw * p.Cg; p.Cout_g = d.Cout / d.groups;
}

extern "C" int fgt_conv_taps_route(const fgt_conv_desc* dd) {
    if (!dd || !taps_enabled() || dd->groups <= 0 || dd->tile != 0) return 0;
    ConvP p{};
    conv_params(*dd, p);
    return fgt_conv_taps_preferred(p) ? 1 : 0;
}

extern "C" int fgt_conv2d(const fgt_conv_desc* dd, const void* x0v, const void* x1v, const float* w_packed,
                          const float* cscale, const float* cbias, const float* aux1, const float* aux2,
                          float* out, void* out_s, void* stream) {
    FGT_REQUIRE(dd && x0v && w_packed, "fgt_conv2d: null pointer");
    const float* x0 = static_cast<const float*>(x0v);
    const float* x1 = static_cast<const float*>(x1v);
    ConvP p;
    p.d = *dd;
    const fgt_conv_desc& d = p.d;
    FGT_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.Cout > 0 && d.groups > 0, "fgt_conv2d: bad sizes");
    FGT_REQUIRE(d.C0 > 0 && d.C0 % d.groups == 0 && d.C1 % d.groups == 0 && d.Cout % d.groups == 0,
                "fgt_conv2d: channels (%d,%d,%d) not divisible by groups %d", d.C0, d.C1, d.Cout, d.groups);
    p.Cg0 = d.C0 / d.groups; p.Cg1 = d.C1 / d.groups; p.Cg = p.Cg0 + p.Cg1; p.Cout_g = d.Cout / d.groups;
    const int gran = d.in_split ? 8 : 4;    // elements per 16-byte gather
    FGT_REQUIRE(d.in_split >= 0 && d.in_split <= 3, "fgt_conv2d: in_split must be 0, 1, 2 or 3");
    FGT_REQUIRE(d.precision >= 0 && d.precision <= 2, "fgt_conv2d: unknown precision %d", d.precision);
    FGT_REQUIRE((d.in_split == 3) == (d.precision == FGT_PREC_F16), "fgt_conv2d: FGT_PREC_F16 and fp16 inputs (in_split = 3) go together");
    if (d.in_split == 2)
        FGT_REQUIRE(p.Cg0 % 32 == 0 && p.Cg1 % 32 == 0 && d.off0 % 32 == 0 && d.off1 % 32 == 0 && d.ld0 % 64 == 0 && (d.C1 == 0 || d.ld1 % 64 == 0),
                    "fgt_conv2d: interleaved split inputs need Cin/groups and offsets multiples of 32, row strides multiples of 64");
    FGT_REQUIRE(d.w_il == 0 || d.precision == FGT_PREC_BF16X3, "fgt_conv2d: w_il needs FGT_PREC_BF16X3");
    FGT_REQUIRE(p.Cg0 % gran == 0 && p.Cg1 % gran == 0, "fgt_conv2d: per-group channels (%d,%d) must be multiples of %d (pad the tensor)", p.Cg0, p.Cg1, gran);
    FGT_REQUIRE(d.ld0 % gran == 0 && d.off0 % gran == 0 && (d.C1 == 0 || (x1 && d.ld1 % gran == 0 && d.off1 % gran == 0)),
                "fgt_conv2d: source strides/offsets must be multiples of %d elements", gran);
    if (d.in_split) {
        FGT_REQUIRE(d.in_split == 3 || d.precision == FGT_PREC_BF16X3, "fgt_conv2d: split inputs need FGT_PREC_BF16X3");
        FGT_REQUIRE(d.in_relu == 0, "fgt_conv2d: in_relu cannot be applied to split inputs (the producer applies it)");
        FGT_REQUIRE(d.in_split >= 2 || (d.ps0 % 8 == 0 && d.ps0 > 0 && (d.C1 == 0 || (d.ps1 % 8 == 0 && d.ps1 > 0))), "fgt_conv2d: plane strides must be positive multiples of 8");
        FGT_REQUIRE(d.in_split != 3 || (d.w_il == 0 && (d.Kpad % 64 == 0 || (p.Cout_g <= 4 && d.tile == 0))), "fgt_conv2d: FGT_PREC_F16 takes the plain fp16 weight image with Kpad %% 64 == 0");
    }
    FGT_REQUIRE(d.out_split >= 0 && d.out_split <= 2, "fgt_conv2d: out_split must be 0, 1 or 2");
    FGT_REQUIRE(d.out_split == 1 || out != nullptr, "fgt_conv2d: null output");
    if (d.out_split) {
        FGT_REQUIRE(out_s != nullptr && ((uintptr_t)out_s & 7) == 0, "fgt_conv2d: out_split needs an 8-byte aligned out_s");
        FGT_REQUIRE(p.Cout_g % 4 == 0 && !d.out_nchw && d.ldo_s % 4 == 0 && d.ooff_s % 4 == 0 && (d.pso == -1 || (d.pso % 4 == 0 && d.pso > 0)),
                    "fgt_conv2d: out_split needs Cout/groups, ldo_s, ooff_s, pso multiples of 4 (pso = -1: fp16 plane) and NHWC output");
        FGT_REQUIRE(d.out_split == 1 || (d.ldo % 4 == 0 && d.ooff % 4 == 0), "fgt_conv2d: out_split = 2 needs ldo, ooff multiples of 4");
        FGT_REQUIRE(d.pso != 32 || ((p.Cout_g % 32 == 0 || d.ps_r > 0) && d.ooff_s % 32 == 0 && d.ldo_s % 64 == 0),
                    "fgt_conv2d: interleaved out_s (pso = 32) needs Cout/groups, ooff_s multiples of 32 and ldo_s a multiple of 64");
        FGT_REQUIRE(d.epi == FGT_EPI_NONE || d.ld_aux1 % 4 == 0, "fgt_conv2d: out_split needs ld_aux1 % 4 == 0");
        FGT_REQUIRE(d.epi < FGT_EPI_GRU || d.ld_aux2 % 4 == 0, "fgt_conv2d: out_split needs ld_aux2 % 4 == 0");
    }
    FGT_REQUIRE(((uintptr_t)x0 & 15) == 0 && ((uintptr_t)x1 & 15) == 0 && ((uintptr_t)w_packed & 15) == 0,
                "fgt_conv2d: pointers must be 16-byte aligned");
    FGT_REQUIRE(d.kh > 0 && d.kw > 0 && d.sh > 0 && d.sw > 0 && d.dh > 0 && d.dw > 0 && d.ph >= 0 && d.pw >= 0, "fgt_conv2d: bad kernel geometry");
    p.Hin = d.H * (d.upsample ? 2 : 1); p.Win = d.W * (d.upsample ? 2 : 1);
    // (ABI 9, ps_phase_pad: the 2x2 sub-pixel form of "nearest x2 + 3x3": padding (1 - a, 1 - b) per sub-pixel, one output row per input pixel)
    const int Ho = d.ps_phase_pad ? p.Hin : (p.Hin + 2 * d.ph - d.dh * (d.kh - 1) - 1) / d.sh + 1;
    const int Wo = d.ps_phase_pad ? p.Win : (p.Win + 2 * d.pw - d.dw * (d.kw - 1) - 1) / d.sw + 1;
    FGT_REQUIRE(Ho == d.Ho && Wo == d.Wo, "fgt_conv2d: output size (%d,%d) != expected (%d,%d)", d.Ho, d.Wo, Ho, Wo);
    p.K = d.kh * d.kw * p.Cg;
    FGT_REQUIRE(d.Kpad % BK == 0 && d.Kpad >= p.K, "fgt_conv2d: Kpad %d invalid for K %d", d.Kpad, p.K);
    FGT_REQUIRE(d.Npad % 128 == 0 && d.Npad >= p.Cout_g, "fgt_conv2d: Npad %d invalid for Cout/groups %d", d.Npad, p.Cout_g);
    FGT_REQUIRE(d.epi >= FGT_EPI_NONE && d.epi <= FGT_EPI_PS_ADD2, "fgt_conv2d: unknown epilogue %d", d.epi);
    if (d.epi != FGT_EPI_NONE) FGT_REQUIRE(aux1 != nullptr, "fgt_conv2d: epilogue needs aux1");
    if (d.epi >= FGT_EPI_GRU) FGT_REQUIRE(aux2 != nullptr, "fgt_conv2d: GRU / affine / sub-pixel-add epilogues need aux2");
    // ---- ABI 7: fold as a convolution (sub-pixel output, per-image aux tables, ky = 0 skipping)
    FGT_REQUIRE(d.ps_r >= 0 && d.ky_skip_n0 >= 0 && (d.aux_per_image == 0 || d.aux_per_image == 1), "fgt_conv2d: bad ps_r / ky_skip_n0 / aux_per_image");
    FGT_REQUIRE(d.epi != FGT_EPI_PS_ADD2 || d.ps_r > 0, "fgt_conv2d: FGT_EPI_PS_ADD2 needs the sub-pixel output (ps_r > 0)");
    FGT_REQUIRE(!(d.ps_r || d.aux_per_image) || d.kw > 1 || d.kh == 1, "fgt_conv2d: sub-pixel output / per-image aux tables are not built for k x 1 layers (transposed tile order)");
    FGT_REQUIRE(!d.aux_per_image || d.epi != FGT_EPI_NONE, "fgt_conv2d: aux_per_image without an epilogue operand");
    FGT_REQUIRE(d.epi < FGT_EPI_AFFINE || (d.ld_aux1 % 4 == 0 && d.ld_aux2 % 4 == 0 && p.Cout_g % 4 == 0), "fgt_conv2d: affine / sub-pixel-add epilogues need ld_aux1, ld_aux2, Cout/groups multiples of 4");
    if (d.ps_r) {
        FGT_REQUIRE(d.groups == 1 && !d.out_nchw && d.ps_c > 0 && d.ps_c % 4 == 0 && d.ps_g0 >= d.ps_r * d.ps_c && d.ps_g0 % 4 == 0 &&
                    d.Cout == d.ps_g0 + (d.ps_r - 1) * d.ps_r * d.ps_c, "fgt_conv2d: sub-pixel output needs groups = 1, NHWC, ps_c %% 4 == 0 and Cout = ps_g0 + (r-1)*r*ps_c (got r %d, c %d, g0 %d, Cout %d)", d.ps_r, d.ps_c, d.ps_g0, d.Cout);
        FGT_REQUIRE(d.ps_H > 0 && d.ps_W > 0 && d.ps_H <= d.ps_r * Ho && d.ps_W <= d.ps_r * Wo, "fgt_conv2d: sub-pixel output map %dx%d does not fit %d x the %dx%d grid", d.ps_H, d.ps_W, d.ps_r, Ho, Wo);
        FGT_REQUIRE((d.out_split == 1 || (d.ldo % 4 == 0 && d.ooff % 4 == 0)) && (d.epi == FGT_EPI_NONE || d.ld_aux1 % 4 == 0), "fgt_conv2d: sub-pixel output needs float4-aligned rows");
        FGT_REQUIRE(!d.out_split || d.pso != 32 || d.ps_c % 32 == 0, "fgt_conv2d: interleaved sub-pixel out_s needs ps_c %% 32 == 0");
    }
    FGT_REQUIRE(!(d.ps_r || d.aux_per_image || d.epi >= FGT_EPI_AFFINE) || p.Cout_g > 4, "fgt_conv2d: the Cout <= 4 kernels have no sub-pixel / per-image-table epilogue");
    FGT_REQUIRE(d.ld_bias >= 0 && (d.tile_order == 0 || d.tile_order == 1) && (d.ld_bias == 0 || (cbias != nullptr && p.Cout_g > 4 && d.ld_bias % 4 == 0 && d.ld_bias >= d.Cout && !d.ps_r && d.in_split != 3)),
                "fgt_conv2d: a bias map (ld_bias > 0) needs cbias, Cout/groups > 4, ld_bias %% 4 == 0, ld_bias >= Cout, no sub-pixel output, no fp16 inputs");
    FGT_REQUIRE(d.ky_skip_n0 == 0 || (d.groups == 1 && d.kh >= 2 && d.upsample == 0), "fgt_conv2d: ky_skip_n0 needs groups = 1, kh >= 2, no upsampling");
    // ---- ABI 8: two heads in one convolution; batched GEMM on the wide kernel
    FGT_REQUIRE(d.dual_n0 >= 0 && d.ps_phase_pad >= 0, "fgt_conv2d: bad dual_n0 / ps_phase_pad");
    // ---- ABI 9: nearest x2 + 3x3 as a 2x2 convolution with a sub-pixel output and per-sub-pixel padding
    if (d.ps_phase_pad)
        FGT_REQUIRE(d.ps_phase_pad <= d.ps_c && d.ps_phase_pad % 4 == 0 && (!d.out_split || d.pso != 32 || d.ps_phase_pad % 32 == 0) &&
                    d.ps_r == 2 && d.ps_g0 == 2 * d.ps_c && d.ps_c % 64 == 0 && d.Cout == 4 * d.ps_c && d.kh == 2 && d.kw == 2 && d.ph == 1 && d.pw == 1 &&
                    d.sh == 1 && d.sw == 1 && d.dh == 1 && d.dw == 1 && !d.upsample && d.pad_mode == 0 && d.groups == 1 && (d.in_split == 1 || d.in_split == 2) &&
                    d.ps_H == 2 * d.H && d.ps_W == 2 * d.W && !d.aux_per_image && !d.dual_n0 && d.ld_bias == 0 && !d.ky_skip_n0 && d.tile_order == 0 &&
                    (d.epi == FGT_EPI_NONE || d.epi == FGT_EPI_MUL || d.epi == FGT_EPI_ADD) && d.tile < FGT_TILE_TAPS,
                    "fgt_conv2d: ps_phase_pad = cv needs cv <= ps_c, cv %% 4 == 0 (%% 32 with an interleaved out_s), ps_r = 2, ps_g0 = 2 * ps_c, ps_c %% 64 == 0, Cout = 4 * ps_c, a 2x2 / stride 1 / pad 1 geometry with zero padding, "
                    "groups = 1, split inputs (in_split = 1 or 2), ps_H x ps_W = 2H x 2W, no epilogue beyond mul / add, and a conv_split / conv_wide tile");
    if (d.dual_n0 > 0)
        FGT_REQUIRE(d.epi == FGT_EPI_MUL && d.out_split == 2 && d.groups == 1 && d.Cout == 2 * d.dual_n0 && d.dual_n0 % 64 == 0 &&
                    !d.ps_r && !d.aux_per_image && !d.out_nchw && (d.in_split == 1 || d.in_split == 2),
                    "fgt_conv2d: two heads (dual_n0 = %d) need FGT_EPI_MUL, out_split = 2, groups = 1, Cout = 2 * dual_n0, dual_n0 %% 64 == 0, NHWC, split inputs (in_split = 1 or 2), no sub-pixel output", d.dual_n0);
    const bool gb = d.gb_x0 != 0 || d.gb_w != 0 || d.gb_o != 0;
    if (gb)
        FGT_REQUIRE(d.gb_x0 > 0 && d.gb_w > 0 && d.gb_o > 0 && d.gb_x0 % 8 == 0 && d.gb_w % 8 == 0 && d.gb_o % 4 == 0 && d.in_split == 2 && d.w_il == 1 &&
                    d.tile >= FGT_TILE_WIDE && d.tile < FGT_TILE_TAPS && d.kh == 1 && d.kw == 1 && d.sh == 1 && d.sw == 1 && d.ph == 0 && d.pw == 0 && d.C1 == 0 && d.N == 1 &&
                    d.out_split == 0 && !d.out_nchw && d.epi == FGT_EPI_NONE && !cscale && !cbias && !d.ps_r && !d.upsample && p.Cout_g % 8 == 0 && d.ldo % 4 == 0 && d.ooff % 4 == 0,
                    "fgt_conv2d: the batched GEMM mode (gb_*) needs interleaved operands (in_split = 2, w_il = 1), an explicit wide tile (100..199), a 1 x 1 / stride 1 / single-source / "
                    "single-image geometry, plain fp32 output without epilogue, scale or bias, Cout/groups %% 8 == 0 and strides that are multiples of 8 (gb_x0, gb_w) / 4 (gb_o, ldo, ooff)");
    const long M = (long)d.N * Ho * Wo;
    FGT_REQUIRE(M < (1l << 31), "fgt_conv2d: M too large");
    p.M = (int)M; p.HoWo = Ho * Wo; p.nk = d.Kpad / (d.in_split == 3 ? 64 : BK);
    p.div_howo = fgt_fastdiv_make((unsigned)(Ho * Wo)); p.div_wo = fgt_fastdiv_make((unsigned)Wo);
    static const int pipe_env = [] { const char* e = getenv("FGT_CONV_PIPE"); return e ? atoi(e) : 1; }();
    p.pipe = pipe_env;
    static const int xcd_env = [] { const char* e = getenv("FGT_CONV_XCD"); return e ? atoi(e) : 1; }();
    p.xcd_swizzle = xcd_env;
    static const int nt_env = [] { const char* e = getenv("FGT_CONV_NT"); return e ? atoi(e) : 1; }();
    p.nt_store = nt_env;

    p.zero_page = fgt_zero_page();
    FGT_REQUIRE(p.zero_page != nullptr, "fgt_conv2d: could not allocate the zero page");
    p.x0 = x0; p.x1 = x1 ? x1 : x0; p.w = w_packed; p.cscale = cscale; p.cbias = cbias; p.aux1 = aux1; p.aux2 = aux2; p.out = out;
    p.out_s = static_cast<__bf16*>(out_s); p.pso = d.pso; p.ps0 = d.ps0; p.ps1 = d.C1 ? d.ps1 : d.ps0;

    int tile = d.tile;
    // tile = 0 or a +200 code on a layer conv_taps.hip serves: that kernel (geometry decides, see taps_enabled); an explicit tile of another
    // family on such a layer selects that family (A/B measurements, tests)
    const bool taps = d.ps_phase_pad ? false : d.tile >= FGT_TILE_TAPS ? fgt_conv_taps_eligible(p) : (d.tile == 0 && taps_enabled() && fgt_conv_taps_preferred(p));
    FGT_REQUIRE(d.tile < FGT_TILE_TAPS || taps, "fgt_conv2d: tile %d (tap-reusing kernel) on a layer it does not serve", d.tile);
    if (taps && tile == 0) tile = FGT_TILE_TAPS + (p.Cout_g <= 192 ? FGT_TILE_128x64 : FGT_TILE_128x128x8);
    FGT_REQUIRE(!taps || d.w_il == (tile >= FGT_TILE_TAPS_BREG ? 2 : 1), "fgt_conv2d: tile %d takes weights with w_il = %d", tile, tile >= FGT_TILE_TAPS_BREG ? 2 : 1);
#ifndef FGT_DIAG
    FGT_REQUIRE(tile < FGT_TILE_TAPS_BREG, "fgt_conv2d: tile %d (register-fed weights) exists in diagnostic builds only", tile);
#endif
    if (tile == 0 && d.ps_phase_pad) tile = d.ps_c % 128 == 0 ? FGT_TILE_128x128 : FGT_TILE_128x64;       // (the tile's N width must divide ps_c)
    if (tile == 0) {
        // static fallback (profiles/r01_run2_tune_conv_*.txt); fgt_amd.ops autotunes per shape on first use
        if (p.Cout_g <= 32) tile = FGT_TILE_128x32;
        else if (p.Cout_g <= 64) tile = FGT_TILE_64x64;
        else {
            const long blocks128 = (long)cdiv(M, 128) * cdiv(p.Cout_g, 128) * d.groups;
            tile = blocks128 >= (d.precision == 0 ? 1024 : 512) ? FGT_TILE_128x128 : FGT_TILE_64x64;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    // Cout <= 4: fp32 VALU kernels (conv_direct.hip).  With fp16 inputs (in_split = 3) only the LDS-tiled 3x3 kernel exists; w_packed is then
    // the fp32 weight image (arithmetic and weights stay fp32: what is fp16 is the feature map in HBM)
    const bool direct = d.tile == 0 && d.out_split == 0 && (d.precision == 0 || (d.in_split == 3 && p.Cout_g <= 4)) && fgt_conv_direct_eligible(p);
    FGT_REQUIRE(!(d.in_split == 3 && p.Cout_g <= 4 && d.tile == 0) || direct, "fgt_conv2d: fp16 inputs with Cout/groups <= 4 need the 3x3 / stride 1 / pad 1 geometry of the LDS-tiled kernel and fp32 output");
    // roofline accounting is about the MFMA kernels only; algorithmic flops use the UNPADDED K (flow 2 -> 4, RGB 3 -> 4 channel
    // padding is not work the reference does): desc.k_alg = kh*kw*Cin_real/groups, 0 = the padded K
    // unique-byte floor: input map(s) once (4 B per value, fp32 or hi + lo), weights once, every output form once, aux operands once
    // (fp16 tensors — in_split = 3, out_s with pso = -1 — are 2 B per value)
    const double in_b = d.in_split == 3 ? 2.0 : 4.0, os_b = (d.out_split && d.pso < 0) ? 2.0 : 4.0;
    // (per-image aux tables are Ho*Wo rows; a sub-pixel map holds ps_H*ps_W*ps_c values per image, and its PS_ADD2 residual as many)
    const double out_vals = d.ps_r ? (double)d.N * d.ps_H * d.ps_W * (d.ps_phase_pad ? d.ps_phase_pad : d.ps_c) : (double)M * d.Cout;
    const double aux1_vals = d.epi == FGT_EPI_NONE ? 0.0 : d.ps_phase_pad ? out_vals : (d.aux_per_image ? (double)Ho * Wo * d.Cout : (double)M * d.Cout);
    const double aux2_vals = d.epi < FGT_EPI_GRU ? 0.0 : d.epi == FGT_EPI_PS_ADD2 ? out_vals : (d.epi == FGT_EPI_AFFINE && d.aux_per_image ? (double)Ho * Wo * d.Cout : (double)M * d.Cout);
    const double bmap_bytes = d.ld_bias > 0 ? 4.0 * (double)M * d.Cout : 0.0;
    // (two heads: each output form and aux1 cover half of the columns)
    const double conv_bytes = bmap_bytes + in_b * ((double)d.N * d.H * d.W * (d.C0 + d.C1) + (double)d.Cout * p.K) +
                              (d.dual_n0 > 0 ? out_vals * 0.5 * (4.0 + os_b) + 2.0 * aux1_vals
                                             : out_vals * ((d.out_split != 1 ? 4.0 : 0.0) + (d.out_split ? os_b : 0.0)) + 4.0 * (aux1_vals + aux2_vals));
    // (Cout <= 4 VALU kernels: an HBM-bound pass over the input map — their own kind, bytes only)
    const int prof = direct ? fgt_prof_begin(FGT_PROF_CONV_SMALL, 0.0, conv_bytes, s) : fgt_prof_begin(FGT_PROF_CONV, 2.0 * (double)M * (d.n_alg > 0 ? d.n_alg : p.Cout_g) * (d.k_alg > 0 ? d.k_alg : p.K) * d.groups, conv_bytes, s);
    int rc;
    if (direct) rc = fgt_conv_direct(p, s);   // Cout <= 4: VALU direct conv (fp32)
    else if (d.in_split == 3) rc = fgt_conv_f16_launch(tile, p, s);
#ifdef FGT_DIAG
    else if (taps && tile >= FGT_TILE_TAPS_BREG) rc = fgt_conv_taps_breg_launch(tile - FGT_TILE_TAPS_BREG, p, s);
#endif
    else if (taps) rc = fgt_conv_taps_launch(tile - FGT_TILE_TAPS, p, s);
    else if (tile == FGT_TILE_C4) {
        if (d.ps_phase_pad || !fgt_conv_c4_eligible(p)) { fgt_set_error("fgt_conv2d: tile %d (4-channel-input kernel) on a layer it does not serve", tile); rc = FGT_EINVAL; }
        else rc = fgt_conv_c4_launch(p, s);
    }
    else if (d.in_split == 2 && tile >= FGT_TILE_WIDE) {
        if (d.w_il != 1 || d.Kpad != p.K) { fgt_set_error("fgt_conv2d: the wide bf16x3 tiles need interleaved weights (w_il = 1) and K %% 32 == 0"); rc = FGT_EINVAL; }
        else rc = fgt_conv_wide_launch(tile - FGT_TILE_WIDE, p, s);
    }
    else if (d.in_split) rc = fgt_conv_split_launch(tile, p, s);
    else if (tile >= FGT_TILE_256x128x8_S3) { fgt_set_error("fgt_conv2d: tile %d needs split inputs", tile); rc = FGT_EINVAL; }
    else rc = d.precision == 0 ? launch_tile<0>(tile, p, s) : launch_tile<1>(tile, p, s);
    fgt_prof_end(prof, s);
    return rc;
}
