// Round 5.  bf16x3 convolution for the 4-CHANNEL INPUT layers (gfx950): the first conv of the frame encoder (3x3 stride 2 on RGB + mask,
// FGT/models/model.py:32), of the flow encoders (5x5 on 2 flow channels zero-padded to 4, model.py:207-209; LAFC/models/lafc.py:23-30) and of RAFT's
// motion encoder (7x7 on the flow, RAFT/update.py:64).  K = k*k*4 is 36 ... 196: two to seven K-steps of 32.  On the register-staged kernel
// (conv_igemm.hip) such a layer is all prologue and epilogue — gather into registers, split, store to LDS, barrier, a handful of MFMAs, barrier —
// 25-36 TFLOP/s algorithmic, 1.9 ms of the FGT step for 0.5 % of its flops.
//
// What bounds these layers is neither their stores (writing fp32 + split instead of one form costs +7 %: profiles/r05_run13_*) nor their MFMAs
// (0.04 ms of 0.37) but LATENCY: a workgroup is one dependent chain — weights in, gathers, a few MFMAs, transpose, stores — and only other
// wavefronts fill it.  The first version of this kernel (256 x 64 per workgroup, 208 registers, two workgroups per CU) removed the LDS staging
// and the K-loop barriers and measured exactly the register-staged kernel's time (profiles/r05_run12_*).  This one halves the tile instead:
// 128 x 64 on 4 wavefronts of 32 x 64 at 141 registers, THREE workgroups per CU: 0.303 vs 0.426 ms on the 5x5 layer (20 frames of 240x432), 0.064
// vs 0.085 on the 3x3 stride-2 one, 0.235 vs 0.344 on LAFC's, 0.110 vs 0.120 on RAFT's 7x7 (profiles/r05_run14_*).  (Four workgroups per CU
// would need <= 128 registers: the shared epilogue's instances spill there.)
//
// The A operand never touches the LDS.  An MFMA lane (row = lane & 31, k-half = lane >> 5) of v_mfma_f32_32x32x16_bf16 holds 8 consecutive k
// values = TWO taps x 4 channels of its output pixel: two 16-byte loads straight from the channels-last input (33 MB per launch, read 9-49
// times: L1 / L2 hits), split into hi / lo in registers.  The gathers of up to seven k16 steps are issued before the first MFMA of the chunk; there
// is no barrier in the K loop.  The weights of the workgroup's 64 output channels (16-56 KB as the usual per-step interleaved hi | lo image) are
// staged in the LDS once, rows padded by 16 bytes so that the 32 rows of a fragment read spread over the banks; the shared epilogue
// (conv_tile.h) does bias / activation / split outputs as for every other conv.
//
// Numerics: the same products as the other bf16x3 kernels (hi = bf16_rne(x), lo = bf16_rne(x - hi); lo*hi + hi*lo + hi*hi per k16 step, fp32
// accumulate) in the order of the packed K axis (ky, kx, ci): measured BIT-IDENTICAL to conv_igemm.hip's tiles on every geometry of
// tests/test_conv_c4_gpu.py, so it is one more candidate of the autotuner (tile code FGT_TILE_C4), not a route.
#include "conv_tile.h"

namespace {

constexpr int C4_BN = 64, C4_WM = 4;
#ifndef FGT_C4_TM
#define FGT_C4_TM 1          // 32-row blocks per wavefront: 1 = 128 x 64 per workgroup at <= 128 registers (3 workgroups per CU), 2 = 256 x 64 at two
#endif
constexpr int C4_TM = FGT_C4_TM, C4_BM = C4_WM * 32 * C4_TM;

template <int KS>
// (KS = 7 stages 64 x 912 B = 58 KB of weights: two workgroups per CU whatever the register count, so it is not held to the 3-workgroup cap)
__global__ void __launch_bounds__(256, C4_TM == 1 ? (KS == 7 ? 2 : 3) : 2) conv_c4_kernel(const ConvP p) {
    constexpr int BM = C4_BM, BN = C4_BN, WM = C4_WM, WN = 1, TM = C4_TM, TN = 2;
    constexpr int NTAP = KS * KS, K = NTAP * 4;
    constexpr int NK16 = (K + 15) / 16;                  // k16 steps that hold a tap (the padding of Kpad beyond them is zero weights: skipped)
    constexpr int NK32 = (K + 31) / 32;
    constexpr int BROW = NK32 * 128 + 16;                // bytes per weight row in the LDS (padded: 32 rows of a fragment read cover 32 x 16 B of distinct banks)
    constexpr int STAGE = 4096;                          // floats: the epilogue's view of the LDS (2 stages = 32 KB = four 32 x 64 wave patches)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN;

    // ---- the 64 weight rows of this N tile: [Kpad/32][hi 32 | lo 32] bf16 per row in global memory -> LDS rows of BROW bytes
    char* const Bs = reinterpret_cast<char*>(smem);
    {
        const char* w = reinterpret_cast<const char*>(p.w) + (long)bn0 * (2 * d.Kpad) * 2;      // row n: 2 * Kpad bf16 = 4 * Kpad bytes
        constexpr int V = NK32 * 8;                      // 16-byte vectors per row
        for (int i = tid; i < BN * V; i += 256) {
            const int r = i / V, v = i - r * V;
            *reinterpret_cast<uint4*>(Bs + r * BROW + v * 16) = *reinterpret_cast<const uint4*>(w + (long)r * (4 * d.Kpad) + v * 16);
        }
    }

    // ---- this lane's two output pixels (one per 32-row block) and the taps it carries: k = 16 s + 8 lh + (0..7) = taps 4 s + 2 lh + {0, 1}
    const float* xb[TM];
    int iy0[TM], ix0[TM];
    bool rowok[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = bm0 + wave * (32 * TM) + i * 32 + l31;
        rowok[i] = m < p.M;
        const int mm = rowok[i] ? m : 0;
        const int n = fgt_fastdiv(mm, p.div_howo), rem = mm - n * p.HoWo;
        const int oy = fgt_fastdiv(rem, p.div_wo), ox = rem - oy * d.Wo;
        iy0[i] = oy * d.sh - d.ph;
        ix0[i] = ox * d.sw - d.pw;
        xb[i] = p.x0 + ((long)n * d.H * d.W) * d.ld0 + d.off0;
    }
    const bool repl = d.pad_mode == 1, relu_in = d.in_relu != 0;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    __syncthreads();                                     // the weight rows are in the LDS

    // K in chunks of at most 7 k16 steps (14 gathers in flight per lane: the 7x7 layer's 13 steps at once would not fit the registers)
    constexpr int CH = 7, NCH = (NK16 + CH - 1) / CH;
    static_for<TM * NCH>([&](auto IC) {
        constexpr int i = decltype(IC)::value / NCH, s0 = (decltype(IC)::value % NCH) * CH;
        constexpr int NS = NK16 - s0 < CH ? NK16 - s0 : CH;
        float4 av[NS][2];
        // all gathers of the chunk first (zero padding: a select on the value, the load reads pixel 0 of the image; replicate: clamped coordinates)
        static_for<NS>([&](auto S) {
            constexpr int s = s0 + decltype(S)::value;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 4 * s + 2 * lh + u;
                const int ky = t / KS, kx = t - ky * KS;
                int iy = iy0[i] + ky * d.dh, ix = ix0[i] + kx * d.dw;
                bool ok = rowok[i] && t < NTAP;
                if (repl) {
                    iy = min(max(iy, 0), d.H - 1);
                    ix = min(max(ix, 0), d.W - 1);
                } else {
                    ok = ok && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                    iy = ok ? iy : 0;
                    ix = ok ? ix : 0;
                }
                float4 v = *reinterpret_cast<const float4*>(xb[i] + ((long)iy * d.W + ix) * d.ld0);
                if (relu_in) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                av[s - s0][u] = ok ? v : zero4;
            }
        });
        static_for<NS>([&](auto S) {
            constexpr int s = s0 + decltype(S)::value;
            uint2 h0, l0, h1, l1;
            split4(av[s - s0][0], h0, l0);
            split4(av[s - s0][1], h1, l1);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
            const bf16x8 al = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
            constexpr int boff = (s / 2) * 128 + (2 * (s % 2)) * 16;      // K-step of 32, k-half of it; + lh * 16 per lane, + 64 for lo
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const char* b = Bs + (j * 32 + l31) * BROW + boff + lh * 16;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(b);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(b + 64);
                // same products as the other bf16x3 kernels: lo*hi, hi*lo, hi*hi
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
            }
        });
        __builtin_amdgcn_sched_barrier(0);               // (the next chunk's gathers are not hoisted above this chunk's MFMAs: registers)
    });
    __syncthreads();                                     // every wavefront has read its last weight fragment: the LDS becomes the epilogue's scratch
    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN, false, false>(conv_epilogue_args(p), acc, smem, bm0, bn0, 0);      // (the 4-channel input layers: no bias maps, no two heads)
}

template <int KS>
int launch(const ConvP& p, hipStream_t s) {
    constexpr int NK32 = (KS * KS * 4 + 31) / 32;
    constexpr size_t bbytes = (size_t)C4_BN * (NK32 * 128 + 16);
    constexpr size_t smem = bbytes > 32768 ? bbytes : 32768;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_c4_kernel<KS>), (int)smem, lds_set, "conv_c4")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, C4_BM);
    q.ntiles = cdiv(p.Cout_g, C4_BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, 1);
    hipLaunchKernelGGL((conv_c4_kernel<KS>), grid, dim3(256), smem, s, q);
    return fgt_check_launch("conv_c4");
}

}  // namespace

// Layers this kernel serves (geometry only): bf16x3 on an fp32 single-source 4-channel input, square 3 / 5 / 7 kernel, interleaved weight image,
// more than 4 output channels, one group, no upsampling.  Tile code FGT_TILE_C4: the autotuner tries it next to the register-staged tiles.
bool fgt_conv_c4_eligible(const ConvP& p) {
    const fgt_conv_desc& d = p.d;
    return d.precision == FGT_PREC_BF16X3 && d.in_split == 0 && d.w_il == 1 && d.groups == 1 && d.C1 == 0 && d.C0 == 4 && d.ld0 % 4 == 0 && d.off0 % 4 == 0 &&
           d.kh == d.kw && (d.kh == 3 || d.kh == 5 || d.kh == 7) && !d.upsample && p.Cout_g > 4 && d.Kpad >= ((d.kh * d.kw * 4 + 31) / 32) * 32 && d.ld_bias == 0 && d.dual_n0 == 0;
}

int fgt_conv_c4_launch(const ConvP& p, hipStream_t s) {
    switch (p.d.kh) {
        case 3: return launch<3>(p, s);
        case 5: return launch<5>(p, s);
        default: return launch<7>(p, s);
    }
}
