// Direct convolution for very small output-channel counts (Cout <= 4: FGT decoder.final 64->3, LAFC decoder.2 24->2,
// edge head 16->1, RAFT flow head 256->2).  An implicit-GEMM tile would waste >90 % of a 32-wide MFMA column block on
// such layers; here LPP = Cin/4 (rounded up to a power of two) lanes share one output pixel, each lane keeps the
// weights of its own 4 input channels for every tap in registers, gathers one float4 per tap (a pixel's channels are
// read by adjacent lanes: fully coalesced) and the Cout partial sums are combined with xor-shuffles.  fp32 VALU FMAs,
// HBM-bound: bytes = input map once + output.
#include "common.h"
#include "conv_params.h"

namespace {

template <int TAPS, int COUT>
__global__ void __launch_bounds__(256) conv_direct_kernel(const ConvP p, int lpp_log2) {
    const fgt_conv_desc& d = p.d;
    const int lpp = 1 << lpp_log2;
    const int sub = threadIdx.x & (lpp - 1);          // which float4 of the input channels
    const int c4n = p.Cg >> 2;
    const bool live = sub < c4n;
    // this lane's weights: w[tap][co][4]
    float w[TAPS][COUT][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live && co < p.Cout_g) v = *reinterpret_cast<const float4*>(p.w + (long)co * d.Kpad + t * p.Cg + sub * 4);
            w[t][co][0] = v.x; w[t][co][1] = v.y; w[t][co][2] = v.z; w[t][co][3] = v.w;
        }
    const int ppb = 256 >> lpp_log2;                    // pixels per block iteration
    const int ush = d.upsample ? 1 : 0;
    for (long m0 = (long)blockIdx.x * ppb; m0 < p.M; m0 += (long)gridDim.x * ppb) {
        const long m = m0 + (threadIdx.x >> lpp_log2);
        const bool mval = m < p.M;
        int n_img = 0, oy = 0, ox = 0;
        if (mval) {
            n_img = (int)(m / p.HoWo);
            const int rem = (int)(m - (long)n_img * p.HoWo);
            oy = rem / d.Wo; ox = rem - oy * d.Wo;
        }
        float acc[COUT];
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int ky = t / d.kw, kx = t - ky * d.kw;
            int iy = oy * d.sh - d.ph + ky * d.dh, ix = ox * d.sw - d.pw + kx * d.dw;
            if (d.pad_mode) { iy = min(max(iy, 0), p.Hin - 1); ix = min(max(ix, 0), p.Win - 1); }
            const bool ok = live && mval && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const int ci = sub * 4;
            const float* src; int ld, ch;
            if (ci < p.Cg0) { src = p.x0; ld = d.ld0; ch = d.off0 + ci; } else { src = p.x1; ld = d.ld1; ch = d.off1 + ci - p.Cg0; }
            const long off = ok ? ((long)(n_img * d.H * d.W + (iy >> ush) * d.W + (ix >> ush)) * ld + ch) : 0l;
            float4 v = *reinterpret_cast<const float4*>(src + off);
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (d.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
#pragma unroll
            for (int co = 0; co < COUT; ++co)
                acc[co] += (v.x * w[t][co][0] + v.y * w[t][co][1]) + (v.z * w[t][co][2] + v.w * w[t][co][3]);
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co)
            for (int o = lpp >> 1; o > 0; o >>= 1) acc[co] += __shfl_xor(acc[co], o);
        if (sub == 0 && mval) {
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                if (co >= p.Cout_g) break;
                const float cs = p.cscale ? p.cscale[co] : 1.f, cb = p.cbias ? p.cbias[co] : 0.f;
                float x = fgt_act(acc[co] * cs + cb, d.act, d.slope) * d.out_scale;
                if (d.epi == FGT_EPI_MUL) x *= p.aux1[m * d.ld_aux1 + co];
                else if (d.epi == FGT_EPI_ADD) x = fgt_act(x + p.aux1[m * d.ld_aux1 + co], d.act2, d.slope);
                else if (d.epi == FGT_EPI_GRU) { const float z = p.aux1[m * d.ld_aux1 + co], hh = p.aux2[m * d.ld_aux2 + co]; x = (1.f - z) * hh + z * x; }
                if (d.out_nchw) p.out[((long)n_img * d.Cout + co) * p.HoWo + (m - (long)n_img * p.HoWo)] = x;
                else p.out[m * d.ldo + d.ooff + co] = x;
            }
        }
    }
}

template <int TAPS, int COUT>
int launch_direct(const ConvP& p, int lpp_log2, hipStream_t s) {
    const int ppb = 256 >> lpp_log2;
    long blocks = (p.M + ppb - 1) / ppb;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((conv_direct_kernel<TAPS, COUT>), dim3((unsigned)blocks), dim3(256), 0, s, p, lpp_log2);
    return fgt_check_launch("conv_direct");
}


// ---- LDS-tiled variant for 3x3 / stride 1 / pad 1 layers whose input map is large (FGT decoder.final: 64 -> 3 at full
// resolution, RAFT flow head 256 -> 2).  A 256-thread block owns an 8 x 32 output tile; the 10 x 34 input halo tile is
// staged through LDS 16 channels at a time ([pixel][16 + 4 pad] floats: conflict-free float4 reads), every thread
// accumulates its pixel's COUT outputs with weights fetched through the scalar cache (uniform addresses).
// Input bytes are read from HBM/L2 once per tile (halo overhead 1.33x) instead of 9x.
// Round 6: rocprofv3 counted 2.05 x the algorithmic bytes for this kernel (round 5; 1.33 x is the halo).  Two causes, two changes: (i) neighbouring
// tiles sat on different XCDs, each fetching the shared halo into its own L2 -> XCD-contiguous tile order (below): RAFT's flow head 96.7 -> 78.2 us
// per launch, decoder.final 474 -> 447; (ii) a 16-channel chunk uses HALF of each 128-byte line per pass -> a 32-channel chunk variant
// (FGT_CONV_SMALL_C32=1): counter traffic 1.00 x the algorithmic bytes, but 49 KB of LDS per workgroup instead of 27 — 116 vs 78 us and 472 vs 447:
// the kernel is bound by its per-chunk barrier chain, not by the bytes, so the variant stays off (profiles/r06_run7_*).
constexpr int TH = 8, TW = 32;

// XH: the input map is an fp16 tensor (fgt_conv_desc.in_split = 3: ld / off in fp16 elements) — the f16 mode hands the decoder's last
// feature map (64 channels at full resolution: the largest activation of the path) over at 2 B per value; weights and arithmetic stay fp32.
template <int COUT, bool XH, int CCH>
__global__ void __launch_bounds__(256) conv3x3_tiled_kernel(const ConvP p) {
    constexpr int PLD = CCH + 4;
    __shared__ __attribute__((aligned(16))) float tile[(TH + 2) * (TW + 2) * PLD];
    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    // XCD-contiguous tile order (1-D grid, a multiple of 8 workgroups): workgroup i runs on XCD i % 8; XCD x takes the x-th eighth of the tiles in
    // (image, tile row, tile column) order, so the halo rows / columns two neighbouring tiles share are fetched into ONE L2 instead of two
    const int ntx = (d.W + TW - 1) / TW, nty = (d.H + TH - 1) / TH;
    const int lt = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (lt >= ntx * nty * d.N) return;
    const int n_img = lt / (ntx * nty), r_ = lt - n_img * (ntx * nty);
    const int x0 = (r_ % ntx) * TW, y0 = (r_ / ntx) * TH;
    const float* __restrict__ wgt = p.w;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    // The halo tile of the NEXT 16 channels is requested into registers before the FMAs of the current 16 run and stored to LDS behind
    // them (one chunk of global-load latency per tile instead of one per chunk: the kernel sat at 0.35 of the HBM roof).
    constexpr int NLD = ((TH + 2) * (TW + 2) * (CCH / 4) + 255) / 256;      // float4 per thread and chunk (6)
    float4 pre[NLD];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            const int idx = tid + l * 256;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < (TH + 2) * (TW + 2) * (CCH / 4)) {
                const int pix = idx / (CCH / 4), c4 = idx % (CCH / 4);
                const int py = pix / (TW + 2), px = pix - py * (TW + 2);
                const int gy = y0 + py - 1, gx = x0 + px - 1;
                if ((unsigned)gy < (unsigned)d.H && (unsigned)gx < (unsigned)d.W) {
                    const int ci = c0 + c4 * 4;
                    const float* src; int ld, ch;
                    if (ci < p.Cg0) { src = p.x0; ld = d.ld0; ch = d.off0 + ci; } else { src = p.x1; ld = d.ld1; ch = d.off1 + ci - p.Cg0; }
                    const long off = ((long)(n_img * d.H + gy) * d.W + gx) * ld + ch;
                    if constexpr (XH) {
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        const h4 hv = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(src) + off);
                        v = make_float4((float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]);
                    } else {
                        v = *reinterpret_cast<const float4*>(src + off);
                    }
                    if (d.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
            }
            pre[l] = v;
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < p.Cg; c0 += CCH) {
        __syncthreads();                                                    // the previous chunk's reads are done
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            const int idx = tid + l * 256;
            if (idx < (TH + 2) * (TW + 2) * (CCH / 4)) *reinterpret_cast<float4*>(tile + (idx / (CCH / 4)) * PLD + (idx % (CCH / 4)) * 4) = pre[l];
        }
        __syncthreads();
        if (c0 + CCH < p.Cg) fetch(c0 + CCH);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float* tp = tile + ((ty + t / 3) * (TW + 2) + tx + t % 3) * PLD;
#pragma unroll
            for (int c4 = 0; c4 < CCH / 4; ++c4) {
                const float4 v = *reinterpret_cast<const float4*>(tp + c4 * 4);
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float* w = wgt + (long)co * d.Kpad + t * p.Cg + c0 + c4 * 4;   // wave-uniform -> scalar loads
                    acc[co] = fmaf(v.x, w[0], fmaf(v.y, w[1], fmaf(v.z, w[2], fmaf(v.w, w[3], acc[co]))));
                }
            }
        }
    }
    const int oy = y0 + ty, ox = x0 + tx;
    if (oy < d.H && ox < d.W) {
        const long m = ((long)n_img * d.H + oy) * d.W + ox;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float cs = p.cscale ? p.cscale[co] : 1.f, cb = p.cbias ? p.cbias[co] : 0.f;
            float x = fgt_act(acc[co] * cs + cb, d.act, d.slope) * d.out_scale;
            if (d.epi == FGT_EPI_MUL) x *= p.aux1[m * d.ld_aux1 + co];
            else if (d.epi == FGT_EPI_ADD) x = fgt_act(x + p.aux1[m * d.ld_aux1 + co], d.act2, d.slope);
            else if (d.epi == FGT_EPI_GRU) { const float z = p.aux1[m * d.ld_aux1 + co], hh = p.aux2[m * d.ld_aux2 + co]; x = (1.f - z) * hh + z * x; }
            if (d.out_nchw) p.out[((long)n_img * d.Cout + co) * p.HoWo + (long)oy * d.W + ox] = x;
            else p.out[m * d.ldo + d.ooff + co] = x;
        }
    }
}

template <int COUT>
int launch_tiled(const ConvP& p, hipStream_t s) {
    dim3 grid((unsigned)(((long)cdiv(p.d.W, TW) * cdiv(p.d.H, TH) * p.d.N + 7) / 8 * 8));
    static const bool c32_env = [] { const char* e = getenv("FGT_CONV_SMALL_C32"); return e && e[0] == '1'; }();      // (measurement switch: see above)
    const bool c32 = c32_env && p.Cg % 32 == 0 && p.Cg0 % 32 == 0 && p.d.off0 % 4 == 0;
    if (p.d.in_split == 3) hipLaunchKernelGGL((conv3x3_tiled_kernel<COUT, true, 16>), grid, dim3(256), 0, s, p);      // (fp16 maps: 32 channels are 64 bytes — already half a line per pass; unchanged)
    else if (c32) hipLaunchKernelGGL((conv3x3_tiled_kernel<COUT, false, 32>), grid, dim3(256), 0, s, p);
    else if (p.Cg % 16 != 0) hipLaunchKernelGGL((conv3x3_tiled_kernel<COUT, false, 8>), grid, dim3(256), 0, s, p);   // (Cin = 24: LAFC's flow output conv, lafc.py:80 — it ran the one-pixel-per-lane-group kernel at 0.36 TB/s)
    else hipLaunchKernelGGL((conv3x3_tiled_kernel<COUT, false, 16>), grid, dim3(256), 0, s, p);
    return fgt_check_launch("conv3x3_tiled");
}

bool tiled_eligible(const ConvP& p) {
    const fgt_conv_desc& d = p.d;
    return d.kh == 3 && d.kw == 3 && d.sh == 1 && d.sw == 1 && d.dh == 1 && d.dw == 1 && d.ph == 1 && d.pw == 1 && !d.upsample &&
           d.pad_mode == 0 && p.Cg % 8 == 0 && (p.Cg % 16 == 0 || (p.Cg0 % 8 == 0 && d.in_split != 3)) && d.N <= 65535 && d.H >= TH && d.W >= TW;
}

}  // namespace

bool fgt_conv_direct_eligible(const ConvP& p) {
    const int taps = p.d.kh * p.d.kw;
    if (p.d.in_split == 3) return p.d.groups == 1 && p.Cout_g <= 4 && tiled_eligible(p);      // fp16 inputs: the LDS-tiled 3x3 kernel only
    return p.d.groups == 1 && p.Cout_g <= 4 && (taps == 9 || taps == 1) && p.Cg <= 256 && p.Cg >= 4;
}

int fgt_conv_direct(const ConvP& p, hipStream_t s) {
    int lpp_log2 = 0;
    while ((1 << lpp_log2) < (p.Cg >> 2)) ++lpp_log2;
    const int taps = p.d.kh * p.d.kw;
    const int co = p.Cout_g;
    if (tiled_eligible(p)) {
        switch (co) {
            case 1: return launch_tiled<1>(p, s);
            case 2: return launch_tiled<2>(p, s);
            case 3: return launch_tiled<3>(p, s);
            default: return launch_tiled<4>(p, s);
        }
    }
    if (taps == 9) {
        if (co <= 2) return launch_direct<9, 2>(p, lpp_log2, s);
        return launch_direct<9, 4>(p, lpp_log2, s);
    }
    if (co <= 2) return launch_direct<1, 2>(p, lpp_log2, s);
    return launch_direct<1, 4>(p, lpp_log2, s);
}
