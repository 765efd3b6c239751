// Launch parameters shared by the conv kernels (implicit GEMM and direct).
#pragma once
#include "common.h"

// Exact n / d for 0 <= n < 2^31 with one 32 x 32 -> 64-bit multiply: mul = ceil(2^sh / d), sh = 31 + ceil(log2 d)  (host: fgt_fastdiv_make)
struct FgtFastDiv {
    unsigned mul;
    int sh;
};
inline FgtFastDiv fgt_fastdiv_make(unsigned d) {
    int s = 0;
    while ((1ull << s) < d) ++s;
    const int sh = 31 + s;
    return FgtFastDiv{(unsigned)(((1ull << sh) + d - 1) / d), sh};
}

struct ConvP {
    fgt_conv_desc d;
    const float *x0, *x1, *w, *cscale, *cbias, *aux1, *aux2;
    float* out;
    __bf16* out_s;            // split output (desc.out_split), plane stride pso
    long pso, ps0, ps1;       // plane strides in bf16 elements
    int M, HoWo, Cg0, Cg1, Cg, K, Cout_g, Hin, Win, nk;
    const float* zero_page;   // 256 zero bytes in device memory: target of out-of-range gathers
    int mtiles, ntiles, mchunk, xcd_swizzle;   // tile grid and XCD-aware ordering (set by the launcher)
    int pipe;   // bf16x3: software-pipelined K loop (FGT_CONV_PIPE=0 selects the plain double-buffered loop)
    int tr_li = 0;  // conv_taps.hip, k x 1 convolutions: != 0 (= H) when the tile's rows walk the image in (n, x, y) order; the epilogue maps them back
    FgtFastDiv div_howo, div_wo;   // rows -> (image, y, x) in the epilogue (sub-pixel output / per-image aux tables)
    int nt_store;   // epilogue stores non-temporal (default; FGT_CONV_NT=0 for A/B measurements): the output is written once and read by the NEXT
                    // kernel — keeping it out of the XCD's L2 leaves the cache to the A / B tiles this kernel re-reads.  Measured per layer
                    // (profiles/r03_run2_split_sweep_nt_stores.txt): K = 512 GEMMs +5...10 %, 3x3 layers +2 %, none slower
};

// runtime.hip: one 256-byte zero-filled device allocation, created on first use (the only memory the library owns)
const float* fgt_zero_page();

// conv_split.hip: bf16x3 with pre-split inputs moved global -> LDS by LDS-DMA
int fgt_conv_split_launch(int tile, const ConvP& p, hipStream_t s);

// conv_wide.hip: bf16x3 on interleaved pre-split inputs (in_split = 2, w_il = 1) with full-line LDS-DMA pieces; `tile` = tile code - 100
int fgt_conv_wide_launch(int tile, const ConvP& p, hipStream_t s);

// conv_taps.hip: bf16x3 for stride-1 "same" convs with kw in {3, 5, 7} on split inputs: the A rows of a (ky, chunk) stay in LDS for all kx taps.
// Selected by GEOMETRY (fgt_conv_taps_eligible), never by tuning: its accumulation order (ky, chunk, kx) differs from the other kernels'.
bool fgt_conv_taps_eligible(const ConvP& p);     // the kernel can run the layer (explicit +200 tiles)
bool fgt_conv_taps_preferred(const ConvP& p);    // ... and tile = 0 routes the layer to it
int fgt_conv_taps_launch(int tile, const ConvP& p, hipStream_t s);
// conv_taps_il.hip: the same arithmetic with the LDS-DMA requests of a step interleaved into its MFMAs and asm-pipelined fragment reads
// (tile codes 200 + FGT_TILE_128x128_EA / FGT_TILE_256x128 / FGT_TILE_256x256_P8 = "128x128it" / "256x128it" / "256x256it")
int fgt_conv_taps_il_launch(int bm, int bn, const ConvP& p, hipStream_t s);
// diag/conv_taps_breg.hip (diagnostic builds only): the same with the weight fragments loaded straight into registers (w_il = 2, tile code - 300)
int fgt_conv_taps_breg_launch(int tile, const ConvP& p, hipStream_t s);

// conv_c4.hip: bf16x3 for the 4-channel-input layers (fp32 input gathered straight into MFMA fragments, weights resident in the LDS): tile code
// FGT_TILE_C4, an autotuner candidate (bit-identical to the register-staged kernel's tiles)
bool fgt_conv_c4_eligible(const ConvP& p);
int fgt_conv_c4_launch(const ConvP& p, hipStream_t s);

// conv_f16.hip: FGT_PREC_F16 — fp16 inputs (one plane) through LDS-DMA, one MFMA per product
int fgt_conv_f16_launch(int tile, const ConvP& p, hipStream_t s);

// conv_direct.hip
bool fgt_conv_direct_eligible(const ConvP& p);
int fgt_conv_direct(const ConvP& p, hipStream_t s);
