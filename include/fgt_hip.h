/*
 * fgt_hip.h — C ABI of libfgt_hip.so, the MI355X (gfx950) hot path of hitachinsk/FGT.
 *
 * Every entry point takes raw device pointers (fp32 unless noted), plain ints and the HIP stream
 * (as void*) to enqueue on.  No entry point synchronises; the only memory the library owns is one 256-byte zero page per
 * device (target of out-of-image gathers), allocated by fgt_init(device) — call it once per device before the first launch
 * and outside stream capture (fgt_amd/_lib.py does so when it loads the library).
 * Return value: 0 on success, a negative FGT_E* code on a rejected argument or a failed launch
 * (never throws across the ABI).  `fgt_last_error()` returns a static description of the last
 * failure on the calling thread.
 *
 * Activations are channels-last: an "image" tensor is [N, H, W, C] fp32, a "token" tensor is
 * [rows, C].  Every tensor argument comes with a pixel/row stride `ld*` (in floats) and a channel
 * offset so that kernels can read and write slices of wider buffers without copies.
 *
 * The reference is pure PyTorch: it has no FFI for this path.  Each entry point below therefore
 * cites the reference ATen call site(s) it replaces (paths relative to the reference root);
 * INTEGRATION.md shows the ctypes binding used by the drop-in nn.Modules.
 */
#ifndef FGT_HIP_H
#define FGT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define FGT_OK 0
#define FGT_EINVAL (-1)  /* rejected argument (shape, alignment, unsupported combination) */
#define FGT_ELAUNCH (-2) /* hipLaunch / hipGetLastError reported a failure */

/* activation codes shared by all epilogues */
#define FGT_ACT_NONE 0
#define FGT_ACT_LRELU 1 /* LeakyReLU(slope) */
#define FGT_ACT_RELU 2
#define FGT_ACT_SIGMOID 3
#define FGT_ACT_TANH 4

/* epilogue combine modes of fgt_conv2d (applied after bias + activation) */
#define FGT_EPI_NONE 0
#define FGT_EPI_MUL 1     /* v *= aux1[m, n]                                         */
#define FGT_EPI_ADD 2     /* v += aux1[m, n]; then act2                              */
#define FGT_EPI_GRU 3     /* v = (1 - aux1[m,n]) * aux2[m,n] + aux1[m,n] * v  (z, h) */
/* ABI 7 — applied BEFORE the activation, on the accumulator (+ cscale / cbias): */
#define FGT_EPI_AFFINE 4  /* v = act(v * aux2[m,n] + aux1[m,n]): per-position scale and offset tables (fold()'s 1/count and summed biases) */
#define FGT_EPI_PS_ADD2 5 /* v = act(v + aux1[m,n] + aux2[pixel,c]): aux1 a table in the conv's own (row, column) layout, aux2 a map in the
                           * SUB-PIXEL output layout (desc.ps_r > 0: the encoder residual of Vec2Patch, model.py:280)               */

const char* fgt_last_error(void);
int fgt_abi_version(void);
/* Allocates (once, thread-safe) the zero page of `device` (-1: the current device).  Every launcher also calls it lazily for
 * its current device, so forgetting it is only illegal inside a stream capture (hipMalloc cannot be captured). */
int fgt_init(int device);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 *
 * out[n, oy, ox, ooff + g*Cout_g + co] = epi( act( cscale[co] * sum_{ky,kx,ci} x(...) * w + cbias[co] ) )      (cbias[m, co] with desc.ld_bias > 0)
 *
 * Input: one or two channels-last sources concatenated per group: group g reads channels
 *   [g*C0/groups, (g+1)*C0/groups) of x0 followed by [g*C1/groups, ...) of x1 (C1 = 0: single source).
 *   This is the reference Encoder's group-interleaved concat (FGT/models/model.py:57-66) and plain
 *   torch.cat for groups = 1 (LAFC/models/lafc.py:100-103, RAFT/update.py:52,97,126).
 * upsample = 1 applies nearest x2 to the input first (network_blocks_2d.py:46-60 VanillaDeconv).
 * pad_mode = 1 clamps coordinates (nn.ReplicationPad2d, FGT/models/model.py:207) instead of zeros.
 * in_relu = 1 applies ReLU to input values as they are gathered (ffn_base.py:40-45: conv2 = ReLU -> Linear).
 * A Linear / matmul is the kh = kw = 1 case with H = 1, W = rows.
 *
 * Replaces: F.conv2d / nn.Conv2d, nn.Conv3d with (1,k,k) or (k,1,1) kernels, nn.Linear and torch.matmul at
 *   FGT/models/model.py:32-51,206-216,96,179-186; transformer_base/attention_base.py:37-40;
 *   attention_flow.py:34-37,52-55; ffn_base.py:39-45; LAFC/models/lafc.py:23-80,111-115,131-139;
 *   RAFT/extractor.py:118-192; RAFT/update.py:6-136; RAFT/corr.py:52-60.
 * ------------------------------------------------------------------------------------------ */
typedef struct fgt_conv_desc {
    int N, H, W;            /* images, input height/width BEFORE the optional x2 upsample            */
    int C0, ld0, off0;      /* source 0: channels used, pixel stride, first channel                   */
    int C1, ld1, off1;      /* source 1 (C1 = 0: absent)                                              */
    int Cout, groups;
    int kh, kw, sh, sw, ph, pw, dh, dw;
    int upsample;           /* 0 | 1                                                                  */
    int pad_mode;           /* 0 zeros | 1 replicate                                                  */
    int in_relu;            /* 0 | 1                                                                  */
    int Ho, Wo;             /* output spatial size (caller computes; validated)                       */
    int ldo, ooff;          /* output pixel stride and first channel                                  */
    int out_nchw;           /* 1: write out[n, co, oy, ox] (contiguous NCHW, ldo/ooff ignored)         */
    int act;  float slope;  /* FGT_ACT_*                                                              */
    int epi;                /* FGT_EPI_*                                                              */
    int act2;               /* activation after FGT_EPI_ADD                                           */
    int ld_aux1, ld_aux2;   /* row strides of aux tensors (indexed [m*ld + g*Cout_g + co])            */
    float out_scale;        /* multiplies the value after act (before epi); 1.0f = off                */
    int Kpad, Npad;         /* packed-weight geometry: w is [groups, Npad, Kpad], k = (ky*kw+kx)*Cg+ci */
    int tile;               /* 0 = auto; otherwise a FGT_TILE_* override (tuning / tests)             */
    int precision;          /* FGT_PREC_FP32 (exact fp32 MFMA) | FGT_PREC_BF16X3 (hi/lo bf16 split, 3 MFMAs) | FGT_PREC_F16.
                             * With FGT_PREC_BF16X3 `w_packed` must be the PRE-SPLIT bf16 image of the packed weights:
                             * [2][groups][Npad][Kpad] bf16, plane 0 = hi = bf16_rne(w), plane 1 = lo = bf16_rne(w - hi)
                             * (same byte count as the fp32 image).
                             * FGT_PREC_F16 (in_split = 3 only): ONE v_mfma_f32_32x32x16_f16 per product on operands rounded once
                             * to fp16 (11 significant bits, fp32 accumulate); `w_packed` = [groups][Npad][Kpad] fp16 = f16_rne(w),
                             * Kpad a multiple of 64 (a K-step is 64 channels: one 128-byte line per row).                     */
    /* "Split" activation tensors (FGT_PREC_BF16X3 only): two bf16 planes with the layout of the fp32 tensor they stand for,
     * hi = bf16_rne(x) at the pointer and lo = bf16_rne(x - hi) `ps` ELEMENTS further (same bytes as fp32).  A producer
     * splits each value once (conv epilogue, fgt_layernorm, fgt_fold, fgt_attention, fgt_split); a consumer conv then moves
     * its im2col tiles global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4) and does no conversion at all.        */
    int in_split;           /* 0: x0/x1 are fp32 (split inside the kernel, per tile)
                             * 1: x0/x1 point to split tensors; ld/off in bf16 elements; needs Cin/groups, ld, off % 8 == 0,
                             *    in_relu == 0 (the producer applies it)
                             * 2: as 1 but hi/lo INTERLEAVED per 32 channels: channel c of a pixel lives at element
                             *    (c/32)*64 + c%32 (hi) and +32 (lo) of a row of 2*C elements, so the 32 hi and 32 lo values of
                             *    one K-step are ONE 128-byte line (planes: two half-used lines).  ld* = row stride in elements
                             *    (>= 2*C), off* = LOGICAL first channel; needs Cin/groups and off % 32 == 0; ps* unused.
                             * 3: (FGT_PREC_F16) x0/x1 point to fp16 tensors = f16_rne(x), ONE plane with the layout of the fp32 tensor
                             *    (half the bytes); ld/off in fp16 elements, Cin/groups, ld, off % 8 == 0, in_relu == 0; ps* unused.
                             *    Written by the same producers as the split format when their plane stride argument is -1.
                             *    Cout/groups <= 4 with tile = 0 (3x3 / stride 1 / pad 1 only): the fp32 VALU kernel reads the fp16 map;
                             *    `w_packed` is then the fp32 image [groups][Npad][Kpad] and the arithmetic fp32.                     */
    int out_split;          /* 0: fp32 `out` only | 1: split `out_s` only | 2: both (needs Cout/groups, ldo_s, ooff_s % 4 == 0,
                             *    out_nchw == 0)                                                                         */
    int ldo_s, ooff_s;      /* row stride / first channel of out_s (bf16 elements; out_split with pso == 32 writes the
                             * interleaved layout of in_split = 2: ldo_s >= 2*Cout, needs Cout/groups % 32 == 0;
                             * pso == -1 writes the single fp16 plane of in_split = 3, whatever `precision` computed it)   */
    int w_il;               /* 1: w_packed is [groups][Npad][Kpad/32][hi 32 | lo 32] (interleaved) instead of two planes; 2 (diagnostic builds): MFMA fragment order */
    int k_alg;              /* profiling only: kh*kw*Cin/groups BEFORE zero-padding of the input channels (flow 2 -> 4, RGB 3 -> 4),
                             * the K that fgt_prof_* credits as algorithmic work; 0 = use the padded K                        */
    long long ps0, ps1, pso;/* plane strides (bf16 elements) of x0, x1, out_s                                            */
    /* ---- ABI 7: nn.Fold after a per-token Linear as ONE stride-1 convolution over the token grid (ffn_base.py:53-77, model.py:102-110).
     * fold(kernel 2r+1, stride r, padding r) of Linear(x) is a transposed convolution; its sub-pixel form: output pixel (r*I + ry, r*J + rx),
     * channel c of the folded map is output column (ry, rx, c) of a 3x3 "same" convolution over the [N, H, W] token grid whose tap
     * (ky, kx) = (di + 1, dj + 1) carries row (c, r + ry - r*di, r + rx - r*dj) of the Linear's weight (zero where that kernel position does
     * not exist).  Column order: [ry = 0: (rx, c), r*ps_c columns | zero padding up to ps_g0 | ry = 1..r-1: (ry, rx, c)]; the columns with
     * ry >= 1 have no ky = 0 tap (ky_skip_n0).  The epilogue scatters: out[n, r*I + ry, r*J + rx, ooff + c] (pixel stride ldo / ldo_s),
     * dropping sub-pixels outside the ps_H x ps_W map (r*H >= ps_H, r*W >= ps_W). */
    int ps_r;               /* 0: off | r: sub-pixel factor (groups == 1, out_nchw == 0, ps_c % 4 == 0)                     */
    int ps_c;               /* channels per output pixel                                                                  */
    int ps_g0;              /* first column of the ry >= 1 block (>= r*ps_c; columns [r*ps_c, ps_g0) are dropped)           */
    int ps_H, ps_W;         /* size of the output map                                                                     */
    int ky_skip_n0;         /* > 0: the weights of output columns >= ky_skip_n0 are all zero for ky = 0; the tap-reusing kernels start the K
                             * walk of tiles that lie entirely in those columns at ky = 1 (other kernels multiply the zeros: same result up to
                             * the sign of zero)                                                                           */
    int aux_per_image;      /* 1: aux1 (and aux2 of FGT_EPI_AFFINE) are [Ho*Wo, Cout] tables shared by all N images (row = m mod Ho*Wo) */
    int n_alg;              /* profiling only: output columns credited as algorithmic work (0 = Cout/groups), see k_alg     */
    int ld_bias;            /* 0: `cbias` is a [Cout] vector (or NULL) | > 0: `cbias` is an [M, Cout] MAP with this row stride, added in front
                             * of the activation like the vector (RAFT's SepConvGRU: the convolution of the iteration-invariant context
                             * features `inp`, RAFT/update.py:45-58, raft.py:112-115, computed once per pair instead of once per iteration) */
    int tile_order;         /* 0: every XCD walks its M tiles with the N tiles of one M tile adjacent in time (they share the im2col rows in that
                             * XCD's L2) | 1: N-major — an XCD finishes all its M tiles for one N tile before the next: the workgroups resident together
                             * read the SAME weight rows (one N tile's K x 128 weights stay in the 4 MB L2) at the price of streaming the input once
                             * per N tile.  For layers whose whole weight matrix does not fit the L2 (the fold convolutions: 7 and 21 MB).  Results
                             * are identical (the order of tiles, not of any sum) */
    /* ---- ABI 8 */
    int dual_n0;            /* 0: off | n0 > 0 (with FGT_EPI_MUL, out_split = 2, groups = 1, split inputs, Cout = 2*n0, n0 % 64 == 0: a wavefront's columns lie in one head):
                             * TWO heads in one convolution.  Columns [0, n0): act(v) -> fp32 `out` channel ooff + n, no combine; columns [n0, 2*n0):
                             * act(v) * aux1[m, n - n0] -> split `out_s` channel ooff_s + n - n0.  RAFT's SepConvGRU: z = sigmoid(convz(hx)) and
                             * r * h = sigmoid(convr(hx)) * h read the same hx (RAFT/update.py:46-49, 53-56): one Cout = 256 launch per GRU half
                             * instead of two of 128, the im2col rows fetched once. */
    int ps_phase_pad;       /* ABI 9 (the field was `reserved8` = 0 in ABI 8).  0: off | cv > 0: "nearest x2 upsampling + 3x3 'same' convolution" (zero padding) as ONE
                             * 2x2 convolution over the LOW-RESOLUTION map with a sub-pixel output (FGT/models/utils/network_blocks_2d.py:46-60: the decoder's
                             * deconv blocks; LAFC/models/utils/network_blocks.py likewise).  Output pixel (2i + a, 2j + b) of the upsampled convolution sees only
                             * input rows {i-1, i} (a = 0) or {i, i+1} (a = 1) — likewise the columns —, so sub-pixel (a, b) is a 2x2 convolution whose taps carry
                             * SUMS of the 3x3 weights (a = 0: [w(-1)], [w(0) + w(+1)]; a = 1: [w(-1) + w(0)], [w(+1)]) and whose padding is (1 - a, 1 - b): 4
                             * multiply-adds per output value and input channel instead of 9.  Needs ps_r = 2, ps_g0 = 2*ps_c, Cout = 4*ps_c (columns ordered
                             * (a, b, c)), kh = kw = 2, ph = pw = 1, stride 1, no dilation, no `upsample`, zero padding, groups = 1, split inputs (in_split 1 | 2),
                             * Ho = H, Wo = W, ps_H = 2*H, ps_W = 2*W, and a tile whose N width divides ps_c (ps_c % 64 == 0): a workgroup's columns then lie in
                             * one sub-pixel and it subtracts (a, b) from (ph, pw).  The VALUE is cv, the number of real channels per sub-pixel (cv <= ps_c,
                             * cv % 4 == 0): the output map has cv channels per pixel and columns (a, b, c >= cv) — zero weight rows that pad a layer's channel
                             * count to the tile width — are dropped.  Served by csrc/conv_split.hip and csrc/conv_wide.hip.  With this mode
                             * FGT_EPI_MUL / FGT_EPI_ADD read aux1 as a map shaped like the OUTPUT ([N, ps_H, ps_W, cv]: row = output pixel, column = c). */
    long long gb_x0, gb_w, gb_o;
                            /* all 0: off | BATCHED GEMM (needs in_split = 2, w_il = 1, a "wide" tile code, 1 x 1, stride 1, one source, N = 1, fp32 output,
                             * no epilogue / bias / scale, Cout/groups % 8 == 0): group g multiplies ITS OWN rows and ITS OWN weights,
                             *   out[g*gb_o + m*ldo + ooff + n] = out_scale * sum_k x0[g*gb_x0 + m*ld0 + k] * w[g*gb_w + n*2*Kpad + k]     (m < H*W, n < Cout/groups)
                             * gb_x0 / gb_w in bf16 elements, gb_o in floats; C0 = groups * K.  The weight operand of a group is any [Cout/groups, 2*Kpad]
                             * interleaved split tensor — e.g. the OTHER frame's feature map as a conv epilogue wrote it: RAFT's all-pairs correlation
                             * corr[b] = fmap1[b] . fmap2[b]^T / sqrt(256) (RAFT/corr.py:52-60) for a whole pair batch in one launch, no weight packing. */
} fgt_conv_desc;

#define FGT_PREC_FP32 0
#define FGT_PREC_BF16X3 1
#define FGT_PREC_F16 2    /* fp16 operands (rounded once by the producer), one MFMA per product, fp32 accumulate: pre-split inputs only */

#define FGT_TILE_128x128 1
#define FGT_TILE_128x64 2
#define FGT_TILE_64x64 3
#define FGT_TILE_128x32 4
#define FGT_TILE_256x128 5
#define FGT_TILE_128x128x8 6   /* 128x128 tile on 8 wavefronts (64x32 each) */
#define FGT_TILE_256x128x16 7  /* 256x128 tile on 16 wavefronts */
#define FGT_TILE_256x64x8 8    /* 256x64 tile on 8 wavefronts (Cout = 64 layers) */
/* 9: retired (256x256 one-workgroup tiles: 8 wavefronts of 128x64 or 16 of 64x64 spill at their VGPR caps and measured slower) */
/* Codes 10-20, 32-35 (planes layout): schedule variants that were measured and NOT adopted.  They are rejected by the product library and exist
 * in diagnostic builds only (csrc/diag/conv_split_variants.hip, `fgt_amd.build.build(variant="diag")`); 17 / 18 + 100 are product tiles of the
 * wide kernel (csrc/conv_wide.hip). */
#define FGT_TILE_C4 40          /* csrc/conv_c4.hip: FGT_PREC_BF16X3 on an fp32 single-source input with C0 = 4 and a square 3 / 5 / 7 kernel (the first conv of the
                                * frame / flow encoders, of LAFC, of RAFT's motion encoder): input gathered straight into MFMA fragments, 128 x 64 tile */
#define FGT_TILE_256x128x8_S3 10  /* split inputs only: 256x128 on 8 wavefronts, 3-stage LDS-DMA ring, one workgroup per CU */
#define FGT_TILE_256x128x16_S3 11 /* split inputs only: 256x128 on 16 wavefronts, 3-stage ring */
#define FGT_TILE_128x128x8_S4 12  /* split inputs only: 128x128 on 8 wavefronts, 4-stage ring */
#define FGT_TILE_256x128x8_PP 13  /* split inputs only: 256x128, 3-stage ring, two ping-pong wavefront groups (R / M phases) */
#define FGT_TILE_128x128x8_PP 14  /* split inputs only: 128x128, 4-stage ring, ping-pong */
/* 15: retired (256x256 on 8 wavefronts of 128x64 with the interleaved schedule: 220 VGPRs, +0...5 % on N >= 512 GEMMs, slower epilogue) */
#define FGT_TILE_256x128x8_IL 16  /* split inputs only: 256x128 on 8 wavefronts of 64x64, interleaved schedule */
#define FGT_TILE_256x256_P8 17    /* DIAGNOSTIC BUILDS ONLY; split inputs only: 256x256 on 8 wavefronts of 128x64, 8-phase schedule: two staggered wavefront
                                   * groups (one loads / issues LDS-DMA while the other owns the matrix pipe), counted vmcnt, s_setprio */
#define FGT_TILE_256x128_P8 18    /* split inputs only: the same schedule on a 256x128 tile (8 wavefronts of 64x64) */
#define FGT_TILE_128x128_EA 26    /* split inputs only: tiles 1, 2, 3, 6, 7, 8 with EARLY STAGE RELEASE — an extra barrier after the fragment reads frees */
#define FGT_TILE_128x64_EA 27     /* the LDS stage for tile kt+2 one step early: two tiles in flight on two stages (the K loop is bound by the */
#define FGT_TILE_64x64_EA 28      /* global -> LDS latency, not by the matrix pipe).  Bit-identical results.                                */
#define FGT_TILE_128x128x8_EA 29
#define FGT_TILE_256x128x16_EA 30
#define FGT_TILE_256x64x8_EA 31
#define FGT_TILE_128x128x8_XY 35  /* split inputs only: tile 29 where only the upper half of the wavefronts (one per SIMD) issues the LDS-DMAs — all of
                                   * them — and the lower half goes from the stage-release barrier straight into its MFMAs.  Bit-identical, not faster. */
#define FGT_TILE_128x128x8_LW 32  /* split inputs only: early-release tiles with two LOADER wavefronts per workgroup (one owns the A tile, one the  */
#define FGT_TILE_128x128_LW 33    /* B tile: address arithmetic + LDS-DMA issue only) next to the 8 / 4 consumer wavefronts (fragment reads + MFMAs): */
#define FGT_TILE_128x64_LW 34     /* the DMA issue no longer sits in front of the MFMAs of the same wavefront.  Bit-identical results; measured slower. */
#define FGT_TILE_256x128_EA 38   /* FGT_PREC_F16 only: 256x128 on 8 wavefronts of 64x64 with early stage release */
#define FGT_TILE_WIDE 100        /* tile code + 100 = the same tile on the "wide" LDS image: a stage row is one full 128-byte line per pixel and
                                  * K-step, an LDS-DMA instruction copies 8 full cache lines instead of 16 half lines.  FGT_PREC_F16 (below) and, since
                                  * ABI 6, FGT_PREC_BF16X3 with INTERLEAVED inputs (in_split = 2, w_il = 1: csrc/conv_wide.hip; codes 1-8, 26-31, 38 and the
                                  * 8-phase tiles 17 / 18).  Bit-identical results. */
#define FGT_TILE_TAPS 200        /* tile code + 200 (codes 1, 2, 3, 4 = 128x64 on 8 wavefronts, 6) = the tap-reusing kernel (csrc/conv_taps.hip) for stride-1 "same"
                                  * convolutions with kw in {3, 5, 7} on split inputs: the im2col rows of a (ky, 32-channel chunk) are loaded once for all kx taps
                                  * (3x3: -31 % LDS-DMA instructions, the im2col stream from memory shrinks by kw).  It accumulates in the order (ky, chunk, kx):
                                  * NOT bit-identical to the other kernels, same error.  fgt_conv2d therefore routes by GEOMETRY: a layer this kernel serves runs
                                  * on it whenever tile = 0 (k x 1 layers only when H >= 16) or a +200 code; an explicit tile of another family selects that family.  Its tiles are bit-identical
                                  * to each other.  FGT_CONV_TAPS=0 (environment) turns the routing off.
                                  * Codes 200 + 26 / 5 / 17 ("128x128it", "256x128it", "256x256it"; round 4) = the same arithmetic with the step's LDS-DMA
                                  * requests and fragment reads interleaved into its MFMAs (csrc/conv_taps_il.hip: 128x128 on 4 wavefronts at two
                                  * workgroups per CU, 256x128 and 256x256 on 8; a wide LDS image for interleaved inputs): bit-identical to the other
                                  * tap tiles; they decline k x 1 and upsampling layers (and 256x256 kw != 3) with FGT_EINVAL. */
#define FGT_TILE_TAPS_BREG 300   /* DIAGNOSTIC BUILDS ONLY: the same kernel with weights in MFMA fragment order (w_il = 2) loaded straight into registers */
#define FGT_TILE_F16_WIDE 100    /* FGT_PREC_F16 only: tile code + 100 = the same tile on the "wide" LDS image (a stage row is the pixel's whole
                                  * 128-byte line of the 64-channel K-step; an LDS-DMA instruction copies 8 full cache lines instead of 16 half
                                  * lines).  Codes 1-8, 26-31 and 38.  Bit-identical results. */
#define FGT_TILE_256x256_P8N 19   /* 17 without s_setprio (A/B measurements) */
#define FGT_TILE_256x256_P8L 20   /* 17 with both wavefront groups in lock step (A/B measurements) */

/* 1 when fgt_conv2d would route this descriptor (with tile = 0) to the tap-reusing kernel: the autotuner of a caller then restricts its
 * candidates to the +200 tile codes, so that tuning never changes which accumulation order a layer gets. */
int fgt_conv_taps_route(const fgt_conv_desc* d);

int fgt_conv2d(const fgt_conv_desc* d, const void* x0, const void* x1, const float* w_packed,
               const float* cscale /* [Cout] or NULL */, const float* cbias /* [Cout] or NULL */,
               const float* aux1, const float* aux2, float* out /* NULL with out_split == 1 */,
               void* out_s /* split output or NULL */, void* stream);

/* fp32 [rows, C] (row stride ldx floats) -> split tensor (hi plane at out_s, lo plane `ps` bf16 elements further, row stride
 * ld_s); relu = 1 applies max(x, 0) first.  For activations whose producer is not one of the fused ones.  C % 4 == 0.
 * ps == -1: out_s is ONE fp16 plane = f16_rne(x) (fgt_conv_desc.in_split = 3).
 * C % 4 == 2 (planes layout only; strides and ps even): channel pairs, e.g. RAFT's 2-channel flow written into the last two channels of the
 * split GRU input buffer (RAFT/update.py:126 `cat([inp, motion_features])`). */
int fgt_split(const float* x, long rows, int C, int ldx, void* out_s, int ld_s, long long ps, int relu, void* stream);

/* Row LayerNorm over the concatenation [x0 | x1] (C1 = 0: single source), eps inside rsqrt.
 * Writes up to two outputs with different affine parameters from one pass over the row:
 *   outA = norm * gA + bA,  outB = norm * gB + bB  (outB = NULL: skipped).
 * Replaces nn.LayerNorm at FGT/models/model.py:126,128,147 and attention_flow.py:84-85,96 (q_norm/k_norm
 * share statistics for window tokens; v_norm).
 * psA / psB > 0: that output is written as a split tensor for the next GEMM (fgt_conv_desc.in_split): the pointer is the bf16 hi
 * plane, ld in bf16 elements, the lo plane ps elements further; 0 = fp32; -1 = one fp16 plane (fgt_conv_desc.in_split = 3);
 * 32 = the interleaved layout of fgt_conv_desc.in_split = 2 (rows of >= 2 * C elements, C % 32 == 0: one 128-byte line per 32 channels). */
int fgt_layernorm(const float* x0, int C0, int ld0, const float* x1, int C1, int ld1, long rows, float eps,
                  const float* gA, const float* bA, float* outA, int ldA,
                  const float* gB, const float* bB, float* outB, int ldB, long long psA, long long psB, void* stream);

/* Fused softmax(Q K^T / sqrt(d)) V with streaming (flash) softmax, d = 128, on fp32 MFMA.
 * mode 0 — temporal zone attention (attention_base.py:16-22 called from :61-69,93-101): tokens of all
 *   `t` frames inside one of group x group spatial zones attend to each other.  Q/K/V are
 *   [b*t, nh, nw, ld] maps (channel offsets qoff/koff/voff select the projection inside a fused QKV
 *   buffer); head hd reads channels [hd*128, hd*128+128).  O has the same geometry (ldo, no offset).
 * mode 1 — spatial window attention with shared global tokens (attention_flow.py:16-22 from :98-108):
 *   each ws x ws window of the padded nh x nw grid attends to its own ws*ws tokens followed by
 *   n_global tokens of the same frame (KG/VG: [bt, n_global, ldg]).  O is written to the CROPPED
 *   [bt, h, w, ldo] grid; rows outside (h, w) are dropped (attention_flow.py:109-110). */
typedef struct fgt_attn_desc {
    int mode;
    int b, t;               /* batch, frames per batch element (mode 1: only b*t is used)             */
    int h, w;               /* un-padded token grid (mode 1 crop)                                     */
    int nh, nw;             /* padded grid the Q/K/V maps live on                                     */
    int heads;              /* head dim is fixed at 128                                               */
    int group;              /* mode 0: zones per side                                                 */
    int ws, n_global;       /* mode 1                                                                 */
    int ldq, qoff, ldk, koff, ldv, voff, ldg_k, ldg_v, ldo;
    int precision;          /* FGT_PREC_FP32 | FGT_PREC_BF16X3 (Q, K, P, V split into hi/lo bf16, 3 MFMAs per product) |
                             * FGT_PREC_F16 (in_split = 2: Q, K, V arrive as fp16, P is rounded to fp16, one MFMA per product) */
    int out_split;          /* 1: O is a split tensor (see fgt_conv_desc): pointer to the bf16 hi plane, ldo in bf16 elements */
    long long pso;          /* plane stride of O in bf16 elements (out_split = 1); -1 = O is one fp16 plane (in_split = 2 only) */
    int in_split;           /* 1 (FGT_PREC_BF16X3 only): Q, K, V (and KG, VG) point to the bf16 hi planes of split tensors written by
                             * the projection GEMMs (fgt_conv_desc.out_split); ld* / *off in bf16 elements (multiples of 8), lo planes
                             * ps* elements further.  K / V tiles then travel global -> LDS by LDS-DMA and nothing is converted in the
                             * kernel (csrc/attention_split.hip); Q is stored unscaled and 1/sqrt(d) multiplies the fp32 scores.
                             * 2 (FGT_PREC_F16 only): the same with ONE fp16 plane per tensor (ps* unused).                      */
    int tq;                 /* mode 0: 0 = every frame queries; 0 < tq <= t: only the first tq frames of each batch element do (K / V still
                             * span all t frames) and O holds b * tq frames compactly — the clip scheduler's last temporal block, whose
                             * other frames nobody reads (tool/video_inpainting.py:727 consumes the neighbour frames only)            */
    long long psq, psk, psv, psg_k, psg_v;
    int compact;            /* mode 1: 1 = the Q / K / V maps hold only the h x w real tokens of every frame ([bt*h*w] rows); the tokens the
                             * reference zero-pads up to the window grid (attention_flow.py:120-124) all read row `pad_row` of the same
                             * maps — their projections are one constant row (LN of a zero row = its bias, then the Linear), so the
                             * padded rows never have to exist.  0 = the maps live on the padded nh x nw grid                     */
    int pad_row;
} fgt_attn_desc;

int fgt_attention(const fgt_attn_desc* d, const void* Q, const void* K, const void* V,
                  const void* KG, const void* VG, float* O, void* stream);

/* Depthwise kxk stride-k convolution over [x0 | x1] maps -> global tokens [bt, (nh/k)*(nw/k), C0+C1]
 * (attention_flow.py:44-48,80-81,87-91).  w is the reference layout [C,1,k,k], bias [C].
 * The maps hold the vh x vw real tokens of every frame ([bt*vh*vw] rows); the rest of the nh x nw window grid is the reference's
 * zero padding (attention_flow.py:120-124) and is never materialised (vh == nh, vw == nw: no padding). */
int fgt_dw_pool(const float* x0, int C0, int ld0, const float* x1, int C1, int ld1, int bt, int nh, int nw, int vh, int vw,
                int k, const float* w, const float* bias, float* out, int ldo, void* stream);

/* Depthwise 3x3, stride 1, pad 1, plus identity: out = dwconv(x) + x  (FGT/models/model.py:76-88).  bt <= 65535 (a grid dimension). */
int fgt_dw3x3_residual(const float* x, int bt, int h, int w, int C, const float* wgt, const float* bias,
                       float* out, void* stream);

/* Overlap-add "fold" of token patches back to a feature map, as a gather:
 *   out[f, y, x, c] = (sum over tokens (i,j) covering (y,x) of Y[f, i*tw+j, ((y-i*s+p)*k + (x-j*s+p))*C + c])
 *                     [/ count(y,x) if normalize] [+ res[f,y,x,c] if res]
 * Y's column order is tap-major (ky,kx,c) — the packed Linear weights are permuted accordingly.
 * Replaces F.fold / nn.Fold at ffn_base.py:56-75 (normalize=1: fold(x)/fold(ones)) and
 * FGT/models/model.py:102-110 (normalize=0) fused with the residual add at model.py:279. */
int fgt_fold(const float* Y, int ldy, int frames, int th, int tw, int C, int k, int s, int p, int Hf, int Wf,
             int normalize, const float* res, int ldres, float* out, int ldo,
             int relu /* max(.,0) last: the FFN's ReLU in front of its second Linear, ffn_base.py:40 */,
             long long ps_out /* > 0: `out` is the hi plane of a split tensor (bf16 elements), lo plane ps_out further (32: interleaved
                               * layout, C % 32 == 0, ldo >= 2 * C); -1: one fp16 plane */,
             int y_f16 /* 1: Y is the fp16 plane a GEMM wrote with pso = -1 (ldy in fp16 elements); the sums stay fp32 */, void* stream);

/* NCHW -> channels-last slice: dst[n, y, x, coff + c] = src[n, c, y, x] * scale + shift for c < C;
 * zero_to > C additionally zero-fills channels [C, zero_to).  (input packing: model.py:253-257) */
int fgt_nchw_to_nhwc(const float* src, int N, int C, int H, int W, float* dst, int ldd, int coff, int zero_to,
                     float scale, float shift, void* stream);
/* channels-last slice -> NCHW */
int fgt_nhwc_to_nchw(const float* src, int lds, int coff, int N, int C, int H, int W, float* dst, void* stream);

/* copy a token map [bt,h,w,C] (row stride lds) into [bt,nh,nw,C]: zero pad where the destination grid is larger
 * (attention_flow.py:66-68, attention_base.py:55-57), crop where it is smaller (attention_base.py:71-72).  bt <= 65535 (a grid dimension). */
int fgt_pad_tokens(const float* src, int lds, int bt, int h, int w, int C, int nh, int nw, float* dst, int ldd,
                   void* stream);

/* Backward bilinear warp, zeros padding (LAFC/models/utils/fbConsistencyCheck.py:8-26 image_warp:
 * grid_sample default align_corners=False on a linspace(-1,1) base grid + flow/((W-1)/2)).
 * img [B,H,W,C] channels-last (ld), flow [B,H,W,2] channels-last; align_corners = 1 gives
 * RAFT/utils/utils.py:57-71 bilinear_sampler semantics with absolute pixel coords in `flow`.  B <= 65535 (a grid dimension), H*W*C < 2^31. */
int fgt_warp(const float* img, int ldi, const float* flow, int B, int H, int W, int C, int align_corners,
             int absolute_coords, float* out, int ldo, void* stream);

/* Forward/backward consistency occlusion masks (fbConsistencyCheck.py:29-47).
 * flow_fw/flow_bw [B,H,W,2]; occ_fw/occ_bw [B,H,W] (1.0 = occluded).  B <= 65535. */
int fgt_fb_consistency(const float* flow_fw, const float* flow_bw, int B, int H, int W, float alpha1, float alpha2,
                       float* occ_fw, float* occ_bw, void* stream);

/* RAFT correlation pyramid: 2x2 average pooling of a [rows, H, W] volume (RAFT/corr.py:23-27). */
int fgt_avgpool2(const float* src, long rows, int H, int W, float* dst, void* stream);

/* RAFT correlation lookup (RAFT/corr.py:29-50): for every query pixel, 4 levels x (2r+1)^2 bilinear taps
 * (align_corners=True, zeros outside).  pyr[l] is [B*H1*W1, H1>>l, W1>>l]; coords [B,H1,W1,2] (x,y);
 * out [B,H1,W1, 4*(2r+1)^2] channels-last with the reference's channel order (level, dx-major "transposed" window). */
int fgt_corr_lookup(const float* const* pyr, int levels, int B, int H1, int W1, int radius, const float* coords,
                    float* out, int ldo, void* stream);
/* The same with an optional split output for the LDS-DMA conv kernels (bf16 hi plane at out_s, lo plane `ps` elements further, row stride
 * ld_s, channels [4*(2r+1)^2, nch_pad) written as zeros so that the consumer's K is a multiple of 32); out may be NULL when out_s is given.
 * Both forms hold the same values (hi + lo is the 16-bit split of the fp32 result). */
int fgt_corr_lookup_split(const float* const* pyr, int levels, int B, int H1, int W1, int radius, const float* coords,
                          float* out, int ldo, void* out_s, int ld_s, long ps, int nch_pad, void* stream);

/* RAFT convex upsampling (RAFT/raft.py:73-84): flow [B,H,W,2] (ld), mask [B,H,W,576] (ld) -> up [B,2,8H,8W] NCHW */
int fgt_convex_upsample(const float* flow, int ldf, const float* mask, int ldm, int B, int H, int W, float* out,
                        void* stream);

/* nn.InstanceNorm2d (affine=False, biased variance; RAFT/extractor.py:27-31) on a channels-last tensor:
 * fgt_instnorm_stats accumulates (sum, sum of squares) per (image, channel) into stats[N][C][2] (fp64 scratch,
 * zeroed by the call); fgt_instnorm_apply writes y = act2( act((x-mean)*rstd) + res ) (res = NULL: y = act(...)). */
int fgt_instnorm_stats(const float* x, int ld, int N, int HW, int C, double* stats, void* stream);
int fgt_instnorm_apply(const float* x, int ld, int N, int HW, int C, const double* stats, float eps, int act,
                       const float* res, int ldres, int act2, float* out, int ldo, void* stream);
/* ABI 8: the same with an optional split output for the LDS-DMA conv kernels behind it (out_s: bf16 hi at the pointer, lo `ps_s` elements further; ps_s = 32:
 * interleaved per 32 channels, C % 32 == 0, ldo_s >= 2 C; ps_s = -1: one fp16 plane); out may be NULL when out_s is given.  C % 4 == 0 runs the vectorised
 * kernels (per-channel mean / rstd once per workgroup); RAFT's fnet hands its activations to the next convolution pre-split this way. */
int fgt_instnorm_apply_split(const float* x, int ld, int N, int HW, int C, const double* stats, float eps, int act,
                             const float* res, int ldres, int act2, float* out, int ldo, void* out_s, int ldo_s, long long ps_s, void* stream);

/* Generic pointwise helper: out = act(a * sa + b * sb) over rows x C slices (b = NULL: single operand);
 * slope is the LeakyReLU slope when act = FGT_ACT_LRELU. */
int fgt_axpby(const float* a, int lda, float sa, const float* b, int ldb, float sb, long rows, int C, int act,
              float slope, float* out, int ldo, void* stream);

/* tool/video_inpainting.py:725-740 on device: comp = trunc_u8((x+1)/2*255)*m + trunc_u8(frame*255)*(1-m);
 * a frame's first visit stores, later visits average 0.5/0.5 (order dependent: call in ascending window order).
 * out_nchw [n,3,H,W] model output for the window's neighbour frames; ids[i] = clip frame index of row i and
 * first[i] = 1 when that frame has not been composed before (both device int32, decided by the host scheduler);
 * frames01 [N,3,H,W], masks [N,1,H,W], comp [N,H,W,3] fp32. */
int fgt_compose_blend(const float* out_nchw, const int* ids, const int* first, int n, const float* frames01,
                      const float* masks, int H, int W, float* comp, void* stream);

/* The same compose fed with already truncated values: filled_u8 [n,3,H,W] = astype(uint8)((x+1)/2*255), produced by
 * fgt_quantize_u8 on the rank that ran the window.  Window outputs travel between ranks in this form (1 byte per value instead
 * of 4): the truncation is the first thing tool/video_inpainting.py:731-733 does with the value, so the composite is unchanged. */
int fgt_compose_blend_u8(const unsigned char* filled_u8, const int* ids, const int* first, int n, const float* frames01,
                         const float* masks, int H, int W, float* comp, void* stream);
/* dst[i] = (unsigned char)(int)((x[i] + 1) / 2 * 255) for `count` values in (-1, 1) (count % 4 == 0). */
int fgt_quantize_u8(const float* x, long count, unsigned char* dst, void* stream);

/* Input packing of the clip-level FGT stage in one pass (tool/video_inpainting.py:697 `frames*2-1`, :719-721
 * `selected_frames * (1 - selected_masks)`, FGT/models/model.py:253-257 `cat(masked_frames, masks)` + NCHW -> channels-last):
 *   dst[i, y, x, 0:3] = (frames01[f, :, y, x] * 2 - 1) * (1 - masks[f, 0, y, x]),  dst[i, y, x, 3] = masks[f, 0, y, x],
 * f = ids[i] (device int32) or i when ids == NULL.  frames01 [N,3,H,W], masks [N,1,H,W], dst [n,H,W,ldd >= 4]. */
int fgt_pack_frames(const float* frames01, const float* masks, const int* ids, int n, int H, int W, float* dst, int ldd,
                    void* stream);

/* norm_flows (tool/video_inpainting.py:402-407; FGT/networks/network.py:80-84): every (frame, channel) map divided by its SIGNED
 * maximum over H*W.  flows [n_src, C, HW] -> out [n_out, C, HW]; output frame i reads source frame min(i, n_src - 1), which is
 * the tool's duplication of the last forward flow to the clip length (:705) when n_out = n_src + 1.  Bit-exact (max, IEEE divide). */
int fgt_norm_flows(const float* flows, int n_src, int n_out, int C, long HW, float* out, void* stream);

/* Row gather dst[i, 0:row_len] = src[ids[i], 0:row_len] (ids: device int32): a window's frames out of the per-frame feature
 * buffers — `tensor[:, neighbor_ids + ref_ids]` of tool/video_inpainting.py:718-722 applied to cached features.  row_len % 4 == 0, n <= 65535 rows per call (a grid dimension). */
int fgt_gather_rows(const float* src, long ld_src, const int* ids, int n, long row_len, float* dst, long ld_dst, void* stream);

/* Laplace ("diffusion") fill of the masked pixels of B scalar H x W maps (flow channels), all maps at once, by conjugate
 * gradients on the masked 5-point stencil:  n(p) x(p) - sum_{masked 4-neighbours} x(q) = sum_{unmasked in-image 4-neighbours} I(q),
 * n(p) = 4 / 3 / 2 in-image neighbours; unmasked pixels are copied.  Problem b uses mask b % n_masks (uint8, non-zero = hole).
 * `iters` CG iterations are always enqueued (no read-back); a map whose residual norm falls below tol * |r0| stops changing.
 * Results are bit-reproducible (ordered two-stage reductions, no atomics).  workspace: fgt_laplace_fill_workspace() bytes, 8-byte
 * aligned.  Replaces: rf.regionfill / diffusion(), tool/utils/region_fill.py:7-63, tool/video_inpainting.py:42-51 (scipy spsolve
 * per map on the CPU). */
long fgt_laplace_fill_workspace(int B, int H, int W);
int fgt_laplace_fill(const float* I, const unsigned char* mask, int B, int n_masks, int H, int W, float* out, void* workspace,
                     int iters, float tol, void* stream);

/* Flow-guided gradient propagation (tool/get_flowNN_gradient.py:11-534 with Nonlocal = False; call site tool/video_inpainting.py:
 * 623-633): for every hole pixel, chase the completed backward / forward flow through the clip to a known pixel (round-trip
 * consistency < consistency_thres, transitive chaining through holes), take the image gradient found there (bilinear) and fuse the
 * two candidates with weights exp(-round-trip error / alpha).  Frame-major channels-last layout:
 *   gx, gy [N,H,W,3] fp32 (gradients, 0 inside the dilated mask), mask [N,H,W] uint8 (non-zero = hole: the tool passes its DILATED
 *   gradient mask), flow_f / flow_b [N-1,H,W,2] fp32 (u, v): completed flows t -> t+1 / t+1 -> t.
 *   out_gx / out_gy [N,H,W,3]; mask_tofill [N,H,W] uint8 = hole pixels that found no flow neighbour.
 * tab: coordinate table of the bilinear sampler that stands for cv2.remap(INTER_LINEAR) (tool/utils/common_utils.py:164,250-251):
 * 32 = OpenCV-style 1/32-pixel quantisation (the specification in oracle/prop_oracle.py), 0 = plain float bilinear.
 * One call enqueues every launch of the clip (one per frame and sweep); workspace: fgt_flow_propagate_workspace() bytes, 8-byte aligned. */
long fgt_flow_propagate_workspace(int N, int H, int W);
int fgt_flow_propagate(const float* gx, const float* gy, const unsigned char* mask, const float* flow_f, const float* flow_b,
                       int N, int H, int W, double consistency_thres, double alpha, int tab, float* out_gx, float* out_gy,
                       unsigned char* mask_tofill, void* workspace, void* stream);

/* Poisson blending of propagated gradients into the frames (tool/utils/Poisson_blend_img.py:19-244; call site
 * tool/video_inpainting.py:644-682), all frames and colour channels of a clip in one call: per hole pixel up to four equations
 * (x_p - x_q = gradient, or x_p = target_q + gradient next to a known pixel) for the neighbours whose gradient lies outside
 * `gmask`; the least-squares solution (the reference: scipy LSQR per frame and channel) by batched conjugate gradients on the
 * normal equations, `iters` iterations always enqueued, a problem freezes at residual <= tol * |r0|.
 *   target [N,H,W,3] fp32 (0..1), gx / gy [N,H,W,3] (gx[y][x] = I[y][x+1] - I[y][x]; last column / row ignored),
 *   hole / gmask [N,H,W] uint8 (non-zero = hole / gradient unknown).
 *   blend [N,H,W,3] = solution inside the hole, target outside; unfilled [N,H,W] uint8 = UnfilledMask (:143-168): hole pixels the
 *   two raster sweeps cannot reach through valid gradients (the tool repaints them).
 * workspace: fgt_poisson_blend_workspace() bytes, 8-byte aligned.  Bit-reproducible (ordered reductions). */
long fgt_poisson_blend_workspace(int N, int H, int W);
int fgt_poisson_blend(const float* target, const float* gx, const float* gy, const unsigned char* hole, const unsigned char* gmask,
                      int N, int H, int W, int iters, float tol, float* blend, unsigned char* unfilled, void* workspace, void* stream);

/* ---- the two solves above with ONE WORKGROUP PER PROBLEM, all iterations inside one launch (csrc/solve_onchip.hip, ABI 6) ----
 * The problems are small (a hole's bounding box: 17-33 k cells) and many (158-240 per clip: one per CU): the search direction lives in LDS
 * as a dense image of the bounding box, x / r / A p in registers, the dot products are fixed-order block reductions (bit-reproducible), a
 * problem leaves its loop when |r| <= tol |r0| (`iters` is the cap).  Same iteration as fgt_laplace_fill / fgt_poisson_blend.
 * fgt_mask_bbox: bbox[m] = (y0, x0, y1, x1) inclusive of the non-zero pixels of mask m (device int32 [n][4]; empty mask: y1 < y0).
 * max_rows / max_cols: host-known upper bounds of the boxes' sizes (the Python wrapper reads `bbox` back once); FGT_EINVAL when such a box
 * does not fit one workgroup (<= 6 144 four-cell strips ~ 24 k cells, (rows + 2) * (strip columns + 8) floats of LDS within 160 KB) or
 * W % 4 != 0 — use the multi-launch entry points then.  status[problem]: 2 * iterations used; 1 = the DEVICE found the box larger than the bounds promised (its hole is
 * filled with NaN, nothing is written out of bounds). */
int fgt_mask_bbox(const unsigned char* mask, int n, int H, int W, int* bbox, void* stream);
int fgt_laplace_fill_onchip(const float* I, const unsigned char* mask, const int* bbox, int B, int n_masks, int H, int W, float* out,
                            int max_rows, int max_cols, int iters, float tol, int* status, void* stream);
int fgt_poisson_blend_onchip(const float* target, const float* gx, const float* gy, const unsigned char* hole, const unsigned char* gmask,
                             const int* bbox /* of `hole`, [N][4] */, int N, int H, int W, int max_rows, int max_cols, int iters, float tol,
                             float* blend, unsigned char* unfilled, int* status /* [3N] */, void* workspace /* fgt_poisson_blend_workspace */, void* stream);

/* ---- per-kernel timing with HIP events on the launch stream (bench.py's roofline blocks) ----
 * fgt_prof_enable(1) makes every fgt_conv2d (MFMA kernels) and fgt_attention launch record an event pair and its ALGORITHMIC
 * flops: conv/GEMM 2*M*Cout_g*K*groups (K before channel padding, fgt_conv_desc.k_alg); attention 4*n_q*n_k*128 per
 * (problem, head) — the two contractions of attention_base.py:16-22 as torch's flop counter counts them.
 * Each record also carries the launch's unique-byte floor (every input / weight / output byte once), so that a kernel can be
 * placed on the HBM roofline as well.  fgt_prof_collect_kind synchronises the events of one kind and returns (and clears) its
 * totals; kind -1 = all. */
#define FGT_PROF_CONV 0
#define FGT_PROF_ATTN_TEMPORAL 1
#define FGT_PROF_ATTN_SPATIAL 2
/* HBM-bound kernels (ABI 6): records carry 0 flops and the launch's ALGORITHMIC bytes — every input / output byte once, the
 * SURVEY.md §8(d) formulas — so that bytes / event time is the figure to hold against the 8 TB/s HBM roof:
 *   LAYERNORM    rows * (C_in * 4 + sum over outputs of C * {4: fp32 or hi + lo, 2: fp16})
 *   FOLD         token matrix once (frames*th*tw*k*k*C * {4 | 2}) + output map (+ residual map)
 *   CONV_SMALL   Cout <= 4 VALU convs (decoder.final 64 -> 3, LAFC 24 -> 2, RAFT flow head): input map + weights + output
 *   DW_POOL      real input tokens * C * 4 + weights + output tokens * C * 4
 *   WARP         (2C + 2) * B*H*W * 4   (image_warp: image in, flow in, image out)
 *   CORR_LOOKUP  per query pixel: levels * (2r+2)^2 * 4 gathered + 8 (coords) + levels * (2r+1)^2 * 4 written
 *   POINTWISE    everything else that is one pass over its data (row gathers, fp32 -> split, pad / crop, 3x3 position embedding,
 *                axpby, layout packing): bytes in + bytes out */
#define FGT_PROF_LAYERNORM 3
#define FGT_PROF_FOLD 4
#define FGT_PROF_CONV_SMALL 5
#define FGT_PROF_DW_POOL 6
#define FGT_PROF_WARP 7
#define FGT_PROF_CORR_LOOKUP 8
#define FGT_PROF_POINTWISE 9
void fgt_prof_enable(int on);                 /* every kind */
void fgt_prof_enable_kinds(unsigned mask);    /* bit k = kind k: an event pair serialises neighbouring launches a little, so bench.py times its
                                               * steps with the three MFMA kinds only and measures the HBM kinds on one extra step */
int fgt_prof_collect_kind(int kind, double* total_ms, double* total_flops, double* total_bytes, long* launches);
int fgt_prof_collect(double* total_ms, double* total_flops, long* launches); /* = kind FGT_PROF_CONV */

/* ---- sustained matrix-core rate of this chip (bench.py: `roofline.sustained`) ----
 * 256 workgroups x 8 wavefronts of back-to-back independent MFMAs (f32 = 0: v_mfma_f32_32x32x16_bf16, 1: v_mfma_f32_32x32x2_f32) on
 * random operands held in registers; returns the rate of the second of two launches and the shader clock the workgroups measured
 * (d s_memtime / d s_memrealtime).  The nominal peaks assume 2.4 GHz; under load the chip clocks to its power budget.
 * Synchronises the stream; workspace >= fgt_mfma_probe_workspace() bytes of device memory. */
long fgt_mfma_probe_workspace(void);
int fgt_mfma_probe(int f32, int iters, void* workspace, double* tflops, double* ghz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FGT_HIP_H */
