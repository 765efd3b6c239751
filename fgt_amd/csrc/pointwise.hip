// Bandwidth-bound kernels of the FGT path: LayerNorm, depthwise convs, fold (overlap-add gather),
// layout packing, zero padding, axpby and the tool's compose/blend step.  All are HBM-bound:
// float4 accesses along the channel dimension of channels-last tensors, one pass over the data.
#include <stdlib.h>
#include "common.h"
#include "conv_params.h"      // FgtFastDiv

namespace {

// Round 6: index arithmetic.  These kernels decoded a 64-bit linear work-item index with `%` and `/` by kernel arguments — four to six 64-bit integer
// divisions per item, ~100 instructions each on a GPU without a divide unit: the "bandwidth" kernels were bound by their index arithmetic
// (the RAFT lookup lost half of its time this way, NOTEBOOK §13.8).  Now the outer dimension is a grid dimension (or a 32-bit index where the item
// count allows), and the inner decode is a multiply-shift by a host-made constant (FgtFastDiv, exact for n < 2^31).
__device__ __forceinline__ int fdiv(int n, const FgtFastDiv f) { return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.sh); }


__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// fp32 store, or (ps > 0) the split format: `out` then points at the hi plane of a bf16 tensor, offsets / ps in bf16 elements;
// ps == -1: `out` is one fp16 plane (offsets in fp16 elements)
// ps == 32: the INTERLEAVED split layout (fgt_conv_desc.in_split = 2): channel c of a row lives at element (c / 32) * 64 + c % 32 (hi) and 32
// further (lo) — `off` is then row * ld + c with ld the row stride in elements (>= 2 * C) and c the LOGICAL channel, passed separately
template <bool NT = false>
__device__ __forceinline__ void store_f32_or_split(float* out, long off, long ps, const float4 v, int c = 0) {
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
    if (ps < 0) {
        const uint2 h = fgt_half4(v);
        if constexpr (NT) __builtin_nontemporal_store(nt_u2{h.x, h.y}, reinterpret_cast<nt_u2*>(reinterpret_cast<__bf16*>(out) + off));
        else *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(out) + off) = h;
    } else if (ps > 0) {
        uint2 hi, lo;
        fgt_split4(v, hi, lo);
        __bf16* o = reinterpret_cast<__bf16*>(out) + (ps == 32 ? off - c + ((c >> 5) << 6) + (c & 31) : off);
        if constexpr (NT) {
            __builtin_nontemporal_store(nt_u2{hi.x, hi.y}, reinterpret_cast<nt_u2*>(o));
            __builtin_nontemporal_store(nt_u2{lo.x, lo.y}, reinterpret_cast<nt_u2*>(o + ps));
        } else {
            *reinterpret_cast<uint2*>(o) = hi;
            *reinterpret_cast<uint2*>(o + ps) = lo;
        }
    } else {
        if constexpr (NT) __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4*>(out + off));
        else *reinterpret_cast<float4*>(out + off) = v;
    }
}

// ------------------------------------------------------------------ LayerNorm: one wavefront per LN_R rows
// (LN_R = 2 rows per wavefront, their loads issued together: with one row per wavefront the kernel sat at 0.57 of the HBM roof — a full
//  chip of one-row wavefronts holds 16 MB in flight, about what 8 TB/s x 2 us needs; per-row arithmetic is unchanged)
constexpr int LN_MAXV = 4;  // float4 per lane -> C <= 1024
constexpr int LN_R = 2;
template <bool NT>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* x0, int C0, int ld0, const float* x1, int C1, int ld1,
                                                        long rows, float eps, const float* gA, const float* bA, float* outA,
                                                        int ldA, const float* gB, const float* bB, float* outB, int ldB,
                                                        long psA, long psB) {
    const int lane = threadIdx.x & 63;
    const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_R;
    if (row0 >= rows) return;
    const int C = C0 + C1, nv = C >> 2;
    float4 v[LN_R][LN_MAXV];
    float s[LN_R];
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        const long row = row0 + r < rows ? row0 + r : row0;          // (a wavefront's last row may not exist: it re-reads row0, nothing is stored)
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) {
                typedef float nt_f4 __attribute__((ext_vector_type(4)));
                const float* src = c < C0 ? x0 + row * ld0 + c : x1 + row * ld1 + (c - C0);
                if constexpr (NT) {
                    const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(src));
                    v[r][i] = make_float4(t.x, t.y, t.z, t.w);
                } else {
                    v[r][i] = *reinterpret_cast<const float4*>(src);
                }
            } else {
                v[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < LN_R; ++r) {
        if (row0 + r >= rows) break;
        const long row = row0 + r;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i)
            if ((lane + 64 * i) * 4 < C) s[r] += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
        const float mean = wave_sum(s[r]) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            if ((lane + 64 * i) < nv) {
                const float a = v[r][i].x - mean, b = v[r][i].y - mean, c = v[r][i].z - mean, e = v[r][i].w - mean;
                ss += (a * a + b * b) + (c * c + e * e);
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < C) {
                const float4 n = make_float4((v[r][i].x - mean) * rstd, (v[r][i].y - mean) * rstd, (v[r][i].z - mean) * rstd,
                                             (v[r][i].w - mean) * rstd);
                const float4 g = *reinterpret_cast<const float4*>(gA + c), b = *reinterpret_cast<const float4*>(bA + c);
                store_f32_or_split<NT>(outA, row * ldA + c, psA, make_float4(n.x * g.x + b.x, n.y * g.y + b.y, n.z * g.z + b.z, n.w * g.w + b.w), c);
                if (outB) {
                    const float4 g2 = *reinterpret_cast<const float4*>(gB + c), b2 = *reinterpret_cast<const float4*>(bB + c);
                    store_f32_or_split<NT>(outB, row * ldB + c, psB, make_float4(n.x * g2.x + b2.x, n.y * g2.y + b2.y, n.z * g2.z + b2.z, n.w * g2.w + b2.w), c);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ depthwise k x k stride k ("global tokens")
__global__ void __launch_bounds__(256) dw_pool_kernel(const float* x0, int C0, int ld0, const float* x1, int C1, int ld1,
                                                      int bt, int nh, int nw, int vh, int vw, int k, const float* w, const float* bias,
                                                      float* out, int ldo) {
    // (vh, vw) < (nh, nw): the maps hold only the vh x vw real tokens of a frame ([bt*vh*vw] rows); the rest of the nh x nw grid is the
    // reference's zero padding (attention_flow.py:120-124) and contributes acc + 0 * w == acc
    const int C = C0 + C1, c4n = C >> 2;
    const int gh = nh / k, gw = nw / k;
    const long total = (long)bt * gh * gw * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        long tok = idx / c4n;
        const int gx = (int)(tok % gw); long r = tok / gw;
        const int gy = (int)(r % gh); const int f = (int)(r / gh);
        float acc[4] = {bias[c], bias[c + 1], bias[c + 2], bias[c + 3]};
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) {
                const int yy = gy * k + a, xx = gx * k + b;
                if (yy >= vh || xx >= vw) continue;
                const long pix = ((long)f * vh + yy) * vw + xx;
                const float4 v = c < C0 ? *reinterpret_cast<const float4*>(x0 + pix * ld0 + c)
                                        : *reinterpret_cast<const float4*>(x1 + pix * ld1 + (c - C0));
                const int wi = a * k + b, kk = k * k;
                acc[0] += v.x * w[(c + 0) * kk + wi];
                acc[1] += v.y * w[(c + 1) * kk + wi];
                acc[2] += v.z * w[(c + 2) * kk + wi];
                acc[3] += v.w * w[(c + 3) * kk + wi];
            }
        *reinterpret_cast<float4*>(out + tok * ldo + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ------------------------------------------------------------------ depthwise 3x3 + identity (AddPosEmb)
__global__ void __launch_bounds__(256) dw3x3_res_kernel(const float* x, int bt, int h, int w, int C, const float* wgt,
                                                        const float* bias, float* out, FgtFastDiv dc4, FgtFastDiv dw_) {
    const int c4n = C >> 2, per_f = h * w * c4n;
    const int f = blockIdx.y;                                     // grid: (items of one frame / 256, frames)
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per_f; e += gridDim.x * 256) {
        const int pl = fdiv(e, dc4), c = (e - pl * c4n) * 4;
        const int py = fdiv(pl, dw_), px = pl - py * w;
        const long pix = (long)f * h * w + pl;
        // the 9 taps unrolled: clamped (always in-range) loads issued back to back, out-of-image taps get a zero WEIGHT
        // (acc + 0 * v == acc exactly), the 36 weights of the 4 channels are one contiguous run read as 9 float4
        float4 v[9];
        float m[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int yy = py + a - 1, xx = px + b - 1;
                m[a * 3 + b] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? 1.f : 0.f;
                const int yc = min(max(yy, 0), h - 1), xc = min(max(xx, 0), w - 1);
                v[a * 3 + b] = *reinterpret_cast<const float4*>(x + (((long)f * h + yc) * w + xc) * C + c);
            }
        float wv[36];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(wgt + c * 9 + q * 4);
            wv[q * 4 + 0] = t.x; wv[q * 4 + 1] = t.y; wv[q * 4 + 2] = t.z; wv[q * 4 + 3] = t.w;
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            acc[0] += v[t].x * (wv[0 * 9 + t] * m[t]);
            acc[1] += v[t].y * (wv[1 * 9 + t] * m[t]);
            acc[2] += v[t].z * (wv[2 * 9 + t] * m[t]);
            acc[3] += v[t].w * (wv[3 * 9 + t] * m[t]);
        }
        const float4 ctr = *reinterpret_cast<const float4*>(x + pix * C + c);
        *reinterpret_cast<float4*>(out + pix * C + c) =
            make_float4((acc[0] + bias[c]) + ctr.x, (acc[1] + bias[c + 1]) + ctr.y, (acc[2] + bias[c + 2]) + ctr.z,
                        (acc[3] + bias[c + 3]) + ctr.w);
    }
}

// k = 4 (the shipped g_downSize): the 16 taps unrolled, the tap loads issued back to back, weights as float4 rows (the generic kernel
// above walks the taps with 4 scalar weight loads each and ran at ~1 TB/s).  Same accumulation order: bit-identical results.
__global__ void __launch_bounds__(256) dw_pool4_kernel(const float* x0, int C0, int ld0, const float* x1, int C1, int ld1,
                                                       int bt, int nh, int nw, int vh, int vw, const float* w, const float* bias,
                                                       float* out, int ldo, FgtFastDiv dc4, FgtFastDiv dgw) {
    const int C = C0 + C1, c4n = C >> 2;
    const int gh = nh / 4, gw = nw / 4, per_f = gh * gw * c4n;
    const int f = blockIdx.y;                                     // grid: (items of one frame / 256, frames)
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per_f; e += gridDim.x * 256) {
        const int tl = fdiv(e, dc4), c = (e - tl * c4n) * 4;
        const int gy = fdiv(tl, dgw), gx = tl - gy * gw;
        const long tok = (long)f * gh * gw + tl;
        const bool s0 = c < C0;
        const float* src = s0 ? x0 + c : x1 + (c - C0);               // select on the address, loads stay straight-line
        const long ld = s0 ? ld0 : ld1;
        const long pix0 = ((long)f * vh + gy * 4) * vw + gx * 4;
        float4 v[16];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const bool in = gy * 4 + a < vh && gx * 4 + b < vw;         // past the real grid: the reference's zero padding
                v[a * 4 + b] = in ? *reinterpret_cast<const float4*>(src + (pix0 + (long)a * vw + b) * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        float wv[4][16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4*>(w + (c + u) * 16 + q * 4);
                wv[u][q * 4 + 0] = t.x; wv[u][q * 4 + 1] = t.y; wv[u][q * 4 + 2] = t.z; wv[u][q * 4 + 3] = t.w;
            }
        const float4 bv = *reinterpret_cast<const float4*>(bias + c);
        float acc[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            acc[0] += v[t].x * wv[0][t];
            acc[1] += v[t].y * wv[1][t];
            acc[2] += v[t].z * wv[2][t];
            acc[3] += v[t].w * wv[3][t];
        }
        *reinterpret_cast<float4*>(out + tok * ldo + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// ------------------------------------------------------------------ fold as a gather
// 4 consecutive channels of a token row: fp32, or (YH) the fp16 plane a GEMM wrote with pso = -1 (half the bytes of the largest
// tensor of a transformer block; the values are summed in fp32 either way)
template <bool YH> __device__ __forceinline__ float4 fold_load4(const float* Y, long off) {
    if constexpr (YH) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 v = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(Y) + off);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    } else {
        return *reinterpret_cast<const float4*>(Y + off);
    }
}

template <bool YH>
__global__ void __launch_bounds__(256) fold_kernel(const float* Y, int ldy, int frames, int th, int tw, int C, int k, int s,
                                                   int p, int Hf, int Wf, int normalize, const float* res, int ldres,
                                                   float* out, int ldo, int relu, long ps_out) {
    const int c4n = C >> 2;
    const long total = (long)frames * Hf * Wf * c4n;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % c4n) * 4;
        long pix = idx / c4n;
        const int x = (int)(pix % Wf); long r = pix / Wf;
        const int y = (int)(r % Hf); const int f = (int)(r / Hf);
        // tokens (i, j) with i*s - p <= y <= i*s - p + k - 1
        const int i_lo = max(0, (y + p - k + 1 + s - 1) / s), i_hi = min(th - 1, (y + p) / s);
        const int j_lo = max(0, (x + p - k + 1 + s - 1) / s), j_hi = min(tw - 1, (x + p) / s);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int cnt = 0;
        if (i_hi >= i_lo && j_hi >= j_lo && i_hi - i_lo < 3 && j_hi - j_lo < 3) {
            // k <= 3s (the shipped 7 / 3): at most 3 x 3 covering tokens.  Straight-line: the nine loads (clamped to a covering token, so
            // always in range) are issued back to back, tokens past the range add 0 (acc + 0 == acc) — same sums in the same order as the loop
            float4 v[9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const int i = min(i_lo + a, i_hi), j = min(j_lo + b, j_hi);
                    const int ky = y + p - i * s, kx = x + p - j * s;
                    v[a * 3 + b] = fold_load4<YH>(Y, ((long)f * th * tw + i * tw + j) * ldy + (ky * k + kx) * C + c);
                }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const bool in = i_lo + a <= i_hi && j_lo + b <= j_hi;
                    acc[0] += in ? v[a * 3 + b].x : 0.f; acc[1] += in ? v[a * 3 + b].y : 0.f;
                    acc[2] += in ? v[a * 3 + b].z : 0.f; acc[3] += in ? v[a * 3 + b].w : 0.f;
                }
            cnt = (i_hi - i_lo + 1) * (j_hi - j_lo + 1);
        } else {
            for (int i = i_lo; i <= i_hi; ++i) {
                const int ky = y + p - i * s;
                for (int j = j_lo; j <= j_hi; ++j) {
                    const int kx = x + p - j * s;
                    const float4 v = fold_load4<YH>(Y, ((long)f * th * tw + i * tw + j) * ldy + (ky * k + kx) * C + c);
                    acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w;
                    ++cnt;
                }
            }
        }
        if (normalize) {
            const float fc = (float)cnt;
            acc[0] /= fc; acc[1] /= fc; acc[2] /= fc; acc[3] /= fc;
        }
        if (res) {
            const float4 rv = *reinterpret_cast<const float4*>(res + pix * ldres + c);
            acc[0] = rv.x + acc[0]; acc[1] = rv.y + acc[1]; acc[2] = rv.z + acc[2]; acc[3] = rv.w + acc[3];
        }
        if (relu) { acc[0] = fmaxf(acc[0], 0.f); acc[1] = fmaxf(acc[1], 0.f); acc[2] = fmaxf(acc[2], 0.f); acc[3] = fmaxf(acc[3], 0.f); }
        store_f32_or_split(out, pix * ldo + c, ps_out, make_float4(acc[0], acc[1], acc[2], acc[3]), c);
    }
}

// ------------------------------------------------------------------ layout packing
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* src, int N, int C, int H, int W, float* dst, int ldd,
                                                           int coff, int zero_to, float scale, float shift) {
    const long HW = (long)H * W, total = (long)N * HW;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long)gridDim.x * blockDim.x) {
        const long n = pix / HW, rem = pix - n * HW;
        float* d = dst + pix * ldd + coff;
        for (int c = 0; c < C; ++c) d[c] = src[(n * C + c) * HW + rem] * scale + shift;
        for (int c = C; c < zero_to; ++c) d[c] = 0.f;
    }
}

__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* src, int lds, int coff, int N, int C, int H, int W,
                                                           float* dst) {
    const long HW = (long)H * W, total = (long)N * HW;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long)gridDim.x * blockDim.x) {
        const long n = pix / HW, rem = pix - n * HW;
        const float* s = src + pix * lds + coff;
        for (int c = 0; c < C; ++c) dst[(n * C + c) * HW + rem] = s[c];
    }
}

__global__ void __launch_bounds__(256) pad_tokens_kernel(const float* src, int lds, int bt, int h, int w, int C, int nh,
                                                         int nw, float* dst, int ldd, FgtFastDiv dc4, FgtFastDiv dnw) {
    const int c4n = C >> 2, per_f = nh * nw * c4n;
    const int f = blockIdx.y;                                     // grid: (items of one frame / 256, frames)
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per_f; e += gridDim.x * 256) {
        const int pl = fdiv(e, dc4), c = (e - pl * c4n) * 4;
        const int y = fdiv(pl, dnw), x = pl - y * nw;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y < h && x < w) v = *reinterpret_cast<const float4*>(src + (((long)f * h + y) * w + x) * lds + c);
        *reinterpret_cast<float4*>(dst + ((long)f * nh * nw + pl) * ldd + c) = v;
    }
}

__global__ void __launch_bounds__(256) axpby_kernel(const float* a, int lda, float sa, const float* b, int ldb, float sb,
                                                    long rows, int C, int act, float slope, float* out, int ldo) {
    const long total = rows * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / C; const int c = (int)(idx - r * C);
        float v = a[r * lda + c] * sa;
        if (b) v += b[r * ldb + c] * sb;
        out[r * ldo + c] = fgt_act(v, act, slope);
    }
}

// tool/video_inpainting.py:725-740.  U8 = false: `out` is the model output (fp32 NCHW, (-1,1)); U8 = true: `out` already holds
// astype(uint8)((x+1)/2*255), the form the window outputs are exchanged in between ranks (a quarter of the bytes; the truncation
// is the first thing the compose does with the value, :731-733, so it commutes with the exchange).
__device__ __forceinline__ float trunc_u8(float o) { return (float)(unsigned char)(int)(((o + 1.f) / 2.f) * 255.f); }

template <bool U8>
__global__ void __launch_bounds__(256) compose_kernel(const void* out_any, const int* ids, const int* first, int n,
                                                      const float* frames01, const float* masks, int H, int W, float* comp) {
    const long HW = (long)H * W, total = (long)n * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / HW); const long rem = idx - (long)i * HW;
        const int fid = ids[i];
        const float m = masks[(long)fid * HW + rem];
        for (int c = 0; c < 3; ++c) {
            const long o = ((long)i * 3 + c) * HW + rem;
            const float fu = U8 ? (float)static_cast<const unsigned char*>(out_any)[o]
                                : trunc_u8(static_cast<const float*>(out_any)[o]);          // astype(uint8): truncation
            const float vu = (float)(unsigned char)(int)(frames01[((long)fid * 3 + c) * HW + rem] * 255.0f);
            const float cv = fu * m + vu * (1.f - m);
            float* dst = comp + ((long)fid * HW + rem) * 3 + c;
            *dst = first[i] ? cv : (*dst * 0.5f + cv * 0.5f);
        }
    }
}

__global__ void __launch_bounds__(256) quantize_u8_kernel(const float* __restrict__ x, long n4, unsigned* __restrict__ dst) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        dst[i] = (unsigned)trunc_u8(v.x) | ((unsigned)trunc_u8(v.y) << 8) | ((unsigned)trunc_u8(v.z) << 16) | ((unsigned)trunc_u8(v.w) << 24);
    }
}

// tool/video_inpainting.py:697,719-721 + FGT/models/model.py:253-257 in one pass: frames in [0,1] -> (f*2-1)*(1-m) | m, channels-last
__global__ void __launch_bounds__(256) pack_frames_kernel(const float* frames01, const float* masks, const int* ids, int n, long HW,
                                                          float* dst, int ldd) {
    const long total = (long)n * HW;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long)gridDim.x * blockDim.x) {
        const long i = pix / HW, rem = pix - i * HW;
        const long f = ids ? ids[i] : i;
        const float m = masks[f * HW + rem];
        const float* s = frames01 + f * 3 * HW + rem;
        const float keep = 1.f - m;
        *reinterpret_cast<float4*>(dst + pix * ldd) = make_float4((s[0] * 2.f - 1.f) * keep, (s[HW] * 2.f - 1.f) * keep,
                                                                    (s[2 * HW] * 2.f - 1.f) * keep, m);
    }
}

// tool/video_inpainting.py:402-407 (+ :705): one workgroup per (output frame, channel); signed maximum over H*W, then x / max
__global__ void __launch_bounds__(256) norm_flows_kernel(const float* src, int n_src, int C, long HW, float* dst) {
    __shared__ float red[4];
    const int f = blockIdx.x / C, c = blockIdx.x - f * C;
    const float* s = src + ((long)min(f, n_src - 1) * C + c) * HW;
    float* d = dst + (long)blockIdx.x * HW;
    float mx = -INFINITY;
    for (long i = threadIdx.x; i < HW; i += 256) mx = fmaxf(mx, s[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    for (long i = threadIdx.x; i < HW; i += 256) d[i] = s[i] / mx;
}

// dst[i, :] = src[ids[i], :] for rows of row_len floats (row_len % 4 == 0): the window's frames out of the per-frame feature cache
__global__ void __launch_bounds__(256) gather_rows_kernel(const float* src, long ld_src, const int* ids, int n, long row4, float* dst,
                                                          long ld_dst) {
    const int i = blockIdx.y;                                     // grid: (float4 of one row / 256, rows)
    const float* s = src + (long)ids[i] * ld_src;
    float* d = dst + (long)i * ld_dst;
    for (long c4 = (long)blockIdx.x * 256 + threadIdx.x; c4 < row4; c4 += (long)gridDim.x * 256)
        *reinterpret_cast<float4*>(d + c4 * 4) = *reinterpret_cast<const float4*>(s + c4 * 4);
}

inline int grid_for(long total, int block = 256) {
    long g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}


// fp32 -> split tensor (hi = bf16_rne(x), lo = bf16_rne(x - hi)); one float4 per thread
__global__ void split_kernel(const float* __restrict__ x, long rows, int C4, int ldx, __bf16* __restrict__ out, int ld_s, long ps, int relu, FgtFastDiv dC) {
    const int total = (int)(rows * C4);                           // (host: < 2^31 items per launch)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const long r = fdiv(i, dC);
        const int c = (i - (int)r * C4) * 4;
        float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (ps < 0) {                                                                  // ps == -1: one fp16 plane
            *reinterpret_cast<uint2*>(out + r * ld_s + c) = fgt_half4(v);
            continue;
        }
        uint2 hi, lo;
        fgt_split4(v, hi, lo);
        __bf16* o = out + r * ld_s + (ps == 32 ? ((c >> 5) << 6) + (c & 31) : c);     // ps == 32: interleaved per 32 channels
        *reinterpret_cast<uint2*>(o) = hi;
        *reinterpret_cast<uint2*>(o + ps) = lo;
    }
}

// C % 4 != 0 but even (RAFT's 2-channel flow into the GRU input buffer): one channel pair per thread, planes layout only
__global__ void split2_kernel(const float* __restrict__ x, long rows, int C2, int ldx, __bf16* __restrict__ out, int ld_s, long ps, int relu, FgtFastDiv dC) {
    const int total = (int)(rows * C2);                           // (host: < 2^31 items per launch)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const long r = fdiv(i, dC);
        const int c = (i - (int)r * C2) * 2;
        float2 v = *reinterpret_cast<const float2*>(x + r * ldx + c);
        if (relu) v = make_float2(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f));
        uint2 hi, lo;
        fgt_split4(make_float4(v.x, v.y, 0.f, 0.f), hi, lo);          // THE definition of the format; the upper pair is unused
        __bf16* o = out + r * ld_s + c;
        *reinterpret_cast<unsigned*>(o) = hi.x;
        *reinterpret_cast<unsigned*>(o + ps) = lo.x;
    }
}

// ---- instance norm: double atomics into stats[N][C][2] = (sum, sumsq)
__global__ void __launch_bounds__(256) in_stats_kernel(const float* x, int ld, int HW, int C, int chunk, double* stats) {
    // grid: (pixel chunks, N).  Consecutive threads own consecutive channels (coalesced); when C < 256 the
    // remaining threads split the chunk's pixels ("sub" lanes).
    const int n = blockIdx.y;
    const long p0 = (long)blockIdx.x * chunk, p1 = min((long)HW, p0 + chunk);
    const int cpb = min(C, 256), lpc = 256 / cpb;
    const int t = threadIdx.x;
    if (t >= cpb * lpc) return;
    const int c0 = t % cpb, sub = t / cpb;
    for (int c = c0; c < C; c += cpb) {
        double s = 0.0, ss = 0.0;
        for (long p = p0 + sub; p < p1; p += lpc) {
            const float v = x[((long)n * HW + p) * ld + c];
            s += v; ss += (double)v * v;
        }
        atomicAdd(&stats[((long)n * C + c) * 2], s);
        atomicAdd(&stats[((long)n * C + c) * 2 + 1], ss);
    }
}

__global__ void __launch_bounds__(256) in_apply_kernel(const float* x, int ld, int N, int HW, int C, const double* stats, float eps,
                                                       int act, const float* res, int ldres, int act2, float* out, int ldo) {
    const long total = (long)N * HW * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C); const long pix = idx / C;
        const long n = pix / HW;
        const double mean = stats[(n * C + c) * 2] / HW;
        const double var = stats[(n * C + c) * 2 + 1] / HW - mean * mean;
        const float rstd = (float)(1.0 / sqrt((var > 0 ? var : 0.0) + (double)eps));
        float v = (x[pix * ld + c] - (float)mean) * rstd;
        v = fgt_act(v, act, 0.2f);
        if (res) { v += res[pix * ldres + c]; v = fgt_act(v, act2, 0.2f); }
        out[pix * ldo + c] = v;
    }
}


// ---- round 6: the vectorised pair for C % 4 == 0 (RAFT's fnet: 64 / 96 / 128 channels).  The kernels above handle one 4-byte element per work item
// with two 64-bit integer divisions, two fp64 divisions, an fp64 square root and an fp64 reciprocal PER ELEMENT: 1.4 TB/s, half of fnet's time
// (tools/raft_encode_breakdown.py).  Same arithmetic — fp64 sums of x and x * x per (image, channel); mean = s / HW, var = ss / HW - mean^2 in fp64,
// rstd = (float)(1 / sqrt(max(var, 0) + eps)), y = (x - (float)mean) * rstd — with the per-channel part done once per workgroup.
// Block = 256 threads = (C / 4 channel quads) x (256 / (C / 4) pixel lanes); grid (pixel chunks, N).
// Pixels per workgroup: enough workgroups to fill the chip on the small maps too (16 images of 60x108 at 2048 pixels per workgroup were 64 workgroups)
inline int in_chunk(int N, int HW) {
    long c = ((long)N * HW + 2047) / 2048;
    return (int)(c < 64 ? 64 : (c > 2048 ? 2048 : c));
}
__global__ void __launch_bounds__(256) in_stats4_kernel(const float* x, int ld, int HW, int C, int chunk, double* stats) {
    __shared__ double red[256][8];
    const int n = blockIdx.y, c4n = C >> 2, lpc = 256 / c4n, t = threadIdx.x;
    const long p0 = (long)blockIdx.x * chunk, p1 = min((long)HW, p0 + chunk);
    const int cq = t % c4n, sub = t / c4n;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    if (sub < lpc) {
        const float* xp = x + (long)n * HW * ld + cq * 4;
#pragma unroll 4
        for (long p = p0 + sub; p < p1; p += lpc) {
            const float4 v = *reinterpret_cast<const float4*>(xp + p * ld);
            s[0] += v.x; ss[0] += (double)v.x * v.x;
            s[1] += v.y; ss[1] += (double)v.y * v.y;
            s[2] += v.z; ss[2] += (double)v.z * v.z;
            s[3] += v.w; ss[3] += (double)v.w * v.w;
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { red[t][u] = s[u]; red[t][4 + u] = ss[u]; }
    __syncthreads();
    if (t < c4n) {                                   // fixed-order sum over the pixel lanes of this quad, then ONE atomic per (channel, moment) and block
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = red[t][u];
        for (int k = 1; k < lpc; ++k)
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] += red[t + k * c4n][u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            atomicAdd(&stats[((long)n * C + t * 4 + u) * 2], a[u]);
            atomicAdd(&stats[((long)n * C + t * 4 + u) * 2 + 1], a[4 + u]);
        }
    }
}

__global__ void __launch_bounds__(256) in_apply4_kernel(const float* x, int ld, int HW, int C, const double* stats, float eps, int act,
                                                        const float* res, int ldres, int act2, float* out, int ldo, float* out_s, int ldo_s, long ps_s, int chunk) {
    __shared__ float mr[1024][2];
    const int n = blockIdx.y, c4n = C >> 2, lpc = 256 / c4n, t = threadIdx.x;
    for (int c = t; c < C; c += 256) {
        const double mean = stats[((long)n * C + c) * 2] / HW;
        const double var = stats[((long)n * C + c) * 2 + 1] / HW - mean * mean;
        mr[c][0] = (float)mean;
        mr[c][1] = (float)(1.0 / sqrt((var > 0 ? var : 0.0) + (double)eps));
    }
    __syncthreads();
    const int cq = t % c4n, sub = t / c4n;
    if (sub >= lpc) return;
    const int c = cq * 4;
    const float m0 = mr[c][0], m1 = mr[c + 1][0], m2 = mr[c + 2][0], m3 = mr[c + 3][0];
    const float r0 = mr[c][1], r1 = mr[c + 1][1], r2 = mr[c + 2][1], r3 = mr[c + 3][1];
    const long p0 = (long)blockIdx.x * chunk, p1 = min((long)HW, p0 + chunk);
    const long base = (long)n * HW;
#pragma unroll 4
    for (long p = p0 + sub; p < p1; p += lpc) {
        const long pix = base + p;
        const float4 v = *reinterpret_cast<const float4*>(x + pix * ld + c);
        float4 y = make_float4(fgt_act((v.x - m0) * r0, act, 0.2f), fgt_act((v.y - m1) * r1, act, 0.2f), fgt_act((v.z - m2) * r2, act, 0.2f),
                               fgt_act((v.w - m3) * r3, act, 0.2f));
        if (res) {
            const float4 q = *reinterpret_cast<const float4*>(res + pix * ldres + c);
            y = make_float4(fgt_act(y.x + q.x, act2, 0.2f), fgt_act(y.y + q.y, act2, 0.2f), fgt_act(y.z + q.z, act2, 0.2f), fgt_act(y.w + q.w, act2, 0.2f));
        }
        if (out) *reinterpret_cast<float4*>(out + pix * ldo + c) = y;
        if (out_s) store_f32_or_split<false>(out_s, pix * ldo_s + c, ps_s, y, c);
    }
}

}  // namespace

extern "C" int fgt_layernorm(const float* x0, int C0, int ld0, const float* x1, int C1, int ld1, long rows, float eps,
                             const float* gA, const float* bA, float* outA, int ldA, const float* gB, const float* bB,
                             float* outB, int ldB, long long psA, long long psB, void* stream) {
    FGT_REQUIRE(x0 && gA && bA && outA && rows > 0, "fgt_layernorm: null pointer / empty");
    FGT_REQUIRE(C0 > 0 && C0 % 4 == 0 && C1 % 4 == 0 && (C0 + C1) <= 256 * LN_MAXV, "fgt_layernorm: C=(%d,%d) unsupported", C0, C1);
    FGT_REQUIRE(ld0 % 4 == 0 && (C1 == 0 || (x1 && ld1 % 4 == 0)) && ldA % 4 == 0 && (!outB || (gB && bB && ldB % 4 == 0)),
                "fgt_layernorm: strides must be multiples of 4 floats");
    FGT_REQUIRE((psA == -1 || (psA >= 0 && psA % 4 == 0)) && (psB == -1 || (psB >= 0 && psB % 4 == 0)),
                "fgt_layernorm: plane strides must be non-negative multiples of 4 (or -1: fp16 plane)");
    FGT_REQUIRE((psA != 32 || ((C0 + C1) % 32 == 0 && ldA >= 2 * (C0 + C1))) && (psB != 32 || ((C0 + C1) % 32 == 0 && ldB >= 2 * (C0 + C1))),
                "fgt_layernorm: an interleaved output (ps = 32) needs C %% 32 == 0 and a row stride of at least 2 * C elements");
    const auto ob = [](long long ps) { return ps < 0 ? 2.0 : 4.0; };
    FgtProfScope prof(FGT_PROF_LAYERNORM, 0.0, (double)rows * ((C0 + C1) * 4.0 + (C0 + C1) * (ob(psA) + (outB ? ob(psB) : 0.0))), stream);
    // FGT_LN_NT (default 1): non-temporal loads / stores — the rows are read once and the outputs are consumed by the NEXT kernel (A/B: =0)
    static const int nt = [] { const char* e = getenv("FGT_LN_NT"); return e ? atoi(e) : 1; }();
    if (nt)
        hipLaunchKernelGGL(layernorm_kernel<true>, dim3(cdiv(rows, 4 * LN_R)), dim3(256), 0, (hipStream_t)stream, x0, C0, ld0, x1, C1, ld1, rows,
                           eps, gA, bA, outA, ldA, gB, bB, outB, ldB, (long)psA, (long)psB);
    else
        hipLaunchKernelGGL(layernorm_kernel<false>, dim3(cdiv(rows, 4 * LN_R)), dim3(256), 0, (hipStream_t)stream, x0, C0, ld0, x1, C1, ld1, rows,
                           eps, gA, bA, outA, ldA, gB, bB, outB, ldB, (long)psA, (long)psB);
    return fgt_check_launch("layernorm");
}

extern "C" int fgt_dw_pool(const float* x0, int C0, int ld0, const float* x1, int C1, int ld1, int bt, int nh, int nw, int vh, int vw,
                           int k, const float* w, const float* bias, float* out, int ldo, void* stream) {
    FGT_REQUIRE(x0 && w && bias && out, "fgt_dw_pool: null pointer");
    FGT_REQUIRE(vh > 0 && vw > 0 && vh <= nh && vw <= nw, "fgt_dw_pool: real grid %dx%d must fit the padded grid %dx%d", vh, vw, nh, nw);
    FGT_REQUIRE(C0 % 4 == 0 && C1 % 4 == 0 && ld0 % 4 == 0 && (C1 == 0 || (x1 && ld1 % 4 == 0)) && ldo % 4 == 0, "fgt_dw_pool: alignment");
    FGT_REQUIRE(k > 0 && nh % k == 0 && nw % k == 0, "fgt_dw_pool: grid %dx%d not divisible by %d", nh, nw, k);
    const long total = (long)bt * (nh / k) * (nw / k) * ((C0 + C1) / 4);
    FgtProfScope prof(FGT_PROF_DW_POOL, 0.0, 4.0 * (C0 + C1) * ((double)bt * vh * vw + (double)k * k + 1.0 + (double)bt * (nh / k) * (nw / k)), stream);
    if (k == 4 && (((uintptr_t)w | (uintptr_t)bias) & 15) == 0 && bt <= 65535)
        hipLaunchKernelGGL(dw_pool4_kernel, dim3(grid_for((long)(nh / 4) * (nw / 4) * ((C0 + C1) / 4)), bt), dim3(256), 0, (hipStream_t)stream, x0, C0, ld0, x1, C1, ld1, bt, nh,
                           nw, vh, vw, w, bias, out, ldo, fgt_fastdiv_make((unsigned)((C0 + C1) / 4)), fgt_fastdiv_make((unsigned)(nw / 4)));
    else
        hipLaunchKernelGGL(dw_pool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x0, C0, ld0, x1, C1, ld1, bt, nh,
                           nw, vh, vw, k, w, bias, out, ldo);
    return fgt_check_launch("dw_pool");
}

extern "C" int fgt_dw3x3_residual(const float* x, int bt, int h, int w, int C, const float* wgt, const float* bias, float* out,
                                  void* stream) {
    FGT_REQUIRE(x && wgt && bias && out && C % 4 == 0 && bt > 0 && bt <= 65535 && (long)h * w * (C / 4) < (1l << 31), "fgt_dw3x3_residual: bad arguments");
    FGT_REQUIRE(((uintptr_t)wgt & 15) == 0, "fgt_dw3x3_residual: weights must be 16-byte aligned");
    FgtProfScope prof(FGT_PROF_POINTWISE, 0.0, 8.0 * (double)bt * h * w * C, stream);
    hipLaunchKernelGGL(dw3x3_res_kernel, dim3(grid_for((long)h * w * (C / 4)), bt), dim3(256), 0, (hipStream_t)stream, x, bt, h, w, C, wgt, bias, out,
                       fgt_fastdiv_make((unsigned)(C / 4)), fgt_fastdiv_make((unsigned)w));
    return fgt_check_launch("dw3x3_residual");
}

extern "C" int fgt_fold(const float* Y, int ldy, int frames, int th, int tw, int C, int k, int s, int p, int Hf, int Wf,
                        int normalize, const float* res, int ldres, float* out, int ldo, int relu, long long ps_out, int y_f16, void* stream) {
    FGT_REQUIRE(Y && out && C % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && (!res || ldres % 4 == 0), "fgt_fold: bad arguments");
    FGT_REQUIRE((Hf + 2 * p - k) / s + 1 == th && (Wf + 2 * p - k) / s + 1 == tw, "fgt_fold: token grid %dx%d does not match output %dx%d", th, tw, Hf, Wf);
    const long total = (long)frames * Hf * Wf * (C / 4);
    FGT_REQUIRE(ps_out == -1 || (ps_out >= 0 && ps_out % 4 == 0), "fgt_fold: plane stride must be a non-negative multiple of 4 (or -1: fp16 plane)");
    FGT_REQUIRE(y_f16 == 0 || y_f16 == 1, "fgt_fold: y_f16 must be 0 or 1");
    FGT_REQUIRE(ps_out != 32 || (C % 32 == 0 && ldo >= 2 * C), "fgt_fold: an interleaved output (ps_out = 32) needs C %% 32 == 0 and ldo >= 2 * C elements");
    FgtProfScope prof(FGT_PROF_FOLD, 0.0, (double)frames * th * tw * k * k * C * (y_f16 ? 2.0 : 4.0) +
                                              (double)frames * Hf * Wf * C * ((ps_out < 0 ? 2.0 : 4.0) + (res ? 4.0 : 0.0)), stream);
    if (y_f16)
        hipLaunchKernelGGL(fold_kernel<true>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, Y, ldy, frames, th, tw, C, k, s, p,
                           Hf, Wf, normalize, res, ldres, out, ldo, relu, (long)ps_out);
    else
        hipLaunchKernelGGL(fold_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, Y, ldy, frames, th, tw, C, k, s, p,
                           Hf, Wf, normalize, res, ldres, out, ldo, relu, (long)ps_out);
    return fgt_check_launch("fold");
}

extern "C" int fgt_nchw_to_nhwc(const float* src, int N, int C, int H, int W, float* dst, int ldd, int coff, int zero_to,
                                float scale, float shift, void* stream) {
    FGT_REQUIRE(src && dst && C > 0, "fgt_nchw_to_nhwc: bad arguments");
    FgtProfScope prof(FGT_PROF_POINTWISE, 0.0, 4.0 * (double)N * H * W * (C + (zero_to > C ? zero_to : C)), stream);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, src, N, C, H, W,
                       dst, ldd, coff, zero_to, scale, shift);
    return fgt_check_launch("nchw_to_nhwc");
}

extern "C" int fgt_nhwc_to_nchw(const float* src, int lds, int coff, int N, int C, int H, int W, float* dst, void* stream) {
    FGT_REQUIRE(src && dst && C > 0, "fgt_nhwc_to_nchw: bad arguments");
    FgtProfScope prof(FGT_PROF_POINTWISE, 0.0, 8.0 * (double)N * H * W * C, stream);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, src, lds, coff, N,
                       C, H, W, dst);
    return fgt_check_launch("nhwc_to_nchw");
}

extern "C" int fgt_pad_tokens(const float* src, int lds, int bt, int h, int w, int C, int nh, int nw, float* dst, int ldd,
                              void* stream) {
    FGT_REQUIRE(src && dst && C % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0, "fgt_pad_tokens: bad arguments");
    FGT_REQUIRE(bt > 0 && bt <= 65535 && (long)nh * nw * (C / 4) < (1l << 31), "fgt_pad_tokens: frame count / frame size out of range");
    FgtProfScope prof(FGT_PROF_POINTWISE, 0.0, 4.0 * C * ((double)bt * (h < nh ? h : nh) * (w < nw ? w : nw) + (double)bt * nh * nw), stream);
    hipLaunchKernelGGL(pad_tokens_kernel, dim3(grid_for((long)nh * nw * (C / 4)), bt), dim3(256), 0, (hipStream_t)stream, src, lds, bt, h, w, C, nh, nw,
                       dst, ldd, fgt_fastdiv_make((unsigned)(C / 4)), fgt_fastdiv_make((unsigned)nw));
    return fgt_check_launch("pad_tokens");
}

extern "C" int fgt_axpby(const float* a, int lda, float sa, const float* b, int ldb, float sb, long rows, int C, int act,
                         float slope, float* out, int ldo, void* stream) {
    FGT_REQUIRE(a && out && rows > 0 && C > 0, "fgt_axpby: bad arguments");
    FgtProfScope prof(FGT_PROF_POINTWISE, 0.0, 4.0 * (double)rows * C * (b ? 3.0 : 2.0), stream);
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, a, lda, sa, b, ldb, sb, rows, C,
                       act, slope, out, ldo);
    return fgt_check_launch("axpby");
}

extern "C" int fgt_compose_blend(const float* out_nchw, const int* ids, const int* first, int n, const float* frames01,
                                 const float* masks, int H, int W, float* comp, void* stream) {
    FGT_REQUIRE(out_nchw && ids && first && frames01 && masks && comp && n > 0, "fgt_compose_blend: bad arguments");
    hipLaunchKernelGGL(compose_kernel<false>, dim3(grid_for((long)n * H * W)), dim3(256), 0, (hipStream_t)stream, out_nchw, ids, first, n,
                       frames01, masks, H, W, comp);
    return fgt_check_launch("compose_blend");
}

extern "C" int fgt_compose_blend_u8(const unsigned char* filled_u8, const int* ids, const int* first, int n, const float* frames01,
                                    const float* masks, int H, int W, float* comp, void* stream) {
    FGT_REQUIRE(filled_u8 && ids && first && frames01 && masks && comp && n > 0, "fgt_compose_blend_u8: bad arguments");
    hipLaunchKernelGGL(compose_kernel<true>, dim3(grid_for((long)n * H * W)), dim3(256), 0, (hipStream_t)stream, filled_u8, ids, first, n,
                       frames01, masks, H, W, comp);
    return fgt_check_launch("compose_blend_u8");
}

extern "C" int fgt_quantize_u8(const float* x, long count, unsigned char* dst, void* stream) {
    FGT_REQUIRE(x && dst && count > 0 && count % 4 == 0 && (((uintptr_t)x & 15) | ((uintptr_t)dst & 3)) == 0,
                "fgt_quantize_u8: count must be a multiple of 4 and the pointers 16- / 4-byte aligned");
    hipLaunchKernelGGL(quantize_u8_kernel, dim3(grid_for(count / 4)), dim3(256), 0, (hipStream_t)stream, x, count / 4,
                       reinterpret_cast<unsigned*>(dst));
    return fgt_check_launch("quantize_u8");
}

extern "C" int fgt_pack_frames(const float* frames01, const float* masks, const int* ids, int n, int H, int W, float* dst, int ldd,
                               void* stream) {
    FGT_REQUIRE(frames01 && masks && dst && n > 0 && H > 0 && W > 0, "fgt_pack_frames: bad arguments");
    FGT_REQUIRE(ldd >= 4 && ldd % 4 == 0 && ((uintptr_t)dst & 15) == 0, "fgt_pack_frames: dst must be float4 aligned with ldd %% 4 == 0");
    FgtProfScope prof(FGT_PROF_POINTWISE, 0.0, 4.0 * (double)n * H * W * 8.0, stream);
    hipLaunchKernelGGL(pack_frames_kernel, dim3(grid_for((long)n * H * W)), dim3(256), 0, (hipStream_t)stream, frames01, masks, ids, n,
                       (long)H * W, dst, ldd);
    return fgt_check_launch("pack_frames");
}

extern "C" int fgt_norm_flows(const float* flows, int n_src, int n_out, int C, long HW, float* out, void* stream) {
    FGT_REQUIRE(flows && out && n_src > 0 && n_out > 0 && C > 0 && HW > 0, "fgt_norm_flows: bad arguments");
    hipLaunchKernelGGL(norm_flows_kernel, dim3(n_out * C), dim3(256), 0, (hipStream_t)stream, flows, n_src, C, HW, out);
    return fgt_check_launch("norm_flows");
}

extern "C" int fgt_gather_rows(const float* src, long ld_src, const int* ids, int n, long row_len, float* dst, long ld_dst, void* stream) {
    FGT_REQUIRE(src && ids && dst && n > 0 && row_len > 0, "fgt_gather_rows: bad arguments");
    FGT_REQUIRE(row_len % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0,
                "fgt_gather_rows: rows must be float4 aligned");
    FgtProfScope prof(FGT_PROF_POINTWISE, 0.0, 8.0 * (double)n * row_len, stream);
    FGT_REQUIRE(n <= 65535, "fgt_gather_rows: at most 65535 rows per call");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(row_len / 4), n), dim3(256), 0, (hipStream_t)stream, src, ld_src, ids, n,
                       row_len / 4, dst, ld_dst);
    return fgt_check_launch("gather_rows");
}

extern "C" int fgt_split(const float* x, long rows, int C, int ldx, void* out_s, int ld_s, long long ps, int relu, void* stream) {
    FGT_REQUIRE(x && out_s && rows > 0 && C > 0, "fgt_split: bad arguments");
    if (C % 4 != 0) {       // channel pairs (planes layout): e.g. a 2-channel slice at the end of a wider split buffer
        FGT_REQUIRE(C % 2 == 0 && ldx % 2 == 0 && ld_s % 2 == 0 && ps > 0 && ps % 2 == 0 && ps != 32 && ((uintptr_t)x & 7) == 0 && ((uintptr_t)out_s & 3) == 0,
                    "fgt_split: C %% 4 != 0 needs even C, strides and plane stride (planes layout) and 8- / 4-byte aligned pointers");
        FgtProfScope prof2(FGT_PROF_POINTWISE, 0.0, (double)rows * C * 8.0, stream);
        FGT_REQUIRE(rows * (C / 2) < (1l << 31), "fgt_split: too many items for one call");
        hipLaunchKernelGGL(split2_kernel, dim3(grid_for(rows * (C / 2))), dim3(256), 0, (hipStream_t)stream, x, rows, C / 2, ldx,
                           static_cast<__bf16*>(out_s), ld_s, (long)ps, relu, fgt_fastdiv_make((unsigned)(C / 2)));
        return fgt_check_launch("split2");
    }
    FGT_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && ld_s % 4 == 0 && (ps == -1 || (ps > 0 && ps % 4 == 0)) && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out_s & 7) == 0,
                "fgt_split: C, strides must be multiples of 4 and pointers aligned");
    FgtProfScope prof(FGT_PROF_POINTWISE, 0.0, (double)rows * C * (4.0 + (ps < 0 ? 2.0 : 4.0)), stream);
    // (32-bit item index in the kernel: rows are taken in chunks of < 2^31 items)
    const long per = ((1l << 31) - 1) / (C / 4);
    for (long r0 = 0; r0 < rows; r0 += per) {
        const long nr = rows - r0 < per ? rows - r0 : per;
        hipLaunchKernelGGL(split_kernel, dim3(grid_for(nr * (C / 4))), dim3(256), 0, (hipStream_t)stream, x + r0 * ldx, nr, C / 4, ldx,
                           static_cast<__bf16*>(out_s) + r0 * ld_s, ld_s, (long)ps, relu, fgt_fastdiv_make((unsigned)(C / 4)));
    }
    return fgt_check_launch("split");
}

extern "C" int fgt_instnorm_stats(const float* x, int ld, int N, int HW, int C, double* stats, void* stream) {
    FGT_REQUIRE(x && stats && N > 0 && HW > 0 && C > 0 && C <= 1024, "fgt_instnorm_stats: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(stats, 0, sizeof(double) * 2 * N * C, s) != hipSuccess) { fgt_set_error("fgt_instnorm_stats: memset failed"); return FGT_ELAUNCH; }
    if (C % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && N <= 65535) {
        const int ck = in_chunk(N, HW);
        hipLaunchKernelGGL(in_stats4_kernel, dim3(cdiv(HW, ck), N), dim3(256), 0, s, x, ld, HW, C, ck, stats);
        return fgt_check_launch("instnorm_stats");
    }
    const int chunk = 512;
    dim3 grid(cdiv(HW, chunk), N);
    hipLaunchKernelGGL(in_stats_kernel, grid, dim3(256), 0, s, x, ld, HW, C, chunk, stats);
    return fgt_check_launch("instnorm_stats");
}

extern "C" int fgt_instnorm_apply_split(const float* x, int ld, int N, int HW, int C, const double* stats, float eps, int act,
                                        const float* res, int ldres, int act2, float* out, int ldo, void* out_s, int ldo_s, long long ps_s, void* stream) {
    FGT_REQUIRE(x && stats && (out || out_s) && N > 0 && HW > 0 && C > 0 && C <= 1024, "fgt_instnorm_apply: bad arguments");
    const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = C % 4 == 0 && ld % 4 == 0 && al16(x) && (!res || (ldres % 4 == 0 && al16(res))) && (!out || (ldo % 4 == 0 && al16(out))) && N <= 65535;
    if (out_s) {
        FGT_REQUIRE(vec, "fgt_instnorm_apply: a split output needs C, strides multiples of 4 and 16-byte aligned pointers");
        FGT_REQUIRE(ldo_s % 4 == 0 && (ps_s == -1 || (ps_s > 0 && ps_s % 4 == 0)) && (reinterpret_cast<uintptr_t>(out_s) & 7) == 0 && (ps_s != 32 || (C % 32 == 0 && ldo_s >= 2 * C)),
                    "fgt_instnorm_apply: split output: ldo_s %d, ps %lld (32 = interleaved: C %% 32 == 0, ldo_s >= 2 C)", ldo_s, ps_s);
    }
    if (vec) {
        const int ck = in_chunk(N, HW);
        hipLaunchKernelGGL(in_apply4_kernel, dim3(cdiv(HW, ck), N), dim3(256), 0, (hipStream_t)stream, x, ld, HW, C, stats, eps, act, res, ldres, act2,
                           out, ldo, static_cast<float*>(out_s), ldo_s, (long)ps_s, ck);
        return fgt_check_launch("instnorm_apply");
    }
    hipLaunchKernelGGL(in_apply_kernel, dim3(grid_for((long)N * HW * C)), dim3(256), 0, (hipStream_t)stream, x, ld, N, HW, C, stats,
                       eps, act, res, ldres, act2, out, ldo);
    return fgt_check_launch("instnorm_apply");
}

extern "C" int fgt_instnorm_apply(const float* x, int ld, int N, int HW, int C, const double* stats, float eps, int act,
                                  const float* res, int ldres, int act2, float* out, int ldo, void* stream) {
    return fgt_instnorm_apply_split(x, ld, N, HW, C, stats, eps, act, res, ldres, act2, out, ldo, nullptr, 0, 0, stream);
}
