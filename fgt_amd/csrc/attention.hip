// Streaming-softmax attention for FGT's two attention flavours, head dim 128, fp32 matrix cores.
//
// A wavefront owns 32 queries.  Both contractions are issued "swapped" so the query index stays in
// the lane (lane & 31) from QK^T through the softmax to PV and no cross-lane transpose is needed:
//   S^T[key][q] = sum_d K[key][d] Q[q][d]      A = K tile (LDS), B = Q (registers)   -> C: col = q, rows = keys
//   O^T[d][q]  += sum_key V[key][d] P[q][key]  A = V tile (LDS), B = P (the S^T accumulator registers themselves)
// The k index of v_mfma_f32_32x32x2_f32 is lane>>5 (h).  For QK^T step s contracts d = 64*h + s, so each
// lane needs a contiguous 64-float run of its Q/K row (float4 LDS reads).  For PV step e contracts the key
// the lane already holds in accumulator register e: key(e, h) = (e&3) + 8*(e>>2) + 4*h.
// Row max / sum need one exchange with lane^32.  O^T keeps q in the lane, so the online-softmax rescale is
// a per-lane scalar multiply.
//
// The zone / window / head gathers of the reference (attention_base.py:61-69, attention_flow.py:76-108)
// are folded into the row addressing: Q, K, V are read in place from the projection GEMM outputs.
#include <stdlib.h>
#include "common.h"

namespace {

struct AttnP {
    fgt_attn_desc d;
    const float *Q, *K, *V, *KG, *VG;
    float* O;
    int n_q, n_k, zh, zw, gh, gw, n_loc;
    float scale_log2e;
};

constexpr int HD = 128;    // head dim
constexpr int KT = 32;     // keys per tile
constexpr int KLD = HD + 4;  // K tile row stride (floats): float4-aligned, conflict-free b128 reads

struct Prob { int frame0, zi, zj, hd; };

// mode 1 with desc.compact: row of the Q / K / V maps that padded-grid pixel `pix` (index into [bt, nh, nw]) reads
__device__ __forceinline__ long attn_map_row(const fgt_attn_desc& d, long pix) {
    if (d.mode != 1 || !d.compact) return pix;
    const int plane = d.nh * d.nw;
    const int fr = (int)(pix / plane), rem = (int)(pix - (long)fr * plane);
    const int y = rem / d.nw, x = rem - y * d.nw;
    return (y < d.h && x < d.w) ? ((long)fr * d.h + y) * d.w + x : (long)d.pad_row;
}


// pixel index (row of a [bt, nh, nw] map) of local token n of the problem
__device__ __forceinline__ int local_pix(const AttnP& p, const Prob& pr, int n) {
    const fgt_attn_desc& d = p.d;
    if (d.mode == 0) {
        const int zsz = p.zh * p.zw;
        const int tt = n / zsz, rem = n - tt * zsz;
        const int i = rem / p.zw, j = rem - i * p.zw;
        return ((pr.frame0 + tt) * d.nh + pr.zi * p.zh + i) * d.nw + pr.zj * p.zw + j;
    } else {
        const int a = n / d.ws, b = n - a * d.ws;
        return (pr.frame0 * d.nh + pr.zi * d.ws + a) * d.nw + pr.zj * d.ws + b;
    }
}

// Row pointers of the KT keys of a tile (zone / window / global-token addressing), computed by KT threads per tile so
// that the tile loaders do no integer divisions.
__device__ __forceinline__ void key_pointers(const AttnP& p, const Prob& pr, int choff, int k0, int tid,
                                             const float** kptr, const float** vptr) {
    if (tid < KT) {
        const fgt_attn_desc& d = p.d;
        const int key = min(k0 + tid, p.n_k - 1);
        if (key < p.n_loc) {
            const long pix = attn_map_row(d, local_pix(p, pr, key));
            kptr[tid] = p.K + pix * d.ldk + d.koff + choff;
            vptr[tid] = p.V + pix * d.ldv + d.voff + choff;
        } else {
            const long gr = (long)pr.frame0 * d.n_global + (key - p.n_loc);
            kptr[tid] = p.KG + gr * d.ldg_k + choff;
            vptr[tid] = p.VG + gr * d.ldg_v + choff;
        }
    }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64, 2) attn_kernel(const AttnP p) {
    constexpr int NT = NW * 64;
    constexpr int LD_IT = KT * (HD / 4) / NT;  // float4 per thread per tensor per tile
    __shared__ __attribute__((aligned(16))) float Ks[KT * KLD];
    __shared__ __attribute__((aligned(16))) float Vs[KT * HD];
    __shared__ const float* kptr[KT];
    __shared__ const float* vptr[KT];

    const fgt_attn_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    Prob pr;
    {
        int y = blockIdx.y;
        pr.hd = y % d.heads; y /= d.heads;
        if (d.mode == 0) {
            pr.zj = y % d.group; y /= d.group;
            pr.zi = y % d.group; y /= d.group;
            pr.frame0 = y * d.t;
        } else {
            pr.zj = y % p.gw; y /= p.gw;
            pr.zi = y % p.gh; y /= p.gh;
            pr.frame0 = y;
        }
    }
    const int choff = pr.hd * HD;

    // ---- Q rows into registers: q[s] = Q[row][64*lh + s]
    const int qi = blockIdx.x * (NW * 32) + wave * 32 + l31;
    const int qpix = local_pix(p, pr, min(qi, p.n_q - 1));
    float q[64];
    {
        const float* qp = p.Q + attn_map_row(d, qpix) * d.ldq + d.qoff + choff + 64 * lh;
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const float4 v = *reinterpret_cast<const float4*>(qp + 4 * s4);
            q[4 * s4 + 0] = v.x; q[4 * s4 + 1] = v.y; q[4 * s4 + 2] = v.z; q[4 * s4 + 3] = v.w;
        }
    }

    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < p.n_k; k0 += KT) {
        key_pointers(p, pr, choff, k0, tid, kptr, vptr);
        __syncthreads();  // previous tile fully consumed, pointer table visible
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx >> 5, c4 = idx & 31;
            const float4 kv = *reinterpret_cast<const float4*>(kptr[row] + c4 * 4);
            const float4 vv = *reinterpret_cast<const float4*>(vptr[row] + c4 * 4);
            *reinterpret_cast<float4*>(Ks + row * KLD + c4 * 4) = kv;
            *reinterpret_cast<float4*>(Vs + row * HD + c4 * 4) = vv;
        }
        __syncthreads();

        // ---- S^T tile = K . Q^T
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        const float* krow = Ks + l31 * KLD + 64 * lh;
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            const float4 kf = *reinterpret_cast<const float4*>(krow + 4 * s4);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, q[4 * s4 + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, q[4 * s4 + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, q[4 * s4 + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, q[4 * s4 + 3], s, 0, 0, 0);
        }
        // ---- online softmax (base-2 exponent), keys of this lane: k0 + (e&3) + 8*(e>>2) + 4*lh
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float v = key < p.n_k ? s[e] * p.scale_log2e : -INFINITY;
            s[e] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pe = exp2f(s[e] - m_new);
            s[e] = pe;
            psum += pe;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float* vrow = Vs + ((e & 3) + 8 * (e >> 2) + 4 * lh) * HD + l31;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[t * 32], s[e], o[t], 0, 0, 0);
        }
    }

    // ---- normalise and store: lane holds q = l31; o[t][e] is d = t*32 + (e&3) + 8*(e>>2) + 4*lh
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    if (qi < p.n_q) {
        long orow;
        bool keep = true;
        if (d.mode == 0) {
            // tq: only the first tq frames of every batch element are queried and O holds b * tq frames, compactly
            orow = qpix - (d.tq > 0 ? (long)(pr.frame0 / d.t) * (d.t - d.tq) * d.nh * d.nw : 0);
        } else {
            const int fr = qpix / (d.nh * d.nw), rem = qpix - fr * (d.nh * d.nw);
            const int y = rem / d.nw, x = rem - y * d.nw;
            keep = y < d.h && x < d.w;
            orow = ((long)fr * d.h + y) * d.w + x;
        }
        if (keep) {
            const long ob = orow * d.ldo + choff + 4 * lh;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const float4 v = make_float4(o[t][4 * e4 + 0] * inv, o[t][4 * e4 + 1] * inv,
                                                 o[t][4 * e4 + 2] * inv, o[t][4 * e4 + 3] * inv);
                    if (d.out_split) {       // O is the hi plane of a split tensor (bf16 elements), lo plane pso further
                        uint2 hi, lo;
                        fgt_split4(v, hi, lo);
                        __bf16* o16 = reinterpret_cast<__bf16*>(p.O) + ob + t * 32 + 8 * e4;
                        *reinterpret_cast<uint2*>(o16) = hi;
                        *reinterpret_cast<uint2*>(o16 + d.pso) = lo;
                    } else {
                        *reinterpret_cast<float4*>(p.O + ob + t * 32 + 8 * e4) = v;
                    }
                }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------
// bf16x3 variant: every fp32 operand (Q, K, P, V) is split into hi = bf16(x), lo = bf16(x - hi) and each product is
// three v_mfma_f32_32x32x16_bf16 (lo*hi + hi*lo + hi*hi, fp32 accumulate) — 5.3x the fp32-MFMA rate at ~2^-16 relative
// error per product.  Same swapped formulation as above; the k index of the 32x32x16 MFMA is (lane>>5)*8 + j:
//   QK^T step s (8 per tile): d = 16*s + 8*h + j          -> K tile rows [key][d] (272-byte rows), Q in registers
//   PV   step ks (2 per tile): key = key(e = 8*ks + j, h)  -> P straight from the S^T accumulator registers 8ks..8ks+7;
//                                                             V is stored TRANSPOSED [d][pos], pos = (2*ks + h)*8 + j,
//                                                             so a lane's 8 keys are one 16-byte LDS read.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KLDB = HD + 8;   // bf16 per K row (272 bytes): conflict-free ds_read_b128 across 16 rows
constexpr int VLDB = KT + 8;   // bf16 per V^T row (80 bytes)

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 l = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xFFFF0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2));
}

struct U4 { unsigned x, y, z, w; };
__device__ __forceinline__ bf16x8 as_bf16x8(unsigned a, unsigned b, unsigned c, unsigned d) {
    const U4 u = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, u);
}

template <int NW, bool PF>
__global__ void __launch_bounds__(NW * 64, 2) attn_bf16x3_kernel(const AttnP p) {
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) __bf16 Khi[KT * KLDB];
    __shared__ __attribute__((aligned(16))) __bf16 Klo[KT * KLDB];
    __shared__ __attribute__((aligned(16))) __bf16 Vhi[HD * VLDB];
    __shared__ __attribute__((aligned(16))) __bf16 Vlo[HD * VLDB];
    __shared__ const float* kptr[KT];
    __shared__ const float* vptr[KT];

    const fgt_attn_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    Prob pr;
    {
        int y = blockIdx.y;
        pr.hd = y % d.heads; y /= d.heads;
        if (d.mode == 0) {
            pr.zj = y % d.group; y /= d.group;
            pr.zi = y % d.group; y /= d.group;
            pr.frame0 = y * d.t;
        } else {
            pr.zj = y % p.gw; y /= p.gw;
            pr.zi = y % p.gh; y /= p.gh;
            pr.frame0 = y;
        }
    }
    const int choff = pr.hd * HD;

    // ---- Q rows into registers, pre-scaled by log2(e)/sqrt(d) and split: step s holds d = 16s + 8h + (0..7)
    const int qi = blockIdx.x * (NW * 32) + wave * 32 + l31;
    const int qpix = local_pix(p, pr, min(qi, p.n_q - 1));
    bf16x8 qh[8], ql[8];
    {
        const float* qp = p.Q + attn_map_row(d, qpix) * d.ldq + d.qoff + choff + 8 * lh;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(qp + 16 * s);
            const float4 b = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
            unsigned h0, h1, h2, h3, l0, l1, l2, l3;
            const float sc = p.scale_log2e;
            split2(a.x * sc, a.y * sc, h0, l0); split2(a.z * sc, a.w * sc, h1, l1);
            split2(b.x * sc, b.y * sc, h2, l2); split2(b.z * sc, b.w * sc, h3, l3);
            qh[s] = as_bf16x8(h0, h1, h2, h3);
            ql[s] = as_bf16x8(l0, l1, l2, l3);
        }
    }

    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // Register prefetch: the raw fp32 K/V values of tile i+1 are fetched while tile i is being multiplied, so the
    // global-load latency (57 % of the wave time was s_waitcnt/barrier in the non-prefetching version) overlaps the MFMAs.
    constexpr int K_IT = KT * (HD / 4) / NT, V_IT = HD * 4 / NT;
    float4 kreg[K_IT];
    float vreg[V_IT][8];
    auto gload = [&]() {
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int idx = tid + it * NT;
            kreg[it] = *reinterpret_cast<const float4*>(kptr[idx >> 5] + (idx & 31) * 4);
        }
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = tid + it * NT;
            const int dd = idx & (HD - 1), g = idx >> 7;        // HD == 128; g = 2*ks + h
            const int kbase = 16 * (g >> 1) + 4 * (g & 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) vreg[it][j] = vptr[kbase + (j & 3) + 8 * (j >> 2)][dd];
        }
    };
    auto lstore = [&]() {
        // K tile: [key][d] rows; V tile transposed [d][pos], pos = (2*ks + h)*8 + j  <->  key(8ks + j, h)
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int idx = tid + it * NT;
            const int row = idx >> 5, c4 = idx & 31;
            unsigned h0, h1, l0, l1;
            split2(kreg[it].x, kreg[it].y, h0, l0); split2(kreg[it].z, kreg[it].w, h1, l1);
            *reinterpret_cast<uint2*>(Khi + row * KLDB + c4 * 4) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(Klo + row * KLDB + c4 * 4) = make_uint2(l0, l1);
        }
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int idx = tid + it * NT;
            const int dd = idx & (HD - 1), g = idx >> 7;
            const float* v = vreg[it];
            unsigned h0, h1, h2, h3, l0, l1, l2, l3;
            split2(v[0], v[1], h0, l0); split2(v[2], v[3], h1, l1); split2(v[4], v[5], h2, l2); split2(v[6], v[7], h3, l3);
            *reinterpret_cast<uint4*>(Vhi + dd * VLDB + g * 8) = make_uint4(h0, h1, h2, h3);
            *reinterpret_cast<uint4*>(Vlo + dd * VLDB + g * 8) = make_uint4(l0, l1, l2, l3);
        }
    };

    if constexpr (PF) {
        key_pointers(p, pr, choff, 0, tid, kptr, vptr);
        __syncthreads();
        gload();
        // every wavefront must have READ tile 0's row pointers before the first loop iteration overwrites the table with tile 1's
        // (without this barrier wavefront 0 could run ahead: found by tests/test_ops_gpu.py::test_attention_temporal_long_zones,
        //  the 8-wavefront instance mixed rows of tile 1 into tile 0)
        __syncthreads();
    }
    for (int k0 = 0; k0 < p.n_k; k0 += KT) {
        if constexpr (PF) {
            lstore();                                                 // tile k0: registers -> LDS (waits for its loads)
            key_pointers(p, pr, choff, k0 + KT, tid, kptr, vptr);     // rows of the next tile (clamped past the end)
            __syncthreads();
            gload();                                                  // next tile in flight during the MFMAs below
        } else {                                                      // short key lists (spatial windows): no prefetch registers
            key_pointers(p, pr, choff, k0, tid, kptr, vptr);
            __syncthreads();
            gload();
            lstore();
            __syncthreads();
        }

        // ---- S^T tile = K . Q^T  (already in log2 units: Q carries the scale)
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        {
            const __bf16* kh = Khi + l31 * KLDB + 8 * lh;
            const __bf16* kl = Klo + l31 * KLDB + 8 * lh;
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(kh + 16 * st);
                const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(kl + 16 * st);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, qh[st], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, ql[st], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, qh[st], s, 0, 0, 0);
            }
        }
        // The softmax bookkeeping is VALU work beside 48 MFMAs per tile (PMC: 8.1 VALU per MFMA, wavefronts 26 % VALU-active):
        // exp2 is the bare v_exp_f32 (arguments <= 0; results below 2^-126 flush to 0, which is what they contribute) instead of the
        // library exp2f with its denormal-range rescaling (~5 instructions per call, 17 calls per tile).  (Masking only the last tile, or
        // skipping the O rescale when no query saw a new maximum, needs a branch that spills registers in the 128-VGPR 8-wave variant.)
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int key = k0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const float v = key < p.n_k ? s[e] : -INFINITY;
            s[e] = v;
            mx = fmaxf(mx, v);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float pe = __builtin_amdgcn_exp2f(s[e] - m_new);
            s[e] = pe;
            psum += pe;
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
        // ---- O^T += V^T . P^T : P split from the accumulator registers (keys 8ks..8ks+7 of this lane half)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            unsigned h0, h1, h2, h3, l0, l1, l2, l3;
            split2(s[8 * ks + 0], s[8 * ks + 1], h0, l0); split2(s[8 * ks + 2], s[8 * ks + 3], h1, l1);
            split2(s[8 * ks + 4], s[8 * ks + 5], h2, l2); split2(s[8 * ks + 6], s[8 * ks + 7], h3, l3);
            const bf16x8 p_h = as_bf16x8(h0, h1, h2, h3), p_l = as_bf16x8(l0, l1, l2, l3);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int off = (t * 32 + l31) * VLDB + (2 * ks + lh) * 8;
                const bf16x8 v_h = *reinterpret_cast<const bf16x8*>(Vhi + off);
                const bf16x8 v_l = *reinterpret_cast<const bf16x8*>(Vlo + off);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_l, p_h, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h, p_l, o[t], 0, 0, 0);
                o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h, p_h, o[t], 0, 0, 0);
            }
        }
        if constexpr (PF) __syncthreads();                            // tile consumed: LDS may be overwritten
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    if (qi < p.n_q) {
        long orow;
        bool keep = true;
        if (d.mode == 0) {
            // tq: only the first tq frames of every batch element are queried and O holds b * tq frames, compactly
            orow = qpix - (d.tq > 0 ? (long)(pr.frame0 / d.t) * (d.t - d.tq) * d.nh * d.nw : 0);
        } else {
            const int fr = qpix / (d.nh * d.nw), rem = qpix - fr * (d.nh * d.nw);
            const int y = rem / d.nw, x = rem - y * d.nw;
            keep = y < d.h && x < d.w;
            orow = ((long)fr * d.h + y) * d.w + x;
        }
        if (keep) {
            const long ob = orow * d.ldo + choff + 4 * lh;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const float4 v = make_float4(o[t][4 * e4 + 0] * inv, o[t][4 * e4 + 1] * inv,
                                                 o[t][4 * e4 + 2] * inv, o[t][4 * e4 + 3] * inv);
                    if (d.out_split) {       // O is the hi plane of a split tensor (bf16 elements), lo plane pso further
                        uint2 hi, lo;
                        fgt_split4(v, hi, lo);
                        __bf16* o16 = reinterpret_cast<__bf16*>(p.O) + ob + t * 32 + 8 * e4;
                        *reinterpret_cast<uint2*>(o16) = hi;
                        *reinterpret_cast<uint2*>(o16 + d.pso) = lo;
                    } else {
                        *reinterpret_cast<float4*>(p.O + ob + t * 32 + 8 * e4) = v;
                    }
                }
        }
    }
}

bool attn_nw8() {
    static const int v = [] { const char* e = getenv("FGT_ATTN_NW8"); return e ? atoi(e) : 1; }();
    return v != 0;
}

}  // namespace

// attention_split.hip
int fgt_attention_split(const fgt_attn_desc* dd, const void* Q, const void* K, const void* V, const void* KG, const void* VG, float* O,
                        int n_q, int n_k, int n_loc, int zh, int zw, int gh, int gw, int problems, float scale_log2e, hipStream_t s);

extern "C" int fgt_attention(const fgt_attn_desc* dd, const void* Qv, const void* Kv, const void* Vv,
                             const void* KGv, const void* VGv, float* O, void* stream) {
    FGT_REQUIRE(dd && Qv && Kv && Vv && O, "fgt_attention: null pointer");
    const float *Q = static_cast<const float*>(Qv), *K = static_cast<const float*>(Kv), *V = static_cast<const float*>(Vv);
    const float *KG = static_cast<const float*>(KGv), *VG = static_cast<const float*>(VGv);
    AttnP p;
    p.d = *dd;
    const fgt_attn_desc& d = p.d;
    FGT_REQUIRE(d.heads > 0 && d.nh > 0 && d.nw > 0 && d.b > 0 && d.t > 0, "fgt_attention: bad sizes");
    FGT_REQUIRE(d.in_split >= 0 && d.in_split <= 2, "fgt_attention: in_split must be 0, 1 or 2");
    FGT_REQUIRE(d.ldq % 4 == 0 && d.ldk % 4 == 0 && d.ldv % 4 == 0 && d.ldo % 4 == 0 && d.qoff % 4 == 0 &&
                d.koff % 4 == 0 && d.voff % 4 == 0, "fgt_attention: strides/offsets must be multiples of 4 floats");
    FGT_REQUIRE((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)KG | (uintptr_t)VG) & 15) == 0,
                "fgt_attention: pointers must be 16-byte aligned");
    p.Q = Q; p.K = K; p.V = V; p.KG = KG; p.VG = VG; p.O = O;
    p.scale_log2e = 1.4426950408889634f / sqrtf((float)HD);
    p.zh = p.zw = p.gh = p.gw = 0;
    int problems;
    if (d.mode == 0) {
        FGT_REQUIRE(d.group > 0 && d.nh % d.group == 0 && d.nw % d.group == 0, "fgt_attention: grid %dx%d not divisible into %d zones", d.nh, d.nw, d.group);
        p.zh = d.nh / d.group; p.zw = d.nw / d.group;
        FGT_REQUIRE(d.tq >= 0 && d.tq <= d.t, "fgt_attention: tq %d outside [0, t = %d]", d.tq, d.t);
        p.n_k = p.n_loc = d.t * p.zh * p.zw;
        p.n_q = (d.tq > 0 ? d.tq : d.t) * p.zh * p.zw;        // zone tokens are ordered (frame, y, x): the queried frames are a prefix
        problems = d.b * d.group * d.group * d.heads;
    } else if (d.mode == 1) {
        FGT_REQUIRE(d.ws > 0 && d.nh % d.ws == 0 && d.nw % d.ws == 0 && d.h <= d.nh && d.w <= d.nw && d.h > 0 && d.w > 0, "fgt_attention: bad window geometry");
        FGT_REQUIRE(d.n_global >= 0 && (d.n_global == 0 || (KG && VG && d.ldg_k % 4 == 0 && d.ldg_v % 4 == 0)), "fgt_attention: global tokens missing");
        FGT_REQUIRE(d.compact == 0 || (d.compact == 1 && d.pad_row >= 0), "fgt_attention: compact maps need pad_row >= 0");
        p.gh = d.nh / d.ws; p.gw = d.nw / d.ws;
        p.n_q = p.n_loc = d.ws * d.ws;
        p.n_k = p.n_loc + d.n_global;
        problems = d.b * d.t * p.gh * p.gw * d.heads;
    } else {
        fgt_set_error("fgt_attention: unknown mode %d", d.mode);
        return FGT_EINVAL;
    }
    FGT_REQUIRE(problems <= 65535, "fgt_attention: too many problems (%d) for grid.y", problems);
    hipStream_t s = (hipStream_t)stream;
    FGT_REQUIRE(d.precision == 0 || d.precision == 1 || (d.precision == FGT_PREC_F16 && d.in_split == 2), "fgt_attention: unknown precision %d (FGT_PREC_F16 needs in_split = 2)", d.precision);
    FGT_REQUIRE(d.out_split == 0 || (d.out_split == 1 && ((d.pso > 0 && d.pso % 4 == 0) || (d.pso == -1 && d.in_split == 2))), "fgt_attention: bad out_split / pso");
    // unique-byte floor: every Q, K, V row of the maps and every global token once, the output once (4 B per value)
    const double rows_in = (double)d.b * d.t * ((d.mode == 1 && d.compact) ? (double)d.h * d.w : (double)d.nh * d.nw), cc = (double)d.heads * HD;
    const double q_frac = (d.mode == 0 && d.tq > 0) ? (double)d.tq / d.t : 1.0;     // Q and O rows that exist
    const double attn_bytes = (d.in_split == 2 ? 2.0 : 4.0) * cc * ((2.0 + q_frac) * rows_in + (d.mode == 1 ? 2.0 * d.b * d.t * d.n_global + (double)d.b * d.t * d.h * d.w : q_frac * rows_in));
    const int prof = fgt_prof_begin(d.mode == 0 ? FGT_PROF_ATTN_TEMPORAL : FGT_PROF_ATTN_SPATIAL,
                                    4.0 * (double)p.n_q * p.n_k * HD * problems, attn_bytes, s);
    if (d.in_split) {
        const int rc = fgt_attention_split(dd, Qv, Kv, Vv, KGv, VGv, O, p.n_q, p.n_k, p.n_loc, p.zh, p.zw, p.gh, p.gw, problems, p.scale_log2e, s);
        fgt_prof_end(prof, s);
        return rc;
    }
    if (p.n_q <= 64) {
        dim3 grid(cdiv(p.n_q, 64), problems);
        if (d.precision == 0) hipLaunchKernelGGL((attn_kernel<2>), grid, dim3(128), 0, s, p);
        else hipLaunchKernelGGL((attn_bf16x3_kernel<2, false>), grid, dim3(128), 0, s, p);
    } else {
        dim3 grid(cdiv(p.n_q, 128), problems);
        if (d.precision == 0) hipLaunchKernelGGL((attn_kernel<4>), grid, dim3(256), 0, s, p);
        else if (attn_nw8() && p.n_q >= 2048) {   // long zones: 8 wavefronts share each K/V tile (345 vs 375 us at t = 17)
            dim3 grid8(cdiv(p.n_q, 256), problems);
            hipLaunchKernelGGL((attn_bf16x3_kernel<8, true>), grid8, dim3(512), 0, s, p);
        } else hipLaunchKernelGGL((attn_bf16x3_kernel<4, true>), grid, dim3(256), 0, s, p);
    }
    fgt_prof_end(prof, s);
    return fgt_check_launch("attn_kernel");
}
