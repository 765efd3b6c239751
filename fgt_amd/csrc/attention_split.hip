// bf16x3 flash attention on PRE-SPLIT operands: Q, K, V arrive as hi/lo bf16 planes (written once by the projection GEMM's epilogue,
// fgt_conv_desc.out_split), K and V tiles are streamed global -> LDS by LDS-DMA, and V is consumed in its natural [key][d] layout
// through gfx950's transposing LDS read.  Same arithmetic family as attn_bf16x3_kernel (attention.hip): every product is three
// v_mfma_f32_32x32x16_bf16 on hi/lo operands with fp32 accumulation, base-2 online softmax, both contractions issued "swapped" so a
// lane keeps its query from QK^T to PV — but no conversion work is left in the kernel:
//   * attn_bf16x3_kernel re-split the K and V tiles of a zone in EVERY 256-query workgroup (12 times per zone at t = 17) through
//     registers (16 global loads + ~56 VALU + 4 LDS stores per thread and tile) and transposed V on the way;
//   * here a tile is 4 LDS-DMA instructions per wavefront (K hi, K lo, V hi, V lo: four key rows of 256 bytes each), issued one tile
//     ahead into the other LDS stage, so the copy of tile i+1 runs under the MFMAs of tile i with ONE barrier per tile.
// The one arithmetic difference to attn_bf16x3_kernel: Q is stored unscaled, so log2(e)/sqrt(d) multiplies the fp32 scores after the
// contraction (the reference's order: attention_base.py:17-18 scales QK^T) instead of the fp32 queries before their split.
//
// LDS image of a stage: [K hi | K lo | V hi | V lo], each 32 key rows x 128 bf16 (256 bytes).  DMA writes are lane-linear (16 bytes per
// lane, 4 rows per instruction), so the bank swizzles are applied on the SOURCE side and undone by the readers:
//   K  (ds_read_b128, rows = lanes):       16-byte chunk c of row r sits in slot  c ^ (r & 15)
//   V  (ds_read_b64_tr_b16, 4 rows x 32 B): 32-byte segment s of row r sits in segment  s ^ ((r & 3) << 1)
// ds_read_b64_tr_b16 (probed on the MI355X, tools/micro/tr_probe.hip): within a 16-lane group, lane i receives element (i & 3) of the
// 8-byte words read by lanes (i >> 2) + 4k, k = 0..3.  With lane j pointing at V[key0 + (j >> 2)][d0 + 4 (j & 3)] the group reads a
// [4 keys][16 d] block and lane i gets V[key0 .. key0+3][d0 + i]: the 4 consecutive keys of one d that half an MFMA operand needs.
// The PV operand of this kernel takes its 8 keys as two such runs: key(e = 8ks + j, h) = 16ks + 4h + (j & 3) + 8 (j >> 2).
//
// H = true (FGT_PREC_F16, fgt_attn_desc.in_split = 2): Q, K, V are ONE fp16 plane each (f16_rne of the projection, written by the GEMM
// epilogue with pso = -1), P is rounded to fp16 in registers, every product is ONE v_mfma_f32_32x32x16_f16.  Same tiles, swizzles,
// transposing V reads and softmax; a stage is [K | V] (half the LDS-DMA pieces and LDS reads, a third of the MFMAs).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int HD = 128, KT = 32;
constexpr int PLANE = KT * HD * 2;        // bytes of one 32 x 128 bf16 plane
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));   // fp16 mode: a stage is [K | V] instead of [K hi | K lo | V hi | V lo]
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct AttnS {
    fgt_attn_desc d;
    const __bf16 *Q, *K, *V, *KG, *VG;    // hi planes; lo planes ps* elements further
    float* O;
    long psq, psk, psv, psgk, psgv;
    int n_q, n_k, zh, zw, gh, gw, n_loc;
    float scale_log2e;
    int gx, gy, per_xcd;                  // work grid (query blocks x problems) and, with the XCD-aware order, work items per XCD (else 0)
};


struct Prob { int frame0, zi, zj, hd; };

// mode 1 with desc.compact: row of the Q / K / V maps that padded-grid pixel `pix` (index into [bt, nh, nw]) reads
__device__ __forceinline__ long attn_map_row(const fgt_attn_desc& d, long pix) {
    if (d.mode != 1 || !d.compact) return pix;
    const int plane = d.nh * d.nw;
    const int fr = (int)(pix / plane), rem = (int)(pix - (long)fr * plane);
    const int y = rem / d.nw, x = rem - y * d.nw;
    return (y < d.h && x < d.w) ? ((long)fr * d.h + y) * d.w + x : (long)d.pad_row;
}


__device__ __forceinline__ int local_pix(const AttnS& p, const Prob& pr, int n) {
    const fgt_attn_desc& d = p.d;
    if (d.mode == 0) {
        const int zsz = p.zh * p.zw;
        const int tt = n / zsz, rem = n - tt * zsz;
        const int i = rem / p.zw, j = rem - i * p.zw;
        return ((pr.frame0 + tt) * d.nh + pr.zi * p.zh + i) * d.nw + pr.zj * p.zw + j;
    }
    const int a = n / d.ws, b = n - a * d.ws;
    return (pr.frame0 * d.nh + pr.zi * d.ws + a) * d.nw + pr.zj * d.ws + b;
}

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const f32x2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    const f32x2 l = {a - __builtin_bit_cast(float, hi << 16), b - __builtin_bit_cast(float, hi & 0xFFFF0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(l, bf16x2));
}

__device__ __forceinline__ unsigned half2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

struct U4 { unsigned x, y, z, w; };
__device__ __forceinline__ bf16x8 as_bf16x8(unsigned a, unsigned b, unsigned c, unsigned d) {
    const U4 u = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, u);
}

struct S8 { s16x4 a, b; };
// Two transposing reads: keys run 0 (j = 0..3) and run 1 (j = 4..7) of this lane's d.  Issued as INLINE ASSEMBLY on purpose: behind the
// __builtin_amdgcn_ds_read_tr16_b64 intrinsic the compiler cannot tell that the read does not alias the LDS-DMA copies in flight and puts an
// `s_waitcnt vmcnt(0)` in front of it — in the middle of every tile, so the K / V tile requested at the top of the tile had to land before its
// PV product: the loop ran at one global -> LDS round trip per tile whatever the rest of it did (rounds 1-2; found in the ISA after halving
// the VALU work, batching the LDS reads and changing the occupancy had all left the kernel's time unchanged).  The reads return through
// lgkmcnt, which the compiler does not track for inline assembly: tr_wait() is the matching wait and ties the registers it covers.
__device__ __forceinline__ bf16x8 tr_pair(const char* lds_lo_run, const char* lds_hi_run) {
    S8 r;
    const unsigned a0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)lds_lo_run;
    const unsigned a1 = (unsigned)(unsigned long)(__attribute__((address_space(3))) const char*)lds_hi_run;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.a) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r.b) : "v"(a1));
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ void tr_wait(bf16x8& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
__device__ __forceinline__ void tr_wait(bf16x8& a, bf16x8& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void tr_wait(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

// NS: depth of the LDS stage ring.  A K / V tile is requested NS - 1 tiles before it is used.  The 8-wavefront instances (long temporal
// zones) run ONE workgroup per CU (164-228 registers per lane), so a tile of the double-buffered loop took exactly one global -> LDS round
// trip (~1 us under load: the loop was latency-bound — halving its VALU instructions, batching its LDS reads and the 4-wavefront instance at
// three workgroups per CU all left its time unchanged); with the LDS of the whole CU to themselves they keep NS - 1 = 3 tiles in flight.
// PF (fp16, NS = 4 only; FGT_ATTN_PREFETCH=1, off by default): operand prefetch one phase ahead in registers, the lever the ablations seemed to
// point at (profiles/r02_run12_attn_ablate.txt).  The V fragments of a tile are requested before its softmax, the K fragments of tile i+1
// under the PV MFMAs of tile i; for that, tile i+1 has landed when tile i starts (two tiles ahead of the MFMAs instead of three).
// Measured with the last GPU seconds of round 2 (tools/attn_prefetch_check.py, profiles/r02_run12_attn_prefetch_check.txt): bit-identical to the
// default kernel in all four cases; 0...5 % SLOWER as first built (b = 8, t = 17: 1 142 -> 1 200 us), **8 % faster** once the timeline
// (tools/attn_trace.py) had shown where a tile's cycles go and the LDS-DMA issue was moved behind its QK^T MFMAs (1 196 -> 1 095 us).  Off by
// default: the full suite has not run on it; round 3 starts here.
// ST (round 4; bf16x3, 8 wavefronts, NS = 4): the two wavefronts of a SIMD run HALF A TILE OUT OF PHASE.  In the plain loop all eight wavefronts
// cross one barrier per tile and then do the same thing at the same time: QK^T (24 MFMAs each), softmax (VALU, matrix pipe idle), PV (24 MFMAs);
// the softmax of both wavefronts of a SIMD coincides, nobody feeds the pipe meanwhile.  With ST wavefronts 4-7 (one per SIMD) carry the PV
// product of tile i-1 into the interval of tile i: G0: | QK(i) softmax(i) PV(i) |, G1: | PV(i-1) QK(i) softmax(i) | ("|" = the tile's barrier):
// G0's softmax runs beside G1's QK^T, G1's beside G0's PV.  The values and their order per wavefront are unchanged (bit-identical); what
// changes is the lifetime of a stage: tile i-1 is read during interval i, so the request issued in interval i is tile i+2 (two tiles in
// flight on the four stages instead of three).
// VB (round 4; bf16x3): the V fragments of a k-half (4 d-blocks x hi / lo = 16 transposing reads) are requested TOGETHER and retired with
// counted lgkmcnt waits (12 / 8 / 4 / 0: LDS returns in order) in front of each d-block's MFMA triple.  The round-2/3 loop requested one
// d-block's four reads and waited lgkmcnt(0) right behind them: eight exposed LDS round trips per tile and wavefront — the PV segment of
// the timeline (profiles/r02_run12_attn_trace.txt) took 1 824 cycles for 768 cycles of MFMAs.  Same values, same order per accumulator.
template <int NW, bool H, bool TEMPORAL, int NS, bool PF = false, bool ST = false, bool VB = false>
__global__ void __launch_bounds__(NW * 64, 2) attn_split_kernel(const AttnS p) {
    constexpr int NT = NW * 64;
    static_assert(NS >= 2 && NS <= 4, "stage ring depth");
    static_assert(!ST || (!H && !PF && NW == 8 && NS == 4), "staggered schedule: the bf16x3 instance on 8 wavefronts and 4 stages");
    static_assert(!PF || (H && NS == 4), "fragment prefetch: fp16 instance on the 4-stage ring");
    constexpr int NPL = H ? 2 : 4;               // planes per stage
    constexpr int STAGE = NPL * PLANE;           // (shadows the namespace constant: this instance's stage)
    constexpr int VOFF = (NPL / 2) * PLANE;      // first V plane
    constexpr int PPW = NPL * 8 / NW;            // DMA pieces (4 key rows of one plane) per wavefront and tile
    static_assert((NPL * 8) % NW == 0, "pieces per wavefront");
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [2 stages][STAGE] tiles

    const fgt_attn_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // Work item of this workgroup.  Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2); with the XCD-aware order every
    // XCD walks a CONTIGUOUS range of (problem, query block) items, so the query blocks of a problem — which stream the same K / V tiles at about
    // the same time — share one L2 instead of fetching the tiles into all eight.
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.per_xcd) {
        const int w = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
        if (w >= p.gx * p.gy) return;
        by = w / p.gx;
        bx = w - by * p.gx;
    }
    Prob pr;
    {
        int y = by;
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

    // ---- row addresses of the K / V tiles, per lane.  Piece i of this wavefront is plane i / NR of key row R_r = 4 * ((wave + r NW) & 7) + (lane >> 4),
    // r = i % NR: a lane needs the addresses of NR = 8 / NW key rows per tile.  It keeps them itself and moves them 32 keys on per tile —
    // for the temporal zones without a division: key = (tt * zh + ti) * zw + tj is carried as (tt, ti, tj).  (Round 1-2 kept a per-tile address
    // table in LDS that wavefront 0 filled with two integer divisions per key: ~150 VALU instructions per tile that the other wavefronts
    // issued as well, exec-masked — half of the loop's VALU work in the fp16 instance, which is VALU-bound — plus an LDS hand-over that
    // needed its own wait.)  Keys past the end read the last key's row and are masked in the softmax.
    constexpr int NR = 8 / NW;
    static_assert(8 % NW == 0 && PPW % NR == 0, "key rows per lane");
    const int rsub = lane >> 4, pc = lane & 15;
    constexpr bool temporal = TEMPORAL;               // (a template parameter: the other mode's address arithmetic is compiled out of the loop)
    auto row_ptrs = [&](int key, unsigned long& kr, unsigned long& vr, bool& glob) __attribute__((always_inline)) {          // any mode, with divisions
        glob = key >= p.n_loc;
        if (!glob) {
            const long pix = attn_map_row(d, local_pix(p, pr, key));
            kr = reinterpret_cast<unsigned long>(p.K + pix * d.ldk + d.koff + choff);
            vr = reinterpret_cast<unsigned long>(p.V + pix * d.ldv + d.voff + choff);
        } else {
            const long gr = (long)pr.frame0 * d.n_global + (key - p.n_loc);
            kr = reinterpret_cast<unsigned long>(p.KG + gr * d.ldg_k + choff);
            vr = reinterpret_cast<unsigned long>(p.VG + gr * d.ldg_v + choff);
        }
    };
    unsigned long k_last, v_last;
    bool g_last;
    row_ptrs(p.n_k - 1, k_last, v_last, g_last);
    // temporal zones: pixel of key (tt, ti, tj) = pix0 + (tt * nh + ti) * nw + tj
    const int pix0 = (pr.frame0 * d.nh + pr.zi * p.zh) * d.nw + pr.zj * p.zw;
    const unsigned long kz = reinterpret_cast<unsigned long>(p.K + d.koff + choff), vz = reinterpret_cast<unsigned long>(p.V + d.voff + choff);
    const long ldk2 = 2l * d.ldk, ldv2 = 2l * d.ldv;
    int r_key[NR], r_tt[NR], r_ti[NR], r_tj[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        r_key[r] = ((wave + r * NW) & 7) * 4 + rsub;
        r_tt[r] = r_ti[r] = r_tj[r] = 0;
        if constexpr (temporal) {
            const int zsz = p.zh * p.zw;
            r_tt[r] = r_key[r] / zsz;
            const int rem = r_key[r] - r_tt[r] * zsz;
            r_ti[r] = rem / p.zw;
            r_tj[r] = rem - r_ti[r] * p.zw;
        }
    }
    // this wavefront's DMA pieces of the NEXT tile (the one its row state points at) into stage `slot`; then the rows move 32 keys on
    auto issue_tile = [&](int slot) __attribute__((always_inline)) {
        char* st = smem + slot * STAGE;
        unsigned long rk[NR], rv[NR];
        bool rg[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r_key[r] >= p.n_k) {
                rk[r] = k_last; rv[r] = v_last; rg[r] = g_last;
            } else if constexpr (temporal) {
                const long pix = pix0 + (r_tt[r] * d.nh + r_ti[r]) * d.nw + r_tj[r];
                rk[r] = kz + pix * ldk2; rv[r] = vz + pix * ldv2; rg[r] = false;
            } else {
                row_ptrs(r_key[r], rk[r], rv[r], rg[r]);
            }
        }
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int r = i % NR, plane = i / NR;
            const int grp = (wave + r * NW) & 7;                         // = (wave + i * NW) & 7
            const int R = grp * 4 + rsub;
            const bool isv = plane >= NPL / 2, islo = !H && (plane & 1);
            const int c = isv ? ((((pc >> 1) ^ ((R & 3) << 1)) << 1) | (pc & 1)) : (pc ^ (R & 15));       // logical 16-byte chunk this lane fetches
            const long ps = rg[r] ? (isv ? p.psgv : p.psgk) : (isv ? p.psv : p.psk);                        // global tokens live in their own tensors
            const unsigned long src = (isv ? rv[r] : rk[r]) + (unsigned long)c * 16 + (islo ? (unsigned long)ps * 2 : 0ul);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(st + plane * PLANE + grp * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            r_key[r] += KT;
            if constexpr (temporal) {
                r_tj[r] += KT;
                while (r_tj[r] >= p.zw) { r_tj[r] -= p.zw; ++r_ti[r]; }
                while (r_ti[r] >= p.zh) { r_ti[r] -= p.zh; ++r_tt[r]; }
            }
        }
    };

    // ---- Q rows into registers, already split: step s holds d = 16s + 8h + (0..7) of the hi and the lo plane
    const int qi = bx * (NW * 32) + wave * 32 + l31;
    const int qpix = local_pix(p, pr, min(qi, p.n_q - 1));
    bf16x8 qh[8], ql[H ? 1 : 8];
    {
        const __bf16* qp = p.Q + attn_map_row(d, qpix) * d.ldq + d.qoff + choff + 8 * lh;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            qh[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
            if constexpr (!H) ql[s] = *reinterpret_cast<const bf16x8*>(qp + p.psq + 16 * s);
        }
    }

    f32x16 o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[t][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (p.n_k + KT - 1) / KT;
#pragma unroll
    for (int t = 0; t < NS - 1 - (ST ? 1 : 0); ++t)
        if (t < ntiles) issue_tile(t);
    bf16x8 kf_pf[PF ? 8 : 1];                                               // PF: K fragments of the tile about to be multiplied

    // per-lane LDS offsets of the operand reads (stage-relative)
    const int krow = l31 * 256;                                             // K: row l31, chunk (2 st + lh) ^ (l31 & 15)
    // V (transposing read): group gi = lane >> 4: d block (gi & 1) * 16, key half lh = gi >> 1; lane j = lane & 15 -> key row (j >> 2), 8-byte word (j & 3)
    const int gi = lane >> 4, j16 = lane & 15;
    const int vrow_in = j16 >> 2, vword = j16 & 3;

    if constexpr (PF) {                                                     // (the launcher guarantees ntiles >= 4)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");     // tile 0 has landed (tiles 1, 2 in flight)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int sx = 0; sx < 8; ++sx) kf_pf[sx] = *reinterpret_cast<const bf16x8*>(smem + krow + (((2 * sx + lh) ^ (l31 & 15)) << 4));
    }

    // one key tile; MASKED = the last, partial tile (keys past n_k get -inf scores): peeled out of the loop so that the full tiles carry no
    // compare / select per score
    auto tile_step = [&](const int it, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const int slot = it % NS;
        // this wavefront's pieces of tile `it` have landed; the (up to NS - 2) younger tiles stay in flight across the barrier
        if constexpr (PF) {
            // tile it+1 has landed as well (its K fragments are read during this tile); only tile it+2 may stay in flight
            if (it + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const int ahead = min(NS - 2, ntiles - 1 - it);
            if (NS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
            else if (NS >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                         // ... everyone's have, and tile it-1 is fully consumed
        if constexpr (!PF)
            if (it + NS - 1 < ntiles) issue_tile((it + NS - 1) % NS);   // its stage held tile it-1; streams under the MFMAs of NS - 1 tiles
        const char* st = smem + slot * STAGE;
        const int k0 = it * KT;

        // ---- S^T tile = K . Q^T
        f32x16 s;
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        bf16x8 vf_pf[PF ? 8 : 1];
        if constexpr (PF) {
            // K fragments were requested during the previous tile's PV product; V fragments of this tile are requested now, ahead of the softmax
#pragma unroll
            for (int sx = 0; sx < 8; ++sx)
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf_pf[sx]), __builtin_bit_cast(f16x8, qh[sx]), s, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int seg = (t * 2 + (gi & 1)) ^ (vrow_in << 1);
                    const int r0 = 16 * ks + 4 * (gi >> 1) + vrow_in;
                    const int a0 = r0 * 256 + seg * 32 + vword * 8, a1 = a0 + 8 * 256;
                    vf_pf[ks * 4 + t] = tr_pair(st + VOFF + a0, st + VOFF + a1);
                }
        }
        if constexpr (H && !PF) {
            // all eight K fragments in flight before the first MFMA (one LDS round trip instead of eight: the per-tile chain of a
            // wavefront is latency-bound, not issue-bound — halving its VALU instructions did not move the kernel)
            bf16x8 kf[8];
#pragma unroll
            for (int sx = 0; sx < 8; ++sx) kf[sx] = *reinterpret_cast<const bf16x8*>(st + krow + (((2 * sx + lh) ^ (l31 & 15)) << 4));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sx = 0; sx < 8; ++sx)
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf[sx]), __builtin_bit_cast(f16x8, qh[sx]), s, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int sx = 0; sx < (H ? 0 : 8); ++sx) {
            const int off = krow + (((2 * sx + lh) ^ (l31 & 15)) << 4);
            const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(st + off);
            if constexpr (H) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_h), __builtin_bit_cast(f16x8, qh[sx]), s, 0, 0, 0);
            } else {
                const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(st + PLANE + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, qh[sx], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, ql[sx], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, qh[sx], s, 0, 0, 0);
            }
        }
        // PF: the LDS-DMAs of a later tile are issued HERE, behind the QK^T MFMAs (their stage held tile it-1: free since the barrier above).  The
        // timeline (tools/attn_trace.py, profiles/r02_run12_attn_trace.txt) shows 540-670 cycles of address arithmetic + DMA issue per tile and
        // wavefront between the barrier and the first K fragment read; with the K fragments already in registers the MFMAs start right behind
        // the barrier and the issue runs while the matrix pipe works: 1 196 -> 1 095 us on the b = 8, t = 17 call (profiles/r02_run12_attn_prefetch_check.txt,
        // second table), bit-identical.  The default kernels keep the issue in front (the same move measured 0...+5 % there).
        if constexpr (PF)
            if (it + NS - 1 < ntiles) issue_tile((it + NS - 1) % NS);
        // ---- online softmax in base 2 on the RAW scores (the scale c = log2(e) / sqrt(d) > 0 commutes with the maximum and is folded into the
        // exponent: p = exp2(s c - m c), one FMA per score); keys of this lane: k0 + (e&3) + 8*(e>>2) + 4*lh, masked in the last tile only
        if constexpr (MASKED) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (k0 + (e & 3) + 8 * (e >> 2) + 4 * lh >= p.n_k) s[e] = -INFINITY;
        }
        float mx = fmaxf(s[0], s[1]);
#pragma unroll
        for (int e = 2; e < 16; e += 2) mx = fmaxf(mx, fmaxf(s[e], s[e + 1]));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float mc = m_new * p.scale_log2e;
        const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run, p.scale_log2e, -mc));
        // exponents and row sum two scores at a time (v_pk_fma_f32 / v_pk_add_f32)
        const f32x2 c2 = {p.scale_log2e, p.scale_log2e}, mc2 = {mc, mc};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
            const f32x2 sv = {s[e], s[e + 1]};
            const f32x2 x = __builtin_elementwise_fma(sv, c2, -mc2);
            const f32x2 pe = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
            s[e] = pe[0];
            s[e + 1] = pe[1];
            ps2 += pe;
        }
        l_run = l_run * alpha + (ps2[0] + ps2[1]);
        m_run = m_new;
        // the running maximum settles after a few tiles: when NO query of this wavefront saw a new one, alpha is exactly 1 for all of
        // them and the 64 multiplies are skipped (bit-identical: x * 1.0f == x)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
        }
        // ---- O^T += V^T . P^T
        if constexpr (PF) {
            tr_wait(vf_pf[0], vf_pf[1], vf_pf[2], vf_pf[3]);
            tr_wait(vf_pf[4], vf_pf[5], vf_pf[6], vf_pf[7]);
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < ntiles) {                                          // K fragments of the next tile: in flight under the MFMAs below
                const char* sn = smem + ((it + 1) % NS) * STAGE;
#pragma unroll
                for (int sx = 0; sx < 8; ++sx) kf_pf[sx] = *reinterpret_cast<const bf16x8*>(sn + krow + (((2 * sx + lh) ^ (l31 & 15)) << 4));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 p_h = as_bf16x8(half2(s[8 * ks + 0], s[8 * ks + 1]), half2(s[8 * ks + 2], s[8 * ks + 3]),
                                             half2(s[8 * ks + 4], s[8 * ks + 5]), half2(s[8 * ks + 6], s[8 * ks + 7]));
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, vf_pf[ks * 4 + t]), __builtin_bit_cast(f16x8, p_h), o[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ks = 0; ks < (PF ? 0 : 2); ++ks) {
            unsigned h0, h1, h2, h3, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
            if constexpr (H) {     // P in [0, 1]: f16_rne
                h0 = half2(s[8 * ks + 0], s[8 * ks + 1]); h1 = half2(s[8 * ks + 2], s[8 * ks + 3]);
                h2 = half2(s[8 * ks + 4], s[8 * ks + 5]); h3 = half2(s[8 * ks + 6], s[8 * ks + 7]);
            } else {
                split2(s[8 * ks + 0], s[8 * ks + 1], h0, l0); split2(s[8 * ks + 2], s[8 * ks + 3], h1, l1);
                split2(s[8 * ks + 4], s[8 * ks + 5], h2, l2); split2(s[8 * ks + 6], s[8 * ks + 7], h3, l3);
            }
            const bf16x8 p_h = as_bf16x8(h0, h1, h2, h3), p_l = as_bf16x8(l0, l1, l2, l3);
            if constexpr (H) {
                bf16x8 vf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int seg = (t * 2 + (gi & 1)) ^ (vrow_in << 1);
                    const int r0 = 16 * ks + 4 * (gi >> 1) + vrow_in;
                    const int a0 = r0 * 256 + seg * 32 + vword * 8, a1 = a0 + 8 * 256;
                    vf[t] = tr_pair(st + VOFF + a0, st + VOFF + a1);
                }
                tr_wait(vf[0], vf[1], vf[2], vf[3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, vf[t]), __builtin_bit_cast(f16x8, p_h), o[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!H && VB) {
                bf16x8 v_h[4], v_l[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int seg = (t * 2 + (gi & 1)) ^ (vrow_in << 1);
                    const int r0 = 16 * ks + 4 * (gi >> 1) + vrow_in;
                    const int a0 = r0 * 256 + seg * 32 + vword * 8, a1 = a0 + 8 * 256;
                    v_h[t] = tr_pair(st + VOFF + a0, st + VOFF + a1);
                    v_l[t] = tr_pair(st + VOFF + PLANE + a0, st + VOFF + PLANE + a1);
                }
                asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(v_h[0]), "+v"(v_l[0]));
                o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_l[0], p_h, o[0], 0, 0, 0);
                o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h[0], p_l, o[0], 0, 0, 0);
                o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h[0], p_h, o[0], 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(v_h[1]), "+v"(v_l[1]));
                o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_l[1], p_h, o[1], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h[1], p_l, o[1], 0, 0, 0);
                o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h[1], p_h, o[1], 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(v_h[2]), "+v"(v_l[2]));
                o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_l[2], p_h, o[2], 0, 0, 0);
                o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h[2], p_l, o[2], 0, 0, 0);
                o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h[2], p_h, o[2], 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v_h[3]), "+v"(v_l[3]));
                o[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_l[3], p_h, o[3], 0, 0, 0);
                o[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h[3], p_l, o[3], 0, 0, 0);
                o[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h[3], p_h, o[3], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < ((H || VB) ? 0 : 4); ++t) {
                // run 0: keys 16ks + 4lh + (0..3); run 1: + 8.  row & 3 = vrow_in for both (16ks, 4lh, 8 are multiples of 4)
                const int seg = (t * 2 + (gi & 1)) ^ (vrow_in << 1);        // swizzled 32-byte segment of d block t*32 + 16 (gi & 1)
                const int r0 = 16 * ks + 4 * (gi >> 1) + vrow_in;
                const int a0 = r0 * 256 + seg * 32 + vword * 8, a1 = a0 + 8 * 256;
                bf16x8 v_h = tr_pair(st + VOFF + a0, st + VOFF + a1);
                if constexpr (H) {
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, v_h), __builtin_bit_cast(f16x8, p_h), o[t], 0, 0, 0);
                } else {
                    bf16x8 v_l = tr_pair(st + VOFF + PLANE + a0, st + VOFF + PLANE + a1);
                    tr_wait(v_h, v_l);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_l, p_h, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h, p_l, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h, p_h, o[t], 0, 0, 0);
                }
            }
        }
    };
    const int nfull = p.n_k / KT;                                           // tiles whose 32 keys all exist
    if constexpr (ST) {
        const bool g1 = wave >= 4;
        f32x16 s;                                                           // scores, then P: G1 keeps P(i-1) across the barrier
        auto qk_softmax = [&](const int it) __attribute__((always_inline)) {
            const char* st = smem + (it % NS) * STAGE;
            const int k0 = it * KT;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
#pragma unroll
            for (int sx = 0; sx < 8; ++sx) {
                const int off = krow + (((2 * sx + lh) ^ (l31 & 15)) << 4);
                const bf16x8 a_h = *reinterpret_cast<const bf16x8*>(st + off);
                const bf16x8 a_l = *reinterpret_cast<const bf16x8*>(st + PLANE + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, qh[sx], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, ql[sx], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, qh[sx], s, 0, 0, 0);
            }
            if (it >= nfull) {                                // the last, partial tile (wave-uniform branch: full tiles carry no compare / select)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (k0 + (e & 3) + 8 * (e >> 2) + 4 * lh >= p.n_k) s[e] = -INFINITY;
            }
            float mx = fmaxf(s[0], s[1]);
#pragma unroll
            for (int e = 2; e < 16; e += 2) mx = fmaxf(mx, fmaxf(s[e], s[e + 1]));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float mc = m_new * p.scale_log2e;
            const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run, p.scale_log2e, -mc));
            const f32x2 c2 = {p.scale_log2e, p.scale_log2e}, mc2 = {mc, mc};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 sv = {s[e], s[e + 1]};
                const f32x2 x = __builtin_elementwise_fma(sv, c2, -mc2);
                const f32x2 pe = f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
                s[e] = pe[0];
                s[e + 1] = pe[1];
                ps2 += pe;
            }
            l_run = l_run * alpha + (ps2[0] + ps2[1]);
            m_run = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[t][e] *= alpha;
            }
        };
        auto pv = [&](const int it) __attribute__((always_inline)) {
            const char* st = smem + (it % NS) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                split2(s[8 * ks + 0], s[8 * ks + 1], h0, l0); split2(s[8 * ks + 2], s[8 * ks + 3], h1, l1);
                split2(s[8 * ks + 4], s[8 * ks + 5], h2, l2); split2(s[8 * ks + 6], s[8 * ks + 7], h3, l3);
                const bf16x8 p_h = as_bf16x8(h0, h1, h2, h3), p_l = as_bf16x8(l0, l1, l2, l3);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int seg = (t * 2 + (gi & 1)) ^ (vrow_in << 1);
                    const int r0 = 16 * ks + 4 * (gi >> 1) + vrow_in;
                    const int a0 = r0 * 256 + seg * 32 + vword * 8, a1 = a0 + 8 * 256;
                    bf16x8 v_h = tr_pair(st + VOFF + a0, st + VOFF + a1);
                    bf16x8 v_l = tr_pair(st + VOFF + PLANE + a0, st + VOFF + PLANE + a1);
                    tr_wait(v_h, v_l);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_l, p_h, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h, p_l, o[t], 0, 0, 0);
                    o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v_h, p_h, o[t], 0, 0, 0);
                }
            }
        };
        // One code site each for QK^T + softmax and for PV: G0 runs  | QK(i) PV(i) |, G1 runs  QK(i) | PV(i)  with its barrier BETWEEN the
        // two — i.e. | PV(i-1) QK(i) | per interval — and one barrier more in front (G0 passes its extra one behind the loop).  Requests: tile
        // i+2 right behind barrier i (its stage held tile i-2: G0 left it in interval i-2, G1 in interval i-1); the two groups' request
        // cursors advance in the same order, G1's one tile ahead of its QK^T.
        if (g1) {
            if (1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // barrier 0
            if (2 < ntiles) issue_tile(2 % NS);
        }
        for (int it = 0; it < ntiles; ++it) {
            if (!g1) {
                // tile `it` has landed (this wavefront's pieces); tile it+1 may stay in flight across the barrier
                if (it + 1 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                 // barrier it
                if (it + 2 < ntiles) issue_tile((it + 2) % NS);
            }
            qk_softmax(it);
            if (g1) {
                // tile it+1 has landed (tile it+2, requested behind barrier it, may stay in flight)
                if (it + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                 // barrier it+1
                if (it + 3 < ntiles) issue_tile((it + 3) % NS);
            }
            pv(it);
        }
        if (!g1) __builtin_amdgcn_s_barrier();                // (G1 passed barrier `ntiles`)
    } else {
    for (int it = 0; it < nfull; ++it) tile_step(it, std::false_type{});
    if (nfull < ntiles) tile_step(nfull, std::true_type{});
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
                    const float4 v = make_float4(o[t][4 * e4 + 0] * inv, o[t][4 * e4 + 1] * inv, o[t][4 * e4 + 2] * inv, o[t][4 * e4 + 3] * inv);
                    if (d.out_split && d.pso < 0) {          // one fp16 plane
                        *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(p.O) + ob + t * 32 + 8 * e4) = fgt_half4(v);
                    } else if (d.out_split) {
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

// FGT_ATTN_XCD=0: plain (query block, problem) grid (A/B measurements)
dim3 attn_grid(AttnS& q, int gx, int gy) {
    static const int xcd = [] { const char* e = getenv("FGT_ATTN_XCD"); return e ? atoi(e) : 1; }();
    q.gx = gx; q.gy = gy;
    q.per_xcd = (xcd && gx > 1) ? cdiv(gx * gy, 8) : 0;
    return q.per_xcd ? dim3(8 * q.per_xcd) : dim3(gx, gy);
}

template <int NW, bool H, bool TEMPORAL>
int launch_mode(const AttnS& p, int problems, hipStream_t s);

// bf16x3, long zones: the staggered schedule (FGT_ATTN_STAGGER=0: the plain loop, A/B measurements); bit-identical
int launch_staggered(const AttnS& p, int problems, hipStream_t s) {
    constexpr int smem = 4 * 4 * PLANE;
    static std::atomic<unsigned long long> lds_set[2];
    AttnS q = p;
    const dim3 grid = attn_grid(q, cdiv(p.n_q, 8 * 32), problems);
    if (p.d.mode == 0) {
        if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&attn_split_kernel<8, false, true, 4, false, true>), smem, lds_set[0], "attn_split")) return rc;
        hipLaunchKernelGGL((attn_split_kernel<8, false, true, 4, false, true>), grid, dim3(8 * 64), smem, s, q);
    } else {
        if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&attn_split_kernel<8, false, false, 4, false, true>), smem, lds_set[1], "attn_split")) return rc;
        hipLaunchKernelGGL((attn_split_kernel<8, false, false, 4, false, true>), grid, dim3(8 * 64), smem, s, q);
    }
    return fgt_check_launch("attn_split_kernel");
}

template <int NW, bool H, bool TEMPORAL>
int launch_mode(const AttnS& p, int problems, hipStream_t s) {
    constexpr int NS = NW == 8 ? 4 : 2;          // long zones (one workgroup per CU): four stages = three tiles in flight
    constexpr int smem = NS * (H ? 2 : 4) * PLANE;
    static_assert(smem <= 160 * 1024, "LDS ring does not fit");
    AttnS q = p;
    const dim3 grid = attn_grid(q, cdiv(p.n_q, NW * 32), problems);
    if constexpr (!H) {
        // FGT_ATTN_VBATCH (default 1): V fragments of a k-half requested together, counted waits (bit-identical; A/B: =0)
        static const int vb = [] { const char* e = getenv("FGT_ATTN_VBATCH"); return e ? atoi(e) : 1; }();
        if (vb) {
            static std::atomic<unsigned long long> lds_set_vb{0};
            if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&attn_split_kernel<NW, H, TEMPORAL, NS, false, false, true>), smem, lds_set_vb, "attn_split")) return rc;
            hipLaunchKernelGGL((attn_split_kernel<NW, H, TEMPORAL, NS, false, false, true>), grid, dim3(NW * 64), smem, s, q);
            return fgt_check_launch("attn_split_kernel");
        }
    }
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&attn_split_kernel<NW, H, TEMPORAL, NS>), smem, lds_set, "attn_split")) return rc;
    hipLaunchKernelGGL((attn_split_kernel<NW, H, TEMPORAL, NS>), grid, dim3(NW * 64), smem, s, q);
    return fgt_check_launch("attn_split_kernel");
}

// fp16, long temporal zones, FGT_ATTN_PREFETCH=1 (explicit variant: bit-identical, measured slower — see the kernel's header comment)
int launch_prefetch(const AttnS& p, int problems, hipStream_t s) {
    constexpr int smem = 4 * 2 * PLANE;
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&attn_split_kernel<8, true, true, 4, true>), smem, lds_set, "attn_split")) return rc;
    AttnS q = p;
    const dim3 grid = attn_grid(q, cdiv(p.n_q, 8 * 32), problems);
    hipLaunchKernelGGL((attn_split_kernel<8, true, true, 4, true>), grid, dim3(8 * 64), smem, s, q);
    return fgt_check_launch("attn_split_kernel");
}

template <int NW, bool H>
int launch(const AttnS& p, int problems, hipStream_t s) {
    return p.d.mode == 0 ? launch_mode<NW, H, true>(p, problems, s) : launch_mode<NW, H, false>(p, problems, s);
}

}  // namespace


// called by fgt_attention (attention.hip) when desc.in_split is set
int fgt_attention_split(const fgt_attn_desc* dd, const void* Q, const void* K, const void* V, const void* KG, const void* VG, float* O,
                        int n_q, int n_k, int n_loc, int zh, int zw, int gh, int gw, int problems, float scale_log2e, hipStream_t s) {
    AttnS p;
    p.d = *dd;
    const fgt_attn_desc& d = p.d;
    const bool h16 = d.in_split == 2;
    FGT_REQUIRE(d.precision == (h16 ? FGT_PREC_F16 : FGT_PREC_BF16X3), "fgt_attention: split inputs need FGT_PREC_BF16X3, fp16 inputs (in_split = 2) FGT_PREC_F16");
    FGT_REQUIRE(d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.qoff % 8 == 0 && d.koff % 8 == 0 && d.voff % 8 == 0,
                "fgt_attention: split / fp16 inputs need strides / offsets that are multiples of 8 elements");
    FGT_REQUIRE(h16 || (d.psq % 8 == 0 && d.psk % 8 == 0 && d.psv % 8 == 0 && d.psq > 0 && d.psk > 0 && d.psv > 0),
                "fgt_attention: split inputs need plane strides that are positive multiples of 8 bf16 elements");
    FGT_REQUIRE(d.n_global == 0 || (d.ldg_k % 8 == 0 && d.ldg_v % 8 == 0 && (h16 || (d.psg_k % 8 == 0 && d.psg_v % 8 == 0 && d.psg_k > 0 && d.psg_v > 0))),
                "fgt_attention: split global tokens need strides / plane strides that are positive multiples of 8");
    FGT_REQUIRE(d.pso >= 0 || h16, "fgt_attention: an fp16 output (pso = -1) needs fp16 inputs");
    p.Q = static_cast<const __bf16*>(Q); p.K = static_cast<const __bf16*>(K); p.V = static_cast<const __bf16*>(V);
    p.KG = static_cast<const __bf16*>(KG); p.VG = static_cast<const __bf16*>(VG); p.O = O;
    p.psq = d.psq; p.psk = d.psk; p.psv = d.psv; p.psgk = d.psg_k; p.psgv = d.psg_v;
    p.n_q = n_q; p.n_k = n_k; p.n_loc = n_loc; p.zh = zh; p.zw = zw; p.gh = gh; p.gw = gw;
    p.scale_log2e = scale_log2e;
    // long zones: 8 wavefronts share each K / V tile (FGT_ATTN_SPLIT_NW=4: A/B switch — the 4-wavefront instance runs three workgroups per CU
    // where the register count of the 8-wavefront one allows a single workgroup)
    static const int nw_long = [] { const char* e = getenv("FGT_ATTN_SPLIT_NW"); return e ? atoi(e) : 8; }();
    const bool big = n_q >= 2048 && nw_long == 8;
    if (h16) {
        static const int prefetch = [] { const char* e = getenv("FGT_ATTN_PREFETCH"); return e ? atoi(e) : 1; }();
        if (n_q <= 64) return launch<2, true>(p, problems, s);
        if (big && prefetch && d.mode == 0 && n_k >= 4 * KT) return launch_prefetch(p, problems, s);
        if (big) return launch<8, true>(p, problems, s);
        return launch<4, true>(p, problems, s);
    }
    if (n_q <= 64) return launch<2, false>(p, problems, s);
    // (FGT_ATTN_STAGGER=1: the staggered schedule — bit-identical, and measured EQUAL to the plain loop: 1 804 vs 1 810 us at b = 8, t = 17,
    //  2 145 vs 2 130 us at t = 18, profiles/r04_run15_attn_stagger_ab.txt — so the plain loop stays the default)
    static const int stagger = [] { const char* e = getenv("FGT_ATTN_STAGGER"); return e ? atoi(e) : 0; }();
    if (big && stagger && n_k >= 3 * KT) return launch_staggered(p, problems, s);
    if (big) return launch<8, false>(p, problems, s);
    return launch<4, false>(p, problems, s);
}
