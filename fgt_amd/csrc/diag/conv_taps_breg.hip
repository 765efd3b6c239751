// DIAGNOSTIC BUILDS ONLY — experiment, measured and NOT adopted (profiles/r03_run8_split_sweep_taps_breg.txt: bit-identical to conv_taps.hip,
// 3-9 % slower on the Cout >= 256 layers, +2...10 % on two decoder layers only): the tap-reusing bf16x3 kernel of conv_taps.hip with the
// WEIGHT fragments loaded straight into registers (tile codes + 300, weights in "fragment order": fgt_conv_desc.w_il = 2).  Neither the
// barriers nor the LDS round trip of the B tile is what bounds conv_taps.hip: the same bytes through plain loads cost the same or more.
//
// conv_taps.hip still moves a 16 KB B (weight) tile through LDS per K-step: 16 of its 22 LDS-DMA instructions, a stage that must be
// released by a barrier after the fragment reads and published by another one after the copy.  Here the weight image is stored the way
// the MFMA consumes it — [kstep][32-channel-out block][hi | lo][k-half][lane] 16 bytes: one wave-wide 16-byte load is 1 KB of consecutive
// memory — so a wavefront fetches its own B fragments with TN * 4 plain global loads per step, one step ahead, into registers.  What is left
// in LDS are the two A buffers of conv_taps.hip (the im2col rows of a (ky, chunk), shared by all kx taps): 37 KB for BM = 128, and ONE
// workgroup barrier per super-step (kw steps) instead of two per step: inside a super-step the wavefronts run free.
// Same products in the same order as conv_taps.hip: bit-identical to it.
#include "../conv_tile.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

constexpr int HALO = 16;

// B fragment loads as inline assembly: hipcc's own s_waitcnt for a tracked load was vmcnt(0) in front of the MFMAs of every second step (LDS-DMA
// requests and register loads share the counter) — it waited for the set requested a few instructions earlier.  The waits are placed by hand.
template <int OFF> __device__ __forceinline__ void gload16(bf16x8& r, const char* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(ptr), "n"(OFF) : "memory");
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

template <int BM, int BN, int WM, int WN, int MINW, int KW>
__global__ void __launch_bounds__(WM* WN * 64, MINW) conv_tapsr_kernel(const ConvP p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int AR = BM + HALO;                        // A rows per plane filled by DMA; row AR is the zero row
    constexpr int APL = (AR + 1) * 64;                   // bytes per A plane
    constexpr int GA = AR / 16;
    constexpr int NPA = 2 * GA;                          // A pieces per (ky, chunk): piece j -> plane j / GA, group j % GA
    constexpr int A_BYTES = 2 * APL;
    constexpr int LDS_BYTES = 2 * A_BYTES;
    constexpr int STAGE = LDS_BYTES / 8;
    constexpr int APW = (NPA + NW - 1) / NW;
    constexpr int ASTEPS = KW - 1;                       // A pieces go out in steps 0 .. KW-2 of the previous super-step
    static_assert(AR % 16 == 0 && TM >= 1 && TN >= 1 && KW >= 3, "tile / wavefront geometry");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const fgt_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    int m_idx, n_idx;
    if (!conv_tile_index(p, m_idx, n_idx)) return;
    const int bm0 = m_idx * BM, bn0 = n_idx * BN, g = blockIdx.y;

    char* const lds = reinterpret_cast<char*>(smem);
    if (tid < 64) reinterpret_cast<float*>(lds + (tid >> 5) * A_BYTES + ((tid >> 4) & 1) * APL + AR * 64)[tid & 15] = 0.f;

    const int W = d.W, H = d.H, dwx = d.dw;
    const int HW = H * W;
    const bool il = d.in_split == 2;
    const int nch0 = p.Cg0 / 32, nch1 = p.Cg1 / 32, nchunk = nch0 + nch1;
    const int nss = d.kh * nchunk;
    const long cstride = il ? 128 : 64;

    // ---- im2col source iterator (as conv_taps.hip)
    auto src_hi = [&](int s) {
        const __bf16* x = reinterpret_cast<const __bf16*>(s ? p.x1 : p.x0);
        const long c0 = s ? (long)d.off1 + (long)g * p.Cg1 : (long)d.off0 + (long)g * p.Cg0;
        return reinterpret_cast<const char*>(x + (il ? 2 * c0 : c0));
    };
    auto src_lo_off = [&](int s) { return il ? 64l : 2 * (s ? p.ps1 : p.ps0); };
    const char* a_hi = src_hi(0);
    const char* a_lo = a_hi + src_lo_off(0);
    int a_ld = d.ld0, a_left = nch0, a_src = 0;
    int a_dy = -d.ph, a_dyW = -d.ph * W, a_o = 0;
    auto a_advance = [&]() {                              // K walks (chunk, ky, kx) as in conv_taps.hip (round 5)
        a_dy += d.dh; a_dyW += d.dh * W;
        if (++a_o < d.kh) return;
        a_o = 0; a_dy = -d.ph; a_dyW = -d.ph * W;
        a_hi += cstride; a_lo += cstride;
        if (--a_left == 0 && a_src == 0 && nch1 > 0) {
            a_src = 1; a_left = nch1; a_ld = d.ld1;
            a_hi = src_hi(1);
            a_lo = a_hi + src_lo_off(1);
        }
    };
    const int lrow = lane >> 2;
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);
    int a_q[APW], a_y[APW];
#pragma unroll
    for (int it = 0; it < APW; ++it) {
        const int j = wave + it * NW;
        const long q = (long)bm0 - d.pw + (j % GA) * 16 + lrow;
        const bool valid = j < NPA && q >= 0 && q < (long)d.N * HW;
        a_q[it] = valid ? (int)q : 0;
        a_y[it] = valid ? (int)(q % HW) / W : -(1 << 30);
    }
    const char* const zp = reinterpret_cast<const char*>(p.zero_page);
    auto issue_A = [&](int it, int ab) {
        const int j = wave + it * NW;
        if (j >= NPA) return 0;
        const int plane = j / GA, grp = j % GA;
        const char* ptr = (plane ? a_lo : a_hi) + 2 * ((long)(a_q[it] + a_dyW) * a_ld + kc * 8);
        const bool ok = (unsigned)(a_y[it] + a_dy) < (unsigned)H;
        glds16(ok ? ptr : zp, lds + ab * A_BYTES + plane * APL + grp * 1024);
        return 1;
    };

    // ---- weights in fragment order: [groups][Kpad/32][Npad/32][hi | lo][k-half][64 lanes] 16 bytes.  This wavefront's TN blocks of a K-step are
    // TN * 4 KB of consecutive memory; K-step order as in conv_taps.hip: kstep(ky, c, kx) = (ky*KW + kx) * nchunk + c
    const long kbytes = (long)(d.Npad / 32) * 4096;      // bytes per K-step
    const char* wq = reinterpret_cast<const char*>(p.w) + (long)g * (d.Kpad / 32) * kbytes + (long)(bn0 / 32 + wn * TN) * 4096 + lane * 16;
    const long dkx = (long)nchunk * kbytes, dchunk = (1 - ((long)d.kh * KW - 1) * nchunk) * kbytes;
    int b_c = 0;
    auto load_B = [&](bf16x8 (&b)[TN][4]) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const char* q = wq + j * 4096;
            gload16<0>(b[j][0], q); gload16<1024>(b[j][1], q); gload16<2048>(b[j][2], q); gload16<3072>(b[j][3], q);
        }
    };
    auto advance_B = [&](bool last_kx) {
        long dlt = dkx;
        if (last_kx && ++b_c == d.kh) { b_c = 0; dlt = dchunk; }
        wq += dlt;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    int Rb[TM], oxp[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        Rb[i] = wm * WTM + i * 32 + l31;
        oxp[i] = (bm0 + Rb[i]) % W - d.pw;
    }

    // ---- prologue: A rows of super-step 0, B fragments of step 0
#pragma unroll
    for (int it = 0; it < APW; ++it) issue_A(it, 0);
    a_advance();
    // B fragments: two register sets, [block][hi k0, hi k1, lo k0, lo k1]; global step n computes from set n & 1 while set (n + 1) & 1 is in flight.
    // KW is odd: the parity of a super-step's first step alternates, so the loop body below is instantiated for both (PAR) and every register
    // index is a compile-time constant (no copies, and the wait for a set is the compiler's s_waitcnt in front of its first MFMA, one step later).
    bf16x8 bq[2][TN][4];
    load_B(bq[0]); advance_B(false);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    auto super_step = [&](auto PARC, int ss) {
        constexpr int PAR = decltype(PARC)::value;
        const bool last = ss + 1 == nss;
        const unsigned Ab = (unsigned)((ss & 1) * A_BYTES);
        static_for<KW>([&](auto KX) {
            constexpr int kx = decltype(KX)::value;
            constexpr int cur = (PAR * KW + kx) & 1;
            // requests: this step's share of the next super-step's A rows, then the next step's B fragments
            int na = 0;                                   // LDS-DMA requests of this wavefront in this step (wave-uniform)
            if constexpr (kx < ASTEPS) {
                if (!last) {
#pragma unroll
                    for (int it = 0; it < APW; ++it)
                        if (it % ASTEPS == kx) na += issue_A(it, (ss + 1) & 1);
                }
            }
            const bool more = kx + 1 < KW || !last;
            if (more) { load_B(bq[cur ^ 1]); advance_B((kx + 1) % KW == KW - 1); }
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 ah[2][TM], al[2][TM];
            {
                int sh = kx * dwx;
                asm volatile("" : "+s"(sh));                     // (fragment addresses recomputed per step: hoisted for all taps they spill at 128 registers)
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const bool xin = (unsigned)(oxp[i] + sh) < (unsigned)W;
                    const int R = xin ? Rb[i] + sh : AR;
                    const unsigned a0 = Ab + (unsigned)R * 64u + (unsigned)(((R >> 2) & 3) ^ lh) * 16u;
                    const unsigned a1 = a0 ^ 32u;
                    ah[0][i] = *reinterpret_cast<const bf16x8*>(lds + a0);
                    al[0][i] = *reinterpret_cast<const bf16x8*>(lds + a0 + APL);
                    ah[1][i] = *reinterpret_cast<const bf16x8*>(lds + a1);
                    al[1][i] = *reinterpret_cast<const bf16x8*>(lds + a1 + APL);
                }
            }
            __builtin_amdgcn_sched_barrier(0);            // all fragment reads of the step ahead of its MFMAs
            // this step's B set (requested one step ago) has landed: everything older than this step's own requests
            constexpr int MAXA = kx < ASTEPS ? (APW + ASTEPS - 1 - kx) / ASTEPS : 0;
            if (more) wait_vmcnt_plus<4 * TN, MAXA>(na); else wait_vmcnt_plus<0, MAXA>(na);
            __builtin_amdgcn_sched_barrier(0);
            // same products as conv_split.hip (lo*hi, hi*lo, hi*hi per k-half)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks][i], bq[cur][j][ks], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bq[cur][j][2 + ks], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks][i], bq[cur][j][ks], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // super-step boundary: this wavefront's A pieces of the next super-step have landed (so has the next step's B set: it was requested
        // a whole step ago), every wavefront has read this super-step's A buffer
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        a_advance();
    };
    int ss = 0;
    for (; ss + 1 < nss; ss += 2) {
        super_step(std::integral_constant<int, 0>{}, ss);
        super_step(std::integral_constant<int, 1>{}, ss + 1);
    }
    if (ss < nss) super_step(std::integral_constant<int, 0>{}, ss);

    conv_epilogue<BM, BN, WM, WN, STAGE, TM, TN>(p, acc, smem, bm0, bn0, g);
}

template <int BM, int BN, int WM, int WN, int MINW, int KW>
int launch_kw(const ConvP& p, hipStream_t s) {
    constexpr int NT = WM * WN * 64;
    constexpr size_t smem = (size_t)2 * (2 * (BM + HALO + 1) * 64);
    static std::atomic<unsigned long long> lds_set{0};
    if (int rc = fgt_set_max_lds(reinterpret_cast<const void*>(&conv_tapsr_kernel<BM, BN, WM, WN, MINW, KW>), (int)smem, lds_set, "conv_taps_breg")) return rc;
    ConvP q = p;
    q.mtiles = cdiv(p.M, BM);
    q.ntiles = cdiv(p.Cout_g, BN);
    q.mchunk = cdiv(q.mtiles, 8);
    dim3 grid(q.xcd_swizzle ? 8 * q.mchunk * q.ntiles : q.mtiles * q.ntiles, p.d.groups);
    hipLaunchKernelGGL((conv_tapsr_kernel<BM, BN, WM, WN, MINW, KW>), grid, dim3(NT), smem, s, q);
    return fgt_check_launch("conv_taps_breg");
}

template <int BM, int BN, int WM, int WN, int MINW>
int launch(const ConvP& p, hipStream_t s) {
    switch (p.d.kw) {
        case 3: return launch_kw<BM, BN, WM, WN, MINW, 3>(p, s);
        case 5: return launch_kw<BM, BN, WM, WN, MINW, 5>(p, s);
        case 7: return launch_kw<BM, BN, WM, WN, MINW, 7>(p, s);
        default: fgt_set_error("fgt_conv2d: the tap-reusing kernel is built for kw = 3, 5, 7 (got %d)", p.d.kw); return FGT_EINVAL;
    }
}

}  // namespace

int fgt_conv_taps_breg_launch(int tile, const ConvP& p, hipStream_t s) {
    switch (tile) {
        case FGT_TILE_128x128x8: return launch<128, 128, 2, 4, 4>(p, s);
        case FGT_TILE_128x128: return launch<128, 128, 2, 2, 2>(p, s);
        case FGT_TILE_128x64: return launch<128, 64, 2, 2, 4>(p, s);
        case FGT_TILE_64x64: return launch<64, 64, 2, 2, 4>(p, s);
        default: fgt_set_error("fgt_conv2d: tile %d is not built for the tap-reusing kernel with register-fed weights", tile + 300); return FGT_EINVAL;
    }
}
