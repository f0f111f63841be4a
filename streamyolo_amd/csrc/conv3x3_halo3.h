// conv3x3_halo3.h — third generation of the 3x3 stride-1 window-in-LDS kernel (forward and data gradient), round 5.
//
// Replaces the same reference code as conv3x3_halo.h (yolox BaseConv's Conv2d of the Bottleneck / head-tower 3x3 layers —
// exps/model/darknet.py:118-165 via CSPLayer, dfp_pafpn.py:33-81, tal_head.py:55-104 — and cuDNN backward-data).
//
// What the counters of the second generation said (profiles/r05, stages a-d; tile 104 = 128 ch x 5 rows, 256->256 @38x60 x 8):
//   * LDS is NOT the wall: SQ_LDS_IDX_ACTIVE = 26 % of the kernel, SQ_LDS_BANK_CONFLICT = 3.7 % of that (the XOR swizzle holds);
//   * a wave is MFMA-busy for 48 % of its life, parked in s_waitcnt / s_barrier for 30 %, and issues 4.3 other instructions per
//     MFMA in the main loop — 1.6 of them VALU address arithmetic for the ds_read_b128 in front of every MFMA; with the LDS
//     fragment reads (and their address / wait instructions) ablated the main loop loses 35 % of its time, with the operand loads
//     ablated another 20 %; a second wave per SIMD (K groups, tiles 120 / 125-127 of stage c) runs in lockstep and buys nothing;
//   * +2 scalar instructions per MFMA (the ablation branches of stage d) cost the loop +40 %: with one wave per SIMD the loop
//     is bound by its in-order instruction stream, not by a memory level.
// So this generation removes instructions from the stream instead of adding waves:
//   * LDS addresses as IMMEDIATES.  The window keeps its 34-row pitch, but the XOR swizzle of a row's 16-byte chunks is keyed on
//     the row's COLUMN index hx = row % 34 instead of the linear row — then address(y, hx, chunk) = y * 34 * 64 + g(hx, chunk), and
//     the 18 x TP fragment reads of a slab are six per-lane base registers (kw x k-half) + compile-time offsets ((tile row + kh) *
//     2176 B): no VALU per read, 6 address registers instead of 9 x TP.  Conflict freedom is unchanged (a b128 lane group reads 16
//     columns of ONE window row: four 4-column runs 0 / 12 / 20 / 24 apart = row quads 0 / 3 / 1 / 2 mod 4).
//   * forward / data gradient as a TEMPLATE parameter (the tap -> window offset map is compile time, as the immediates need);
//   * the last slab PEELED: no `live` selects on the fragment loads / DMA pieces of the slab after the last one (they are not
//     issued at all — no dead loads for the final wait to drain);
//   * DMA pieces take the slab's channel offset through the SGPR offset operand of buffer_load ... lds (no VALU add per piece);
//     the halo-buffer parity is two v_add per (kw, k-half) base and slab instead of one per read;
//   * the stream is PLACED BY HAND (one fragment read and at most one global load behind every MFMA, scheduling fences), one explicit
//     LGKM wait per step instead of one per MFMA, and the fragment ring runs on ACROSS the slab boundary: the last steps of a slab
//     prefetch the first steps of the next window, the slab rendezvous stands in front of them (see the kernel).
// Main loop at 256->256 @38x60 x 8 (720 MFMAs per wave): 18.8 us (tile 104) -> 17.2 (immediates) -> 16.4 us (placed stream, cross-slab
// prefetch), instructions per MFMA besides the MFMA 4.3 -> 2.2; kernels +7-17 % in tools/conv_probe.py (profiles/r05 stages e-h).
// Tile = 4 waves x (32 ch x TP rows x 32 px) as tiles 117 / 107 / 118 / 104 -> 101 / 100 / 98 / 109 (the grids that fit the layer sizes).
// Epilogue: conv_epilogue of conv_igemm_impl.h through the TilePixels mapper.
#pragma once
#include "conv3x3_halo.h"

namespace sy_conv {

// LDS-DMA with a wave-uniform byte offset in the instruction's SGPR-offset operand (range check on the VGPR offset alone:
// 0xFFFFFFFF stays out of range whatever the scalar adds)
#ifdef SY_EMU
static inline void sy_glds16_buf_at_s(const sy_buffer& b, unsigned voff, unsigned soff, sy_lds_base_t base, unsigned off) {
    sy_glds16_buf(b, voff == 0xFFFFFFFFu ? 0xFFFFFFFFu : voff + soff, base + off);
}
#else
__device__ __forceinline__ void sy_glds16_buf_at_s(const sy_buffer& b, unsigned voff, unsigned soff, sy_lds_base_t base, unsigned off) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(base + off);
    asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(b), "s"(soff), "{m0}"(dst) : "memory");
}
#endif

// explicit LGKM wait (LDS reads): at most N outstanding.  A real S_WAITCNT (not inline asm), so the compiler's own wait insertion
// sees it and drops the per-MFMA waits it covers.
#ifdef SY_EMU
template <int N> static inline void sy_wait_lgkm() {}
#else
template <int N> __device__ __forceinline__ void sy_wait_lgkm() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    __builtin_amdgcn_s_waitcnt(0xC07F | (N << 8));      // vmcnt = 63, expcnt = 7: untouched
}
#endif

// Loads issued behind the last DMA piece (G = 1 step of tap NI - 1, behind its second MFMA) up to the start of step SB of the same
// slab: k-half 1 of that tap + two fragment loads per later G = 1 step
constexpr int halo3_loads_behind_pieces(int NI, int SB) {
    int n = 1;
    for (int s = 2 * NI; s < SB; ++s) n += (s & 1) ? 2 : 0;
    return n;
}

// BD: pixel-fragment ring — the fragments of step S + BD - 1 are requested while step S multiplies.
// ILV = 1: this generation's stream; ILV = 0 keeps the compiler-ordered step of the first halo3 measurement (profiles/r05 stage e).
// WC: waves over the channel blocks (4 = the 128-channel tiles; 2 = 64 channels x two groups of TP tile rows, for the 64-channel
// layers of the 150x240 maps, where the 128-channel tile leaves two of its four waves without channels)
template <typename T, int TP, int DGRAD, int BD, int ILV = 1, int WC = 4>
__global__ __launch_bounds__(256, 2) void conv3x3_halo3_kernel(ConvArgs p) {
    SY_TL_BEGIN(2 + (DGRAD ? 32 : 0));
    constexpr int NW = 4, WP = NW / WC;
    constexpr int EPC = T::kEPC;
    constexpr int ESZ = 16 / EPC;
    constexpr int BK = 4 * EPC;
    constexpr int TH = WP * TP;
    constexpr int RW = kHaloW;                             // 34 window columns per tile row
    constexpr int HR = (TH + 2) * RW;
    constexpr int NI = ((HR + 15) / 16 + NW - 1) / NW;     // DMA pieces (16 window rows each) per wave and slab
    constexpr int BUF = NW * NI * 16 * 64;
    constexpr int RB = RW * 64;                            // bytes between the windows of consecutive tile rows
    constexpr int PF = BD - 1;                             // prefetch distance in (tap, k-half) steps
    constexpr int SB = 18 - PF;                            // the step whose prefetch is the NEXT slab's step 0: the slab rendezvous sits in front of it
    static_assert(2 * NI - 1 < SB, "the slab's DMA pieces are issued before the rendezvous");
    static_assert(TP * PF <= 15, "outstanding fragment reads fit the 4-bit LGKM counter");
    static_assert(18 % BD == 0, "the fragment ring runs on across slabs: a slab's 18 steps are a whole number of turns");
    static_assert(BUF + (TH + 1) * RB + 34 * 64 < 65536, "fragment offsets are 16-bit immediates");

    SY_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sy_uniform(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;               // channel block, group of TP tile rows
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const sy_block_id bid = sy_xcd_block_id();
    const int tiles_w = (p.Wo + 31) >> 5, tiles_h = (p.Ho + TH - 1) / TH;
    const int tw = bid.y % tiles_w, th_ = (bid.y / tiles_w) % tiles_h, n = bid.y / (tiles_w * tiles_h);
    const int h0 = th_ * TH, w0 = tw * 32;

    // ---- DMA assignment: piece i of this wave fills window rows [16 (wave + 4 i), + 16); lane -> (row, physical 16-byte slot);
    //      the slot of logical chunk c of a row in window column hx is c ^ ((hx >> 2) & 3)
    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    unsigned voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = (wave + i * NW) * 16 + (lane >> 2);
        const int hy = r / RW, hx = r - hy * RW;
        const int h = h0 - 1 + hy, w = w0 - 1 + hx;
        const int chunk = (lane & 3) ^ ((hx >> 2) & 3);
        const bool ok = r < HR && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W && !(p.ablate & 1);
        voff[i] = ok ? (unsigned)((((long long)n * p.xbs + ((long long)h * p.W + w) * p.ldx) + chunk * EPC) * ESZ) : 0xFFFFFFFFu;
    }
    const sy_lds_base_t lds0 = sy_lds_base(smem);
    const int ncs = sy_uniform(p.Cin / BK);
    auto issue_piece = [&](auto i_, int j) {               // piece I of slab j into buffer j & 1
        constexpr int I = decltype(i_)::value;
        sy_glds16_buf_at_s(bufx, voff[I], (unsigned)(j * BK * ESZ), lds0, (unsigned)((j & 1) * BUF + (wave + I * NW) * 1024));
    };

    // ---- weight fragments [ct][slab][tap][g][64 lanes][16 B]: the wave's 32 channels, one per-lane offset register per k-half
    const sy_buffer buff = sy_make_buffer(p.wfrag, p.wfrag_extent);
    const int ntile32 = (p.Cout + 31) / 32;
    const int ct = bid.x * WC + wc;
    const unsigned foff = (ct < ntile32 && !(p.ablate & 2)) ? (unsigned)((((long long)ct * ncs * 9) * 128 + lane) * 16) : 0xFFFFFFFFu;
    const unsigned foff1 = foff == 0xFFFFFFFFu ? 0xFFFFFFFFu : foff + 1024u;      // k-half 1
    uint4 fr[9][2];

    f32x16 acc[1][TP];
#pragma unroll
    for (int u = 0; u < TP; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][u][r] = 0.0f;

    // ---- fragment read bases: window column hx = l31 + ox (ox = the tap's column offset), k-half G; everything else is an immediate
    unsigned ba[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int hx = l31 + (DGRAD ? 2 - kw : kw);
#pragma unroll
        for (int g = 0; g < 2; ++g) ba[kw][g] = (unsigned)(wp * TP * RB + hx * 64 + (((g * 2 + half) ^ ((hx >> 2) & 3)) << 4));
    }
    uint4 b[BD][TP];
    auto read_frag = [&](auto s_, auto u_) {               // pixel fragment of step S (mod 18: of the window `ba` points at), tile row U
        constexpr int S = decltype(s_)::value % 18, U = decltype(u_)::value;
        constexpr int TAP = S >> 1, G = S & 1, KH = TAP / 3, KW = TAP % 3;
        constexpr int OY = DGRAD ? 2 - KH : KH;
        b[decltype(s_)::value % BD][U] = *reinterpret_cast<const uint4*>(smem + ba[KW][G] + (unsigned)((U + OY) * RB));
    };

    sy_probe(0);
    sy_static_for<0, NI>([&](auto i_) { issue_piece(i_, 0); });
    sy_static_for<0, 9>([&](auto t_) {                     // in tap order: the compiler's vmcnt waits count on it
        constexpr int TAP = decltype(t_)::value;
        fr[TAP][0] = sy_buffer_load16_s(buff, foff, (unsigned)(TAP * 2048));
        fr[TAP][1] = sy_buffer_load16_s(buff, foff1, (unsigned)(TAP * 2048));
        sy_sched_fence();
    });
    sy_probe(1);
    sy_wait_vmcnt<18>();                          // this wave's pieces of slab 0 (older than the 18 fragment loads)
    sy_barrier();                                 // ... everybody's
    sy_probe(2);
    sy_static_for<0, PF>([&](auto s_) { sy_static_for<0, TP>([&](auto u_) { read_frag(s_, u_); }); });

    // One slab = 18 (tap, k-half) steps of TP MFMAs.  The stream is placed by hand: behind EVERY MFMA exactly one fragment read of the
    // step PF ahead and at most one global load, a scheduling fence after each such group — an in-order wave can use an MFMA's 32-cycle
    // shadow only for what stands in front of the NEXT MFMA in program order (left to itself the compiler gathers a step's reads, two
    // fragment loads and the DMA piece behind one MFMA and issues the other four back to back).  One explicit LGKM wait per step (the
    // step's fragments were requested PF steps ago) instead of the compiler's wait in front of every MFMA.  The steps SB .. 17 prefetch
    // steps 0 .. PF - 1 of the NEXT slab's window, so the rendezvous that publishes that window (my pieces landed; every read of this
    // window complete, for every wave: its buffer is DMA'd again one slab later) stands in front of step SB, not between the slabs.
    auto slab = [&](auto last_, int j) {
        constexpr bool LAST = decltype(last_)::value != 0;
        sy_static_for<0, 18>([&](auto s_) {
            constexpr int S = decltype(s_)::value;
            constexpr int TAP = S >> 1, G = S & 1;
            if constexpr (!LAST && S == SB) {
                sy_wait_vmcnt<halo3_loads_behind_pieces(NI, SB)>();
                sy_wait_lgkm<0>();
                sy_barrier();
                const unsigned flip = (j & 1) ? (unsigned)(-BUF) : (unsigned)BUF;       // the other halo buffer from here on
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) { ba[kw][0] += flip; ba[kw][1] += flip; }
            } else {
                sy_wait_lgkm<(LAST && S + PF > 18) ? TP * (17 - S) : TP * (PF - 1)>();   // this step's fragments are in
            }
            sy_static_for<0, TP>([&](auto u_) {
                constexpr int U = decltype(u_)::value;
                acc[0][U] = sy_mfma_group(T(), fr[TAP][G], b[S % BD][U], acc[0][U]);
                if constexpr (!LAST || S + PF < 18) read_frag(sy_int<S + PF>(), u_);
                if constexpr (G == 1 && !LAST) {
                    const unsigned s_f = (unsigned)(((j + 1) * 9 + TAP) * 2048);
                    if constexpr (U == 0) fr[TAP][0] = sy_buffer_load16_s(buff, foff, s_f);          // k-half 0: its MFMAs are a step behind
                    if constexpr (U == (TP > 2 ? 1 : 0) && TAP < NI) issue_piece(sy_int<TAP>(), j + 1);
                    if constexpr (U == TP - 1) fr[TAP][1] = sy_buffer_load16_s(buff, foff1, s_f);     // k-half 1: behind its last MFMA
                }
                sy_sched_fence();
            });
        });
    };

    for (int j = 0; j < ncs - 1; ++j) slab(sy_int<0>(), j);
    slab(sy_int<1>(), ncs - 1);
    sy_barrier();                                 // every wave is done with the window: the epilogue reuses the LDS
    sy_probe(7);

    SY_LATE_ARGS(ConvArgs, p);
    sy_probe(3);
    int e_bx = bid.x, e_n = n, e_h0 = h0, e_w0 = w0, e_by = bid.y;
    SY_LAUNDER_INT(e_bx); SY_LAUNDER_INT(e_n); SY_LAUNDER_INT(e_h0); SY_LAUNDER_INT(e_w0); SY_LAUNDER_INT(e_by);
    TilePixels mp;
    mp.n = e_n; mp.h0 = e_h0; mp.w0 = e_w0; mp.Ho = p_late.Ho; mp.Wo = p_late.Wo; mp.rep = e_by;
    mp.seg = p_late.seg_M > 0 ? (e_n * p_late.HoWo) / p_late.seg_M : 0;
    // output pixels of the tile outside the image still see valid input through their window: clear them (the BatchNorm
    // statistics of the epilogue sum every accumulator)
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        int n_, rem_;
        if (!mp.map((wp * TP + u) * 32 + l31, n_, rem_)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][u][r] = 0.0f;
        }
    }
    conv_epilogue<T, WC, WP, 1, TP>(p_late, mp, e_bx, acc, smem, tid);
    sy_probe(6);
    SY_TL_END();
}

template <typename T, int TP, int BD, int ILV = 1, int WC = 4>
int launch_halo3(const ConvArgs& a_in, void* stream) {
    constexpr int NW = 4, WP = NW / WC, CT = WC * 32, TH = WP * TP, PT = TH * 32;
    constexpr int HR = (TH + 2) * kHaloW, NI = ((HR + 15) / 16 + NW - 1) / NW, BUF = NW * NI * 16 * 64;
    ConvArgs a = a_in;
    a.s2_classes = 0;
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W || a.ksplit > 1) return SY_ERR_UNSUPPORTED;
    if (a.Cin % (4 * T::kEPC) != 0 || a.Cin < 4 * T::kEPC || a.x_extent == 0 || a.wfrag == nullptr || a.wfrag_extent == 0) return SY_ERR_UNSUPPORTED;
    constexpr size_t smem_k = (size_t)2 * BUF;
    constexpr size_t smem_e = (size_t)EpiLds<WP, CT>::kStatBytes + (size_t)PT * (CT * 2 + 16) + (size_t)PT * 8;
    constexpr bool can_stage = (T::kEPC == 8 && smem_e <= StageLimit<WC, WP, 1, TP>::kBytes);
    constexpr size_t smem_s = (size_t)WP * CT * 8;
    constexpr size_t smem = (can_stage && smem_e > smem_k) ? smem_e : (smem_s > smem_k ? smem_s : smem_k);
    const int tiles = a.N * ((a.Ho + TH - 1) / TH) * ((a.Wo + 31) / 32);
    dim3 grid((a.Cout + CT - 1) / CT, tiles, 1);
    const bool dgrad = a.mode == SY_CONV_DGRAD;
#ifndef SY_EMU
    static sy_dev_once attr_done;
    if (attr_done.need()) {
        if (hipFuncSetAttribute((const void*)conv3x3_halo3_kernel<T, TP, 0, BD, ILV, WC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv3x3_halo3_kernel<T, TP, 1, BD, ILV, WC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return SY_ERR_LAUNCH;
        attr_done.mark();
    }
#endif
    if (dgrad) { SY_LAUNCH((conv3x3_halo3_kernel<T, TP, 1, BD, ILV, WC>), grid, dim3(NW * 64), smem, stream, a); }
    else { SY_LAUNCH((conv3x3_halo3_kernel<T, TP, 0, BD, ILV, WC>), grid, dim3(NW * 64), smem, stream, a); }
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

// tile codes 98, 100, 101, 109 (128 channels) and 96, 97 (64 channels) of sy_conv_desc::tile.  (A ring of 6 for the 2- / 3-row tiles — five steps of reads in flight — measured
// the same or slower than the ring of 3: profiles/r05 stage h.)
template <typename T>
int launch_halo3_typed(const ConvArgs& a, void* stream) {
    switch (a.tile) {
        case 101: return launch_halo3<T, 2, 3>(a, stream);     // 128 ch x (2 rows x 32 px)   (second generation: 117)
        case 100: return launch_halo3<T, 3, 3>(a, stream);     // 128 ch x (3 rows x 32 px)   (107)
        case 98: return launch_halo3<T, 4, 3>(a, stream);      // 128 ch x (4 rows x 32 px)   (118)
        case 109: return launch_halo3<T, 5, 3>(a, stream);     // 128 ch x (5 rows x 32 px)   (104)
        case 96: return launch_halo3<T, 2, 3, 1, 2>(a, stream);   //  64 ch x (4 rows x 32 px): 2 channel blocks x 2 row groups
        case 97: return launch_halo3<T, 3, 3, 1, 2>(a, stream);   //  64 ch x (6 rows x 32 px)
        default: return SY_ERR_ARG;
    }
}

}  // namespace sy_conv
