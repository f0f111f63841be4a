// conv3x3_halo.h — 3x3 stride-1 convolution (forward and data gradient) with the input tile + halo resident in LDS.
//
// Replaces the same reference code as conv_igemm_impl.h (yolox BaseConv's Conv2d of the Bottleneck / head-tower 3x3 layers —
// exps/model/darknet.py:118-165 via CSPLayer, dfp_pafpn.py:33-81, tal_head.py:55-104 — and cuDNN backward-data) for the layers
// that carry 55 % of the step's convolution FLOPs.
//
// Why a second kernel: the implicit-GEMM kernel walks K = (tap, channel slab) and re-stages the pixel operand for every tap —
// nine trips L2 -> LDS per input element, 4-8 MFMAs per wave between barriers.  Here a workgroup owns TH image rows x 32
// pixels of ONE image and, per 64-byte channel slab, parks the (TH + 2) x 34 halo tile in LDS ONCE; the nine taps are nine
// shifted windows of that tile:
//   * pixel operand (MFMA B):  L2 -> LDS by LDS-DMA (buffer_load ... lds, no VGPR round trip, out-of-image pixels arrive as
//     zeros through the buffer bounds check), two halo buffers, ONE barrier per channel slab = 18 x TC x TP MFMAs per wave
//     between barriers (72-144 instead of 4-8), global -> LDS traffic 1.3-2.1x the tile instead of 9x;
//   * a pixel tile of the MFMA = 32 consecutive pixels of one image row, so a lane's fragment address is
//     (row + tap offset) * 64 B: rows of 64 B land lane-linear from the DMA, physical 16-byte slot = logical chunk ^ ((row >> 2) & 3)
//     (swizzle applied on the SOURCE side of the DMA and on the read; conflict free for every tap offset: the four 4-row
//     runs of a ds_read_b128 lane group start 0 / 12 / 20 / 24 rows apart = 0 / 3 / 1 / 2 mod 4 row quads);
//   * weight operand (MFMA A): fragment-packed weights straight from L2 into VGPRs (the SY_TILE_WR layout of
//     conv_igemm_impl.h), prefetched two taps ahead through a 3-stage register ring, no LDS, no barrier;
//   * the data gradient is the same kernel reading tap (kh, kw) at window offset (2 - kh, 2 - kw) with the transposed weights.
// Epilogue: conv_epilogue of conv_igemm_impl.h through the TilePixels mapper (BN statistics, staged coalesced write-out).
#pragma once
#include "conv_igemm_impl.h"

#ifndef SY_HALO2_BD
#define SY_HALO2_BD 3
#endif

namespace sy_conv {

constexpr int kHaloW = 34;            // 32 pixels + 1 halo pixel on each side

template <typename T, int WC, int WP, int TC, int TP>
__global__ __launch_bounds__(WC * WP * 64, (TC * TP <= 2 ? 3 : 2)) void conv3x3_halo_kernel(ConvArgs p) {
    SY_TL_BEGIN(3 + (p.mode == SY_CONV_DGRAD ? 32 : 0));
    constexpr int kThreads = WC * WP * 64;
    constexpr int NW = WC * WP;
    constexpr int EPC = T::kEPC;
    constexpr int ESZ = 16 / EPC;
    constexpr int BK = 4 * EPC;                  // channels per 64-byte slab
    constexpr int CT = WC * TC * 32;
    constexpr int TH = WP * TP;                  // image rows per tile (one 32-pixel row segment per MFMA pixel tile)
    constexpr int PT = TH * 32;
    constexpr int HR = (TH + 2) * kHaloW;        // halo rows (pixels) per slab
    constexpr int NI = ((HR + 15) / 16 + NW - 1) / NW;   // DMA wave-instructions (16 rows each) per wave per slab
    constexpr int BUF = NW * NI * 16 * 64;       // bytes of one halo buffer (rows past HR: zero filler, never read)
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");

    SY_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sy_uniform(tid >> 6);
    const int wc = wave / WP;
    const int wp = wave % WP;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const sy_block_id bid = sy_xcd_block_id();   // x: channel tile (fastest: the channel tiles of a pixel tile share its halo in L2)
    const int tiles_w = (p.Wo + 31) >> 5, tiles_h = (p.Ho + TH - 1) / TH;
    const int tw = bid.y % tiles_w, th_ = (bid.y / tiles_w) % tiles_h, n = bid.y / (tiles_w * tiles_h);
    const int h0 = th_ * TH, w0 = tw * 32;

    // ---- DMA assignment: instruction j = wave + i * NW fills halo rows [16 j, 16 j + 16); lane -> (row, physical 16-byte slot)
    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    unsigned voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = (wave + i * NW) * 16 + (lane >> 2);
        const int hy = r / kHaloW, hx = r - hy * kHaloW;
        const int h = h0 - 1 + hy, w = w0 - 1 + hx;
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        const bool ok = r < HR && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W && !(p.ablate & 1);
        voff[i] = ok ? (unsigned)((((long long)n * p.xbs + ((long long)h * p.W + w) * p.ldx) + chunk * EPC) * ESZ) : 0xFFFFFFFFu;
    }
    const sy_lds_base_t lds0 = sy_lds_base(smem);
    // One DMA piece (16 halo rows) of the next slab.  VMEM loads retire IN ORDER, and the weight-fragment stream below is
    // consumed two taps after it is issued — so a wait for fragments also waits for every DMA piece issued before them.
    // The pieces are therefore spread over the first NI taps of a slab, one per tap behind that tap's fragment fetch: each
    // has ~3 taps of MFMA time to land, none ever stalls a fragment wait by more than that, and all of them are in LDS long
    // before the slab-end barrier.
    auto issue_piece = [&](auto i_, int buf, int cslab) {
        constexpr int I = decltype(i_)::value;
        const unsigned s_x = (unsigned)(cslab * BK * ESZ);
        sy_glds16_buf_at(bufx, voff[I] == 0xFFFFFFFFu ? 0xFFFFFFFFu : voff[I] + s_x, lds0,
                         (unsigned)(buf * BUF + (wave + I * NW) * 1024));
    };

    // ---- weight fragments: [ct][cslab][tap][g][half][32 rows][EPC], 1 KiB per (ct, cslab, tap, g)
    const sy_buffer buff = sy_make_buffer(p.wfrag, p.wfrag_extent);
    const int ncs = p.Cin / BK;
    const int ntile32 = (p.Cout + 31) / 32;
    unsigned foff[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int ct = bid.x * (CT / 32) + wc * TC + t;
        foff[t] = (ct < ntile32 && !(p.ablate & 2)) ? (unsigned)((((long long)ct * ncs * 9) * 128 + lane) * 16) : 0xFFFFFFFFu;
    }
    uint4 fr[3][TC][2];
    int f_cs = 0, f_tap = 0;                                  // (channel slab, tap) of the next fragment set to fetch
    auto fetch = [&](auto st_) {
        constexpr int S = decltype(st_)::value;
        const unsigned s_f = (unsigned)((f_cs * 9 + f_tap) * 2048);
        const bool live = f_cs < ncs;
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                fr[S][t][g] = sy_buffer_load16_s(buff, (live && foff[t] != 0xFFFFFFFFu) ? foff[t] + (unsigned)(g * 1024) : 0xFFFFFFFFu, s_f);
        if (++f_tap == 9) { f_tap = 0; ++f_cs; }
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    // window origin of this lane's pixel of tile u (halo row index of tap offset (0, 0))
    int rbase[TP];
#pragma unroll
    for (int u = 0; u < TP; ++u) rbase[u] = (wp * TP + u) * kHaloW + l31;
    const bool fwd = (p.mode == SY_CONV_FWD);

    static_assert(NI <= 9, "one DMA piece per tap");
    sy_static_for<0, NI>([&](auto i_) { issue_piece(i_, 0, 0); });
    fetch(sy_int<0>());
    fetch(sy_int<1>());
    for (int cs = 0; cs < ncs; ++cs) {
        sy_wait_vmcnt<0>();                       // this wave's part of slab cs has landed (and its fragment prefetches)
        sy_barrier();                             // ... everybody's; every wave is done reading the other buffer
        const bool more = cs + 1 < ncs;
        const unsigned char* const hb = smem + (cs & 1) * BUF;
        // taps: kh is a RUNTIME loop (three trips), kw unrolled — the fragment ring position (tap % 3 == kw) stays a compile-time
        // index, the code is a third of the fully unrolled form and the register allocator no longer hoists nine taps of
        // address arithmetic and LDS reads (the 64-register tile spilled, tools/regs_census.sh)
        for (int kh = 0; kh < 3; ++kh) {
            sy_static_for<0, 3>([&](auto kw_) {
                constexpr int KW = decltype(kw_)::value;
                fetch(sy_int<(KW + 2) % 3>());    // two taps ahead (its registers held the previous tap)
                if (more) {                       // DMA piece number (3 kh + kw) of the next slab
                    if constexpr (KW < NI) { if (kh == 0) issue_piece(sy_int<KW>(), (cs + 1) & 1, cs + 1); }
                    if constexpr (3 + KW < NI) { if (kh == 1) issue_piece(sy_int<3 + KW>(), (cs + 1) & 1, cs + 1); }
                    if constexpr (6 + KW < NI) { if (kh == 2) issue_piece(sy_int<6 + KW>(), (cs + 1) & 1, cs + 1); }
                }
                const int toff = fwd ? (kh * kHaloW + KW) : ((2 - kh) * kHaloW + (2 - KW));
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    uint4 b[TP];
#pragma unroll
                    for (int u = 0; u < TP; ++u) {
                        const int row = rbase[u] + toff;
                        b[u] = *reinterpret_cast<const uint4*>(hb + row * 64 + (((g * 2 + half) ^ ((row >> 2) & 3)) << 4));
                    }
#pragma unroll
                    for (int t = 0; t < TC; ++t)
#pragma unroll
                        for (int u = 0; u < TP; ++u) acc[t][u] = sy_mfma_group(T(), fr[KW][t][g], b[u], acc[t][u]);
                }
            });
        }
    }

    SY_LATE_ARGS(ConvArgs, p);
    int e_bx = bid.x, e_n = n, e_h0 = h0, e_w0 = w0, e_by = bid.y;
    SY_LAUNDER_INT(e_bx); SY_LAUNDER_INT(e_n); SY_LAUNDER_INT(e_h0); SY_LAUNDER_INT(e_w0); SY_LAUNDER_INT(e_by);
    TilePixels mp;
    mp.n = e_n; mp.h0 = e_h0; mp.w0 = e_w0; mp.Ho = p_late.Ho; mp.Wo = p_late.Wo; mp.rep = e_by;
    mp.seg = p_late.seg_M > 0 ? (e_n * p_late.HoWo) / p_late.seg_M : 0;
    // Output pixels of the tile that lie outside the image still see valid input through their window (the implicit-GEMM
    // kernel's out-of-range rows read zeros): clear them, the BatchNorm statistics of the epilogue sum every accumulator.
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        int n_, rem_;
        if (!mp.map((wp * TP + u) * 32 + l31, n_, rem_)) {
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;
        }
    }
    conv_epilogue<T, WC, WP, TC, TP>(p_late, mp, e_bx, acc, smem, tid);
    SY_TL_END();
}

// ---- second generation: the same tile, software-pipelined inside the wave ---------------------------------------------
// The ISA of the kernel above shows why its waves sit at ~35 % MFMA issue: every pair of MFMAs waits on the ds_read_b128
// issued right in front of it (lgkmcnt(0), ~120 cycles), the weight fragments arrive two taps = 4 MFMAs = 128 cycles after
// they were requested from L2 (~500+ cycles), and the slab-top `vmcnt(0)` drains the fragment prefetch every 36 MFMAs.
// Here, per wave:
//   * weight fragments: ALL nine taps of a slab live in registers (72 x TC VGPRs); the slot of tap t is refilled with tap t
//     of the NEXT slab right after its MFMAs — every fragment has a whole slab (18 x TC x TP MFMAs) to arrive;
//   * pixel fragments: a 3-deep register ring over the 18 (tap, k-half) steps of a slab, read two steps ahead of the MFMAs
//     from LDS addresses precomputed once per kernel (one v_add per read for the halo-buffer parity);
//   * the slab-top wait is COUNTED: only the slab's DMA pieces (issued behind taps 0 .. NI-1 of the previous slab) must
//     have landed — the (9 - NI) x 2 TC fragment loads issued after the last piece stay in flight across the barrier;
//   * no branches in the slab body: the DMA pieces of the slab after the last one are issued out of range (zeros into the
//     idle buffer, drained before the epilogue reuses the LDS).
//   * S2 (forward only): the same kernel for STRIDE 2.  The tile of TH x 32 output pixels needs (2 TH + 1) x 65 input pixels; they
//     are parked with the columns split by parity — LDS row of input pixel (hy, hx) = hy * 66 + (hx & 1) * 33 + (hx >> 1) — so
//     that the 32 lanes of a tap (input column 2 * lane + kw) read 32 CONSECUTIVE rows exactly as at stride 1 (same XOR swizzle,
//     conflict free), and the parity split costs nothing: every DMA lane picks its own global pixel anyway.
//   * KS > 1 (small launches, forward): the workgroup's waves are KS groups of WC x WP, group g contracting channel slabs g, g + KS,
//     ... of the workgroup's range into its own accumulators from its own pair of LDS buffers — every wave still fetches only the
//     weights of ITS channels x ITS slabs, so the bytes a wave must pull through its 72-register fragment window halve (KS = 2)
//     or quarter: at batch 1 the main loop is bound by exactly that (one wave per SIMD, ~18 KB in flight per wave; 113's eight
//     waves that DUPLICATE the weights are slower than 117's four — profiles/r04 stage ad).  The groups' partial tiles are
//     summed through LDS in group order (deterministic), groups > 0 exit, group 0 runs the ordinary epilogue.
template <typename T, int WC, int WP, int TC, int TP, int S2 = 0, int KS = 1>
__global__ __launch_bounds__(WC * WP * KS * 64, (KS > 1 ? (WC * WP * KS <= 4 ? 3 : 1) : (TC * TP <= 2 ? 3 : (TC * TP >= 8 ? 1 : 2))))
void conv3x3_halo2_kernel(ConvArgs p) {
    SY_TL_BEGIN(2 + (p.mode == SY_CONV_DGRAD ? 32 : 0));
    constexpr int NW = WC * WP;
    constexpr int EPC = T::kEPC;
    constexpr int ESZ = 16 / EPC;
    constexpr int BK = 4 * EPC;
    constexpr int CT = WC * TC * 32;
    constexpr int TH = WP * TP;
    constexpr int RW = S2 ? 66 : kHaloW;                   // LDS rows per input row of the tile
    constexpr int HR = (S2 ? 2 * TH + 1 : TH + 2) * RW;
    constexpr int NI = ((HR + 15) / 16 + NW - 1) / NW;
    constexpr int BUF = NW * NI * 16 * 64;
    constexpr int BD = (TC * TP <= 2) ? SY_HALO2_BD : 3;   // pixel-fragment ring: BD - 1 (tap, k-half) steps of reads in flight
    static_assert(NW == 4 || NW == 8 || (KS > 1 && NW <= 2), "4 or 8 waves (1 or 2 per K group)");
    static_assert(NI <= 9, "one DMA piece per tap");

    SY_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = sy_uniform(tid >> 6);
    const int kg = KS > 1 ? wave_all / NW : 0;             // K group of this wave
    const int wave = KS > 1 ? wave_all % NW : wave_all;    // wave inside its group
    const int wc = wave / WP;
    const int wp = wave % WP;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const sy_block_id bid = sy_xcd_block_id();
    const int tiles_w = (p.Wo + 31) >> 5, tiles_h = (p.Ho + TH - 1) / TH;
    const int tw = bid.y % tiles_w, th_ = (bid.y / tiles_w) % tiles_h, n = bid.y / (tiles_w * tiles_h);
    const int h0 = th_ * TH, w0 = tw * 32;

    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    unsigned voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = (wave + i * NW) * 16 + (lane >> 2);
        const int hy = r / RW;
        int hx = r - hy * RW;
        bool col_ok = true;
        if constexpr (S2) { col_ok = hx != 65; hx = hx >= 33 ? 2 * (hx - 33) + 1 : 2 * hx; }     // parity planes: even columns, then odd
        const int h = (S2 ? 2 * h0 : h0) - 1 + hy, w = (S2 ? 2 * w0 : w0) - 1 + hx;
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        const bool ok = r < HR && col_ok && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W && !(p.ablate & 1);
        voff[i] = ok ? (unsigned)((((long long)n * p.xbs + ((long long)h * p.W + w) * p.ldx) + chunk * EPC) * ESZ) : 0xFFFFFFFFu;
    }
    const sy_lds_base_t lds0 = sy_lds_base(smem);
    // split-K (p.ksplit > 1): this workgroup contracts channel slabs [cs_begin, ncs) of the layer only (gridDim.z ranges) and
    // writes fp32 partial sums; `ncs` below is the END of its range
    const int ncs_all = p.Cin / BK;
    const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
    const int kz = ksplit > 1 ? bid.z : 0;
    const int cs_begin = sy_uniform((kz * ncs_all) / ksplit);
    const int ncs = sy_uniform(((kz + 1) * ncs_all) / ksplit);
    // trip j of the slab loop: group kg contracts slab cs_begin + j * KS + kg (KS == 1: slab cs_begin + j) from LDS buffer j & 1 of its pair
    const unsigned gbo = (unsigned)(kg * 2 * BUF);
    auto issue_piece = [&](auto i_, int j) {               // piece I of trip j's slab (out of range past the last slab)
        constexpr int I = decltype(i_)::value;
        const int cslab = cs_begin + j * KS + kg;
        const unsigned s_x = (unsigned)(cslab * BK * ESZ);
        const bool dead = voff[I] == 0xFFFFFFFFu || cslab >= ncs;
        sy_glds16_buf_at(bufx, dead ? 0xFFFFFFFFu : voff[I] + s_x, lds0, gbo + (unsigned)((j & 1) * BUF + (wave + I * NW) * 1024));
    };

    const sy_buffer buff = sy_make_buffer(p.wfrag, p.wfrag_extent);
    const int ntile32 = (p.Cout + 31) / 32;
    unsigned foff[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int ct = bid.x * (CT / 32) + wc * TC + t;
        foff[t] = (ct < ntile32 && !(p.ablate & 2)) ? (unsigned)((((long long)ct * ncs_all * 9) * 128 + lane) * 16) : 0xFFFFFFFFu;
    }
    uint4 fr[9][TC][2];
    auto fetch = [&](auto tap_, int j) {                   // fragments of tap TAP of trip j's slab into their slot
        constexpr int TAP = decltype(tap_)::value;
        const int cslab = cs_begin + j * KS + kg;
        const unsigned s_f = (unsigned)((cslab * 9 + TAP) * 2048);
        const bool live = cslab < ncs;
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                fr[TAP][t][g] = sy_buffer_load16_s(buff, (live && foff[t] != 0xFFFFFFFFu) ? foff[t] + (unsigned)(g * 1024) : 0xFFFFFFFFu, s_f);
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    // LDS byte offset of this lane's fragment row for (tap, pixel tile), k-half 0 (k-half 1 = ^ 32: the swizzle is an XOR)
    const bool fwd = (p.mode == SY_CONV_FWD);
    unsigned ba[9][TP];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap % 3;
        const int toff = S2 ? (kh * RW + (kw & 1) * 33 + (kw >> 1)) : (fwd ? (kh * kHaloW + kw) : ((2 - kh) * kHaloW + (2 - kw)));
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            const int row = (S2 ? 2 : 1) * (wp * TP + u) * RW + l31 + toff;
            ba[tap][u] = (unsigned)(row * 64 + ((half ^ ((row >> 2) & 3)) << 4));
        }
    }

    sy_probe(0);
    sy_static_for<0, NI>([&](auto i_) { issue_piece(i_, 0); });
    sy_static_for<0, 9>([&](auto t_) { fetch(t_, 0); sy_sched_fence(); });   // in tap order: the compiler's vmcnt waits count on it
    sy_probe(1);
    const int trips = (ncs - cs_begin + KS - 1) / KS;      // the same for every group (slabs past the range contribute zeros)
    for (int j = 0; j < trips; ++j) {
        sy_wait_vmcnt<(9 - NI) * 2 * TC>();      // the slab's DMA pieces (older than the last (9 - NI) taps of fragment loads)
        sy_barrier();                             // ... everybody's; every wave is done reading the other buffer
        if (j == 0) sy_probe(2);
        const unsigned hbo = gbo + (unsigned)((j & 1) * BUF);
        uint4 b[BD][TP];
        auto read_step = [&](auto s_) {
            constexpr int S = decltype(s_)::value;
            constexpr int TAP = S >> 1, G = S & 1;
#pragma unroll
            for (int u = 0; u < TP; ++u)
                b[S % BD][u] = *reinterpret_cast<const uint4*>(smem + ((hbo + ba[TAP][u]) ^ (unsigned)(G * 32)));
        };
        sy_static_for<0, BD - 1>([&](auto s_) { read_step(s_); });
        sy_static_for<0, 18>([&](auto s_) {
            constexpr int S = decltype(s_)::value;
            constexpr int TAP = S >> 1, G = S & 1;
            if constexpr (S + BD - 1 < 18) read_step(sy_int<S + BD - 1>());
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int u = 0; u < TP; ++u) acc[t][u] = sy_mfma_group(T(), fr[TAP][t][G], b[S % BD][u], acc[t][u]);
            if constexpr (G == 1) {
                fetch(sy_int<TAP>(), j + 1);
                if constexpr (TAP < NI) issue_piece(sy_int<TAP>(), j + 1);
            }
            sy_sched_fence();
        });
    }
    sy_wait_vmcnt<0>();                           // the out-of-range pieces of the slab after the last one
    sy_barrier();
    sy_probe(7);
    if constexpr (KS > 1) {
        // partial tiles of groups 1 .. KS-1 through LDS ([group - 1][wave][register][lane]: conflict-free), summed by group 0 in
        // group order; the other groups' waves END here (an ended wave no longer counts at s_barrier)
        float* const red = reinterpret_cast<float*>(smem);
        constexpr int RW_ = TC * TP * 16;
        if (kg > 0) {
            float* const dst = red + ((size_t)((kg - 1) * NW + wave) * RW_) * 64 + lane;
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int u = 0; u < TP; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((t * TP + u) * 16 + r) * 64] = acc[t][u][r];
        }
        sy_barrier_lds();                         // the partial tiles are ds_write stores: drained before the rendezvous
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const float* const src = red + ((size_t)((g - 1) * NW + wave) * RW_) * 64 + lane;
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int u = 0; u < TP; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][u][r] += src[((t * TP + u) * 16 + r) * 64];
        }
        sy_barrier();                             // (group 0 only from here on) the epilogue reuses the LDS
    }

    SY_LATE_ARGS(ConvArgs, p);
    sy_probe(3);
    int e_bx = bid.x, e_n = n + kz * p.N, e_h0 = h0, e_w0 = w0, e_by = bid.y;      // split z: partial "images" [z*N, (z+1)*N)
    SY_LAUNDER_INT(e_bx); SY_LAUNDER_INT(e_n); SY_LAUNDER_INT(e_h0); SY_LAUNDER_INT(e_w0); SY_LAUNDER_INT(e_by);
    TilePixels mp;
    mp.n = e_n; mp.h0 = e_h0; mp.w0 = e_w0; mp.Ho = p_late.Ho; mp.Wo = p_late.Wo; mp.rep = e_by;
    mp.seg = p_late.seg_M > 0 ? (e_n * p_late.HoWo) / p_late.seg_M : 0;
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        int n_, rem_;
        if (!mp.map((wp * TP + u) * 32 + l31, n_, rem_)) {
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;
        }
    }
    conv_epilogue<T, WC, WP, TC, TP>(p_late, mp, e_bx, acc, smem, tid);
    sy_probe(6);
    SY_TL_END();
}

template <typename T, int WC, int WP, int TC, int TP, int GEN = 1, int S2 = 0, int KS = 1>
int launch_halo(const ConvArgs& a_in, void* stream) {
    constexpr int NW = WC * WP, CT = WC * TC * 32, TH = WP * TP, PT = TH * 32;
    static_assert(KS == 1 || GEN == 2, "K groups: second-generation kernel");
    constexpr int HR = S2 ? (2 * TH + 1) * 66 : (TH + 2) * kHaloW, NI = ((HR + 15) / 16 + NW - 1) / NW, BUF = NW * NI * 16 * 64;
    static_assert(!S2 || GEN == 2, "stride 2: second-generation kernel, forward");
    ConvArgs a = a_in;
    a.s2_classes = 0;
    // 3x3, stride 1, "same" padding, whole channel slabs, 32-bit addressable input, fragment-packed weights
    if constexpr (S2) {
        if (a.KH != 3 || a.KW != 3 || a.stride != 2 || a.pad != 1 || a.Ho != (a.H + 1) / 2 || a.Wo != (a.W + 1) / 2 || a.mode != SY_CONV_FWD ||
            a.ksplit > 1)
            return SY_ERR_UNSUPPORTED;
    } else {
        if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W) return SY_ERR_UNSUPPORTED;
    }
    if (a.Cin % (4 * T::kEPC) != 0 || a.x_extent == 0 || a.wfrag == nullptr || a.wfrag_extent == 0) return SY_ERR_UNSUPPORTED;
    constexpr size_t smem_r = (size_t)(KS - 1) * NW * 64 * TC * TP * 16 * 4;      // K groups: partial tiles of groups 1 .. KS-1
    constexpr size_t smem_k = (size_t)KS * 2 * BUF > smem_r ? (size_t)KS * 2 * BUF : smem_r;
    constexpr size_t smem_e = (size_t)EpiLds<WP, CT>::kStatBytes + (size_t)PT * (CT * 2 + 16) + (size_t)PT * 8;
    constexpr bool can_stage = (T::kEPC == 8 && smem_e <= StageLimit<WC, WP, TC, TP>::kBytes);
    constexpr size_t smem_s = (size_t)WP * CT * 8;            // statistics scratch of the un-staged epilogue
    constexpr size_t smem = (can_stage && smem_e > smem_k) ? smem_e : (smem_s > smem_k ? smem_s : smem_k);
    const int tiles = a.N * ((a.Ho + TH - 1) / TH) * ((a.Wo + 31) / 32);
    if (a.ksplit > 1 && (GEN != 2 || KS > 1 || a.ksplit > a.Cin / (4 * T::kEPC))) return SY_ERR_UNSUPPORTED;
    dim3 grid((a.Cout + CT - 1) / CT, tiles, a.ksplit > 1 ? a.ksplit : 1);
#ifndef SY_EMU
    static sy_dev_once attr_done;
    if (attr_done.need()) {
        const void* fn;                                        // (if constexpr: only the generation this tile code launches is instantiated)
        if constexpr (GEN == 2) fn = (const void*)conv3x3_halo2_kernel<T, WC, WP, TC, TP, S2, KS>;
        else fn = (const void*)conv3x3_halo_kernel<T, WC, WP, TC, TP>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return SY_ERR_LAUNCH;
        attr_done.mark();
    }
#endif
    if constexpr (GEN == 2) {
        SY_LAUNCH((conv3x3_halo2_kernel<T, WC, WP, TC, TP, S2, KS>), grid, dim3(NW * KS * 64), smem, stream, a);
    } else {
        SY_LAUNCH((conv3x3_halo_kernel<T, WC, WP, TC, TP>), grid, dim3(NW * 64), smem, stream, a);
    }
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

// tile codes 104..107, 110..118 of sy_conv_desc::tile
template <typename T>
int launch_halo_typed(const ConvArgs& a, void* stream) {
    switch (a.tile) {
        case 114: return launch_halo<T, 4, 2, 1, 1>(a, stream);     // 128 ch x ( 2 rows x 32 px), 8 waves x (32 ch x 32 px)
        case 115: return launch_halo<T, 4, 1, 1, 2>(a, stream);     // 128 ch x ( 2 rows x 32 px)
        case 116: return launch_halo<T, 1, 4, 2, 2>(a, stream);     //  64 ch x ( 8 rows x 32 px)
        // second generation (in-wave software pipeline) of 115 / 113
        case 117: return launch_halo<T, 4, 1, 1, 2, 2>(a, stream);
        case 118: return launch_halo<T, 4, 1, 1, 4, 2>(a, stream);
        case 107: return launch_halo<T, 4, 1, 1, 3, 2>(a, stream);     // 128 ch x (3 rows x 32 px): between 117 and 118
        case 104: return launch_halo<T, 4, 1, 1, 5, 2>(a, stream);     // 128 ch x (5 rows x 32 px)
        // small launches (the streaming frame, the 19x30 maps): one 32 ch x 32 px MFMA tile per wave — half the serial MFMA chain
        // of 117 per wave, and (112) twice the workgroups over which the layer's weights are fetched
        case 112: return launch_halo<T, 2, 2, 1, 1, 2>(a, stream);  //  64 ch x ( 2 rows x 32 px), 4 waves
        case 113: return launch_halo<T, 4, 2, 1, 1, 2>(a, stream);  // 128 ch x ( 2 rows x 32 px), 8 waves
        // K groups inside the workgroup (the fp32 summation order differs from the other tiles': plans opt in)
        case 111: return launch_halo<T, 2, 1, 1, 2, 2, 0, 2>(a, stream);  //  64 ch x (2 rows x 32 px), 2 groups of 2 waves
        // (64 ch x 4 groups and 128 ch x 2 groups measured no better than 111 / 117: the main loop follows the weight bytes per CU)
        case 106: return launch_halo<T, 1, 1, 1, 2, 2, 0, 4>(a, stream);  //  32 ch x (2 rows x 32 px), 4 groups of 1 wave
        case 105: return launch_halo<T, 2, 1, 1, 1, 2, 1, 2>(a, stream);  // STRIDE 2 forward, 64 ch x (1 row x 32 px), 2 groups of 2 waves
        // (round 5: 104 / 107 / 117 / 118 as TWO K groups of four waves = two waves per SIMD contracting alternate channel slabs were
        //  measured and removed — the two waves run in lockstep, the SIMD's MFMA rate does not change (main loop 17.9 vs 18.8 us at
        //  256->256 @38x60 x 8) and the partial-tile exchange costs 0.8 us: 5-20 % slower on every training shape, profiles/r05 stage c)
        // STRIDE 2, forward (tile 117's configuration over a parity-split input window): +7 % / +26 % over the implicit-GEMM
        // variants on dark2.0 / dark4.0 (profiles/r04/a_probe_s2_stats.txt)
        case 110: return launch_halo<T, 4, 1, 1, 2, 2, 1>(a, stream);
        default: return SY_ERR_ARG;
    }
}

}  // namespace sy_conv
