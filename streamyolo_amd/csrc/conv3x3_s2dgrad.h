// conv3x3_s2dgrad.h — data gradient of the 3x3 STRIDE-2 convolutions (tile code 108 of sy_conv_desc::tile), the window-in-LDS way.
//
// Reference op: the autograd backward of nn.Conv2d(k=3, s=2, p=1) inside yolox BaseConv (CSPDarknet stems dark2-5,
// exps/model/darknet.py; the PAN bu_convs, exps/model/dfp_pafpn.py:96-105) = cuDNN backward-data.  Today it runs on the implicit-GEMM
// kernel as four output-parity classes (conv_igemm_impl.h, s2_classes).
//
// With y[ho, wo] += x[2 ho - 1 + kh, 2 wo - 1 + kw] * W[kh, kw], the gradient of input pixel (h, w) collects dy[(h + 1 - kh) / 2,
// (w + 1 - kw) / 2] * W[kh, kw] over the taps for which both divisions are exact.  For the pixels of ONE parity class (ph, pw) =
// (h & 1, w & 1), written dx[2 i + ph, 2 j + pw], that is a dense STRIDE-1 stencil over dy with at most four taps:
//     ph = 0: kh = 1 reads dy row i            ph = 1: kh = 0 reads row i + 1, kh = 2 reads row i
//     pw = 0: kw = 1 reads dy column j         pw = 1: kw = 0 reads column j + 1, kw = 2 reads column j
// i.e. 1 / 2 / 2 / 4 taps for the classes (0,0) / (0,1) / (1,0) / (1,1).  A workgroup owns TH x 32 class pixels of one class
// (gridDim.z = class) and 128 output channels; per 64-byte slab of dy channels the (TH + 1) x 34 window of dy is parked once in LDS by
// LDS-DMA (two buffers, the same XOR swizzle as conv3x3_halo.h, out-of-image pixels = zeros), the taps are pixel offsets of one
// ds_read_b128 fragment read, the transposed weights come fragment-packed straight into VGPRs one slab ahead.  The epilogue is
// conv_epilogue of conv_igemm_impl.h through a mapper that scatters the tile to the class's pixels of dx (first write or +=).
#pragma once
#include "conv3x3_halo.h"

namespace sy_conv {

struct ClassPixels {              // TH rows x 32 class pixels (i, j) of parity class (ph, pw): dx pixel (2 i + ph, 2 j + pw)
    int n, i0, j0, ph, pw, Ho, Wo, seg, rep;
    __device__ __forceinline__ bool map(int lp, int& n_, int& rem) const {
        const int h = 2 * (i0 + (lp >> 5)) + ph, w = 2 * (j0 + (lp & 31)) + pw;
        n_ = n;
        if (h >= Ho || w >= Wo) { rem = 0; return false; }
        rem = h * Wo + w;
        return true;
    }
};

template <typename T, int WC, int WP, int TC, int TP>
__global__ __launch_bounds__(WC * WP * 64, 3) void conv3x3_s2dgrad_kernel(ConvArgs p) {
    SY_TL_BEGIN(5 + 32);
    constexpr int NW = WC * WP;
    constexpr int EPC = T::kEPC;
    constexpr int ESZ = 16 / EPC;
    constexpr int BK = 4 * EPC;
    constexpr int CT = WC * TC * 32;
    constexpr int TH = WP * TP;
    constexpr int HR = (TH + 1) * kHaloW;                  // window rows x 34 columns (33 used)
    constexpr int NI = ((HR + 15) / 16 + NW - 1) / NW;     // DMA pieces per wave and slab
    constexpr int BUF = NW * NI * 16 * 64;
    static_assert(NW == 4, "four waves");

    SY_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sy_uniform(tid >> 6);
    const int wc = wave / WP;
    const int wp = wave % WP;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const sy_block_id bid = sy_xcd_block_id();
    const int cls = bid.z, ph = cls >> 1, pw = cls & 1;
    const int Hc = (p.Ho + 1) >> 1, Wc = (p.Wo + 1) >> 1;            // class grid of the largest class
    const int tiles_w = (Wc + 31) >> 5, tiles_h = (Hc + TH - 1) / TH;
    const int tw = bid.y % tiles_w, th_ = (bid.y / tiles_w) % tiles_h, n = bid.y / (tiles_w * tiles_h);
    const int i0 = th_ * TH, j0 = tw * 32;

    // ---- the class's taps: slot q -> (weight tap kh * 3 + kw, window offset dh * 34 + dw)
    const int nth = 1 + ph, ntw = 1 + pw, nt = sy_uniform(nth * ntw);
    int tap_w[4], tap_o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int qh = ntw == 2 ? (q >> 1) : q, qw = ntw == 2 ? (q & 1) : 0;
        const int kh = ph == 0 ? 1 : (qh == 0 ? 0 : 2), dh = (ph == 1 && qh == 0) ? 1 : 0;
        const int kw = pw == 0 ? 1 : (qw == 0 ? 0 : 2), dw = (pw == 1 && qw == 0) ? 1 : 0;
        tap_w[q] = sy_uniform(kh * 3 + kw);
        tap_o[q] = sy_uniform(dh * kHaloW + dw);
    }

    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    unsigned voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = (wave + i * NW) * 16 + (lane >> 2);
        const int hy = r / kHaloW, hx = r - hy * kHaloW;
        const int h = i0 + hy, w = j0 + hx;                            // dy pixel (p.H x p.W = the SMALL map of this launch's input)
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        const bool ok = r < HR && h < p.H && w < p.W;
        voff[i] = ok ? (unsigned)((((long long)n * p.xbs + ((long long)h * p.W + w) * p.ldx) + chunk * EPC) * ESZ) : 0xFFFFFFFFu;
    }
    const sy_lds_base_t lds0 = sy_lds_base(smem);
    const int ncs = p.Cin / BK;
    auto issue_pieces = [&](int cslab) {
        const unsigned s_x = (unsigned)(cslab * BK * ESZ);
        sy_static_for<0, NI>([&](auto i_) {
            constexpr int I = decltype(i_)::value;
            const bool dead = voff[I] == 0xFFFFFFFFu || cslab >= ncs;
            sy_glds16_buf_at(bufx, dead ? 0xFFFFFFFFu : voff[I] + s_x, lds0, (unsigned)((cslab & 1) * BUF + (wave + I * NW) * 1024));
        });
    };

    const sy_buffer buff = sy_make_buffer(p.wfrag, p.wfrag_extent);
    const int ntile32 = (p.Cout + 31) / 32;
    unsigned foff[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int ct = bid.x * (CT / 32) + wc * TC + t;
        foff[t] = ct < ntile32 ? (unsigned)((((long long)ct * ncs * 9) * 128 + lane) * 16) : 0xFFFFFFFFu;
    }
    uint4 fr[4][TC][2];
    auto fetch = [&](auto q_, int cslab) {                  // weight fragments of slot Q of slab `cslab`
        constexpr int Q = decltype(q_)::value;
        const unsigned s_f = (unsigned)((cslab * 9 + tap_w[Q]) * 2048);
        const bool live = cslab < ncs;
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                fr[Q][t][g] = sy_buffer_load16_s(buff, (live && foff[t] != 0xFFFFFFFFu) ? foff[t] + (unsigned)(g * 1024) : 0xFFFFFFFFu, s_f);
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    unsigned ba[4][TP];                                     // LDS byte offset of this lane's fragment row, k-half 0 (k-half 1 = ^ 32)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            const int row = (wp * TP + u) * kHaloW + l31 + tap_o[q];
            ba[q][u] = (unsigned)(row * 64 + ((half ^ ((row >> 2) & 3)) << 4));
        }

    // VMEM queue order per slab: the NI DMA pieces, then the fragments of the class's slots — at the top of a slab the pieces are older
    // than nt * 2 * TC fragment loads
    issue_pieces(0);
    sy_static_for<0, 4>([&](auto q_) { if (decltype(q_)::value < nt) fetch(q_, 0); sy_sched_fence(); });
    for (int cs = 0; cs < ncs; ++cs) {
        if (nt == 1) sy_wait_vmcnt<1 * 2 * TC>(); else if (nt == 2) sy_wait_vmcnt<2 * 2 * TC>(); else sy_wait_vmcnt<4 * 2 * TC>();
        sy_barrier();                                       // everybody's pieces; every wave is done reading the other buffer
        const unsigned hbo = (unsigned)((cs & 1) * BUF);
        issue_pieces(cs + 1);                               // (dead past the last slab: zeros into the idle buffer)
        sy_static_for<0, 4>([&](auto q_) {
            constexpr int Q = decltype(q_)::value;
            if (Q < nt) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    uint4 b[TP];
#pragma unroll
                    for (int u = 0; u < TP; ++u) b[u] = *reinterpret_cast<const uint4*>(smem + ((hbo + ba[Q][u]) ^ (unsigned)(g * 32)));
#pragma unroll
                    for (int t = 0; t < TC; ++t)
#pragma unroll
                        for (int u = 0; u < TP; ++u) acc[t][u] = sy_mfma_group(T(), fr[Q][t][g], b[u], acc[t][u]);
                }
                fetch(q_, cs + 1);
            }
            sy_sched_fence();
        });
    }
    sy_wait_vmcnt<0>();
    sy_barrier();

    SY_LATE_ARGS(ConvArgs, p);
    int e_bx = bid.x, e_n = n, e_i0 = i0, e_j0 = j0, e_by = bid.y, e_cls = cls;
    SY_LAUNDER_INT(e_bx); SY_LAUNDER_INT(e_n); SY_LAUNDER_INT(e_i0); SY_LAUNDER_INT(e_j0); SY_LAUNDER_INT(e_by); SY_LAUNDER_INT(e_cls);
    ClassPixels mp;
    mp.n = e_n; mp.i0 = e_i0; mp.j0 = e_j0; mp.ph = e_cls >> 1; mp.pw = e_cls & 1; mp.Ho = p_late.Ho; mp.Wo = p_late.Wo;
    mp.rep = e_by; mp.seg = 0;
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

// tile code 108: 4 waves x (32 channels x 2 class rows x 32 class pixels), the configuration of halo tile 117
template <typename T>
int launch_s2dgrad(const ConvArgs& a_in, void* stream) {
    constexpr int WC = 4, WP = 1, TC = 1, TP = 2;
    constexpr int NW = WC * WP, CT = WC * TC * 32, TH = WP * TP, PT = TH * 32;
    constexpr int HR = (TH + 1) * kHaloW, NI = ((HR + 15) / 16 + NW - 1) / NW, BUF = NW * NI * 16 * 64;
    ConvArgs a = a_in;
    a.s2_classes = 0;
    // launch input = dy [N, H, W, Cin] on the small map, output = dx [N, Ho, Wo, Cout] with H = ceil(Ho / 2), W = ceil(Wo / 2)
    if (a.KH != 3 || a.KW != 3 || a.stride != 2 || a.pad != 1 || a.mode != SY_CONV_DGRAD || a.H != (a.Ho + 1) / 2 || a.W != (a.Wo + 1) / 2)
        return SY_ERR_UNSUPPORTED;
    if (a.Cin % (4 * T::kEPC) != 0 || a.x_extent == 0 || a.wfrag == nullptr || a.wfrag_extent == 0 || a.stat_sum != nullptr || a.ksplit > 1)
        return SY_ERR_UNSUPPORTED;
    constexpr size_t smem_k = 2 * (size_t)BUF;
    constexpr size_t smem_e = (size_t)EpiLds<WP, CT>::kStatBytes + (size_t)PT * (CT * 2 + 16) + (size_t)PT * 8;
    constexpr bool can_stage = (T::kEPC == 8 && smem_e <= StageLimit<WC, WP, TC, TP>::kBytes);
    constexpr size_t smem_s = (size_t)WP * CT * 8;
    constexpr size_t smem = (can_stage && smem_e > smem_k) ? smem_e : (smem_s > smem_k ? smem_s : smem_k);
    const int tiles = a.N * ((((a.Ho + 1) >> 1) + TH - 1) / TH) * ((((a.Wo + 1) >> 1) + 31) / 32);
    dim3 grid((a.Cout + CT - 1) / CT, tiles, 4);
#ifndef SY_EMU
    static sy_dev_once attr_done;
    if (attr_done.need()) {
        if (hipFuncSetAttribute((const void*)conv3x3_s2dgrad_kernel<T, WC, WP, TC, TP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
            hipSuccess)
            return SY_ERR_LAUNCH;
        attr_done.mark();
    }
#endif
    SY_LAUNCH((conv3x3_s2dgrad_kernel<T, WC, WP, TC, TP>), grid, dim3(NW * 64), smem, stream, a);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}


// ---- all four parity classes in ONE workgroup (tile codes 126 / 127; round 6) ------------------------------------------------------
// The kernel above gives every parity class its own workgroups (gridDim.z = class): the (TH + 1) x 34 window of dy is parked four
// times per pixel tile, a class-(0,0) workgroup runs 4 MFMAs per wave between two barriers and a class-(1,1) one 16 (the launch ends
// on the heavy class), and at 8 pairs the four stride-2 data gradients of a frame were 0.93 ms of the step ONE AT A TIME at 300-440
// TF/s — on the two frame chains, where nothing overlaps them but the other frame (launch timeline, profiles/r05: 141 us per launch).
// Every one of the nine taps belongs to exactly ONE class — tap (kh, kw) feeds class (ph, pw) = (kh != 1, kw != 1) from window
// offset (dh, dw) = (kh == 0, kw == 0) — so a workgroup that owns the SAME TH x 32 class pixels of all four classes (a dense 2 TH x 64
// block of dx) parks the window once and runs the full 3x3's 18 TP MFMAs per wave and slab, like the stride-1 window kernels: four
// accumulator sets (one per class), the window fragments of the four offsets read once per slab, 4x fewer L2 -> LDS bytes and
// barriers per MFMA, no light / heavy workgroups.  The epilogue is conv_epilogue once per class through the class's pixel mapper.
template <typename T, int WC, int WP, int TC, int TP, int OCC>
__global__ __launch_bounds__(WC * WP * 64, OCC) void conv3x3_s2dgrad4_kernel(ConvArgs p) {
    SY_TL_BEGIN(5 + 32);
    constexpr int NW = WC * WP;
    constexpr int EPC = T::kEPC;
    constexpr int ESZ = 16 / EPC;
    constexpr int BK = 4 * EPC;
    constexpr int CT = WC * TC * 32;
    constexpr int TH = WP * TP;
    constexpr int HR = (TH + 1) * kHaloW;                  // window rows x 34 columns (33 used)
    constexpr int NI = ((HR + 15) / 16 + NW - 1) / NW;     // DMA pieces per wave and slab
    constexpr int BUF = NW * NI * 16 * 64;
    static_assert(NW == 4, "four waves");

    SY_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sy_uniform(tid >> 6);
    const int wc = wave / WP;
    const int wp = wave % WP;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const sy_block_id bid = sy_xcd_block_id();
    const int Hc = (p.Ho + 1) >> 1, Wc = (p.Wo + 1) >> 1;            // class grid of the largest class
    const int tiles_w = (Wc + 31) >> 5, tiles_h = (Hc + TH - 1) / TH;
    const int tw = bid.y % tiles_w, th_ = (bid.y / tiles_w) % tiles_h, n = bid.y / (tiles_w * tiles_h);
    const int i0 = th_ * TH, j0 = tw * 32;

    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    unsigned voff[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = (wave + i * NW) * 16 + (lane >> 2);
        const int hy = r / kHaloW, hx = r - hy * kHaloW;
        const int h = i0 + hy, w = j0 + hx;                            // dy pixel (p.H x p.W = the SMALL map of this launch's input)
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        const bool ok = r < HR && h < p.H && w < p.W;
        voff[i] = ok ? (unsigned)((((long long)n * p.xbs + ((long long)h * p.W + w) * p.ldx) + chunk * EPC) * ESZ) : 0xFFFFFFFFu;
    }
    const sy_lds_base_t lds0 = sy_lds_base(smem);
    const int ncs = p.Cin / BK;
    auto issue_pieces = [&](int cslab) {
        const unsigned s_x = (unsigned)(cslab * BK * ESZ);
        sy_static_for<0, NI>([&](auto i_) {
            constexpr int I = decltype(i_)::value;
            const bool dead = voff[I] == 0xFFFFFFFFu || cslab >= ncs;
            sy_glds16_buf_at(bufx, dead ? 0xFFFFFFFFu : voff[I] + s_x, lds0, (unsigned)((cslab & 1) * BUF + (wave + I * NW) * 1024));
        });
    };

    const sy_buffer buff = sy_make_buffer(p.wfrag, p.wfrag_extent);
    const int ntile32 = (p.Cout + 31) / 32;
    unsigned foff[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int ct = bid.x * (CT / 32) + wc * TC + t;
        foff[t] = ct < ntile32 ? (unsigned)((((long long)ct * ncs * 9) * 128 + lane) * 16) : 0xFFFFFFFFu;
    }
    uint4 fr[9][TC][2];                                     // weight fragments of the nine taps, refilled one slab ahead behind each tap's MFMAs
    auto fetch = [&](auto tap_, int cslab) {
        constexpr int TAP = decltype(tap_)::value;
        const unsigned s_f = (unsigned)((cslab * 9 + TAP) * 2048);
        const bool live = cslab < ncs;
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                fr[TAP][t][g] = sy_buffer_load16_s(buff, (live && foff[t] != 0xFFFFFFFFu) ? foff[t] + (unsigned)(g * 1024) : 0xFFFFFFFFu, s_f);
    };

    f32x16 acc[4][TC][TP];                                  // [class ph * 2 + pw]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int u = 0; u < TP; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][t][u][r] = 0.0f;

    unsigned ba[4][TP];                                     // LDS byte offset of this lane's fragment row at window offset o = dh * 2 + dw, k-half 0
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            const int row = (wp * TP + u) * kHaloW + l31 + (o >> 1) * kHaloW + (o & 1);
            ba[o][u] = (unsigned)(row * 64 + ((half ^ ((row >> 2) & 3)) << 4));
        }

    // VMEM queue order per slab: the NI DMA pieces, then the 9 x 2 TC fragment loads of the slab — at the top of a slab the pieces are
    // older than those 18 TC loads
    issue_pieces(0);
    sy_static_for<0, 9>([&](auto tap_) { fetch(tap_, 0); });
    sy_sched_fence();
    for (int cs = 0; cs < ncs; ++cs) {
        sy_wait_vmcnt<9 * 2 * TC>();
        sy_barrier();                                       // everybody's pieces; every wave is done reading the other buffer
        const unsigned hbo = (unsigned)((cs & 1) * BUF);
        issue_pieces(cs + 1);                               // (dead past the last slab: zeros into the idle buffer)
        // window offset o = (dh, dw): its fragments are read once and feed every tap that reads the window there
        sy_static_for<0, 4>([&](auto o_) {
            constexpr int O = decltype(o_)::value, DH = O >> 1, DW = O & 1;
            uint4 b[2][TP];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int u = 0; u < TP; ++u) b[g][u] = *reinterpret_cast<const uint4*>(smem + ((hbo + ba[O][u]) ^ (unsigned)(g * 32)));
            sy_static_for<0, 9>([&](auto tap_) {
                constexpr int TAP = decltype(tap_)::value, KH = TAP / 3, KW = TAP % 3;
                if constexpr ((KH == 0 ? 1 : 0) == DH && (KW == 0 ? 1 : 0) == DW) {
                    constexpr int CLS = (KH != 1 ? 2 : 0) + (KW != 1 ? 1 : 0);
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int t = 0; t < TC; ++t)
#pragma unroll
                            for (int u = 0; u < TP; ++u) acc[CLS][t][u] = sy_mfma_group(T(), fr[TAP][t][g], b[g][u], acc[CLS][t][u]);
                    fetch(tap_, cs + 1);
                }
            });
            sy_sched_fence();
        });
    }
    sy_wait_vmcnt<0>();
    sy_barrier();

    SY_LATE_ARGS(ConvArgs, p);
    int e_bx = bid.x, e_n = n, e_i0 = i0, e_j0 = j0, e_by = bid.y;
    SY_LAUNDER_INT(e_bx); SY_LAUNDER_INT(e_n); SY_LAUNDER_INT(e_i0); SY_LAUNDER_INT(e_j0); SY_LAUNDER_INT(e_by);
    // ONE copy of the epilogue's code for the four classes (four inlined copies were 32 000 instructions, each run once per workgroup:
    // instruction-cache misses instead of work): the class's accumulators are moved into `cur` by a uniform switch
    for (int cls = 0; cls < 4; ++cls) {
        f32x16 cur[TC][TP];
        sy_static_for<0, 4>([&](auto c_) {
            constexpr int CLS = decltype(c_)::value;
            if (cls == CLS) {
#pragma unroll
                for (int t = 0; t < TC; ++t)
#pragma unroll
                    for (int u = 0; u < TP; ++u) cur[t][u] = acc[CLS][t][u];
            }
        });
        ClassPixels mp;
        mp.n = e_n; mp.i0 = e_i0; mp.j0 = e_j0; mp.ph = cls >> 1; mp.pw = cls & 1; mp.Ho = p_late.Ho; mp.Wo = p_late.Wo;
        mp.rep = e_by; mp.seg = 0;
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            int n_, rem_;
            if (!mp.map((wp * TP + u) * 32 + l31, n_, rem_)) {
#pragma unroll
                for (int t = 0; t < TC; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[t][u][r] = 0.0f;
            }
        }
        // (conv_epilogue opens with a workgroup barrier when it stages: every wave has finished the previous class's write-out pass
        //  before the staging rows are overwritten)
        conv_epilogue<T, WC, WP, TC, TP>(p_late, mp, e_bx, cur, smem, tid);
    }
    SY_TL_END();
}

// tile codes 126: 4 waves x 32 channels (128) x 2 class rows x 32 class pixels x 4 classes — 128 accumulator registers + the nine
// taps' fragments + the epilogue's `cur`: one wave per SIMD (within 256 registers the compiler spilled INSIDE the slab loop); 125: the
// same on ONE class row (64 accumulators, two waves per SIMD, twice the weight-fragment loads per MFMA); 127: 2 x 2 waves, 64 channels
// (the 64-channel gradient of dark2's stride-2 conv, where the 128-channel tiles would idle half their waves)
template <typename T, int WC, int WP, int TC, int TP, int OCC>
int launch_s2dgrad4_cfg(const ConvArgs& a_in, void* stream) {
    constexpr int NW = WC * WP, CT = WC * TC * 32, TH = WP * TP, PT = TH * 32;
    constexpr int HR = (TH + 1) * kHaloW, NI = ((HR + 15) / 16 + NW - 1) / NW, BUF = NW * NI * 16 * 64;
    ConvArgs a = a_in;
    a.s2_classes = 0;
    constexpr size_t smem_k = 2 * (size_t)BUF;
    constexpr size_t smem_e = (size_t)EpiLds<WP, CT>::kStatBytes + (size_t)PT * (CT * 2 + 16) + (size_t)PT * 8;
    constexpr bool can_stage = (T::kEPC == 8 && smem_e <= StageLimit<WC, WP, TC, TP>::kBytes);
    constexpr size_t smem_s = (size_t)WP * CT * 8;
    constexpr size_t smem = (can_stage && smem_e > smem_k) ? smem_e : (smem_s > smem_k ? smem_s : smem_k);
    const int tiles = a.N * ((((a.Ho + 1) >> 1) + TH - 1) / TH) * ((((a.Wo + 1) >> 1) + 31) / 32);
    dim3 grid((a.Cout + CT - 1) / CT, tiles, 1);
#ifndef SY_EMU
    static sy_dev_once attr_done;
    if (attr_done.need()) {
        if (hipFuncSetAttribute((const void*)conv3x3_s2dgrad4_kernel<T, WC, WP, TC, TP, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
            hipSuccess)
            return SY_ERR_LAUNCH;
        attr_done.mark();
    }
#endif
    SY_LAUNCH((conv3x3_s2dgrad4_kernel<T, WC, WP, TC, TP, OCC>), grid, dim3(NW * 64), smem, stream, a);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

template <typename T>
int launch_s2dgrad4(const ConvArgs& a, void* stream) {
    // launch input = dy [N, H, W, Cin] on the small map, output = dx [N, Ho, Wo, Cout] with H = ceil(Ho / 2), W = ceil(Wo / 2)
    if (a.KH != 3 || a.KW != 3 || a.stride != 2 || a.pad != 1 || a.mode != SY_CONV_DGRAD || a.H != (a.Ho + 1) / 2 || a.W != (a.Wo + 1) / 2)
        return SY_ERR_UNSUPPORTED;
    if (a.Cin % (4 * T::kEPC) != 0 || a.x_extent == 0 || a.wfrag == nullptr || a.wfrag_extent == 0 || a.stat_sum != nullptr || a.ksplit > 1)
        return SY_ERR_UNSUPPORTED;
    if (a.tile == 127) return launch_s2dgrad4_cfg<T, 2, 2, 1, 1, 2>(a, stream);
    if (a.tile == 125) return launch_s2dgrad4_cfg<T, 4, 1, 1, 1, 2>(a, stream);
    return launch_s2dgrad4_cfg<T, 4, 1, 1, 2, 1>(a, stream);
}

}  // namespace sy_conv
