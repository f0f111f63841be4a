// conv1x1_tile.h — 1x1 stride-1 convolution (forward and data gradient) with the WHOLE K extent of the tile in flight at once.
//
// Replaces the same reference code as conv_igemm_impl.h for the 1x1 layers (yolox BaseConv's Conv2d of CSPLayer conv1/2/3,
// Bottleneck conv1, the lateral / reduce convs of the PAFPN and the head stems — exps/model/darknet.py:118-165 via CSPLayer,
// dfp_pafpn.py:33-105, tal_head.py:55-104 — and cuDNN backward-data): 40 % of the launches of a step.
//
// Why a third kernel: at K = Cin <= 512 the implicit-GEMM loop is 2-16 slabs long, and every slab costs a barrier, ~50 scalar
// instructions of generic (tap, channel) bookkeeping and a global-load latency that only one or two slabs of MFMAs cover —
// the ISA shows `s_waitcnt vmcnt(0)` in front of every 4 MFMAs.  A 1x1 tile needs no K loop at all:
//   * pixel operand (MFMA B): the tile's PT pixel rows x Cin channels are requested from L2 in ONE burst of LDS-DMA pieces
//     (16 rows x 64 B each, slab-major LDS image [slab][pixel][64 B] with the halo kernel's source-side XOR swizzle), every
//     wave issuing its share up front;
//   * weight operand (MFMA A): the wave's 32 x TC output channels x ALL of K straight from the fragment-packed layout into
//     VGPRs (8 x TC x Cin/32 registers), requested in the same burst, slab by slab between the pieces;
//   * NG barrier groups: the wave waits (counted — VMEM retires in order) for the pieces of group g only, the fragments and
//     pieces of the later groups stay in flight while group g's MFMAs run; NG = 1 for short K;
//   * inside a group the pixel fragments are read from LDS two (slab, k-half) steps ahead of the MFMAs (3-deep register ring).
// Epilogue: conv_epilogue of conv_igemm_impl.h through the LinearPixels mapper — BN statistics, affine / SiLU / residual /
// decode, staged coalesced write-out, `+=` — so training forward, data gradient and the eval convs all qualify.
#pragma once
#include "conv_igemm_impl.h"

namespace sy_conv {

// NCH > 1 (Cin = NCH x 512): the K extent is walked in NCH chunks of NS = 16 slabs through the same LDS image and weight
// registers (accumulators persist; a barrier after a chunk's MFMAs frees the image for the next burst).
template <typename T, int WC, int WP, int TC, int TP, int NS, int NCH = 1>
__global__ __launch_bounds__(WC * WP * 64, (TC * TP <= 2 && NS <= 8 ? 3 : 2)) void conv1x1_tile_kernel(ConvArgs p) {
    SY_TL_BEGIN(4 + (p.mode == SY_CONV_DGRAD ? 32 : 0));
    constexpr int NW = WC * WP;
    constexpr int EPC = T::kEPC;
    constexpr int ESZ = 16 / EPC;
    constexpr int BK = 4 * EPC;                  // channels per 64-byte slab
    constexpr int CT = WC * TC * 32;
    constexpr int PT = WP * TP * 32;
    constexpr int PPS = PT / 16;                 // DMA pieces (16 pixel rows x 64 B) per slab
    constexpr int NPIECE = NS * PPS;
    static_assert(NW == 4, "4 waves");
    static_assert(PPS % NW == 0, "every wave issues the same number of pieces per slab");
    constexpr int PW = PPS / NW;                 // pieces per wave per slab
    constexpr int NG = NS >= 8 ? 2 : 1;          // barrier groups
    constexpr int SPG = NS / NG;                 // slabs per group
    constexpr int BD = 3;                        // pixel-fragment register ring
    static_assert(NS % NG == 0, "whole groups");

    SY_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sy_uniform(tid >> 6);
    const int wc = wave / WP;
    const int wp = wave % WP;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const sy_block_id bid = sy_xcd_block_id();   // x: channel tile (fastest: the channel tiles of a pixel tile share its rows in L2)
    sy_probe(0);
    const int HW = p.HoWo;
    const int m_end = p.seg_M > 0 ? (bid.z + 1) * p.seg_M : p.M;
    const int m0 = bid.z * (p.seg_M > 0 ? p.seg_M : 0) + bid.y * PT;

    // ---- the burst: piece j = slab j / PPS, pixel rows [16 (j % PPS), +16); wave w issues pieces w, w + NW, ... of every slab
    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    const sy_buffer buff = sy_make_buffer(p.wfrag, p.wfrag_extent);
    const sy_lds_base_t lds0 = sy_lds_base(smem);
    unsigned voff[PW];
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int r = (wave + i * NW) * 16 + (lane >> 2);
        const int m = m0 + r;
        const int n = m / HW, rem = m - n * HW;
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        const bool ok = m < m_end && !(p.ablate & 1);
        voff[i] = ok ? (unsigned)(((long long)n * p.xbs + (long long)rem * p.ldx + chunk * EPC) * ESZ) : 0xFFFFFFFFu;
    }
    const int ntile32 = (p.Cout + 31) / 32;
    unsigned foff[TC];
#pragma unroll
    for (int t = 0; t < TC; ++t) {
        const int ct = bid.x * (CT / 32) + wc * TC + t;
        foff[t] = (ct < ntile32 && !(p.ablate & 2)) ? (unsigned)((((long long)ct * NS * NCH) * 128 + lane) * 16) : 0xFFFFFFFFu;
    }
    uint4 fr[NS][TC][2];
    auto burst = [&](int chunk) {
        const unsigned cx = (unsigned)(chunk * NS * BK * ESZ), cf = (unsigned)(chunk * NS * 2048);
        sy_static_for<0, NS>([&](auto s_) {
            constexpr int S = decltype(s_)::value;
#pragma unroll
            for (int i = 0; i < PW; ++i)
                sy_glds16_buf_at(bufx, voff[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : voff[i] + cx + (unsigned)(S * BK * ESZ), lds0,
                                 (unsigned)((S * PPS + wave + i * NW) * 1024));
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    fr[S][t][g] = sy_buffer_load16_s(buff, foff[t] == 0xFFFFFFFFu ? 0xFFFFFFFFu : foff[t] + (unsigned)(g * 1024), cf + (unsigned)(S * 2048));
            sy_sched_fence();                    // slab order: the counted waits below (and the compiler's own) rely on it
        });
    };
    burst(0);
    sy_probe(1);

    f32x16 acc[TC][TP];
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    // LDS byte offset of this lane's fragment row inside a slab image, k-half 0 (k-half 1 = ^ 32)
    unsigned ba[TP];
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        const int row = (wp * TP + u) * 32 + l31;
        ba[u] = (unsigned)(row * 64 + ((half ^ ((row >> 2) & 3)) << 4));
    }

    for (int chunk = 0; chunk < NCH; ++chunk) {
    if (chunk > 0) {
        sy_barrier();                             // every wave is done with the previous chunk's image
        burst(chunk);
    }
    sy_static_for<0, NG>([&](auto g_) {
        constexpr int GI = decltype(g_)::value;
        // VMEM operations issued after the last piece of this group: its own last slab's fragments + everything of the later slabs
        sy_wait_vmcnt<2 * TC + (NS - (GI + 1) * SPG) * (PW + 2 * TC)>();
        sy_barrier();
        if constexpr (GI == 0) sy_probe(2);
        uint4 b[BD][TP];
        auto read_step = [&](auto s_) {          // step = (slab, k-half) inside the group
            constexpr int ST = decltype(s_)::value;
            constexpr int S = GI * SPG + (ST >> 1), G = ST & 1;
#pragma unroll
            for (int u = 0; u < TP; ++u)
                b[ST % BD][u] = *reinterpret_cast<const uint4*>(smem + S * PT * 64 + (ba[u] ^ (unsigned)(G * 32)));
        };
        sy_static_for<0, BD - 1>([&](auto s_) { read_step(s_); });
        sy_static_for<0, 2 * SPG>([&](auto s_) {
            constexpr int ST = decltype(s_)::value;
            constexpr int S = GI * SPG + (ST >> 1), G = ST & 1;
            if constexpr (ST + BD - 1 < 2 * SPG) read_step(sy_int<ST + BD - 1>());
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int u = 0; u < TP; ++u) acc[t][u] = sy_mfma_group(T(), fr[S][t][G], b[ST % BD][u], acc[t][u]);
            sy_sched_fence();
        });
    });
    }

    sy_probe(3);
    SY_LATE_ARGS(ConvArgs, p);
    int e_bx = bid.x, e_by = bid.y, e_bz = bid.z;
    SY_LAUNDER_INT(e_bx); SY_LAUNDER_INT(e_by); SY_LAUNDER_INT(e_bz);
    const LinearPixels mp(p_late, e_by, e_bz, PT);
    conv_epilogue<T, WC, WP, TC, TP>(p_late, mp, e_bx, acc, smem, tid);
    sy_probe(6);
    SY_TL_END();
}

// (A persistent variant — weights loaded once per workgroup, pixel tiles walked through two tile buffers, BatchNorm sums carried
//  in registers and reduced once — was built, parity-green and 10-25 % SLOWER on every layer but one
//  (profiles/r02/x_conv_probe_persistent_1x1.txt: three barriers per tile and half the resident workgroups cost more than the
//  saved prologue / statistics work: these kernels live on latency hiding across workgroups, not on instruction count.)

template <typename T, int WC, int WP, int TC, int TP, int NS, int NCH = 1>
int launch_1x1_tile_ns(const ConvArgs& a_in, void* stream) {
    constexpr int CT = WC * TC * 32, PT = WP * TP * 32;
    ConvArgs a = a_in;
    a.s2_classes = 0;
    const int nseg = a.seg_M > 0 ? a.M / a.seg_M : 1;
    dim3 grid((a.Cout + CT - 1) / CT, ((a.seg_M > 0 ? a.seg_M : a.M) + PT - 1) / PT, nseg);
    constexpr size_t smem_k = (size_t)NS * PT * 64;
    constexpr size_t smem_e = (size_t)EpiLds<WP, CT>::kStatBytes + (size_t)PT * (CT * 2 + 16) + (size_t)PT * 8;
    constexpr size_t smem_s = (size_t)WP * CT * 8;
    constexpr size_t smem = (smem_e <= 48 * 1024 && smem_e > smem_k) ? smem_e : (smem_s > smem_k ? smem_s : smem_k);
#ifndef SY_EMU
    static sy_dev_once attr_done;
    if (attr_done.need()) {
        if (hipFuncSetAttribute((const void*)conv1x1_tile_kernel<T, WC, WP, TC, TP, NS, NCH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return SY_ERR_LAUNCH;
        attr_done.mark();
    }
#endif
    SY_LAUNCH((conv1x1_tile_kernel<T, WC, WP, TC, TP, NS, NCH>), grid, dim3(WC * WP * 64), smem, stream, a);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

template <typename T, int WC, int WP, int TC, int TP>
int launch_1x1_tile_cfg(const ConvArgs& a, void* stream) {
    switch (a.Cin / (4 * T::kEPC)) {
        case 2: return launch_1x1_tile_ns<T, WC, WP, TC, TP, 2>(a, stream);
        case 4: return launch_1x1_tile_ns<T, WC, WP, TC, TP, 4>(a, stream);
        case 8: return launch_1x1_tile_ns<T, WC, WP, TC, TP, 8>(a, stream);
        case 16:
            if constexpr (TC * TP <= 2) return launch_1x1_tile_ns<T, WC, WP, TC, TP, 16>(a, stream);
            return SY_ERR_UNSUPPORTED;
        case 32:                                  // Cin 1024 / 2048: two / four chunks of 512 channels
            if constexpr ((WC == 4 || TC * TP == 1) && TC * TP <= 2) return launch_1x1_tile_ns<T, WC, WP, TC, TP, 16, 2>(a, stream);
            return SY_ERR_UNSUPPORTED;
        case 64:
            if constexpr ((WC == 4 || TC * TP == 1) && TC * TP <= 2) return launch_1x1_tile_ns<T, WC, WP, TC, TP, 16, 4>(a, stream);
            return SY_ERR_UNSUPPORTED;
        default: return SY_ERR_UNSUPPORTED;
    }
}

// tile codes 121..124 (SY_TILE_1X1K + k)
template <typename T>
int launch_1x1_tile(const ConvArgs& a, void* stream) {
    if constexpr (T::kEPC != 8) {
        return SY_ERR_UNSUPPORTED;
    } else {
        // 1x1 stride 1 (forward and data gradient are the same gather), whole 64-byte channel slabs, 32-bit addressable input,
        // fragment-packed weights, Cin in {64, 128, 256, 512, 1024, 2048}
        if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.Ho != a.H || a.Wo != a.W) return SY_ERR_UNSUPPORTED;
        if (a.Cin % 32 != 0 || a.x_extent == 0 || a.wfrag == nullptr || a.wfrag_extent == 0) return SY_ERR_UNSUPPORTED;
        switch (a.tile) {
            case 121: return launch_1x1_tile_cfg<T, 4, 1, 1, 2>(a, stream);     // 128 ch x  64 px
            case 122: return launch_1x1_tile_cfg<T, 2, 2, 1, 2>(a, stream);     //  64 ch x 128 px
            case 123: return launch_1x1_tile_cfg<T, 4, 1, 1, 4>(a, stream);     // 128 ch x 128 px
            // small launches with a long K (the 19x30 maps' 1024 / 2048-channel layers of a streamed frame): half the weight bytes
            // per workgroup, twice the workgroups (at Cin <= 512 it measured the same as 121: not a candidate there)
            case 124: return launch_1x1_tile_cfg<T, 2, 2, 1, 1>(a, stream);     //  64 ch x  64 px
            default: return SY_ERR_ARG;
        }
    }
}

}  // namespace sy_conv
