// bottleneck_fused.h — yolox Bottleneck forward (eval) as ONE launch: conv1 (1x1) -> BatchNorm -> SiLU -> conv2 (3x3) -> BatchNorm
// -> SiLU (+ x), with conv1's output kept in LDS (tile code 119 of sy_conv2d, sy_conv_desc::pre_*).
//
// Replaces, for inference / streaming plans, two BaseConv launches of the reference's Bottleneck (yolox network_blocks.Bottleneck
// inside CSPLayer — call sites exps/model/darknet.py:118-165, dfp_pafpn.py:33-81) and the HBM round trip of the
// hidden activation between them.  Training keeps the two launches: batch statistics need a global reduction between the
// convolution and its activation.
//
// A workgroup (4 waves, 8 for layers wider than 128 channels) owns an output tile of (32 x waves) channels x 2 rows x 32 pixels of
// one image (tile 117's shape).
//   stage 1  the (2 + 2) x 34 halo window of the Bottleneck's INPUT x, all K1 channels, is parked in LDS once (LDS-DMA, slab-major
//            image [K1 / 32][144 rows][64 B], the halo kernels' source-side XOR swizzle); the four waves share the hidden channels
//            (32-channel tiles round robin) and compute h = silu(scale1 * (W1 x) + shift1) for the 136 window pixels (five 32-pixel
//            MFMA tiles, fragment-packed W1 straight from L2 into VGPRs), round to the storage dtype and write h into a second
//            slab-major LDS image in exactly the layout stage 2 reads — zeros for window pixels outside the image (the 3x3
//            convolution's zero padding applies to h, not to x).
//   stage 2  conv3x3_halo2_kernel's main loop over the RESIDENT slabs of h: nine taps = nine pixel offsets of the same ds_read_b128
//            fragment read, W2 fragments prefetched one slab ahead; no DMA, no per-slab barrier.
//   epilogue conv_epilogue (affine, SiLU, residual x, staged coalesced write-out).
// The window is recomputed per output-channel tile (512-channel layers: twice) and its 2 x 34 halo makes stage 1 do 2.1x the
// 1x1 layer's products: the kernel trades MFMA work for a launch and an HBM round trip, which pays where the step is bound by
// dispatch and latency (the batch-1 streaming step: 72 workgroups of a 38 x 60 map on a 256-CU chip) — the plan's tuner decides per
// layer against the two separate launches (ops.tuned_bottleneck).  LDS: (K1 + hidden) / 32 x 9 KiB, so K1 + hidden <= 544.
#pragma once
#include "conv3x3_halo.h"

namespace sy_conv {

constexpr int kBnkRows = 144;                    // LDS rows per slab: nine 16-row DMA pieces cover the 136 window pixels
constexpr int kBnkSlab = kBnkRows * 64;          // bytes per 32-channel slab image
constexpr int kBnkHR = 4 * kHaloW;               // window pixels (TH + 2 = 4 rows x 34)

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 2 : 1)) void bottleneck_fused_kernel(ConvArgs p) {
    SY_TL_BEGIN(16);
    constexpr int TC = 1, TP = 2, TH = 2, CT = NW * 32;
    constexpr int NPW = (9 + NW - 1) / NW;        // DMA pieces per wave per slab
    constexpr int PF = 8;                         // stage 1: W1 fragments in flight per wave (L2 latency >> the 5 MFMAs of a step)
    constexpr int EPC = T::kEPC;
    constexpr int ESZ = 16 / EPC;
    constexpr int BK = 4 * EPC;
    constexpr int BD = 3;
    static_assert(EPC == 8, "16-bit storage types");

    SY_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sy_uniform(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const sy_block_id bid = sy_xcd_block_id();
    const int tiles_w = (p.Wo + 31) >> 5, tiles_h = (p.Ho + TH - 1) / TH;
    const int tw = bid.y % tiles_w, th_ = (bid.y / tiles_w) % tiles_h, n = bid.y / (tiles_w * tiles_h);
    const int h0 = th_ * TH, w0 = tw * 32;
    const int n1 = p.pre_cin / BK;               // slabs of the input x
    const int nh = p.Cin / BK;                   // slabs of the hidden activation (= the 3x3 convolution's input channels)
    unsigned char* const sX = smem;
    unsigned char* const sH = smem + n1 * kBnkSlab;
    const sy_lds_base_t lds0 = sy_lds_base(smem);

    // ---- stage 1a: the window of x, every slab, in one burst.  Piece j (16 rows) of slab s; wave w issues pieces w, w + NW, ...
    {
        const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
        unsigned voff[NPW];
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int piece = wave + i * NW;
            const int r = piece * 16 + (lane >> 2);
            const int hy = r / kHaloW, hx = r - hy * kHaloW;
            const int h = h0 - 1 + hy, w = w0 - 1 + hx;
            const int chunk = (lane & 3) ^ ((r >> 2) & 3);
            const bool ok = piece < 9 && r < kBnkHR && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W;
            voff[i] = ok ? (unsigned)((((long long)n * p.xbs + ((long long)h * p.W + w) * p.ldx) + chunk * EPC) * ESZ) : 0xFFFFFFFFu;
        }
        for (int s = 0; s < n1; ++s) {
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                const int piece = wave + i * NW;
                if (piece < 9)                    // (wave-uniform)
                    sy_glds16_buf_at(bufx, voff[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : voff[i] + (unsigned)(s * BK * ESZ), lds0,
                                     (unsigned)(s * kBnkSlab + piece * 1024));
            }
        }
    }
    // which window pixels lie inside the image (lane = pixel column of an MFMA tile): bit pt of `inside`
    unsigned inside = 0;
#pragma unroll
    for (int pt = 0; pt < 5; ++pt) {
        const int r = pt * 32 + l31;
        const int hy = r / kHaloW, hx = r - hy * kHaloW;
        const int h = h0 - 1 + hy, w = w0 - 1 + hx;
        if (r < kBnkHR && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W) inside |= 1u << pt;
    }
    sy_wait_vmcnt<0>();
    __syncthreads();                              // the window has landed, for every wave

    // ---- stage 1b: h = silu(scale1 * (W1 x) + shift1), hidden-channel tile ct = wave, wave + 4, ...
    {
        const sy_buffer bufw1 = sy_make_buffer(p.pre_w, p.pre_w_extent);
        for (int ct = wave; ct < nh; ct += NW) {
            f32x16 a1[5];
#pragma unroll
            for (int pt = 0; pt < 5; ++pt)
#pragma unroll
                for (int r = 0; r < 16; ++r) a1[pt][r] = 0.0f;
            const unsigned fbase = (unsigned)((((long long)ct * n1) * 128 + lane) * 16);
            const int steps = 2 * n1;                              // step = (slab, k-half); fragments requested PF steps ahead
            uint4 fa[PF];
#pragma unroll
            for (int j = 0; j < PF; ++j) fa[j] = sy_buffer_load16_s(bufw1, j < steps ? fbase : 0xFFFFFFFFu, (unsigned)(j * 1024));
            for (int base = 0; base < steps; base += PF) {
                sy_static_for<0, PF>([&](auto j_) {
                    constexpr int J = decltype(j_)::value;
                    const int st = base + J;
                    if (st < steps) {                              // (uniform)
                        const uint4 cur = fa[J];
                        fa[J] = sy_buffer_load16_s(bufw1, st + PF < steps ? fbase : 0xFFFFFFFFu, (unsigned)((st + PF) * 1024));
                        const int s = st >> 1, g = st & 1;
#pragma unroll
                        for (int pt = 0; pt < 5; ++pt) {
                            const int row = pt * 32 + l31;
                            const unsigned off = (unsigned)(s * kBnkSlab + row * 64 + ((half ^ ((row >> 2) & 3)) << 4)) ^ (unsigned)(g * 32);
                            const uint4 b = *reinterpret_cast<const uint4*>(sX + off);
                            a1[pt] = sy_mfma_group(T(), cur, b, a1[pt]);
                        }
                    }
                });
            }
            // affine + SiLU of this lane's 16 channels (4 groups of 4 consecutive ones), rounded, into the hidden image
            float sc[16], sh[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 s4 = *reinterpret_cast<const float4*>(p.pre_scale + ct * 32 + q * 8 + half * 4);
                const float4 h4 = *reinterpret_cast<const float4*>(p.pre_shift + ct * 32 + q * 8 + half * 4);
                sc[q * 4 + 0] = s4.x; sc[q * 4 + 1] = s4.y; sc[q * 4 + 2] = s4.z; sc[q * 4 + 3] = s4.w;
                sh[q * 4 + 0] = h4.x; sh[q * 4 + 1] = h4.y; sh[q * 4 + 2] = h4.z; sh[q * 4 + 3] = h4.w;
            }
#pragma unroll
            for (int pt = 0; pt < 5; ++pt) {
                const int row = pt * 32 + l31;
                if (row >= kBnkRows) continue;                     // beyond the slab image (rows 136 .. 143: padding rows, written as zeros)
                const bool in = (inside >> pt) & 1u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = in ? sy_silu(a1[pt][q * 4 + j] * sc[q * 4 + j] + sh[q * 4 + j]) : 0.0f;
                    unsigned char* const dst = sH + ct * kBnkSlab + row * 64 + ((q ^ ((row >> 2) & 3)) << 4) + half * 8;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(T::pack2(v[0], v[1]), T::pack2(v[2], v[3]));
                }
            }
        }
    }
    __syncthreads();                              // the hidden image is complete

    // ---- stage 2: 3x3 over the resident slabs (conv3x3_halo2_kernel's pipeline without its DMA)
    const sy_buffer buff = sy_make_buffer(p.wfrag, p.wfrag_extent);
    const int ntile32 = (p.Cout + 31) / 32;
    const int ct2 = bid.x * (CT / 32) + wave;
    const unsigned foff = ct2 < ntile32 ? (unsigned)((((long long)ct2 * nh * 9) * 128 + lane) * 16) : 0xFFFFFFFFu;
    uint4 fr[9][2];
    auto fetch = [&](auto tap_, int cslab) {
        constexpr int TAP = decltype(tap_)::value;
        const unsigned s_f = (unsigned)((cslab * 9 + TAP) * 2048);
        const bool live = cslab < nh && foff != 0xFFFFFFFFu;
#pragma unroll
        for (int g = 0; g < 2; ++g) fr[TAP][g] = sy_buffer_load16_s(buff, live ? foff + (unsigned)(g * 1024) : 0xFFFFFFFFu, s_f);
    };
    f32x16 acc[TC][TP];
#pragma unroll
    for (int u = 0; u < TP; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][u][r] = 0.0f;
    unsigned ba[9][TP];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int toff = (tap / 3) * kHaloW + (tap % 3);
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            const int row = u * kHaloW + l31 + toff;
            ba[tap][u] = (unsigned)(row * 64 + ((half ^ ((row >> 2) & 3)) << 4));
        }
    }
    sy_static_for<0, 9>([&](auto t_) { fetch(t_, 0); sy_sched_fence(); });
    for (int cs = 0; cs < nh; ++cs) {
        const unsigned char* const hb = sH + cs * kBnkSlab;
        uint4 b[BD][TP];
        auto read_step = [&](auto s_) {
            constexpr int S = decltype(s_)::value;
            constexpr int TAP = S >> 1, G = S & 1;
#pragma unroll
            for (int u = 0; u < TP; ++u) b[S % BD][u] = *reinterpret_cast<const uint4*>(hb + (ba[TAP][u] ^ (unsigned)(G * 32)));
        };
        sy_static_for<0, BD - 1>([&](auto s_) { read_step(s_); });
        sy_static_for<0, 18>([&](auto s_) {
            constexpr int S = decltype(s_)::value;
            constexpr int TAP = S >> 1, G = S & 1;
            if constexpr (S + BD - 1 < 18) read_step(sy_int<S + BD - 1>());
#pragma unroll
            for (int u = 0; u < TP; ++u) acc[0][u] = sy_mfma_group(T(), fr[TAP][G], b[S % BD][u], acc[0][u]);
            if constexpr (G == 1) fetch(sy_int<TAP>(), cs + 1);
            sy_sched_fence();
        });
    }
    __syncthreads();                              // every wave is done with the hidden image: the epilogue stages through the same LDS

    SY_LATE_ARGS(ConvArgs, p);
    int e_bx = bid.x, e_n = n, e_h0 = h0, e_w0 = w0, e_by = bid.y;
    SY_LAUNDER_INT(e_bx); SY_LAUNDER_INT(e_n); SY_LAUNDER_INT(e_h0); SY_LAUNDER_INT(e_w0); SY_LAUNDER_INT(e_by);
    TilePixels mp;
    mp.n = e_n; mp.h0 = e_h0; mp.w0 = e_w0; mp.Ho = p_late.Ho; mp.Wo = p_late.Wo; mp.rep = e_by;
    mp.seg = 0;
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        int n_, rem_;
        if (!mp.map(u * 32 + l31, n_, rem_)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][u][r] = 0.0f;
        }
    }
    conv_epilogue<T, NW, 1, 1, 2>(p_late, mp, e_bx, acc, smem, tid);
    SY_TL_END();
}

template <typename T, int NW>
int launch_bottleneck_fused_nw(const ConvArgs& a_in, void* stream) {
    if constexpr (T::kEPC != 8) {
        return SY_ERR_UNSUPPORTED;
    } else {
        constexpr int CT = NW * 32, PT = 64, TH = 2;
        ConvArgs a = a_in;
        a.s2_classes = 0;
        if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W || a.mode != SY_CONV_FWD) return SY_ERR_UNSUPPORTED;
        if (a.pre_w == nullptr || a.pre_scale == nullptr || a.pre_shift == nullptr || a.pre_w_extent == 0) return SY_ERR_ARG;
        if (a.Cin % 32 != 0 || a.pre_cin % 32 != 0 || a.pre_cin <= 0 || a.x_extent == 0 || a.wfrag == nullptr || a.wfrag_extent == 0)
            return SY_ERR_UNSUPPORTED;
        if (a.stat_sum != nullptr || a.ksplit > 1 || a.accumulate || a.y_f32) return SY_ERR_UNSUPPORTED;
        const size_t smem_k = (size_t)(a.Cin / 32 + a.pre_cin / 32) * kBnkSlab;
        constexpr size_t smem_e = (size_t)EpiLds<1, CT>::kStatBytes + (size_t)PT * (CT * 2 + 16) + (size_t)PT * 8;
        if (smem_k > 156 * 1024) return SY_ERR_UNSUPPORTED;           // K1 + hidden <= 544 channels (160 KiB of LDS per CU)
        const size_t smem = smem_k > smem_e ? smem_k : smem_e;
        const int tiles = a.N * ((a.Ho + TH - 1) / TH) * ((a.Wo + 31) / 32);
        dim3 grid((a.Cout + CT - 1) / CT, tiles, 1);
#ifndef SY_EMU
        static sy_dev_once attr_done;
        if (attr_done.need()) {
            if (hipFuncSetAttribute((const void*)bottleneck_fused_kernel<T, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
                return SY_ERR_LAUNCH;
            attr_done.mark();
        }
#endif
        SY_LAUNCH((bottleneck_fused_kernel<T, NW>), grid, dim3(NW * 64), smem, stream, a);
        return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
    }
}

// 128 output channels per workgroup on four waves, or — wide layers — 256 on eight: the window (stage 1) is computed once per
// workgroup, so a 256-channel Bottleneck on the four-wave tile would compute it twice
template <typename T>
int launch_bottleneck_fused(const ConvArgs& a, void* stream) {
    return a.Cout > 128 ? launch_bottleneck_fused_nw<T, 8>(a, stream) : launch_bottleneck_fused_nw<T, 4>(a, stream);
}

}  // namespace sy_conv
