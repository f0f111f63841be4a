// conv1x1_stream.h — 1x1 stride-1 convolution (training forward with BatchNorm statistics, and data gradient) as a
// weight-stationary pixel stream.
//
// Replaces the same reference code as conv_igemm_impl.h (yolox BaseConv's Conv2d for the 1x1 layers: CSPLayer conv1 / conv2 /
// conv3 and Bottleneck conv1 — exps/model/darknet.py:118-165, dfp_pafpn.py:33-105 — and cuDNN backward-data), ~45 % of the
// convolution launches of a step.  These layers are HBM-bound (one read of the input, one write of the output, K <= 512),
// and the implicit-GEMM kernel ran them at 75-320 TF/s: one 64-128 pixel tile per workgroup, so every workgroup re-reads
// its 32-64 KB weight tile from L2, nothing is in flight during its prologue / epilogue, and each ends with 2 x CT atomics.
// Here (tile code SY_TILE_STREAM1X1):
//   * a workgroup = 4 waves x 32 output channels; each wave loads ITS weight rows once — all of K, in MFMA-fragment order
//     (the SY_TILE_WR layout) — into Cin / 4 VGPRs and keeps them for its whole life;
//   * the workgroup then streams a contiguous range of 64-pixel tiles: whole pixel rows go L2/HBM -> LDS by LDS-DMA (1 KiB
//     per wave instruction, fully coalesced; the 16-byte chunks of a row are XOR-permuted by (row & 15) on the source side so
//     the ds_read_b128 fragment reads are conflict free), two tile buffers, ONE barrier per tile;
//   * per tile every wave converts its 32 ch x 64 px accumulators, transposes them through a private 5 KB LDS patch and
//     writes 64 contiguous bytes per pixel; BatchNorm sum / sum^2 stay in registers across ALL tiles of the workgroup and
//     cost one shuffle reduction + 64 atomics per wave at the very end (a quarter of the atomics of the tiled kernel).
#pragma once
#include "conv_igemm_impl.h"

namespace sy_conv {

template <typename T, int NS>     // NS = Cin / 32 channel slabs (16-bit types): 2, 4, 8
__global__ __launch_bounds__(256, (NS <= 4 ? 3 : 2)) void conv1x1_stream_kernel(ConvArgs p) {
    typedef typename T::elem elem;
    constexpr int NW = 4, TP = 2, PT = TP * 32, CT = NW * 32;
    constexpr int ROWB = NS * 64;                 // bytes per pixel row in LDS (= Cin * 2)
    constexpr int CPRW = NS * 4;                  // 16-byte chunks per row
    constexpr int XM = (CPRW < 16 ? CPRW : 16) - 1;   // chunk permutation mask
    constexpr int TILEB = PT * ROWB;
    constexpr int NI = TILEB / 1024 / NW;         // DMA wave instructions per wave per tile
    constexpr int PATCH = 64 * 80;                // per-wave output transpose patch: 64 pixels x (64 B + 16 B pad)
    static_assert(T::kEPC == 8 && NI >= 1, "16-bit elements");
    SY_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = sy_uniform(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int c0 = blockIdx.x * CT;
    // pixel range of this workgroup: tiles [t0, t1) of statistics segment blockIdx.z
    const int seg_M = p.seg_M > 0 ? p.seg_M : p.M;
    const int seg0 = blockIdx.z * seg_M, seg_end = seg0 + seg_M;
    const int ntiles = (seg_M + PT - 1) / PT;
    const int per_wg = (ntiles + gridDim.y - 1) / gridDim.y;
    const int t0 = blockIdx.y * per_wg;
    int t1 = t0 + per_wg;
    if (t1 > ntiles) t1 = ntiles;
    if (t0 >= t1) return;                            // (never taken: the launcher sizes gridDim.y so that every workgroup owns a tile)

    // ---- this wave's weights: 32 output channels x all of K, fragment order, straight into registers
    const sy_buffer buff = sy_make_buffer(p.wfrag, p.wfrag_extent);
    uint4 fa[NS][2];
    {
        const int ct = blockIdx.x * NW + wave;
        const bool live = ct < (p.Cout + 31) / 32;
        const unsigned base = (unsigned)((((long long)ct * NS) * 128 + lane) * 16);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int g = 0; g < 2; ++g) fa[s][g] = sy_buffer_load16(buff, live ? base + (unsigned)((s * 2 + g) * 1024) : 0xFFFFFFFFu);
    }

    // ---- DMA assignment: instruction j = wave + i * NW covers LDS bytes [1024 j, 1024 j + 1024) of the tile; lane -> linear
    //      chunk q = 64 j + lane -> (row, physical chunk); it fetches logical chunk (physical ^ (row & XM)) of that pixel row
    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    const sy_lds_base_t lds0 = sy_lds_base(smem);
    int d_row[NI];
    unsigned d_off[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int q = (wave + i * NW) * 64 + lane;
        d_row[i] = q / CPRW;
        const int phys = q - d_row[i] * CPRW;
        d_off[i] = (unsigned)((d_row[i] * p.ldx + (phys ^ (d_row[i] & XM)) * 8) * 2);
    }
    auto issue_tile = [&](int t, int buf) {
        const int m0 = seg0 + t * PT;
        const unsigned s_x = (unsigned)((long long)m0 * p.ldx * 2);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            sy_glds16_buf_at(bufx, (m0 + d_row[i] < seg_end) ? d_off[i] + s_x : 0xFFFFFFFFu, lds0,
                             (unsigned)(buf * TILEB + (wave + i * NW) * 1024));
    };

    float ssum[16], ssq[16];
    const bool want_stats = p.stat_sum != nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
    unsigned char* const patch = smem + 2 * TILEB + wave * PATCH;

    issue_tile(t0, 0);
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        sy_wait_vmcnt<0>();                       // this wave's share of tile t has landed (and its earlier stores retired)
        sy_barrier();                             // ... everybody's; every wave is done reading the other buffer
        if (t + 1 < t1) issue_tile(t + 1, buf ^ 1);
        const unsigned char* const xb = smem + buf * TILEB;
        f32x16 acc[TP];
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint4 b[TP];
#pragma unroll
                for (int u = 0; u < TP; ++u) {
                    const int row = u * 32 + l31;
                    b[u] = *reinterpret_cast<const uint4*>(xb + row * ROWB + (((s * 4 + g * 2 + half) ^ (row & XM)) << 4));
                }
#pragma unroll
                for (int u = 0; u < TP; ++u) acc[u] = sy_mfma_group(T(), fa[s][g], b[u], acc[u]);
            }
        // ---- tile epilogue: statistics in registers, 16-bit conversion, transpose through the wave's LDS patch, 64 B per pixel
        const int m0 = seg0 + t * PT;
#pragma unroll
        for (int u = 0; u < TP; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = acc[u][q * 4 + j];
                    ssum[q * 4 + j] += v[j];
                    ssq[q * 4 + j] += v[j] * v[j];
                }
                *reinterpret_cast<uint2*>(patch + (u * 32 + l31) * 80 + (q * 8 + half * 4) * 2) =
                    make_uint2(T::pack2(v[0], v[1]), T::pack2(v[2], v[3]));
            }
        }
        sy_wave_fence();                          // wave-private patch: LDS operations of one wave complete in order, no barrier
        {
            const int m = m0 + lane;              // lane <-> pixel of the tile
            const int co = c0 + wave * 32;
            if (m < seg_end && co < p.Cout) {
                elem* dst = reinterpret_cast<elem*>(p.y) + (long long)m * p.ldy + co;
                const int nch = p.Cout - co < 32 ? p.Cout - co : 32;        // ragged last channel tile: whole 8-channel chunks
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k * 8 >= nch) break;
                    uint4 v = *reinterpret_cast<const uint4*>(patch + lane * 80 + k * 16);
                    if (p.accumulate) {
                        const uint4 o = *reinterpret_cast<const uint4*>(dst + k * 8);
                        elem ev[8], eo[8];
                        __builtin_memcpy(ev, &v, 16);
                        __builtin_memcpy(eo, &o, 16);
                        unsigned w[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            w[j] = T::pack2(T::to_f32(ev[2 * j]) + T::to_f32(eo[2 * j]), T::to_f32(ev[2 * j + 1]) + T::to_f32(eo[2 * j + 1]));
                        v = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                    *reinterpret_cast<uint4*>(dst + k * 8) = v;
                }
            }
        }
        sy_wave_fence();                          // the patch is rewritten by the next tile
    }
    if (want_stats) {
        // reduce over the 32 pixels of each half-wave, then one atomic per channel and kind from lanes 16 / 48
        const int copy = blockIdx.z * p.stat_copies + (int)((unsigned)blockIdx.y % (unsigned)p.stat_copies);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = sy_sum32_upper(ssum[r]), b = sy_sum32_upper(ssq[r]);
            const int co = c0 + wave * 32 + (r >> 2) * 8 + half * 4 + (r & 3);
            if (l31 == 16 && co < p.Cout) {
                atomicAdd(p.stat_sum + (long long)copy * p.Cout + co, a);
                atomicAdd(p.stat_sq + (long long)copy * p.Cout + co, b);
            }
        }
        if (p.fin != nullptr) conv_stats_finalize<128, 256>(p, (int)blockIdx.z, c0, tid, smem);
    }
}

template <typename T, int NS>
int launch_1x1_stream_ns(const ConvArgs& a, void* stream) {
    constexpr int PT = 64, CT = 128;
    constexpr size_t smem = 2 * (size_t)PT * NS * 64 + 4 * 64 * 80;
    const int nseg = a.seg_M > 0 ? a.M / a.seg_M : 1;
    const int seg_M = a.seg_M > 0 ? a.seg_M : a.M;
    const int ntiles = (seg_M + PT - 1) / PT;
    const int gx = (a.Cout + CT - 1) / CT;
    // about 768 workgroups per launch, at least 4 tiles each
    int gy = 768 / (gx * nseg);
    if (gy < 1) gy = 1;
    if (gy > (ntiles + 3) / 4) gy = (ntiles + 3) / 4;
    if (gy < 1) gy = 1;
    // no idle workgroups: with per = ceil(ntiles / gy) tiles each, ceil(ntiles / per) workgroups cover the range (sy_conv_desc::fin
    // counts one ticket per workgroup of a channel tile)
    gy = (ntiles + (ntiles + gy - 1) / gy - 1) / ((ntiles + gy - 1) / gy);
#ifndef SY_EMU
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)conv1x1_stream_kernel<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return SY_ERR_LAUNCH;
        attr_done = true;
    }
#endif
    SY_LAUNCH((conv1x1_stream_kernel<T, NS>), dim3(gx, gy, nseg), dim3(256), smem, stream, a);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

// tile code 120 (SY_TILE_STREAM1X1)
template <typename T>
int launch_1x1_stream(const ConvArgs& a, void* stream) {
    if constexpr (T::kEPC != 8) {
        return SY_ERR_UNSUPPORTED;
    } else {
        // 1x1 stride 1 (forward and data gradient are the same gather), dense pixel rows, raw 16-bit output (training forward /
        // data gradient: no affine, activation, residual), fragment-packed weights
        if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.Ho != a.H || a.Wo != a.W) return SY_ERR_UNSUPPORTED;
        if (a.epilogue != SY_EPI_LINEAR || a.scale != nullptr || a.shift != nullptr || a.res != nullptr || a.y_f32) return SY_ERR_UNSUPPORTED;
        if (a.x_extent == 0 || a.wfrag == nullptr || a.wfrag_extent == 0) return SY_ERR_UNSUPPORTED;
        if (a.xbs != (long long)a.H * a.W * a.ldx || a.ybs != (long long)a.Ho * a.Wo * a.ldy) return SY_ERR_UNSUPPORTED;
        if ((a.ldy & 7) || (a.Cout & 7) || (reinterpret_cast<unsigned long long>(a.y) & 15ull)) return SY_ERR_UNSUPPORTED;
        switch (a.Cin) {
            case 64: return launch_1x1_stream_ns<T, 2>(a, stream);
            case 128: return launch_1x1_stream_ns<T, 4>(a, stream);
            case 256: return launch_1x1_stream_ns<T, 8>(a, stream);
            // (Cin 512 = 128 weight registers per lane: spills at the 256-register cap, tools/regs_census.sh — left to the tiled kernel)
            default: return SY_ERR_UNSUPPORTED;
        }
    }
}

}  // namespace sy_conv
