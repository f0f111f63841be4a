// conv_igemm_impl.h — NHWC implicit-GEMM convolution on the CDNA4 matrix cores, fused epilogue.
//
// Replaces (reference, per call): nn.Conv2d -> nn.BatchNorm2d -> nn.SiLU of yolox BaseConv, three
// cuDNN/ATen kernels plus the cat/add kernels around them, and cuDNN's
// backward-data under autograd.
//
// GEMM view:  D[co][p] = sum_k Wp[co][k] * X[p][k],   k = (tap, ci),  p = (n, ho, wo)
//   * MFMA "A/row" operand = packed weights [Cout][K] (K-contiguous), "B/col" operand = pixels:
//     each lane then owns 4 CONSECUTIVE output channels of one pixel per accumulator quad, so the
//     epilogue stores 8 B (16-bit types) / 16 B (fp32) per lane straight from registers — NHWC
//     output needs no LDS transpose.
//   * K is walked in 64-byte slabs per row (32 bf16/f16 or 16 fp32 elements), 16-byte chunks.
//   * wave64: 4 or 8 waves per workgroup arranged WC x WP over (channels x pixels); each wave owns
//     TC x TP accumulator tiles of 32x32 (16 fp32 registers each).
//   * gather modes: forward (stride 1/2, zero padding) and data-gradient (transposed conv).
//   * two loaders (template FAST):
//       FAST    Cin is a whole number of slabs: taps are the INNER loop (the kh*kw gathers of one channel
//               slab hit the same cache lines one slab apart) and every address is a per-row 32-bit
//               offset + ONE wave-uniform offset, fetched with bounds-checked buffer loads: out-of-image
//               taps and ragged edges get offset 0xFFFFFFFF and read zeros.  ~2 VALU per 16-byte load.
//       generic any Cin % 8 == 0 (stem: 16 channels, taps straddle slabs): per-chunk (tap, ci) tracking and
//               64-bit address math (~40 VALU per load; made the first version instruction-issue bound).
//   * two staging strategies (template RS):
//       RS = 1  register-staged: one slab of loads held in VGPRs while the previous one feeds the MFMAs,
//               single padded LDS buffer (80-byte pitch), two barriers per slab; ~20 KiB of LDS so several
//               workgroups share a CU and hide each other's latency.  Fastest on MI355X today.
//       RS = 0  4-stage LDS ring filled by LDS-DMA (global/buffer_load ... lds in inline asm, counted
//               s_waitcnt vmcnt(N), raw s_barrier, XOR swizzle applied on the source side so the
//               ds_read_b128 fragment reads stay conflict free).
//   * epilogue (all fp32): z = acc*scale[co] + shift[co]; linear | SiLU | sigmoid | box decode;
//     optional residual add, optional += into the destination, optional per-channel sum / sum of
//     squares of the raw accumulator for training-mode BatchNorm (wave shuffles, LDS fold over the
//     waves, one atomic per channel per workgroup into one of `stat_copies` replicas).
#pragma once
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

namespace sy_conv {

struct ConvArgs {
    const unsigned char* x;
    const unsigned char* w;
    const float* scale;
    const float* shift;
    const unsigned char* res;
    unsigned char* y;
    float* stat_sum;
    float* stat_sq;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int ldx, ldy, ldr;
    long long xbs, ybs, rbs;
    int y_f32, mode, epilogue, accumulate;
    float dec_stride;
    int M, K, HoWo;
    int stat_copies;            // replicas of the statistics arrays (atomic-contention control)
    int seg_M;                  // > 0: pixels per statistics segment (gridDim.z segments, tiles never straddle one)
    int s2_classes;             // stride-2 data gradient split into 4 output-parity classes (gridDim.z), see the kernel
    int tile;                   // 0 = heuristic, else a forced tile configuration (tests / tuning)
    unsigned x_extent, w_extent; // bytes addressable from x / w (FAST loader's buffer bounds)
    const unsigned char* wfrag; // fragment-packed weights (STG 5), [Cout/32][slab][2][64 lanes][16 B]
    unsigned wfrag_extent;
    const unsigned char* pre_w; // fused Bottleneck (bottleneck_fused.h, tile 119): fragment-packed 1x1 weights [Cin][pre_cin] of the conv in front
    unsigned pre_w_extent;
    const float* pre_scale;     // ... its folded BatchNorm affine [Cin]
    const float* pre_shift;
    int pre_cin;                // ... and its input channels (x has pre_cin channels then, Cin = the hidden width)
    int ksplit;                 // > 1: split-K over channel-slab ranges (conv3x3_halo2_kernel), gridDim.z splits, fp32 partial outputs
    int wt;                     // write the output through the L2 (sy_store16_wt): set by sy_conv2d for outputs of SY_WT_MIN_BYTES or more
    int ablate;                 // profiling only (tools/conv_probe.py): 1 = no pixel loads, 2 = no weight loads, 8 = no statistics atomics, 16 = no cross-lane statistics reduction
};

// Staged write-out needs PT * (2 * CT + 16) bytes of LDS: up to 48 KiB (the tiles share a CU with other workgroups)
template <int WC, int WP, int TC, int TP> struct StageLimit { static constexpr size_t kBytes = 48 * 1024; };

// LDS in front of the staged output tile: BN-statistics scratch [WP][CT][2] floats
template <int WP, int CT> struct EpiLds { static constexpr int kStatBytes = WP * CT * 8; };

constexpr int kRowB = 64;           // DMA ring: bytes of K per LDS row per slab (unpadded: LDS-DMA lands lane-linear)
constexpr int kPitchRS = 80;        // register-staged: 64 B of K + 16 B pad per LDS row (conflict-free ds_read_b128)

static __device__ uint4 g_zero16[4] = {};  // DMA generic loader: source of every predicated-off 16-byte chunk

// Pixel mapping of a workgroup's tile for the epilogue: local pixel lp in [0, PT) -> (image n, pixel index `rem` inside the
// image) of the OUTPUT tensor; `seg` = statistics segment of the tile, `rep` = tile index used to pick a replica array.
struct LinearPixels {             // implicit-GEMM kernel: PT consecutive pixels of the flattened (n, ho, wo) range
    int m0, m_end, cls_hw, Wc, Wo, cls_ph, cls_pw, s2, seg, rep;
    template <typename Args> __device__ __forceinline__ LinearPixels(const Args& p, int by, int bz, int PT) {
        cls_ph = 0; cls_pw = 0; Wc = p.Wo; Wo = p.Wo; s2 = p.s2_classes;
        int Hc = p.Ho, cls_M = p.M;
        if (p.s2_classes) {
            cls_ph = bz >> 1; cls_pw = bz & 1;
            Hc = (p.Ho - cls_ph + 1) >> 1; Wc = (p.Wo - cls_pw + 1) >> 1;
            cls_M = p.N * Hc * Wc;
        }
        cls_hw = Hc * Wc;
        m_end = p.seg_M > 0 ? (bz + 1) * p.seg_M : cls_M;
        m0 = (p.s2_classes ? 0 : bz * p.seg_M) + by * PT;
        seg = bz; rep = by;
    }
    __device__ __forceinline__ bool map(int lp, int& n, int& rem) const {
        const int m = m0 + lp;
        if (m >= m_end) { n = 0; rem = 0; return false; }
        n = m / cls_hw;
        rem = m - n * cls_hw;
        if (s2) { const int i2 = rem / Wc; rem = (2 * i2 + cls_ph) * Wo + 2 * (rem - i2 * Wc) + cls_pw; }
        return true;
    }
};
struct TilePixels {               // halo kernel: TH rows of 32 consecutive pixels of one image, origin (h0, w0)
    int n, h0, w0, Ho, Wo, seg, rep;
    __device__ __forceinline__ bool map(int lp, int& n_, int& rem) const {
        const int h = h0 + (lp >> 5), w = w0 + (lp & 31);
        n_ = n;
        if (h >= Ho || w >= Wo) { rem = 0; return false; }
        rem = h * Wo + w;
        return true;
    }
};

template <typename T, int WC, int WP, int TC, int TP, typename Args, typename Map>
__device__ __forceinline__ void conv_epilogue(const Args& p, const Map& mp, int bx, f32x16 (&acc)[TC][TP],
                                              unsigned char* smem, int tid);

// STG: staging strategy — 1 = register-staged; 2 / 3 / 4 (0 = 4) = LDS-DMA ring of that depth (depth-1 slabs of
// loads in flight while one feeds the MFMAs; a shallow ring costs less LDS, so more workgroups share a CU).
template <typename T, int WC, int WP, int TC, int TP, int STG, int FAST>
__global__ __launch_bounds__(WC * WP * 64, ((TC * TP <= 4 && !(STG == 6 && TC * TP == 4)) ? 4 : 2)) void conv_igemm_kernel(ConvArgs p) {
    SY_TL_BEGIN(1 + (p.mode == SY_CONV_DGRAD ? 32 : 0));
    typedef typename T::elem elem;
    constexpr int RS = (STG == 1 || STG == 5 || STG == 6) ? 1 : 0;
    constexpr int WR = (STG == 5 || STG == 6) ? 1 : 0;   // weights: fragment-packed, global -> VGPR, never in LDS
    constexpr int WR3 = (STG == 6) ? 1 : 0;         // ... with three register stages and two LDS pixel buffers
    constexpr int kStages = RS ? 1 : (STG == 0 ? 4 : STG);
    constexpr int kThreads = WC * WP * 64;       // 4 or 8 waves
    constexpr int RPI = kThreads / 4;            // rows staged per sweep of the workgroup (one 16-byte chunk per lane)
    constexpr int EPC = T::kEPC;                 // elements per 16-byte chunk
    constexpr int ESZ = 16 / EPC;                // bytes per element
    constexpr int BK = 4 * EPC;                  // elements per 64-byte K slab
    constexpr int CT = WC * TC * 32;             // channels per workgroup
    constexpr int PT = WP * TP * 32;             // pixels per workgroup
    constexpr int WCH = (CT * 4 + kThreads - 1) / kThreads;   // weight chunks per thread per slab
    constexpr int XCH = (PT * 4 + kThreads - 1) / kThreads;   // pixel chunks per thread per slab
    static_assert(WC * WP == 4 || WC * WP == 8, "4 or 8 waves per workgroup");
    static_assert(PT % RPI == 0, "every wave stages pixel rows");

    // ONE LDS object (a second one makes hipcc drain the LDS-DMA queue before every ds_read — guide §5)
    SY_DYN_SMEM(smem);
    unsigned char* const sW = smem;
    unsigned char* const sX = smem + (WR ? 0 : (RS ? CT * kPitchRS : kStages * CT * kRowB));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave / WP;
    const int wp = wave % WP;
    const sy_block_id bid = sy_xcd_block_id();   // logical tile of this workgroup (XCD-contiguous order)
    const int c0 = bid.x * CT;
    // statistics segments (training forward of a frame pair): segment z owns pixels [z * seg_M, (z + 1) * seg_M)
    // Stride-2 data gradient: an output pixel (ho, wo) only receives the taps with kh = ho + pad, kw = wo + pad (mod 2)
    // — a 3x3 kernel has 1, 2, 2 or 4 of them, never 9.  gridDim.z enumerates the four parity classes; a workgroup
    // tiles the half-resolution grid of ITS class and walks only that class's taps, so no MFMA multiplies padding.
    int cls_ph = 0, cls_pw = 0, Hc = p.Ho, Wc = p.Wo, cls_M = p.M;
    if (p.s2_classes) {
        cls_ph = bid.z >> 1; cls_pw = bid.z & 1;
        Hc = (p.Ho - cls_ph + 1) >> 1; Wc = (p.Wo - cls_pw + 1) >> 1;
        cls_M = p.N * Hc * Wc;
        if (bid.y * PT >= cls_M) return;              // uniform: the smaller classes need fewer tiles
    }
    const int cls_hw = Hc * Wc;
    const int m_end = p.seg_M > 0 ? (bid.z + 1) * p.seg_M : cls_M;
    const int m0 = (p.s2_classes ? 0 : bid.z * p.seg_M) + bid.y * PT;
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // ---- staging assignment.  Thread t owns rows (t>>2) + RPI*i and 16-byte slot t&3 of each.  In the DMA
    //      ring one wave instruction fills 16 rows x 64 B lane-linearly, and the physical slot s of row r holds
    //      logical K-chunk s ^ ((r>>2)&3) (XOR swizzle on the source side); for a thread's rows (r>>2)&3 ==
    //      (t>>4)&3, so its logical chunk is the same for every row.
    const int kc = RS ? (tid & 3) : ((tid & 3) ^ ((tid >> 4) & 3));
    const int row0 = tid >> 2;
    const bool w_active = (wave * 16) < CT;      // DMA: narrow weight tiles are staged by the first waves only
    const int ntaps = p.KH * p.KW;

    // output pixel of staged row i -> image index and gather origin
    auto row_geom = [&](int i, bool& ok, long long& base, int& h0, int& w0) {
        const int m = m0 + row0 + i * RPI;
        ok = (m < m_end);
        const int mm = ok ? m : 0;
        const int n = mm / cls_hw;
        const int rem = mm - n * cls_hw;
        int ho = rem / Wc;
        int wo = rem - ho * Wc;
        if (p.s2_classes) { ho = 2 * ho + cls_ph; wo = 2 * wo + cls_pw; }
        base = (long long)n * p.xbs;
        if (p.mode == SY_CONV_FWD) { h0 = ho * p.stride - p.pad; w0 = wo * p.stride - p.pad; }
        else { h0 = ho + p.pad; w0 = wo + p.pad; }
    };
    auto tap_coords = [&](int h0, int w0, int kh, int kw, int& hi, int& wi) -> bool {   // gather source of one tap
        bool ok = true;
        if (p.mode == SY_CONV_FWD) { hi = h0 + kh; wi = w0 + kw; }
        else {
            hi = h0 - kh; wi = w0 - kw;
            if (p.stride == 2) { ok = ((hi & 1) == 0) && ((wi & 1) == 0); hi >>= 1; wi >>= 1; }
        }
        return ok && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
    };

    f32x16 acc[TC][TP];
#pragma unroll
    for (int t = 0; t < TC; ++t)
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    const int nslab_all = (p.K + BK - 1) / BK;                 // K slabs of the whole filter (fragment-packed weight pitch)
    const int kh0 = p.s2_classes ? ((cls_ph + p.pad) & 1) : 0, kw0 = p.s2_classes ? ((cls_pw + p.pad) & 1) : 0;
    const int kstep = p.s2_classes ? 2 : 1;
    // slabs this workgroup walks: all of them, or (class taps) x (channel slabs)
    const int nslab = p.s2_classes ? ((p.KH - kh0 + 1) / 2) * ((p.KW - kw0 + 1) / 2) * (p.Cin / BK) : nslab_all;

    // one slab of MFMAs from LDS rows `bw`/`bx` with row pitch PITCH and per-lane 16-byte slot selector
    auto compute_slab = [&](const unsigned char* bw, const unsigned char* bx, int pitch, int swz) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int slot = ((g * 2 + half) ^ swz) * 16;
            uint4 a[TC], b[TP];
#pragma unroll
            for (int t = 0; t < TC; ++t) a[t] = *reinterpret_cast<const uint4*>(bw + ((wc * TC + t) * 32 + l31) * pitch + slot);
#pragma unroll
            for (int u = 0; u < TP; ++u) b[u] = *reinterpret_cast<const uint4*>(bx + ((wp * TP + u) * 32 + l31) * pitch + slot);
#pragma unroll
            for (int t = 0; t < TC; ++t)
#pragma unroll
                for (int u = 0; u < TP; ++u) acc[t][u] = sy_mfma_group(T(), a[t], b[u], acc[t][u]);
        }
    };

    if constexpr (FAST) {
        // ================= FAST loader: per-row 32-bit offsets + one wave-uniform offset per slab =================
        const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
        const sy_buffer bufw = sy_make_buffer(p.w, p.w_extent);
        const bool half_res = (p.mode != SY_CONV_FWD) && p.stride == 2;
        unsigned xoff[XCH], xmask[XCH], woff[WCH];
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            bool ok; long long base; int h0, w0;
            row_geom(i, ok, base, h0, w0);
            unsigned mask = 0u;
            for (int t = 0; t < ntaps; ++t) {
                const int kh = t / p.KW, kw = t - kh * p.KW;
                int hi, wi;
                if (ok && tap_coords(h0, w0, kh, kw, hi, wi)) mask |= (1u << t);
            }
            xmask[i] = (p.ablate & 1) ? 0u : mask;
            const int hb = half_res ? (h0 >> 1) : h0, wb = half_res ? (w0 >> 1) : w0;
            xoff[i] = (unsigned)((base + ((long long)hb * p.W + wb) * p.ldx + kc * EPC) * ESZ);
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int r = row0 + i * RPI;
            const int co = c0 + r;
            woff[i] = (r < CT && co < p.Cout && !(p.ablate & 2)) ? (unsigned)(((long long)co * p.K + kc * EPC) * ESZ) : 0xFFFFFFFFu;
        }
        int f_kh = kh0, f_kw = kw0, f_c = 0, f_t = kh0 * p.KW + kw0, f_cs = 0;   // wave-uniform K position of the next slab to fetch
        unsigned s_x = 0, s_w = 0, s_f = 0;
        // pixel delta of tap (kh, kw): forward +(kh*W + kw); data gradient -(kh*W + kw), at half resolution for stride 2
        const int tap_sh = half_res ? 1 : 0;
        const int tap_ld = (p.mode == SY_CONV_FWD) ? p.ldx : -p.ldx;
        auto slab_offsets = [&]() {                          // uniform offsets of slab (f_t, f_c); then advance
            const int dpix = (f_kh >> tap_sh) * p.W + (f_kw >> tap_sh);
            s_x = (unsigned)((dpix * tap_ld + f_c) * ESZ);
            s_w = (unsigned)((f_t * p.Cin + f_c) * ESZ);
            s_f = (unsigned)((f_cs * ntaps + f_t) * 2048);   // fragment-packed weights: [cslab][tap] blocks of 2 KiB
        };
        auto advance = [&]() {
            f_kw += kstep;
            if (f_kw >= p.KW) {
                f_kw = kw0;
                f_kh += kstep;
                if (f_kh >= p.KH) { f_kh = kh0; f_c += BK; ++f_cs; }
            }
            f_t = f_kh * p.KW + f_kw;
        };
        if constexpr (WR) {
            // ---- weights straight to VGPRs.  The host packs them in MFMA-fragment order along the kernel's own
            //      K traversal, so one wave instruction reads 1 KiB contiguous (perfectly coalesced, L2 resident)
            //      and the weight operand costs no LDS write, no LDS read and no barrier traffic at all.
            const sy_buffer buff = sy_make_buffer(p.wfrag, p.wfrag_extent);
            const int ntile32 = (p.Cout + 31) / 32;
            unsigned foff[TC];
#pragma unroll
            for (int t = 0; t < TC; ++t) {
                const int ct = bid.x * (CT / 32) + wc * TC + t;
                foff[t] = (ct < ntile32 && !(p.ablate & 2)) ? (unsigned)((((long long)ct * nslab_all) * 128 + lane) * 16) : 0xFFFFFFFFu;
            }
            if constexpr (WR3) {
                // ---- three register stages, two LDS pixel buffers.  Iteration s: (1) issue the pixel loads of slab
                //      s+3, (2) park slab s+1's pixels (landed by now) in the other LDS buffer, (3) multiply slab s,
                //      (4) issue the weight fragments of slab s+3 into the registers slab s just used, (5) ONE barrier.
                //      Loads get two full iterations to land (the single-stage loop above waits for them in the
                //      iteration that issued them) and the second barrier per slab is gone.
                uint4 fr[3][TC][2], rx3[3][XCH];
                unsigned h_f[3] = {0u, 0u, 0u};                     // fragment offset of the slab whose pixels sit in set i
                auto load_x = [&](auto set_) {
                    constexpr int S = decltype(set_)::value;
                    slab_offsets();
                    h_f[S] = s_f;
#pragma unroll
                    for (int i = 0; i < XCH; ++i)
                        rx3[S][i] = sy_buffer_load16(bufx, ((xmask[i] >> f_t) & 1u) ? xoff[i] + s_x : 0xFFFFFFFFu);
                    advance();
                };
                auto load_f = [&](auto set_) {
                    constexpr int S = decltype(set_)::value;
#pragma unroll
                    for (int t = 0; t < TC; ++t)
#pragma unroll
                        for (int g = 0; g < 2; ++g)
                            fr[S][t][g] = sy_buffer_load16_s(buff, foff[t] == 0xFFFFFFFFu ? 0xFFFFFFFFu : foff[t] + (unsigned)(g * 1024), h_f[S]);
                };
                auto store_x = [&](auto set_, int buf) {
                    constexpr int S = decltype(set_)::value;
#pragma unroll
                    for (int i = 0; i < XCH; ++i)
                        *reinterpret_cast<uint4*>(sX + buf * (PT * kPitchRS) + (row0 + i * RPI) * kPitchRS + kc * 16) = rx3[S][i];
                };
                auto mma = [&](auto set_, int buf) {
                    constexpr int S = decltype(set_)::value;
                    const unsigned char* bx = sX + buf * (PT * kPitchRS);
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        uint4 b[TP];
#pragma unroll
                        for (int u = 0; u < TP; ++u)
                            b[u] = *reinterpret_cast<const uint4*>(bx + ((wp * TP + u) * 32 + l31) * kPitchRS + (g * 2 + half) * 16);
#pragma unroll
                        for (int t = 0; t < TC; ++t)
#pragma unroll
                            for (int u = 0; u < TP; ++u) acc[t][u] = sy_mfma_group(T(), fr[S][t][g], b[u], acc[t][u]);
                    }
                };
                load_x(sy_int<0>()); load_f(sy_int<0>());
                if (nslab > 1) { load_x(sy_int<1>()); load_f(sy_int<1>()); }
                if (nslab > 2) { load_x(sy_int<2>()); load_f(sy_int<2>()); }
                store_x(sy_int<0>(), 0);
                __syncthreads();
                for (int s0 = 0; s0 < nslab; s0 += 3) {
                    sy_static_for<0, 3>([&](auto j_) {
                        constexpr int J = decltype(j_)::value;
                        const int s = s0 + J;
                        if (s < nslab) {                                  // uniform
                            if (s + 3 < nslab) load_x(sy_int<J>());       // pixel registers of set J were parked last iteration
                            if (s + 1 < nslab) store_x(sy_int<(J + 1) % 3>(), (s + 1) & 1);
                            mma(sy_int<J>(), s & 1);
                            if (s + 3 < nslab) load_f(sy_int<J>());
                            __syncthreads();
                        }
                    });
                }
            } else {
            uint4 fa[TC][2], fn[TC][2], rx[XCH];
            auto load_slab = [&]() {
                slab_offsets();
                // fragment loads: the per-lane offset is fixed for the whole kernel, the slab position rides in the
                // scalar offset operand of the buffer load (no VALU address math)
#pragma unroll
                for (int t = 0; t < TC; ++t)
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        fn[t][g] = sy_buffer_load16_s(buff, foff[t] == 0xFFFFFFFFu ? 0xFFFFFFFFu : foff[t] + (unsigned)(g * 1024), s_f);
#pragma unroll
                for (int i = 0; i < XCH; ++i) rx[i] = sy_buffer_load16(bufx, ((xmask[i] >> f_t) & 1u) ? xoff[i] + s_x : 0xFFFFFFFFu);
                advance();
            };
            auto store_slab = [&]() {
#pragma unroll
                for (int i = 0; i < XCH; ++i) *reinterpret_cast<uint4*>(sX + (row0 + i * RPI) * kPitchRS + kc * 16) = rx[i];
#pragma unroll
                for (int t = 0; t < TC; ++t) { fa[t][0] = fn[t][0]; fa[t][1] = fn[t][1]; }
            };
            load_slab();
            store_slab();
            __syncthreads();
            for (int s = 0; s < nslab; ++s) {
                const bool more = (s + 1 < nslab);
                if (more) load_slab();
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    uint4 b[TP];
#pragma unroll
                    for (int u = 0; u < TP; ++u)
                        b[u] = *reinterpret_cast<const uint4*>(sX + ((wp * TP + u) * 32 + l31) * kPitchRS + (g * 2 + half) * 16);
#pragma unroll
                    for (int t = 0; t < TC; ++t)
#pragma unroll
                        for (int u = 0; u < TP; ++u) acc[t][u] = sy_mfma_group(T(), fa[t][g], b[u], acc[t][u]);
                }
                __syncthreads();
                if (more) {
                    store_slab();
                    __syncthreads();
                }
            }
            }
        } else if constexpr (RS) {
            uint4 rw[WCH], rx[XCH];
            auto load_slab = [&]() {
                slab_offsets();
#pragma unroll
                for (int i = 0; i < WCH; ++i) rw[i] = sy_buffer_load16(bufw, woff[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : woff[i] + s_w);
#pragma unroll
                for (int i = 0; i < XCH; ++i) rx[i] = sy_buffer_load16(bufx, ((xmask[i] >> f_t) & 1u) ? xoff[i] + s_x : 0xFFFFFFFFu);
                advance();
            };
            auto store_slab = [&]() {
#pragma unroll
                for (int i = 0; i < WCH; ++i) {
                    const int r = row0 + i * RPI;
                    if (r < CT) *reinterpret_cast<uint4*>(sW + r * kPitchRS + kc * 16) = rw[i];
                }
#pragma unroll
                for (int i = 0; i < XCH; ++i) *reinterpret_cast<uint4*>(sX + (row0 + i * RPI) * kPitchRS + kc * 16) = rx[i];
            };
            load_slab();
            store_slab();
            __syncthreads();
            for (int s = 0; s < nslab; ++s) {
                const bool more = (s + 1 < nslab);
                if (more) load_slab();
                compute_slab(sW, sX, kPitchRS, 0);
                __syncthreads();
                if (more) {
                    store_slab();
                    __syncthreads();
                }
            }
        } else {
            int issued = 0;
            auto issue_slab = [&]() {
                const int stage = issued % kStages;
                slab_offsets();
                if (w_active) {
#pragma unroll
                    for (int i = 0; i < WCH; ++i)
                        sy_glds16_buf(bufw, woff[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : woff[i] + s_w,
                                      sW + (stage * CT + wave * 16 + i * RPI) * kRowB);
                }
#pragma unroll
                for (int i = 0; i < XCH; ++i)
                    sy_glds16_buf(bufx, ((xmask[i] >> f_t) & 1u) ? xoff[i] + s_x : 0xFFFFFFFFu,
                                  sX + (stage * PT + wave * 16 + i * RPI) * kRowB);
                advance();
                ++issued;
            };
            auto wait_slab = [&](int ahead) {      // at most `ahead` later slabs of THIS wave's loads still in flight
                constexpr int LA = WCH + XCH, LI = XCH;
                if (w_active) {
                    if (ahead >= 2) sy_wait_vmcnt<2 * LA>(); else if (ahead == 1) sy_wait_vmcnt<LA>(); else sy_wait_vmcnt<0>();
                } else {
                    if (ahead >= 2) sy_wait_vmcnt<2 * LI>(); else if (ahead == 1) sy_wait_vmcnt<LI>(); else sy_wait_vmcnt<0>();
                }
            };
            const int swz = (l31 >> 2) & 3;
            for (int j = 0; j < kStages - 1 && j < nslab; ++j) issue_slab();
            for (int s = 0; s < nslab; ++s) {
                wait_slab(issued - s - 1);               // slab s has landed (this wave's part) ...
                sy_barrier();                            // ... and everybody's; all waves are done reading slab s-1
                if (issued < nslab) issue_slab();        // refill the buffer slab s-1 occupied
                compute_slab(sW + (s % kStages) * CT * kRowB, sX + (s % kStages) * PT * kRowB, kRowB, swz);
            }
        }
    } else {
        // ================= generic loader: per-chunk (tap, ci), 64-bit addresses =================
        int px_h0[XCH], px_w0[XCH];
        long long px_base[XCH];
        bool px_ok[XCH];
#pragma unroll
        for (int i = 0; i < XCH; ++i) row_geom(i, px_ok[i], px_base[i], px_h0[i], px_w0[i]);
        const bool tap_inner = (ntaps > 1) && (p.Cin % BK == 0);
        int k_el = kc * EPC;            // this thread's element offset inside the K range of the slab being fetched
        int tap = tap_inner ? 0 : k_el / p.Cin;
        int ci = tap_inner ? k_el : k_el - tap * p.Cin;
        auto advance_k = [&]() {
            if (tap_inner) {
                if (++tap == ntaps) { tap = 0; ci += BK; }
                k_el = (ci < p.Cin) ? tap * p.Cin + ci : p.K;
            } else {
                k_el += BK;
                ci += BK;
                while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
            }
        };
        auto w_src = [&](int i) -> const unsigned char* {       // nullptr = zeros
            const int r = row0 + i * RPI;
            const int co = c0 + r;
            if (k_el < p.K && r < CT && co < p.Cout && !(p.ablate & 2)) return p.w + ((long long)co * p.K + k_el) * ESZ;
            return nullptr;
        };
        auto x_src = [&](int i, int kh, int kw) -> const unsigned char* {
            int hi, wi;
            if (k_el < p.K && px_ok[i] && tap_coords(px_h0[i], px_w0[i], kh, kw, hi, wi) && !(p.ablate & 1))
                return p.x + (px_base[i] + ((long long)hi * p.W + wi) * p.ldx + ci) * ESZ;
            return nullptr;
        };
        if constexpr (RS) {
            uint4 rw[WCH], rx[XCH];
            auto load_slab = [&]() {
#pragma unroll
                for (int i = 0; i < WCH; ++i) {
                    const unsigned char* src = w_src(i);
                    rw[i] = src ? *reinterpret_cast<const uint4*>(src) : make_uint4(0u, 0u, 0u, 0u);
                }
                const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const unsigned char* src = x_src(i, kh, kw);
                    rx[i] = src ? *reinterpret_cast<const uint4*>(src) : make_uint4(0u, 0u, 0u, 0u);
                }
                advance_k();
            };
            auto store_slab = [&]() {
#pragma unroll
                for (int i = 0; i < WCH; ++i) {
                    const int r = row0 + i * RPI;
                    if (r < CT) *reinterpret_cast<uint4*>(sW + r * kPitchRS + kc * 16) = rw[i];
                }
#pragma unroll
                for (int i = 0; i < XCH; ++i) *reinterpret_cast<uint4*>(sX + (row0 + i * RPI) * kPitchRS + kc * 16) = rx[i];
            };
            load_slab();
            store_slab();
            __syncthreads();
            for (int s = 0; s < nslab; ++s) {
                const bool more = (s + 1 < nslab);
                if (more) load_slab();
                compute_slab(sW, sX, kPitchRS, 0);
                __syncthreads();
                if (more) {
                    store_slab();
                    __syncthreads();
                }
            }
        } else {
            const unsigned char* const zero = reinterpret_cast<const unsigned char*>(g_zero16);
            int issued = 0;
            auto issue_slab = [&]() {
                const int stage = issued % kStages;
                if (w_active) {
#pragma unroll
                    for (int i = 0; i < WCH; ++i) {
                        const unsigned char* src = w_src(i);
                        sy_glds16(src ? src : zero, sW + (stage * CT + wave * 16 + i * RPI) * kRowB);
                    }
                }
                const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
                for (int i = 0; i < XCH; ++i) {
                    const unsigned char* src = x_src(i, kh, kw);
                    sy_glds16(src ? src : zero, sX + (stage * PT + wave * 16 + i * RPI) * kRowB);
                }
                advance_k();
                ++issued;
            };
            auto wait_slab = [&](int ahead) {
                constexpr int LA = WCH + XCH, LI = XCH;
                if (w_active) {
                    if (ahead >= 2) sy_wait_vmcnt<2 * LA>(); else if (ahead == 1) sy_wait_vmcnt<LA>(); else sy_wait_vmcnt<0>();
                } else {
                    if (ahead >= 2) sy_wait_vmcnt<2 * LI>(); else if (ahead == 1) sy_wait_vmcnt<LI>(); else sy_wait_vmcnt<0>();
                }
            };
            const int swz = (l31 >> 2) & 3;
            for (int j = 0; j < kStages - 1 && j < nslab; ++j) issue_slab();
            for (int s = 0; s < nslab; ++s) {
                wait_slab(issued - s - 1);
                sy_barrier();
                if (issued < nslab) issue_slab();
                compute_slab(sW + (s % kStages) * CT * kRowB, sX + (s % kStages) * PT * kRowB, kRowB, swz);
            }
        }
    }

    // ---- epilogue -----------------------------------------------------------------------------
    // Everything below sees the kernel arguments and the tile geometry through LATE copies (SY_LATE_ARGS): values
    // used only here are not kept in SGPRs across the K loop.
    SY_LATE_ARGS(ConvArgs, p);
    int e_bx = bid.x, e_by = bid.y, e_bz = bid.z;
    SY_LAUNDER_INT(e_bx); SY_LAUNDER_INT(e_by); SY_LAUNDER_INT(e_bz);
    const LinearPixels mp(p_late, e_by, e_bz, PT);
    conv_epilogue<T, WC, WP, TC, TP>(p_late, mp, e_bx, acc, smem, tid);
    SY_TL_END();
}

// The epilogue as a separate (inlined) function: its only inputs are the late argument view, the logical tile and
// the accumulators, so none of the prologue's uniforms can be referenced (and kept alive) by accident.
template <typename T, int WC, int WP, int TC, int TP, typename Args, typename Map>
__device__ __forceinline__ void conv_epilogue(const Args& p, const Map& mp, int bx, f32x16 (&acc)[TC][TP],
                                              unsigned char* smem, int tid) {
    typedef typename T::elem elem;
    constexpr int kThreads = WC * WP * 64;
    constexpr int EPC = T::kEPC;
    constexpr int ESZ = 16 / EPC;
    constexpr int CT = WC * TC * 32;
    constexpr int PT = WP * TP * 32;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wc = wave / WP;
    const int wp = wave % WP;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int c0 = bx * CT;
    const bool vec_ok = (p.epilogue != SY_EPI_DECODE) && ((p.Cout & 3) == 0) && ((p.ldy & 3) == 0) &&
                        (p.res == nullptr || (p.ldr & 3) == 0);
    const bool bnr = (p.epilogue == SY_EPI_BNR);                 // fused BatchNorm-backward reduce of the producing layer (data gradient)
    const bool want_stats = (p.stat_sum != nullptr) && !bnr;
    // Output staging (16-bit outputs without residual / accumulate): the MFMA accumulator layout gives a lane 4
    // channels of one pixel, i.e. 8-byte stores 2*ldy bytes apart — 64 partial cache lines per store instruction,
    // and the 1x1 layers were bound by exactly that.  Instead the wave parks its converted tile in LDS as
    // [pixel][channel] and the workgroup writes whole pixel rows, 16 bytes per lane, consecutive lanes consecutive
    // addresses (the K loop is over, its LDS is free).
    constexpr int kStagePitch = CT * 2 + 16;                    // bytes per staged pixel row (+16: bank spread)
    constexpr int kStatBytes = EpiLds<WP, CT>::kStatBytes;      // BN-statistics scratch [WP][CT][2] floats (aliases too)
    constexpr bool kCanStage = (ESZ == 2) && ((size_t)PT * kStagePitch + PT * 8 + kStatBytes <= StageLimit<WC, WP, TC, TP>::kBytes);
    // (+= outputs — data gradients of activations with several consumers — are staged too: the write-out pass reads
    //  the old row chunk, adds in fp32 and stores, all coalesced; the staged value was already rounded to 16 bits,
    //  one extra rounding the gradient path tolerates.  Residual adds keep the direct path: single rounding.)
    // (residual adds — the eval Bottleneck convs — take the staged path too when the residual tile can be loaded as whole pixel
    //  rows: it is parked in the staging rows first, every lane adds its own 4-channel groups in fp32 and rounds ONCE, in place)
    constexpr bool kResStage = (WC * TC >= TP);                 // residual row offsets reuse the statistics scratch [WP][CT][2]
    const bool res_ok = p.res == nullptr || bnr || (kResStage && !want_stats && !p.accumulate && ((p.ldr & 7) == 0) &&
                                             (p.epilogue == SY_EPI_LINEAR || p.epilogue == SY_EPI_SILU) &&      // = the lean path below
                                             ((reinterpret_cast<unsigned long long>(p.res) & 15ull) == 0));
    const bool stage_out = kCanStage && vec_ok && !p.y_f32 && res_ok && ((p.ldy & 7) == 0) &&
                           ((reinterpret_cast<unsigned long long>(p.y) & 15ull) == 0);
    unsigned char* const stg = smem + kStatBytes;               // [PT][kStagePitch]
    long long* const stg_off = reinterpret_cast<long long*>(smem + kStatBytes + PT * kStagePitch);   // [PT] element offsets
    if (want_stats || stage_out) __syncthreads();               // every wave is done with the operand tiles
    // ---- lean path for staged outputs (every training forward / first-write data gradient / eval conv without a
    //      residual): the mode, affine and statistics decisions are taken ONCE here, not per element — the general
    //      loop below re-tests them for each of its 16 x TP values per tile, and at 1x1 layers (4-16 slabs of MFMA
    //      per tile) those ~2500 scalar-ish instructions per wave were the whole kernel time.
    const bool lean = kCanStage && stage_out && (p.epilogue == SY_EPI_LINEAR || p.epilogue == SY_EPI_SILU || bnr) &&
                      (!want_stats || (p.epilogue == SY_EPI_LINEAR && p.scale == nullptr && p.shift == nullptr));
    if (lean) {
        if constexpr (kCanStage) {
            auto body = [&](auto silu_, auto aff_, auto stats_, auto res_) {
                // res_ = 2: BNR — the staged rows are first filled with the producing layer's raw output z (as a residual tile would
                // be, through the output offsets: z has y's strides), every lane then turns its own values into g = acc * silu'(scale z
                // + shift), accumulates sum g / sum g z per channel (the statistics reduction below) and overwrites z with acc
                constexpr bool SILU = decltype(silu_)::value != 0, AFF = decltype(aff_)::value != 0, BNR = decltype(res_)::value == 2,
                               STATS = decltype(stats_)::value != 0 || BNR, RES = decltype(res_)::value == 1;
                long long* const stg_roff = reinterpret_cast<long long*>(smem);     // [PT] residual row offsets (RES only)
                if (wc == 0 && half == 0) {                         // output element offset of every tile pixel, once
#pragma unroll
                    for (int u = 0; u < TP; ++u) {
                        long long off = -1, roff = -1;
                        int n, rem;
                        if (mp.map((wp * TP + u) * 32 + l31, n, rem)) {
                            off = (long long)n * p.ybs + (long long)rem * p.ldy;
                            roff = (long long)n * p.rbs + (long long)rem * p.ldr;
                        }
                        stg_off[(wp * TP + u) * 32 + l31] = off;
                        if (RES) stg_roff[(wp * TP + u) * 32 + l31] = roff;
                    }
                }
                if constexpr (BNR) {                                // z of the producing layer, whole pixel rows, into the staging rows
                    __syncthreads();
                    constexpr int CPR = CT / 8;
                    for (int i = tid; i < PT * CPR; i += kThreads) {
                        const int px = i / CPR, ck = i - px * CPR;
                        const long long off = stg_off[px];
                        const int co = c0 + ck * 8;
                        uint4 v = make_uint4(0u, 0u, 0u, 0u);
                        if (off >= 0 && co < p.Cout) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const elem*>(p.res) + off + co);
                        *reinterpret_cast<uint4*>(stg + px * kStagePitch + ck * 16) = v;
                    }
                    __syncthreads();
                }
                if constexpr (RES) {                                // the residual tile, whole pixel rows, into the staging rows
                    __syncthreads();
                    constexpr int CPR = CT / 8;
                    for (int i = tid; i < PT * CPR; i += kThreads) {
                        const int px = i / CPR, ck = i - px * CPR;
                        const long long roff = stg_roff[px];
                        const int co = c0 + ck * 8;
                        uint4 v = make_uint4(0u, 0u, 0u, 0u);
                        if (roff >= 0 && co < p.Cout) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const elem*>(p.res) + roff + co);
                        *reinterpret_cast<uint4*>(stg + px * kStagePitch + ck * 16) = v;
                    }
                    __syncthreads();
                }
                sy_static_for<0, TC>([&](auto tc_) {
                    constexpr int t = decltype(tc_)::value;
                    float sc[16], sh[16], ssum[16], ssq[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (AFF || BNR) {
                            const int co = sy_min(c0 + (wc * TC + t) * 32 + (r >> 2) * 8 + half * 4 + (r & 3), p.Cout - 1);
                            sc[r] = p.scale[co]; sh[r] = p.shift[co];
                        }
                        if (STATS) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
                    }
#pragma unroll
                    for (int u = 0; u < TP; ++u) {
                        unsigned char* const row = stg + ((wp * TP + u) * 32 + l31) * kStagePitch + ((wc * TC + t) * 32 + half * 4) * 2;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float v[4];
                            if constexpr (BNR) {
                                const uint2 zz = *reinterpret_cast<const uint2*>(row + q * 16);
                                elem ze[4];
                                __builtin_memcpy(ze, &zz, 8);
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const float a = acc[t][u][q * 4 + j], zf = T::to_f32(ze[j]);
                                    const float g = a * sy_silu_grad(zf * sc[q * 4 + j] + sh[q * 4 + j]);
                                    ssum[q * 4 + j] += g; ssq[q * 4 + j] += g * zf;
                                    v[j] = a;
                                }
                            } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float a = acc[t][u][q * 4 + j];
                                if (STATS) { ssum[q * 4 + j] += a; ssq[q * 4 + j] += a * a; }
                                float z = AFF ? a * sc[q * 4 + j] + sh[q * 4 + j] : a;
                                if (SILU) z = sy_silu(z);
                                v[j] = z;
                            }
                            }
                            if (RES) {                              // this lane's own 4 channels of its pixel: read, add, round once
                                const uint2 rr = *reinterpret_cast<const uint2*>(row + q * 16);
                                elem re[4];
                                __builtin_memcpy(re, &rr, 8);
#pragma unroll
                                for (int j = 0; j < 4; ++j) v[j] += T::to_f32(re[j]);
                            }
                            *reinterpret_cast<uint2*>(row + q * 16) = make_uint2(T::pack2(v[0], v[1]), T::pack2(v[2], v[3]));
                        }
                    }
                    if (STATS) {
                        // Per-channel totals over the tile's 32 pixel columns: value-halving butterfly (sy_reduce16_over32) — lane l
                        // ends up with the totals of accumulator register l & 15, lanes 0-15 of each half write them.
                        float* red = reinterpret_cast<float*>(smem);          // [WP][CT][2]
                        float a = 0.0f, b = 0.0f;
                        if (!(p.ablate & 16)) { a = sy_reduce16_over32(ssum); b = sy_reduce16_over32(ssq); }
                        const int r = l31 & 15;
                        const int cl = (wc * TC + t) * 32 + (r >> 2) * 8 + half * 4 + (r & 3);
                        if (l31 < 16) { red[(wp * CT + cl) * 2 + 0] = a; red[(wp * CT + cl) * 2 + 1] = b; }
                    }
                });
            };
            const bool aff = p.scale != nullptr && p.shift != nullptr;
            const bool silu = p.epilogue == SY_EPI_SILU;
            if (bnr) body(sy_int<0>(), sy_int<0>(), sy_int<0>(), sy_int<2>());                 // data gradient + the producer's BatchNorm-backward reduce
            else if (want_stats) body(sy_int<0>(), sy_int<0>(), sy_int<1>(), sy_int<0>());     // training forward: raw output + statistics
            else if (p.res != nullptr) {
                if constexpr (kResStage) {
                    if (silu && aff) body(sy_int<1>(), sy_int<1>(), sy_int<0>(), sy_int<1>());   // eval Bottleneck conv with shortcut
                    else if (silu) body(sy_int<1>(), sy_int<0>(), sy_int<0>(), sy_int<1>());
                    else if (aff) body(sy_int<0>(), sy_int<1>(), sy_int<0>(), sy_int<1>());
                    else body(sy_int<0>(), sy_int<0>(), sy_int<0>(), sy_int<1>());
                }
            }
            else if (silu && aff) body(sy_int<1>(), sy_int<1>(), sy_int<0>(), sy_int<0>());    // eval BaseConv
            else if (!silu && !aff) body(sy_int<0>(), sy_int<0>(), sy_int<0>(), sy_int<0>());  // data gradient
            else if (silu) body(sy_int<1>(), sy_int<0>(), sy_int<0>(), sy_int<0>());
            else body(sy_int<0>(), sy_int<1>(), sy_int<0>(), sy_int<0>());
        }
    } else
    // compile-time loop over the wave's channel tiles: accumulator indices must be constants (a runtime
    // index would push the 128-register accumulator file of the 256x256 tile into scratch)
    sy_static_for<0, TC>([&](auto tc_) {
        constexpr int t = decltype(tc_)::value;
        float ssum[16], ssq[16];
        if (want_stats) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { ssum[r] = 0.0f; ssq[r] = 0.0f; }
        }
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            int n, rem;
            const bool m_ok = mp.map((wp * TP + u) * 32 + l31, n, rem);
            const long long yoff = (long long)n * p.ybs + (long long)rem * p.ldy;
            const long long roff = (long long)n * p.rbs + (long long)rem * p.ldr;
            int gy = 0, gx = 0;
            if (p.epilogue == SY_EPI_DECODE) { gy = rem / p.Wo; gx = rem - gy * p.Wo; }
            if (kCanStage && stage_out && t == 0 && wc == 0 && half == 0)
                stg_off[(wp * TP + u) * 32 + l31] = m_ok ? yoff : -1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cb = c0 + (wc * TC + t) * 32 + q * 8 + half * 4;   // first of 4 channels
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = acc[t][u][q * 4 + j];
                    if (want_stats) { ssum[q * 4 + j] += a; ssq[q * 4 + j] += a * a; }
                    const int co = cb + j;
                    const float sc = (p.scale != nullptr && co < p.Cout) ? p.scale[co] : 1.0f;
                    const float sh = (p.shift != nullptr && co < p.Cout) ? p.shift[co] : 0.0f;
                    float z = a * sc + sh;
                    if (p.epilogue == SY_EPI_SILU) z = sy_silu(z);
                    else if (p.epilogue == SY_EPI_SIGMOID) z = sy_sigmoid(z);
                    else if (p.epilogue == SY_EPI_DECODE) {
                        if (co == 0) z = (z + (float)gx) * p.dec_stride;
                        else if (co == 1) z = (z + (float)gy) * p.dec_stride;
                        else if (co == 2 || co == 3) z = sy_exp(z) * p.dec_stride;
                        else z = sy_sigmoid(z);
                    }
                    v[j] = z;
                }
                if (kCanStage && stage_out) {
                    elem e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] = T::from_f32(v[j]);
                    uint2 o;
                    __builtin_memcpy(&o, e, 8);
                    *reinterpret_cast<uint2*>(stg + ((wp * TP + u) * 32 + l31) * kStagePitch +
                                              ((wc * TC + t) * 32 + q * 8 + half * 4) * 2) = o;
                    continue;
                }
                if (!m_ok || cb >= p.Cout) continue;   // (inside the q loop)
                if (vec_ok) {
                    if (p.res != nullptr) {
                        const elem* rp = reinterpret_cast<const elem*>(p.res) + roff + cb;
                        if (ESZ == 2) {
                            uint2 rv = *reinterpret_cast<const uint2*>(rp);
                            elem e[4];
                            __builtin_memcpy(e, &rv, 8);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += T::to_f32(e[j]);
                        } else {
                            float4 rv = *reinterpret_cast<const float4*>(rp);
                            v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
                        }
                    }
                    if (p.y_f32 || ESZ == 4) {
                        float* yp = reinterpret_cast<float*>(p.y) + yoff + cb;
                        if (p.accumulate) {
                            float4 o = *reinterpret_cast<const float4*>(yp);
                            v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
                        }
                        *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        elem* yp = reinterpret_cast<elem*>(p.y) + yoff + cb;
                        elem e[4];
                        if (p.accumulate) {
                            uint2 o = *reinterpret_cast<const uint2*>(yp);
                            __builtin_memcpy(e, &o, 8);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] += T::to_f32(e[j]);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) e[j] = T::from_f32(v[j]);
                        uint2 o;
                        __builtin_memcpy(&o, e, 8);
                        *reinterpret_cast<uint2*>(yp) = o;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = cb + j;
                        if (co >= p.Cout) continue;
                        float z = v[j];
                        if (p.res != nullptr) z += T::to_f32(reinterpret_cast<const elem*>(p.res)[roff + co]);
                        if (p.y_f32 || ESZ == 4) {
                            float* yp = reinterpret_cast<float*>(p.y) + yoff + co;
                            if (p.accumulate) z += *yp;
                            *yp = z;
                        } else {
                            elem* yp = reinterpret_cast<elem*>(p.y) + yoff + co;
                            if (p.accumulate) z += T::to_f32(*yp);
                            *yp = T::from_f32(z);
                        }
                    }
                }
            }
        }
        if (want_stats) {
            // reduce over the 32 pixels held by lanes with equal `half`; park the wave's 32 channel sums in LDS
            // (the K loop is over: sW is free after the barrier below), fold the WP waves, one atomic per channel.
            float* red = reinterpret_cast<float*>(smem);          // [WP][CT][2]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = sy_sum32_upper(ssum[r]), b = sy_sum32_upper(ssq[r]);
                const int cl = (wc * TC + t) * 32 + (r >> 2) * 8 + half * 4 + (r & 3);
                if (l31 == 16) {
                    red[(wp * CT + cl) * 2 + 0] = a;
                    red[(wp * CT + cl) * 2 + 1] = b;
                }
            }
        }
    });
    if (want_stats || (kCanStage && stage_out)) __syncthreads();
    sy_probe(4);
    if (kCanStage && stage_out && !p.accumulate) {
        // first-write rows (every training forward, most data gradients, the eval convs): the chunks of a thread are independent —
        // all of a batch's LDS reads go out before its first store instead of one read -> store round trip per chunk (round 5: the
        // write-out of a 128 ch x 160 px tile was ten serial LDS round trips, 1.1 of the workgroup's 20 us)
        constexpr int CPR = CT / 8;                             // 16-byte chunks per staged pixel row
        constexpr int NCH = (PT * CPR + kThreads - 1) / kThreads;
        constexpr int UB = NCH < 5 ? NCH : 5;
#pragma unroll 1
        for (int b0 = 0; b0 < NCH; b0 += UB) {
            uint4 v[UB];
            long long off[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int i = tid + (b0 + k) * kThreads;
                const int ck = i % CPR;
                const bool ok = i < PT * CPR && c0 + ck * 8 < p.Cout;
                const int px = ok ? i / CPR : 0;                 // (unconditional reads of a valid row: no exec-mask branches, the
                const long long o_ = stg_off[px];                //  batch stays in registers)
                v[k] = *reinterpret_cast<const uint4*>(stg + px * kStagePitch + ck * 16);
                off[k] = ok ? o_ : -1;
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int i = tid + (b0 + k) * kThreads;
                const int ck = i % CPR;
                if (off[k] >= 0) {
                    if (SY_WT_CONV && p.wt) sy_store16_wt(reinterpret_cast<elem*>(p.y) + off[k] + c0 + ck * 8, v[k]);
                    else *reinterpret_cast<uint4*>(reinterpret_cast<elem*>(p.y) + off[k] + c0 + ck * 8) = v[k];
                }
            }
        }
    } else if (kCanStage && stage_out) {
        constexpr int CPR = CT / 8;                             // 16-byte chunks per staged pixel row
        for (int i = tid; i < PT * CPR; i += kThreads) {
            const int px = i / CPR, ck = i - px * CPR;
            const long long off = stg_off[px];
            const int co = c0 + ck * 8;
            if (off < 0 || co >= p.Cout) continue;
            uint4 v = *reinterpret_cast<const uint4*>(stg + px * kStagePitch + ck * 16);
            uint4* const dst = reinterpret_cast<uint4*>(reinterpret_cast<elem*>(p.y) + off + co);
            if (p.accumulate) {
                const uint4 o = *dst;
                elem ev[8], eo[8];
                __builtin_memcpy(ev, &v, 16);
                __builtin_memcpy(eo, &o, 16);
                unsigned w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w[j] = T::pack2(T::to_f32(ev[2 * j]) + T::to_f32(eo[2 * j]), T::to_f32(ev[2 * j + 1]) + T::to_f32(eo[2 * j + 1]));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            if (SY_WT_CONV && p.wt) sy_store16_wt(dst, v); else *dst = v;
        }
    }
    sy_probe(5);
    if (bnr && lean) {                                          // sums [copies][2][Cout]: row 0 = sum g, row 1 = sum g z (raw moment)
        const float* red = reinterpret_cast<const float*>(smem);
        float* const dst = p.stat_sum + (long long)((unsigned)mp.rep % (unsigned)p.stat_copies) * 2 * p.Cout;
        for (int cl = tid; cl < CT; cl += kThreads) {
            const int co = c0 + cl;
            if (co >= p.Cout) continue;
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w = 0; w < WP; ++w) { a += red[(w * CT + cl) * 2]; b += red[(w * CT + cl) * 2 + 1]; }
            atomicAdd(dst + co, a);
            atomicAdd(dst + p.Cout + co, b);
        }
    }
    if (want_stats && !(p.ablate & 8)) {
        const float* red = reinterpret_cast<const float*>(smem);
        const int copy = mp.seg * p.stat_copies + (int)((unsigned)mp.rep % (unsigned)p.stat_copies);
        for (int cl = tid; cl < CT; cl += kThreads) {
            const int co = c0 + cl;
            if (co >= p.Cout) continue;
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int w = 0; w < WP; ++w) { a += red[(w * CT + cl) * 2]; b += red[(w * CT + cl) * 2 + 1]; }
            atomicAdd(p.stat_sum + (long long)copy * p.Cout + co, a);
            atomicAdd(p.stat_sq + (long long)copy * p.Cout + co, b);
        }
    }
}

template <typename T, int WC, int WP, int TC, int TP, int STG, int FAST>
int launch_one(const ConvArgs& a, void* stream) {
    constexpr int RS = (STG == 1 || STG == 5 || STG == 6) ? 1 : 0;
    constexpr int kStages = RS ? 1 : (STG == 0 ? 4 : STG);
    constexpr int CT = WC * TC * 32, PT = WP * TP * 32;
    const int nseg = a.seg_M > 0 ? a.M / a.seg_M : 1;
    dim3 grid((a.Cout + CT - 1) / CT, ((a.seg_M > 0 ? a.seg_M : a.M) + PT - 1) / PT, nseg);
    if (a.s2_classes) grid = dim3(grid.x, (a.N * ((a.Ho + 1) / 2) * ((a.Wo + 1) / 2) + PT - 1) / PT, 4);
    // STG 5 keeps only the pixel tile in LDS (the BN-statistics scratch [WP][CT][2] floats aliases it after the K loop)
    constexpr size_t smem_k = (STG == 6) ? (size_t)(2 * PT * kPitchRS > WP * CT * 8 ? 2 * PT * kPitchRS : WP * CT * 8)
                              : (STG == 5) ? (size_t)(PT * kPitchRS > WP * CT * 8 ? PT * kPitchRS : WP * CT * 8)
                                         : (RS ? (size_t)(CT + PT) * kPitchRS : (size_t)kStages * (CT + PT) * kRowB);
    // epilogue staging of 16-bit outputs (see the kernel): statistics scratch + [PT][CT*2+16] + [PT] offsets
    constexpr size_t smem_e = (size_t)EpiLds<WP, CT>::kStatBytes + (size_t)PT * (CT * 2 + 16) + (size_t)PT * 8;
    constexpr size_t smem = (T::kEPC == 8 && smem_e <= 48 * 1024 && smem_e > smem_k) ? smem_e : smem_k;
#ifndef SY_EMU
    static sy_dev_once attr_done;          // > 64 KiB of dynamic LDS needs the opt-in once per kernel instance
    if (attr_done.need()) {
        if (hipFuncSetAttribute((const void*)conv_igemm_kernel<T, WC, WP, TC, TP, STG, FAST>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return SY_ERR_LAUNCH;
        attr_done.mark();
    }
#endif
    SY_LAUNCH((conv_igemm_kernel<T, WC, WP, TC, TP, STG, FAST>), grid, dim3(WC * WP * 64), smem, stream, a);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

template <typename T, int WC, int WP, int TC, int TP, int STG = 0>
int launch_cfg(const ConvArgs& a_in, void* stream) {
    // FAST loader preconditions: whole slabs per tap, 32-bit addressable operands, taps fit the validity mask
    ConvArgs a = a_in;
    const bool fast = (a.Cin % (4 * T::kEPC) == 0) && a.x_extent != 0 && a.w_extent != 0 && a.KH * a.KW <= 32;
    a.s2_classes = (fast && a.mode != SY_CONV_FWD && a.stride == 2 && a.KH >= 2 && a.KW >= 2 && !(a_in.ablate & 4)) ? 1 : 0;
    if constexpr (STG == 5 || STG == 6) {   // fragment-packed weights exist only for the FAST traversal; else plain register staging
        if (fast && a.wfrag != nullptr && a.wfrag_extent != 0) return launch_one<T, WC, WP, TC, TP, STG, 1>(a, stream);
        return fast ? launch_one<T, WC, WP, TC, TP, 1, 1>(a, stream) : launch_one<T, WC, WP, TC, TP, 1, 0>(a, stream);
    } else {
        return fast ? launch_one<T, WC, WP, TC, TP, STG, 1>(a, stream) : launch_one<T, WC, WP, TC, TP, STG, 0>(a, stream);
    }
}

template <typename T> int launch_halo_typed(const ConvArgs& a, void* stream);      // conv3x3_halo.h (tile codes 104..107, 110..118)
template <typename T> int launch_halo3_typed(const ConvArgs& a, void* stream);     // conv3x3_halo3.h (tile codes 96..98, 100, 101, 109)
template <typename T> int launch_s2dgrad(const ConvArgs& a, void* stream);         // conv3x3_s2dgrad.h (tile code 108)
template <typename T> int launch_s2dgrad4(const ConvArgs& a, void* stream);        // conv3x3_s2dgrad.h (tile codes 125 - 127: all four classes per workgroup)
template <typename T> int launch_1x1_tile(const ConvArgs& a, void* stream);        // conv1x1_tile.h (tile codes 121..124)
template <typename T> int launch_bottleneck_fused(const ConvArgs& a, void* stream); // bottleneck_fused.h (tile code 119)

template <typename T>
int launch_typed(const ConvArgs& a, void* stream) {
    if ((a.tile >= 104 && a.tile <= 107) || (a.tile >= 110 && a.tile <= 118)) return launch_halo_typed<T>(a, stream);
    if (a.tile == 109 || (a.tile >= 96 && a.tile <= 98) || a.tile == 100 || a.tile == 101) return launch_halo3_typed<T>(a, stream);
    if (a.tile == 119) return launch_bottleneck_fused<T>(a, stream);
    if (a.tile == 108) return launch_s2dgrad<T>(a, stream);
    if (a.tile >= 125 && a.tile <= 127) return launch_s2dgrad4<T>(a, stream);
    if (a.tile >= 121 && a.tile <= 124) return launch_1x1_tile<T>(a, stream);
    // Tile choice.  The kernel is fed from L2: bytes staged per MFMA flop fall with the tile area, so wide
    // layers use 256 ch x 256 px (8 waves, 128 accumulator registers per lane).  Layers too small to give
    // every CU a large tile fall back to 128 x 128 (4 waves); narrow layers trade channels for pixels.
    switch (a.tile) {
        case SY_TILE_256x256: return launch_cfg<T, 2, 4, 4, 2>(a, stream);
        case SY_TILE_128x256: return launch_cfg<T, 1, 8, 4, 1>(a, stream);
        case SY_TILE_128x128: return launch_cfg<T, 2, 2, 2, 2>(a, stream);
        case SY_TILE_64x256: return launch_cfg<T, 1, 4, 2, 2>(a, stream);
        case SY_TILE_32x256: return launch_cfg<T, 1, 4, 1, 2>(a, stream);
        case SY_TILE_128x64: return launch_cfg<T, 4, 1, 1, 2>(a, stream);
        case SY_TILE_64x64: return launch_cfg<T, 2, 2, 1, 1>(a, stream);
        case SY_TILE_RS + SY_TILE_128x64: return launch_cfg<T, 4, 1, 1, 2, 1>(a, stream);
        case SY_TILE_RS + SY_TILE_64x64: return launch_cfg<T, 2, 2, 1, 1, 1>(a, stream);
        case 32 + SY_TILE_128x128: return launch_cfg<T, 2, 2, 2, 2, 2>(a, stream);       // 2-deep DMA ring
        case 32 + SY_TILE_128x64: return launch_cfg<T, 4, 1, 1, 2, 2>(a, stream);
        case 32 + SY_TILE_64x64: return launch_cfg<T, 2, 2, 1, 1, 2>(a, stream);
        case 32 + SY_TILE_64x256: return launch_cfg<T, 1, 4, 2, 2, 2>(a, stream);
        case 48 + SY_TILE_128x128: return launch_cfg<T, 2, 2, 2, 2, 3>(a, stream);       // 3-deep DMA ring
        case 48 + SY_TILE_128x64: return launch_cfg<T, 4, 1, 1, 2, 3>(a, stream);
        case 48 + SY_TILE_64x64: return launch_cfg<T, 2, 2, 1, 1, 3>(a, stream);
        case 48 + SY_TILE_64x256: return launch_cfg<T, 1, 4, 2, 2, 3>(a, stream);
        case SY_TILE_RS + SY_TILE_256x64: return launch_cfg<T, 4, 1, 2, 2, 1>(a, stream);
        case 80 + SY_TILE_256x64: return launch_cfg<T, 4, 1, 2, 2, 5>(a, stream);         // 4 waves x (64 ch x 64 px), shared pixel tile
        case 80 + SY_TILE_256x128: return launch_cfg<T, 4, 1, 2, 4, 5>(a, stream);        // 4 waves x (64 ch x 128 px)
        case 80 + SY_TILE_128x128: return launch_cfg<T, 2, 2, 2, 2, 5>(a, stream);       // weights in registers
        case 80 + SY_TILE_128x64: return launch_cfg<T, 4, 1, 1, 2, 5>(a, stream);
        // + 96: weights in registers, three register stages + two LDS pixel buffers (one barrier per slab)
        case 96 + SY_TILE_128x64: return launch_cfg<T, 4, 1, 1, 2, 6>(a, stream);
        case 96 + SY_TILE_64x64: return launch_cfg<T, 2, 2, 1, 1, 6>(a, stream);
        case 96 + SY_TILE_128x128: return launch_cfg<T, 2, 2, 2, 2, 6>(a, stream);
        case 80 + SY_TILE_64x64: return launch_cfg<T, 2, 2, 1, 1, 5>(a, stream);
        case 80 + SY_TILE_64x256: return launch_cfg<T, 1, 4, 2, 2, 5>(a, stream);
        case 80 + SY_TILE_32x256: return launch_cfg<T, 1, 4, 1, 2, 5>(a, stream);
        case SY_TILE_RS + SY_TILE_256x256: return launch_cfg<T, 2, 4, 4, 2, 1>(a, stream);
        case SY_TILE_RS + SY_TILE_128x256: return launch_cfg<T, 1, 8, 4, 1, 1>(a, stream);
        case SY_TILE_RS + SY_TILE_128x128: return launch_cfg<T, 2, 2, 2, 2, 1>(a, stream);
        case SY_TILE_RS + SY_TILE_64x256: return launch_cfg<T, 1, 4, 2, 2, 1>(a, stream);
        case SY_TILE_RS + SY_TILE_32x256: return launch_cfg<T, 1, 4, 1, 2, 1>(a, stream);
        default: break;
    }
    // Heuristic (tools/conv_probe.py on MI355X): the register-staged variant wins at every layer shape of the
    // path because 4-5 of its workgroups share a CU; pick the largest tile that still yields >= 2 workgroups
    // per CU (256 CUs), trading channels for pixels on narrow layers.
    auto blocks = [&](int ct, int pt) { return (long long)((a.Cout + ct - 1) / ct) * ((a.M + pt - 1) / pt); };
    if (a.Cout <= 32) return launch_cfg<T, 1, 4, 1, 2, 1>(a, stream);                       //  32 ch x 256 px
    if (a.Cout <= 64) return launch_cfg<T, 1, 4, 2, 2, 1>(a, stream);                       //  64 ch x 256 px
    if (blocks(128, 128) >= 512) return launch_cfg<T, 2, 2, 2, 2, 1>(a, stream);            // 128 ch x 128 px
    if (blocks(128, 64) >= 512) return launch_cfg<T, 4, 1, 1, 2, 1>(a, stream);             // 128 ch x  64 px
    return launch_cfg<T, 2, 2, 1, 1, 1>(a, stream);                                         //  64 ch x  64 px
}

}  // namespace sy_conv
