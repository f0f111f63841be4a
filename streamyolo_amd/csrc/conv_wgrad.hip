// conv_wgrad.hip — convolution weight gradient on the CDNA4 matrix cores.
//
// Replaces cuDNN backward-filter under autograd (reference: loss.backward() in
// exps/train_utils/double_trainer.py:114).
//
// GEMM view:  dW[co][k] += sum_p  dY[p][co] * X[gather(p, tap)][ci],    k = (tap, ci)
// The contraction index is the PIXEL, which is the slow axis of both NHWC operands.  Two staging schemes:
//  * 16-bit types (the production path, conv_wgrad_tr_kernel below): tiles land in LDS untransposed by LDS-DMA
//    and the K-major fragments are gathered with ds_read_b64_tr_b16 — 1.6-2x the scatter kernel on MI355X;
//  * fp32 (parity runs) and shapes the first scheme does not cover (conv_wgrad_kernel): both tiles
// are transposed on their way into LDS: a lane loads 16 bytes (8 / 4 consecutive channels of one
// pixel) and scatters them as 32-bit words into channel-major rows [channel][pixel] (two adjacent
// pixels are packed per word for 16-bit types).  Row pitch 72 B: the scatter is at most 2-way bank
// conflicted (free on ds_write_b32) and the MFMA fragments read back as conflict-free ds_read_b64.
//   rows  (MFMA A operand) = X^T : (tap, ci)      cols (MFMA B operand) = dY^T : co
// so a lane's accumulator quad is 4 consecutive k of one co = 4 consecutive floats of dW[co][:].
// The pixel range is split over gridDim.z; each split writes its partial tile to a private slab of the
// caller's workspace (plain 16-byte stores) and a fold kernel sums the slabs INTO dW (+=), which is
// how the current-frame and support-frame passes of the shared backbone weights add up
// (SURVEY.md §8(e)); the caller zeroes the gradient arena once per step.  Deterministic: no atomics.
// (Round 6 measured the fold INSIDE the launch — the last workgroup of an output tile to publish its slab, behind an agent-scope
//  release / ticket / acquire, adds the tile's slabs in split order; parity-green on the MI355X — and removed it: a tile's `splits`
//  slabs are 0.5-1.8 MB that ONE workgroup then pulls alone, the weight gradients took 15.75 instead of 8.95 ms one at a time and the
//  l step 24.9-25.1 instead of 21.9-22.0 ms on the same box (profiles/r06 stage b); the fold stays a chip-wide launch.)
#include <stdlib.h>
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

namespace {

struct WgradArgs {
    const unsigned char* x;
    const unsigned char* dy;
    float* dw;
    int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int ldx, lddy;
    long long xbs, dybs;
    int M, K, slabs_per_split;
    int oihw;                   // 1: dW laid out [Cout][Cin][KH][KW] (the nn.Parameter layout), 0: [Cout][tap][Cin]
    int tile, target_blocks;    // tuning knobs: 0 = heuristics
    unsigned x_extent, dy_extent; // bytes addressable from x / dy (0: 64-bit pointer loads)
    float* part;                // split > 1: partial slabs [splits][K / 4][Cout][4], folded by wgrad_fold_kernel
    int splits;
    int part_bf16;              // slabs stored as bf16 (bf16 compute mode): half the slab traffic of the kernels and of the fold
    int ablate;                 // profiling only (tools/wgrad_probe.py): 1 = no x loads, 2 = no dy loads
};

constexpr int kPitchT = 72;
constexpr int kThreadsW = 256;

// One float4 of a split's partial slab, layout [split][K / 4][Cout][4]: the 32 lanes of a half-wave hold 32 consecutive output
// channels, so their stores are one contiguous run (the former [Cout][K] layout scattered 16-byte pieces over 32 rows per store
// instruction: -0.3 ms per l step, profiles/r03/q_*); wgrad_fold_kernel walks the slabs in this order.  bf16 mode stores the
// partial sums as bf16 (the final sum over the splits is fp32): the reference's own autocast returns the whole weight gradient
// of a bf16 convolution rounded to bf16 (torch conv backward under autocast), so this is no coarser than the reference.
__device__ __forceinline__ void wgrad_store_slab(const WgradArgs& p, int split, int co, int kb, float v0, float v1, float v2,
                                                 float v3) {
    const long long g = ((long long)split * (p.K >> 2) + (kb >> 2)) * p.Cout + co;
    if (p.part_bf16) {
        uint2 u;
        u.x = BF16::pack2(v0, v1);
        u.y = BF16::pack2(v2, v3);
#if SY_WT_SLAB
        sy_store8_wt(reinterpret_cast<uint2*>(p.part) + g, u);
#else
        reinterpret_cast<uint2*>(p.part)[g] = u;
#endif
    } else {
        const float4 f = make_float4(v0, v1, v2, v3);
        uint4 q;
        __builtin_memcpy(&q, &f, 16);
#if SY_WT_SLAB
        sy_store16_wt(reinterpret_cast<float4*>(p.part) + g, q);
#else
        reinterpret_cast<float4*>(p.part)[g] = f;
#endif
    }
}

struct PixelCursor {            // (n, ho, wo) of one output pixel, advanced slab by slab
    int n, ho, wo, m;
    __device__ __forceinline__ void init(int m_, int Ho, int Wo) {
        m = m_;
        const int hw = Ho * Wo;
        n = m_ / hw;
        const int rem = m_ - n * hw;
        ho = rem / Wo;
        wo = rem - ho * Wo;
    }
    __device__ __forceinline__ void advance(int step, int Ho, int Wo) {
        m += step;
        wo += step;
        while (wo >= Wo) {
            wo -= Wo;
            if (++ho == Ho) { ho = 0; ++n; }
        }
    }
};

// ---- epilogue: D[row = k][col = co].  One split: every dW element belongs to exactly one workgroup, so a
//      plain += into dW is race free.  Several splits: each writes its tile to a private slab (16-byte
//      stores, no atomics) and wgrad_fold_kernel sums the slabs into dW.
template <int TR, int TC>
__device__ __forceinline__ void wgrad_epilogue(const WgradArgs& p, f32x16 (&acc)[TR][TC], int r0, int c0, int wr, int wcn,
                                               int lane, int split) {
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int taps = p.KH * p.KW;
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int u = 0; u < TC; ++u) {
            const int co = c0 + (wcn * TC + u) * 32 + l31;
            if (co >= p.Cout) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int kb = r0 + (wr * TR + t) * 32 + q * 8 + half * 4;
                if (kb >= p.K) continue;                       // Cin % 4 == 0: the quad shares one tap and K % 4 == 0
                const float v0 = acc[t][u][q * 4 + 0], v1 = acc[t][u][q * 4 + 1], v2 = acc[t][u][q * 4 + 2],
                            v3 = acc[t][u][q * 4 + 3];
                if (p.splits > 1) {
                    wgrad_store_slab(p, split, co, kb, v0, v1, v2, v3);
                } else if (p.oihw) {
                    const int tap = kb / p.Cin;
                    const int ci = kb - tap * p.Cin;
                    float* row = p.dw + (long long)co * p.K + (long long)ci * taps + tap;
                    row[0] += v0; row[taps] += v1; row[2 * taps] += v2; row[3 * taps] += v3;
                } else {
                    float4* dst = reinterpret_cast<float4*>(p.dw + (long long)co * p.K + kb);
                    float4 o = *dst;
                    o.x += v0; o.y += v1; o.z += v2; o.w += v3;
                    *dst = o;
                }
            }
        }
}

template <typename T, int WR, int WC, int TR, int TC>
__global__ __launch_bounds__(kThreadsW) void conv_wgrad_kernel(WgradArgs p) {
    SY_TL_BEGIN(8);
    typedef typename T::elem elem;
    constexpr int EPC = T::kEPC;
    constexpr int ESZ = 16 / EPC;
    constexpr int SLAB = 4 * EPC;                 // pixels per slab (64 bytes per transposed row)
    constexpr int PPU = (ESZ == 2) ? 2 : 1;       // pixels per staging unit (pair-packed for 16-bit)
    constexpr int SLOTS = SLAB / PPU;             // 16 pixel slots per slab
    constexpr int RT = WR * TR * 32;              // rows: (tap, ci)
    constexpr int CT = WC * TC * 32;              // cols: co
    constexpr int UA = ((RT / EPC) * SLOTS + kThreadsW - 1) / kThreadsW;
    constexpr int UB = ((CT / EPC) * SLOTS + kThreadsW - 1) / kThreadsW;
    static_assert(WR * WC == 4, "4 waves per workgroup");
    static_assert(SLOTS == 16, "slot arithmetic below assumes 16 slots");

    __shared__ __attribute__((aligned(16))) unsigned char sA[RT * kPitchT];
    __shared__ __attribute__((aligned(16))) unsigned char sB[CT * kPitchT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WC;
    const int wcn = wave % WC;
    const sy_block_id bid = sy_xcd_block_id();   // the row / column tiles of one pixel range share an XCD's L2
    const int r0 = bid.x * RT;
    const int c0 = bid.y * CT;
    const int slab0 = bid.z * p.slabs_per_split;
    int nslab = p.slabs_per_split;
    const int slabs_total = (p.M + SLAB - 1) / SLAB;
    if (slab0 + nslab > slabs_total) nslab = slabs_total - slab0;
    if (nslab <= 0) return;                       // uniform for the whole workgroup

    const int slot = tid & 15;                    // pixel slot inside the slab (fastest across lanes)
    const int g0 = tid >> 4;                      // channel group; unit i uses group g0 + 16*i

    // per-unit constants
    int a_tap_h[UA], a_tap_w[UA], a_ci[UA];
    bool a_ok[UA];
#pragma unroll
    for (int i = 0; i < UA; ++i) {
        const int g = g0 + 16 * i;
        const int k = r0 + g * EPC;
        a_ok[i] = (g < RT / EPC) && (k < p.K);
        const int kk = a_ok[i] ? k : 0;
        const int tap = kk / p.Cin;
        a_ci[i] = kk - tap * p.Cin;
        a_tap_h[i] = tap / p.KW;
        a_tap_w[i] = tap - a_tap_h[i] * p.KW;
    }
    int b_co[UB];
    bool b_ok[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) {
        const int g = g0 + 16 * i;
        b_co[i] = c0 + g * EPC;
        b_ok[i] = (g < CT / EPC) && (b_co[i] < p.Cout);
    }

    PixelCursor cur[PPU];
#pragma unroll
    for (int q = 0; q < PPU; ++q) cur[q].init(slab0 * SLAB + slot * PPU + q, p.Ho, p.Wo);

    uint4 ra[UA][PPU], rb[UB][PPU];

    // Loads go through bounds-checked buffer descriptors with 32-bit byte offsets: the per-load address math is a
    // handful of 32-bit integer ops and out-of-image taps / ragged tails read zeros (offset 0xFFFFFFFF).
    const bool use_buf = p.x_extent != 0 && p.dy_extent != 0;
    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    const sy_buffer bufdy = sy_make_buffer(p.dy, p.dy_extent);
    int a_rel[UA];                         // element offset of (tap, ci) relative to the gather origin
#pragma unroll
    for (int i = 0; i < UA; ++i) a_rel[i] = (a_tap_h[i] * p.W + a_tap_w[i]) * p.ldx + a_ci[i];
    auto load_slab = [&]() {
#pragma unroll
        for (int q = 0; q < PPU; ++q) {
            const bool m_ok = cur[q].m < p.M;
            const int hb = cur[q].ho * p.stride - p.pad, wb = cur[q].wo * p.stride - p.pad;
            if (use_buf) {
                const int xo = cur[q].n * (int)p.xbs + (hb * p.W + wb) * p.ldx;
                const int yo = cur[q].n * (int)p.dybs + (cur[q].ho * p.Wo + cur[q].wo) * p.lddy;
#pragma unroll
                for (int i = 0; i < UA; ++i) {
                    const int hi = hb + a_tap_h[i], wi = wb + a_tap_w[i];
                    const bool ok = m_ok && a_ok[i] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W;
                    ra[i][q] = sy_buffer_load16(bufx, ok ? (unsigned)((xo + a_rel[i]) * ESZ) : 0xFFFFFFFFu);
                }
#pragma unroll
                for (int i = 0; i < UB; ++i)
                    rb[i][q] = sy_buffer_load16(bufdy, (m_ok && b_ok[i]) ? (unsigned)((yo + b_co[i]) * ESZ) : 0xFFFFFFFFu);
                continue;
            }
#pragma unroll
            for (int i = 0; i < UA; ++i) {
                const int hi = hb + a_tap_h[i], wi = wb + a_tap_w[i];
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (m_ok && a_ok[i] && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
                    v = *reinterpret_cast<const uint4*>(
                        p.x + ((long long)cur[q].n * p.xbs + ((long long)hi * p.W + wi) * p.ldx + a_ci[i]) * ESZ);
                ra[i][q] = v;
            }
#pragma unroll
            for (int i = 0; i < UB; ++i) {
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (m_ok && b_ok[i])
                    v = *reinterpret_cast<const uint4*>(
                        p.dy + ((long long)cur[q].n * p.dybs +
                                ((long long)cur[q].ho * p.Wo + cur[q].wo) * p.lddy + b_co[i]) * ESZ);
                rb[i][q] = v;
            }
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int q = 0; q < PPU; ++q) cur[q].advance(SLAB, p.Ho, p.Wo);
    };
    // transpose-scatter one unit: element j of the chunk goes to row (g*EPC + j), 32-bit word `slot`
    auto scatter = [&](unsigned char* base, int g, const uint4 (&v)[PPU]) {
        if (ESZ == 2) {
            unsigned short e0[8], e1[8];
            __builtin_memcpy(e0, &v[0], 16);
            __builtin_memcpy(e1, &v[PPU - 1], 16);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *reinterpret_cast<unsigned*>(base + (g * 8 + j) * kPitchT + slot * 4) =
                    (unsigned)e0[j] | ((unsigned)e1[j] << 16);
        } else {
            unsigned w[4];
            __builtin_memcpy(w, &v[0], 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<unsigned*>(base + (g * 4 + j) * kPitchT + slot * 4) = w[j];
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int i = 0; i < UA; ++i) {
            const int g = g0 + 16 * i;
            if (g < RT / EPC) scatter(sA, g, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int g = g0 + 16 * i;
            if (g < CT / EPC) scatter(sB, g, rb[i]);
        }
    };

    f32x16 acc[TR][TC];
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int u = 0; u < TC; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    const int l31 = lane & 31;
    const int hi16 = (lane >> 5) * 16;

    load_slab();
    store_slab();
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const bool more = (s + 1 < nslab);
        if (more) { advance(); load_slab(); }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 a[TR], b[TC];
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                const unsigned char* ptr = sA + ((wr * TR + t) * 32 + l31) * kPitchT + g * 32 + hi16;
                const uint2 lo = *reinterpret_cast<const uint2*>(ptr);
                const uint2 hi = *reinterpret_cast<const uint2*>(ptr + 8);
                a[t] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const unsigned char* ptr = sB + ((wcn * TC + u) * 32 + l31) * kPitchT + g * 32 + hi16;
                const uint2 lo = *reinterpret_cast<const uint2*>(ptr);
                const uint2 hi = *reinterpret_cast<const uint2*>(ptr + 8);
                b[u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int t = 0; t < TR; ++t)
#pragma unroll
                for (int u = 0; u < TC; ++u) acc[t][u] = sy_mfma_group(T(), a[t], b[u], acc[t][u]);
        }
        __syncthreads();
        if (more) {
            store_slab();
            __syncthreads();
        }
    }

    wgrad_epilogue<TR, TC>(p, acc, r0, c0, wr, wcn, lane, bid.z);
    SY_TL_END();
}

// ---- transpose-read variant (16-bit types) ------------------------------------------------------------------
// Both operand tiles land in LDS exactly as they lie in HBM — row-major [pixel][channel] — by LDS-DMA, and the
// K-major MFMA fragments are gathered with ds_read_b64_tr_b16 (sy_lds_read_tr16): no register round trip, no
// scatter, and the loads of the next slabs stay in flight behind the MFMAs (ring of STG slabs).
// LDS image of one slab: (RT + CT) / 16 subtiles of [32 pixels][16 channels] (1 KiB, one DMA wave instruction:
// lane = (pixel, 16-byte half)), 128 B of padding between subtiles so that the two subtiles a half-wave reads
// (MFMA rows 0-15 / 16-31) fall on disjoint banks.  Requires Cin % 16 == 0 and Cout % 16 == 0 (a subtile never
// straddles a tap) and buffer-addressable operands.
constexpr int kSubPitch = 1024 + 128;

// LIN = 1: 1x1 stride-1 layer over dense tensors (xbs = H W ldx, dybs = Ho Wo lddy): pixel m lives at element m * ld of both
// operands, a slab's addresses are per-lane constants + one wave-uniform offset, validity is m < M.
template <typename T, int WR, int WC, int TR, int TC, int STG, int LIN>
__global__ __launch_bounds__(kThreadsW) void conv_wgrad_tr_kernel(WgradArgs p) {
    SY_TL_BEGIN(7);
    constexpr int RT = WR * TR * 32, CT = WC * TC * 32;
    constexpr int SA = RT / 16, SB = CT / 16, NS = SA + SB;
    constexpr int STAGE = NS * kSubPitch;
    constexpr int SLAB = 32;
    static_assert(WR * WC == 4 && NS % 4 == 0 && T::kEPC == 8, "4 waves, whole DMA rounds, 16-bit elements");
    SY_DYN_SMEM(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave / WC;
    const int wcn = wave % WC;
    const sy_block_id bid = sy_xcd_block_id();   // the row / column tiles of one pixel range share an XCD's L2
    const int r0 = bid.x * RT;
    const int c0 = bid.y * CT;
    const int slab0 = bid.z * p.slabs_per_split;
    int nslab = p.slabs_per_split;
    const int slabs_total = (p.M + SLAB - 1) / SLAB;
    if (slab0 + nslab > slabs_total) nslab = slabs_total - slab0;
    if (nslab <= 0) return;

    // ---- DMA assignment: wave w fills X subtiles w + 4 j (j < LA) and dY subtiles w + 4 j (j < LB).
    //      Lane -> (pixel lane>>1, 16-byte channel half lane&1).
    // These kernels are instruction-issue bound (profiles/r02/r_pmc_wgrad.txt: 40 instructions per MFMA on the 1x1 layers):
    // the ring is kept full past the last slab (out-of-range pieces deposit zeros into an idle stage) so the wait in front of a
    // slab is ONE counted vmcnt, a piece's offset is OR-ed with an all-ones mask instead of selected (hipcc turns the select
    // into an exec-mask branch), and all fragments of a slab are requested before its first MFMA.
    constexpr int LA = SA / 4, LB = SB / 4;
    static_assert(SA % 4 == 0 && SB % 4 == 0, "every wave stages the same number of subtiles of each operand");
    const int wv = sy_uniform(wave);
    const int half8 = (lane & 1) * 8;
    bool a_ok[LA], b_ok[LB];
    int a_dh[LA], a_dw[LA], a_rel[LA], b_rel[LB];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int k = r0 + (wv + 4 * j) * 16;
        a_ok[j] = k < p.K;
        const int kk = a_ok[j] ? k : 0;
        const int tap = LIN ? 0 : kk / p.Cin;
        const int ci = kk - tap * p.Cin;
        a_dh[j] = LIN ? 0 : tap / p.KW;
        a_dw[j] = tap - a_dh[j] * p.KW;
        a_rel[j] = (a_dh[j] * p.W + a_dw[j]) * p.ldx + ci + half8;
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int co = c0 + (wv + 4 * j) * 16;
        b_ok[j] = co < p.Cout;
        b_rel[j] = co + half8;
    }
    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    const sy_buffer bufdy = sy_make_buffer(p.dy, p.dy_extent);
    const sy_lds_base_t lds0 = sy_lds_base(smem);
    PixelCursor cur;
    if (!LIN) cur.init(slab0 * SLAB + (lane >> 1), p.Ho, p.Wo);
    int m_next = slab0 * SLAB;                    // LIN: first pixel of the next slab to issue (wave-uniform)
    const int m_lane = lane >> 1;

    int issued = 0, stage_w = 0;
    auto issue_slab = [&]() {
        const unsigned stage = (unsigned)(stage_w * STAGE + wv * kSubPitch);
        const bool live = issued < nslab;
        if constexpr (LIN) {
            const bool m_ok = live && m_next + m_lane < p.M;
            const int xo = (m_next + m_lane) * p.ldx, yo = (m_next + m_lane) * p.lddy;
#pragma unroll
            for (int j = 0; j < LA; ++j)
                sy_glds16_buf_at(bufx, (unsigned)((xo + a_rel[j]) * 2) | ((m_ok && a_ok[j]) ? 0u : 0xFFFFFFFFu), lds0, stage + 4 * j * kSubPitch);
#pragma unroll
            for (int j = 0; j < LB; ++j)
                sy_glds16_buf_at(bufdy, (unsigned)((yo + b_rel[j]) * 2) | ((m_ok && b_ok[j]) ? 0u : 0xFFFFFFFFu), lds0,
                                 stage + (SA + 4 * j) * kSubPitch);
            m_next += SLAB;
        } else {
            const bool m_ok = live && cur.m < p.M;
            const int hb = cur.ho * p.stride - p.pad, wb = cur.wo * p.stride - p.pad;
            const int xo = cur.n * (int)p.xbs + (hb * p.W + wb) * p.ldx;
            const int yo = cur.n * (int)p.dybs + (cur.ho * p.Wo + cur.wo) * p.lddy;
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const int hi = hb + a_dh[j], wi = wb + a_dw[j];
                const bool ok = m_ok && a_ok[j] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                sy_glds16_buf_at(bufx, (unsigned)((xo + a_rel[j]) * 2) | (ok ? 0u : 0xFFFFFFFFu), lds0, stage + 4 * j * kSubPitch);
            }
#pragma unroll
            for (int j = 0; j < LB; ++j)
                sy_glds16_buf_at(bufdy, (unsigned)((yo + b_rel[j]) * 2) | ((m_ok && b_ok[j]) ? 0u : 0xFFFFFFFFu), lds0,
                                 stage + (SA + 4 * j) * kSubPitch);
            if (live) cur.advance(SLAB, p.Ho, p.Wo);
        }
        ++issued;
        stage_w = (stage_w + 1 == STG) ? 0 : stage_w + 1;
    };

    f32x16 acc[TR][TC];
#pragma unroll
    for (int t = 0; t < TR; ++t)
#pragma unroll
        for (int u = 0; u < TC; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.0f;

    // fragment gather: lane (g = lane>>5, parity = (lane>>4)&1, i = lane&15) addresses pixel row 8g + (i>>2)
    // [+4 for the second half of its 8 k-values], column quad i&3 of subtile 2*tile + parity
    const int lane_off = ((lane >> 4) & 1) * kSubPitch + ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;

    for (int j = 0; j < STG - 1; ++j) issue_slab();
    int stage_r = 0;
    for (int s = 0; s < nslab; ++s) {
        sy_wait_vmcnt<(STG - 2) * (LA + LB)>();   // slab s landed; the STG - 2 slabs behind it stay in flight
        sy_barrier();                             // ... for every wave; everyone is past slab s-1
        issue_slab();
        const unsigned char* const base = smem + stage_r * STAGE + lane_off;
        stage_r = (stage_r + 1 == STG) ? 0 : stage_r + 1;
        uint4 a[2][TR], b[2][TC];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                const unsigned char* ptr = base + ((wr * TR + t) * 2) * kSubPitch + ks * 512;
                const uint2 lo = sy_lds_read_tr16(ptr), hi = sy_lds_read_tr16(ptr + 128);
                a[ks][t] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
#pragma unroll
            for (int u = 0; u < TC; ++u) {
                const unsigned char* ptr = base + (SA + (wcn * TC + u) * 2) * kSubPitch + ks * 512;
                const uint2 lo = sy_lds_read_tr16(ptr), hi = sy_lds_read_tr16(ptr + 128);
                b[ks][u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < TR; ++t)
#pragma unroll
                for (int u = 0; u < TC; ++u) acc[t][u] = sy_mfma_group(T(), a[ks][t], b[ks][u], acc[t][u]);
    }
    sy_wait_vmcnt<0>();                           // the out-of-range pieces past the last slab (LDS must be quiet at exit)
    wgrad_epilogue<TR, TC>(p, acc, r0, c0, wr, wcn, lane, bid.z);
    SY_TL_END();
}

// ---- 3x3 stride-1 "all taps" variant (16-bit types) ------------------------------------------------------------
// The transpose-read kernel above treats every (tap, ci) row tile as its own GEMM tile: a 3x3 layer re-stages the x operand
// nine times (once per tap) and the dy operand Cin * 9 / 128 times.  Here a workgroup owns 32 input channels x ALL NINE taps
// (288 k-rows) x 128 output channels; a slab = 32 consecutive pixels of ONE image row, and the x operand of the slab is the
// 3 x 34 halo window around them, parked ONCE in LDS (row-major [pixel][16 channels] subtiles, LDS-DMA): the nine taps are
// nine pixel offsets ((kh * 34 + kw) * 32 bytes) of the same ds_read_b64_tr_b16 gather.  Per slab: 14.5 KB staged for 72
// MFMAs instead of 16 KB for 32.  Wave w owns output channels [32 w, 32 w + 32) and all nine taps: 9 accumulator tiles.
constexpr int kHaloPix = 3 * 34;                  // halo pixels of a slab (3 rows x (32 + 2))
constexpr int kXSub = 128 * 32 + 128;             // x subtile: 128 pixel rows (102 used) x 16 channels + bank padding

// The x fragments of step (k-half, tap) + 2 are read while step (k-half, tap) multiplies (ring of three fragments; reading 4 / 6 steps
// ahead was measured twice, round 4 and round 5 stage ze: no faster); one slab per rendezvous (two: +0.5-4 % alone, nothing in the
// step — round 5 stage j); rings of 6 / 8 slabs: no faster (round 4).
// Launch bounds: at most 256 registers per lane (accumulators included; the kernel takes ~180), so that two four-wave workgroups — or
// another chain's waves — share a CU with this kernel.  (Rounds 2-4 ran it without that bound: 464 registers, one workgroup owned
// the CU — tile codes 49 / 65, no longer instantiated.)
// CI2 = 2 (round 5): EIGHT waves, 64 input channels per workgroup — wave (cw = w & 3, ch = w >> 2) owns output channels [32 cw, + 32)
// x input channels [32 ch, + 32) x nine taps.  The dy slab (the larger stream, re-read by every input-channel tile of the layer) is
// staged once for twice the MFMAs: 21.3 KB of LDS-DMA per 144 MFMAs instead of 14.5 KB per 72 (-27 % L2 -> LDS bytes per MFMA), three
// DMA pieces per wave and slab instead of four.  Run chip-wide the four-wave kernel is bound by exactly that stream (profiles/r05 k).
//
// PIPE = 1 (round 5): the slab loop as ONE instruction stream.  In the loop above a slab is [wait + barrier, ~55 scalar / vector
// instructions of slab bookkeeping and DMA issue, ten fragment reads, a full LGKM drain] and THEN 18 MFMAs: the in-order wave issues
// nothing into the matrix pipe for ~45 % of a slab (main loop 45 us for 20.5 us of MFMA at 256->256 @38x60 x 16), and neither fewer
// L2 -> LDS bytes (tile 53), deeper fragment prefetch, two slabs per rendezvous nor deeper rings moved it.  Here
// the rendezvous of slab s + 1 sits in front of MFMA 9 of slab s (ring of >= 4 slabs: the DMA of slab s + STG - 1 is issued behind
// it, one piece behind each of the next MFMAs), the first fragments of slab s + 1 are read behind MFMAs 14 / 16 / 17 of slab s, and
// the fragment ring (18 % 3 == 0) runs on across the slab boundary — every MFMA has two fragment reads and at most one DMA piece or
// one extra fragment behind it, and no MFMA waits for a barrier.
template <typename T, int STG, int CI2 = 1, int PIPE = 0>
__global__ __launch_bounds__(kThreadsW * CI2, CI2 == 1 ? 2 : 1) void conv_wgrad9_kernel(WgradArgs p) {
    SY_TL_BEGIN(6);
    constexpr int CT = 128, CIT = 32 * CI2;
    constexpr int SB = CT / 16;                   // dy subtiles per slab
    constexpr int NWV = 4 * CI2;                  // waves
    constexpr int XP = 8 * CI2 / NWV, YP = SB / NWV;             // x / dy DMA pieces per wave and slab (2 / 2, or 2 / 1)
    constexpr int STAGE = 2 * CI2 * kXSub + SB * kSubPitch;
    static_assert(T::kEPC == 8, "16-bit elements");
    SY_DYN_SMEM(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = sy_uniform(tid >> 6);
    const sy_block_id bid = sy_xcd_block_id();
    const int ci0 = bid.x * CIT;
    const int c0 = bid.y * CT;
    const int wsegs = (p.Wo + 31) >> 5;
    const int slabs_total = p.N * p.Ho * wsegs;
    const int slab0 = bid.z * p.slabs_per_split;
    int nslab = p.slabs_per_split;
    if (slab0 + nslab > slabs_total) nslab = slabs_total - slab0;
    if (nslab <= 0) return;

    // ---- DMA assignment per slab: wave w issues x instructions {w, w + 4} (subtile j >> 2, 32-pixel quarter j & 3 of the
    //      flattened 3 x 34 window) and dy subtiles {w, w + 4}; lane -> (pixel lane >> 1, 16-byte channel half lane & 1).
    //      The kernel is instruction-issue bound (profiles/r02/r_pmc_wgrad.txt: 183 instructions per 18 MFMAs), so a slab's
    //      addresses are per-lane constants + ONE wave-uniform element offset per operand, validity is two unsigned compares per
    //      piece, and the ring is kept full past the last slab (out-of-range pieces: zeros into an idle stage) so that the wait in
    //      front of every slab is the same counted `vmcnt`.
    const sy_buffer bufx = sy_make_buffer(p.x, p.x_extent);
    const sy_buffer bufdy = sy_make_buffer(p.dy, p.dy_extent);
    const sy_lds_base_t lds0 = sy_lds_base(smem);
    const int half8 = (lane & 1) * 8;
    int x_rel[XP], x_hy1[XP], x_hx1[XP];            // element offset relative to pixel (cur_h, w0); window row / column - 1
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int j = wv + NWV * i;
        const int pp = (j & 3) * 32 + (lane >> 1);             // flattened halo pixel
        const int hy = pp / 34, hx = pp - hy * 34;
        x_hy1[i] = pp < kHaloPix ? hy - 1 : 0x40000000;        // beyond the window: never valid
        x_hx1[i] = hx - 1;
        x_rel[i] = ((hy - 1) * p.W + (hx - 1)) * p.ldx + ci0 + (j >> 2) * 16 + half8;
    }
    int y_rel[YP];
    bool y_cok[YP];
#pragma unroll
    for (int i = 0; i < YP; ++i) {
        const int co = c0 + (wv + NWV * i) * 16 + half8;
        y_cok[i] = co < p.Cout;
        y_rel[i] = (lane >> 1) * p.lddy + co;
    }
    const int y_px = lane >> 1;
    int cur_n, cur_h, cur_ws;                                   // slab cursor of the NEXT slab to issue
    {
        const int per_img = p.Ho * wsegs;
        cur_n = slab0 / per_img;
        const int rem = slab0 - cur_n * per_img;
        cur_h = rem / wsegs;
        cur_ws = rem - cur_h * wsegs;
    }
    int issued = 0, stage_w = 0;
    // one slab's DMA = issue_begin (wave-uniform bases of the slab under the cursor), XP + YP pieces, issue_end (cursor on)
    unsigned is_stage = 0;
    int is_w0 = 0, is_xbase = 0, is_ybase = 0;
    bool is_live = false;
    auto issue_begin = [&]() {
        is_stage = (unsigned)(stage_w * STAGE);
        is_w0 = cur_ws * 32;
        is_live = issued < nslab;
        is_xbase = cur_n * (int)p.xbs + (cur_h * p.W + is_w0) * p.ldx;        // wave-uniform
        is_ybase = cur_n * (int)p.dybs + (cur_h * p.Wo + is_w0) * p.lddy;
    };
    auto issue_piece = [&](auto i_) {
        constexpr int I = decltype(i_)::value;
        if constexpr (I < XP) {
            const int j = wv + NWV * I;
            const bool ok = is_live && (unsigned)(x_hy1[I] + cur_h) < (unsigned)p.H && (unsigned)(x_hx1[I] + is_w0) < (unsigned)p.W &&
                            !(p.ablate & 1);
            // (OR with an all-ones mask instead of a select: hipcc turns the select into an exec-mask branch around the add)
            sy_glds16_buf_at(bufx, (unsigned)((is_xbase + x_rel[I]) * 2) | (ok ? 0u : 0xFFFFFFFFu), lds0,
                             is_stage + (unsigned)((j >> 2) * kXSub + (j & 3) * 1024));
        } else {
            constexpr int Y = I - XP;
            const int j = wv + NWV * Y;
            const bool ok = is_live && y_cok[Y] && is_w0 + y_px < p.Wo && !(p.ablate & 2);
            sy_glds16_buf_at(bufdy, (unsigned)((is_ybase + y_rel[Y]) * 2) | (ok ? 0u : 0xFFFFFFFFu), lds0,
                             is_stage + (unsigned)(2 * CI2 * kXSub + j * kSubPitch));
        }
    };
    auto issue_end = [&]() {
        if (++cur_ws == wsegs) { cur_ws = 0; if (++cur_h == p.Ho) { cur_h = 0; ++cur_n; } }
        ++issued;
        stage_w = (stage_w + 1 == STG) ? 0 : stage_w + 1;
    };
    auto issue_slab = [&]() {
        issue_begin();
        sy_static_for<0, XP + YP>([&](auto i_) { issue_piece(i_); });
        issue_end();
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // fragment gather (see conv_wgrad_tr_kernel): lane (g = lane >> 5, parity = (lane >> 4) & 1, i = lane & 15) addresses pixel
    // row 8 g + (i >> 2) (+ 4 for the second half of its 8 k-values), channel quad i & 3 of subtile `parity`
    const int cw = CI2 > 1 ? (wv & 3) : wv, ch = CI2 > 1 ? (wv >> 2) : 0;      // output-channel block, input-channel half of this wave
    const int x_lane = (2 * ch + ((lane >> 4) & 1)) * kXSub + ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
    const int y_lane = ((lane >> 4) & 1) * kSubPitch + ((lane >> 5) * 8 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;

    static_assert(STG >= 3, "ring: one slab multiplies while at least one is in flight");
    sy_probe(0);
    if constexpr (PIPE) {
        static_assert(STG >= 4, "pipelined loop: ring of >= 4 slabs");
        constexpr int NP = XP + YP;
        constexpr int RV = 9;                     // the rendezvous of the NEXT slab sits in front of this MFMA
        for (int j = 0; j < STG - 1; ++j) issue_slab();
        sy_probe(1);
        sy_wait_vmcnt<NP * (STG - 2)>();          // slab 0 landed
        sy_barrier();
        sy_probe(2);
        uint4 a[3], b[2];
        auto read_a = [&](const unsigned char* xb, auto st_) {
            constexpr int ST = decltype(st_)::value;
            constexpr int KS = ST / 9, TAP = ST % 9;
            const unsigned char* ptr = xb + ((TAP / 3) * 34 + (TAP % 3)) * 32 + KS * 512;
            const uint2 lo = sy_lds_read_tr16(ptr), hi = sy_lds_read_tr16(ptr + 128);
            a[ST % 3] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        };
        auto read_b = [&](const unsigned char* yb, auto ks_) {
            constexpr int KS = decltype(ks_)::value;
            const uint2 lo = sy_lds_read_tr16(yb + KS * 512), hi = sy_lds_read_tr16(yb + KS * 512 + 128);
            b[KS] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        };
        const unsigned char* const x0 = smem + x_lane;
        const unsigned char* const y0 = smem + 2 * CI2 * kXSub + (cw * 2) * kSubPitch + y_lane;
        int stage_r = 0;
        read_b(y0, sy_int<0>());
        read_a(x0, sy_int<0>());
        read_a(x0, sy_int<1>());
        for (int s = 0; s < nslab; ++s) {
            const int stage_n = (stage_r + 1 == STG) ? 0 : stage_r + 1;
            const unsigned char* const xb = x0 + stage_r * STAGE;
            const unsigned char* const yb = y0 + stage_r * STAGE;
            const unsigned char* const xn = x0 + stage_n * STAGE;
            const unsigned char* const yn = y0 + stage_n * STAGE;
            stage_r = stage_n;
            sy_static_for<0, 18>([&](auto st_) {
                constexpr int ST = decltype(st_)::value;
                if constexpr (ST == RV) {
                    sy_wait_vmcnt<NP * (STG - 3)>();      // my pieces of slab s + 1 landed (slabs s + 2 .. s + STG - 2 stay in flight)
                    sy_barrier();                         // ... everyone's; and everyone is done with slab s - 1: its stage is free
                    issue_begin();
                }
                if constexpr (ST + 2 < 18) read_a(xb, sy_int<ST + 2>());
                else read_a(xn, sy_int<ST + 2 - 18>());
                if constexpr (ST == 1) read_b(yb, sy_int<1>());                       // (its registers: free since MFMA 17 of slab s - 1)
                if constexpr (ST >= RV && ST < RV + NP) issue_piece(sy_int<ST - RV>());
                if constexpr (ST == RV + NP) issue_end();
                if constexpr (ST == 14) read_b(yn, sy_int<0>());                      // (free since MFMA 8)
                acc[ST % 9] = sy_mfma_group(T(), a[ST % 3], b[ST / 9], acc[ST % 9]);
                sy_sched_fence();
            });
        }
    } else {
    for (int j = 0; j < STG - 1; ++j) issue_slab();
    sy_probe(1);
    int stage_r = 0;
    for (int s = 0; s < nslab; ++s) {
        sy_wait_vmcnt<(XP + YP) * (STG - 2)>();   // slab s landed; the STG - 2 slabs behind it stay in flight
        sy_barrier();                             // ... for every wave; everyone is past slab s - 1
        if (s == 0) sy_probe(2);
        issue_slab();                             // (a slab past the last one: zeros from the out-of-range pieces)
        const unsigned char* const xb = smem + stage_r * STAGE + x_lane;
        const unsigned char* const yb = smem + stage_r * STAGE + 2 * CI2 * kXSub + (cw * 2) * kSubPitch + y_lane;
        stage_r = (stage_r + 1 == STG) ? 0 : stage_r + 1;
        uint4 a[3], b[2];
        auto read_a = [&](auto st_) {
            constexpr int ST = decltype(st_)::value;
            constexpr int KS = ST / 9, TAP = ST % 9;
            const unsigned char* ptr = xb + ((TAP / 3) * 34 + (TAP % 3)) * 32 + KS * 512;
            const uint2 lo = sy_lds_read_tr16(ptr), hi = sy_lds_read_tr16(ptr + 128);
            a[ST % 3] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        };
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uint2 lo = sy_lds_read_tr16(yb + ks * 512), hi = sy_lds_read_tr16(yb + ks * 512 + 128);
            b[ks] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        read_a(sy_int<0>());
        read_a(sy_int<1>());
        sy_static_for<0, 18>([&](auto st_) {
            constexpr int ST = decltype(st_)::value;
            if constexpr (ST + 2 < 18) read_a(sy_int<ST + 2>());
            acc[ST % 9] = sy_mfma_group(T(), a[ST % 3], b[ST / 9], acc[ST % 9]);
            sy_sched_fence();
        });
    }
    }
    sy_wait_vmcnt<0>();                           // the out-of-range pieces past the last slab (LDS must be quiet at exit)
    sy_probe(3);

    // ---- epilogue: D[row = (tap, ci)][col = co]; partial slab of this split, or += into dW (one split)
    const int l31 = lane & 31, half = lane >> 5;
    const int co = c0 + cw * 32 + l31;
    if (co >= p.Cout) return;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ci = ci0 + ch * 32 + q * 8 + half * 4;
            const int kb = t * p.Cin + ci;
            const float v0 = acc[t][q * 4 + 0], v1 = acc[t][q * 4 + 1], v2 = acc[t][q * 4 + 2], v3 = acc[t][q * 4 + 3];
            if (p.splits > 1) {
                wgrad_store_slab(p, bid.z, co, kb, v0, v1, v2, v3);
            } else if (p.oihw) {
                float* row = p.dw + (long long)co * p.K + (long long)ci * 9 + t;
                row[0] += v0; row[9] += v1; row[18] += v2; row[27] += v3;
            } else {
                float4* dst = reinterpret_cast<float4*>(p.dw + (long long)co * p.K + kb);
                float4 o = *dst;
                o.x += v0; o.y += v1; o.z += v2; o.w += v3;
                *dst = o;
            }
        }
    sy_probe(6);
    SY_TL_END();
}

// dW (+)= sum over splits of the partial slabs; also applies the slab -> packed / OIHW layout change.
// Slabs are [split][K / 4][Cout][4] (coalesced stores in the weight-gradient kernels); the fold walks them in that order —
// coalesced reads of the `splits` x larger side — and scatters its single read-modify-write of dW.
// A workgroup = (256 / ZL) consecutive elements x ZL split lanes: many-split folds of small weight tensors are
// latency bound, so the split loop is spread over ZL threads per element and combined through LDS.
template <int ZL, bool BF>
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const float* part, float* dw, int Cout, int K, int Cin, int taps,
                                                         int splits, int oihw) {
    SY_TL_BEGIN(9);
    constexpr int E = 256 / ZL;
    __shared__ float red[256];
    const long long total = (long long)Cout * K;
    const int e = threadIdx.x % E, zl = threadIdx.x / E;
    const long long row4 = (long long)Cout * 4;
    for (long long base = (long long)blockIdx.x * E; base < total; base += (long long)gridDim.x * E) {
        const long long i = base + e;
        float v = 0.0f;
        if (i < total)
            for (int z = zl; z < splits; z += ZL)
                v += BF ? BF16::to_f32(reinterpret_cast<const unsigned short*>(part)[(long long)z * total + i]) : part[(long long)z * total + i];
        if (ZL > 1) {
            red[threadIdx.x] = v;
            __syncthreads();
            if (zl == 0) {
#pragma unroll
                for (int k = 1; k < ZL; ++k) v += red[k * E + e];
            }
            __syncthreads();
        }
        if (zl == 0 && i < total) {
            const long long kq = i / row4;
            const int rem = (int)(i - kq * row4);
            const int co = rem >> 2, k = (int)(kq * 4) + (rem & 3);
            long long o = (long long)co * K + k;
            if (oihw) {
                const int tap = k / Cin, ci = k - tap * Cin;
                o = (long long)co * K + (long long)ci * taps + tap;
            }
            const float nv = dw[o] + v;
            unsigned nb;
            __builtin_memcpy(&nb, &nv, 4);
#if SY_WT_FOLD
            sy_store4_wt(dw + o, nb);
#else
            dw[o] = nv;
#endif
        }
    }
    SY_TL_END();
}


// fold launch: ZL split lanes per element by split count, slab element type by mode
static int launch_fold(const WgradArgs& a, int splits, int taps, void* stream) {
    const long long work = (long long)a.Cout * a.K;
    auto grid_for = [&](int e) { long long b = (work + e - 1) / e; return (int)(b > 4096 ? 4096 : b); };
#define SY_FOLD(ZL, E)                                                                                                       \
    do {                                                                                                                     \
        if (a.part_bf16)                                                                                                     \
            SY_LAUNCH((wgrad_fold_kernel<ZL, true>), dim3(grid_for(E)), dim3(256), 0, stream, (const float*)a.part, a.dw,    \
                      a.Cout, a.K, a.Cin, taps, splits, a.oihw);                                                             \
        else                                                                                                                 \
            SY_LAUNCH((wgrad_fold_kernel<ZL, false>), dim3(grid_for(E)), dim3(256), 0, stream, (const float*)a.part, a.dw,   \
                      a.Cout, a.K, a.Cin, taps, splits, a.oihw);                                                             \
    } while (0)
    if (splits >= 64) SY_FOLD(16, 16);
    else if (splits >= 16) SY_FOLD(4, 64);
    else SY_FOLD(1, 256);
#undef SY_FOLD
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

template <typename T, int WR, int WC, int TR, int TC, int STG, int LIN>
int launch_tr_kernel_lin(const WgradArgs& a, dim3 grid, void* stream) {
    constexpr size_t smem = (size_t)STG * ((WR * TR + WC * TC) * 2) * kSubPitch;
#ifndef SY_EMU
    static sy_dev_once attr_done;
    if (attr_done.need()) {
        if (hipFuncSetAttribute((const void*)conv_wgrad_tr_kernel<T, WR, WC, TR, TC, STG, LIN>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return SY_ERR_LAUNCH;
        attr_done.mark();
    }
#endif
    SY_LAUNCH((conv_wgrad_tr_kernel<T, WR, WC, TR, TC, STG, LIN>), grid, dim3(kThreadsW), smem, stream, a);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

template <typename T, int WR, int WC, int TR, int TC, int STG>
int launch_tr_kernel(const WgradArgs& a, dim3 grid, void* stream) {
    if constexpr (T::kEPC == 8 && ((WR * TR + WC * TC) * 2) % 4 == 0) {
        // 1x1 stride 1 over dense tensors: linear pixel addressing
        const bool lin = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.Ho == a.H && a.Wo == a.W &&
                         a.xbs == (long long)a.H * a.W * a.ldx && a.dybs == (long long)a.Ho * a.Wo * a.lddy &&
                         (long long)a.M * a.ldx < 0x3fffffffLL && (long long)a.M * a.lddy < 0x3fffffffLL;
        return lin ? launch_tr_kernel_lin<T, WR, WC, TR, TC, STG, 1>(a, grid, stream)
                   : launch_tr_kernel_lin<T, WR, WC, TR, TC, STG, 0>(a, grid, stream);
    } else {
        return SY_ERR_UNSUPPORTED;
    }
}

template <typename T, int WR, int WC, int TR, int TC, int TRV = 0>
int launch_wgrad_cfg(WgradArgs a, long long ws_bytes, void* stream) {
    constexpr int RT = WR * TR * 32, CT = WC * TC * 32, SLAB = 4 * T::kEPC;
    const int gx = (a.K + RT - 1) / RT, gy = (a.Cout + CT - 1) / CT;
    const int slabs_total = (a.M + SLAB - 1) / SLAB;
    // about two workgroups per CU (256 CUs), at least 8 slabs of work each, and the slabs must fit the workspace
    const int target = a.target_blocks > 0 ? a.target_blocks : 512;
    int splits = (target + gx * gy - 1) / (gx * gy);
    const int max_splits = (slabs_total + 7) / 8;
    if (splits > max_splits) splits = max_splits;
    const long long slab_bytes = (long long)a.Cout * a.K * 4;
    if (a.part == nullptr || (a.K & 3)) splits = 1;               // the slab layout is in float4 groups along K
    else if ((long long)splits * slab_bytes > ws_bytes) splits = (int)(ws_bytes / slab_bytes);
    if (splits < 1) splits = 1;
    a.slabs_per_split = (slabs_total + splits - 1) / splits;
    splits = (slabs_total + a.slabs_per_split - 1) / a.slabs_per_split;
    a.splits = splits;
    if constexpr (TRV != 0) {
        // transpose-read variant preconditions; otherwise the scatter kernel of the same tile runs
        const bool tr_ok = T::kEPC == 8 && a.Cin % 16 == 0 && a.Cout % 16 == 0 && a.x_extent != 0 && a.dy_extent != 0;
        if (tr_ok) {
            const int rc = launch_tr_kernel<T, WR, WC, TR, TC, TRV>(a, dim3(gx, gy, splits), stream);
            if (rc != SY_OK) return rc;
        } else {
            SY_LAUNCH((conv_wgrad_kernel<T, WR, WC, TR, TC>), dim3(gx, gy, splits), dim3(kThreadsW), 0, stream, a);
        }
    } else {
        SY_LAUNCH((conv_wgrad_kernel<T, WR, WC, TR, TC>), dim3(gx, gy, splits), dim3(kThreadsW), 0, stream, a);
    }
    if (SY_LAUNCH_OK() != 0) return SY_ERR_LAUNCH;
    if (splits > 1) return launch_fold(a, splits, a.KH * a.KW, stream);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

template <typename T, int STG, int CI2 = 1, int PIPE = 0>
int launch_wgrad9(WgradArgs a, long long ws_bytes, void* stream) {
    if constexpr (T::kEPC != 8) {
        return SY_ERR_UNSUPPORTED;
    } else {
        if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Ho != a.H || a.Wo != a.W || a.Cin % (32 * CI2) != 0 ||
            a.Cout % 16 != 0 || a.x_extent == 0 || a.dy_extent == 0)
            return SY_ERR_UNSUPPORTED;
        const int gx = a.Cin / (32 * CI2), gy = (a.Cout + 127) / 128;
        const int slabs_total = a.N * a.Ho * ((a.Wo + 31) / 32);
        const int target = a.target_blocks > 0 ? a.target_blocks : 512;
        int splits = (target + gx * gy - 1) / (gx * gy);
        const int max_splits = (slabs_total + 7) / 8;
        if (splits > max_splits) splits = max_splits;
        const long long slab_bytes = (long long)a.Cout * a.K * 4;
        if (a.part == nullptr) splits = 1;
        else if ((long long)splits * slab_bytes > ws_bytes) splits = (int)(ws_bytes / slab_bytes);
        if (splits < 1) splits = 1;
        a.slabs_per_split = (slabs_total + splits - 1) / splits;
        splits = (slabs_total + a.slabs_per_split - 1) / a.slabs_per_split;
        a.splits = splits;
        constexpr size_t smem = (size_t)STG * (2 * CI2 * kXSub + 8 * kSubPitch);
#ifndef SY_EMU
        static sy_dev_once attr_done;
        if (attr_done.need()) {
            const void* fn = (const void*)conv_wgrad9_kernel<T, STG, CI2, PIPE>;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return SY_ERR_LAUNCH;
            attr_done.mark();
        }
#endif
        SY_LAUNCH((conv_wgrad9_kernel<T, STG, CI2, PIPE>), dim3(gx, gy, splits), dim3(kThreadsW * CI2), smem, stream, a);
        if (SY_LAUNCH_OK() != 0) return SY_ERR_LAUNCH;
        if (splits > 1) return launch_fold(a, splits, 9, stream);
        return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
    }
}

template <typename T>
int launch_wgrad_typed(const WgradArgs& a, long long ws_bytes, void* stream) {
    if (a.tile == 52) return launch_wgrad9<T, 3>(a, ws_bytes, stream);        // 3x3 stride 1: all nine taps per workgroup, halo in LDS
    if (a.tile == 59) return launch_wgrad9<T, 4, 1, 1>(a, ws_bytes, stream);  // ... the slab loop as one instruction stream (PIPE)
    if (a.tile == 60) return launch_wgrad9<T, 4, 2, 1>(a, ws_bytes, stream);  // ... on eight waves, 64 input channels per workgroup
    // (49 / 65: tile 52 without the register bound; 53: tile 60 without PIPE — measured references of rounds 2-5, never the tuner's
    //  choice any more, not instantiated)
    switch (a.tile) {          // (k rows x output channels) per workgroup
        case 1: return launch_wgrad_cfg<T, 2, 2, 2, 2>(a, ws_bytes, stream);   // 128 x 128
        case 2: return launch_wgrad_cfg<T, 4, 1, 1, 2>(a, ws_bytes, stream);   // 128 x  64
        case 3: return launch_wgrad_cfg<T, 4, 1, 1, 1>(a, ws_bytes, stream);   // 128 x  32
        case 4: return launch_wgrad_cfg<T, 2, 2, 1, 1>(a, ws_bytes, stream);   //  64 x  64
        case 5: return launch_wgrad_cfg<T, 1, 4, 2, 1>(a, ws_bytes, stream);   //  64 x 128
        case 6: return launch_wgrad_cfg<T, 2, 2, 1, 2>(a, ws_bytes, stream);   //  64 x 128 (2x2 waves)
        // + 16: transpose-read variants (LDS-DMA ring of 3 slabs + ds_read_b64_tr_b16), + 32: ring of 4
        case 17: return launch_wgrad_cfg<T, 2, 2, 2, 2, 3>(a, ws_bytes, stream);
        case 18: return launch_wgrad_cfg<T, 4, 1, 1, 2, 3>(a, ws_bytes, stream);
        case 20: return launch_wgrad_cfg<T, 2, 2, 1, 1, 3>(a, ws_bytes, stream);
        case 21: return launch_wgrad_cfg<T, 1, 4, 2, 1, 3>(a, ws_bytes, stream);
        case 22: return launch_wgrad_cfg<T, 2, 2, 1, 2, 3>(a, ws_bytes, stream);
        case 33: return launch_wgrad_cfg<T, 2, 2, 2, 2, 4>(a, ws_bytes, stream);
        case 34: return launch_wgrad_cfg<T, 4, 1, 1, 2, 4>(a, ws_bytes, stream);
        default: break;
    }
    // defaults: transpose-read variants (they fall back to the scatter kernel of the same tile when not applicable)
    if (a.Cout > 64) return launch_wgrad_cfg<T, 2, 2, 2, 2, 3>(a, ws_bytes, stream);  // 128 k x 128 co
    if (a.Cout > 32) return launch_wgrad_cfg<T, 4, 1, 1, 2, 3>(a, ws_bytes, stream);  // 128 k x  64 co
    return launch_wgrad_cfg<T, 4, 1, 1, 1>(a, ws_bytes, stream);                      // 128 k x  32 co
}

}  // namespace

extern "C" int sy_conv2d_wgrad(const sy_wgrad_desc* d, void* stream) {
    if (d == nullptr || d->x == nullptr || d->dy == nullptr || d->dw == nullptr) return SY_ERR_ARG;
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0) return SY_ERR_ARG;
    if (d->dtype < SY_DT_BF16 || d->dtype > SY_DT_F32) return SY_ERR_ARG;
    const int epc = d->dtype == SY_DT_F32 ? 4 : 8;
    if (d->Cin % epc || d->Cout % epc || d->ldx % epc || d->lddy % epc) return SY_ERR_UNSUPPORTED;
    if (d->stride != 1 && d->stride != 2) return SY_ERR_UNSUPPORTED;
    if ((long long)d->N * d->Ho * d->Wo > 0x7fffffffLL) return SY_ERR_UNSUPPORTED;
    WgradArgs a;
    a.x = (const unsigned char*)d->x; a.dy = (const unsigned char*)d->dy; a.dw = d->dw;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.ldx = d->ldx; a.lddy = d->lddy; a.xbs = d->xbs; a.dybs = d->dybs;
    a.M = d->N * d->Ho * d->Wo; a.K = d->KH * d->KW * d->Cin; a.slabs_per_split = 0;
    a.oihw = d->dw_oihw;
    a.tile = d->tile & 0xff;
    a.ablate = (d->tile >> 8) & 3;
    a.x_extent = (d->x_bytes > 0 && d->x_bytes < 0x7FFFFFF0LL) ? (unsigned)d->x_bytes : 0u;
    a.dy_extent = (d->dy_bytes > 0 && d->dy_bytes < 0x7FFFFFF0LL) ? (unsigned)d->dy_bytes : 0u;
    a.target_blocks = d->target_blocks;
    a.part = (float*)d->workspace;
    a.splits = 1;
    static const bool slab_bf16 = [] { const char* e = getenv("SY_WGRAD_SLAB_BF16"); return e == nullptr || atoi(e) != 0; }();
    a.part_bf16 = (d->dtype == SY_DT_BF16 && slab_bf16) ? 1 : 0;
    const long long wsb = d->workspace != nullptr ? (long long)d->workspace_bytes : 0;
    switch (d->dtype) {
        case SY_DT_BF16: return launch_wgrad_typed<BF16>(a, wsb, stream);
        case SY_DT_F16: return launch_wgrad_typed<F16>(a, wsb, stream);
        default: return launch_wgrad_typed<F32>(a, wsb, stream);
    }
}
SY_PROBE_READER(sy_probe_read_wgrad)
