// train_ops.hip — SPP pooling (forward + arg-max-routed backward) and the training-mode
// BatchNorm + SiLU passes around the MFMA convolutions.  All HBM-bound: 16 bytes per lane along the
// NHWC channel axis; a thread's channel chunk is FIXED across its grid-stride loop (the stride is a
// multiple of the chunks-per-pixel), so per-channel parameters live in registers.
//
// Replaces nn.BatchNorm2d (training mode, eps/momentum patched by init_yolo — cfgs/<name>.py:40-44 of
// the reference), nn.SiLU, SPPBottleneck's three nn.MaxPool2d + cat (exps/model/darknet.py:156) and
// their autograd backward kernels.
#include <stdlib.h>
#include "sy_pointwise.h"

#ifndef SY_BN_ROWS_IN_FLIGHT
#define SY_BN_ROWS_IN_FLIGHT 2          // rows (16-byte chunks per operand) a thread of the BatchNorm row kernels keeps in flight
#endif

namespace {

// ---- SPP: max over 5x5, 9x9, 13x13 windows, stride 1, -inf padding (nested windows share loads) ----
// Optional arg-max record (training): for level l and channel c the window offset (dh+6)*13+(dw+6)
// of the FIRST maximum in row-major scan order, one byte per (pixel, level, channel).
template <typename T>
__global__ __launch_bounds__(kBlock) void spp_pool_kernel(typename T::elem* buf, int N, int H, int W, int C, int ld,
                                                          long long bs, unsigned char* argmax) {
    const int cpp = C / T::kEPC;
    const long long total = (long long)N * H * W * cpp;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int cc = (int)(i % cpp);
        long long pix = i / cpp;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const int n = (int)(pix / ((long long)W * H));
        float m5[T::kEPC], m9[T::kEPC], m13[T::kEPC];
        unsigned char a5[T::kEPC], a9[T::kEPC], a13[T::kEPC];
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) { m5[j] = -INFINITY; m9[j] = -INFINITY; m13[j] = -INFINITY; a5[j] = 0; a9[j] = 0; a13[j] = 0; }
        const typename T::elem* base = buf + n * bs + cc * T::kEPC;
        for (int dh = -6; dh <= 6; ++dh) {
            const int hh = h + dh;
            if (hh < 0 || hh >= H) continue;
            const int ah = dh < 0 ? -dh : dh;
            for (int dw = -6; dw <= 6; ++dw) {
                const int ww = w + dw;
                if (ww < 0 || ww >= W) continue;
                const int aw = dw < 0 ? -dw : dw;
                const int d = ah > aw ? ah : aw;            // Chebyshev ring index
                const unsigned char code = (unsigned char)((dh + 6) * 13 + (dw + 6));
                Chunk<T> c = Chunk<T>::load(base + ((long long)hh * W + ww) * ld);
#pragma unroll
                for (int j = 0; j < T::kEPC; ++j) {
                    const float v = T::to_f32(c.e[j]);
                    if (v > m13[j]) { m13[j] = v; a13[j] = code; }
                    if (d <= 4 && v > m9[j]) { m9[j] = v; a9[j] = code; }
                    if (d <= 2 && v > m5[j]) { m5[j] = v; a5[j] = code; }
                }
            }
        }
        Chunk<T> o5, o9, o13;
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) { o5.e[j] = T::from_f32(m5[j]); o9.e[j] = T::from_f32(m9[j]); o13.e[j] = T::from_f32(m13[j]); }
        typename T::elem* dst = buf + n * bs + ((long long)h * W + w) * ld + cc * T::kEPC;
        o5.store(dst + C);
        o9.store(dst + 2 * C);
        o13.store(dst + 3 * C);
        if (argmax != nullptr) {
            unsigned char* am = argmax + (pix * 3) * C + cc * T::kEPC;
#pragma unroll
            for (int j = 0; j < T::kEPC; ++j) { am[j] = a5[j]; am[C + j] = a9[j]; am[2 * C + j] = a13[j]; }
        }
    }
}

// LDS-tiled form of the same pooling (H*W small enough for one image's channel chunk to sit in LDS, which is
// every SPP on the path: 19x30 at 600x960): one workgroup per (image, 16-byte channel chunk).  The square
// windows are separable — a horizontal pass keeps, per row and level, the row maximum and the dw of its FIRST
// occurrence, a vertical pass takes the first row (smallest dh) holding the maximum — which is exactly the first
// maximum of the row-major scan above, so pooled values AND arg-max bytes are bit-identical to
// spp_pool_kernel with 13 + 27 LDS reads per pixel instead of 169 global loads.
// LDS layout: in[HW] chunks | rv[3][HW] chunks (row maxima) | ra[3][HW][kEPC] bytes (dw + 6 of the row maximum).
// LSEL >= 0: only pooling level LSEL (5 + 4 LSEL wide) — the eval launch of a small batch runs the three levels as three
// workgroups per channel chunk (gridDim.y = 3: 64 workgroups of 39 us -> 192 of ~15 at one streamed frame); same arithmetic.
// Threads of a tile workgroup: one pass over its work items where that fits (19x30 at 600x960: 576 threads for the 570 pixels of the
// forward passes, 1024 for the 1710 (level, row-stage element) items of the backward's phase A) — round 6: with 256 threads the
// 512 workgroups of a training launch were 4 waves per CU (backward: 109 KB of LDS, one workgroup per CU) walking 3-7 dependent
// passes, 66 / 149 us per launch on each frame chain's critical path.
constexpr int kSppTileThreads = 1024;
inline int spp_tile_threads(long long items) {
    long long t = (items + 63) / 64 * 64;
    return (int)(t < 256 ? 256 : (t > kSppTileThreads ? kSppTileThreads : t));
}

template <typename T, int LSEL>
__device__ __forceinline__ void spp_pool_tile_body(typename T::elem* buf, int H, int W, int C, int ld, long long bs,
                                                   unsigned char* argmax, unsigned char* smem, int bx) {
    typedef typename T::elem elem;
    constexpr int E = T::kEPC;
    constexpr int L0 = LSEL < 0 ? 0 : LSEL, L1 = LSEL < 0 ? 3 : LSEL + 1, RH = 2 * L1;
    const int HW = H * W, nt = (int)blockDim.x;
    const int cpp = C / E;
    const int cc = bx % cpp, n = bx / cpp;
    unsigned char* s_in = smem;
    unsigned char* s_rv = smem + (size_t)HW * 16;
    unsigned char* s_ra = s_rv + (size_t)3 * HW * 16;
    elem* base = buf + n * bs + cc * E;
    for (int p = threadIdx.x; p < HW; p += nt) Chunk<T>::load(base + (long long)p * ld).store(s_in + (size_t)p * 16);
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += nt) {
        const int w = p % W, row = p - w;
        float m[3][E];
        unsigned char a[3][E];
#pragma unroll
        for (int l = 0; l < 3; ++l)
#pragma unroll
            for (int j = 0; j < E; ++j) { m[l][j] = -INFINITY; a[l][j] = 6; }
        for (int dw = -RH; dw <= RH; ++dw) {
            const int ww = w + dw;
            if (ww < 0 || ww >= W) continue;
            const int aw = dw < 0 ? -dw : dw;
            Chunk<T> c = Chunk<T>::load(s_in + (size_t)(row + ww) * 16);
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const float v = T::to_f32(c.e[j]);
                if (L1 == 3 && v > m[2][j]) { m[2][j] = v; a[2][j] = (unsigned char)(dw + 6); }
                if (L0 <= 1 && L1 >= 2 && aw <= 4 && v > m[1][j]) { m[1][j] = v; a[1][j] = (unsigned char)(dw + 6); }
                if (L0 == 0 && aw <= 2 && v > m[0][j]) { m[0][j] = v; a[0][j] = (unsigned char)(dw + 6); }
            }
        }
#pragma unroll
        for (int l = L0; l < L1; ++l) {
            Chunk<T> o;
#pragma unroll
            for (int j = 0; j < E; ++j) o.e[j] = T::from_f32(m[l][j]);
            o.store(s_rv + ((size_t)l * HW + p) * 16);
            __builtin_memcpy(s_ra + ((size_t)l * HW + p) * E, a[l], E);
        }
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += nt) {
        const int w = p % W, h = p / W;
        elem* dst = base + (long long)p * ld;
#pragma unroll
        for (int l = L0; l < L1; ++l) {
            const int r = 2 * (l + 1);
            float m[E];
            unsigned char a[E];
#pragma unroll
            for (int j = 0; j < E; ++j) { m[j] = -INFINITY; a[j] = 0; }
            for (int dh = -r; dh <= r; ++dh) {
                const int hh = h + dh;
                if (hh < 0 || hh >= H) continue;
                const int q = hh * W + w;
                Chunk<T> c = Chunk<T>::load(s_rv + ((size_t)l * HW + q) * 16);
                unsigned char ra[E];
                __builtin_memcpy(ra, s_ra + ((size_t)l * HW + q) * E, E);
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const float v = T::to_f32(c.e[j]);
                    if (v > m[j]) { m[j] = v; a[j] = (unsigned char)((dh + 6) * 13 + ra[j]); }
                }
            }
            Chunk<T> o;
#pragma unroll
            for (int j = 0; j < E; ++j) o.e[j] = T::from_f32(m[j]);
            o.store(dst + (l + 1) * C);
            if (argmax != nullptr)
                __builtin_memcpy(argmax + (((long long)n * HW + p) * 3 + l) * C + cc * E, a, E);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kSppTileThreads) void spp_pool_tile_kernel(typename T::elem* buf, int H, int W, int C, int ld,
                                                               long long bs, unsigned char* argmax) {
    SY_TL_BEGIN(15);
    SY_DYN_SMEM(smem);
    // XCD-contiguous order: a workgroup touches 16 bytes of every pixel's channel row, the workgroups of the neighbouring channel
    // chunks the rest of the same 128-byte lines — in hardware order they sit on eight different L2s and every XCD fetches every
    // line (round 6)
    const sy_block_id b = sy_xcd_block_id();
    if (gridDim.y == 1) spp_pool_tile_body<T, -1>(buf, H, W, C, ld, bs, argmax, smem, b.x);
    else if (b.y == 0) spp_pool_tile_body<T, 0>(buf, H, W, C, ld, bs, argmax, smem, b.x);
    else if (b.y == 1) spp_pool_tile_body<T, 1>(buf, H, W, C, ld, bs, argmax, smem, b.x);
    else spp_pool_tile_body<T, 2>(buf, H, W, C, ld, bs, argmax, smem, b.x);
    SY_TL_END();
}

// backward (gather form, deterministic, no atomics): source pixel (h,w) collects the pooled gradient of
// every window whose recorded arg-max is (h,w), and adds it onto slice 0's gradient.
template <typename T>
__global__ __launch_bounds__(kBlock) void spp_pool_bwd_kernel(typename T::elem* dbuf, const unsigned char* argmax, int N,
                                                              int H, int W, int C, int ld, long long bs) {
    const int cpp = C / T::kEPC;
    const long long total = (long long)N * H * W * cpp;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int cc = (int)(i % cpp);
        long long pix = i / cpp;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const int n = (int)(pix / ((long long)W * H));
        typename T::elem* gslot0 = dbuf + n * bs + ((long long)h * W + w) * ld + cc * T::kEPC;
        Chunk<T> g0 = Chunk<T>::load(gslot0);
        float acc[T::kEPC];
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) acc[j] = T::to_f32(g0.e[j]);
        for (int dh = -6; dh <= 6; ++dh) {
            const int ch = h - dh;                                  // window centre such that (h,w) sits at offset (dh,dw)
            if (ch < 0 || ch >= H) continue;
            const int ah = dh < 0 ? -dh : dh;
            for (int dw = -6; dw <= 6; ++dw) {
                const int cw = w - dw;
                if (cw < 0 || cw >= W) continue;
                const int aw = dw < 0 ? -dw : dw;
                const int d = ah > aw ? ah : aw;
                const unsigned char code = (unsigned char)((dh + 6) * 13 + (dw + 6));
                const long long cpix = ((long long)n * H + ch) * W + cw;
                const unsigned char* am = argmax + (cpix * 3) * C + cc * T::kEPC;
                const typename T::elem* gp = dbuf + n * bs + ((long long)ch * W + cw) * ld + cc * T::kEPC;
                const int l0 = d <= 2 ? 0 : (d <= 4 ? 1 : 2);       // smallest level whose window reaches (dh,dw)
                for (int l = l0; l < 3; ++l) {
                    unsigned char cb[8];
                    __builtin_memcpy(cb, am + l * C, T::kEPC);
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < T::kEPC; ++j) any = any || (cb[j] == code);
                    if (!any) continue;
                    Chunk<T> g = Chunk<T>::load(gp + (l + 1) * C);
#pragma unroll
                    for (int j = 0; j < T::kEPC; ++j)
                        if (cb[j] == code) acc[j] += T::to_f32(g.e[j]);
                }
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) o.e[j] = T::from_f32(acc[j]);
        o.store(gslot0);
    }
}

// LDS-tiled backward, separable like the forward: the arg-max of window (h, w, l) is reached through its row-stage
// element (h + dh, w) and that element's own dw, so the routing factors into
//   phase A  T[l][(h', w)] = sum of the pooled gradients of the <= 2r+1 centres of column w whose code selects row h'
//            (dh ascending), and the dw they all carry (one row-stage element has ONE arg-max);
//   phase B  source (h', x) += T[l][(h', x - dw)] for the <= 2r+1 row-stage elements whose dw points at x (l, dw ascending)
// — 27 + 27 LDS probes per pixel instead of the 507 of the scan form.  No atomics: deterministic.  The additions are
// grouped by row, so fp32 partial sums may differ from spp_pool_bwd_kernel's in the last bit.
// LDS layout: g[HW][3] chunks | am[HW][3][E] bytes | T[3][HW][E] fp32 | dw[3][HW][E] bytes (0xFF: no contribution).
template <typename T>
__global__ __launch_bounds__(kSppTileThreads) void spp_pool_bwd_tile_kernel(typename T::elem* dbuf, const unsigned char* argmax,
                                                                   int H, int W, int C, int ld, long long bs) {
    SY_TL_BEGIN(15);
    typedef typename T::elem elem;
    constexpr int E = T::kEPC;
    SY_DYN_SMEM(smem);
    const int HW = H * W, nt = (int)blockDim.x;
    const int cpp = C / E;
    const int bx = sy_xcd_block_id().x;                             // XCD-contiguous order, as in the forward kernel
    const int cc = bx % cpp, n = bx / cpp;
    unsigned char* s_g = smem;
    unsigned char* s_am = s_g + (size_t)HW * 3 * 16;
    float* s_T = reinterpret_cast<float*>(s_am + (size_t)HW * 3 * E);      // HW * 3 * (16 + E) is a multiple of 4
    unsigned char* s_dw = reinterpret_cast<unsigned char*>(s_T + (size_t)3 * HW * E);
    elem* base = dbuf + n * bs + cc * E;
    for (int i = threadIdx.x; i < HW * 3; i += nt) {
        const int p = i / 3, l = i - 3 * p;
        Chunk<T>::load(base + (long long)p * ld + (l + 1) * C).store(s_g + (size_t)i * 16);
        __builtin_memcpy(s_am + (size_t)i * E, argmax + (((long long)n * HW + p) * 3 + l) * C + cc * E, E);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HW * 3; i += nt) {            // phase A: (level, row-stage element)
        const int l = i / HW, p = i - l * HW;
        const int w = p % W, hr = p / W;
        const int r = 2 * (l + 1);
        float t[E];
        unsigned char rec[E];
#pragma unroll
        for (int j = 0; j < E; ++j) { t[j] = 0.0f; rec[j] = 0xFF; }
        for (int dh = -r; dh <= r; ++dh) {
            const int h = hr - dh;                                   // centre whose window row dh is this row
            if (h < 0 || h >= H) continue;
            const int q = h * W + w;
            unsigned char cb[E];
            __builtin_memcpy(cb, s_am + ((size_t)q * 3 + l) * E, E);
            Chunk<T> g = Chunk<T>::load(s_g + ((size_t)q * 3 + l) * 16);
            const int code0 = (dh + 6) * 13;
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const unsigned d = (unsigned)((int)cb[j] - code0);
                if (d < 13u) { t[j] += T::to_f32(g.e[j]); rec[j] = (unsigned char)d; }
            }
        }
        __builtin_memcpy(s_T + (size_t)i * E, t, sizeof(t));
        __builtin_memcpy(s_dw + (size_t)i * E, rec, E);
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += nt) {                // phase B: source pixels
        const int x = p % W, row = p - x;
        elem* gslot0 = base + (long long)p * ld;
        Chunk<T> g0 = Chunk<T>::load(gslot0);
        float acc[E];
#pragma unroll
        for (int j = 0; j < E; ++j) acc[j] = T::to_f32(g0.e[j]);
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int r = 2 * (l + 1);
            for (int dw = -r; dw <= r; ++dw) {
                const int w = x - dw;
                if (w < 0 || w >= W) continue;
                const size_t i = (size_t)l * HW + row + w;
                unsigned char rec[E];
                float t[E];
                __builtin_memcpy(rec, s_dw + i * E, E);
                __builtin_memcpy(t, s_T + i * E, sizeof(t));
#pragma unroll
                for (int j = 0; j < E; ++j)
                    if (rec[j] == (unsigned char)(dw + 6)) acc[j] += t[j];
            }
        }
        Chunk<T> o;
#pragma unroll
        for (int j = 0; j < E; ++j) o.e[j] = T::from_f32(acc[j]);
        o.store(gslot0);
    }
    SY_TL_END();
}

// ---- training-mode BatchNorm -------------------------------------------------------------------------
// The conv epilogue spreads its per-channel partial sums over `copies` replicas (contention control);
// finalize folds them, updates the running statistics and emits the per-channel affine.
// One workgroup = 32 channels; thread = (channel c = tid & 31, replica group r = tid >> 5): the replica
// loads of a channel are independent and run in parallel, then an LDS fold.
// Many replica rows (the exact mode: one row per convolution workgroup, thousands per layer) are first folded into the
// first kFoldTo rows IN PLACE by fold_rows_kernel: output row r = the rows r, r + kFoldTo, r + 2 kFoldTo, ... added in index order
// by four partial sums (rows j = q mod 4 of that list) that are combined in order q — a fixed tree, so the result does not
// depend on scheduling — with (C / 64) x kFoldTo workgroups instead of eight serial threads per channel.
constexpr int kFoldTo = 32, kFoldAbove = 64;
__global__ __launch_bounds__(kBlock) void fold_rows_kernel(float* a0, float* a1, int C, int copies) {
    __shared__ float part[4][64];
    float* const arr = (blockIdx.z & 1) ? a1 : a0;
    float* const base = arr + (long long)(blockIdx.z >> 1) * copies * C;
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, r = blockIdx.y;
    float acc = 0.0f;
    if (c < C) {
        int j = q;
        const int J = (copies - r + kFoldTo - 1) / kFoldTo;            // rows of this class
        for (; j + 12 < J; j += 16) {                                 // four independent loads in flight, added in index order
            const float v0 = base[(long long)(r + kFoldTo * j) * C + c], v1 = base[(long long)(r + kFoldTo * (j + 4)) * C + c];
            const float v2 = base[(long long)(r + kFoldTo * (j + 8)) * C + c], v3 = base[(long long)(r + kFoldTo * (j + 12)) * C + c];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; j < J; j += 4) acc += base[(long long)(r + kFoldTo * j) * C + c];
    }
    part[q][cl] = acc;
    __syncthreads();
    if (q == 0 && c < C) base[(long long)r * C + c] = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
}

// `copies` rows are read, `stride` rows lie between the segments (stride > copies after fold_rows_kernel)
__global__ __launch_bounds__(kBlock) void bn_finalize_kernel(const float* sum, const float* sqsum, int C, int copies, int stride,
                                                             double count, const float* gamma, const float* beta,
                                                             float eps, float momentum, float* running_mean,
                                                             float* running_var, float* scale, float* shift,
                                                             float* mean_out, float* invstd_out) {
    SY_TL_BEGIN(11);
    __shared__ float s_s[8][32], s_q[8][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    {   // segment blockIdx.y: statistics [seg][copies][C], outputs [seg][C] (gamma / beta are shared)
        const long long so = (long long)blockIdx.y * stride * C, ao = (long long)blockIdx.y * C;
        sum += so; sqsum += so; scale += ao; shift += ao;
        if (mean_out != nullptr) mean_out += ao;
        if (invstd_out != nullptr) invstd_out += ao;
    }
    float s = 0.0f, q = 0.0f;
    if (c < C)
        for (int k = rg; k < copies; k += 8) { s += sum[(long long)k * C + c]; q += sqsum[(long long)k * C + c]; }
    s_s[rg][cl] = s;
    s_q[rg][cl] = q;
    __syncthreads();
    if (rg != 0 || c >= C) return;
    double ds = 0.0, dq = 0.0;
    for (int k = 0; k < 8; ++k) { ds += (double)s_s[k][cl]; dq += (double)s_q[k][cl]; }
    const double mean = ds / count;
    double var = dq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - (float)mean * sc;
    if (mean_out != nullptr) mean_out[c] = (float)mean;
    if (invstd_out != nullptr) invstd_out[c] = invstd;
    if (running_mean != nullptr) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    SY_TL_END();
}

// Running statistics of every BatchNorm of the step in one launch (sy_bn_running_update): workgroup = (entry,
// 32-channel block); the replica fold repeats bn_finalize_kernel's order so both see the same batch statistics.
__global__ __launch_bounds__(kBlock) void bn_running_update_kernel(const sy_bn_running_entry* entries) {
    SY_TL_BEGIN(15);
    __shared__ float s_s[8][32], s_q[8][32];
    const sy_bn_running_entry e = entries[blockIdx.x];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.y * 32 + cl;
    if ((int)blockIdx.y * 32 >= e.C) return;                     // uniform for the workgroup
    float rm = 0.0f, rv = 0.0f;
    const long long ld = e.ld > 0 ? e.ld : e.C;                  // replica pitch (channel slice of a wider statistics array)
    if (rg == 0 && c < e.C) { rm = e.running_mean[c]; rv = e.running_var[c]; }
    for (int j = 0; j < e.calls; ++j) {
        float s = 0.0f, q = 0.0f;
        if (c < e.C)
            for (int k = rg; k < e.copies; k += 8) { s += e.sum[j][k * ld + c]; q += e.sqsum[j][k * ld + c]; }
        s_s[rg][cl] = s;
        s_q[rg][cl] = q;
        __syncthreads();
        if (rg == 0 && c < e.C) {
            double ds = 0.0, dq = 0.0;
            for (int k = 0; k < 8; ++k) { ds += (double)s_s[k][cl]; dq += (double)s_q[k][cl]; }
            const double count = e.count[j];
            const double mean = ds / count;
            double var = dq / count - mean * mean;
            if (var < 0.0) var = 0.0;
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            rm = (1.0f - e.momentum) * rm + e.momentum * (float)mean;
            rv = (1.0f - e.momentum) * rv + e.momentum * (float)unbiased;
        }
        __syncthreads();
    }
    if (rg == 0 && c < e.C) { e.running_mean[c] = rm; e.running_var[c] = rv; }
    if (blockIdx.y == 0 && threadIdx.x == 0 && e.num_batches_tracked != nullptr) *e.num_batches_tracked += e.calls;
    SY_TL_END();
}

template <typename T>
__global__ __launch_bounds__(kBlock) void bn_silu_apply_kernel(const typename T::elem* y, int ldy, const float* scale,
                                                               const float* shift, const typename T::elem* res, int ldr,
                                                               typename T::elem* out, int ldo, long long pixels, int C) {
    SY_TL_BEGIN(10);
    const int cpp = C / T::kEPC;
    const int cc = threadIdx.x % cpp;
    const int c0 = cc * T::kEPC;
    const int rows = kBlock / cpp;
    if ((int)threadIdx.x >= rows * cpp) return;        // idle tail when cpp does not divide the workgroup
    {   // segment blockIdx.y: rows [seg * pixels, (seg + 1) * pixels), affine [seg][C]
        const long long ro = (long long)blockIdx.y * pixels;
        y += ro * ldy; out += ro * ldo; scale += blockIdx.y * C; shift += blockIdx.y * C;
        if (res != nullptr) res += ro * ldr;
    }
    // two rows in flight per thread: the next row's loads are issued before this row's arithmetic, and the FIRST row's
    // before the per-channel parameters (independent latencies overlap: these launches last 10-20 us, a dependent chain of
    // three memory round trips at their head was a quarter of that)
    const long long step = (long long)gridDim.x * rows;
    long long pix = (long long)blockIdx.x * rows + threadIdx.x / cpp;
    constexpr int D = SY_BN_ROWS_IN_FLIGHT;
    Chunk<T> vq[D], rq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const long long pd = pix + d * step;
        if (pd < pixels) {
            vq[d] = Chunk<T>::load(y + pd * ldy + c0);
            if (res != nullptr) rq[d] = Chunk<T>::load(res + pd * ldr + c0);
        }
    }
    float sc[T::kEPC], sh[T::kEPC];
#pragma unroll
    for (int j = 0; j < T::kEPC; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
    for (; pix < pixels; pix += step) {
        const Chunk<T> v = vq[0], rv = rq[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) { vq[d] = vq[d + 1]; rq[d] = rq[d + 1]; }
        const long long nxt = pix + D * step;
        if (nxt < pixels) {
            vq[D - 1] = Chunk<T>::load(y + nxt * ldy + c0);
            if (res != nullptr) rq[D - 1] = Chunk<T>::load(res + nxt * ldr + c0);
        }
        Chunk<T> o;
        float r[T::kEPC];
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) r[j] = res != nullptr ? T::to_f32(rv.e[j]) : 0.0f;
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) o.e[j] = T::from_f32(sy_silu(T::to_f32(v.e[j]) * sc[j] + sh[j]) + r[j]);
        if constexpr (SY_WT_BNF) o.store_wt(out + pix * ldo + c0); else o.store(out + pix * ldo + c0);
    }
    SY_TL_END();
}

// bn_finalize + bn_silu_apply in ONE launch.  A dependent launch on the step's critical path costs ~13 us whatever it does
// (kernel boundary: drain, L2 write-back, dispatch — measured by skipping the 128 finalize launches of an l step: -1.7 ms,
// profiles/r02/s_*), so the 4 us finalize kernel is folded into the apply pass: a workgroup owns a channel SLICE (CS <= 64
// channels = one 128-byte line per pixel row, blockIdx.z) of a row range, folds the statistic replicas of ITS slice only
// (2 x copies x CS loads per workgroup, in the order bn_finalize_kernel / bn_running_update_kernel use: eight float partials per
// channel, summed in double), derives the affine in LDS, and streams its rows; the first row's loads are in flight meanwhile.
// The workgroups with blockIdx.x == 0 also publish scale / shift / mean / invstd for the backward pass.
template <typename T>
__global__ __launch_bounds__(kBlock) void bn_finalize_apply_kernel(const float* sum, const float* sqsum, int copies, double count,
                                                                   const float* gamma, const float* beta, float eps,
                                                                   float* scale, float* shift, float* mean_out,
                                                                   float* invstd_out, const typename T::elem* y, int ldy,
                                                                   const typename T::elem* res, int ldr,
                                                                   typename T::elem* out, int ldo, long long pixels, int C,
                                                                   int CS) {
    SY_TL_BEGIN(10);
    __shared__ float s_part[2][8][64];
    __shared__ float s_aff[2][64];
    const int cpp = CS / T::kEPC;
    const int rows = kBlock / cpp;
    const int cc = threadIdx.x % cpp;
    const int cb = blockIdx.z * CS;
    const int c0 = cb + cc * T::kEPC;
    {   // segment blockIdx.y: rows [seg * pixels, (seg + 1) * pixels), statistics [seg][copies][C], affine [seg][C]
        const long long ro = (long long)blockIdx.y * pixels;
        const long long so = (long long)blockIdx.y * copies * C;
        const int ao = blockIdx.y * C;
        y += ro * ldy; out += ro * ldo;
        if (res != nullptr) res += ro * ldr;
        sum += so; sqsum += so; scale += ao; shift += ao; mean_out += ao; invstd_out += ao;
    }
    const bool live = (int)threadIdx.x < rows * cpp;
    const long long step = (long long)gridDim.x * rows;
    long long pix = (long long)blockIdx.x * rows + threadIdx.x / cpp;
    Chunk<T> v, rv, vn, rn;
    if (live && pix < pixels) {
        v = Chunk<T>::load(y + pix * ldy + c0);
        if (res != nullptr) rv = Chunk<T>::load(res + pix * ldr + c0);
    }
    // ---- fold the replicas of this slice: item = (kind, replica group rg, channel c)
    for (int it = threadIdx.x; it < 2 * 8 * CS; it += kBlock) {
        const int c = it % CS, rg = (it / CS) & 7, kind = it / (8 * CS);
        const float* src = (kind ? sqsum : sum) + cb + c;
        float a = 0.0f;
        for (int k = rg; k < copies; k += 8) a += src[(long long)k * C];
        s_part[kind][rg][c] = a;
    }
    __syncthreads();
    if ((int)threadIdx.x < CS) {
        const int c = threadIdx.x;
        double ds = 0.0, dq = 0.0;
        for (int k = 0; k < 8; ++k) { ds += (double)s_part[0][k][c]; dq += (double)s_part[1][k][c]; }
        const double mean = ds / count;
        double var = dq / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[cb + c] * invstd;
        const float sh = beta[cb + c] - (float)mean * sc;
        s_aff[0][c] = sc;
        s_aff[1][c] = sh;
        if (blockIdx.x == 0) {
            scale[cb + c] = sc; shift[cb + c] = sh; mean_out[cb + c] = (float)mean; invstd_out[cb + c] = invstd;
        }
    }
    __syncthreads();
    if (!live) return;
    float sc[T::kEPC], sh[T::kEPC];
#pragma unroll
    for (int j = 0; j < T::kEPC; ++j) { sc[j] = s_aff[0][cc * T::kEPC + j]; sh[j] = s_aff[1][cc * T::kEPC + j]; }
    for (; pix < pixels; pix += step) {
        const long long nxt = pix + step;
        if (nxt < pixels) {
            vn = Chunk<T>::load(y + nxt * ldy + c0);
            if (res != nullptr) rn = Chunk<T>::load(res + nxt * ldr + c0);
        }
        Chunk<T> o;
        float r[T::kEPC];
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) r[j] = res != nullptr ? T::to_f32(rv.e[j]) : 0.0f;
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) o.e[j] = T::from_f32(sy_silu(T::to_f32(v.e[j]) * sc[j] + sh[j]) + r[j]);
        if constexpr (SY_WT_BNF) o.store_wt(out + pix * ldo + c0); else o.store(out + pix * ldo + c0);
        v = vn; rv = rn;
    }
    SY_TL_END();
}

// reduce: sums[0:C] += sum dz, sums[C:2C] += sum dz*xhat.  Thread = (pixel row, channel chunk); register
// partials, LDS tree over the rows of the workgroup, then ONE atomic per channel per workgroup.
// A workgroup owns a channel SLICE of CS channels (blockIdx.z; 64 channels = one 128-byte line per pixel row), not all C:
// the closing atomics are what this kernel waits for — the device retires ~40 G float atomics/s whatever their
// addresses (profiles/r02/k_stats_copies.txt), 1024 workgroups x 2 C atomics were 13 us of a 21 us launch at C = 256 and 27 us
// at C = 512 — and with slices a launch issues (workgroups x 2 CS) of them whatever C is.
template <typename T>
__global__ __launch_bounds__(kBlock) void bn_silu_bwd_reduce_kernel(const typename T::elem* y, int ldy,
                                                                    const typename T::elem* da, int ldda,
                                                                    const float* scale, const float* shift,
                                                                    const float* mean, const float* invstd, float* sums,
                                                                    long long pixels, int C, int CS, int copies) {
    SY_TL_BEGIN(12);
    __shared__ float red[kBlock * 2 * 8];
    const int cpp = CS / T::kEPC;
    const int rows = kBlock / cpp;
    const int cc = threadIdx.x % cpp;
    const int pr = threadIdx.x / cpp;
    const int cb = blockIdx.z * CS;                    // first channel of the slice
    const int c0 = cb + cc * T::kEPC;
    {   // segment blockIdx.y
        const long long ro = (long long)blockIdx.y * pixels;
        const int ao = blockIdx.y * C;
        y += ro * ldy; da += ro * ldda; scale += ao; shift += ao; mean += ao; invstd += ao;
        sums += (long long)blockIdx.y * copies * 2 * C;
    }
    sy_probe(0);
    const long long first = pr < rows ? (long long)blockIdx.x * rows + pr : pixels;     // idle tail threads skip the loop
    const long long step = (long long)gridDim.x * rows;
    constexpr int D = SY_BN_ROWS_IN_FLIGHT;   // rows in flight per thread; the first rows' loads go out before the parameters'
    Chunk<T> yq[D], gq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const long long pd = first + d * step;
        if (pd < pixels) { yq[d] = Chunk<T>::load(y + pd * ldy + c0); gq[d] = Chunk<T>::load(da + pd * ldda + c0); }
    }
    float sc[T::kEPC], sh[T::kEPC], mu[T::kEPC], is[T::kEPC], s0[T::kEPC], s1[T::kEPC];
#pragma unroll
    for (int j = 0; j < T::kEPC; ++j) {
        sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j];
        s0[j] = 0.0f; s1[j] = 0.0f;
    }
    sy_probe(1);
    for (long long pix = first; pix < pixels; pix += step) {
        const Chunk<T> yv = yq[0], gv = gq[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) { yq[d] = yq[d + 1]; gq[d] = gq[d + 1]; }
        const long long nxt = pix + D * step;
        if (nxt < pixels) { yq[D - 1] = Chunk<T>::load(y + nxt * ldy + c0); gq[D - 1] = Chunk<T>::load(da + nxt * ldda + c0); }
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) {
            const float yy = T::to_f32(yv.e[j]);
            const float dz = T::to_f32(gv.e[j]) * sy_silu_grad(yy * sc[j] + sh[j]);
            s0[j] += dz;
            s1[j] += dz * ((yy - mu[j]) * is[j]);
        }
    }
    sy_probe(3);
#pragma unroll
    for (int j = 0; j < T::kEPC; ++j) {
        red[(j * 2 + 0) * kBlock + threadIdx.x] = s0[j];
        red[(j * 2 + 1) * kBlock + threadIdx.x] = s1[j];
    }
    __syncthreads();
    // thread t < cpp*EPC*2 folds one (chunk, element, kind) column over the `rows` pixel rows
    for (int t = threadIdx.x; t < cpp * T::kEPC * 2; t += kBlock) {
        const int kind = t & 1, j = (t >> 1) % T::kEPC, ch = (t >> 1) / T::kEPC;
        float v = 0.0f;
        for (int r = 0; r < rows; ++r) v += red[(j * 2 + kind) * kBlock + r * cpp + ch];
        atomicAdd(sums + (long long)(blockIdx.x % copies) * 2 * C + kind * C + cb + ch * T::kEPC + j, v);
    }
    sy_probe(6);
    SY_TL_END();
}

template <typename T>
__global__ __launch_bounds__(kBlock) void bn_silu_bwd_apply_kernel(const typename T::elem* y, int ldy,
                                                                   const typename T::elem* da, int ldda,
                                                                   const float* scale, const float* shift,
                                                                   const float* mean, const float* invstd,
                                                                   const float* gamma, const float* sums,
                                                                   typename T::elem* dy, int lddy, long long pixels,
                                                                   int C, int copies, float* dgamma, float* dbeta,
                                                                   long long seg_sum_stride, typename T::elem* dres,
                                                                   int lddres, int dres_acc) {
    SY_TL_BEGIN(13);
    // fold the replicas of the two reduction sums once per workgroup, cooperatively, through LDS
    __shared__ float s_fold[2 * 1024];
    {   // segment blockIdx.y
        const long long ro = (long long)blockIdx.y * pixels;
        const int ao = blockIdx.y * C;
        y += ro * ldy; da += ro * ldda; dy += ro * lddy; scale += ao; shift += ao; mean += ao; invstd += ao;
        sums += blockIdx.y * seg_sum_stride;
        if (dres != nullptr) dres += ro * lddres;
    }
    const int cpp = C / T::kEPC;
    const int rows = kBlock / cpp;
    const int cc = threadIdx.x % cpp;
    const int c0 = cc * T::kEPC;
    const float inv_m = 1.0f / (float)pixels;
    // the first row's loads go out before the replica fold, its barrier and the parameter loads (independent latencies overlap)
    const long long step = (long long)gridDim.x * rows;
    long long pix = (long long)blockIdx.x * rows + threadIdx.x / cpp;
    const bool live = (int)threadIdx.x < rows * cpp;
    constexpr int D = SY_BN_ROWS_IN_FLIGHT;   // rows in flight per thread
    Chunk<T> yq[D], gq[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const long long pd = pix + d * step;
        if (live && pd < pixels) { yq[d] = Chunk<T>::load(y + pd * ldy + c0); gq[d] = Chunk<T>::load(da + pd * ldda + c0); }
    }
    for (int i = threadIdx.x; i < 2 * C; i += kBlock) {
        float a = 0.0f;
        for (int k = 0; k < copies; ++k) a += sums[(long long)k * 2 * C + i];
        s_fold[i] = a;
    }
    __syncthreads();
    if (dres_acc & 4) {                                 // row 1 holds the RAW moment sum dz*y (a data gradient's fused reduce,
        for (int c = threadIdx.x; c < C; c += kBlock)   // SY_EPI_BNR): sum dz*xhat = invstd * (sum dz*y - mean * sum dz)
            // in double: for a channel with |mean| >> std the fp32 difference cancels catastrophically
            s_fold[C + c] = (float)((double)invstd[c] * ((double)s_fold[C + c] - (double)mean[c] * (double)s_fold[c]));
        __syncthreads();
    }
    if (blockIdx.x == 0 && dgamma != nullptr) {
        if (gridDim.y == 1 && !(dres_acc & 2)) {        // launches on one stream are ordered: plain += is race free
            for (int c = threadIdx.x; c < C; c += kBlock) { dbeta[c] += s_fold[c]; dgamma[c] += s_fold[C + c]; }
        } else {                                        // the segments (frames) share gamma / beta (bit 1 of dres_acc: the two
                                                        // frames' launches run on different streams: atomics as well)
            for (int c = threadIdx.x; c < C; c += kBlock) { atomicAdd(dbeta + c, s_fold[c]); atomicAdd(dgamma + c, s_fold[C + c]); }
        }
    }
    if ((int)threadIdx.x >= rows * cpp) return;
    float sc[T::kEPC], sh[T::kEPC], mu[T::kEPC], is[T::kEPC], gi[T::kEPC], m0[T::kEPC], m1[T::kEPC];
#pragma unroll
    for (int j = 0; j < T::kEPC; ++j) {
        const int c = c0 + j;
        sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; is[j] = invstd[c];
        gi[j] = gamma[c] * invstd[c];
        m0[j] = s_fold[c] * inv_m;
        m1[j] = s_fold[C + c] * inv_m;
    }
    for (; pix < pixels; pix += step) {
        Chunk<T> yv = yq[0], gv = gq[0];
#pragma unroll
        for (int d = 0; d + 1 < D; ++d) { yq[d] = yq[d + 1]; gq[d] = gq[d + 1]; }
        const long long nxt = pix + D * step;
        if (nxt < pixels) { yq[D - 1] = Chunk<T>::load(y + nxt * ldy + c0); gq[D - 1] = Chunk<T>::load(da + nxt * ldda + c0); }
        Chunk<T> o;
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) {
            const float yy = T::to_f32(yv.e[j]);
            const float dz = T::to_f32(gv.e[j]) * sy_silu_grad(yy * sc[j] + sh[j]);
            o.e[j] = T::from_f32(gi[j] * (dz - m0[j] - (yy - mu[j]) * is[j] * m1[j]));
        }
        if constexpr (SY_WT_BNB) o.store_wt(dy + pix * lddy + c0); else o.store(dy + pix * lddy + c0);
        if (dres != nullptr) {          // y = silu(bn(conv)) + res: the residual branch receives da unchanged (view_copy fused)
            typename T::elem* dst = dres + pix * lddres + c0;
            if (dres_acc & 1) {
                Chunk<T> r = Chunk<T>::load(dst);
#pragma unroll
                for (int j = 0; j < T::kEPC; ++j) gv.e[j] = T::from_f32(T::to_f32(gv.e[j]) + T::to_f32(r.e[j]));
            }
            if constexpr (SY_WT_BNB) gv.store_wt(dst); else gv.store(dst);
        }
    }
    SY_TL_END();
}

// sums[0][i] = sum over replicas k of sums[k][i]  (i in [0, 2C)): run once before the apply pass
// (64 elements x 4 row classes per workgroup: class q adds the rows k = q mod 4 in index order, the four partial sums are combined
//  in order q — a fixed tree; the exact mode folds 768 rows per layer here, a serial loop per element was 30 % of its step)
__global__ __launch_bounds__(kBlock) void fold_replicas_kernel(float* sums, int n, int copies) {
    __shared__ float part[4][64];
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + el;
    sums += (long long)blockIdx.y * copies * n;
    float a = 0.0f;
    if (i < n) {
        int k = q;
        for (; k + 12 < copies; k += 16) {
            const float v0 = sums[(long long)k * n + i], v1 = sums[(long long)(k + 4) * n + i];
            const float v2 = sums[(long long)(k + 8) * n + i], v3 = sums[(long long)(k + 12) * n + i];
            a += v0; a += v1; a += v2; a += v3;
        }
        for (; k < copies; k += 4) a += sums[(long long)k * n + i];
    }
    part[q][el] = a;
    __syncthreads();
    if (q == 0 && i < n) sums[i] = ((part[0][el] + part[1][el]) + part[2][el]) + part[3][el];
}

inline bool chunk_rows_ok(int C, int e) { const int cpp = C / e; return cpp >= 1 && cpp <= kBlock; }

inline int env_cap(const char* name, int dflt) {          // tuning knob (tools/): workgroups per launch of the row kernels
    const char* v = getenv(name);
    return (v != nullptr && atoi(v) > 0) ? atoi(v) : dflt;
}

// workgroup caps of the four row kernels (sy_bn_grid_caps): environment defaults read once, then explicit values
enum { kCapApply = 0, kCapFApply = 1, kCapReduce = 2, kCapBApply = 3 };
inline std::atomic<int>& bn_cap(int which) {
    static std::atomic<int> caps[4] = {{env_cap("SY_BN_APPLY_BLOCKS", 2048)}, {env_cap("SY_BN_FAPPLY_BLOCKS", 2048)},
                                       {env_cap("SY_BN_REDUCE_BLOCKS", 768)}, {env_cap("SY_BN_BAPPLY_BLOCKS", 1024)}};
    return caps[which];
}

inline int row_grid(long long pixels, int C, int e, int cap) {
    const int rows = kBlock / (C / e);
    long long b = (pixels + rows - 1) / rows;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// the LDS-tiled SPP kernels take up to 144 KiB of dynamic LDS (of the CU's 160 KiB); larger feature maps use the scan kernels
constexpr size_t kSppTileLds = 144 * 1024;
inline bool spp_force_scan() { const char* v = getenv("SY_SPP_SCAN"); return v != nullptr && v[0] == '1'; }
inline bool spp_lds_ok(const void* fn, int slot) {           // slot = kernel (0 fwd, 1 bwd) * 3 + dtype: set once each
#ifdef SY_EMU
    return true;
#else
    static sy_dev_once done[6];
    if (done[slot].need()) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSppTileLds) != hipSuccess) return false;
        done[slot].mark();
    }
    return true;
#endif
}

}  // namespace

extern "C" int sy_bn_grid_caps(const int32_t* set4, int32_t* get4) {
    for (int k = 0; k < 4; ++k) {
        if (set4 != nullptr && set4[k] > 0) bn_cap(k).store(set4[k], std::memory_order_relaxed);
        if (get4 != nullptr) get4[k] = bn_cap(k).load(std::memory_order_relaxed);
    }
    return SY_OK;
}

extern "C" int sy_spp_pool(void* buf, int N, int H, int W, int C, int ld, int64_t bs, void* argmax, int dtype,
                           void* stream) {
    if (buf == nullptr || N <= 0 || C <= 0 || ld < 4 * C) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ld % e) return SY_ERR_UNSUPPORTED;
    const long long work = (long long)N * H * W * (C / e);
    const size_t tile_lds = (size_t)H * W * (16 + 3 * 16 + 3 * e);      // in | row maxima | row arg-max bytes
    if (tile_lds <= kSppTileLds && !spp_force_scan()) {
        SY_DISPATCH_DTYPE(dtype, if (!spp_lds_ok((const void*)spp_pool_tile_kernel<T>, T::kCode)) return SY_ERR_LAUNCH;
                          SY_LAUNCH((spp_pool_tile_kernel<T>), dim3(N * (C / e), N * (C / e) < 256 ? 3 : 1), dim3(spp_tile_threads((long long)H * W)), tile_lds, stream,
                                    (typename T::elem*)buf, H, W, C, ld, (long long)bs, (unsigned char*)argmax));
    }
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((spp_pool_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       (typename T::elem*)buf, N, H, W, C, ld, (long long)bs, (unsigned char*)argmax));
}

extern "C" int sy_spp_pool_bwd(void* dbuf, const void* argmax, int N, int H, int W, int C, int ld, int64_t bs, int dtype,
                               void* stream) {
    if (argmax == nullptr || dbuf == nullptr || N <= 0 || C <= 0 || ld < 4 * C) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ld % e) return SY_ERR_UNSUPPORTED;
    const long long work = (long long)N * H * W * (C / e);
    const size_t tile_lds = (size_t)H * W * 3 * (16 + e + 4 * e + e);    // pooled gradients | arg-max bytes | row sums | row dw
    if (tile_lds <= kSppTileLds && !spp_force_scan()) {
        SY_DISPATCH_DTYPE(dtype, if (!spp_lds_ok((const void*)spp_pool_bwd_tile_kernel<T>, 3 + T::kCode)) return SY_ERR_LAUNCH;
                          SY_LAUNCH((spp_pool_bwd_tile_kernel<T>), dim3(N * (C / e)), dim3(spp_tile_threads(3LL * H * W)), tile_lds, stream,
                                    (typename T::elem*)dbuf, (const unsigned char*)argmax, H, W, C, ld, (long long)bs));
    }
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((spp_pool_bwd_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       (typename T::elem*)dbuf, (const unsigned char*)argmax, N, H, W, C, ld,
                                       (long long)bs));
}

extern "C" int sy_bn_finalize(const float* sum, const float* sqsum, int C, int copies, double count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              float* scale, float* shift, float* mean, float* invstd, int nseg, void* stream) {
    if (sum == nullptr || sqsum == nullptr || gamma == nullptr || beta == nullptr || scale == nullptr ||
        shift == nullptr || C <= 0 || copies <= 0 || count <= 0.0)
        return SY_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SY_ERR_ARG;
    if (nseg < 1 || (nseg > 1 && running_mean != nullptr)) return SY_ERR_ARG;   // running stats of several calls: sy_bn_running_update
    int read = copies;
    if (copies > kFoldAbove) {          // (writes the folded rows back into the caller's arrays: rows [0, kFoldTo) then hold the totals)
        SY_LAUNCH(fold_rows_kernel, dim3((C + 63) / 64, kFoldTo, 2 * nseg), dim3(kBlock), 0, stream, const_cast<float*>(sum),
                  const_cast<float*>(sqsum), C, copies);
        read = kFoldTo;
    }
    SY_LAUNCH(bn_finalize_kernel, dim3((C + 31) / 32, nseg), dim3(kBlock), 0, stream, sum, sqsum, C, read, copies, count,
              gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

extern "C" int sy_bn_running_update(const sy_bn_running_entry* entries, int n_entries, int max_C, void* stream) {
    if (entries == nullptr || n_entries <= 0 || max_C <= 0) return SY_ERR_ARG;
    SY_LAUNCH(bn_running_update_kernel, dim3(n_entries, (max_C + 31) / 32), dim3(kBlock), 0, stream, entries);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

extern "C" int sy_bn_silu_apply(const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldr,
                                void* out, int ldo, int64_t pixels, int C, int dtype, int nseg, void* stream) {
    if (y == nullptr || out == nullptr || scale == nullptr || shift == nullptr || pixels <= 0 || C <= 0 || nseg < 1) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ldy % e || ldo % e || (res != nullptr && ldr % e)) return SY_ERR_UNSUPPORTED;
    if (!chunk_rows_ok(C, e)) return SY_ERR_UNSUPPORTED;
    const int cap_apply = bn_cap(kCapApply).load(std::memory_order_relaxed);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((bn_silu_apply_kernel<T>), dim3(row_grid(pixels, C, e, cap_apply / nseg), nseg), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)y, ldy, scale, shift, (const typename T::elem*)res, ldr,
                                       (typename T::elem*)out, ldo, (long long)pixels, C));
}

extern "C" int sy_bn_finalize_apply(const float* sum, const float* sqsum, int copies, double count, const float* gamma,
                                    const float* beta, float eps, float* scale, float* shift, float* mean, float* invstd,
                                    const void* y, int ldy, const void* res, int ldr, void* out, int ldo, int64_t pixels, int C,
                                    int dtype, int nseg, void* stream) {
    if (sum == nullptr || sqsum == nullptr || gamma == nullptr || beta == nullptr || scale == nullptr || shift == nullptr ||
        mean == nullptr || invstd == nullptr || y == nullptr || out == nullptr || copies <= 0 || count <= 0.0 || pixels <= 0 ||
        C <= 0 || nseg < 1)
        return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ldy % e || ldo % e || (res != nullptr && ldr % e)) return SY_ERR_UNSUPPORTED;
    // channel slice of a workgroup: the largest chunk multiple <= 64 channels that divides C
    int CS = C;
    for (int c = (64 / e) * e; c >= e; c -= e)
        if (c <= C && C % c == 0) { CS = c; break; }
    if (CS > 64) return SY_ERR_UNSUPPORTED;
    const int nsl = C / CS;
    const int cap_fa = bn_cap(kCapFApply).load(std::memory_order_relaxed);
    int cap = cap_fa / (nseg * nsl);
    if (cap < 1) cap = 1;
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((bn_finalize_apply_kernel<T>), dim3(row_grid(pixels, CS, e, cap), nseg, nsl), dim3(kBlock), 0,
                                       stream, sum, sqsum, copies, count, gamma, beta, eps, scale, shift, mean, invstd,
                                       (const typename T::elem*)y, ldy, (const typename T::elem*)res, ldr,
                                       (typename T::elem*)out, ldo, (long long)pixels, C, CS));
}

extern "C" int sy_bn_silu_bwd_reduce(const void* y, int ldy, const void* da, int ldda, const float* scale,
                                     const float* shift, const float* mean, const float* invstd, float* sums,
                                     int copies, int64_t pixels, int C, int dtype, int nseg, void* stream) {
    if (y == nullptr || da == nullptr || sums == nullptr || pixels <= 0 || C <= 0 || copies <= 0 || nseg < 1) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ldy % e || ldda % e) return SY_ERR_UNSUPPORTED;
    if (!chunk_rows_ok(C, e)) return SY_ERR_UNSUPPORTED;
    const int cap_reduce = bn_cap(kCapReduce).load(std::memory_order_relaxed);
    static const int slice_max = env_cap("SY_BN_REDUCE_SLICE", 64);
    // channel slice of a workgroup: the largest chunk multiple <= 64 channels that divides C (64 = one 128-byte line per row)
    int CS = C;
    for (int c = (slice_max / e) * e; c >= e; c -= e)
        if (c <= C && C % c == 0) { CS = c; break; }
    const int nsl = C / CS;
    int cap = cap_reduce / (nseg * nsl);
    if (cap < 1) cap = 1;
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((bn_silu_bwd_reduce_kernel<T>), dim3(row_grid(pixels, CS, e, cap), nseg, nsl), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)y, ldy, (const typename T::elem*)da, ldda, scale, shift,
                                       mean, invstd, sums, (long long)pixels, C, CS, copies));
}

extern "C" int sy_bn_silu_bwd_apply(const void* y, int ldy, const void* da, int ldda, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, const float* gamma,
                                    const float* sums, int copies, void* dy, int lddy, int64_t pixels, int C,
                                    float* dgamma, float* dbeta, void* dres, int lddres, int dres_accumulate, int dtype,
                                    int nseg, void* stream) {
    if (y == nullptr || da == nullptr || sums == nullptr || dy == nullptr || pixels <= 0 || C <= 0 || copies <= 0 || nseg < 1)
        return SY_ERR_ARG;
    const long long seg_sum_stride = (long long)copies * 2 * C;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ldy % e || ldda % e || lddy % e || (dres != nullptr && lddres % e)) return SY_ERR_UNSUPPORTED;
    if (!chunk_rows_ok(C, e) || C > 1024) return SY_ERR_UNSUPPORTED;
    static const int fold_inline = env_cap("SY_BN_FOLD_INLINE", 2);      // replicas <= this: every workgroup folds them itself
    if (copies > fold_inline) {       // many replicas: fold once here instead of in every workgroup of the apply pass
        SY_LAUNCH(fold_replicas_kernel, dim3((2 * C + 63) / 64, nseg), dim3(kBlock), 0, stream,
                  const_cast<float*>(sums), 2 * C, copies);
        copies = 1;
    }
    // (grid caps re-swept after the instruction diet of these kernels, profiles/r02/ai_*, aj_*: fewer, fatter workgroups —
    //  apply 4096 -> 2048, backward reduce 1024 -> 768, backward apply 2048 -> 1024: -0.16 ... -0.27 ms per l step)
    const int cap_bapply = bn_cap(kCapBApply).load(std::memory_order_relaxed);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((bn_silu_bwd_apply_kernel<T>), dim3(row_grid(pixels, C, e, cap_bapply / nseg), nseg), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)y, ldy, (const typename T::elem*)da, ldda, scale, shift,
                                       mean, invstd, gamma, sums, (typename T::elem*)dy, lddy, (long long)pixels, C, copies,
                                       dgamma, dbeta, seg_sum_stride, (typename T::elem*)dres, lddres, dres_accumulate));
}
SY_PROBE_READER(sy_probe_read_train_ops)
