// pointwise.hip — the HBM-bound glue kernels of the path: every one moves 16 bytes per lane,
// coalesced along the NHWC channel axis, grid-strided, and writes straight into the channel slice
// its consumer reads (no torch.cat / split / interpolate kernels survive — SURVEY.md traps T3-T6).
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

namespace {

constexpr int kBlock = 256;

inline int grid_for(long long work) {
    long long b = (work + kBlock - 1) / kBlock;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;        // 256 CUs x 8 workgroups, grid-stride beyond
    return (int)b;
}

template <typename T> struct Chunk {
    typedef typename T::elem elem;
    static constexpr int N = T::kEPC;
    elem e[N];
    __device__ __forceinline__ static Chunk load(const void* p) {
        Chunk c;
        uint4 v = *reinterpret_cast<const uint4*>(p);
        __builtin_memcpy(c.e, &v, 16);
        return c;
    }
    __device__ __forceinline__ void store(void* p) const {
        uint4 v;
        __builtin_memcpy(&v, e, 16);
        *reinterpret_cast<uint4*>(p) = v;
    }
};

// ---- Focus: NCHW fp32 planes -> NHWC 16-channel (12 used) ------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void focus_pack_kernel(const float* in, int N, int Ctot, int c0, int H, int W,
                                                            typename T::elem* out) {
    typedef typename T::elem elem;
    const int H2 = H >> 1, W2 = W >> 1;
    const long long total = (long long)N * H2 * W2;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int w2 = (int)(i % W2);
        const int h2 = (int)((i / W2) % H2);
        const int n = (int)(i / ((long long)W2 * H2));
        elem o[16];
        // patch order TL, BL, TR, BR (yolox Focus; trap T4): q -> (dy, dx) = (q&1, q>>1)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int hh = 2 * h2 + (q & 1), ww = 2 * w2 + (q >> 1);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                o[q * 3 + c] = T::from_f32(in[(((long long)n * Ctot + c0 + c) * H + hh) * W + ww]);
        }
#pragma unroll
        for (int c = 12; c < 16; ++c) o[c] = T::from_f32(0.0f);
        uint4* dst = reinterpret_cast<uint4*>(out + i * 16);
        constexpr int NV = (int)(16 * sizeof(elem) / 16);
        uint4 v[NV];
        __builtin_memcpy(v, o, sizeof(o));
#pragma unroll
        for (int k = 0; k < NV; ++k) dst[k] = v[k];
    }
}

// ---- nearest resize to a target size (PyTorch 'nearest': src = min(floor(dst * in/out), in-1)) ----
__device__ __forceinline__ int nearest_src(int dst, int in_sz, int out_sz) {
    const float scale = (float)in_sz / (float)out_sz;
    int s = (int)floorf((float)dst * scale);
    return s < in_sz - 1 ? s : in_sz - 1;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void resize_nearest_kernel(const typename T::elem* in, int N, int Hi, int Wi, int C,
                                                                int ldi, long long ibs, typename T::elem* out, int Ho,
                                                                int Wo, int ldo, long long obs) {
    const int cpp = C / T::kEPC;                       // chunks per pixel
    const long long total = (long long)N * Ho * Wo * cpp;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int cc = (int)(i % cpp);
        long long pix = i / cpp;
        const int wo = (int)(pix % Wo);
        const int ho = (int)((pix / Wo) % Ho);
        const int n = (int)(pix / ((long long)Wo * Ho));
        const int hs = nearest_src(ho, Hi, Ho), ws = nearest_src(wo, Wi, Wo);
        Chunk<T> c = Chunk<T>::load(in + n * ibs + ((long long)hs * Wi + ws) * ldi + cc * T::kEPC);
        c.store(out + n * obs + ((long long)ho * Wo + wo) * ldo + cc * T::kEPC);
    }
}

// backward: every source pixel sums the destination pixels that read it (gather form, no atomics)
template <typename T>
__global__ __launch_bounds__(kBlock) void resize_nearest_bwd_kernel(const typename T::elem* dout, int N, int Ho, int Wo,
                                                                    int C, int lddo, long long dobs,
                                                                    typename T::elem* din, int Hi, int Wi, int lddi,
                                                                    long long dibs, int accumulate) {
    const int cpp = C / T::kEPC;
    const long long total = (long long)N * Hi * Wi * cpp;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int cc = (int)(i % cpp);
        long long pix = i / cpp;
        const int ws = (int)(pix % Wi);
        const int hs = (int)((pix / Wi) % Hi);
        const int n = (int)(pix / ((long long)Wi * Hi));
        float acc[T::kEPC];
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) acc[j] = 0.0f;
        // candidate destination rows/cols: the preimage of hs under nearest_src is a short run near hs*Ho/Hi
        int h_lo = (int)((long long)hs * Ho / Hi) - 1; if (h_lo < 0) h_lo = 0;
        int w_lo = (int)((long long)ws * Wo / Wi) - 1; if (w_lo < 0) w_lo = 0;
        const int h_span = Ho / Hi + 3, w_span = Wo / Wi + 3;
        for (int dh = 0; dh < h_span; ++dh) {
            const int ho = h_lo + dh;
            if (ho >= Ho || nearest_src(ho, Hi, Ho) != hs) continue;
            for (int dw = 0; dw < w_span; ++dw) {
                const int wo = w_lo + dw;
                if (wo >= Wo || nearest_src(wo, Wi, Wo) != ws) continue;
                Chunk<T> c = Chunk<T>::load(dout + n * dobs + ((long long)ho * Wo + wo) * lddo + cc * T::kEPC);
#pragma unroll
                for (int j = 0; j < T::kEPC; ++j) acc[j] += T::to_f32(c.e[j]);
            }
        }
        typename T::elem* dst = din + n * dibs + ((long long)hs * Wi + ws) * lddi + cc * T::kEPC;
        Chunk<T> o;
        if (accumulate) {
            Chunk<T> old = Chunk<T>::load(dst);
#pragma unroll
            for (int j = 0; j < T::kEPC; ++j) acc[j] += T::to_f32(old.e[j]);
        }
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) o.e[j] = T::from_f32(acc[j]);
        o.store(dst);
    }
}

// ---- SPP: max over 5x5, 9x9, 13x13 windows, stride 1, -inf padding (nested windows share loads) ----
template <typename T>
__global__ __launch_bounds__(kBlock) void spp_pool_kernel(typename T::elem* buf, int N, int H, int W, int C, int ld,
                                                          long long bs) {
    const int cpp = C / T::kEPC;
    const long long total = (long long)N * H * W * cpp;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int cc = (int)(i % cpp);
        long long pix = i / cpp;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const int n = (int)(pix / ((long long)W * H));
        float m5[T::kEPC], m9[T::kEPC], m13[T::kEPC];
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) { m5[j] = -INFINITY; m9[j] = -INFINITY; m13[j] = -INFINITY; }
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
                Chunk<T> c = Chunk<T>::load(base + ((long long)hh * W + ww) * ld);
#pragma unroll
                for (int j = 0; j < T::kEPC; ++j) {
                    const float v = T::to_f32(c.e[j]);
                    m13[j] = fmaxf(m13[j], v);
                    if (d <= 4) m9[j] = fmaxf(m9[j], v);
                    if (d <= 2) m5[j] = fmaxf(m5[j], v);
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
    }
}

// backward of the three max-pools: route each pooled gradient to the arg-max position (first max in
// row-major window order, as ATen's max_pool2d does) and add onto slice 0's gradient.  Gather form:
// each source pixel scans the windows that contain it.
template <typename T>
__global__ __launch_bounds__(kBlock) void spp_pool_bwd_kernel(const typename T::elem* buf, typename T::elem* dbuf, int N,
                                                              int H, int W, int C, int ld, long long bs) {
    const long long total = (long long)N * H * W * C;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int c = (int)(i % C);
        long long pix = i / C;
        const int w = (int)(pix % W);
        const int h = (int)((pix / W) % H);
        const int n = (int)(pix / ((long long)W * H));
        const typename T::elem* xb = buf + n * bs + c;
        const typename T::elem* gb = dbuf + n * bs + c;
        const float xv = T::to_f32(xb[((long long)h * W + w) * ld]);
        float g = T::to_f32(gb[((long long)h * W + w) * ld]);
        for (int lvl = 0; lvl < 3; ++lvl) {
            const int R = 2 + 2 * lvl;                              // window radius 2, 4, 6
            // windows centred at (ch, cw) with |ch-h|<=R, |cw-w|<=R contain (h, w)
            for (int ch = h - R; ch <= h + R; ++ch) {
                if (ch < 0 || ch >= H) continue;
                for (int cw = w - R; cw <= w + R; ++cw) {
                    if (cw < 0 || cw >= W) continue;
                    const float pooled = T::to_f32(xb[((long long)ch * W + cw) * ld + (lvl + 1) * C]);
                    if (pooled != xv) continue;
                    // (h,w) is the arg-max iff no earlier (row-major) element of the window equals the max
                    bool first = true;
                    for (int hh = ch - R; hh <= h && first; ++hh) {
                        if (hh < 0 || hh >= H) continue;
                        const int w_end = (hh == h) ? w - 1 : cw + R;
                        for (int ww = cw - R; ww <= w_end; ++ww) {
                            if (ww < 0 || ww >= W) continue;
                            if (T::to_f32(xb[((long long)hh * W + ww) * ld]) == xv) { first = false; break; }
                        }
                    }
                    if (first) g += T::to_f32(gb[((long long)ch * W + cw) * ld + (lvl + 1) * C]);
                }
            }
        }
        dbuf[n * bs + ((long long)h * W + w) * ld + c] = T::from_f32(g);
    }
}

// ---- view copy / accumulate -----------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void view_copy_kernel(const typename T::elem* in, int ldi, typename T::elem* out,
                                                           int ldo, long long pixels, int C, int accumulate) {
    const int cpp = C / T::kEPC;
    const long long total = pixels * cpp;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int cc = (int)(i % cpp);
        const long long pix = i / cpp;
        Chunk<T> c = Chunk<T>::load(in + pix * ldi + cc * T::kEPC);
        typename T::elem* dst = out + pix * ldo + cc * T::kEPC;
        if (accumulate) {
            Chunk<T> o = Chunk<T>::load(dst);
#pragma unroll
            for (int j = 0; j < T::kEPC; ++j) c.e[j] = T::from_f32(T::to_f32(c.e[j]) + T::to_f32(o.e[j]));
        }
        c.store(dst);
    }
}

// ---- training-mode BatchNorm -------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void bn_finalize_kernel(const float* sum, const float* sqsum, int C, double count,
                                                             const float* gamma, const float* beta, float eps,
                                                             float momentum, float* running_mean, float* running_var,
                                                             float* scale, float* shift, float* mean_out,
                                                             float* invstd_out) {
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= C) return;
    const double mean = (double)sum[c] / count;
    double var = (double)sqsum[c] / count - mean * mean;
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
}

template <typename T>
__global__ __launch_bounds__(kBlock) void bn_silu_apply_kernel(const typename T::elem* y, int ldy, const float* scale,
                                                               const float* shift, const typename T::elem* res, int ldr,
                                                               typename T::elem* out, int ldo, long long pixels, int C) {
    const int cpp = C / T::kEPC;
    const long long total = pixels * cpp;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int cc = (int)(i % cpp);
        const long long pix = i / cpp;
        const int c0 = cc * T::kEPC;
        Chunk<T> v = Chunk<T>::load(y + pix * ldy + c0);
        Chunk<T> o;
        float r[T::kEPC];
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) r[j] = 0.0f;
        if (res != nullptr) {
            Chunk<T> rv = Chunk<T>::load(res + pix * ldr + c0);
#pragma unroll
            for (int j = 0; j < T::kEPC; ++j) r[j] = T::to_f32(rv.e[j]);
        }
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j)
            o.e[j] = T::from_f32(sy_silu(T::to_f32(v.e[j]) * scale[c0 + j] + shift[c0 + j]) + r[j]);
        o.store(out + pix * ldo + c0);
    }
}

// Each workgroup owns a strip of pixels and all C channels; a thread owns one 16-byte channel chunk
// (fixed across its loop) so per-channel partial sums live in registers; then LDS tree + atomics.
template <typename T>
__global__ __launch_bounds__(kBlock) void bn_silu_bwd_reduce_kernel(const typename T::elem* y, int ldy,
                                                                    const typename T::elem* da, int ldda,
                                                                    const float* scale, const float* shift,
                                                                    const float* mean, const float* invstd, float* sums,
                                                                    long long pixels, int C) {
    const int cpp = C / T::kEPC;
    const int rows = kBlock / cpp > 0 ? kBlock / cpp : 1;          // pixel rows processed per sweep
    const int cc = threadIdx.x % cpp;
    const int pr = threadIdx.x / cpp;
    float s0[T::kEPC], s1[T::kEPC];
#pragma unroll
    for (int j = 0; j < T::kEPC; ++j) { s0[j] = 0.0f; s1[j] = 0.0f; }
    const int c0 = cc * T::kEPC;
    if (pr < rows && cpp <= kBlock) {
        for (long long pix = (long long)blockIdx.x * rows + pr; pix < pixels; pix += (long long)gridDim.x * rows) {
            Chunk<T> yv = Chunk<T>::load(y + pix * ldy + c0);
            Chunk<T> gv = Chunk<T>::load(da + pix * ldda + c0);
#pragma unroll
            for (int j = 0; j < T::kEPC; ++j) {
                const float yy = T::to_f32(yv.e[j]);
                const float z = yy * scale[c0 + j] + shift[c0 + j];
                const float dz = T::to_f32(gv.e[j]) * sy_silu_grad(z);
                const float xhat = (yy - mean[c0 + j]) * invstd[c0 + j];
                s0[j] += dz;
                s1[j] += dz * xhat;
            }
        }
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) {
            atomicAdd(sums + c0 + j, s0[j]);
            atomicAdd(sums + C + c0 + j, s1[j]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void bn_silu_bwd_apply_kernel(const typename T::elem* y, int ldy,
                                                                   const typename T::elem* da, int ldda,
                                                                   const float* scale, const float* shift,
                                                                   const float* mean, const float* invstd,
                                                                   const float* gamma, const float* sums,
                                                                   typename T::elem* dy, int lddy, long long pixels,
                                                                   int C, float* dgamma, float* dbeta) {
    const int cpp = C / T::kEPC;
    const long long total = pixels * cpp;
    const float inv_m = 1.0f / (float)pixels;
    if (blockIdx.x == 0 && dgamma != nullptr) {        // launches on one stream are ordered: plain += is race free
        for (int c = threadIdx.x; c < C; c += kBlock) { dgamma[c] += sums[C + c]; dbeta[c] += sums[c]; }
    }
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int cc = (int)(i % cpp);
        const long long pix = i / cpp;
        const int c0 = cc * T::kEPC;
        Chunk<T> yv = Chunk<T>::load(y + pix * ldy + c0);
        Chunk<T> gv = Chunk<T>::load(da + pix * ldda + c0);
        Chunk<T> o;
#pragma unroll
        for (int j = 0; j < T::kEPC; ++j) {
            const int c = c0 + j;
            const float yy = T::to_f32(yv.e[j]);
            const float z = yy * scale[c] + shift[c];
            const float dz = T::to_f32(gv.e[j]) * sy_silu_grad(z);
            const float xhat = (yy - mean[c]) * invstd[c];
            o.e[j] = T::from_f32(gamma[c] * invstd[c] * (dz - sums[c] * inv_m - xhat * sums[C + c] * inv_m));
        }
        o.store(dy + pix * lddy + c0);
    }
}

}  // namespace

#define SY_DISPATCH_DTYPE(dtype, CALL)                 \
    switch (dtype) {                                   \
        case SY_DT_BF16: { typedef BF16 T; CALL; break; } \
        case SY_DT_F16: { typedef F16 T; CALL; break; }   \
        case SY_DT_F32: { typedef F32 T; CALL; break; }   \
        default: return SY_ERR_ARG;                    \
    }                                                  \
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH

static inline int epc_of(int dtype) { return dtype == SY_DT_F32 ? 4 : 8; }

extern "C" int sy_focus_pack(const float* in, int N, int Ctot, int c0, int H, int W, void* out, int dtype, void* stream) {
    if (in == nullptr || out == nullptr || N <= 0 || (H & 1) || (W & 1) || c0 < 0 || c0 + 3 > Ctot) return SY_ERR_ARG;
    const long long work = (long long)N * (H / 2) * (W / 2);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((focus_pack_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream, in, N, Ctot, c0, H, W,
                                       (typename T::elem*)out));
}

extern "C" int sy_resize_nearest(const void* in, int N, int Hi, int Wi, int C, int ldi, int64_t ibs, void* out, int Ho,
                                 int Wo, int ldo, int64_t obs, int dtype, void* stream) {
    if (in == nullptr || out == nullptr || N <= 0 || C <= 0) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ldi % e || ldo % e) return SY_ERR_UNSUPPORTED;
    const long long work = (long long)N * Ho * Wo * (C / e);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((resize_nearest_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)in, N, Hi, Wi, C, ldi, (long long)ibs,
                                       (typename T::elem*)out, Ho, Wo, ldo, (long long)obs));
}

extern "C" int sy_resize_nearest_bwd(const void* dout, int N, int Ho, int Wo, int C, int lddo, int64_t dobs, void* din,
                                     int Hi, int Wi, int lddi, int64_t dibs, int accumulate, int dtype, void* stream) {
    if (dout == nullptr || din == nullptr || N <= 0 || C <= 0) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || lddo % e || lddi % e) return SY_ERR_UNSUPPORTED;
    const long long work = (long long)N * Hi * Wi * (C / e);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((resize_nearest_bwd_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)dout, N, Ho, Wo, C, lddo, (long long)dobs,
                                       (typename T::elem*)din, Hi, Wi, lddi, (long long)dibs, accumulate));
}

extern "C" int sy_spp_pool(void* buf, int N, int H, int W, int C, int ld, int64_t bs, int dtype, void* stream) {
    if (buf == nullptr || N <= 0 || C <= 0 || ld < 4 * C) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ld % e) return SY_ERR_UNSUPPORTED;
    const long long work = (long long)N * H * W * (C / e);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((spp_pool_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       (typename T::elem*)buf, N, H, W, C, ld, (long long)bs));
}

extern "C" int sy_spp_pool_bwd(const void* buf, void* dbuf, int N, int H, int W, int C, int ld, int64_t bs, int dtype,
                               void* stream) {
    if (buf == nullptr || dbuf == nullptr || N <= 0 || C <= 0 || ld < 4 * C) return SY_ERR_ARG;
    const long long work = (long long)N * H * W * C;
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((spp_pool_bwd_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)buf, (typename T::elem*)dbuf, N, H, W, C, ld,
                                       (long long)bs));
}

extern "C" int sy_view_copy(const void* in, int ldi, void* out, int ldo, int64_t pixels, int C, int dtype,
                            int accumulate, void* stream) {
    if (in == nullptr || out == nullptr || pixels <= 0 || C <= 0) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ldi % e || ldo % e) return SY_ERR_UNSUPPORTED;
    const long long work = (long long)pixels * (C / e);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((view_copy_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)in, ldi, (typename T::elem*)out, ldo,
                                       (long long)pixels, C, accumulate));
}

extern "C" int sy_bn_finalize(const float* sum, const float* sqsum, int C, double count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              float* scale, float* shift, float* mean, float* invstd, void* stream) {
    if (sum == nullptr || sqsum == nullptr || gamma == nullptr || beta == nullptr || scale == nullptr ||
        shift == nullptr || C <= 0 || count <= 0.0)
        return SY_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return SY_ERR_ARG;
    SY_LAUNCH(bn_finalize_kernel, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, sum, sqsum, C, count, gamma,
              beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

extern "C" int sy_bn_silu_apply(const void* y, int ldy, const float* scale, const float* shift, const void* res, int ldr,
                                void* out, int ldo, int64_t pixels, int C, int dtype, void* stream) {
    if (y == nullptr || out == nullptr || scale == nullptr || shift == nullptr || pixels <= 0 || C <= 0) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ldy % e || ldo % e || (res != nullptr && ldr % e)) return SY_ERR_UNSUPPORTED;
    const long long work = (long long)pixels * (C / e);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((bn_silu_apply_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)y, ldy, scale, shift, (const typename T::elem*)res, ldr,
                                       (typename T::elem*)out, ldo, (long long)pixels, C));
}

extern "C" int sy_bn_silu_bwd_reduce(const void* y, int ldy, const void* da, int ldda, const float* scale,
                                     const float* shift, const float* mean, const float* invstd, float* sums,
                                     int64_t pixels, int C, int dtype, void* stream) {
    if (y == nullptr || da == nullptr || sums == nullptr || pixels <= 0 || C <= 0) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ldy % e || ldda % e) return SY_ERR_UNSUPPORTED;
    if (C / e > kBlock) return SY_ERR_UNSUPPORTED;
    const int rows = kBlock / (C / e);
    long long blocks = (pixels + rows - 1) / rows;
    if (blocks > 1024) blocks = 1024;
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((bn_silu_bwd_reduce_kernel<T>), dim3((int)blocks), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)y, ldy, (const typename T::elem*)da, ldda, scale, shift,
                                       mean, invstd, sums, (long long)pixels, C));
}

extern "C" int sy_bn_silu_bwd_apply(const void* y, int ldy, const void* da, int ldda, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, const float* gamma,
                                    const float* sums, void* dy, int lddy, int64_t pixels, int C, float* dgamma,
                                    float* dbeta, int dtype, void* stream) {
    if (y == nullptr || da == nullptr || sums == nullptr || dy == nullptr || pixels <= 0 || C <= 0) return SY_ERR_ARG;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return SY_ERR_ARG;
    const int e = epc_of(dtype);
    if (C % e || ldy % e || ldda % e || lddy % e) return SY_ERR_UNSUPPORTED;
    const long long work = (long long)pixels * (C / e);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((bn_silu_bwd_apply_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       (const typename T::elem*)y, ldy, (const typename T::elem*)da, ldda, scale, shift,
                                       mean, invstd, gamma, sums, (typename T::elem*)dy, lddy, (long long)pixels, C, dgamma,
                                       dbeta));
}
