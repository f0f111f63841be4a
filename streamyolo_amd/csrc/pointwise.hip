// pointwise.hip — the HBM-bound glue kernels of the path: every one moves 16 bytes per lane,
// coalesced along the NHWC channel axis, grid-strided, and writes straight into the channel slice
// its consumer reads (no torch.cat / split / interpolate kernels survive — SURVEY.md traps T3-T6).
#include "sy_device.h"
#include "../../include/streamyolo_hip.h"

#include "sy_pointwise.h"

namespace {

// ---- Focus: NCHW fp32 planes -> NHWC 16-channel (12 used) ------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void focus_pack_kernel(const float* in, int N, int Ctot, int c0, int H, int W,
                                                            typename T::elem* out) {
    SY_TL_BEGIN(15);
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
    SY_TL_END();
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
    SY_TL_BEGIN(15);
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
    SY_TL_END();
}

// backward: every source pixel sums the destination pixels that read it (gather form, no atomics)
template <typename T>
__global__ __launch_bounds__(kBlock) void resize_nearest_bwd_kernel(const typename T::elem* dout, int N, int Ho, int Wo,
                                                                    int C, int lddo, long long dobs,
                                                                    typename T::elem* din, int Hi, int Wi, int lddi,
                                                                    long long dibs, int accumulate) {
    SY_TL_BEGIN(15);
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
    SY_TL_END();
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

// ---- zero fill of a (strided) byte region: the plan's clears as launches of its own tape (no ATen fill kernels in a step) ----
// rows x row_bytes bytes at pitch_bytes; everything a multiple of 4 bytes (16-byte stores where pointer, row and pitch allow).
template <typename V>
__global__ __launch_bounds__(kBlock) void zero_rows_kernel(V* p, long long rows, long long row_v, long long pitch_v) {
    const long long total = rows * row_v;
    V z;
    __builtin_memset(&z, 0, sizeof(V));
    if (row_v == pitch_v) {
        for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
            if constexpr (sizeof(V) == 16 && SY_WT_ZERO) sy_store16_wt(p + i, *reinterpret_cast<const uint4*>(&z)); else p[i] = z;
        }
    } else {
        for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
            const long long r = i / row_v;
            p[r * pitch_v + (i - r * row_v)] = z;
        }
    }
}

}  // namespace

extern "C" int sy_zero_rows(void* ptr, int64_t rows, int64_t row_bytes, int64_t pitch_bytes, void* stream) {
    if (ptr == nullptr || rows < 0 || row_bytes < 0 || (rows > 1 && pitch_bytes < row_bytes)) return SY_ERR_ARG;
    if (rows == 0 || row_bytes == 0) return SY_OK;
    if (rows == 1) pitch_bytes = row_bytes;
    const unsigned long long a = (unsigned long long)ptr | (unsigned long long)row_bytes | (unsigned long long)pitch_bytes;
    if (a & 3) return SY_ERR_UNSUPPORTED;
    if ((a & 15) == 0) {
        SY_LAUNCH((zero_rows_kernel<uint4>), dim3(grid_for(rows * (row_bytes / 16))), dim3(kBlock), 0, stream, (uint4*)ptr,
                  (long long)rows, (long long)(row_bytes / 16), (long long)(pitch_bytes / 16));
    } else {
        SY_LAUNCH((zero_rows_kernel<unsigned>), dim3(grid_for(rows * (row_bytes / 4))), dim3(kBlock), 0, stream, (unsigned*)ptr,
                  (long long)rows, (long long)(row_bytes / 4), (long long)(pitch_bytes / 4));
    }
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}

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


// ---- second half of a split-K convolution (sy_conv_desc::k_splits) ---------------------------------------------------------
// One thread = one 16-byte output chunk: sums its channels' fp32 partials over the splits in split order, then applies the
// convolution epilogue's arithmetic (acc * scale + shift -> SiLU -> + residual, rounded once).
namespace {
template <typename T>
__global__ __launch_bounds__(kBlock) void splitk_epilogue_kernel(const float* part, int splits, long long pixels, int C,
                                                                 const float* scale, const float* shift,
                                                                 const typename T::elem* res, int ldr, typename T::elem* y,
                                                                 int ldy, int silu) {
    constexpr int E = T::kEPC;
    const int cpp = C / E;
    const long long total = pixels * cpp;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const long long p = i / cpp;
        const int c0 = (int)(i - p * cpp) * E;
        float v[E];
#pragma unroll
        for (int j = 0; j < E; ++j) v[j] = 0.0f;
        for (int z = 0; z < splits; ++z) {
            const float4* src = reinterpret_cast<const float4*>(part + ((long long)z * pixels + p) * C + c0);
#pragma unroll
            for (int q = 0; q < E / 4; ++q) {
                const float4 u = src[q];
                v[4 * q + 0] += u.x; v[4 * q + 1] += u.y; v[4 * q + 2] += u.z; v[4 * q + 3] += u.w;
            }
        }
        Chunk<T> r, o;
        if (res != nullptr) r = Chunk<T>::load(res + p * ldr + c0);
#pragma unroll
        for (int j = 0; j < E; ++j) {
            float t = v[j] * (scale != nullptr ? scale[c0 + j] : 1.0f) + (shift != nullptr ? shift[c0 + j] : 0.0f);
            if (silu) t = sy_silu(t);
            if (res != nullptr) t += T::to_f32(r.e[j]);
            o.e[j] = T::from_f32(t);
        }
        o.store(y + p * ldy + c0);
    }
}
}  // namespace

extern "C" int sy_splitk_epilogue(const float* part, int splits, int64_t pixels, int C, const float* scale, const float* shift,
                                  const void* res, int ldr, void* y, int ldy, int dtype, int epilogue, void* stream) {
    if (part == nullptr || y == nullptr || splits < 1 || pixels <= 0 || C <= 0) return SY_ERR_ARG;
    if (epilogue != SY_EPI_LINEAR && epilogue != SY_EPI_SILU) return SY_ERR_UNSUPPORTED;
    const int e = epc_of(dtype);
    if (C % e || ldy % e || (res != nullptr && ldr % e)) return SY_ERR_UNSUPPORTED;
    const long long work = pixels * (C / e);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((splitk_epilogue_kernel<T>), dim3(grid_for(work)), dim3(kBlock), 0, stream, part, splits,
                                       (long long)pixels, C, scale, shift, (const typename T::elem*)res, ldr,
                                       (typename T::elem*)y, ldy, epilogue == SY_EPI_SILU ? 1 : 0));
}
SY_PROBE_READER(sy_probe_read_pointwise)
