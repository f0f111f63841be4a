// input_pipeline.hip — the frame-pair input pipeline on the device.
//
// The reference prepares every sample on CPU workers — `_mirror` + `preproc` (letterbox on a 114 canvas,
// HWC uint8 -> CHW float32; exps/data/data_augment_flip.py:140-167), `np.concatenate((img, support_img))`
// (exps/data/tal_flip_mosaicdetection.py:257) — ships fp32 [B,6,H,W] over PCIe (13.8 MB per 600x960 pair,
// exps/train_utils/double_data_prefetcher.py:50-51) and, for multi-scale training, resizes the batch with
// `F.interpolate(mode="bilinear", align_corners=False)` (Exp.preprocess, cfgs/l_s50_onex_dfp_tal_filp.py:161-172).
// The streaming loop does the same per camera frame inside its timed region (sAP/streamyolo/streamyolo_det.py:176-179).
//
// Here the uint8 HWC frames go to the device as they are (1/4 of the bytes) and ONE kernel does mirror, letterbox,
// the optional exact 2x decimation, the optional bilinear resize to the training size and the layout change —
// either to the reference's fp32 NCHW tensor (drop-in) or straight to the Focus-packed NHWC16 operand of the stem
// convolution (space-to-depth order TL, BL, TR, BR — trap T4).  HBM-bound: 3 B read, <= 4 B written per pixel/channel.
//
// cv2.resize parity (OpenCV is not available where this was built: all three cases are restated from OpenCV's published
// resize.cpp, PARITY UNPINNED against a cv2 binary — oracle/input_oracle.py states the same arithmetic in numpy):
//   decimate 1   same size: copy;
//   decimate 2   exact 2x: cv::resize turns INTER_LINEAR into the INTER_AREA fast path when both scale factors are exactly
//                2, (a + b + c + d + 2) >> 2 — Argoverse's 1200x1920 -> 600x960;
//   decimate 0   any other camera size: `preproc`'s r = min(H / h, W / w), cv2.resize(img, (int(w r), int(h r)),
//                INTER_LINEAR) (data_augment_flip.py:151-167) = OpenCV's fixed-point bilinear for 8-bit images: source
//                coordinate fx = (float)((dx + 0.5) * scale - 0.5), taps sx = floor(fx), sx + 1 (horizontally: clamped at the
//                borders with the fraction zeroed; vertically: row indices clamped, fraction kept), coefficients saturate_cast<short>(frac * 2048) (round half to even), horizontal pass
//                in int32, vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
#include "sy_pointwise.h"

namespace {

struct FrameSrc {               // uint8 HWC frames of one batch: current and (optional) support
    const unsigned char* img[2];
    long long image_stride;     // bytes between images
    int row_stride;             // bytes between rows
    int hs, ws;                 // size AFTER the load-time resize (decimation or general ratio)
    int dec;                    // 1 or 2; 0 = general-ratio fixed-point bilinear from a src_h x src_w frame
    int src_h, src_w;
    double scale_y, scale_x;    // dec == 0: source / destination size ratios (OpenCV: 1. / ((double)dst / src))
    const unsigned char* mirror;    // [B] flags or nullptr
    // OpenCV's INTER_LINEAR tap set of one destination coordinate (resize.cpp, 8-bit fixed-point path)
    struct Tap { int s0, s1; int a0, a1; };
    // HORIZ: OpenCV zeroes the fraction at the left / right border (xofs / alpha tables); the vertical pass keeps the fraction
    // and only clamps the two row indices (srows[k] = clip(sy + k)) — both taps then read the same row with b0 + b1 = 2048
    template <bool HORIZ> static __device__ __forceinline__ Tap tap(int d, double scale, int ssize) {
        float f = (float)(((double)d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (HORIZ) {
            if (s < 0) { f = 0.0f; s = 0; }
            if (s >= ssize - 1) { f = 0.0f; s = ssize - 1; }
        }
        Tap t;
        t.s0 = s < 0 ? 0 : (s > ssize - 1 ? ssize - 1 : s);
        t.s1 = s + 1 < 0 ? 0 : (s + 1 > ssize - 1 ? ssize - 1 : s + 1);
        t.a0 = (int)rintf((1.0f - f) * 2048.0f);            // saturate_cast<short>(float) = cvRound (round half to even)
        t.a1 = (int)rintf(f * 2048.0f);
        return t;
    }
    // value of channel c at (y, x) of the letterboxed H x W canvas of image n, frame f
    __device__ __forceinline__ float at(int n, int f, int c, int y, int x) const {
        if (y >= hs || x >= ws) return 114.0f;
        if (mirror != nullptr && mirror[n]) x = ws - 1 - x;
        const unsigned char* p = img[f] + n * image_stride;
        if (dec == 1) return (float)p[(long long)y * row_stride + x * 3 + c];
        if (dec == 0) {
            const Tap ty = tap<false>(y, scale_y, src_h), tx = tap<true>(x, scale_x, src_w);
            const unsigned char* r0 = p + (long long)ty.s0 * row_stride + c;
            const unsigned char* r1 = p + (long long)ty.s1 * row_stride + c;
            const int S0 = (int)r0[tx.s0 * 3] * tx.a0 + (int)r0[tx.s1 * 3] * tx.a1;      // horizontal pass, scaled by 2^11
            const int S1 = (int)r1[tx.s0 * 3] * tx.a0 + (int)r1[tx.s1 * 3] * tx.a1;
            const int v = (((ty.a0 * (S0 >> 4)) >> 16) + ((ty.a1 * (S1 >> 4)) >> 16) + 2) >> 2;
            return (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
        const unsigned char* r0 = p + (long long)(2 * y) * row_stride + (2 * x) * 3 + c;
        const unsigned char* r1 = r0 + row_stride;
        return (float)(((int)r0[0] + (int)r0[3] + (int)r1[0] + (int)r1[3] + 2) >> 2);
    }
};

struct PlaneSrc {               // fp32 NCHW tensor (Exp.preprocess's input)
    const float* in;
    int C, H, W;
    __device__ __forceinline__ float at(int n, int f, int c, int y, int x) const {
        return in[(((long long)n * C + f * 3 + c) * H + y) * W + x];
    }
};

// torch's upsample_bilinear2d, align_corners=False: scale = in / out (float), src = max(0, scale * (dst + 0.5) - 0.5),
// i0 = (int)src, i1 = i0 + (i0 < in - 1), lambda1 = src - i0 (aten/src/ATen/native/UpSample.h area_pixel_compute_*)
struct Lerp {
    int i0, i1;
    float l0, l1;
    __device__ __forceinline__ Lerp(int dst, float scale, int in_size) {
        float s = scale * ((float)dst + 0.5f) - 0.5f;
        if (s < 0.0f) s = 0.0f;
        i0 = (int)s;
        if (i0 > in_size - 1) i0 = in_size - 1;
        i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
        l1 = s - (float)i0;
        l0 = 1.0f - l1;
    }
};

template <typename S>
__device__ __forceinline__ float sample(const S& src, int n, int f, int c, int y, int x, bool resize, float sy, float sx,
                                        int H, int W) {
    if (!resize) return src.at(n, f, c, y, x);
    const Lerp ly(y, sy, H), lx(x, sx, W);
    return ly.l0 * (lx.l0 * src.at(n, f, c, ly.i0, lx.i0) + lx.l1 * src.at(n, f, c, ly.i0, lx.i1)) +
           ly.l1 * (lx.l0 * src.at(n, f, c, ly.i1, lx.i0) + lx.l1 * src.at(n, f, c, ly.i1, lx.i1));
}

// -> fp32 NCHW [B, 3*F, Ho, Wo]; one thread per output element, x fastest (coalesced stores)
template <typename S>
__global__ __launch_bounds__(kBlock) void frames_to_nchw_kernel(S src, int B, int F, int H, int W, int Ho, int Wo,
                                                                float* out) {
    const bool resize = Ho != H || Wo != W;
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    const long long total = (long long)B * F * 3 * Ho * Wo;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
        const int x = (int)(i % Wo);
        const int y = (int)((i / Wo) % Ho);
        const int ch = (int)((i / ((long long)Wo * Ho)) % (3 * F));
        const int n = (int)(i / ((long long)Wo * Ho * 3 * F));
        out[i] = sample(src, n, ch / 3, ch % 3, y, x, resize, sy, sx, H, W);
    }
}

// -> Focus-packed NHWC16 [B, Ho/2, Wo/2, 16] per frame (12 real channels, q*3 + c with q = TL, BL, TR, BR; 4 zero)
template <typename T, typename S>
__global__ __launch_bounds__(kBlock) void frames_to_focus_kernel(S src, int B, int F, int H, int W, int Ho, int Wo,
                                                                 typename T::elem* out0, typename T::elem* out1) {
    typedef typename T::elem elem;
    const bool resize = Ho != H || Wo != W;
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
    const int H2 = Ho >> 1, W2 = Wo >> 1;
    const long long per_frame = (long long)B * H2 * W2;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < per_frame * F; i += (long long)gridDim.x * kBlock) {
        const int f = (int)(i / per_frame);
        const long long r = i - f * per_frame;
        const int w2 = (int)(r % W2);
        const int h2 = (int)((r / W2) % H2);
        const int n = (int)(r / ((long long)W2 * H2));
        elem o[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int y = 2 * h2 + (q & 1), x = 2 * w2 + (q >> 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) o[q * 3 + c] = T::from_f32(sample(src, n, f, c, y, x, resize, sy, sx, H, W));
        }
#pragma unroll
        for (int c = 12; c < 16; ++c) o[c] = T::from_f32(0.0f);
        uint4* dst = reinterpret_cast<uint4*>((f == 0 ? out0 : out1) + r * 16);
        constexpr int NV = (int)(16 * sizeof(elem) / 16);
        uint4 v[NV];
        __builtin_memcpy(v, o, sizeof(o));
#pragma unroll
        for (int k = 0; k < NV; ++k) dst[k] = v[k];
    }
}

}  // namespace

extern "C" int sy_frames_u8_pack(const uint8_t* cur, const uint8_t* sup, int B, int Hs, int Ws, int64_t image_stride,
                                 int row_stride, int decimate, const uint8_t* mirror, int H, int W, int Ho, int Wo,
                                 int layout, void* out_cur, void* out_sup, int dtype, void* stream) {
    if (cur == nullptr || out_cur == nullptr || B <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0)
        return SY_ERR_ARG;
    if (decimate < -1 || decimate > 2) return SY_ERR_UNSUPPORTED;
    if (decimate == 2 && ((Hs & 1) || (Ws & 1))) return SY_ERR_UNSUPPORTED;
    if (row_stride < Ws * 3 || image_stride < (int64_t)row_stride * Hs) return SY_ERR_ARG;
    int hs, ws;
    FrameSrc src;
    src.src_h = Hs; src.src_w = Ws; src.scale_y = 1.0; src.scale_x = 1.0;
    if (decimate == 0 || decimate == -1) {
        if (decimate == 0) {
            // preproc: r = min(H / h, W / w); resized to (int(w * r), int(h * r)) — the same double arithmetic as the Python code
            const double r = ((double)H / Hs < (double)W / Ws) ? (double)H / Hs : (double)W / Ws;
            hs = (int)(Hs * r); ws = (int)(Ws * r);
        } else {
            // streaming: cv2.resize(img, (w_img, h_img)) with the canvas = (int(h * in_scale), int(w * in_scale)) — each axis
            // stretched on its own, no letterbox (sAP/streamyolo/streamyolo_det.py:176-178)
            hs = H; ws = W;
            decimate = 0;
        }
        if (hs < 1 || ws < 1 || hs > H || ws > W) return SY_ERR_UNSUPPORTED;
        src.scale_y = 1.0 / ((double)hs / Hs); src.scale_x = 1.0 / ((double)ws / Ws);
        if (hs == Hs && ws == Ws) decimate = 1;                          // r == 1: cv::resize copies
        else if (2 * hs == Hs && 2 * ws == Ws) decimate = 2;             // both factors exactly 2: OpenCV's INTER_AREA fast path
    } else {
        hs = Hs / decimate; ws = Ws / decimate;
        // preproc's ratio r = min(H / h, W / w) must be exactly 1 after the decimation (see the header)
        if (hs > H || ws > W || (hs != H && ws != W)) return SY_ERR_UNSUPPORTED;
    }
    const int F = sup != nullptr ? 2 : 1;
    src.img[0] = cur; src.img[1] = sup; src.image_stride = image_stride; src.row_stride = row_stride;
    src.hs = hs; src.ws = ws; src.dec = decimate; src.mirror = mirror;
    if (layout == SY_FRAMES_NCHW) {
        const long long work = (long long)B * F * 3 * Ho * Wo;
        SY_LAUNCH((frames_to_nchw_kernel<FrameSrc>), dim3(grid_for(work)), dim3(kBlock), 0, stream, src, B, F, H, W, Ho, Wo,
                  (float*)out_cur);
        return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
    }
    if (layout != SY_FRAMES_FOCUS) return SY_ERR_ARG;
    if ((Ho & 1) || (Wo & 1) || (F == 2 && out_sup == nullptr)) return SY_ERR_ARG;
    const long long work = (long long)B * F * (Ho / 2) * (Wo / 2);
    SY_DISPATCH_DTYPE(dtype, SY_LAUNCH((frames_to_focus_kernel<T, FrameSrc>), dim3(grid_for(work)), dim3(kBlock), 0, stream,
                                       src, B, F, H, W, Ho, Wo, (typename T::elem*)out_cur, (typename T::elem*)out_sup));
}

extern "C" int sy_resize_bilinear_nchw(const float* in, int N, int C, int H, int W, float* out, int Ho, int Wo,
                                       void* stream) {
    if (in == nullptr || out == nullptr || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return SY_ERR_ARG;
    if (C % 3) return SY_ERR_UNSUPPORTED;                   // frames are 3-channel planes (3 = on_pipe, 6 = a pair)
    PlaneSrc src;
    src.in = in; src.C = C; src.H = H; src.W = W;
    const long long work = (long long)N * C * Ho * Wo;
    SY_LAUNCH((frames_to_nchw_kernel<PlaneSrc>), dim3(grid_for(work)), dim3(kBlock), 0, stream, src, N, C / 3, H, W, Ho, Wo, out);
    return SY_LAUNCH_OK() == 0 ? SY_OK : SY_ERR_LAUNCH;
}
