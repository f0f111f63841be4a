// conv_igemm.hip — C-ABI entry point of the implicit-GEMM convolution (kernel: conv_igemm_impl.h; one
// translation unit per storage dtype so the instantiations compile in parallel).
#include "conv_igemm_impl.h"

using sy_conv::ConvArgs;
int sy_conv_launch_bf16(const ConvArgs& a, void* stream);
int sy_conv_launch_f16(const ConvArgs& a, void* stream);
int sy_conv_launch_f32(const ConvArgs& a, void* stream);

constexpr long long kWtMinBytes = 8ll << 20;   // outputs of 8 MB or more are written through (measured: stages W, X of profiles/r06)

extern "C" int sy_conv2d(const sy_conv_desc* d, void* stream) {
    if (d == nullptr || d->x == nullptr || d->w == nullptr || d->y == nullptr) return SY_ERR_ARG;
    if (d->N <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0) return SY_ERR_ARG;
    if (d->dtype < SY_DT_BF16 || d->dtype > SY_DT_F32) return SY_ERR_ARG;
    const int epc = d->dtype == SY_DT_F32 ? 4 : 8;
    // 16-byte chunks must never straddle a tap or a pixel
    if (d->Cin % epc != 0 || d->ldx % epc != 0) return SY_ERR_UNSUPPORTED;
    if (d->stride != 1 && d->stride != 2) return SY_ERR_UNSUPPORTED;
    if (d->KH < 1 || d->KW < 1 || d->KH * d->KW > 49) return SY_ERR_UNSUPPORTED;
    if ((d->stat_sum == nullptr) != (d->stat_sqsum == nullptr)) return SY_ERR_ARG;
    if (d->epilogue < SY_EPI_LINEAR || d->epilogue > SY_EPI_BNR) return SY_ERR_ARG;
    if (d->epilogue == SY_EPI_BNR) {    // fused BatchNorm-backward reduce: data gradient, first write, z in `res` with y's strides
        if (d->mode != SY_CONV_DGRAD || d->accumulate || d->y_f32 || d->dtype == SY_DT_F32 || d->res == nullptr || d->scale == nullptr ||
            d->shift == nullptr || d->stat_sum == nullptr || d->ldr != d->ldy || d->rbs != d->ybs || d->stat_segments > 1)
            return SY_ERR_UNSUPPORTED;
        const int t = d->tile & 0xff;       // the staged epilogue of the 128-channel 3x3 stride-1 window tiles carries it
        if (t != 98 && t != 100 && t != 101 && t != 109 && t != 104 && t != 107 && t != 117 && t != 118) return SY_ERR_UNSUPPORTED;
        if (d->KH != 3 || d->stride != 1 || (d->ldy & 7) != 0 || (d->Cout & 7) != 0) return SY_ERR_UNSUPPORTED;
        // the reduce exists ONLY in the staged write-out of conv_epilogue (16-byte stores of y, uint4 loads of z): its run-time
        // preconditions are checked here, because the general epilogue loop has no BNR case and would treat scale / shift / res as
        // an affine + residual of the gradient with no sums accumulated — a silently wrong gradient through a public entry point
        // (ADVICE r05): 16-byte aligned y and z, batch strides that keep every pixel row aligned
        if ((((uintptr_t)d->y | (uintptr_t)d->res) & 15) != 0 || (d->ybs & 7) != 0) return SY_ERR_UNSUPPORTED;
    }
    if ((long long)d->N * d->Ho * d->Wo > 0x7fffffffLL) return SY_ERR_UNSUPPORTED;
    ConvArgs a;
    a.x = (const unsigned char*)d->x; a.w = (const unsigned char*)d->w;
    a.scale = d->scale; a.shift = d->shift; a.res = (const unsigned char*)d->res; a.y = (unsigned char*)d->y;
    a.stat_sum = d->stat_sum; a.stat_sq = d->stat_sqsum;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad;
    a.ldx = d->ldx; a.ldy = d->ldy; a.ldr = d->ldr; a.xbs = d->xbs; a.ybs = d->ybs; a.rbs = d->rbs;
    a.y_f32 = d->y_f32; a.mode = d->mode; a.epilogue = d->epilogue; a.accumulate = d->accumulate;
    a.dec_stride = d->dec_stride;
    a.stat_copies = d->stat_copies > 0 ? d->stat_copies : 1;
    a.seg_M = 0;
    a.s2_classes = 0;
    if (d->stat_segments > 1) {
        if (d->stat_sum == nullptr || d->N % d->stat_segments != 0 || d->mode != SY_CONV_FWD) return SY_ERR_ARG;
        a.seg_M = (d->N / d->stat_segments) * d->Ho * d->Wo;
    }
    a.x_extent = (d->x_bytes > 0 && d->x_bytes < 0xFFFFFFF0LL) ? (unsigned)d->x_bytes : 0u;
    a.w_extent = (d->w_bytes > 0 && d->w_bytes < 0xFFFFFFF0LL) ? (unsigned)d->w_bytes : 0u;
    a.wfrag = (const unsigned char*)d->wfrag;
    a.wfrag_extent = (d->wfrag != nullptr && d->wfrag_bytes > 0 && d->wfrag_bytes < 0xFFFFFFF0LL) ? (unsigned)d->wfrag_bytes : 0u;
    a.pre_w = (const unsigned char*)d->pre_w; a.pre_scale = d->pre_scale; a.pre_shift = d->pre_shift; a.pre_cin = d->pre_cin;
    a.pre_w_extent = (d->pre_w != nullptr && d->pre_w_bytes > 0 && d->pre_w_bytes < 0xFFFFFFF0LL) ? (unsigned)d->pre_w_bytes : 0u;
    if ((d->pre_w != nullptr) != ((d->tile & 0xff) == 119)) return SY_ERR_ARG;       // the fused Bottleneck kernel, and only it, takes pre_*
    a.ksplit = 0;
    if (d->k_splits > 1) {
        if (((d->tile & 0xff) != 117 && (d->tile & 0xff) != 118) || d->mode != SY_CONV_FWD || !d->y_f32 || d->epilogue != SY_EPI_LINEAR ||
            d->scale != nullptr || d->shift != nullptr || d->res != nullptr || d->stat_sum != nullptr || d->accumulate || d->k_splits > 16)
            return SY_ERR_UNSUPPORTED;
        a.ksplit = d->k_splits;
    }
    {   // Write-through output stores (sy_device.h, sy_store16_wt) for outputs the L2s would not keep anyway: the kernel boundary behind
        // a launch that leaves tens of MB of dirty lines costs 2.6-3.2 us instead of 1.2 (profiles/r06 stages N, R-X: l step -0.17 ms,
        // s inference at 8 pairs -3 %).  Small outputs (the batch-1 streaming frame: 1.55 vs 1.52 ms when everything is written
        // through) stay in the L2 for their consumer.  SY_WT_MIN_BYTES overrides the threshold (A/B).
        static const long long wt_min = [] { const char* e = std::getenv("SY_WT_MIN_BYTES"); return e ? std::atoll(e) : (long long)kWtMinBytes; }();
        const long long out_bytes = (long long)d->N * d->Ho * d->Wo * d->Cout * (d->y_f32 ? 4 : (d->dtype == SY_DT_F32 ? 4 : 2));
        a.wt = out_bytes >= wt_min ? 1 : 0;
    }
    a.tile = d->tile & 0xff;
    a.ablate = (d->tile >> 8) & 1023;    // profiling / tests: 1 no pixel loads, 2 no weight loads, 4 no stride-2 parity classes, 8 / 16 statistics, 32.. see ConvArgs
    a.HoWo = d->Ho * d->Wo; a.M = d->N * a.HoWo; a.K = d->KH * d->KW * d->Cin;
    switch (d->dtype) {
        case SY_DT_BF16: return sy_conv_launch_bf16(a, stream);
        case SY_DT_F16: return sy_conv_launch_f16(a, stream);
        default: return sy_conv_launch_f32(a, stream);
    }
}
