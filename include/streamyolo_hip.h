/* streamyolo_hip.h — C ABI of libstreamyolo_hip.so (MI355X / gfx950 only).
 *
 * The reference (yancie-yjr/StreamYOLO) has NO native boundary on this path: its hot path is a
 * tree of Python nn.Modules dispatching cuDNN/ATen/torchvision kernels one op at a time
 * (SURVEY.md §0, §8(b)).  The drop-in surface a user sees stays Python (package `streamyolo_amd`
 * mirrors exps/model/ and cfgs/); THIS header is the seam underneath it — plain pointers, sizes
 * and a hipStream_t, no torch types — and each entry point names the reference code it replaces.
 *
 * Conventions
 *   - activations are NHWC "views": (ptr, ld = elements between consecutive pixels, batch stride);
 *     a view may be a channel slice of a wider buffer, which is how every torch.cat on the path
 *     (trap T6) disappears: producers write straight into their slice.
 *   - dtype codes: 0 = bf16, 1 = fp16, 2 = fp32 (storage of activations and packed weights;
 *     accumulation and epilogues are always fp32).
 *   - every function returns 0 on success, >0 on error (1 bad argument, 2 launch failure,
 *     3 unsupported shape); nothing aborts, nothing synchronises, everything is enqueued on `stream`.
 */
#ifndef STREAMYOLO_HIP_H
#define STREAMYOLO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SY_ABI_VERSION 7
#define SY_API __attribute__((visibility("default")))

enum { SY_DT_BF16 = 0, SY_DT_F16 = 1, SY_DT_F32 = 2 };

/* epilogue selectors of sy_conv2d */
enum {
    SY_EPI_LINEAR = 0,      /* y = acc*scale + shift                                   */
    SY_EPI_SILU = 1,        /* y = silu(acc*scale + shift) [+ residual]  (BaseConv)    */
    SY_EPI_SIGMOID = 2,     /* y = sigmoid(acc*scale + shift)            (cls preds)   */
    SY_EPI_DECODE = 3,      /* ch0,1: (v+grid)*stride; ch2,3: exp(v)*stride; ch4: sigmoid (reg+obj preds) */
    SY_EPI_BNR = 4          /* SY_CONV_DGRAD only (16-bit types, first write): y = acc, and the BatchNorm-backward REDUCE of the layer whose
                               activation gradient this launch produces rides in the epilogue — with z = that layer's raw conv output
                               (passed as `res`, same pixel / batch strides as y), its folded affine in `scale` / `shift`:
                               g = acc * silu'(scale*z + shift);  stat_sum[r][0][c] += sum g,  stat_sum[r][1][c] += sum g*z  over the
                               tile's pixels, replica r = tile % stat_copies ([copies][2][Cout] as sy_bn_silu_bwd_reduce's `sums`, but
                               the second row is the RAW moment: sy_bn_silu_bwd_apply takes it with bit 2 of dres_accumulate).
                               Replaces one sy_bn_silu_bwd_reduce launch and its read of dA (autograd of BatchNorm2d + SiLU,
                               exps/model/darknet.py:115-165 BaseConv). */
};

/* workgroup tiles of sy_conv2d (output channels x output pixels) */
enum { SY_TILE_AUTO = 0, SY_TILE_256x256 = 1, SY_TILE_128x256 = 2, SY_TILE_128x128 = 3, SY_TILE_64x256 = 4, SY_TILE_32x256 = 5,
       SY_TILE_128x64 = 6, SY_TILE_64x64 = 7, SY_TILE_256x64 = 8, SY_TILE_256x128 = 9,
       SY_TILE_RS = 16,   /* add to a tile code: register-staged variant instead of the 4-deep LDS-DMA ring */
       SY_TILE_DMA2 = 32, /* add: 2-deep LDS-DMA ring */
       SY_TILE_DMA3 = 48, /* add: 3-deep LDS-DMA ring */
       SY_TILE_WR = 80,   /* add: register-staged pixels + fragment-packed weights loaded straight into VGPRs */
       SY_TILE_HALO = 112, /* 114..118 (HALO + 2..6): 3x3 stride-1 kernel with the input tile + halo resident in LDS (csrc/conv3x3_halo.h;
                             needs wfrag): 128 ch x 2 rows x 32 px (8 waves) | 128 x 2 | 64 x 8 | 128 x 2 and 128 x 4 software-pipelined;
                             112, 113: software-pipelined, one 32 x 32 MFMA tile per wave (small launches): 64 ch x 2 rows | 128 ch x 2 rows;
                             107, 104: software-pipelined, 128 ch x 3 rows | 128 ch x 5 rows (three / five MFMAs per weight fragment);
                             101, 100, 98, 109 (96, 97: on 64 channels): THIRD generation (csrc/conv3x3_halo3.h) of 117 / 107 / 118 / 104: fragment reads with immediate
                             offsets, hand-placed instruction stream, prefetch across the channel-slab boundary (same summation order);
                             105, 106, 111: K GROUPS inside the workgroup (small launches; partial tiles summed through LDS in group order — another
                             fp32 summation order than the tiles above): 105 stride 2 forward 64 ch x 1 row, 2 groups | 106 32 ch, 4 groups |
                             111 64 ch, 2 groups (all 2 rows x 32 px);
                             119: FUSED Bottleneck forward (1x1 -> 3x3, the hidden activation kept in LDS; sy_conv_desc::pre_*);
                             110: STRIDE 2 forward, 128 ch x 2 output rows x 32 px, input window split by column parity;
                             108: STRIDE 2 data gradient (csrc/conv3x3_s2dgrad.h), four output-parity classes */
       SY_TILE_1X1K = 121 /* 121..124: 1x1 stride-1 kernel with the tile's whole K extent requested in one burst (csrc/conv1x1_tile.h; needs
                             wfrag, 16-bit types, Cin 64 / 128 / 256 / 512): 128 ch x 64 px | 64 ch x 128 px | 128 ch x 128 px | 64 ch x 64 px */ };

/* gather modes of sy_conv2d */
enum {
    SY_CONV_FWD = 0,        /* out(ho,wo) <- in(ho*s - p + kh, wo*s - p + kw)                  */
    SY_CONV_DGRAD = 1       /* out(h,w)  <- in((h + p - kh)/s, (w + p - kw)/s) when divisible  */
};

typedef struct sy_conv_desc {
    /* tensors */
    const void* x;          /* input view,  [N, H, W, Cin]   */
    const void* w;          /* packed weights [Cout][KH*KW][Cin] in `dtype`, K-contiguous      */
    const float* scale;     /* [Cout] fp32 (folded BN gamma/sqrt(var+eps), or NULL = 1)        */
    const float* shift;     /* [Cout] fp32 (folded BN shift / conv bias, or NULL = 0)          */
    const void* res;        /* optional residual view added after the activation (or NULL)    */
    void* y;                /* output view, [N, Ho, Wo, Cout]                                  */
    float* stat_sum;        /* optional [Cout]: += sum over pixels of acc   (training BN)      */
    float* stat_sqsum;      /* optional [Cout]: += sum over pixels of acc^2                    */
    /* geometry */
    int32_t N, H, W, Cin;
    int32_t Ho, Wo, Cout;
    int32_t KH, KW, stride, pad;
    int32_t ldx, ldy, ldr;              /* pixel strides, elements                              */
    int64_t xbs, ybs, rbs;              /* batch strides, elements                              */
    /* behaviour */
    int32_t dtype;                      /* SY_DT_* of x, w, res (and y unless y_f32)            */
    int32_t y_f32;                      /* 1: y is fp32 regardless of dtype                      */
    int32_t mode;                       /* SY_CONV_FWD | SY_CONV_DGRAD                           */
    int32_t epilogue;                   /* SY_EPI_*                                              */
    int32_t accumulate;                 /* 1: y += result (gradient fan-in)                      */
    float dec_stride;                   /* SY_EPI_DECODE: the level's stride (8/16/32)           */
    int32_t stat_copies;                /* stat arrays hold this many replicas [copies][Cout] (>=1) */
    int32_t stat_segments;              /* 0/1: one set of statistics; S > 1: the N images form S equal groups with
                                           separate statistics, arrays [S][copies][Cout] (N % S == 0) */
    int32_t tile;                       /* SY_TILE_AUTO or a forced workgroup tile (channels x pixels) */
    int64_t x_bytes, w_bytes;           /* bytes addressable from x / w (buffer bounds of the fast gather; 0 = unknown) */
    const void* wfrag;                  /* optional: weights re-packed in MFMA-fragment order (SY_TILE_WR variants) */
    int64_t wfrag_bytes;
    /* k_splits S > 1 (tiles 117 / 118, forward, y_f32 = 1, SY_EPI_LINEAR, no scale / shift / res / statistics): the channel slabs
       of the contraction are cut into S ranges, one per gridDim.z; split z writes its fp32 PARTIAL sums as images [z*N, (z+1)*N) of
       y (which must hold S*N images, batch stride ybs).  sy_splitk_epilogue sums the partials and applies the epilogue.  For the
       deep small-map layers of the batch-1 streaming step: 36-72 workgroups become 144-288. */
    int32_t k_splits;
    /* Fused Bottleneck forward (tile 119, eval epilogues only; csrc/bottleneck_fused.h): this 3x3 stride-1 launch first computes its
       own input h = silu(pre_scale * (pre_w x) + pre_shift) — the 1x1 BaseConv in front of it — for the tile's halo window and keeps
       it in LDS.  x is then the 1x1 convolution's input ([N, H, W, pre_cin]); Cin stays the 3x3 convolution's input width (the hidden
       channels); pre_w: that 1x1 convolution's weights in MFMA-fragment order ([Cin][pre_cin], as `wfrag` of a 1x1 launch);
       pre_scale / pre_shift: its folded BatchNorm affine, fp32 [Cin].  pre_cin + Cin <= 544, both multiples of 32.
       Replaces yolox Bottleneck.conv1 -> conv2 (+ x) as two launches. */
    int32_t pre_cin;
    const void* pre_w;
    int64_t pre_w_bytes;
    const float* pre_scale;
    const float* pre_shift;
} sy_conv_desc;

/* Implicit-GEMM convolution on the MFMA units with the fused epilogue.
 * Replaces: yolox BaseConv.forward / fuseforward = Conv2d -> BatchNorm2d -> SiLU as three kernels
 * (call sites exps/model/darknet.py:115-165, dfp_pafpn.py:33-105,168-170, tal_head.py:55-131,162-171),
 * the Bottleneck shortcut add, the DFP `+ current` add (dfp_pafpn.py:168-170), the eval-time
 * sigmoid/decode of tal_head.py:197-199,245-260, and (mode DGRAD) cuDNN's backward-data. */
SY_API int sy_conv2d(const sy_conv_desc* d, void* stream);

/* Weight gradient (fp32, +=): dw[co][tap][ci] = sum_pixels dy[p][co] * x[p@tap][ci]; with
 * dw_oihw = 1 the result lands in the nn.Parameter layout [Cout][Cin][KH][KW] (so .grad can be a
 * view of the gradient arena).  Replaces cuDNN backward-filter under autograd (double_trainer.py:114). */
typedef struct sy_wgrad_desc {
    const void* x;  const void* dy;  float* dw;
    int32_t N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
    int32_t ldx, lddy;  int64_t xbs, dybs;
    int32_t dtype;
    int32_t dw_oihw;
    void* workspace;                    /* optional fp32 scratch for split-K slabs (NULL: one split) */
    int64_t workspace_bytes;
    int32_t tile, target_blocks;        /* tuning knobs (0 = heuristic): workgroup tile, workgroups aimed for by split-K.
                                         * tile: 1-6 = (k rows x output channels) per workgroup, operands transposed into LDS by
                                         * scatter stores; +16 / +32 = the same tiles with both operands parked in LDS as they lie in
                                         * HBM (LDS-DMA ring of 3 / 4 slabs) and gathered by ds_read_b64_tr_b16; 3x3 stride-1, 16-bit,
                                         * Cin % 32 == 0, Cout % 16 == 0 only: 52 = all nine taps per workgroup (x halo window in LDS),
                                         * 59 = 52 with the slab loop as one instruction stream, 60 = 59 on eight waves with 64 input
                                         * channels per workgroup (Cin % 64 == 0).  52 / 59 / 60 on a layer they do not cover:
                                         * SY_ERR_UNSUPPORTED; any other unknown code: the heuristic tile. */
    int64_t x_bytes, dy_bytes;          /* bytes addressable from x / dy (buffer bounds; 0 = unknown) */
} sy_wgrad_desc;
SY_API int sy_conv2d_wgrad(const sy_wgrad_desc* d, void* stream);

/* Focus space-to-depth (trap T4: TL,BL,TR,BR) fused with NCHW fp32 -> NHWC `dtype` conversion.
 * in: [N, Ctot, H, W] fp32 planes, takes channels [c0, c0+3); out: [N, H/2, W/2, 16] (12 used, 4 zero).
 * Replaces yolox Focus.forward slicing + cat (exps/model/darknet.py:115) and torch.split (dfp_pafpn.py:120,145). */
SY_API int sy_focus_pack(const float* in, int N, int Ctot, int c0, int H, int W, void* out, int dtype, void* stream);

/* Device-side input pipeline: uint8 HWC frames -> model input in ONE launch.
 * cur / sup: [B, Hs, Ws, 3] uint8 (BGR as cv2.imread delivers them; sup = NULL for a single on_pipe frame),
 * image_stride / row_stride in bytes.  Steps, in the reference's order:
 *   decimate (1 | 2 | 0 | -1): the load-time cv2.resize (exps/dataset/tal_flip_one_future_argoversedataset.py:179-187,
 *       streamyolo_det.py:177): 1 = copy, 2 = exact 2x = (a+b+c+d+2)>>2 (OpenCV's INTER_AREA fast path), 0 = any other
 *       camera size: r = min(H/Hs, W/Ws), OpenCV's fixed-point INTER_LINEAR to (int(Ws r), int(Hs r)) (`preproc`,
 *       exps/data/data_augment_flip.py:151-167); restated from OpenCV's source, unpinned against a cv2 binary;
 *       -1 = the streaming detector's cv2.resize(img, (W, H)): each axis stretched on its own to exactly the H x W canvas
 *       (sAP/streamyolo/streamyolo_det.py:176-178, canvas = (int(Hs * in_scale), int(Ws * in_scale))), same arithmetic;
 *   mirror[b] != 0: `_mirror`'s image[:, ::-1], the same flag for both frames of a pair
 *       (exps/data/data_augment_flip.py:140-148, DoubleTrainTransform :219-222);
 *   letterbox onto an H x W canvas filled with 114, image at the top-left (`preproc` :151-167);
 *   bilinear resize of the canvas to Ho x Wo when they differ (Exp.preprocess's F.interpolate(mode="bilinear",
 *       align_corners=False), cfgs/l_s50_onex_dfp_tal_filp.py:161-172);
 *   layout SY_FRAMES_NCHW: out_cur = fp32 [B, 3 or 6, Ho, Wo] (current channels first — np.concatenate((img,
 *       support_img)), exps/data/tal_flip_mosaicdetection.py:257); out_sup unused;
 *   layout SY_FRAMES_FOCUS: out_cur / out_sup = Focus-packed [B, Ho/2, Wo/2, 16] of `dtype` (what sy_focus_pack
 *       would produce from the NCHW tensor: the stem convolution's operand). */
enum { SY_FRAMES_NCHW = 0, SY_FRAMES_FOCUS = 1 };
SY_API int sy_frames_u8_pack(const uint8_t* cur, const uint8_t* sup, int B, int Hs, int Ws, int64_t image_stride,
                             int row_stride, int decimate, const uint8_t* mirror, int H, int W, int Ho, int Wo,
                             int layout, void* out_cur, void* out_sup, int dtype, void* stream);

/* Exp.preprocess's multi-scale resize on the reference's own tensor: fp32 NCHW [N, C, H, W] -> [N, C, Ho, Wo],
 * torch bilinear, align_corners=False (cfgs/l_s50_onex_dfp_tal_filp.py:161-172).  C = 3 or 6. */
SY_API int sy_resize_bilinear_nchw(const float* in, int N, int C, int H, int W, float* out, int Ho, int Wo, void* stream);

/* Nearest-neighbour resize to a target SIZE (trap T3), writing into a channel slice.
 * Replaces F.interpolate(size=..., mode='nearest') + torch.cat (dfp_pafpn.py:125-126,130-131). */
SY_API int sy_resize_nearest(const void* in, int N, int Hi, int Wi, int C, int ldi, int64_t ibs,
                      void* out, int Ho, int Wo, int ldo, int64_t obs, int dtype, void* stream);

/* SPP pooling: buf holds x in channels [0,C) of a 4C-wide view; writes maxpool 5/9/13 (stride 1,
 * -inf padding) into [C,2C), [2C,3C), [3C,4C) (trap T5).  Replaces SPPBottleneck's three
 * nn.MaxPool2d + cat (exps/model/darknet.py:156).  argmax (optional, training): [N,H,W,3,C] bytes,
 * the window offset of each pooled maximum, consumed by sy_spp_pool_bwd. */
SY_API int sy_spp_pool(void* buf, int N, int H, int W, int C, int ld, int64_t bs, void* argmax, int dtype,
                       void* stream);

/* The head's output transform as a stand-alone in-place pass over out [B, A, 5+nc] fp32 (anchors level-major, row-major):
 * flags & 1: boxes (xy + grid) * stride, exp(wh) * stride — TALHead.decode_outputs (exps/model/tal_head.py:245-260), what
 * tools/eval.py:187-188 calls when the head ran with decode_in_inference = False; flags & 2: sigmoid of the objectness column
 * (tal_head.py:197-199 applies it in both modes); flags & 4 (after the others): boxes (cx, cy, w, h) -> corners (cx - w/2, cy - h/2,
 * cx + w/2, cy + h/2), the in-place rewrite yolox.utils.postprocess performs on its argument before the confidence filter (the
 * reference's evaluator sees it: exps/evaluators/onex_stream_evaluator.py:148-150).  level_h / level_w / level_stride: HOST arrays of nlevels (<= 8) entries
 * (copied into the launch by value), sum h*w == A. */
SY_API int sy_head_decode(float* out, int B, int A, int nch, const int32_t* level_h, const int32_t* level_w,
                          const float* level_stride, int nlevels, int flags, void* stream);

/* Box decode + confidence filter + class-aware greedy NMS for a batch of images.
 * pred: [B, A, 5+nc] fp32 (cx,cy,w,h,obj,cls...) as produced by the head (tal_head.py:245-260).
 * Outputs per image i: out_count[i] kept detections; out_det[i][k][0..6] = x1,y1,x2,y2,obj,
 * class_conf,class_id; out_index[i][k] = anchor index; kept in descending score order.
 * Replaces yolox.utils.postprocess (exps/evaluators/onex_stream_evaluator.py:148-150) and the
 * inline `inference()` of sAP/streamyolo/streamyolo_det.py:62-83 incl. torchvision batched_nms.
 * `workspace` must hold sy_postprocess_workspace_bytes(B, A) bytes. */
SY_API int64_t sy_postprocess_workspace_bytes(int B, int A);
SY_API int sy_postprocess(const float* pred, int B, int A, int num_classes, float conf_thre, float nms_thre,
                   int max_det, float* out_det, int32_t* out_index, int32_t* out_count,
                   void* workspace, void* stream);

/* Optimizer + EMA step of one training iteration in one launch.  Replaces
 * `scaler.step(optimizer)` + `ema_model.update(model)` (exps/train_utils/double_trainer.py:115-119):
 * torch.optim.SGD(momentum, nesterov=True) semantics per entry — d = g*grad_scale; d += weight_decay*p;
 * buf = first_step ? d : momentum*buf + d; d += momentum*buf; p -= lr*lr_mult*d — then yolox ModelEMA:
 * ema = ema_decay*ema + (1-ema_decay)*p on the UPDATED p.  g == NULL: EMA only (BatchNorm running statistics);
 * buf == NULL: no momentum; ema == NULL: no EMA.  grad_scale folds GradScaler's 1/scale and the 1/world_size of the
 * gradient all-reduce.  chunk0 = prefix sum of ceil(n/1024) over the preceding entries; `entries` is DEVICE memory. */
typedef struct sy_optim_entry {
    float* p;
    const float* g;
    float* buf;
    float* ema;
    int64_t n;
    float weight_decay;
    float lr_mult;
    int32_t chunk0;
    int32_t reserved;
} sy_optim_entry;
SY_API int sy_sgd_ema_step(const sy_optim_entry* entries, int n_entries, int total_chunks, float lr, float momentum,
                           float grad_scale, float ema_decay, int first_step, void* stream);

/* Per-step weight staging: the nn.Parameters stay OIHW fp32 (the optimizer's master copy,
 * exps/train_utils/double_trainer.py:114-119); ONE launch re-derives, for every convolution of the step, the
 * layouts the MFMA kernels read.  Entry: a source block w[co_n][ci_n][taps] fp32 written as rows r0..r0+co_n of
 *   packed   [R][taps][CI]       forward / weight-gradient operand (K-contiguous), CI >= ci_n (zero padded once)
 *   packed_t [CI][taps][R_t]     data-gradient operand, element [ci][tap][r0 + co]
 *   frag / frag_t                the same two matrices in SY_TILE_WR fragment order
 *                                [rows/32][channels/BK][taps][2][2][32][EPC]  (BK = 64 bytes of channels)
 * Any destination may be NULL.  Several entries may target one destination (reg + obj predictors share a matrix).
 * Padding rows / channels are never written: allocate destinations zeroed.  `entries` is DEVICE memory. */
typedef struct sy_pack_entry {
    const float* w;
    void* packed;
    void* packed_t;
    void* frag;
    void* frag_t;
    int32_t co_n, ci_n, taps, r0, R, R_t, CI, dtype;
    int32_t tile0;              /* prefix sum over the preceding entries of ceil(co_n/32) * ceil(ci_n/32) */
    int32_t reserved;
} sy_pack_entry;
/* total_tiles = the prefix sum over ALL entries (one workgroup per 32x32 tile); taps <= 9. */
SY_API int sy_pack_weights(const sy_pack_entry* entries, int n_entries, int total_tiles, void* stream);

/* SEGMENTS (nseg >= 1): the two frames of a pair pass through the shared backbone separately in the reference, so each
 * BatchNorm normalises them with separate batch statistics (dfp_pafpn.py:120-165).  Here both frames run in ONE launch
 * per layer: segment s owns rows [s*pixels, (s+1)*pixels) of every activation view (`pixels` = rows PER segment,
 * `count` likewise), its own statistics [s][copies][C] / sums [s][copies][2][C] and its own affine / mean / invstd
 * [s][C]; gamma, beta, dgamma, dbeta are shared.  sy_conv2d's stat_segments splits its pixel range the same way.
 * nseg = 1 is an ordinary single-call BatchNorm.
 *
 * Training-mode BatchNorm helpers around sy_conv2d(stat_sum/stat_sqsum).
 * Replaces nn.BatchNorm2d in training mode (momentum/eps patched by init_yolo, cfgs/<name>.py:40-44). */
SY_API int sy_bn_finalize(const float* sum, const float* sqsum, int C, int copies, double count, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean,
                   float* running_var, float* scale, float* shift, float* mean, float* invstd,
                   int nseg, void* stream);
/* Running-statistics update of MANY BatchNorm modules in one launch, from the same replica arrays sy_bn_finalize
 * folds (pass running_mean = NULL there).  Entry i describes one nn.BatchNorm2d and the 1 or 2 calls it received
 * this step IN CALL ORDER (the backbone / neck modules are called once per frame, current frame first —
 * exps/model/dfp_pafpn.py:120-165): for each call, running = (1 - momentum) * running + momentum * batch_stat
 * with the unbiased variance, exactly torch.nn.functional.batch_norm(training=True).  `entries` is DEVICE memory. */
typedef struct sy_bn_running_entry {
    float* running_mean;
    float* running_var;
    const float* sum[2];        /* [copies][C] replica arrays of call 0 / call 1 (call 1 NULL when calls == 1) */
    const float* sqsum[2];
    double count[2];            /* elements per channel of each call */
    int32_t C, copies, calls;
    float momentum;
    int32_t ld;                 /* elements between consecutive replicas (0 = C): > C when the module's channels are a slice of a
                                   wider statistics array (sibling convolutions stacked into one launch) */
    int32_t reserved;
    int64_t* num_batches_tracked; /* NULL, or the module's counter: += calls (nn.BatchNorm2d increments it once per training-mode
                                   forward; the shared backbone / neck modules are called once per frame) — ABI 7 */
} sy_bn_running_entry;
SY_API int sy_bn_running_update(const sy_bn_running_entry* entries, int n_entries, int max_C, void* stream);
/* sy_bn_finalize + sy_bn_silu_apply in one launch (the training forward of every BaseConv: nn.BatchNorm2d in training mode
   followed by nn.SiLU, yolox BaseConv.forward; one kernel boundary less on the critical path per layer).  Statistics
   [nseg][copies][C] as sy_conv2d's epilogue leaves them, count = elements per channel per segment; scale / shift / mean /
   invstd [nseg][C] are OUTPUTS (the backward pass reads them); running statistics: sy_bn_running_update.  C must have a
   divisor <= 64 that is a multiple of the 16-byte chunk (every width of the reference's models has). */
SY_API int sy_bn_finalize_apply(const float* sum, const float* sqsum, int copies, double count, const float* gamma,
                                const float* beta, float eps, float* scale, float* shift, float* mean, float* invstd,
                                const void* y, int ldy, const void* res, int ldr, void* out, int ldo, int64_t pixels, int C,
                                int dtype, int nseg, void* stream);

/* a = silu(scale*y + shift) [+ res], y raw conv output; views as in sy_conv2d. */
SY_API int sy_bn_silu_apply(const void* y, int ldy, const float* scale, const float* shift, const void* res,
                     int ldr, void* out, int ldo, int64_t pixels, int C, int dtype, int nseg, void* stream);
/* Backward of (BN-train + SiLU): reduce pass then apply pass.
 * reduce: sums[r][0][c] += sum dz, sums[r][1][c] += sum dz*xhat over replica r = workgroup % copies,
 * with dz = da * silu'(scale*y+shift). */
SY_API int sy_bn_silu_bwd_reduce(const void* y, int ldy, const void* da, int ldda, const float* scale,
                                 const float* shift, const float* mean, const float* invstd, float* sums,
                                 int copies, int64_t pixels, int C, int dtype, int nseg, void* stream);
/* apply: dy = gamma*invstd*(dz - S0/M - xhat*S1/M) with S = sums folded over its `copies` replicas
 * ([copies][2][C]); optionally dgamma += S1, dbeta += S0.  dres (optional): the gradient view of the residual input of
 * y = silu(bn(conv)) + res (Bottleneck shortcut, DFP add): dres = da, or dres += da when (dres_accumulate & 1) — the same
 * pass that already reads da (replaces a separate sy_view_copy).  dres_accumulate & 2: dgamma / dbeta are accumulated with
 * atomics even for nseg == 1 (the two frames of a pair go through separate launches on different streams).
 * dres_accumulate & 4: row 1 of `sums` holds the raw moment sum dz*y as a data gradient's fused reduce leaves it (SY_EPI_BNR);
 * converted here: sum dz*xhat = invstd * (sum dz*y - mean * sum dz). */
SY_API int sy_bn_silu_bwd_apply(const void* y, int ldy, const void* da, int ldda, const float* scale,
                                const float* shift, const float* mean, const float* invstd,
                                const float* gamma, const float* sums, int copies, void* dy, int lddy,
                                int64_t pixels, int C, float* dgamma, float* dbeta, void* dres, int lddres,
                                int dres_accumulate, int dtype, int nseg, void* stream);

/* SimOTA assignment + Trend-Aware loss, forward and gradient, for a whole batch, no host sync.
 * raw [B, A, 5+nc] fp32 raw head logits (reg4, obj, cls); labels/support [B, max_labels, 5] fp32 rows
 * (cls, cx, cy, w, h) zero padded; level_h/w/stride describe the anchor grid (level-major, row-major).
 * Outputs: d_raw [B, A, 5+nc] = d total_loss / d raw; losses[8] = total, 5*iou, l1, conf, cls,
 * num_fg/num_gt, num_fg, num_gt (device); optional fg_mask [B, A] int32; optional d_pad [B, A, 16] in `pad_dtype` (SY_DT_*,
 * num_classes <= 8): the same gradient as columns [reg 4 | obj | 5..7 untouched | cls nc | rest untouched] — the operand layout
 * of the prediction convolutions' data / weight gradients, so that no repacking pass stands between the loss and the backward.
 * Replaces TALHead.get_losses / get_assignments / dynamic_k_matching (exps/model/tal_head.py:262-712)
 * and autograd's backward over them. */
SY_API int64_t sy_tal_loss_workspace_bytes(int B, int A, int max_gt);
SY_API int sy_tal_loss(const float* raw, int B, int A, int num_classes, const float* labels,
                       const float* support, int max_labels, const int32_t* level_h,
                       const int32_t* level_w, const float* level_stride, int nlevels, float gamma,
                       float ignore_thr, float ignore_value, int use_l1, float* d_raw, float* losses,
                       int32_t* fg_mask, void* workspace, void* d_pad, int pad_dtype, void* stream);

/* Assignment of the LAST sy_tal_loss call on `workspace` (same B, A, max_labels): matched_gt [B, A] int32 = index of the matched
 * ground truth of every anchor, -1 for background; matched_iou [B, A] fp32 = its IoU (either may be NULL).  The reference's
 * get_assignments returns the same pair for the foreground anchors (exps/model/tal_head.py:559-600); diagnostics and parity tests. */
SY_API int sy_tal_loss_assignment(const void* workspace, int B, int A, int max_labels, int32_t* matched_gt,
                                  float* matched_iou, void* stream);

/* Workgroup caps of the BatchNorm row kernels (grid-stride maps over pixel rows: results do not depend on them, except that the
 * backward reduce's cap is the number of replica rows the exact mode sizes its sums for): caps4 = {sy_bn_silu_apply,
 * sy_bn_finalize_apply, sy_bn_silu_bwd_reduce, sy_bn_silu_bwd_apply}.  set4 (may be NULL): entries > 0 replace the cap, others keep
 * it; get4 (may be NULL) receives the caps in force afterwards.  Process-wide; defaults 2048 / 2048 / 768 / 1024 (or SY_BN_*_BLOCKS
 * in the environment, read at first use).  Takes no stream: nothing is launched (ABI 7; replaces an import-time environment write). */
SY_API int sy_bn_grid_caps(const int32_t* set4, int32_t* get4);

/* Zero fill of rows x row_bytes bytes at pitch_bytes (all multiples of 4; rows == 1: one dense run): the clears of a training
 * step — statistics / gradient arenas, the not-yet-written channel ranges of a gradient buffer before an accumulating data
 * gradient (autograd's zero-initialised .grad accumulation) — as launches of the plan's own tape instead of ATen fills (ABI 7). */
SY_API int sy_zero_rows(void* ptr, int64_t rows, int64_t row_bytes, int64_t pitch_bytes, void* stream);

/* elementwise helpers on views: out (+)= in */
SY_API int sy_view_copy(const void* in, int ldi, void* out, int ldo, int64_t pixels, int C, int dtype,
                 int accumulate, void* stream);

/* backward of sy_resize_nearest (scatter-add into the source grid) */
SY_API int sy_resize_nearest_bwd(const void* dout, int N, int Ho, int Wo, int C, int lddo, int64_t dobs,
                          void* din, int Hi, int Wi, int lddi, int64_t dibs, int accumulate,
                          int dtype, void* stream);
/* backward of sy_spp_pool: dbuf holds grads of the 4 slices; routes pooled grads to their arg-max into slice 0 */
SY_API int sy_spp_pool_bwd(void* dbuf, const void* argmax, int N, int H, int W, int C, int ld, int64_t bs,
                    int dtype, void* stream);

/* Second half of a split-K convolution (sy_conv_desc::k_splits): y[p][c] = act(scale[c] * sum_z part[z][p][c] + shift[c]) (+ res),
 * part = fp32 [splits][pixels][C] dense, act per `epilogue` (SY_EPI_LINEAR | SY_EPI_SILU), y / res views in `dtype` with pixel
 * strides ldy / ldr.  The partials are summed in split order (deterministic). */
SY_API int sy_splitk_epilogue(const float* part, int splits, int64_t pixels, int C, const float* scale, const float* shift,
                              const void* res, int ldr, void* y, int ldy, int dtype, int epilogue, void* stream);

/* ---- small fp32 helpers of the backward plan ---------------------------------------------------------------------
 * sy_rows_add_f32: dst[r][0..cols) += src[r][0..cols) for r < rows (row pitches in elements); zero_src != 0 zeroes the
 * source behind the read (a scratch the next accumulating sy_conv2d_wgrad launch reuses).  Used where a weight gradient is
 * computed in a padded layout (Focus stem: 12 real + 4 zero channels).
 * sy_pred_grad_fold: the parameter gradients of one head level's three 1x1 prediction convs (autograd of
 * tal_head.py:167-171): bias gradients (+)= column sums of d_raw over the level's B x rows anchors (deterministic two-stage
 * sum), weight gradients (+)= the [reg 0-3 | obj 4] / [cls] rows of `scratch` ([2][scratch_rows][ld] fp32, filled by two
 * sy_conv2d_wgrad launches), scratch zeroed behind the read.  `workspace`: sy_pred_grad_fold_workspace_floats floats. */
SY_API int sy_rows_add_f32(float* dst, int64_t ldd, float* src, int64_t lds, int rows, int cols, int zero_src, void* stream);
SY_API int64_t sy_pred_grad_fold_workspace_floats(int num_classes);
SY_API int sy_pred_grad_fold(const float* d_raw, int B, int64_t batch_stride, int rows, int num_classes, float* scratch,
                             int scratch_rows, int ld, int cin, float* g_reg, float* g_obj, float* g_cls, float* gb_reg, float* gb_obj, float* gb_cls,
                             float* workspace, void* stream);

/* ---- native launch tapes ---------------------------------------------------------------------------------------
 * Replaces the per-iteration Python dispatch of the reference trainer / evaluator loops
 * (exps/train_utils/double_trainer.py:95-131: ~2000 eager ops per train_one_iter; sAP/streamyolo/streamyolo_det.py:176-184
 * per streamed frame).  A plan's step is a fixed launch list: sy_tape_begin() opens a recording on the calling thread —
 * every kernel launch of every entry point above still executes, and its closure (kernel, grid, block, LDS size, argument
 * values) is appended; sy_tape_mark() appends stream-control entries between them; sy_tape_end() returns the tape.
 * sy_tape_replay() re-issues the list from C on the given streams, starting at entry *pos: it returns at a BREAK entry (and
 * at BUCKET entries when stop_buckets != 0) with *pos = the entry to resume at, *stop_kind / *stop_arg = that entry and
 * *stop_on_side = 1 if the launch cursor is on the side stream; at the end *stop_kind = SY_TAPE_END.  side_stream == NULL
 * (or == main_stream): one-stream replay, the stream marks are ignored.  A tape holds raw pointers: it is valid while the
 * buffers it was recorded over live.  Not thread safe per tape; recordings do not nest. */
enum { SY_TAPE_END = -1, SY_TAPE_LAUNCH = 0,
       SY_TAPE_SIDE = 1,     /* side stream waits for main; cursor -> side                                    */
       SY_TAPE_FORK = 2,     /* side stream waits for main; cursor unchanged                                  */
       SY_TAPE_SIDE_NW = 3,  /* cursor -> side, no new dependency                                             */
       SY_TAPE_MAIN = 4,     /* cursor -> main; arg >= 0: event on side = "ring slot `arg` is free again"      */
       SY_TAPE_ACQUIRE = 5,  /* main waits for the event of ring slot `arg` (if one is pending)               */
       SY_TAPE_JOIN = 6,     /* main waits for everything issued on side                                      */
       SY_TAPE_BREAK = 7,    /* return to the caller (host-side snippet `arg` runs there)                     */
       SY_TAPE_BUCKET = 8,   /* gradient bucket `arg` is final on every stream                                */
       /* plans with more than two chains (sy_tape_replay_n: streams[0] = main, [1] = side, [2..] = further chains) */
       SY_TAPE_CUR = 9,      /* cursor -> stream `arg`, no new dependency                                     */
       SY_TAPE_DEP = 10,     /* stream (arg >> 4) records an event, stream (arg & 15) waits for it            */
       SY_TAPE_SLOT_DONE = 11,   /* event on the CURRENT stream = "this stream's readers of ring slot `arg` are done"; a slot keeps
                                    one such event PER STREAM, the first one after an acquisition starts a new set      */
       SY_TAPE_ACQUIRE_CUR = 12  /* the CURRENT stream waits for every event of slot `arg`'s set recorded on ANOTHER stream
                                    (the set stays: several chains may acquire their part of one slot)                  */ };
SY_API void* sy_tape_begin(void);
SY_API int sy_tape_mark(int kind, int arg);
SY_API void* sy_tape_end(void);
SY_API int sy_tape_size(const void* tape, int* n_entries, int* n_launches);
/* events recorded / stream-waits issued by the last (or current) replay pass of `tape` */
SY_API int sy_tape_counters(const void* tape, int* n_events, int* n_waits);
SY_API int sy_tape_replay(void* tape, void* main_stream, void* side_stream, int* pos, int stop_buckets, int* stop_kind,
                          int* stop_arg, int* stop_on_side);
/* the same over n_streams (1..8) streams; *stop_stream = index of the cursor stream at a BREAK / BUCKET return.  A chain index
 * beyond n_streams (or a NULL entry) runs on streams[0]: a three-chain tape replays correctly on one or two streams. */
SY_API int sy_tape_replay_n(void* tape, void* const* streams, int n_streams, int* pos, int stop_buckets, int* stop_kind,
                            int* stop_arg, int* stop_stream);
SY_API void sy_tape_free(void* tape);

SY_API const char* sy_version(void);
SY_API int sy_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* STREAMYOLO_HIP_H */
