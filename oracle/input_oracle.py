"""CPU restatement of the reference's input pipeline between `cv2.imread` and `model(inps, targets)` — TEST
INFRASTRUCTURE ONLY (imported by tests/, never by the product; streamyolo_amd/data.py has no CPU path).

Pinned against the reference's own code run in the build container: oracle/make_golden_input.py imports
exps/data/data_augment_flip.py unmodified (through oracle/ref_shim, incl. a `cv2` stand-in that only knows the
same-size and exact-2x resize) and stores its outputs in tests/golden/input_pipeline.npz; tests/test_data.py checks
this file against them.  The 2x decimation itself is restated from OpenCV's published source — PARITY UNPINNED for
that one step (no OpenCV here); the bilinear multi-scale resize IS torch's F.interpolate, as in the reference.
"""
import numpy as np
import torch
import torch.nn.functional as F


def load_time_resize(img, decimate):
    """exps/dataset/tal_flip_one_future_argoversedataset.py:179-187 / streamyolo_det.py:176-177 for the two ratios
    that need no OpenCV tables: r == 1 (copy) and r == 1/2 exactly (INTER_AREA fast path, (a+b+c+d+2)>>2)."""
    if decimate == 1:
        return img.copy()
    assert decimate == 2 and img.shape[0] % 2 == 0 and img.shape[1] % 2 == 0
    s = img.astype(np.int32)
    return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)


def cv2_resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize=(w, h), interpolation=cv2.INTER_LINEAR) for uint8 HWC images, restated from OpenCV's
    modules/imgproc/src/resize.cpp (8-bit fixed-point path; INTER_RESIZE_COEF_BITS = 11):
        scale = 1. / ((double)dst / src);  fx = (float)((dx + 0.5) * scale - 0.5);  sx = floor(fx);  fx -= sx
        HORIZONTAL only: sx < 0 -> (fx, sx) = (0, 0);  sx >= src - 1 -> (fx, sx) = (0, src - 1)
        VERTICAL: the fraction is kept, the two ROW INDICES are clamped (srows[k] = clip(sy + k, 0, src - 1)) — at the top /
        bottom border both taps read the same row with b0 + b1 = 2048, and the two separate >> 16 floors can land 1 LSB below
        the single-row value
        alpha / beta = saturate_cast<short>({1 - f, f} * 2048)              (cvRound: round half to even)
        horizontal:  D = S[sx] * alpha0 + S[sx + 1] * alpha1                 (int32)
        vertical:    dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    (both scale factors exactly 2 take the INTER_AREA fast path instead: load_time_resize(img, 2)).
    PARITY UNPINNED: no OpenCV binary exists in the build container."""
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = img.shape[:2]
    if sh == 2 * dh and sw == 2 * dw:
        return load_time_resize(img, 2)

    def taps(dn, sn, horizontal):
        scale = 1.0 / (float(dn) / float(sn))
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if horizontal:
            lo, hi = s < 0, s >= sn - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, sn - 1, s))
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return np.clip(s, 0, sn - 1), np.clip(s + 1, 0, sn - 1), a0, a1
    x0, x1, ax0, ax1 = taps(dw, sw, True)
    y0, y1, ay0, ay1 = taps(dh, sh, False)
    src = img.astype(np.int64)
    hor = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]           # [sh, dw, 3], scaled by 2^11
    S0, S1 = hor[y0], hor[y1]
    out = (((ay0[:, None, None] * (S0 >> 4)) >> 16) + ((ay1[:, None, None] * (S1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def load_time_resize_general(img, input_size):
    """`preproc`'s own resize for an arbitrary camera size (data_augment_flip.py:151-160):
    r = min(H / h, W / w); cv2.resize(img, (int(w * r), int(h * r)), INTER_LINEAR)."""
    r = min(input_size[0] / img.shape[0], input_size[1] / img.shape[1])
    return cv2_resize_linear_u8(img, (int(img.shape[1] * r), int(img.shape[0] * r)))


def mirror_image(img, mirror):
    """`_mirror`, image part: image[:, ::-1] (exps/data/data_augment_flip.py:140-148)."""
    return img[:, ::-1] if mirror else img


def preproc(img, input_size):
    """`preproc` with r == 1 (data_augment_flip.py:151-167): uint8 canvas of 114, image top-left, HWC -> CHW fp32."""
    assert img.shape[0] <= input_size[0] and img.shape[1] <= input_size[1]
    assert min(input_size[0] / img.shape[0], input_size[1] / img.shape[1]) == 1
    padded = np.ones((input_size[0], input_size[1], 3), dtype=np.uint8) * 114
    padded[:img.shape[0], :img.shape[1]] = img
    return np.ascontiguousarray(padded.transpose(2, 0, 1), dtype=np.float32)


def pair_tensor(cur, sup, input_size, decimate=1, mirror=None):
    """[B, 6, H, W] fp32 as the DataLoader collates it: per sample np.concatenate((img, support_img), axis=0)
    (exps/data/tal_flip_mosaicdetection.py:257) of the two `preproc` outputs, one mirror flag per pair
    (DoubleTrainTransform, data_augment_flip.py:213-222).  cur / sup: uint8 [B, Hs, Ws, 3] arrays (sup may be None)."""
    out = []
    rs = (lambda im: load_time_resize_general(im, input_size)) if decimate == 0 else (lambda im: load_time_resize(im, decimate))
    for b in range(cur.shape[0]):
        m = bool(mirror[b]) if mirror is not None else False
        planes = [preproc(mirror_image(rs(f[b]), m), input_size)
                  for f in ((cur, sup) if sup is not None else (cur,))]
        out.append(np.concatenate(planes, axis=0))
    return torch.from_numpy(np.stack(out))


def exp_preprocess(inputs, targets, tsize, input_size):
    """`Exp.preprocess` (cfgs/l_s50_onex_dfp_tal_filp.py:161-172), `self.input_size` passed explicitly."""
    scale_y = tsize[0] / input_size[0]
    scale_x = tsize[1] / input_size[1]
    if scale_x != 1 or scale_y != 1:
        inputs = F.interpolate(inputs, size=tsize, mode="bilinear", align_corners=False)
        targets[0][..., 1::2] = targets[0][..., 1::2] * scale_x
        targets[0][..., 2::2] = targets[0][..., 2::2] * scale_y
        targets[1][..., 1::2] = targets[1][..., 1::2] * scale_x
        targets[1][..., 2::2] = targets[1][..., 2::2] * scale_y
    return inputs, targets


def focus_pack(x):
    """[B, 3, H, W] -> [B, H/2, W/2, 12]: yolox Focus slicing, patch order TL, BL, TR, BR (trap T4), channels-last."""
    tl, bl, tr, br = x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]
    return torch.cat((tl, bl, tr, br), dim=1).permute(0, 2, 3, 1).contiguous()
