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
    for b in range(cur.shape[0]):
        m = bool(mirror[b]) if mirror is not None else False
        planes = [preproc(mirror_image(load_time_resize(f[b], decimate), m), input_size)
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
