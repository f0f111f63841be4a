#!/usr/bin/env python3
"""Mint tests/golden/input_pipeline.npz from the REFERENCE's own input code: exps/data/data_augment_flip.py
(`preproc`, `_mirror`, `DoubleTrainTransform`) imported unmodified through oracle/ref_shim (yolox + a cv2 stand-in that
only resizes by 1x / exact 2x), and `Exp.preprocess`'s arithmetic (cfgs/l_s50_onex_dfp_tal_filp.py:161-172 — the cfg
file itself needs the full yolox.exp package, so its eight lines are executed verbatim below on torch).
Also asserts oracle/input_oracle.py reproduces every stored array.  Runs only where /root/reference exists."""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("STREAMYOLO_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "ref_shim"))
sys.path.insert(0, REF)

from oracle import input_oracle as IO                      # noqa: E402


def main():
    import cv2
    from exps.data.data_augment_flip import preproc, _mirror, DoubleTrainTransform
    rng = np.random.RandomState(7)
    B, H, W = 2, 24, 40
    out = {}
    # case A: frames already at the canvas size (the steady state of the reference's loader), one pair mirrored
    cur = rng.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    sup = rng.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    mirror = np.array([1, 0], dtype=np.uint8)
    ref = []
    for b in range(B):
        boxes = np.zeros((1, 4))
        ic, _ = _mirror(cur[b], boxes, mirror=bool(mirror[b]))
        isup, _ = _mirror(sup[b], boxes, mirror=bool(mirror[b]))
        ref.append(np.concatenate((preproc(ic, (H, W))[0], preproc(isup, (H, W))[0]), axis=0))
    out.update(a_cur=cur, a_sup=sup, a_mirror=mirror, a_ref=np.stack(ref))
    # case B: letterbox (shorter and narrower-than-canvas frames: same width, fewer rows)
    cur_b = rng.randint(0, 256, (B, H - 6, W, 3)).astype(np.uint8)
    out.update(b_cur=cur_b, b_ref=np.stack([preproc(cur_b[b], (H, W))[0] for b in range(B)]))
    # case C: full-resolution frames, exact 2x load-time resize (streamyolo_det.py:176-177 calls preproc on the raw frame)
    cur_c = rng.randint(0, 256, (B, 2 * H, 2 * W, 3)).astype(np.uint8)
    out.update(c_cur=cur_c, c_ref=np.stack([preproc(cur_c[b], (H, W))[0] for b in range(B)]))
    # case D: DoubleTrainTransform end to end on a pair with boxes (targets are CPU-side in the reference; stored so the
    # test can check the image halves and document the label layout [cls, cx, cy, w, h])
    random.seed(3)
    t = DoubleTrainTransform(max_labels=5, hsv=False, flip=True)
    boxes = np.array([[4.0, 3.0, 20.0, 15.0, 2.0], [10.0, 8.0, 30.0, 20.0, 5.0]])
    a_flag = None
    st = random.getstate()
    a_flag = random.randrange(2)
    random.setstate(st)
    i1, i2, l1, l2 = t((cur[0].copy(), sup[0].copy()), (boxes.copy(), boxes.copy()), (H, W))
    out.update(d_flag=np.array([a_flag], dtype=np.uint8), d_img=np.concatenate((i1, i2), axis=0), d_lab=l1, d_lab_sup=l2)
    # case E: Exp.preprocess — the cfg's eight lines, verbatim, on the case-A tensor
    inputs = torch.from_numpy(out["a_ref"]).clone()
    targets = (torch.tensor([[[1.0, 10.0, 6.0, 8.0, 4.0]]]).repeat(B, 1, 1), torch.tensor([[[1.0, 11.0, 7.0, 8.0, 4.0]]]).repeat(B, 1, 1))
    tsize, input_size = (32, 56), (H, W)
    scale_y = tsize[0] / input_size[0]
    scale_x = tsize[1] / input_size[1]
    if scale_x != 1 or scale_y != 1:
        inputs = torch.nn.functional.interpolate(inputs, size=tsize, mode="bilinear", align_corners=False)
        targets[0][..., 1::2] = targets[0][..., 1::2] * scale_x
        targets[0][..., 2::2] = targets[0][..., 2::2] * scale_y
        targets[1][..., 1::2] = targets[1][..., 1::2] * scale_x
        targets[1][..., 2::2] = targets[1][..., 2::2] * scale_y
    out.update(e_tsize=np.array(tsize), e_ref=inputs.numpy(), e_t0=targets[0].numpy(), e_t1=targets[1].numpy())

    # the oracle restatement must reproduce all of it
    assert np.array_equal(IO.pair_tensor(cur, sup, (H, W), 1, mirror).numpy(), out["a_ref"])
    assert np.array_equal(IO.pair_tensor(cur_b, None, (H, W)).numpy(), out["b_ref"])
    assert np.array_equal(IO.pair_tensor(cur_c, None, (H, W), 2).numpy(), out["c_ref"])
    assert np.array_equal(IO.pair_tensor(cur[:1], sup[:1], (H, W), 1, out["d_flag"]).numpy()[0], out["d_img"])
    t0 = (torch.tensor([[[1.0, 10.0, 6.0, 8.0, 4.0]]]).repeat(B, 1, 1), torch.tensor([[[1.0, 11.0, 7.0, 8.0, 4.0]]]).repeat(B, 1, 1))
    oi, ot = IO.exp_preprocess(torch.from_numpy(out["a_ref"]).clone(), t0, tsize, input_size)
    assert np.array_equal(oi.numpy(), out["e_ref"]) and np.array_equal(ot[0].numpy(), out["e_t0"])
    path = os.path.join(ROOT, "tests", "golden", "input_pipeline.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
