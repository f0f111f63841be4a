#!/usr/bin/env python3
"""Mint tests/golden/nano_pipe_train_2x64x96.npz from the REFERENCE's PIPEHead (exps/model/pipe_head.py, the head of
cfgs/l_s50_still_dfp_flip.py:37,49) — same procedure as oracle/make_golden.py (reference imported unmodified through
oracle/ref_shim, synthetic weights / frames / labels from streamyolo_amd.utils.synth), one label tensor, with and
without the L1 term.  Also asserts that the TAL oracle fed with support = labels reproduces PIPEHead (the identity
streamyolo_amd/model/pipe_head.py relies on).  Test infrastructure only; runs where /root/reference exists."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("STREAMYOLO_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "ref_shim"))
sys.path.insert(0, REF)

from oracle import streamyolo_oracle as O                      # noqa: E402
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels  # noqa: E402

NAMES = ("total_loss", "iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg")


def main():
    import torch.nn as nn
    from exps.model.yolox import YOLOX
    from exps.model.dfp_pafpn import DFPPAFPN
    from exps.model.pipe_head import PIPEHead
    cfg = O.OracleConfig.named("nano")
    B, H, W = 2, 64, 96
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    x = synth_frames(B, H, W, seed=2)
    lab, _ = synth_labels(B, H, W, cfg.num_classes, num_gt=6, seed=3)
    out = {"shape": np.array([B, H, W], dtype=np.int32)}
    for use_l1 in (True, False):
        ref = YOLOX(DFPPAFPN(cfg.depth, cfg.width, in_channels=list(cfg.in_channels)),
                    PIPEHead(cfg.num_classes, cfg.width, in_channels=list(cfg.in_channels)))
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = cfg.bn_eps, cfg.bn_momentum
        ref.head.initialize_biases(1e-2)
        ref.load_state_dict(sd, strict=True)
        ref.train()
        ref.head.use_l1 = use_l1
        res = ref(x.clone(), lab.clone())
        res["total_loss"].backward()
        losses = np.array([float(res[k]) for k in NAMES], dtype=np.float64)
        tag = "l1" if use_l1 else "nol1"
        out["losses_" + tag] = losses
        grads = {k: p.grad.clone() for k, p in ref.named_parameters()}
        out["grad_norms_" + tag] = np.array([float(grads[k].double().norm()) for k in sorted(grads)])
        for k in ("head.reg_preds.0.weight", "head.cls_preds.1.bias", "backbone.backbone.dark3.1.m.0.conv2.conv.weight",
                  "backbone.jian1.conv.weight"):
            out["grad_%s:%s" % (tag, k)] = grads[k].numpy()
        if use_l1:                                           # the identity the HIP path uses: TAL(support = labels) == PIPE
            osd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running_" not in k else v.clone())
                   for k, v in sd.items()}
            o = O.forward_train(osd, x, lab, lab, cfg)
            lo = np.array([float(o[k]) for k in NAMES])
            assert np.abs(lo - losses).max() / np.abs(losses).max() < 1e-5, (lo, losses)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nano_pipe_train_2x64x96.npz"), **out)
    print("PIPEHead golden:", out["losses_l1"], out["losses_nol1"])


if __name__ == "__main__":
    main()
