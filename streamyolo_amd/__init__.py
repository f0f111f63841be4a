"""streamyolo_amd — MI355X-native (gfx950) implementation of StreamYOLO's dual-frame detection hot path.

Public surface mirrors the reference's exps/model package:
    from streamyolo_amd import YOLOX, DFPPAFPN, TALHead, PIPEHead, CSPDarknet, postprocess
Every numeric result comes from hand-written HIP kernels (streamyolo_amd/csrc) behind a C ABI
(include/streamyolo_hip.h); importing this package without the built library fails on first use.
"""
from .model.darknet import CSPDarknet
from .model.dfp_pafpn import DFPPAFPN
from .model.pipe_head import PIPEHead
from .model.tal_head import TALHead
from .model.yolox import YOLOX
from .postprocess import postprocess

__all__ = ["YOLOX", "DFPPAFPN", "TALHead", "PIPEHead", "CSPDarknet", "postprocess", "build_model"]
__version__ = "0.1.0"

MODEL_ZOO = {
    # name: (depth, width, ignore_thr, ignore_value) — cfgs/{s,m,l}_s50_onex_dfp_tal_flip.py, l_s50_twox_dfp_tal_flip.py
    "s": (0.33, 0.50, 0.5, 1.5),
    "m": (0.67, 0.75, 0.4, 1.7),
    "l": (1.0, 1.0, 0.5, 1.6),
    "l2x": (1.0, 1.0, 0.4, 1.5),
    "nano": (0.33, 0.125, 0.5, 1.5),       # test-sized variant (not a reference cfg)
}


def build_model(name="s", num_classes=8, head="tal"):
    """What `Exp.get_model()` of the reference cfgs builds (cfgs/s_s50_onex_dfp_tal_flip.py:34-55):
    YOLOX(DFPPAFPN, TALHead) with BN eps 1e-3 / momentum 0.03 and `initialize_biases(1e-2)`; head="pipe": the still-image
    configuration YOLOX(DFPPAFPN, PIPEHead) of cfgs/l_s50_still_dfp_flip.py:34-54."""
    import torch.nn as nn
    depth, width, thr, val = MODEL_ZOO[name]
    in_channels = [256, 512, 1024]
    if head == "pipe":
        h = PIPEHead(num_classes, width, in_channels=in_channels)
    else:
        h = TALHead(num_classes, width, in_channels=in_channels, gamma=1.0, ignore_thr=thr, ignore_value=val)
    model = YOLOX(DFPPAFPN(depth, width, in_channels=in_channels), h)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03
    model.head.initialize_biases(1e-2)
    return model
