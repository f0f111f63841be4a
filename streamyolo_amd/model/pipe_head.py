"""PIPEHead — the "still image" head of the reference (exps/model/pipe_head.py, used by
cfgs/l_s50_still_dfp_flip.py:37,49): the same decoupled YOLOX head and SimOTA assignment as TALHead, trained with ONE
label tensor and without the Trend-Aware weights (pipe_head.py:254-422 vs tal_head.py:262-470: the `ious_targets` /
`weight` block is absent, and the L1 term is guarded by `use_l1`).

On the HIP plan this is TALHead's loss kernel fed with `support = labels`: every ground truth then matches itself with
IoU 1 (>= any ignore_thr), every trend weight is the same constant and the normalised weights
`w * sum(l) / sum(w * l)` are 1 — sy_tal_loss computes exactly PIPEHead's loss and gradient."""
from .tal_head import TALHead


class PIPEHead(TALHead):
    single_labels = True          # YOLOX.forward / TrainStep pass `targets` (one [B, L, 5] tensor) as labels AND support

    def __init__(self, num_classes, width=1.0, strides=[8, 16, 32], in_channels=[256, 512, 1024], act="silu",
                 depthwise=False):
        super().__init__(num_classes, width=width, strides=strides, in_channels=in_channels, act=act,
                         depthwise=depthwise, gamma=1.0, ignore_thr=0.0, ignore_value=1.0)
