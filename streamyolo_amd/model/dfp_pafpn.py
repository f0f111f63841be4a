"""DFPPAFPN — dual-frame PAFPN + Dual-Flow-Perception fusion on the HIP execution plan.

Drop-in for exps/model/dfp_pafpn.py of the reference: same constructor (:18-26), attribute names /
state_dict keys (:27-105) and `forward(input, buffer=None, mode='off_pipe')` contract (:232-255),
including the 3-channel -> duplicated 6-channel branch (:236-238) and the on_pipe asserts (:251-252).
Outputs are NCHW-shaped float tensors (channels-last memory) exactly where the reference returns
NCHW tensors.
"""
import torch
import torch.nn as nn

from .darknet import CSPDarknet
from .network_blocks import BaseConv, CSPLayer
from .plan_cache import PlanCache


class DFPPAFPN(nn.Module):
    def __init__(self, depth=1.0, width=1.0, in_features=("dark3", "dark4", "dark5"),
                 in_channels=[256, 512, 1024], depthwise=False, act="silu"):
        super().__init__()
        if depthwise:
            raise NotImplementedError("depthwise=True is not used by any StreamYOLO cfg")
        self.backbone = CSPDarknet(depth, width, depthwise=depthwise, act=act)
        self.in_features = in_features
        self.in_channels = in_channels
        self.depth, self.width = depth, width
        c3, c4, c5 = [int(c * width) for c in in_channels]
        n = round(3 * depth)
        self.lateral_conv0 = BaseConv(c5, c4, 1, 1, act=act)
        self.C3_p4 = CSPLayer(2 * c4, c4, n, False, depthwise=depthwise, act=act)
        self.reduce_conv1 = BaseConv(c4, c3, 1, 1, act=act)
        self.C3_p3 = CSPLayer(2 * c3, c3, n, False, depthwise=depthwise, act=act)
        self.bu_conv2 = BaseConv(c3, c3, 3, 2, act=act)
        self.C3_n3 = CSPLayer(2 * c3, c4, n, False, depthwise=depthwise, act=act)
        self.bu_conv1 = BaseConv(c4, c4, 3, 2, act=act)
        self.C3_n4 = CSPLayer(2 * c4, c5, n, False, depthwise=depthwise, act=act)
        self.jian2 = BaseConv(c3, c3 // 2, 1, 1, act=act)
        self.jian1 = BaseConv(c4, c4 // 2, 1, 1, act=act)
        self.jian0 = BaseConv(c5, c5 // 2, 1, 1, act=act)
        self._plans = PlanCache()

    def forward(self, input, buffer=None, mode="off_pipe"):
        if self.training:
            # stand-alone training-mode call (the first half of the reference's YOLOX.forward, yolox.py:32): batch statistics,
            # running-statistics updates and an autograd node whose backward runs the HIP backward plan of the backbone.
            # YOLOX.forward itself uses the single fused forward + loss + backward plan instead (train_engine.train_forward).
            assert mode == "off_pipe", "on_pipe is an inference mode (dfp_pafpn.py:177-228 runs it under eval)"
            from ..train_engine import backbone_train_forward
            return backbone_train_forward(self, input)
        if mode == "off_pipe":
            if input.size()[1] == 3:
                input = torch.cat([input, input], dim=1)
            assert input.size()[1] == 6
            plan = self._plans.inference(self, None, "off_pipe", input)
            fused = plan.run_backbone(input)
            return tuple(f.export() for f in fused)
        elif mode == "on_pipe":
            if buffer is not None:
                assert len(buffer) == 3
                assert input.size()[1] == 3
            plan = self._plans.inference(self, None, "on_pipe", input)
            fused = plan.run_backbone(input, buffer)
            return tuple(f.export() for f in fused), plan.export_buffer()
        raise AssertionError(mode)
