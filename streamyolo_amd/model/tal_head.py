"""TALHead — YOLOX decoupled head supervised by the Trend-Aware Loss, on the HIP execution plan.

Drop-in for exps/model/tal_head.py of the reference: constructor (:20-31), parameter tree /
state_dict keys (:47-131), `initialize_biases` (:141-150), `forward(xin, labels=None, imgs=None)`
(:152-223), `decode_outputs` (:245-260), and the attributes other code pokes (`use_l1`,
`decode_in_inference`, `hw`, `strides`, `num_classes`, `n_anchors` — SURVEY.md §8(b)).
"""
import math

import torch
import torch.nn as nn

from .network_blocks import BaseConv
from .plan_cache import PlanCache


class TALHead(nn.Module):
    def __init__(self, num_classes, width=1.0, strides=[8, 16, 32], in_channels=[256, 512, 1024], act="silu",
                 depthwise=False, gamma=1.5, ignore_thr=0.2, ignore_value=0.2):
        super().__init__()
        if depthwise:
            raise NotImplementedError("depthwise=True is not used by any StreamYOLO cfg")
        self.gamma, self.ignore_thr, self.ignore_value = gamma, ignore_thr, ignore_value
        self.n_anchors = 1
        self.num_classes = num_classes
        self.decode_in_inference = True          # for deploy, set to False (tal_head.py:45)
        self.width = width
        self.in_channels_ = list(in_channels)
        hw = int(256 * width)
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.stems = nn.ModuleList()
        for c in in_channels:
            self.stems.append(BaseConv(int(c * width), hw, 1, 1, act=act))
            self.cls_convs.append(nn.Sequential(BaseConv(hw, hw, 3, 1, act=act), BaseConv(hw, hw, 3, 1, act=act)))
            self.reg_convs.append(nn.Sequential(BaseConv(hw, hw, 3, 1, act=act), BaseConv(hw, hw, 3, 1, act=act)))
            self.cls_preds.append(nn.Conv2d(hw, self.n_anchors * num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(hw, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(hw, self.n_anchors * 1, 1, 1, 0))
        self.use_l1 = False
        self.strides = strides
        self.hw = None
        self._plans = PlanCache()

    def initialize_biases(self, prior_prob):
        v = -math.log((1 - prior_prob) / prior_prob)
        for conv in list(self.cls_preds) + list(self.obj_preds):
            b = conv.bias.view(self.n_anchors, -1)
            b.data.fill_(v)
            conv.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def forward(self, xin, labels=None, imgs=None):
        """On already-fused FPN features (NCHW tensors).  eval: decoded predictions; training (`labels` = the (targets,
        support targets) pair, tal_head.py:152-209): the reference's loss 6-tuple through a head-only forward + loss +
        backward plan (YOLOX.forward itself runs the single fused plan of the whole model)."""
        if self.training:
            assert labels is not None
            from ..train_engine import head_train_forward
            return head_train_forward(self, xin, labels)
        x0 = xin[0]
        B = x0.shape[0]
        s0 = self.strides[0]
        probe = x0.new_empty((B, 3, x0.shape[2] * s0, x0.shape[3] * s0))
        plan = self._plans.inference(None, self, "off_pipe", probe, decode=self.decode_in_inference)
        for v, t in zip(plan.fused, xin):
            assert tuple(t.shape) == (v.N, v.C, v.H, v.W), "feature map %s does not fit the head plan" % (tuple(t.shape),)
            v.set_nchw(t)
        out = plan.run_head()
        self.hw = [(v.H, v.W) for v in plan.fused]
        return out.clone().to(x0.dtype if x0.dtype.is_floating_point else torch.float32)

    def decode_outputs(self, outputs, dtype):
        """(xy + grid) * stride, exp(wh) * stride on a [B, A, 5+nc] tensor (tal_head.py:245-260; used by tools/eval.py:187-188
        when decode_in_inference is False): one sy_head_decode launch on the fp32 tensor; a 16-bit tensor (model.half()) is
        widened for it and rounded back once."""
        from .. import ops
        if outputs.dtype == torch.float32 and outputs.is_contiguous():
            return ops.head_decode(outputs, self.hw, self.strides)
        wide = outputs.float().contiguous()
        ops.head_decode(wide, self.hw, self.strides)
        outputs.copy_(wide)
        return outputs
