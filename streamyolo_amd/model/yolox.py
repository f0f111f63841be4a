"""YOLOX — top-level StreamYOLO module (mirrors exps/model/yolox.py:12-55 of the reference).

forward(x, targets=None, buffer=None, mode='off_pipe'):
  off_pipe + eval   : decoded [B, A, 5+nc]                       (yolox.py:47-49)
  off_pipe + train  : dict(total_loss, iou_loss, l1_loss, conf_loss, cls_loss, num_fg)   (:33-46)
  on_pipe           : (decoded, buffer_)                         (:51-55)
All arithmetic runs in HIP execution plans (streamyolo_amd/engine.py, train_engine.py).
"""
import torch
import torch.nn as nn

from ..data import FramePairsU8
from .dfp_pafpn import DFPPAFPN
from .plan_cache import PlanCache
from .tal_head import TALHead


class YOLOX(nn.Module):
    def __init__(self, backbone=None, head=None):
        super().__init__()
        if backbone is None:
            backbone = DFPPAFPN()
        if head is None:
            head = TALHead(20)
        self.backbone = backbone
        self.head = head
        self.compute_dtype = None          # None: follow autocast / parameter dtype; or "fp32" | "bf16" | "fp16"
        self._plans = PlanCache()

    def set_compute_dtype(self, dtype):
        """'fp32' = exact-f32 MFMA parity mode, 'bf16' / 'fp16' = speed modes (fp32 accumulate)."""
        self.compute_dtype = dtype
        self.backbone.compute_dtype = dtype
        self.head.compute_dtype = dtype
        for m in (self, self.backbone, self.head):
            m._plans.clear()
        return self

    def forward(self, x, targets=None, buffer=None, mode="off_pipe"):
        assert mode in ["off_pipe", "on_pipe"]
        out_dtype = x.dtype if x.dtype.is_floating_point else torch.float32
        if mode == "off_pipe":
            if self.training:
                assert targets is not None
                from ..train_engine import train_forward
                return train_forward(self, x, targets)
            if x.size()[1] == 3:                                   # dfp_pafpn.py:236-238
                x = x.paired_with_self() if isinstance(x, FramePairsU8) else torch.cat([x, x], dim=1)
            assert x.size()[1] == 6
            plan = self._plans.inference(self.backbone, self.head, "off_pipe", x,
                                         decode=self.head.decode_in_inference, owner=self)
            out = plan.run(x)
            self.head.hw = [(v.H, v.W) for v in plan.fused]
            return out.clone().to(out_dtype)
        if buffer is not None:
            assert len(buffer) == 3
            assert x.size()[1] == 3
        plan = self._plans.inference(self.backbone, self.head, "on_pipe", x,
                                     decode=self.head.decode_in_inference, owner=self)
        out = plan.run(x, buffer)
        self.head.hw = [(v.H, v.W) for v in plan.fused]
        return out.clone().to(out_dtype), plan.export_buffer()
