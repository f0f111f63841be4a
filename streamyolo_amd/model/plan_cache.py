"""Per-module cache of compiled execution plans, keyed by (mode, batch, H, W, dtype, device)."""
import torch

from .. import ops


def compute_dtype_for(module, x):
    """Storage/MFMA dtype of a plan.  Explicit `module.compute_dtype` wins; else torch.autocast's
    dtype when active (the reference trainer's `--fp16`, double_trainer.py:108); else the parameter
    dtype (so `model.half()` — tools/eval.py:--fp16, streamyolo_det.py:109 — selects fp16)."""
    forced = getattr(module, "compute_dtype", None)
    if forced is not None:
        return ops.dtype_code(forced)
    dev = "cuda" if x.is_cuda else "cpu"
    if torch.is_autocast_enabled(dev):
        return ops.dtype_code(torch.get_autocast_dtype(dev))
    p = next(module.parameters())
    return ops.dtype_code(p.dtype if p.dtype in ops.DTYPE_CODE else torch.float32)


class PlanCache:
    """Not an nn.Module attribute that state_dict / deepcopy should carry: deepcopy yields an empty cache."""

    def __init__(self):
        self.plans = {}

    def __deepcopy__(self, memo):
        return PlanCache()

    def __getstate__(self):
        return {}

    def __setstate__(self, st):
        self.plans = {}

    def inference(self, pafpn, head, mode, x, decode=True, owner=None):
        from ..engine import InferencePlan
        owner = owner if owner is not None else (pafpn if pafpn is not None else head)
        dt = compute_dtype_for(owner, x)
        B, _, H, W = x.shape
        key = ("inf", mode, B, H, W, dt, str(x.device), decode, pafpn is not None, head is not None)
        plan = self.plans.get(key)
        if plan is None:
            plan = InferencePlan(pafpn, head, mode, B, H, W, dt, x.device, decode=decode)
            self.plans[key] = plan
        return plan

    def clear(self):
        self.plans.clear()
