"""Per-module cache of compiled execution plans, keyed by (mode, batch, H, W, dtype, device).

Bounded: the reference trainer changes the input size every 10 iterations (multi-scale training, cfgs/l_s50_onex_dfp_tal_flip.py:
139-158 `random_resize`, exps/train_utils/double_trainer.py:276-279), eleven sizes for the l cfg — a plan owns a full set of
activation / gradient buffers, so only the most recently used few stay alive (LRU); an evicted plan's buffers go back to torch's
caching allocator and the next plan is carved out of them.  Shared scratch (split-K workspace, raw-gradient ring) is ONE
allocation per module, sized by the largest plan seen."""
import os
from collections import OrderedDict

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

    MAX_TRAIN_PLANS = int(os.environ.get("STREAMYOLO_MAX_TRAIN_PLANS", "3"))
    MAX_INFER_PLANS = int(os.environ.get("STREAMYOLO_MAX_INFER_PLANS", "8"))

    def __init__(self):
        self.plans = OrderedDict()
        self.scratch = {}                 # name -> uint8 tensor shared by every plan of this module (grows, never shrinks)
        self.scratch_gen = 0              # bumped when a scratch tensor is re-allocated: plans re-record their launch tapes

    def __deepcopy__(self, memo):
        return PlanCache()

    def __getstate__(self):
        return {}

    def __setstate__(self, st):
        self.__init__()

    def get(self, key, build):
        """The plan for `key` (most recently used last); `build()` makes it on a miss, the least recently used plan of the
        same kind ("train*" / "inf") beyond the bound is dropped."""
        plan = self.plans.get(key)
        if plan is not None:
            self.plans.move_to_end(key)
            return plan
        plan = build()
        self.plans[key] = plan
        kind = "train" if str(key[0]).startswith("train") else "inf"
        bound = self.MAX_TRAIN_PLANS if kind == "train" else self.MAX_INFER_PLANS
        same = [k for k in self.plans if ("train" if str(k[0]).startswith("train") else "inf") == kind]
        # never evict a plan whose forward has run but whose backward is still pending (its autograd node holds it)
        over = max(0, len(same) - bound)
        evicted = [k for k in same[:-1] if getattr(self.plans[k], "pending", 0) == 0][:over]
        for k in evicted:
            old = self.plans.pop(k)
            release = getattr(old, "release", None)
            if release is not None:
                release()
            del old
        if evicted:
            import gc
            gc.collect()                      # plans are webs of closures / views: return their buffers to the allocator now
        return plan

    def shared_scratch(self, name, nbytes, device):
        """One uint8 tensor per name, at least `nbytes` long, shared by all plans of the module (they never run concurrently:
        one training step at a time).  Growing it invalidates recorded raw pointers: scratch_gen changes."""
        t = self.scratch.get(name)
        if t is None or t.numel() < nbytes or t.device != device:
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self.scratch[name] = t
            self.scratch_gen += 1
        return t

    def inference(self, pafpn, head, mode, x, decode=True, owner=None, split_k=False):
        """split_k: the streaming step of this plan may run its deep small-map 3x3 layers as split-K (another fp32 summation
        order) — part of the key, so the facade's exact on_pipe path never shares a plan object with a StreamingDetector."""
        from ..engine import InferencePlan
        owner = owner if owner is not None else (pafpn if pafpn is not None else head)
        dt = compute_dtype_for(owner, x)
        B, _, H, W = x.shape
        key = ("inf", mode, B, H, W, dt, str(x.device), decode, pafpn is not None, head is not None, bool(split_k))

        def build():
            plan = InferencePlan(pafpn, head, mode, B, H, W, dt, x.device, decode=decode)
            plan.allow_split_k = bool(split_k)
            return plan
        return self.get(key, build)

    def clear(self):
        self.plans.clear()
        self.scratch.clear()
        self.scratch_gen += 1
