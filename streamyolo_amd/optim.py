"""Fused SGD-nesterov + EMA step: what `scaler.step(optimizer)` followed by
`ema_model.update(model)` do in the reference trainer (exps/train_utils/double_trainer.py:115-119), as ONE launch
of sy_sgd_ema_step over a table of tensors.

Parameter groups are yolox.exp.Exp.get_optimizer's (un-vendored yolox==0.3, restated): BatchNorm weights (no
decay), every other module weight (weight decay 5e-4), biases (no decay); SGD(lr, momentum 0.9, nesterov=True).
EMA is yolox.utils.ModelEMA: every floating state_dict entry, decay(updates) = 0.9998 * (1 - exp(-updates/2000)).
Gradients are read where the training plan leaves them (`p.grad`, views of the flat arena)."""
import math

import torch

from . import _lib, ops


def yolox_param_groups(model):
    """(bn_weights, decayed_weights, biases) exactly as yolox's get_optimizer walks `named_modules()`."""
    pg0, pg1, pg2 = [], [], []
    for name, m in model.named_modules():
        if hasattr(m, "bias") and isinstance(m.bias, torch.nn.Parameter):
            pg2.append(m.bias)
        if isinstance(m, torch.nn.BatchNorm2d) or "bn" in name:
            pg0.append(m.weight)
        elif hasattr(m, "weight") and isinstance(m.weight, torch.nn.Parameter):
            pg1.append(m.weight)
    return pg0, pg1, pg2


class FusedSGDEMA:
    def __init__(self, model, momentum=0.9, weight_decay=5e-4, ema_decay=0.9998, ema=True, ema_updates=0):
        self.model, self.momentum, self.ema_base = model, momentum, ema_decay
        self.updates, self.steps = int(ema_updates), 0
        pg0, pg1, pg2 = yolox_param_groups(model)
        wd = {id(p): weight_decay for p in pg1}
        self._wd = wd
        params = [p for p in pg0 + pg1 + pg2 if p.requires_grad]
        assert len({id(p) for p in params}) == len(params)
        dev = params[0].device
        self.device = dev
        self.params = params
        self.buf = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)      # momentum, flat
        sd = model.state_dict()
        self.ema_keys = [k for k, v in sd.items() if v.dtype.is_floating_point] if ema else []
        self.ema_other = {k: v.clone() for k, v in sd.items() if not v.dtype.is_floating_point} if ema else {}
        self.ema_flat = torch.empty(sum(sd[k].numel() for k in self.ema_keys), dtype=torch.float32, device=dev)
        self.ema_view, o = {}, 0
        for k in self.ema_keys:                                  # EMA starts as a copy of the model (ModelEMA.__init__)
            n = sd[k].numel()
            self.ema_view[k] = self.ema_flat[o:o + n].view(sd[k].shape)
            self.ema_view[k].copy_(sd[k])
            o += n
        self._sd_ptr = {v.data_ptr(): k for k, v in sd.items()}
        self._table_sig = None
        self._build(wd, sd)

    def _build(self, wd, sd):
        rows, o, chunks = [], 0, 0
        done = set()
        for p in self.params:
            e = _lib.OptimEntry()
            e.p, e.g = p.data_ptr(), (p.grad.data_ptr() if p.grad is not None else None)
            e.buf = self.buf[o:o + p.numel()].data_ptr()
            key = self._sd_ptr.get(p.data_ptr())
            e.ema = self.ema_view[key].data_ptr() if key in self.ema_view else None
            e.n, e.weight_decay, e.lr_mult, e.chunk0 = p.numel(), wd.get(id(p), 0.0), 1.0, chunks
            chunks += -(-p.numel() // 1024)
            o += p.numel()
            rows.append(e)
            done.add(key)
        for k in self.ema_keys:                                  # buffers (BatchNorm running statistics): EMA only
            if k in done:
                continue
            e = _lib.OptimEntry()
            e.p, e.g, e.buf, e.ema = sd[k].data_ptr(), None, None, self.ema_view[k].data_ptr()
            e.n, e.weight_decay, e.lr_mult, e.chunk0 = sd[k].numel(), 0.0, 1.0, chunks
            chunks += -(-sd[k].numel() // 1024)
            rows.append(e)
        arr = (_lib.OptimEntry * len(rows))(*rows)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self.n, self.chunks = len(rows), chunks
        self._table_sig = self._sig()

    def _sig(self):
        return tuple((p.data_ptr(), None if p.grad is None else p.grad.data_ptr()) for p in self.params)

    def step(self, lr, grad_scale=1.0):
        """One iteration: SGD update of every parameter from its .grad, then the EMA update (decay from the update
        counter, as ModelEMA.update).  `grad_scale` multiplies the gradients first (1/loss_scale, 1/world_size)."""
        if self._sig() != self._table_sig:                       # a .grad or parameter moved: re-point the table
            self._sd_ptr = {v.data_ptr(): k for k, v in self.model.state_dict().items()}
            self._build(self._wd, self.model.state_dict())
        assert all(p.grad is not None for p in self.params), "FusedSGDEMA.step: a parameter has no gradient"
        self.updates += 1
        d = self.ema_base * (1.0 - math.exp(-self.updates / 2000.0)) if self.ema_keys else 0.0
        ops.check(_lib.lib().sy_sgd_ema_step(self.table.data_ptr(), self.n, self.chunks, float(lr), float(self.momentum),
                                             float(grad_scale), float(d), 1 if self.steps == 0 else 0,
                                             ops.stream_of(self.table)), "sy_sgd_ema_step")
        self.steps += 1
        ops.bump_weights_epoch()                   # parameters rewritten through raw pointers (tensor._version unchanged)

    def ema_state_dict(self):
        """The EMA weights under the reference's checkpoint keys (what `ema_model.ema.state_dict()` returns)."""
        out = {k: v.clone() for k, v in self.ema_view.items()}
        out.update({k: v.clone() for k, v in self.ema_other.items()})
        return out
