"""TEST INFRASTRUCTURE ONLY — CPU oracle of the optimizer + EMA step.

Restates what the reference trainer runs per iteration (exps/train_utils/double_trainer.py:115-119):
`scaler.step(self.optimizer)` with the optimizer of yolox.exp.Exp.get_optimizer and `self.ema_model.update(model)`
with yolox.utils.ModelEMA (decay 0.9998, double_trainer.py:173-175).  Both classes live in the un-vendored
`yolox==0.3` dependency (README.md:67 of the reference); their published behaviour is restated here on top of
torch.optim.SGD itself, which pins the arithmetic:

* get_optimizer: walk `model.named_modules()`; every `.bias` Parameter -> group 2 (no decay); the `.weight` of a
  BatchNorm2d (or of any module whose name contains "bn") -> group 0 (no decay); every other `.weight` Parameter ->
  group 1 (weight_decay 5e-4); `SGD(group0, lr, momentum=0.9, nesterov=True)` + the two added groups.
* ModelEMA.update: `updates += 1; d = decay * (1 - exp(-updates / 2000))`; for every floating entry of
  `ema.state_dict()`: `v *= d; v += (1 - d) * model_state[k]`.

Parity unpinned by the reference itself (no tests, no golden vectors for the optimizer); pinned to torch.optim.SGD.
Only tests/ may import this module."""
import copy
import math

import torch


def param_groups(model):
    pg0, pg1, pg2 = [], [], []
    for name, m in model.named_modules():
        if hasattr(m, "bias") and isinstance(m.bias, torch.nn.Parameter):
            pg2.append(m.bias)
        if isinstance(m, torch.nn.BatchNorm2d) or "bn" in name:
            pg0.append(m.weight)
        elif hasattr(m, "weight") and isinstance(m.weight, torch.nn.Parameter):
            pg1.append(m.weight)
    return pg0, pg1, pg2


class ReferenceOptimEMA:
    def __init__(self, model, lr, momentum=0.9, weight_decay=5e-4, ema_decay=0.9998):
        pg0, pg1, pg2 = param_groups(model)
        self.model = model
        self.opt = torch.optim.SGD(pg0, lr=lr, momentum=momentum, nesterov=True)
        self.opt.add_param_group({"params": pg1, "weight_decay": weight_decay})
        self.opt.add_param_group({"params": pg2})
        self.ema = copy.deepcopy(model).eval()
        self.updates, self.decay = 0, ema_decay

    def step(self, lr, grad_scale=1.0):
        for g in self.opt.param_groups:
            g["lr"] = lr
            for p in g["params"]:
                if grad_scale != 1.0:
                    p.grad.mul_(grad_scale)
        self.opt.step()
        with torch.no_grad():
            self.updates += 1
            d = self.decay * (1 - math.exp(-self.updates / 2000))
            msd = self.model.state_dict()
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    v *= d
                    v += (1.0 - d) * msd[k].detach()
