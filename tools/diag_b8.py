#!/usr/bin/env python3
"""Diagnosis: per-parameter gradient error of the l 8x600x960 exact-mode step vs the oracle (worst 40), under optional env switches."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import streamyolo_amd as sy
from test_lowp_yardstick import _oracle_grads, NAMES
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg, sd, x, lab, sup, ref, rgrads = _oracle_grads("l", B, 600, 960, 16)
if os.environ.get("DIRTY"):
    # fill the caching allocator's free blocks with a poison value: every torch.empty buffer of the plan then starts dirty
    poison = float(os.environ["DIRTY"])
    junk = [torch.full((1 << 28,), poison, device=dev) for _ in range(40)]      # 40 GiB
    junk += [torch.full((1 << 22,), poison, device=dev) for _ in range(256)]
    del junk
model = sy.build_model("l"); model.load_state_dict(sd, strict=True)
model = model.to(dev).train().set_compute_dtype("fp32"); model.head.use_l1 = True
for it in range(2):
    for p in model.parameters(): p.grad = None
    model.load_state_dict(sd, strict=True)
    out = model(x.to(dev), (lab.to(dev), sup.to(dev))); out["total_loss"].backward()
    errs = sorted(((float((p.grad.detach().cpu().double() - rgrads[n]).norm() / rgrads[n].norm().clamp_min(1e-30)), n,
                    float(p.grad.detach().cpu().double().norm() / rgrads[n].norm().clamp_min(1e-30))) for n, p in model.named_parameters()), reverse=True)
    print("step", it, "loss", float(out["total_loss"]), float(ref["total_loss"]))
    for e, n, r in errs[:25]:
        print("   %.4e  norm ratio %.4f  %s" % (e, r, n))
    print("   median %.3e" % errs[len(errs) // 2][0])
