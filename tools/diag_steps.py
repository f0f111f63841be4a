#!/usr/bin/env python3
"""Diagnosis: do the FIRST step's gradients (kernel variants are tuned during it: candidate launches on dummy tensors run between
the plan's own launches) equal the second step's (no tuning)?  l, 8 pairs, exact-fp32 mode, same weights and inputs."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import streamyolo_amd as sy
from oracle import streamyolo_oracle as O
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels
dev = torch.device("cuda:0")
dt = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cfg = O.OracleConfig.named("l")
sd = synth_state_dict(O.param_shapes(cfg), seed=0)
x = synth_frames(8, 600, 960, seed=2).to(dev)
lab, sup = synth_labels(8, 600, 960, cfg.num_classes, num_gt=16, seed=3)
lab, sup = lab.to(dev), sup.to(dev)
model = sy.build_model("l"); model.load_state_dict(sd, strict=True)
model = model.to(dev).train().set_compute_dtype(dt); model.head.use_l1 = True
grads = []
for it in range(3):
    for p in model.parameters(): p.grad = None
    model.load_state_dict(sd, strict=True)
    out = model(x, (lab, sup)); out["total_loss"].backward()
    grads.append({n: p.grad.detach().double().clone() for n, p in model.named_parameters()})
for a, b in ((0, 1), (1, 2)):
    errs = sorted(((float((grads[a][n] - grads[b][n]).norm() / grads[b][n].norm().clamp_min(1e-30)), n) for n in grads[a]), reverse=True)
    bad = [e for e in errs if e[0] > 1e-2]
    print("step %d vs %d: worst %.3e (%s), median %.2e, %d parameters above 1e-2" % (a, b, errs[0][0], errs[0][1], errs[len(errs) // 2][0], len(bad)))
    for e, n in bad[:12]:
        print("    %.3e %s" % (e, n))
