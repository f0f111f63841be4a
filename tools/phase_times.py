#!/usr/bin/env python3
"""Wall time of the three phases of the taped training step, each bracketed by synchronisations (so their sum exceeds the step:
phases overlap a little across their borders): forward, loss, backward.  python tools/phase_times.py [--model l] [--batch 8]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="l")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
import streamyolo_amd as sy                                             # noqa: E402
from oracle import streamyolo_oracle as O                                # noqa: E402
from streamyolo_amd.train_engine import TrainStep                        # noqa: E402
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels, load_bn_stats   # noqa: E402

dev = torch.device("cuda:0")
cfg = O.OracleConfig.named(a.model)
model = sy.build_model(a.model)
model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats(a.model)), strict=True)
model = model.to(dev).set_compute_dtype(a.dtype)
x = synth_frames(a.batch, 600, 960, seed=2).to(dev)
lab, sup = synth_labels(a.batch, 600, 960, cfg.num_classes, seed=3)
lab, sup = lab.to(dev), sup.to(dev)
st = TrainStep(model, graph=False)
for _ in range(4):
    st.step(x, (lab, sup))
plan = st.plan
torch.cuda.synchronize()
t = {"forward": 0.0, "loss": 0.0, "backward": 0.0, "step": 0.0}
for _ in range(a.steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    plan.forward(x)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    _, d_raw = plan.loss(lab, sup)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    plan.backward(d_raw)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    t["forward"] += t1 - t0; t["loss"] += t2 - t1; t["backward"] += t3 - t2
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    st.step(x, (lab, sup))
torch.cuda.synchronize()
t["step"] = time.perf_counter() - t0
print("%s, %d pairs, %s: ms per step — " % (a.model, a.batch, a.dtype) + "  ".join("%s %.3f" % (k, v / a.steps * 1e3) for k, v in t.items()))
