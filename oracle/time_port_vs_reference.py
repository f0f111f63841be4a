#!/usr/bin/env python3
"""bench.py's `cpu_baseline` is kind "port": the GPU box has no reference checkout, so it times oracle/streamyolo_oracle.py (the
functional restatement) on the host cores.  BASELINE.md section 3 prescribes the reference's OWN modules; this script records —
once, in the build container, where /root/reference exists — that the two time the same, so the port's number stands for the
reference's: the reference's unmodified exps/model/*.py (through oracle/ref_shim, as oracle/make_golden.py builds them) and the
port, same weights, same frames, same targets, forward + TAL loss + backward (or eval forward), same thread count, alternating.

Test infrastructure only.  Usage:  python oracle/time_port_vs_reference.py [--model s] [--height 600 --width 960] [--reps 5]
"""
import argparse
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import build_reference, clone_sd, O, synth_state_dict, synth_frames, synth_labels   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="s")
ap.add_argument("--height", type=int, default=600)
ap.add_argument("--width", type=int, default=960)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
a = ap.parse_args()
torch.set_num_threads(a.threads)
cfg = O.OracleConfig.named(a.model)
sd = synth_state_dict(O.param_shapes(cfg), seed=0)
x = synth_frames(1, a.height, a.width, seed=2)
lab, sup = synth_labels(1, a.height, a.width, cfg.num_classes, seed=3)
ref = build_reference(cfg)
ref.load_state_dict(sd, strict=True)
ref.head.use_l1 = True


def ref_train():
    ref.train()
    ref.zero_grad(set_to_none=True)
    out = ref(x, (lab, sup))
    out["total_loss"].backward()
    return float(out["total_loss"])


def port_train():
    s = clone_sd(sd, grad=True)
    out = O.forward_train(s, x, lab, sup, cfg)
    out["total_loss"].backward()
    return float(out["total_loss"])


def ref_eval():
    ref.eval()
    with torch.no_grad():
        return float(ref(x).sum())


def port_eval():
    with torch.no_grad():
        return float(O.forward_eval(sd, x, cfg).sum())


print("StreamYOLO-%s %dx%d, 1 frame pair, torch %s CPU fp32, %d threads; seconds per call, median of %d (alternating)" %
      (a.model, a.height, a.width, torch.__version__, a.threads, a.reps))
for name, f_ref, f_port in (("fwd + TAL loss + bwd", ref_train, port_train), ("eval forward + decode", ref_eval, port_eval)):
    f_ref(); f_port()                       # warm-up (oneDNN primitive caches); also resets the running statistics' drift equally
    tr, tp = [], []
    for _ in range(a.reps):
        t0 = time.perf_counter(); f_ref(); tr.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); f_port(); tp.append(time.perf_counter() - t0)
    mr, mp_ = sorted(tr)[len(tr) // 2], sorted(tp)[len(tp) // 2]
    print("%-24s reference %.3f s   port %.3f s   port / reference = %.3f" % (name, mr, mp_, mp_ / mr))
