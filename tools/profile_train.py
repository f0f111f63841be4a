#!/usr/bin/env python3
"""Per-(kind, shape) timing table of one training step (HIP events on the launch stream).
Usage: python tools/profile_train.py --model l --batch 8 --dtype bf16"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import streamyolo_amd as sy                                                     # noqa: E402
from streamyolo_amd.train_engine import TrainStep                              # noqa: E402
from streamyolo_amd.utils.synth import synth_frames, synth_labels, synth_state_dict, load_bn_stats  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import streamyolo_oracle as O                                                   # noqa: E402  (shapes only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="l")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = O.OracleConfig.named(a.model)
    model = sy.build_model(a.model)
    bn = load_bn_stats(a.model) if a.model in ("nano", "s", "l") else None
    model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=bn), strict=True)
    model = model.to(dev).set_compute_dtype(a.dtype)
    x = synth_frames(a.batch, 600, 960, seed=2).to(dev)
    lab, sup = synth_labels(a.batch, 600, 960, cfg.num_classes, seed=3)
    st = TrainStep(model)
    for _ in range(3):
        st.step(x, (lab.to(dev), sup.to(dev)))
    rows = st.plan.profile(x, (lab.to(dev), sup.to(dev)), a.iters, detail=True)
    rows.sort(key=lambda r: -r[3])
    tot = sum(r[3] for r in rows)
    print("%-20s %-36s %6s %9s %6s %9s" % ("kind", "shape", "calls", "ms/step", "%", "TFLOP/s"))
    for kind, shape, calls, ms, fl in rows[:70]:
        print("%-20s %-36s %6.1f %9.4f %6.2f %9.1f" % (kind, shape, calls, ms, 100 * ms / tot, fl / ms / 1e9 if ms else 0))
    print("TOTAL %.3f ms" % tot)
    # which kernel variant the per-shape autotuner picked (tile codes: include/streamyolo_hip.h; wgrad: (tile, split-K workgroups))
    from collections import Counter
    picks = Counter()
    for op in st.plan.ops:
        if op.kind == "conv":
            for k, t in op._tiles.items():
                picks[("N%d %dx%d %d->%d k%d s%d" % (op.y.N, op.y.H, op.y.W, op.x.C, op.y.C, op.k, op.stride), k, str(t))] += 1
    print("tuned variants (shape, launch kind, code): launches")
    for (shape, k, t), n in sorted(picks.items()):
        print("  %-34s %-12s %-12s x%d" % (shape, k, t, n))


if __name__ == "__main__":
    main()
