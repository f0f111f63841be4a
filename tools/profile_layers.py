#!/usr/bin/env python3
"""Per-launch timing table of an inference plan (HIP events on the launch stream): which conv
shapes are far from the MFMA roofline.  Usage: python tools/profile_layers.py --model l --batch 8 --dtype bf16"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import streamyolo_amd as sy                                           # noqa: E402
from streamyolo_amd.utils.synth import synth_frames                  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="l")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--mode", default="off_pipe", choices=["off_pipe", "on_pipe"],
                    help="on_pipe: the streaming plan (one frame per step; BASELINE.json configs[4] with --batch 1 --dtype fp16)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = sy.build_model(a.model).to(dev).eval().set_compute_dtype(a.dtype)
    x = synth_frames(a.batch, 600, 960).to(dev)
    if a.mode == "on_pipe":
        x = x[:, 0:3].contiguous()
    plan = model._plans.inference(model.backbone, model.head, a.mode, x, owner=model)
    with torch.no_grad():
        for _ in range(2):
            if a.mode == "on_pipe":
                plan.run_stream(x, first=True)
            else:
                plan.run(x)
    rows = {}
    for it in range(a.iters):
        for i, op in enumerate(plan.ops):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            plan._run_op(op)
            e.record()
            rows.setdefault(i, []).append((s, e))
    torch.cuda.synchronize()
    tot_ms = tot_fl = 0.0
    print("%-28s %-34s %9s %9s" % ("op", "shape", "ms", "TFLOP/s"))
    for i, op in enumerate(plan.ops):
        ms = sorted(s.elapsed_time(e) for s, e in rows[i])[len(rows[i]) // 2]
        if op.kind == "conv":
            fl = 2.0 * op.x.C * op.y.C * op.k * op.k * op.y.pixels
            shape = "N%d %dx%d %d->%d k%d s%d t%s" % (op.x.N, op.y.H, op.y.W, op.x.C, op.y.C, op.k, op.stride, op._tiles.get("fwd"))
            tag = op.tag
        else:
            fl, shape, tag = 0.0, "", op.kind
        tot_ms += ms
        tot_fl += fl
        print("%-28s %-34s %9.4f %9.1f" % (tag, shape, ms, fl / ms / 1e9 if ms > 0 else 0))
    print("TOTAL %.3f ms  %.1f TFLOP/s (conv flops only, incl. launch gaps between events)" % (tot_ms, tot_fl / tot_ms / 1e9))


if __name__ == "__main__":
    main()
