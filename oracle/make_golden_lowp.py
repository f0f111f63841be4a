#!/usr/bin/env python3
"""Yardstick for the 16-bit speed modes: how far do the REFERENCE's own gradients move when the
reference itself runs in reduced precision?

Runs only where /root/reference exists.  The reference's unmodified exps/model/* (through oracle/ref_shim, as
oracle/make_golden.py) does one training-mode forward + TAL loss + backward twice on the same synthetic weights / frames /
targets the GPU tests use: once in fp32, once under torch.autocast("cpu", dtype) — the CPU counterpart of the trainer's
`torch.cuda.amp.autocast` (exps/train_utils/double_trainer.py:100-108).  Recorded per case and dtype: the loss error and,
for every parameter, ||g_lowp - g_fp32|| / ||g_fp32||, plus median / p90 / max per parameter group (backbone, neck, head), and — because on the deep l model both the reference's
and our 16-bit gradients are uncorrelated with fp32 parameter by parameter (rel-L2 ~ sqrt(2)) — two statistics that still
separate "rounding noise" from "defect" there: the cosine to the fp32 gradient and |log(||g_lowp|| / ||g32||)|.
tests/test_lowp_yardstick.py asserts that the HIP path's bf16 / fp16 step stays within 2x of these numbers.

Test infrastructure only.  Usage:  python oracle/make_golden_lowp.py   -> tests/golden/lowp_yardstick.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden import build_reference                         # noqa: E402  (also puts ref_shim + the reference on sys.path)
from oracle import streamyolo_oracle as O                        # noqa: E402
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
NAMES = ("total_loss", "iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg")


def group_of(name):
    """backbone = CSPDarknet, neck = PAFPN + DFP jian convs, head = TALHead."""
    if name.startswith("head."):
        return "head"
    return "backbone" if name.startswith("backbone.backbone.") else "neck"


def summarise(errs, cos=None, ratio=None):
    out = {}
    for g in ("backbone", "neck", "head", "all"):
        names = [n for n in errs if g == "all" or group_of(n) == g]
        v = np.array(sorted(errs[n] for n in names))
        out[g] = {"n": int(v.size), "median": float(np.median(v)), "p90": float(v[int(round(0.9 * (v.size - 1)))]),
                  "max": float(v[-1])}
        if cos is not None:
            c = np.array(sorted(cos[n] for n in names))
            r = np.array(sorted(abs(np.log(max(ratio[n], 1e-30))) for n in names))
            out[g].update({"cos_median": float(np.median(c)), "cos_p10": float(c[int(round(0.1 * (c.size - 1)))]),
                           "abs_log_norm_ratio_median": float(np.median(r)),
                           "abs_log_norm_ratio_p90": float(r[int(round(0.9 * (r.size - 1)))])})
    return out


def run(cfg, sd, x, lab, sup, dtype):
    ref = build_reference(cfg)
    ref.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    ref.train()
    ref.head.use_l1 = True
    t0 = time.perf_counter()
    if dtype is None:
        out = ref(x.clone(), (lab.clone(), sup.clone()))
    else:
        with torch.autocast("cpu", dtype=dtype):
            out = ref(x.clone(), (lab.clone(), sup.clone()))
    out["total_loss"].float().backward()
    dt = time.perf_counter() - t0
    losses = [float(out[k]) for k in NAMES]
    return losses, {k: p.grad.detach().double() for k, p in ref.named_parameters()}, dt


def case(name, B, H, W, ngt, dtypes):
    cfg = O.OracleConfig.named(name)
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    x = synth_frames(B, H, W, seed=2)
    lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=ngt, seed=3)
    l32, g32, dt = run(cfg, sd, x, lab, sup, None)
    res = {"model": name, "shape": [B, H, W], "num_gt": ngt, "loss_fp32": l32, "seconds_fp32": round(dt, 2), "dtypes": {}}
    for dname, dtype in dtypes:
        l, g, dt = run(cfg, sd, x, lab, sup, dtype)
        errs = {k: float((g[k] - g32[k]).norm() / g32[k].norm().clamp_min(1e-30)) for k in g32}
        cos = {k: float((g[k] * g32[k]).sum() / (g[k].norm() * g32[k].norm()).clamp_min(1e-30)) for k in g32}
        ratio = {k: float(g[k].norm() / g32[k].norm().clamp_min(1e-30)) for k in g32}
        res["dtypes"][dname] = {"loss": l, "loss_rel": abs(l[0] - l32[0]) / abs(l32[0]), "groups": summarise(errs, cos, ratio),
                                "per_param": {k: round(v, 6) for k, v in errs.items()}, "seconds": round(dt, 2)}
        print(name, (B, H, W), dname, "loss rel %.3e" % res["dtypes"][dname]["loss_rel"],
              {g_: (round(s["median"], 4), round(s["p90"], 4), round(s["max"], 4)) for g_, s in res["dtypes"][dname]["groups"].items()},
              "%.1f s" % dt, flush=True)
    return res


def main():
    torch.set_num_threads(os.cpu_count())
    torch.manual_seed(0)
    cases = [("s", 2, 160, 256, 6), ("l", 1, 600, 960, 16)]
    if len(sys.argv) > 1:
        cases = [c for c in cases if c[0] in sys.argv[1:]]
    out = {"torch": torch.__version__, "what": "reference (exps/model/*, unmodified) under torch.autocast('cpu', dtype) vs its own fp32 "
           "run: per-parameter ||g - g32|| / ||g32||", "cases": []}
    for name, B, H, W, ngt in cases:
        out["cases"].append(case(name, B, H, W, ngt, [("bf16", torch.bfloat16), ("fp16", torch.float16)]))
    with open(os.path.join(GOLD, "lowp_yardstick.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("written", os.path.join(GOLD, "lowp_yardstick.json"))


if __name__ == "__main__":
    main()
