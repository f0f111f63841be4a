#!/usr/bin/env python3
"""Loss curves of N fused SGD + EMA steps on one fixed synthetic batch, exact-fp32 mode (twice: the run-to-run floor) and bf16,
for a few learning rates — the exploration behind tests/test_lowp_yardstick.py::test_30_sgd_steps_bf16_tracks_fp32."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_lowp_yardstick import _run_curve                                  # noqa: E402

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for lr in [float(v) for v in (sys.argv[2:] or ["1e-5", "1.25e-4", "1e-3"])]:
    a = _run_curve("l", 8, 600, 960, "fp32", steps, lr, dev)
    b = _run_curve("l", 8, 600, 960, "fp32", steps, lr, dev)
    c = _run_curve("l", 8, 600, 960, "bf16", steps, lr, dev)
    idx = [0, 1, 2, 4, 9, 14, 19, 24, 29][:max(1, min(9, steps))]
    idx = [i for i in idx if i < steps]
    print("lr %g" % lr)
    print("  fp32 A  ", np.round(a[idx], 4))
    print("  fp32 B  ", np.round(b[idx], 4))
    print("  bf16    ", np.round(c[idx], 4))
    print("  |A-B|/A max %.3e   |bf16-A|/A max %.3e   at steps %d / %d" % (
        (np.abs(a - b) / np.abs(a)).max(), (np.abs(c - a) / np.abs(a)).max(), int((np.abs(a - b) / np.abs(a)).argmax()),
        int((np.abs(c - a) / np.abs(a)).argmax())))
