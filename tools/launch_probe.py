#!/usr/bin/env python3
"""Back-to-back launch cost of tiny kernels on one stream (the fixed per-kernel price of a dependent launch)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamyolo_amd import ops, _lib                                   # noqa: E402

dev = torch.device("cuda:0")
C = 32
ssum = torch.zeros(C, device=dev); ssq = torch.ones(C, device=dev)
g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
out = [torch.empty(C, device=dev) for _ in range(4)]


def one():
    ops.bn_finalize(ssum, ssq, 100, g, b, 1e-3, 0.03, None, None, *out)


for n in (1, 10, 100, 1000):
    one(); torch.cuda.synchronize()
    res = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s.record()
        for _ in range(n):
            one()
        e.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        res.append((s.elapsed_time(e) * 1e3 / n, (t1 - t0) * 1e6 / n))
    res.sort()
    print("chain %4d: %.2f us / launch on the GPU timeline, %.2f us host issue" % (n, res[2][0], res[2][1]))
# the same through a recorded tape (no Python wrapper cost)
with _lib.record() as tape:
    one()
h = ops.stream_of(ssum)
for n in (100, 1000):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    s.record()
    for _ in range(n):
        _lib.replay(tape, h)
    e.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("tape  %4d: %.2f us / launch on the GPU timeline, %.2f us host issue" % (n, s.elapsed_time(e) * 1e3 / n, (t1 - t0) * 1e6 / n))
