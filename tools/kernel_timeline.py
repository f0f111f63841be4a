#!/usr/bin/env python3
"""In-kernel timeline of one launch, from the probe build of the library (make -C streamyolo_amd/csrc probe):

    STREAMYOLO_HIP_LIB=tools/probes/_build/libstreamyolo_probe.so python tools/kernel_timeline.py --kind conv --shape 10 --tile 117 --mode stats

Every workgroup's thread 0 stamps the device's 100 MHz clock at fixed points (sy_probe(slot), sy_device.h): 0 entry, 1 first
loads issued, 2 first operands landed (after the wait + barrier), 3 main loop done, 4 epilogue: tile converted / statistics
reduced, 5 output rows written (issued), 6 exit.  Printed: when workgroups start and end relative to the first start (the launch's
dispatch ramp and tail), and the median / p90 duration of each phase — where a latency-bound kernel spends its life.
kinds: conv (tools/conv_probe.py's SHAPES table; --mode fwd | stats | dgrad), wgrad (3x3 layers: tile 52 / 59 / 60), bnred (BatchNorm
backward reduce on the layer's output tensor)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from streamyolo_amd import _lib, ops                                  # noqa: E402
from streamyolo_amd.ops import View                                   # noqa: E402
from tools.conv_probe import SHAPES                                   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="conv", choices=["conv", "wgrad", "bnred"])
ap.add_argument("--shape", type=int, default=10)
ap.add_argument("--tile", type=int, default=117)
ap.add_argument("--mode", default="stats", choices=["fwd", "stats", "dgrad"])
ap.add_argument("--half", type=int, default=1, help="1: one frame's launch (N / 2 images), as the chain schedule issues it")
ap.add_argument("--batch", type=int, default=0, help="images in the launch (overrides the table's batch and --half): 1 = the streaming step")
ap.add_argument("--blocks", type=int, default=256, help="wgrad: split-K target workgroups")
a = ap.parse_args()
lib = _lib.lib()
dev = torch.device("cuda:0")
name, N, Ho, Wo, cin, cout, k, st = SHAPES[a.shape]
if a.batch:
    N = a.batch
elif a.half:
    N //= 2
H, W = Ho * st, Wo * st
g = torch.Generator().manual_seed(1)
x = View.alloc(N, H, W, cin, "bf16", dev); x.buf.copy_(torch.randn(x.buf.shape, generator=g).to(x.buf.dtype))
y = View.alloc(N, Ho, Wo, cout, "bf16", dev); y.buf.copy_(torch.randn(y.buf.shape, generator=g).to(y.buf.dtype))
w = (torch.randn(cout, k * k * cin, generator=g) / (cin * k * k) ** 0.5).to(x.buf.dtype).to(dev)
from streamyolo_amd.model.packing import pack_conv_weight_frag      # noqa: E402
wf = pack_conv_weight_frag(w, k)
reader = {"conv": ("sy_probe_read_conv_halo3_bf16" if (a.tile & 255) in (109, 100, 101, 98, 96, 97) else "sy_probe_read_conv_extra") if (((a.tile & 255) >= 104 or (a.tile & 255) in (96, 97, 98, 100, 101)) and (a.tile & 255) != 119) else "sy_probe_read_conv_igemm", "wgrad": "sy_probe_read_wgrad",
          "bnred": "sy_probe_read_train_ops"}[a.kind]
read = getattr(C.CDLL(_lib.library_path()), reader)
read.argtypes, read.restype = [C.c_void_p, C.c_int], C.c_int
NWG, NS = 8192, 8
buf = np.zeros(NWG * NS, dtype=np.uint64)


def launch():
    if a.kind == "conv":
        if a.mode == "stats":
            ops.conv2d(x, w, y, k, st, stats=(stats[0], stats[1]), tile=a.tile, wfrag=wf)
        elif a.mode == "dgrad":
            ops.conv2d(y, wt, x, k, st, mode=ops.CONV_DGRAD, tile=a.tile, wfrag=wft)
        else:
            ops.conv2d(x, w, y, k, st, torch.ones(cout, device=dev), torch.zeros(cout, device=dev), epilogue=ops.EPI_SILU, tile=a.tile, wfrag=wf)
    elif a.kind == "wgrad":
        ops.conv2d_wgrad(x, y, dw, k, st, oihw=True, workspace=ws, tile=a.tile, target_blocks=a.blocks)
    else:
        ops.bn_silu_bwd_reduce(y, y2, *aff, sums)


stats = (torch.zeros(32 * cout, device=dev), torch.zeros(32 * cout, device=dev))
wt = (torch.randn(cin, k * k * cout, generator=g) / (cout * k * k) ** 0.5).to(x.buf.dtype).to(dev)
wft = pack_conv_weight_frag(wt, k)
dw = torch.zeros(cout, cin, k, k, device=dev)
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
y2 = View.alloc(N, Ho, Wo, cout, "bf16", dev); y2.buf.copy_(y.buf)
aff = [torch.ones(cout, device=dev), torch.zeros(cout, device=dev), torch.zeros(cout, device=dev), torch.ones(cout, device=dev)]
sums = torch.zeros(4 * cout, device=dev)
for _ in range(3):
    launch()
torch.cuda.synchronize()
assert read(buf.ctypes.data, 1) == 0
s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s_.record(); launch(); e_.record()
torch.cuda.synchronize()
assert read(buf.ctypes.data, 0) == 0
t = buf.reshape(NWG, NS).astype(np.float64)
live = t[:, 0] > 0
t = t[live]
n = t.shape[0]
t0 = t[:, 0].min()
us = lambda v: (v - t0) / 100.0                                        # noqa: E731  (100 MHz ticks -> us)
print("%s %s: N%d %dx%d %d->%d k%d s%d, tile %d, mode %s — %d workgroups stamped (of <= %d), event-timed launch %.1f us"
      % (a.kind, name, N, Ho, Wo, cin, cout, k, st, a.tile, a.mode, n, NWG, s_.elapsed_time(e_) * 1e3))
pc = lambda v, q: float(np.percentile(v, q))                           # noqa: E731
st_, en_ = us(t[:, 0]), us(t[:, 6])
print("  workgroup START after the first: p50 %.2f  p90 %.2f  max %.2f us" % (pc(st_, 50), pc(st_, 90), st_.max()))
print("  workgroup END   after the first start: min %.2f  p50 %.2f  p90 %.2f  max %.2f us  (kernel span as the workgroups see it)" % (en_.min(), pc(en_, 50), pc(en_, 90), en_.max()))
names = ["entry -> loads issued", "loads issued -> first operands landed", "first operands landed -> main loop done",
         "main loop done -> tile converted / stats reduced", "-> output rows issued", "-> exit"]
prev = 0
for i, nm in enumerate(names, start=1):
    if not (t[:, i] > 0).all():
        continue
    d = (t[:, i] - t[:, prev]) / 100.0
    print("  %-52s p50 %6.2f  p90 %6.2f us" % (nm, pc(d, 50), pc(d, 90)))
    prev = i
if (t[:, 7] > 0).all():
    d = (t[:, 3] - t[:, 7]) / 100.0
    print("  %-52s p50 %6.2f  p90 %6.2f us" % ("(of the main loop: K-group reduction / loop exit -> slot 3)", pc(d, 50), pc(d, 90)))
life = (t[:, 6] - t[:, 0]) / 100.0
print("  workgroup lifetime                                    p50 %6.2f  p90 %6.2f us" % (pc(life, 50), pc(life, 90)))
