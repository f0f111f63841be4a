#!/usr/bin/env python3
"""Fused Bottleneck forward (tile 119: 1x1 -> 3x3 in one launch, hidden activation in LDS) against the two launches it replaces,
on StreamYOLO-l's Bottleneck shapes at batch 1 (one streaming frame) and batch 16 (8 frame pairs of the eval forward).
us per Bottleneck, median; the `chain` launches of a timed event pair are back to back on one stream."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamyolo_amd import ops                                        # noqa: E402
from streamyolo_amd.ops import View                                   # noqa: E402
from streamyolo_amd.model.packing import pack_conv_weight_frag      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="fp16")
ap.add_argument("--chain", type=int, default=16)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
SHAPES = [("dark2.m", 150, 240, 64), ("dark3.m", 75, 120, 128), ("dark4.m", 38, 60, 256), ("C3_p3.m", 75, 120, 128), ("C3_n3.m", 38, 60, 256)]
print("%-9s %-22s %12s %12s %8s" % ("layer", "shape", "two launches", "fused (119)", "ratio"))
for N in (1, 16):
    for name, H, W, hid in SHAPES:
        g = torch.Generator().manual_seed(hid)
        code = ops.dtype_code(a.dtype)
        x = View.alloc(N, H, W, hid, a.dtype, dev); x.buf.copy_(torch.randn(x.buf.shape, generator=g).to(x.buf.dtype))
        w1 = (torch.randn(hid, hid, generator=g) / hid ** 0.5).to(x.buf.dtype).to(dev)
        w2 = (torch.randn(hid, 9 * hid, generator=g) / (9 * hid) ** 0.5).to(x.buf.dtype).to(dev)
        w1f, w2f = pack_conv_weight_frag(w1, 1), pack_conv_weight_frag(w2, 3)
        s = torch.ones(hid, device=dev); b = torch.zeros(hid, device=dev)
        h = View.alloc(N, H, W, hid, a.dtype, dev)
        y = View.alloc(N, H, W, hid, a.dtype, dev)
        t1 = ops.tuned_tile(ops.CONV_FWD, code, N, H, W, hid, hid, 1, 1, dev)
        t3 = ops.tuned_tile(ops.CONV_FWD, code, N, H, W, hid, hid, 3, 1, dev)

        def two():
            ops.conv2d(x, w1, h, 1, 1, s, b, epilogue=ops.EPI_SILU, tile=t1, wfrag=w1f if t1 >= ops.TILE_WR else None)
            ops.conv2d(h, w2, y, 3, 1, s, b, res=x, epilogue=ops.EPI_SILU, tile=t3, wfrag=w2f if t3 >= ops.TILE_WR else None)

        def one():
            ops.conv2d(x, w2, y, 3, 1, s, b, res=x, epilogue=ops.EPI_SILU, tile=119, wfrag=w2f, pre=(w1f, s, b))
        res = []
        for fn in (two, one):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()                      # replayed from a hipGraph, as the streaming step is: no Python between launches
            with torch.cuda.graph(gr):
                for _ in range(a.chain):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                gr.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / a.chain * 1e3)
            res.append(sorted(ts)[len(ts) // 2])
        print("%-9s %-22s %9.1f us %9.1f us %8.2f   (tiles %d + %d)" % (name, "N%d %dx%d c%d" % (N, H, W, hid), res[0], res[1], res[0] / res[1], t1, t3))
