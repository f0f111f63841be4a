#!/usr/bin/env python3
"""Time sy_conv2d_wgrad variants (workgroup tile x split-K target) on representative StreamYOLO-l layers."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamyolo_amd import ops                                        # noqa: E402
from streamyolo_amd.ops import View                                   # noqa: E402
from tools.conv_probe import SHAPES                                   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--chain", type=int, default=1, help="launches per timed event pair")
    ap.add_argument("--variants", default="", help="tile/target pairs, e.g. 17/2048,18/256")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    variants = [(1, 1024), (2, 512), (4, 1024), (17, 512), (17, 1024), (33, 1024), (18, 512), (18, 1024), (34, 1024), (20, 1024), (22, 1024), (21, 1024)]
    if a.variants:
        variants = [tuple(int(v) for v in p.split("/")) for p in a.variants.replace("+", ",").split(",")]
    sel = [int(i) for i in a.shapes.replace("+", ",").split(",")] if a.shapes else range(len(SHAPES))
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    print("%-10s %-28s " % ("layer", "shape") + " ".join("%9s" % ("t%d/%d" % v) for v in variants) + "   (TFLOP/s)")
    for i in sel:
        name, N, Ho, Wo, cin, cout, k, st = SHAPES[i]
        H, W = Ho * st, Wo * st
        g = torch.Generator().manual_seed(i)                   # random operands (zero-filled ones inflate TF/s through DVFS)
        x = View.alloc(N, H, W, cin, "bf16", dev)
        x.buf.copy_(torch.randn(x.buf.shape, generator=g).to(x.buf.dtype))
        dy = View.alloc(N, ops.conv_out_size(H, k, st), ops.conv_out_size(W, k, st), cout, "bf16", dev)
        dy.buf.copy_(torch.randn(dy.buf.shape, generator=g).to(dy.buf.dtype))
        dw = torch.zeros(cout, cin, k, k, device=dev)
        flops = 2.0 * cin * cout * k * k * dy.pixels
        res = []
        for (t, tb) in variants:
            if (t & 255) in (52, 59, 60) and (k != 3 or st != 1 or cin % (64 if (t & 255) == 60 else 32)):
                res.append(float("nan")); continue
            if (t & 255) < 48 and (t & 15) in (1, 5, 6) and cout < 128 and t != 0:
                res.append(float("nan")); continue
            try:
                ops.conv2d_wgrad(x, dy, dw, k, st, oihw=True, workspace=ws, tile=t, target_blocks=tb)
                torch.cuda.synchronize()
                ts = []
                for _ in range(a.reps):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(a.chain):
                        ops.conv2d_wgrad(x, dy, dw, k, st, oihw=True, workspace=ws, tile=t, target_blocks=tb)
                    e.record()
                    torch.cuda.synchronize()
                    ts.append(s.elapsed_time(e) / a.chain)
                ts.sort()
                res.append(flops / (ts[len(ts) // 2] * 1e-3) / 1e12)
            except Exception as ex:                                    # noqa: BLE001
                res.append(float("nan"))
        print("%-10s %-28s " % (name, "N%d %dx%d %d->%d k%d s%d" % (N, dy.H, dy.W, cin, cout, k, st)) + " ".join("%9.1f" % r for r in res))


if __name__ == "__main__":
    main()
