#!/usr/bin/env python3
"""Time every sy_conv2d kernel variant (workgroup tile x staging strategy) on representative layer
shapes of StreamYOLO-l at batch 8.  Usage: python tools/conv_probe.py [--shapes i,j] [--tiles 1,3,19] [--reps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamyolo_amd import ops                                        # noqa: E402
from streamyolo_amd.ops import View                                   # noqa: E402

# (name, N, H(out), W(out), Cin, Cout, k, stride)
SHAPES = [
    ("stem", 16, 300, 480, 16, 64, 3, 1),
    ("dark2.0", 16, 150, 240, 64, 128, 3, 2),
    ("d2.m.c1", 16, 150, 240, 64, 64, 1, 1),
    ("d2.m.c2", 16, 150, 240, 64, 64, 3, 1),
    ("d2.conv3", 16, 150, 240, 128, 128, 1, 1),
    ("d3.m.c1", 16, 75, 120, 128, 128, 1, 1),
    ("d3.m.c2", 16, 75, 120, 128, 128, 3, 1),
    ("d3.conv3", 16, 75, 120, 256, 256, 1, 1),
    ("dark4.0", 16, 38, 60, 256, 512, 3, 2),
    ("d4.m.c1", 16, 38, 60, 256, 256, 1, 1),
    ("d4.m.c2", 16, 38, 60, 256, 256, 3, 1),
    ("d5.m.c2", 16, 19, 30, 512, 512, 3, 1),
    ("spp.c2", 16, 19, 30, 2048, 1024, 1, 1),
    ("head0.3x3", 8, 75, 120, 256, 256, 3, 1),
    ("head1.3x3", 8, 38, 60, 256, 256, 3, 1),
    ("head2.3x3", 8, 19, 30, 256, 256, 3, 1),
    ("p4.reduce", 16, 38, 60, 512, 256, 1, 1),
    ("d5.m.c1", 16, 19, 30, 512, 512, 1, 1),
    ("d2.conv1", 16, 150, 240, 128, 64, 1, 1),
    ("d3.conv1", 16, 75, 120, 256, 128, 1, 1),
    ("head0.stem", 8, 75, 120, 256, 256, 1, 1),
    ("d5.conv3", 16, 19, 30, 1024, 1024, 1, 1),
    ("p5.csp.c1", 16, 19, 30, 1024, 512, 1, 1),
    ("p4.csp.c1", 16, 38, 60, 1024, 256, 1, 1),
    # one frame's launches (8 images) of the training step's chains
    ("d3.m.c2/f", 8, 75, 120, 128, 128, 3, 1),
    ("d5.m.c2/f", 8, 19, 30, 512, 512, 3, 1),
    ("d2.m.c2/f", 8, 150, 240, 64, 64, 3, 1),
    # the stride-2 layers of one frame (indices 27-32): --mode dgrad times their data gradients (tiles 108, 125-127 + the implicit GEMM)
    ("dark2.0/f", 8, 150, 240, 64, 128, 3, 2),
    ("dark3.0/f", 8, 75, 120, 128, 256, 3, 2),
    ("dark4.0/f", 8, 38, 60, 256, 512, 3, 2),
    ("dark5.0/f", 8, 19, 30, 512, 1024, 3, 2),
    ("bu_conv2/f", 8, 38, 60, 256, 256, 3, 2),
    ("bu_conv1/f", 8, 19, 30, 512, 512, 3, 2),
    # one frame's 1x1 launches (indices 33-36): --mode stats with tiles 121, 121 + 8 * 256 (no atomics), 121 + 24 * 256 (no statistics)
    ("d3.m.c1/f", 8, 75, 120, 128, 128, 1, 1),
    ("d4.m.c1/f", 8, 38, 60, 256, 256, 1, 1),
    ("d5.m.c1/f", 8, 19, 30, 512, 512, 1, 1),
    ("d4.conv3/f", 8, 38, 60, 512, 512, 1, 1),
]
NAMES = {0: "auto", 1: "dma256x256", 2: "dma128x256", 3: "dma128x128", 4: "dma64x256", 5: "dma32x256", 6: "dma128x64",
         7: "dma64x64", 17: "rs256x256", 18: "rs128x256", 19: "rs128x128", 20: "rs64x256", 21: "rs32x256",
         22: "rs128x64", 23: "rs64x64", 35: "d2-128x128", 36: "d2-64x256", 38: "d2-128x64", 39: "d2-64x64",
         51: "d3-128x128", 52: "d3-64x256", 54: "d3-128x64", 55: "d3-64x64",
         96: "halo3-64x4", 97: "halo3-64x6", 100: "halo3-128x3", 101: "halo3-128x2", 98: "halo3-128x4", 121: "k1-128x64", 122: "k1-64x128", 123: "k1-128x128", 117: "halo2-128x2", 118: "halo2-128x4w", 107: "halo2-128x3", 104: "halo2-128x5",
         112: "halo128x4", 113: "halo128x4w", 114: "halo128x2-8w", 115: "halo128x2", 116: "halo64x8",
         119: "halo2-128x8/8acc", 111: "halo2-256x4/8acc", 109: "halo3-128x5", 110: "halo2-s2-128x2", 108: "s2dgrad-128x2", 125: "s2dgrad4-128x1", 126: "s2dgrad4-128x2", 127: "s2dgrad4-64x2",
         99: "w3-128x128", 102: "w3-128x64", 103: "w3-64x64", 24: "rs256x64", 88: "wr256x64", 89: "wr256x128", 83: "wr128x128", 84: "wr64x256", 85: "wr32x256", 86: "wr128x64", 87: "wr64x64"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="")
    ap.add_argument("--tiles", default="0,22,38,83,86,87,88,89,24")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--chain", type=int, default=1, help="launches per timed event pair (amortises the event / launch gap)")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "dgrad", "stats"],
                    help="dgrad: the data gradient of the layer (3x3 s1 / 1x1 shapes); stats: training forward (raw output + BatchNorm sums)")
    ap.add_argument("--copies", type=int, default=16, help="stats mode: replicas of the statistics arrays per segment")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    # ("+" works as a separator too: tools/gpu.sh splits task arguments at commas)
    tiles = [int(t) for t in a.tiles.replace("+", ",").split(",")]
    sel = [int(i) for i in a.shapes.replace("+", ",").split(",")] if a.shapes else range(len(SHAPES))
    def nm(t):
        return NAMES.get(t & 255, "tile%d" % (t & 255)) + {0: "", 1: "-noX", 2: "-noW", 3: "-noXW", 8: "-noAtom", 24: "-noStatRed"}[t >> 8]
    print("%-10s %-28s " % ("layer", "shape") + " ".join("%15s" % nm(t) for t in tiles) + "   (TFLOP/s, median of %d)" % a.reps)
    for i in sel:
        name, N, Ho, Wo, cin, cout, k, st = SHAPES[i]
        H, W = Ho * st, Wo * st
        g = torch.Generator().manual_seed(i)
        x = View.alloc(N, H, W, cin, a.dtype, dev)
        x.buf.copy_(torch.randn(x.buf.shape, generator=g).to(x.buf.dtype))
        w = (torch.randn(cout, k * k * cin, generator=g) / (cin * k * k) ** 0.5).to(x.buf.dtype).to(dev)
        y = View.alloc(N, ops.conv_out_size(H, k, st), ops.conv_out_size(W, k, st), cout, a.dtype, dev)
        from streamyolo_amd.model.packing import pack_conv_weight_frag
        wf = pack_conv_weight_frag(w, k)
        scale = torch.ones(cout, device=dev)
        shift = torch.zeros(cout, device=dev)
        kw = dict(epilogue=ops.EPI_SILU)
        if a.mode == "dgrad":
            scale = shift = None
            kw = dict(mode=ops.CONV_DGRAD)
            if st != 1:
                # stride 2: the launch's input is dy on the SMALL map, its output dx on the large one, weights transposed
                x, y = View.alloc(N, y.H, y.W, cout, a.dtype, dev), View.alloc(N, H, W, cin, a.dtype, dev)
                x.buf.copy_(torch.randn(x.buf.shape, generator=g).to(x.buf.dtype))
                w = (torch.randn(cin, k * k * cout, generator=g) / (cout * k * k) ** 0.5).to(x.buf.dtype).to(dev)
                wf = pack_conv_weight_frag(w, k)
        if a.mode == "stats":
            scale = shift = None
            kw = dict(stats=(torch.zeros(2 * a.copies * cout, device=dev), torch.zeros(2 * a.copies * cout, device=dev)), segments=2)
        flops = 2.0 * cin * cout * k * k * (x.pixels if (a.mode == "dgrad" and st != 1) else y.pixels)   # products of the forward op
        res = []
        for t in tiles:
            if 121 <= (t & 255) <= 123 and (k != 1 or st != 1 or cin not in (64, 128, 256, 512, 1024, 2048) or ((t & 255) == 122 and cout > 64)
                                            or ((t & 255) == 123 and cin > 256) or ((t & 255) != 121 and cin > 512)):
                res.append(float("nan"))
                continue
            if (112 <= (t & 255) < 120 or (t & 255) in (111, 104, 106, 107, 109, 100, 101, 98, 96, 97)) and (k != 3 or st != 1):
                res.append(float("nan"))
                continue
            if (t & 255) in (108, 125, 126, 127) and (k != 3 or st != 2 or a.mode != "dgrad"):
                res.append(float("nan"))
                continue
            if (t & 255) == 110 and (k != 3 or st != 2 or a.mode == "dgrad"):
                res.append(float("nan"))
                continue
            if (t & 255) < 104 and (t & 255) not in (96, 97, 98, 100, 101) and (((t & 15) in (4, 5) and cout > 64) or ((t & 15) in (8, 9) and cout < 256)):
                res.append(float("nan"))
                continue
            try:
                for _ in range(2):
                    ops.conv2d(x, w, y, k, st, scale, shift, tile=t, wfrag=wf, **kw)
                torch.cuda.synchronize()
                ts = []
                for _ in range(a.reps):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(a.chain):
                        ops.conv2d(x, w, y, k, st, scale, shift, tile=t, wfrag=wf, **kw)
                    e.record()
                    torch.cuda.synchronize()
                    ts.append(s.elapsed_time(e) / a.chain)
                ts.sort()
                res.append(flops / (ts[len(ts) // 2] * 1e-3) / 1e12)
            except Exception as ex:                                    # noqa: BLE001
                res.append(float("nan"))
                print("   tile %d failed: %s" % (t, ex))
        shp = "N%d %dx%d %d->%d k%d s%d" % (N, y.H, y.W, cin, cout, k, st)
        print("%-10s %-28s " % (name, shp) + " ".join("%15.1f" % r for r in res))


if __name__ == "__main__":
    main()
