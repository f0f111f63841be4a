#!/usr/bin/env python3
"""Where does the g-space (BatchNorm-backward fusion) write-out of a data-gradient launch spend its time?  One layer, the
plain launch vs the fused one, with the fused path ablated piece by piece (tile bits 8.. = ConvArgs::ablate):
8 no global atomics, 16 no LDS fold, 32 no raw loads / arithmetic, 64 no affine fill; and 2 vs 32 replicas."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamyolo_amd import ops                                        # noqa: E402
from streamyolo_amd.ops import View                                   # noqa: E402
from streamyolo_amd.model.packing import pack_conv_weight_frag        # noqa: E402

dev = torch.device("cuda:0")


def run(N, H, W, cin, cout, k, tile, reps=7, chain=10):
    g = torch.Generator().manual_seed(1)
    dy = View.alloc(N, H, W, cin, "bf16", dev); dy.buf.copy_(torch.randn(dy.buf.shape, generator=g).to(dy.buf.dtype))
    w = (torch.randn(cout, k * k * cin, generator=g) / (cin * k * k) ** 0.5).to(dy.buf.dtype).to(dev)
    wf = pack_conv_weight_frag(w, k)
    dx = View.alloc(N, H, W, cout, "bf16", dev)
    raw = View.alloc(N, H, W, cout, "bf16", dev); raw.buf.copy_(torch.randn(raw.buf.shape, generator=g).to(raw.buf.dtype))
    aff = [torch.rand(2 * cout, device=dev) + 0.5 for _ in range(4)]
    out = []
    for label, copies, abl, fused in (("plain", 2, 0, False), ("gs c2", 2, 0, True), ("gs c32", 32, 0, True),
                                      ("gs noGatom", 2, 8, True), ("gs noLDSfold", 2, 8 + 16, True),
                                      ("gs noMath", 2, 8 + 16 + 32, True), ("gs noAff", 2, 8 + 16 + 32 + 64, True)):
        sums = torch.zeros(2 * copies * 2 * cout, device=dev)
        gs = [dict(c0=0, c1=cout, raw=raw, scale=aff[0], shift=aff[1], mean=aff[2], invstd=aff[3], sums=sums, copies=copies)] if fused else None
        t = tile + (abl << 8)

        def once():
            ops.conv2d(dy, w, dx, k, 1, mode=ops.CONV_DGRAD, tile=t, wfrag=wf, gs=gs, gs_segments=2)
        once(); once()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(chain):
                once()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / chain * 1e3)
        ts.sort()
        out.append("%s %.1f" % (label, ts[len(ts) // 2]))
    print("N%d %dx%d %d->%d k%d tile %d (us): " % (N, H, W, cin, cout, k, tile) + " | ".join(out))


for shape in ((16, 38, 60, 256, 256, 1), (16, 38, 60, 256, 256, 3), (16, 75, 120, 128, 128, 1), (16, 150, 240, 64, 64, 3)):
    for tile in (86, 102, 83):
        run(*shape, tile)
