#!/usr/bin/env python3
"""VERDICT r02 "next" #3 measured: BatchNorm.SiLU of the producer applied by the 3x3 CONSUMER in LDS (sy_conv2d in_scale /
in_shift, conv3x3_halo2_kernel NORM) against the separate sy_bn_silu_apply pass + the plain kernel, on the 3x3 stride-1 layer
shapes of StreamYOLO-l at batch 8 (frame pairs: 16 images, two segments).  Training forward: raw output + statistics.

    per layer:  apply (us) | conv (us) | apply + conv | conv with in-LDS normalisation | saved (+) / lost (-) per launch
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from streamyolo_amd import ops                                        # noqa: E402
from streamyolo_amd.ops import View                                   # noqa: E402
from streamyolo_amd.model.packing import pack_conv_weight_frag       # noqa: E402

# (name, N, H, W, C (= Cin = Cout), launches of this shape per l step whose producer is a 1x1 Bottleneck conv1 / head conv)
SHAPES = [("d2.m.c2", 16, 150, 240, 64, 3), ("d3.m.c2", 16, 75, 120, 128, 12), ("d4.m.c2", 16, 38, 60, 256, 15),
          ("d5.m.c2", 16, 19, 30, 512, 6), ("head0", 8, 75, 120, 256, 4), ("head1", 8, 38, 60, 256, 4), ("head2", 8, 19, 30, 256, 4)]


def timeit(fn, reps, chain):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(chain):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / chain * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--chain", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print("%-9s %-24s %5s %9s %9s %9s %9s %9s %8s" % ("layer", "shape", "tile", "apply us", "conv us", "sum us", "fused us", "gain us", "per step"))
    total = 0.0
    for name, N, H, W, Cc, per_step in SHAPES:
        g = torch.Generator().manual_seed(H)
        raw = View.alloc(N, H, W, Cc, a.dtype, dev)
        raw.buf.copy_(torch.randn(raw.buf.shape, generator=g).to(raw.buf.dtype))
        act = View.alloc(N, H, W, Cc, a.dtype, dev)
        y = View.alloc(N, H, W, Cc, a.dtype, dev)
        w = (torch.randn(Cc, 9 * Cc, generator=g) / (9 * Cc) ** 0.5).to(raw.buf.dtype).to(dev)
        wf = pack_conv_weight_frag(w, 3)
        scale = (torch.rand(2 * Cc, generator=g) + 0.5).to(dev)
        shift = (torch.randn(2 * Cc, generator=g) * 0.3).to(dev)
        stats = (torch.zeros(2 * 32 * Cc, device=dev), torch.zeros(2 * 32 * Cc, device=dev))
        best_plain, best_fused, t_apply = float("inf"), float("inf"), 0.0
        for tile in (117, 118):
            t_apply = timeit(lambda: ops.bn_silu_apply(raw, scale, shift, act, nseg=2), a.reps, a.chain)
            t_conv = timeit(lambda: ops.conv2d(act, w, y, 3, 1, stats=stats, tile=tile, wfrag=wf, segments=2), a.reps, a.chain)
            t_fused = timeit(lambda: ops.conv2d(raw, w, y, 3, 1, stats=stats, tile=tile, wfrag=wf, segments=2,
                                                in_affine=(scale, shift), in_segments=2), a.reps, a.chain)
            print("%-9s %-24s %5d %9.1f %9.1f %9.1f %9.1f %9.1f" % (name, "N%d %dx%d c%d" % (N, H, W, Cc), tile, t_apply, t_conv,
                                                                  t_apply + t_conv, t_fused, t_apply + t_conv - t_fused))
            best_plain, best_fused = min(best_plain, t_apply + t_conv), min(best_fused, t_fused)
        gain = best_plain - best_fused
        total += gain * per_step
        print("%-9s best apply + conv %.1f us, best fused %.1f us: %+.1f us x %d launches = %+.3f ms per step"
              % (name, best_plain, best_fused, gain, per_step, gain * per_step * 1e-3))
    print("forward total over the %d candidate launches of an l step: %+.3f ms (wgrad9 of the same layers would have to transform "
          "its operand the same way)" % (sum(s[5] for s in SHAPES), total * 1e-3))


if __name__ == "__main__":
    main()
