#!/usr/bin/env python3
"""Tune kernel variants IN the training step instead of alone.

    python tools/tune_in_situ.py [--model l] [--batch 8] [--kinds wgrad,fwd,dgrad] [--save 1]
    python tools/tune_in_situ.py --workload stream [--dtype fp16]        # the streamed frame's launch tape

ops.tuned_tile / tuned_wgrad time every kernel variant alone on an idle chip.  In the step the launches of four chains share the
CUs, and what is best alone is not what is best beside the neighbours (profiles/r04 stages r-v, aj, as: `wgrad9` on half the chip is
1.5x slower alone and 0.6 ms better in the step).  This tool measures the STEP: for every group of launches of one shape and kind
it swaps the group's variant for each alternative, re-records the launch tapes (one step) and times the step (median of --steps
replays); a swap is kept when it beats the current best by --gain ms in two measurements.  Kept choices are written into the
tuner's persisted cache (streamyolo_amd/lib/tune_cache.json, ops.save_tuned) under the same keys the per-kernel tuner uses, so
every later plan of this shape starts from them.  One pass over the weight-gradient groups of the l step takes ~1 GPU-minute.
--workload stream does the same for the streaming plan's per-layer decisions (tile, K-group tile / split-K, fused Bottleneck): the
per-kernel tuner times a layer with its weights hot in L2, in the frame every layer's weights are touched once."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="l")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--kinds", default="wgrad,fwd,dgrad")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--gain", type=float, default=0.06, help="ms a swap must win by, twice")
ap.add_argument("--min-ms", type=float, default=0.0, help="skip groups whose launches sum to less than this per step (profile estimate)")
ap.add_argument("--save", type=int, default=1)
ap.add_argument("--workload", default="train", choices=["train", "stream"])
ap.add_argument("--frames", type=int, default=60, help="stream: frames per timing")
a = ap.parse_args()
import streamyolo_amd as sy                                             # noqa: E402
from oracle import streamyolo_oracle as O                                # noqa: E402
from streamyolo_amd import ops                                           # noqa: E402
from streamyolo_amd.train_engine import TrainStep                        # noqa: E402
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels, load_bn_stats   # noqa: E402

dev = torch.device("cuda:0")
cfg = O.OracleConfig.named(a.model)
model = sy.build_model(a.model)
model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats(a.model)), strict=True)
model = model.to(dev).set_compute_dtype(a.dtype)
if a.workload == "stream":
    from tools._tune_stream import tune_stream                            # noqa: E402
    tune_stream(a, model, cfg, dev)
    sys.exit(0)
x = synth_frames(a.batch, 600, 960, seed=2).to(dev)
lab, sup = synth_labels(a.batch, 600, 960, cfg.num_classes, seed=3)
lab, sup = lab.to(dev), sup.to(dev)
st = TrainStep(model, graph=False)
for _ in range(4):
    st.step(x, (lab, sup))
torch.cuda.synchronize()
plan = st.plan


def measure():
    """ms per step: median of --steps replays (the first two steps after a change re-record the tapes)."""
    for _ in range(3):
        st.step(x, (lab, sup))
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st.step(x, (lab, sup))
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def apply(group, slot, value):
    for op in group:
        op._tiles[slot] = value
    plan.programs.clear()                                                # the next step records the tapes again


HALO = [117, 118, 107, 104, 101, 100, 98, 109]        # second / third generation of the 128-channel window tiles
WG3 = [(52, 128), (52, 96), (59, 128), (59, 96), (60, 96), (60, 128), (17, 256), (17, 512), (18, 256), (18, 512), (33, 256), (33, 512)]
WG1 = [(17, 256), (17, 512), (18, 256), (18, 512), (33, 256), (33, 512)]
convs = [op for op in plan.ops if op.kind == "conv"]
groups = {}                                                              # (slot, shape...) -> [ops]
for op in convs:
    for slot, val in list(op._tiles.items()):
        kind = "wgrad" if slot.startswith("wgrad") else ("fwd" if slot.startswith("fwd") else ("dgrad" if slot.startswith("dgrad") else None))
        if kind is None or kind not in a.kinds.split(","):
            continue
        sig = (slot, op.x.H, op.x.W, op.x.C, op.y.C, op.k, op.stride, val if kind == "wgrad" else int(val))
        groups.setdefault(sig, []).append(op)
print("%d groups over %d convolutions" % (len(groups), len(convs)))
base = measure()
print("step %.3f ms with the per-kernel tuner's choices" % base)
best = base
kept = []
for sig, group in sorted(groups.items(), key=lambda kv: -len(kv[1]) * kv[0][1] * kv[0][2] * kv[0][3] * kv[0][4] * kv[0][5] ** 2):
    slot, H, W, cin, cout, k, stride, cur = sig
    if slot.startswith("wgrad"):
        cands = [c for c in (WG3 if (k == 3 and stride == 1) else WG1) if c != tuple(cur)]
        # the per-kernel tuner's own applicability rules (ops.tuned_wgrad)
        cands = [c for c in cands if not (c[0] in (52, 59, 60) and (cin % 32 or cout % 16)) and not (c[0] == 60 and cin % 64) and not ((c[0] & 15) == 1 and c[0] < 49 and cout < 128)
                 and not ((c[0] & 15) == 2 and cout < 64)]
    elif k == 3 and stride == 1 and int(cur) in HALO and cin % 32 == 0:
        cands = [t for t in HALO if t != int(cur)]
    else:
        continue
    flops = 2.0 * len(group) * H * W * cin * cout * k * k
    if flops < 2e9 * a.batch:                                            # tiny groups cannot move the step by the threshold
        continue
    won = cur
    for c in cands:
        try:
            apply(group, slot, c)
            t1 = measure()
        except ops._lib.HipLibraryError:
            continue
        if t1 < best - a.gain:
            apply(group, slot, won)
            t0 = measure()                                               # the incumbent again, then the challenger again
            apply(group, slot, c)
            t2 = measure()
            if t2 < t0 - a.gain:
                print("  %-8s %3dx%-3d %4d->%-4d k%d s%d x%-2d  %s -> %s   %.3f / %.3f vs %.3f ms" %
                      (slot, H, W, cin, cout, k, stride, len(group), won, c, t1, t2, t0))
                won, best = c, min(t1, t2)
                continue
        apply(group, slot, won)
    if won != cur:
        kept.append((sig, won))
final = measure()
print("step %.3f ms with %d in-situ choices (was %.3f)" % (final, len(kept), base))
if a.save and kept and final < base - a.gain:
    code = ops.dtype_code(a.dtype)
    for (slot, H, W, cin, cout, k, stride, cur), won in kept:
        op = groups[(slot, H, W, cin, cout, k, stride, cur)][0]
        if slot.startswith("wgrad"):
            N = int(slot[5:])
            dy = op.y
            ops._wgrad_cache[(code, N, op.x.H, op.x.W, op.x.C, dy.H, dy.W, dy.C, k, stride, str(dev))] = tuple(won)
        else:
            N = op.x.N * (2 if slot.endswith("2") else 1)                  # "...2": the launch over both frames
            if slot.startswith("fwd"):
                ops._tile_cache[(ops.CONV_FWD, code, N, op.x.H, op.x.W, op.x.C, op.y.C, k, stride, True, str(dev))] = int(won)
            else:
                ops._tile_cache[(ops.CONV_DGRAD, code, N, op.y.H, op.y.W, op.y.C, op.x.C, k, stride, False, str(dev))] = int(won)
    ops._tune_store.dirty = True
    ops.save_tuned()
    print("saved to the tuner cache")
