#!/usr/bin/env python3
"""Upper bounds for fusion work: the training step with one family of launches REMOVED from its launch tapes.

    python tools/ablate_step.py [--model l] [--batch 8] [--steps 10] [--only base,bn_apply,...] [--emu]

bench.py's training configuration (same synthetic weights, inputs and labels; the input is the same every step, so a skipped
launch leaves the previous step's — correct — values in place wherever a layer owns its buffer; the plan and its buffers are
kept across configurations, only the tapes are recorded again).  The numbers are NOT results
of the framework: they say how much of the step a family of launches costs under the overlapped schedule, i.e. what fusing
that family away could buy at most.  Nothing in the package knows about this tool: the launch wrappers of streamyolo_amd.ops
are replaced from here, per configuration, before a fresh plan records its tapes."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="l")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--height", type=int, default=600)
ap.add_argument("--width", type=int, default=960)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--only", default="")
ap.add_argument("--emu", action="store_true", help="self-test on the SIMT emulator (CPU): checks the tool, measures nothing")
args = ap.parse_args()

if args.emu:
    from streamyolo_amd import _lib
    _lib.use_library(os.path.join(ROOT, "tests", "emu", "_build", "libstreamyolo_emu.so"))
import streamyolo_amd as sy                                             # noqa: E402
from oracle import streamyolo_oracle as O                                # noqa: E402  (synthetic weight shapes only)
from streamyolo_amd import ops                                           # noqa: E402
from streamyolo_amd.train_engine import TrainStep                        # noqa: E402
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels, load_bn_stats   # noqa: E402

dev = torch.device("cpu" if args.emu else "cuda:0")
cfg = O.OracleConfig.named(args.model)
model = sy.build_model(args.model)
model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats(args.model)), strict=True)
model = model.to(dev).set_compute_dtype(args.dtype)
x = synth_frames(args.batch, args.height, args.width, seed=2).to(dev)
lab, sup = synth_labels(args.batch, args.height, args.width, cfg.num_classes, seed=3)
lab, sup = lab.to(dev), sup.to(dev)

REAL = {n: getattr(ops, n) for n in ("conv2d", "conv2d_wgrad", "bn_finalize", "bn_finalize_apply", "bn_silu_apply",
                                     "bn_silu_bwd_reduce", "bn_silu_bwd_apply", "resize_nearest", "resize_nearest_bwd", "spp_pool",
                                     "spp_pool_bwd", "view_copy", "rows_add_f32", "pred_grad_fold")}
_dot = ops.View.alloc(1, 1, 1, 8, args.dtype, dev, zero=True)


def tiny(n=1):
    """n launches of a one-pixel kernel on the current launch stream: the step's launch / dependency skeleton without its work."""
    def f(*a, **k):
        for _ in range(n):
            REAL["view_copy"](_dot, _dot)
    return f


def nothing(*a, **k):
    return None


def conv_without(mode):
    def conv2d(*a, **k):
        if k.get("mode", ops.CONV_FWD) == mode:
            return None
        return REAL["conv2d"](*a, **k)
    return conv2d


CONFIGS = [
    ("base", {}),
    ("bn_all", {"bn_silu_apply": nothing, "bn_finalize_apply": nothing, "bn_silu_bwd_reduce": nothing, "bn_silu_bwd_apply": nothing,
                "bn_finalize": nothing}),
    ("bn_bwd", {"bn_silu_bwd_reduce": nothing, "bn_silu_bwd_apply": nothing}),
    ("bn_apply", {"bn_silu_apply": nothing, "bn_finalize_apply": nothing}),
    ("wgrad", {"conv2d_wgrad": nothing}),
    ("bn_bwd_reduce", {"bn_silu_bwd_reduce": nothing}),
    ("dgrad", {"conv2d": conv_without(ops.CONV_DGRAD)}),
    ("conv_fwd", {"conv2d": conv_without(ops.CONV_FWD)}),
    ("bn_finalize", {"bn_finalize": nothing}),
    ("all_bn_wgrad", {"bn_silu_apply": nothing, "bn_finalize_apply": nothing, "bn_silu_bwd_reduce": nothing, "bn_silu_bwd_apply": nothing,
                      "bn_finalize": nothing, "conv2d_wgrad": nothing}),
    # every launch of the forward / backward passes replaced by a one-pixel kernel on the same stream (the loss stays): what the
    # step costs as a STRUCTURE — ~1600 launches, their stream switches and event pairs — before any of its work
    ("skeleton", {"conv2d": tiny(), "conv2d_wgrad": tiny(2), "bn_finalize": tiny(), "bn_finalize_apply": tiny(), "bn_silu_apply": tiny(),
                  "bn_silu_bwd_reduce": tiny(), "bn_silu_bwd_apply": tiny(), "resize_nearest": tiny(), "resize_nearest_bwd": tiny(),
                  "spp_pool": tiny(), "spp_pool_bwd": tiny(), "view_copy": tiny(), "rows_add_f32": tiny(), "pred_grad_fold": tiny()}),
    ("base_again", {}),
]
only = [s for s in args.only.replace("+", ",").split(",") if s]


def sync():
    if dev.type == "cuda":
        torch.cuda.synchronize()


print("%s, %d pairs, %dx%d, %s: ms per step with a family of launches removed (%d steps each)" %
      (args.model, args.batch, args.height, args.width, args.dtype, args.steps), flush=True)
base = None
st = TrainStep(model, world_size=1, process_group=None, graph=False)
for _ in range(3):                                       # direct (tuning) step, recording step, a replay: every buffer holds its
    st.step(x, (lab, sup))                               # layer's values from here on
for name, patch in CONFIGS:
    if only and name not in only:
        continue
    for n, f in REAL.items():
        setattr(ops, n, patch.get(n, f))
    sync()
    st.plan.programs.clear()                             # the next step records its tapes again, through the patched wrappers
    for _ in range(3):
        st.step(x, (lab, sup))
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.step(x, (lab, sup))
    sync()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    base = ms if base is None else base
    print("%-14s %8.3f ms   %+7.3f ms vs base" % (name, ms, ms - base), flush=True)
for n, f in REAL.items():
    setattr(ops, n, f)
