#!/usr/bin/env python3
"""The REAL concurrent timeline of one taped training step, from the probe build of the library (make -C streamyolo_amd/csrc probe):

    STREAMYOLO_HIP_LIB=tools/probes/_build/libstreamyolo_probe.so python tools/step_timeline.py [--model l] [--batch 8]

Every launch folds its workgroups' entry / exit stamps (the device's 100 MHz clock) into one record (sy_device.h, sy_tl_*).
rocprofv3's kernel trace serialises the streams on this stack (every kernel on one queue, union busy == sum of durations), so it
cannot say how the chains of the step overlap; these records can.  Printed: wall time of the step as the kernels see it, the time
during which 0 / 1 / 2 / 3+ launches are resident, the time an MFMA kernel is resident, per family the summed launch spans and the
time during which ONLY that family is resident, and the forward / backward split."""
import argparse
import ctypes as C
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
ap.add_argument("--dump", default="")
ap.add_argument("--json", default="", help="write the summary (step, resident-launch histogram, MFMA-resident time) with the kernel-source key: bench.py reports it as roofline.in_step")
ap.add_argument("--workload", default="train", choices=["train", "stream"], help="stream: one on_pipe frame (batch 1), decode + NMS")
ap.add_argument("--bins", type=float, default=0.0, help="print the residency profile in bins of this many ms: mean launches resident, share of the bin with 0 / 1 resident, busiest families")
ap.add_argument("--list", type=int, default=0, help="print the first N launches in start order (start, span, gap to the previous end)")
a = ap.parse_args()
import streamyolo_amd as sy                                             # noqa: E402
from oracle import streamyolo_oracle as O                                # noqa: E402
from streamyolo_amd import _lib                                          # noqa: E402
from streamyolo_amd.train_engine import TrainStep                        # noqa: E402
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels, load_bn_stats   # noqa: E402

dev = torch.device("cuda:0")
cfg = O.OracleConfig.named(a.model)
model = sy.build_model(a.model)
model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats(a.model)), strict=True)
model = model.to(dev).set_compute_dtype(a.dtype)
if a.workload == "stream":
    a.batch = 1
x = synth_frames(a.batch, 600, 960, seed=2).to(dev)
if a.workload == "train":
    lab, sup = synth_labels(a.batch, 600, 960, cfg.num_classes, seed=3)
    lab, sup = lab.to(dev), sup.to(dev)
    st = TrainStep(model, graph=False)
    run = lambda: st.step(x, (lab, sup))                               # noqa: E731
else:
    from streamyolo_amd.postprocess import postprocess_device
    model.eval()
    frame = x[:, 0:3].contiguous()
    plan = model._plans.inference(model.backbone, model.head, "on_pipe", frame, owner=model, split_k=a.dtype != "fp32")
    post = lambda out: postprocess_device(out, cfg.num_classes, 0.01, 0.65)        # noqa: E731
    with torch.no_grad():
        plan.run_stream(frame, first=True)

    def run():
        with torch.no_grad():
            return plan.run_stream_taped(frame, post=post, check_params=False)
for _ in range(4):
    run()
torch.cuda.synchronize()
lib = C.CDLL(_lib.library_path())
READERS = ["sy_probe_read_conv_extra_tl", "sy_probe_read_conv_igemm_tl", "sy_probe_read_conv_halo3_bf16_tl", "sy_probe_read_conv_halo3_f16_tl",
           "sy_probe_read_conv_halo3_f32_tl", "sy_probe_read_wgrad_tl", "sy_probe_read_train_ops_tl",
           "sy_probe_read_tal_loss_tl", "sy_probe_read_api_misc_tl", "sy_probe_read_pointwise_tl"]
ent = np.dtype([("key", "<u8"), ("t0", "<u8"), ("t1", "<u8"), ("tag", "<u4"), ("count", "<u4")])
buf = np.zeros(65536, dtype=ent)


def read_all(clear):
    out = []
    for r in READERS:
        f = getattr(lib, r)
        f.argtypes, f.restype = [C.c_void_p, C.c_int], C.c_int
        assert f(buf.ctypes.data, clear) == 0, r
        out.append(buf[buf["key"] != 0].copy())
    return np.concatenate(out)


read_all(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
rec = read_all(0)
rec = rec[rec["t1"] > 0]
rec.sort(order="t0")
NAMES = {1: "conv igemm", 2: "conv halo2 3x3", 3: "conv halo 3x3", 4: "conv 1x1 tile", 33: "dgrad igemm", 34: "dgrad halo2 3x3", 35: "dgrad halo 3x3",
         36: "dgrad 1x1 tile", 37: "dgrad s2 window", 6: "wgrad9", 7: "wgrad tr", 8: "wgrad scatter", 9: "wgrad fold", 10: "bn apply",
         11: "bn finalize", 12: "bn bwd reduce", 13: "bn bwd apply", 14: "loss", 15: "other", 16: "bottleneck fused"}
MFMA = {1, 2, 3, 4, 33, 34, 35, 36, 37, 6, 7, 8}
t0 = rec["t0"].min()
s = (rec["t0"] - t0) / 100.0
e = (rec["t1"] - t0) / 100.0
wall = e.max()
print("%s, %d %s, %s: %d launches recorded, step %.3f ms by events, %.3f ms first entry -> last exit" %
      (a.model, a.batch, "pairs" if a.workload == "train" else "frame", a.dtype, len(rec), e0.elapsed_time(e1), wall / 1e3))
# sweep
ev = sorted([(v, 1, int(t)) for v, t in zip(s, rec["tag"])] + [(v, -1, int(t)) for v, t in zip(e, rec["tag"])])
act = {}
conc = [0.0] * 6
mfma_t = only = 0.0
only_f = {}
prev = 0.0
for tm, d, tag in ev:
    n = sum(act.values())
    dt = tm - prev
    if dt > 0:
        conc[min(n, 5)] += dt
        if any(act.get(t, 0) > 0 for t in MFMA):
            mfma_t += dt
        live = [t for t, c in act.items() if c > 0]
        if len(live) == 1:
            only_f[live[0]] = only_f.get(live[0], 0.0) + dt
    act[tag] = act.get(tag, 0) + d
    prev = tm
print("resident launches   0: %.3f  1: %.3f  2: %.3f  3: %.3f  4: %.3f  5+: %.3f ms" % tuple(c / 1e3 for c in conc))
print("an MFMA kernel resident: %.3f ms of %.3f (%.0f %%)" % (mfma_t / 1e3, wall / 1e3, 100 * mfma_t / wall))
loss_rows = rec[rec["tag"] == 14]
if len(loss_rows):
    l0, l1 = (loss_rows["t0"].min() - t0) / 100.0, (loss_rows["t1"].max() - t0) / 100.0
    print("forward %.3f ms | loss %.3f ms | backward %.3f ms" % (l0 / 1e3, (l1 - l0) / 1e3, (wall - l1) / 1e3))
print("%-18s %6s %10s %10s %12s" % ("family", "n", "sum spans", "avg span", "alone"))
for tag in sorted(set(rec["tag"].tolist()), key=lambda t: -float(((e - s)[rec["tag"] == t]).sum())):
    m = rec["tag"] == tag
    sp = (e - s)[m]
    print("%-18s %6d %8.3f ms %8.1f us %9.3f ms" % (NAMES.get(tag, "tag %d" % tag), m.sum(), sp.sum() / 1e3, sp.mean(), only_f.get(tag, 0.0) / 1e3))
if a.json:
    import json
    import subprocess
    try:
        commit = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    if not commit and os.path.exists(os.path.join(ROOT, "tools", ".head_commit")):
        commit = open(os.path.join(ROOT, "tools", ".head_commit")).read().strip()
    with open(a.json, "w") as fh:
        json.dump({"workload": a.workload, "model": a.model, "batch": a.batch, "dtype": a.dtype, "kernel_source_key": _lib.kernel_source_key(),
                   "commit": commit, "launches": int(len(rec)), "step_ms_events": e0.elapsed_time(e1), "step_ms_first_entry_to_last_exit": wall / 1e3,
                   "resident_launches_ms": [c / 1e3 for c in conc], "mfma_kernel_resident_ms": mfma_t / 1e3,
                   "note": "probe build (sy_tl_* records of every launch; ~3 % slower than the product build)"}, fh)
if a.bins > 0:
    # residency profile: where in the step the chip holds fewer than two launches
    nb = int(wall / 1e3 / a.bins) + 1
    w_us = a.bins * 1e3
    res = np.zeros(nb); low = np.zeros(nb); fam = [dict() for _ in range(nb)]
    prev, act = 0.0, {}
    for tm, d, tag in ev:
        n = sum(act.values())
        t_ = prev
        while t_ < tm:                                  # spread [prev, tm) over the bins it crosses
            b = min(int(t_ / w_us), nb - 1)
            hi_ = min(tm, (b + 1) * w_us)
            dt = hi_ - t_
            res[b] += n * dt
            if n <= 1:
                low[b] += dt
            for t2, c in act.items():
                if c > 0:
                    fam[b][t2] = fam[b].get(t2, 0.0) + c * dt
            t_ = hi_ if hi_ > t_ else tm
        act[tag] = act.get(tag, 0) + d
        prev = tm
    print("residency profile, %.2f ms bins: mean launches resident | share of the bin with <= 1 resident | busiest families (launch-ms)" % a.bins)
    for b in range(nb):
        top = sorted(fam[b].items(), key=lambda kv: -kv[1])[:3]
        print("  %6.2f ms  %4.2f  %3.0f %%  %s" % (b * a.bins, res[b] / w_us, 100 * low[b] / w_us,
                                                  ", ".join("%s %.2f" % (NAMES.get(t2, "tag %d" % t2), v / 1e3) for t2, v in top)))
if a.list:
    print("launches in start order: start us, span us, workgroups, gap to the latest end so far")
    hi = 0.0
    for i in range(min(a.list, len(rec))):
        print("  %9.1f %8.1f %6d %8.1f  %s" % (s[i], e[i] - s[i], rec["count"][i], s[i] - hi, NAMES.get(int(rec["tag"][i]), "?")))
        hi = max(hi, e[i])
if a.dump:
    np.save(a.dump, np.stack([s, e, rec["tag"].astype(np.float64), rec["count"].astype(np.float64)], 1))
