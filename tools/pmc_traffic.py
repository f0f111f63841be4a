#!/usr/bin/env python3
"""HBM traffic of one training / inference step from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM section):
FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC slot budget), both are reported in KiB, and on
gfx950 FETCH_SIZE counts wide coalesced reads at half their size -> doubled here (WRITE_SIZE is left as reported:
the guide calls it uncalibrated).  One step = the dispatches between the last two launches of a
once-per-step kernel (weight staging in training, Focus packing in inference).

Usage (GPU box):  python tools/pmc_traffic.py --out profiles/r01/traffic_train_l.json -- --workload train --model l
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = {"conv_igemm_kernel": "conv_igemm", "conv3x3_halo": "conv_halo", "conv1x1_tile_kernel": "conv_tile1x1",
            "conv_wgrad9_kernel": "conv_wgrad9", "conv_wgrad_tr_kernel": "conv_wgrad", "conv_wgrad_kernel": "conv_wgrad",
            "wgrad_fold_kernel": "wgrad_fold", "bn_silu_bwd_apply": "bn_silu_bwd_apply", "bn_silu_bwd_reduce": "bn_silu_bwd_reduce",
            "bn_silu_apply": "bn_silu_apply", "bn_finalize": "bn_finalize", "fold_replicas": "fold_replicas",
            "pack_weights": "pack_weights", "tal_": "tal_loss", "spp_pool": "spp_pool"}


def family(name):
    for k, v in FAMILIES.items():
        if k in name:
            return v
    return "other"


def one_pass(counter, bench_args, workdir):
    out = os.path.join(workdir, counter)
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
           sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
    subprocess.run(cmd, cwd="/tmp", check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   env=dict(os.environ, TMPDIR="/tmp"))
    rows = []
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == counter:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def step_slice(rows):
    """One step = the dispatches between the last two launches of a once-per-step kernel: the weight staging kernel
    (training) or the Focus packing kernel (inference: one launch per forward, also inside hipGraph replays)."""
    for mark in ("pack_weights_kernel", "focus_pack_kernel"):
        marks = [i for i, r in enumerate(rows) if mark in r[1]]
        marks = [m for k, m in enumerate(marks) if k == 0 or m - marks[k - 1] > 1]     # back-to-back launches = one step start
        if len(marks) >= 2:
            return rows[marks[-2]:marks[-1]], 1
    raise SystemExit("pmc_traffic: no per-step delimiter kernel found")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--commit", default=None, help="tree the counters were taken on (default: tools/.head_commit, written before gpurun)")
    ap.add_argument("bench", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    bench_args = [x for x in a.bench if x != "--"] + ["--steps", str(a.steps), "--warmup", "3", "--no-cpu-baseline", "--extras", "0"]
    commit = a.commit
    if commit is None:
        try:
            commit = open(os.path.join(ROOT, "tools", ".head_commit")).read().strip()
        except OSError:
            commit = None
    sys.path.insert(0, ROOT)
    from streamyolo_amd import _lib
    res = {"commit": commit, "kernel_source_key": _lib.kernel_source_key(), "bench_args": bench_args, "units": "bytes per step", "fetch_correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950)",
           "write_correction": "WRITE_SIZE KiB x 1024 (uncalibrated)", "families": {}}
    with tempfile.TemporaryDirectory(dir="/tmp") as wd:
        for counter, scale, key in (("FETCH_SIZE", 2048.0, "read"), ("WRITE_SIZE", 1024.0, "write")):
            rows, n = step_slice(one_pass(counter, bench_args, wd))
            res["dispatches_in_step"] = len(rows)
            for _, name, v in rows:
                fam = res["families"].setdefault(family(name), {"read": 0.0, "write": 0.0, "launches": 0})
                fam[key] += v * scale
                if key == "read":
                    fam["launches"] += 1
    res["total_read"] = sum(f["read"] for f in res["families"].values())
    res["total_write"] = sum(f["write"] for f in res["families"].values())
    mf = [res["families"].get(k, {"read": 0, "write": 0})
          for k in ("conv_igemm", "conv_halo", "conv_tile1x1", "conv_wgrad", "conv_wgrad9", "wgrad_fold")]
    res["mfma_kernels_bytes"] = sum(f["read"] + f["write"] for f in mf)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "families"}))
    for k, f in sorted(res["families"].items(), key=lambda kv: -(kv[1]["read"] + kv[1]["write"])):
        print("%-22s launches %5d  read %9.1f MB  write %9.1f MB" % (k, f["launches"], f["read"] / 1e6, f["write"] / 1e6))


if __name__ == "__main__":
    main()
