#!/usr/bin/env python3
"""bench.py — throughput of the StreamYOLO dual-frame hot path on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W [--workload train|infer] [--model l] [--batch 8]
    N>1: `python bench.py --gpus N` launches its own N ranks (one process per GPU, re-executing itself under
    torch.distributed.run on 127.0.0.1, as the reference's tools/train.py:133-141 `launch(main, num_gpu, ...)` does); started
    under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it uses the ranks it was given.  Either way
    WORLD_SIZE must equal --gpus and N devices must be visible, otherwise it exits with an error (never a 1-GPU line).

A "step" = one pass of the hot path over one batch of synthetic 600x960 frame pairs that are already
resident in HBM when the timed region starts:
    train : forward + Trend-Aware loss + backward (+ RCCL gradient all-reduce when N>1), BASELINE.json configs[2]
    infer : eval forward + decode (configs[1]-style)
One process per GPU; W untimed warm-up steps, then EXACTLY K steps between barrier+synchronize
pairs; time = MAX over ranks; rank 0 prints ONE JSON line.  `value` = frame pairs per second over
all N GPUs (weak scaling: the per-GPU batch is fixed).

Extra objects on the line:
  roofline     — dominant kernel (conv_igemm / conv_wgrad on the MFMA units): algorithmic conv FLOPs per
                 step (2*Cin*Cout*k^2*Ho*Wo, x3 for fwd+bwd) / the summed duration of
                 those kernels per step, measured live with HIP events on the launch stream.
  cpu_baseline — the CPU oracle (oracle/streamyolo_oracle.py, a torch-CPU restatement pinned to the
                 reference; kind "port") timed on this box's host cores on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}     # /opt/skills/guides/MI355X_MICROARCH.md (dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=[None, "train", "infer", "stream"])
    ap.add_argument("--model", default="l")
    ap.add_argument("--batch", type=int, default=None, help="frame pairs (frames for stream) per GPU per step; default 8, stream 1")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--graph", type=int, default=None,
                    help="infer / stream: 1 = replay the step from a hipGraph (infer default; one stream), 2 = stream only (its default): "
                         "from a launch tape over three chains (what StreamingDetector uses), 0 = Python wrappers")
    ap.add_argument("--with-optimizer", type=int, default=0,
                    help="train: also run the fused SGD-nesterov + EMA step (sy_sgd_ema_step) inside every timed step "
                         "(off by default: BASELINE.json's metric is forward + loss + backward)")
    ap.add_argument("--u8-input", type=int, default=0,
                    help="stream: feed the raw uint8 HWC camera frame (2H x 2W, e.g. 1200x1920) — exact-2x decimation, "
                         "letterbox and Focus packing run on the device (sy_frames_u8_pack) inside the timed step; "
                         "train: feed uint8 HWC frame pairs with per-pair mirror flags instead of the fp32 [B,6,H,W] tensor")
    ap.add_argument("--h2d", type=int, default=0,
                    help="stream: also copy the frame from pinned host memory inside every step (PCIe-inclusive latency; "
                         "never the headline value)")
    ap.add_argument("--train-graph", type=int, default=0, help="train: hipGraph replay instead of launch tapes (slower on ROCm 7)")
    ap.add_argument("--candidates", default="all", choices=["all", "realistic"],
                    help="stream: 'all' = random-init weights as they are — EVERY one of the 11 850 anchors passes conf 0.01, the NMS "
                         "worst case; 'realistic' = the objectness biases are shifted (calibrated on the synthetic frame, before the "
                         "timed region) so that ~1 %% of the anchors pass, as with a trained checkpoint")
    ap.add_argument("--split-k", type=int, default=1,
                    help="stream: split-K for the deep small-map 3x3 layers (16-bit modes; tuned per layer, off where it does not pay)")
    ap.add_argument("--extras", type=int, default=1,
                    help="train, 1 GPU, default path: after the timed region also time (a) the drop-in boundary "
                         "(model(x, targets)['total_loss'].backward(), the unchanged trainer's call sequence) and (b) the same "
                         "step at 4 frame pairs per GPU (BASELINE.json configs[3]'s per-GPU load = the weak-scaling denominator "
                         "of an 8-GPU global-batch-32 run) and report them beside `value`")
    ap.add_argument("--path", default="trainstep", choices=["trainstep", "dropin"],
                    help="train: 'trainstep' = streamyolo_amd.TrainStep (sync-free fast path, gradients stay in the flat arena); "
                         "'dropin' = the UNCHANGED reference trainer's call sequence, model(inps, targets)['total_loss'].backward() "
                         "(exps/train_utils/double_trainer.py:107-114), through the autograd.Function of train_forward")
    return ap.parse_args()


def effective_cores():
    """Host cores this process may really use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole socket inside a container and oversubscribes torch)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(args, workload, flops_pair):
    """Oracle (reference restatement) on the host cores: bounded sample, same shapes, batch 1."""
    from oracle import streamyolo_oracle as O
    from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels
    cores = effective_cores()
    torch.set_num_threads(cores)
    cfg = O.OracleConfig.named(args.model)
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    x = synth_frames(1, args.height, args.width, seed=2)
    lab, sup = synth_labels(1, args.height, args.width, cfg.num_classes, seed=3)

    def once():
        if workload == "train":
            s = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running_" not in k else v.clone())
                 for k, v in sd.items()}
            out = O.forward_train(s, x, lab, sup, cfg)
            out["total_loss"].backward()
        else:
            O.forward_eval(sd, x, cfg)
    once()                                        # warm-up (oneDNN primitive caches)
    t0 = time.perf_counter()
    n = 0
    while True:
        once()
        n += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or n >= 50:
            break
    out = {"value": n / el, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
           "sample": "%d x (StreamYOLO-%s %dx%d batch 1 %s, torch CPU fp32 oracle restatement of the reference)"
                     % (n, args.model, args.height, args.width,
                        "fwd+TAL loss+bwd" if workload == "train" else "eval fwd+decode"),
           "gflops_effective": n * flops_pair / el / 1e9}
    if workload == "train" and args.model == "l":
        # BASELINE.json configs[0] itself: StreamYOLO-s, ONE 600x960 frame pair, eval forward (off_pipe) + decode on the host cores
        # (cfgs/s_s50_onex_dfp_tal_flip.py:34-55 builds the model) — VERDICT r05 "missing" #3.  A few seconds of CPU time.
        cfg_s = O.OracleConfig.named("s")
        sd_s = synth_state_dict(O.param_shapes(cfg_s), seed=0)
        with torch.no_grad():
            O.forward_eval(sd_s, x, cfg_s)
            t0 = time.perf_counter()
            m = 0
            while True:
                O.forward_eval(sd_s, x, cfg_s)
                m += 1
                el_s = time.perf_counter() - t0
                if el_s >= min(5.0, args.cpu_seconds) or m >= 50:
                    break
        out["configs0"] = {"value": m / el_s, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
                           "sample": "%d x (StreamYOLO-s %dx%d batch 1 eval fwd off_pipe + decode, torch CPU fp32 oracle restatement)"
                                     % (m, args.height, args.width),
                           "gflops_effective": m * O.conv_flops_per_pair(cfg_s, args.height, args.width) / el_s / 1e9}
    return out


def pmc_traffic_full(workload, args, B):
    """HBM-side bytes per step from the committed PMC passes of this exact configuration (tools/pmc_traffic.py: rocprofv3
    FETCH_SIZE x2 (gfx950) + WRITE_SIZE in separate passes): the MFMA kernels' share (`traffic`) and the WHOLE step (`traffic_total`:
    every dispatch between two weight-staging launches — BatchNorm row passes, folds, loss, pooling included).
    PMC collection serialises kernels, so it is not re-run inside the timed bench.  A file counts only when it was measured on
    THIS tree's kernels (`kernel_source_key` = hash of csrc/, VERDICT r03 "weak" #6a): otherwise both are null and the stale source
    is named.  -> (mfma bytes or None, whole-step bytes or None, source path, commit, note)"""
    import glob
    from streamyolo_amd import _lib
    if (args.height, args.width) != (600, 960) or args.dtype != "bf16" or B != 8:
        return None, None, None, None, "no PMC measurement of this configuration"
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic_%s_%s.json" % (workload, args.model))))
    if not hits:
        return None, None, None, None, "no PMC measurement committed"
    try:
        with open(hits[-1]) as fh:
            d = json.load(fh)
        src = os.path.relpath(hits[-1], ROOT)
        total = float(d["total_read"]) + float(d["total_write"])
        if d.get("kernel_source_key") != _lib.kernel_source_key():
            return None, None, src, d.get("commit"), ("stale: measured on other kernel sources (MFMA kernels %.4g, whole step %.4g bytes there)"
                                                      % (float(d["mfma_kernels_bytes"]), total))
        return float(d["mfma_kernels_bytes"]), total, src, d.get("commit"), "measured on this tree's kernels"
    except (OSError, ValueError, KeyError):
        return None, None, None, None, "unreadable traffic file"


def in_step_utilisation(workload, args, B, flops_step):
    """What the chip does DURING the overlapped step: from the committed launch timeline of this configuration
    (tools/step_timeline.py --json, probe build: every launch's first workgroup entry / last exit on the device clock) — the share
    of the step during which at least one MFMA kernel is resident, the conv FLOPs over THAT time (not over the one-at-a-time sum
    of kernel durations, which the co-scheduled tuning has made longer than the step), and the resident-launch histogram.  Counts
    only when measured on this tree's kernel sources."""
    import glob
    from streamyolo_amd import _lib
    if (args.height, args.width) != (600, 960) or args.dtype != "bf16" or B != 8:
        return None
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "step_timeline_%s_%s.json" % (workload, args.model))))
    if not hits:
        return None
    try:
        with open(hits[-1]) as fh:
            d = json.load(fh)
        src = os.path.relpath(hits[-1], ROOT)
        out = {"source": src, "commit": d.get("commit"), "key_matches_tree": d.get("kernel_source_key") == _lib.kernel_source_key()}
        wall, res = float(d["step_ms_first_entry_to_last_exit"]), float(d["mfma_kernel_resident_ms"])
        out.update({"step_ms_probe_build": round(wall, 3), "mfma_kernel_resident_ms": round(res, 3), "mfma_kernel_resident_frac": round(res / wall, 4),
                    "achieved_tflops_over_mfma_resident_time": round(flops_step / (res * 1e-3) / 1e12, 1),
                    "frac_of_peak_over_mfma_resident_time": round(flops_step / (res * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4),
                    "resident_launches_ms": [round(v, 3) for v in d["resident_launches_ms"]]})
        return out
    except (OSError, ValueError, KeyError):
        return None


def rocprof_step(workload, args, B):
    """The committed rocprofv3 kernel trace of this configuration's taped step (tools/trace_analyze.py --json; last step between two
    weight-staging launches): per-kernel durations of the step's own launch list.  None unless it exists for this configuration."""
    import glob
    from streamyolo_amd import _lib
    if (args.height, args.width) != (600, 960) or args.dtype != "bf16" or B != 8:
        return None
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "rocprof_step_%s_%s.json" % (workload, args.model))))
    if not hits:
        return None
    try:
        with open(hits[-1]) as fh:
            d = json.load(fh)
        d["source"] = os.path.relpath(hits[-1], ROOT)
        d["key_matches_tree"] = d.get("kernel_source_key") == _lib.kernel_source_key()
        return d
    except (OSError, ValueError):
        return None


def parity_record(args, workload, B):
    """Parity of the BENCHMARKED dtype at the benchmarked configuration against the reference: the figures the
    GPU parity tests print on the MI355X, committed as profiles/r*/parity_table.json by tools/parity_table.py (the same
    comparisons as tests/test_model_train.py / test_lowp_yardstick.py, written out instead of asserted)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "parity_table.json")))
    if not hits:
        return None
    try:
        with open(hits[-1]) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        return None
    key = "%s_%s_b%d_%s" % (workload, args.model, B, args.dtype)
    out = {"source": os.path.relpath(hits[-1], ROOT), "commit": d.get("commit"), "config": key, "this_config": d.get("rows", {}).get(key),
           "north_star_bound": "1e-3 relative (fp32 mode); NMS keep indices bit-exact",
           "fp32_mode_same_config": d.get("rows", {}).get("%s_%s_b%d_fp32" % (workload, args.model, B))}
    return out


def other_configs(args):
    """The other BASELINE.json configurations and the exact-fp32 mode of the headline step, each a short run of THIS file in a child
    process (own plan, own tuner entries; the parent's GPU memory stays allocated — 288 GB), reduced to the figures that matter:
      configs[4]  StreamYOLO-l, fp16, batch-1 streaming step incl. decode + NMS, realistic candidate count  -> ms / frame
      configs[1]  StreamYOLO-s, bf16, eval forward (8 pairs and 1 pair per step)                              -> pairs / s
      exact mode  the headline training step in the fp32 mode (exact-f32 MFMA, peak 157.3 TF/s): the mode whose parity with
                  the reference is within 1e-3 (tests/test_model_train.py::test_headline_batch_l_8x600x960_exact_mode_vs_oracle)"""
    runs = {
        "configs4_stream_l_fp16": ["--workload", "stream", "--model", "l", "--dtype", "fp16", "--candidates", "realistic", "--steps", "200",
                                   "--warmup", "20", "--u8-input", "1"],
        "configs1_infer_s_bf16_b8": ["--workload", "infer", "--model", "s", "--dtype", "bf16", "--batch", "8", "--steps", "100", "--warmup", "10"],
        "configs1_infer_s_bf16_b1": ["--workload", "infer", "--model", "s", "--dtype", "bf16", "--batch", "1", "--steps", "200", "--warmup", "20"],
        "train_l_exact_fp32": ["--workload", "train", "--model", "l", "--dtype", "fp32", "--batch", "8", "--steps", "5", "--warmup", "3"],
        # the reference's own --fp16 (tools/train.py:61-67 -> double_trainer.py amp autocast = fp16): same step, fp16 storage
        "train_l_fp16": ["--workload", "train", "--model", "l", "--dtype", "fp16", "--batch", "8", "--steps", "10", "--warmup", "4"],
    }
    out = {}
    for name, extra in runs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--extras", "0",
               "--height", str(args.height), "--width", str(args.width)] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rf = d.get("roofline") or {}
            out[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "steps": d["steps"],
                         "workload": d["config"]["workload"], "nms_candidates": d["config"].get("nms_candidates"),
                         "roofline_frac": rf.get("frac"), "roofline_peak": rf.get("peak"), "whole_step_frac": rf.get("whole_step_frac")}
        except (subprocess.SubprocessError, ValueError, KeyError, IndexError, OSError) as e:
            out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if "value" in out.get("train_l_exact_fp32", {}):
        out["train_l_exact_fp32"]["parity"] = ("fp32 mode vs the reference: s eval 1.2e-6, l eval 5.2e-4, l 8-pair loss dict 1e-3 / "
                                               "gradients 1e-2 of their norm (DESIGN.md section 4; the bf16 headline mode: loss 1e-2)")
    return out


# TEST-SUITE ONLY (tests/test_bench_spawn.py): run the launcher / rank / reduction plumbing of this file against the
# SIMT-emulator build of the kernels on CPU tensors with the gloo backend.  The line it prints says so and is not a measurement.
EMU_SELFTEST = os.environ.get("STREAMYOLO_BENCH_EMU", "0") == "1"


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command line (one per GPU) and pass rank 0's
    JSON line through.  The reference launches its own workers the same way (tools/train.py:133-141)."""
    n = args.gpus
    if not EMU_SELFTEST:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible on this node" % (n, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, effective_cores() // n)))
    raise SystemExit(subprocess.call(cmd, env=env))


def pin_rank_to_cores(local, world):
    """One slice of the host cores per rank: a step is ~1500 kernel launches issued by ONE thread per rank, and eight ranks whose
    launch threads migrate over (or share) the same cores issue more slowly than one does.  The process's allowed cores (affinity
    mask, capped by the cgroup quota) are cut into `world` contiguous slices; rank `local` stays inside its own.  Returns the
    cores it was pinned to (None when there are fewer cores than ranks or the platform has no affinity call)."""
    if world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cores = sorted(os.sched_getaffinity(0))[:effective_cores()]
    per = len(cores) // world
    if per < 1:
        return None
    mine = cores[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(per, int(os.environ.get("OMP_NUM_THREADS", per)))))
    return mine


class _Mark:
    """Per-step time mark: a HIP event on the launch stream (GPU) or the host clock (emulator self-test)."""

    def __init__(self, cuda):
        self.ev = torch.cuda.Event(enable_timing=True) if cuda else None
        self.t = 0.0

    def record(self):
        if self.ev is not None:
            self.ev.record()
        else:
            self.t = time.perf_counter()

    def ms_to(self, other):
        return self.ev.elapsed_time(other.ev) if self.ev is not None else (other.t - self.t) * 1e3


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                          # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    pinned = pin_rank_to_cores(local, world)
    if world > 1 and args.path == "dropin":
        raise SystemExit("bench.py: --path dropin times the single-process boundary; the multi-GPU measurement is the default path "
                         "(the drop-in path under DDP is covered by tests/test_distributed_gloo.py)")
    dist = None
    if EMU_SELFTEST:
        from streamyolo_amd import _lib
        _lib.use_library(os.path.join(ROOT, "tests", "emu", "_build", "libstreamyolo_emu.so"))
        dev = torch.device("cpu")
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback exists)"
        if torch.cuda.device_count() < world:
            raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=dev)            # "nccl" is RCCL on ROCm
    on_gpu = dev.type == "cuda"

    import streamyolo_amd as sy
    from oracle import streamyolo_oracle as O                   # FLOP accounting + cpu_baseline leg only
    from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels, load_bn_stats

    workload = args.workload
    if workload is None:
        try:
            from streamyolo_amd import train_engine  # noqa: F401
            workload = "train"
        except ImportError:
            workload = "infer"

    cfg = O.OracleConfig.named(args.model)
    flops_fwd = O.conv_flops_per_pair(cfg, args.height, args.width)
    flops_pair = flops_fwd * (3.0 if workload == "train" else 1.0)
    if args.graph is None:
        # stream: the launch tape — its three chains (the detection levels beside the bottom-up path) are what StreamingDetector
        # runs and 5 % faster than the one-stream hipGraph; (no hipGraphs on the emulator)
        args.graph = 0 if EMU_SELFTEST else (2 if workload == "stream" else 1)
    if workload == "stream":
        flops_pair = O.conv_flops_per_pair(cfg, args.height, args.width, mode="on_pipe")

    model = sy.build_model(args.model)
    bn = load_bn_stats(args.model) if args.model in ("nano", "s", "m", "l") else None
    model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=bn), strict=True)
    model = model.to(dev).set_compute_dtype(args.dtype)
    B = args.batch if args.batch is not None else (1 if workload == "stream" else 8)
    x = synth_frames(B, args.height, args.width, seed=2 + rank).to(dev)

    if workload == "train":
        from streamyolo_amd.train_engine import TrainStep
        lab, sup = synth_labels(B, args.height, args.width, cfg.num_classes, seed=3 + rank)
        stepper = TrainStep(model, world_size=world, process_group=dist, graph=bool(args.train_graph))
        lab, sup = lab.to(dev), sup.to(dev)
        if args.u8_input:
            # the same pictures as uint8 HWC frame pairs (what a loader hands to DevicePrefetcher), every second pair
            # mirrored: mirror + letterbox + Focus packing then run on the device inside the step (sy_frames_u8_pack)
            from streamyolo_amd.data import FramePairsU8
            u8 = x.round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            flags = torch.tensor([i % 2 for i in range(B)], dtype=torch.uint8, device=dev)
            x = FramePairsU8(u8[..., 0:3].contiguous(), u8[..., 3:6].contiguous(), (args.height, args.width), mirror=flags)

        opt = [None]

        def step():
            if args.path == "dropin":
                for p_ in model.parameters():                 # optimizer.zero_grad() (set_to_none)
                    p_.grad = None
                out = model(x, (lab, sup))
                out["total_loss"].backward()
            else:
                out = stepper.step(x, (lab, sup))
            if args.with_optimizer:
                if opt[0] is None:
                    from streamyolo_amd.optim import FusedSGDEMA
                    opt[0] = FusedSGDEMA(model)
                opt[0].step(1e-5)               # tiny lr: the synthetic batch must not blow the random-init weights up
            return out
        if args.path == "dropin":
            model.train()
            model.head.use_l1 = True
            from streamyolo_amd.train_engine import get_train_plan
            profile_rows = lambda n: get_train_plan(model, x).profile(x, (lab, sup), n, detail=True)      # noqa: E731
        else:
            profile_rows = lambda n: stepper.plan.profile(x, (lab, sup), n, detail=True)                  # noqa: E731
    elif workload == "stream":
        # BASELINE.json configs[4]: on_pipe steady state, one 600x960 frame per step, decode + NMS included
        from streamyolo_amd.postprocess import postprocess_device
        model.eval()
        frame = x[:, 0:3].contiguous()
        if args.u8_input:
            from streamyolo_amd.data import FramePairsU8
            # same picture as the fp32 run (rounded to uint8), every pixel repeated 2x2 so the decimation returns it
            raw = frame.round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
            raw = raw.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2).contiguous()
            frame = FramePairsU8(raw, None, (args.height, args.width), decimate=2)
        plan = model._plans.inference(model.backbone, model.head, "on_pipe", frame, owner=model,
                                      split_k=bool(args.split_k) and args.dtype != "fp32")   # what StreamingDetector does in the 16-bit modes
        graph = None
        n_candidates = None
        if args.candidates == "realistic":
            # a trained StreamYOLO passes ~1 % of its anchors through conf 0.01; random-init weights pass all of them.  Shift the
            # three objectness biases by ONE constant found by bisection on this frame's own scores (outside the timed region).
            with torch.no_grad():
                dec = plan.run_stream(frame, first=True).clone()
                obj, cls = dec[0, :, 4].double().clamp(1e-12, 1 - 1e-12), dec[0, :, 5:].double().max(dim=1).values
                logit = torch.log(obj / (1 - obj))
                lo, hi = -40.0, 0.0
                for _ in range(40):
                    mid = 0.5 * (lo + hi)
                    frac = float(((torch.sigmoid(logit + mid) * cls) >= 0.01).double().mean())
                    lo, hi = (mid, hi) if frac < 0.01 else (lo, mid)
                for conv in model.head.obj_preds:
                    conv.bias.add_(0.5 * (lo + hi))
                dec = plan.run_stream(frame, first=True)
                n_candidates = int(((dec[0, :, 4] * dec[0, :, 5:].max(dim=1).values) >= 0.01).sum())

        dev_buf = frame.cur if args.u8_input else frame
        host_buf = dev_buf.cpu().pin_memory() if args.h2d else None

        post = lambda out: postprocess_device(out, cfg.num_classes, 0.01, 0.65)        # noqa: E731

        def eager():
            with torch.no_grad():
                if host_buf is not None:
                    dev_buf.copy_(host_buf, non_blocking=True)
                if args.graph == 2:
                    return plan.run_stream_taped(frame, post=post, check_params=False)
                out = plan.run_stream(frame)
                return post(out)
        with torch.no_grad():
            plan.run_stream(frame, first=True)
        if args.graph == 1:
            for _ in range(2):
                eager()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eager()

        def step():
            if graph is not None:
                graph.replay()
            else:
                eager()
        profile = lambda n: plan.profile(frame, n)                          # noqa: E731
    else:
        model.eval()
        plan = model._plans.inference(model.backbone, model.head, "off_pipe", x, owner=model)
        graph = None

        def eager():
            with torch.no_grad():
                return plan.run(x)
        if args.graph:   # infer: 1 (hipGraph); a tape of the off_pipe plan is not wired up
            for _ in range(2):
                eager()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                eager()

        def step():
            if graph is not None:
                graph.replay()
            else:
                eager()
        profile = lambda n: plan.profile(x, n)                  # noqa: E731

    def barrier():
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def timed(fn, warmup, steps):
        """W untimed steps, barrier + synchronize, EXACTLY `steps` steps, barrier + synchronize -> (elapsed s on this rank,
        host-side launch ms per step, sorted per-step ms from marks on the launch stream)."""
        for _ in range(warmup):
            fn()
        barrier()
        # per-step durations: one HIP event on the launch stream after every step (no synchronisation inside the timed
        # region); SURVEY.md §8(d): median and p10 / p90 over the timed steps beside the mean that `value` is computed from
        marks = [_Mark(on_gpu) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            fn()
            marks[i + 1].record()
        host = (time.perf_counter() - t0) / steps * 1e3          # launch-side time (the GPU runs behind it)
        barrier()
        el = time.perf_counter() - t0
        return el, host, sorted(marks[i].ms_to(marks[i + 1]) for i in range(steps))

    elapsed, host_ms, per_step = timed(step, args.warmup, args.steps)
    pct = lambda q: per_step[min(len(per_step) - 1, max(0, int(round(q * (len(per_step) - 1)))))]      # noqa: E731
    step_stats = {"median": round(pct(0.5), 4), "p10": round(pct(0.1), 4), "p90": round(pct(0.9), 4)}
    rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(e.item()) / args.steps * 1e3 for e in every]
        elapsed = max(float(e.item()) for e in every)            # MAX over ranks
        # every rank's core slice (first core, count; -1 = not pinned), for the line's comm record
        ct = torch.tensor([pinned[0] if pinned else -1, len(pinned) if pinned else 0], device=dev, dtype=torch.int64)
        cores_all = [torch.zeros_like(ct) for _ in range(world)]
        dist.all_gather(cores_all, ct)
        all_rank_cores = [None if int(c[0]) < 0 else [int(c[0]), int(c[0]) + int(c[1]) - 1] for c in cores_all]

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- roofline of the dominant kernels, HIP events on the launch stream (rank 0) -----------------
    def host_issue(fn, n=5):
        """Host time to ISSUE one step into empty queues (synchronise, issue, stop the clock before the GPU has finished): median
        of n.  The loop-level figure of timed() also contains the time the host spends blocked on full hardware queues once it
        runs a step ahead of the GPU (back-pressure) — at 8 pairs that was most of it (profiles/r04/c_ablate_skeleton.txt: the
        step's ~1600 launches + stream events issue in 6.6 ms)."""
        ts = []
        for _ in range(n):
            if on_gpu:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        if on_gpu:
            torch.cuda.synchronize()
        return sorted(ts)[len(ts) // 2]
    host_issue_ms = host_issue(step)
    launches = None
    if workload == "train" and args.path == "trainstep" and stepper.plan is not None and stepper.plan.programs:
        launches = sum(prog[1].size()[1] for prog in stepper.plan.programs.values())

    roofline = None
    if rank == 0 and on_gpu:
        dominant = None
        if workload == "train":
            rows = profile_rows(3)                              # [(kind, shape, launches, ms, flops)] per step
            prof = {}
            for kind, _, _, ms, _ in rows:
                prof[kind] = prof.get(kind, 0.0) + ms
            # the single largest kernel of the family: the 3x3 stride-1 forward + data-gradient layers (conv3x3_halo2_kernel,
            # 55 % of a pair's conv FLOPs) with ITS OWN fraction of the MFMA peak, next to the family's
            d_ms = sum(ms for kind, shp, _, ms, _ in rows if kind in ("conv", "dgrad") and shp.endswith("k3 s1"))
            d_fl = sum(fl for kind, shp, _, _, fl in rows if kind in ("conv", "dgrad") and shp.endswith("k3 s1"))
            w_ms = sum(ms for kind, shp, _, ms, _ in rows if kind == "wgrad")
            w_fl = sum(fl for kind, shp, _, _, fl in rows if kind == "wgrad")
            if d_ms > 0:
                dominant = {"kernel": "conv3x3_halo3_kernel / conv3x3_halo2_kernel (3x3 stride-1 forward + data gradient)", "ms_per_step": round(d_ms, 4),
                            "achieved": d_fl / (d_ms * 1e-3) / 1e12, "frac": d_fl / (d_ms * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype],
                            "second": {"kernel": "conv_wgrad9_kernel / conv_wgrad_tr_kernel (+ wgrad_fold)", "ms_per_step": round(w_ms, 4),
                                       "achieved": w_fl / (w_ms * 1e-3) / 1e12 if w_ms > 0 else 0.0,
                                       "frac": (w_fl / (w_ms * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype]) if w_ms > 0 else 0.0},
                            "hbm_bound_passes_ms": {k: round(v, 4) for k, v in prof.items() if k.startswith("bn_")}}
                rp = rocprof_step(workload, args, B)
                if rp is not None and rp.get("conv3x3_halo_ms"):
                    # the same FLOPs over the durations the rocprofv3 kernel trace of the taped step gives these kernels (every
                    # conv3x3_halo2 / halo3 launch of the last step; they also run the few stride-2 forward layers whose FLOPs are not
                    # in d_fl, so this is a lower bound of the 3x3 stride-1 rate)
                    h_ms = float(rp["conv3x3_halo_ms"])
                    dominant["in_step_ms"] = round(h_ms, 4)
                    dominant["in_step_frac"] = d_fl / (h_ms * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype]
                    dominant["in_step_source"] = {"file": rp["source"], "commit": rp.get("commit"), "key_matches_tree": rp["key_matches_tree"],
                                                  "launches_in_step": rp.get("launches_in_step"), "sum_kernel_ms": rp.get("sum_kernel_ms")}
        else:
            prof = profile(3)                                   # {kind: ms per step}
        mfma_ms = sum(v for k, v in prof.items() if k in ("conv", "pred", "dgrad", "wgrad", "conv(pred)", "dgrad(pred)", "wgrad(pred)"))
        ach = flops_pair * B / (mfma_ms * 1e-3) / 1e12 if mfma_ms > 0 else 0.0
        traffic, traffic_total, traffic_src, traffic_commit, traffic_note = pmc_traffic_full(workload, args, B)
        roofline = {"bound": "mfma", "kernel": "conv_igemm_kernel+conv3x3_halo(2,3)_kernel+conv1x1_tile_kernel" +
                    ("+conv_wgrad_tr_kernel+conv_wgrad9_kernel(+wgrad_fold)" if workload == "train" else ""),
                    "achieved": ach, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": ach / PEAK_TFLOPS[args.dtype], "traffic": traffic, "traffic_unit": "bytes/step (MFMA kernels)",
                    "traffic_total": traffic_total, "traffic_total_unit": "bytes/step (every dispatch of the step: MFMA kernels + BatchNorm row "
                                                                          "passes + folds + loss + pooling)",
                    "traffic_source": traffic_src, "traffic_commit": traffic_commit, "traffic_note": traffic_note,
                    "flops_per_step": flops_pair * B, "kernel_ms_per_step": mfma_ms,
                    "per_kind_ms": {k: round(v, 4) for k, v in prof.items()},
                    "dominant": dominant,
                    "kernel_ms_note": "kernel_ms_per_step / per_kind_ms / dominant: every launch timed ALONE (HIP events, one at a time); the "
                                      "step overlaps three chains, see in_step",
                    "in_step": in_step_utilisation(workload, args, B, flops_pair * B),
                    "whole_step_frac": flops_pair * B / (ms_per_step * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype]}

    # ---- beside the headline (1 GPU, default training path): the drop-in boundary and configs[3]'s per-GPU load ----------
    extras = None
    if workload == "train" and world == 1 and args.extras and args.path == "trainstep" and not args.train_graph:
        extras = {}
        ks, kw = max(5, args.steps // 2), 3

        def dropin_step():
            for p_ in model.parameters():                     # optimizer.zero_grad() (set_to_none)
                p_.grad = None
            out = model(x, (lab, sup))
            out["total_loss"].backward()
        el, hst, _ = timed(dropin_step, kw, ks)
        extras["dropin"] = {"ms_per_step": el / ks * 1e3, "value": B * ks / el, "steps": ks, "host_issue_ms_per_step": round(host_issue(dropin_step), 3),
                            "what": "model(x, targets)['total_loss'].backward() — the unchanged trainer's call sequence "
                                    "(exps/train_utils/double_trainer.py:107-114) through train_forward's autograd.Function"}
        if B == 8 and not args.u8_input:
            x4, lab4, sup4 = x[:4].contiguous(), lab[:4].contiguous(), sup[:4].contiguous()
            st4 = TrainStep(model, world_size=1, process_group=None, graph=False)
            el, hst, _ = timed(lambda: st4.step(x4, (lab4, sup4)), kw + 2, ks)
            extras["per_gpu_batch_4"] = {"ms_per_step": el / ks * 1e3, "value": 4 * ks / el, "steps": ks,
                                         "host_issue_ms_per_step": round(host_issue(lambda: st4.step(x4, (lab4, sup4))), 3),
                                         "what": "same step at 4 frame pairs / GPU: BASELINE.json configs[3] (global batch 32 on 8 "
                                                 "GPUs) per-GPU load, the 1-GPU denominator of its weak-scaling efficiency"}
        if not EMU_SELFTEST and args.model == "l" and B == 8 and args.dtype == "bf16" and (args.height, args.width) == (600, 960):
            extras["other_configs"] = other_configs(args)

    if workload == "stream" and world == 1 and args.extras and plan.allow_split_k:
        # the same step with every layer on its single-pass kernel: what split-K buys at batch 1 (same process, same box)
        chosen = [(op.x.H, op.x.W, op.x.C, op.y.C) + tuple(op._tiles["splitk"]) for op in plan.ops
                  if op.kind == "conv" and op._tiles.get("splitk", (1,))[0] > 1]
        extras = {"split_k_layers": [{"H": h, "W": w_, "Cin": ci, "Cout": co, "splits": s_, "tile": t_} for h, w_, ci, co, s_, t_ in chosen]}
        if chosen:
            plan.allow_split_k = False
            plan._stream_tape = None
            if args.graph == 1:
                with torch.no_grad():
                    plan.run_stream(frame, first=True)
                for _ in range(2):
                    eager()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    eager()
            ks = max(20, args.steps // 2)
            el, hst, _ = timed(step, 5, ks)
            extras["single_pass"] = {"ms_per_step": el / ks * 1e3, "steps": ks,
                                     "what": "same step, --split-k 0 (every 3x3 layer as one pass over its input channels)"}
            plan.allow_split_k = True

    comm = None
    if workload == "train" and world > 1 and stepper.plan is not None:
        pl = stepper.plan
        exposed = stepper.exposed_allreduce()
        comm = {"backend": "gloo (emulator self-test)" if EMU_SELFTEST else "nccl (RCCL over xGMI)",
                "grad_comm_dtype": "bf16" if stepper.comm_bf16 else "fp32",
                "allreduce_bytes_per_step": int(pl.arena.numel()) * (2 if stepper.comm_bf16 else 4), "buckets": len(pl.buckets),
                "bucket_bytes": [(hi - lo) * (2 if stepper.comm_bf16 else 4) for lo, hi, _ in pl.buckets],
                "bucket_order": "all-reduces start back to front (head first); bucket 0 (stem / dark2 / dark3) is final last and is the small one",
                "buckets_overlapped_with_backward": int(getattr(stepper, "last_overlapped", 0)),
                "exposed_allreduce_ms": None if exposed is None else round(exposed, 4),
                "exposed_note": "last timed step, rank 0: end of backward -> end of the gradient exchange incl. averaging (what backward did not hide)",
                "rank_cores": pinned,
                "all_rank_cores": all_rank_cores,           # [first, last] host core of every rank's slice (None: not pinned)
                "rank_ms_per_step": [round(v, 4) for v in rank_ms]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not EMU_SELFTEST:
        cpu = cpu_baseline(args, workload, flops_pair)

    if rank == 0:
        line = {
            "metric": ("frames/sec (600x960) StreamYOLO-%s on_pipe fwd + decode + NMS" % args.model) if workload == "stream"
            else "frame-pairs/sec (600x960) StreamYOLO-%s %s" % (args.model, "fwd+bwd" if workload == "train" else "fwd (eval)"),
            "value": value, "unit": "frames/s" if workload == "stream" else "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "step_ms": step_stats, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic" if not EMU_SELFTEST else "synthetic (SIMT-EMULATOR SELF-TEST on CPU: NOT a measurement)",
            "config": {"workload": "StreamYOLO-%s %dx%d %s, %d %s/GPU/step, %s"
                                   % (args.model, args.height, args.width,
                                      "training step: dual-frame forward + TAL loss + backward" if workload == "train"
                                      else ("streaming on_pipe forward + decode + NMS (conf 0.01, IoU 0.65)" if workload == "stream"
                                            else "eval forward off_pipe + decode"), B,
                                      "frame(s)" if workload == "stream" else "frame pairs",
                                      "random-init synthetic weights (utils/synth.py)"),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "hipgraph": bool(args.train_graph if workload == "train" else args.graph == 1),
                       "launch_tape": True if workload == "train" and not args.train_graph else (args.graph == 2 if workload == "stream" else False),
                       "optimizer_in_step": bool(args.with_optimizer) if workload == "train" else None,
                       "path": args.path if workload == "train" else None,
                       "u8_input": bool(args.u8_input) if workload in ("stream", "train") else None,
                       "h2d_in_step": bool(args.h2d) if workload == "stream" else None,
                       "nms_candidates": (("all %d anchors pass conf 0.01 (random-init weights: NMS worst case)" % plan.A)
                                          if args.candidates == "all" else
                                          "%d of %d anchors pass conf 0.01 (objectness bias calibrated to ~1 %%, as a trained "
                                          "checkpoint)" % (n_candidates, plan.A)) if workload == "stream" else None,
                       "launches_per_step": launches,
                       "host_issue_ms_per_step": round(host_issue_ms, 3),
                       "host_loop_ms_per_step": round(host_ms, 3),
                       "host_note": "host_issue = one step issued into empty queues (median of 5); host_loop = per step inside the "
                                    "timed loop, which includes blocking on full hardware queues while the GPU is the bottleneck"},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity_record(args, workload, B), "extras": extras, "comm": comm,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
