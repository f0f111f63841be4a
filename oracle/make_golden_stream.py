#!/usr/bin/env python3
"""Mint tests/golden/stream_glue.json from the REFERENCE's own streaming / evaluator code.

The reference keeps this logic inside scripts whose module-level imports (pycocotools, cv2, mmcv-based `det`, yolox.exp)
do not exist here, partly inline in `main()`.  So the function bodies are pulled out of the reference files with `ast`
— executed UNMODIFIED — inside namespaces that stub only their surroundings (dataset object, pickle files, clock, model):
  * `inference`                      sAP/streamyolo/streamyolo_det.py:62-83   (batched_nms: oracle/ref_shim restatement)
  * `main` of streamyolo_det.py      :85-229, the real-time scheduling loop, under a SIMULATED clock that the stub model
                                     advances by a scripted latency per call
  * `main` of streaming_eval.py      :46-160, the time-based pairing of results with ground-truth frames
  * `ONEX_COCOEvaluator.convert_to_coco_format`   exps/evaluators/onex_stream_evaluator.py:167-209
  * `print_stats`, `ltrb2ltwh(_)`    sAP/util/__init__.py:13-36, sAP/util/bbox.py
Test infrastructure only; runs where /root/reference exists."""
import ast
import io
import json
import os
import sys
import types
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("STREAMYOLO_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_shim"))


def pull(path, name, cls=None):
    """Source of top-level function `name` (or method `name` of class `cls`) of a reference file, verbatim."""
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    body = tree.body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    import textwrap
    return textwrap.dedent("\n".join(src.splitlines()[node.lineno - 1:node.end_lineno]))


def run(src, ns):
    exec(compile(src, "<reference>", "exec"), ns)
    return ns


class Clock:
    def __init__(self):
        self.t = 100.0
        self.tick = 1e-4          # every look at the clock costs a little simulated time

    def __call__(self):
        self.t += self.tick
        return self.t


def mint_schedule(latencies, n_frames, fps, det_stride, dynamic):
    """Reference main() of streamyolo_det.py with everything around the loop stubbed; returns the per-sequence pickle."""
    clock = Clock()
    lat = iter(latencies)
    dumped = {}

    class FakeModel:
        def cuda(self): return self
        def eval(self): return self
        def half(self): return self
        def load_state_dict(self, sd): pass
        def __call__(self, frame, buffer=None, mode=None):
            if frame is not WARM:
                clock.t += next(lat)
            return ["res%d" % frame if frame is not WARM else "warm"], "buf"

    WARM = object()

    class FakeTensor:
        def __init__(self, v): self.v = v
        def unsqueeze(self, d): return self
        def type(self, t): return self.v

    class FakeTorch:
        class cuda:
            HalfTensor = "half"
            @staticmethod
            def synchronize(): pass
        @staticmethod
        def load(*a, **k): return {"model": {}}
        @staticmethod
        def ones(*a): return FakeTensor(WARM)
        @staticmethod
        def from_numpy(f): return FakeTensor(f)
        class no_grad:
            def __enter__(self): return self
            def __exit__(self, *a): return False

    imgs = {i: {"sid": 0, "name": "f%d.jpg" % i, "width": 1920, "height": 1200} for i in range(n_frames)}
    db = types.SimpleNamespace(dataset={"sequences": ["seq0"], "seq_dirs": ["dir0"]}, imgs=imgs)
    opts = types.SimpleNamespace(out_dir="out", annot_path="a", config="c", weights="w", data_root="d", fps=fps,
                                 det_stride=det_stride, dynamic_schedule=dynamic, in_scale=0.5, overwrite=True)
    frame_no = iter(range(n_frames))
    ns = {
        "parse_args": lambda: opts, "mkdir2": lambda p: p, "COCO": lambda p: db,
        "get_exp": lambda c, n: types.SimpleNamespace(get_model=lambda: FakeModel()),
        "torch": FakeTorch, "tqdm": lambda x: x, "join": os.path.join, "isfile": lambda p: False,
        "cv2": types.SimpleNamespace(imread=lambda p: next(frame_no)), "perf_counter": clock, "np": np,
        "preproc": lambda frame, input_size: frame, "inference": lambda r: ("b%s" % r, "s", "l", None),
        "pickle": types.SimpleNamespace(dump=lambda obj, fh: dumped.__setitem__(fh, obj)),
        "open": lambda p, m="r": p, "print_stats": lambda *a, **k: None, "print": lambda *a, **k: None,
    }
    run(pull("sAP/streamyolo/streamyolo_det.py", "main"), ns)["main"]()
    seq = dumped[os.path.join("out", "seq0.pkl")]
    info = dumped[os.path.join("out", "time_info.pkl")]
    return seq, info


def mint_pairing(results, n_frames, fps, eta):
    dumped = {}
    imgs = {i: {"sid": 0, "name": "f%d.jpg" % i, "id": 1000 + i} for i in range(n_frames)}
    db = types.SimpleNamespace(dataset={"sequences": ["seq0"], "seq_dirs": ["dir0"], "categories": [{"name": "c"}]}, imgs=imgs)
    opts = types.SimpleNamespace(out_dir=None, result_dir="res", vis_dir=None, annot_path="a", no_class_mapping=True, fps=fps,
                                 eta=eta, overwrite=True, no_eval=True, eval_mask=False, data_root="d", vis_scale=1)
    ns = {"parse_args": lambda: opts, "mkdir2": lambda p: p, "COCO": lambda p: db, "tqdm": lambda x: x, "np": np,
          "join": os.path.join, "isfile": lambda p: False, "open": lambda p, m="r": p, "print": lambda *a, **k: None,
          "pickle": types.SimpleNamespace(load=lambda fh: results, dump=lambda obj, fh: dumped.__setitem__(fh, obj))}
    run(pull("sAP/util/bbox.py", "ltrb2ltwh_"), ns)
    run(pull("sAP/util/bbox.py", "ltrb2ltwh"), ns)
    run(pull("sAP/streamyolo/streaming_eval.py", "main"), ns)["main"]()
    return dumped[os.path.join("res", "results_ccf.pkl")], dumped[os.path.join("res", "eval_assoc.pkl")]


def jsonable(o):
    if isinstance(o, dict):
        return {str(k): jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.floating,)):
        return float(o)
    return o


def main():
    out = {}
    # ---- scheduling loop: three regimes (fast detector, slow detector with skipped frames, stride 2, dynamic schedule)
    cases = []
    rng = np.random.RandomState(0)
    for name, n_frames, fps, stride, dyn, lat in (
            ("fast", 12, 30.0, 1, False, [0.010] * 40),
            ("slow_mixed", 20, 30.0, 1, False, list(rng.uniform(0.02, 0.09, 40))),
            ("stride2", 16, 30.0, 2, False, [0.012] * 40),
            ("dynamic", 20, 30.0, 1, True, list(rng.uniform(0.015, 0.06, 60)))):
        seq, info = mint_schedule(lat, n_frames, fps, stride, dyn)
        cases.append({"name": name, "n_frames": n_frames, "fps": fps, "det_stride": stride, "dynamic": dyn, "latencies": lat,
                      "timestamps": seq["timestamps"], "input_fidx": seq["input_fidx"], "runtime": seq["runtime"],
                      "results_parsed": [list(r) for r in seq["results_parsed"]],
                      "n_processed": info["n_processed"], "n_total": info["n_total"], "n_small_runtime": int(info["n_small_runtime"])})
    out["schedule"] = cases
    # ---- pairing: feed the slow_mixed schedule's result (boxes made concrete) to streaming_eval.main
    pair = []
    for c in (cases[1], cases[3]):
        for eta in (0.0, -1.0):
            rp = [(rng.uniform(0, 100, (k % 3, 4)).astype(np.float32) + np.array([0, 0, 100, 100], dtype=np.float32),
                   rng.uniform(0, 1, k % 3).astype(np.float32), rng.randint(0, 8, k % 3).astype(np.int32), None)
                  for k in range(len(c["timestamps"]))]
            results = {"results_parsed": rp, "timestamps": c["timestamps"], "input_fidx": c["input_fidx"]}
            rows, assoc = mint_pairing(results, c["n_frames"], c["fps"], eta)
            pair.append({"schedule": c["name"], "eta": eta, "results_parsed": [[r[0], r[1], r[2]] for r in rp],
                         "rows": rows, "assoc": assoc})
    out["pairing"] = pair
    # ---- print_stats
    ns = run(pull("sAP/util/__init__.py", "print_stats"), {"np": np})
    buf = io.StringIO()
    with redirect_stdout(buf):
        ns["print_stats"](np.asarray(cases[1]["runtime"]), "Runtime (ms)", cvt=lambda x: 1e3 * x)
    out["print_stats"] = {"runtime": cases[1]["runtime"], "line": buf.getvalue().strip()}
    # ---- inference(): decoded rows -> boxes / scores / labels
    from yolox.utils import batched_nms, xyxy2xywh
    g = torch.Generator().manual_seed(4)
    A = 400
    dec = torch.zeros(A, 13)
    dec[:, 0:2] = torch.rand(A, 2, generator=g) * 400 + 50
    dec[:, 2:4] = torch.rand(A, 2, generator=g) * 120 + 10
    dec[:, 4] = torch.rand(A, generator=g) ** 3
    dec[:, 5:] = torch.rand(A, 8, generator=g) ** 2
    ns = run(pull("sAP/streamyolo/streamyolo_det.py", "inference"), {"torch": torch, "np": np, "batched_nms": batched_nms})
    b, s, l, _ = ns["inference"](dec.clone())
    out["inference"] = {"decoded": dec.tolist(), "bboxes": b.tolist(), "scores": s.tolist(), "labels": l.tolist()}
    # ---- convert_to_coco_format
    images = [{"fid": f} for f in (0, 1, 2, 3, 0, 1, 2, 0, 1)]
    me = types.SimpleNamespace(img_size=(600, 960), dataloader=types.SimpleNamespace(dataset=types.SimpleNamespace(
        class_ids=[1, 2, 3, 4, 6, 8, 10, 13], coco=types.SimpleNamespace(dataset={"images": images}))))
    outs = []
    for i in range(7):
        n = [2, 0, 3, 1, 2, 1, 2][i]
        o = torch.zeros(n, 7)
        o[:, 0:2] = torch.rand(n, 2, generator=g) * 300
        o[:, 2:4] = o[:, 0:2] + torch.rand(n, 2, generator=g) * 200 + 5
        o[:, 4:6] = torch.rand(n, 2, generator=g)
        o[:, 6] = torch.randint(0, 8, (n,), generator=g).float()
        outs.append(o if n else None)
    ids = torch.arange(7)
    info = (torch.full((7,), 1200), torch.full((7,), 1920))
    ns = run(pull("exps/evaluators/onex_stream_evaluator.py", "convert_to_coco_format", cls="ONEX_COCOEvaluator"), {"xyxy2xywh": xyxy2xywh})
    rows = ns["convert_to_coco_format"](me, [None if o is None else o.clone() for o in outs], info, ids)
    out["coco"] = {"outputs": [None if o is None else o.tolist() for o in outs], "images": images, "rows": rows,
                   "class_ids": me.dataloader.dataset.class_ids}
    path = os.path.join(ROOT, "tests", "golden", "stream_glue.json")
    with open(path, "w") as fh:
        json.dump(jsonable(out), fh)
    print("wrote", path, {k: len(v) for k, v in out.items()},
          [(c["name"], len(c["timestamps"]), c["input_fidx"][:8]) for c in cases], [p["assoc"] for p in pair], len(rows))


if __name__ == "__main__":
    main()
