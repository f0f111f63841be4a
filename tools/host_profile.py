#!/usr/bin/env python3
"""Where does the HOST time of a taped training step go?  cProfile over a few replayed steps (l, B=8, bf16)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import streamyolo_amd as sy                                              # noqa: E402
from oracle import streamyolo_oracle as O                                # noqa: E402
from streamyolo_amd.train_engine import TrainStep                        # noqa: E402
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels, load_bn_stats  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "l"
dev = torch.device("cuda:0")
cfg = O.OracleConfig.named(name)
model = sy.build_model(name)
model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats(name)), strict=True)
model = model.to(dev).set_compute_dtype("bf16")
x = synth_frames(8, 600, 960, seed=2).to(dev)
lab, sup = synth_labels(8, 600, 960, cfg.num_classes, seed=3)
lab, sup = lab.to(dev), sup.to(dev)
st = TrainStep(model)
for _ in range(4):
    st.step(x, (lab, sup))
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    st.step(x, (lab, sup))
host = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
print("host issue %.2f ms / step, wall %.2f ms / step" % (host * 1e3, wall * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    st.step(x, (lab, sup))
pr.disable()
torch.cuda.synchronize()
ps = pstats.Stats(pr)
ps.sort_stats("tottime")
ps.print_stats(22)
for k, (key, t) in st.plan.programs.items():
    n_entries, n_launches = t.size()
    print("%s: %d tape entries, %d kernel launches, %d host snippets" % (k, n_entries, n_launches, len(t.snippets)))
