"""The drop-in boundary BASELINE.json's north_star names: the reference's own cfgs/*.py build the MI355X-native model
UNCHANGED when `compat/` sits ahead of the reference checkout on PYTHONPATH.

Build-container test (needs /root/reference; the GPU box has no reference checkout, so there it is skipped): a child
interpreter gets PYTHONPATH = compat : oracle/ref_shim (yolox.exp / loguru stand-ins, test infrastructure) : reference :
repo, imports each of the five reference cfg files plus the shipped `flip` alias through `yolox.exp.get_exp`, calls
`Exp().get_model()` (cfgs/l_s50_onex_dfp_tal_filp.py:34-55), and checks
  * the classes are streamyolo_amd's, the state_dict key set is the reference's (golden key lists / oracle inventory),
    a synthetic checkpoint strict-loads, BN eps / momentum and the bias prior are what init_yolo / initialize_biases set;
  * the s cfg narrowed with `exp.merge(["width", "0.125"])` (tools/train.py's `-o` mechanism) reproduces the reference's
    golden eval tensor, losses, gradients and running statistics on the kernels (SIMT-emulator build of the same sources);
  * the reference's `exps.data` / `exps.dataset` modules stay importable next to the alias package (`__path__` extension).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("STREAMYOLO_REFERENCE", "/root/reference")

CHILD = r'''
import json, os, sys
import numpy as np
import torch
ROOT, REF = sys.argv[1], sys.argv[2]
from yolox.exp import get_exp
import streamyolo_amd as sy
from streamyolo_amd import _lib
from oracle import streamyolo_oracle as O
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels, load_bn_stats
import exps.model.yolox as alias_yolox
assert alias_yolox.YOLOX is sy.YOLOX, "exps.model.yolox did not resolve to the alias package"
import exps.data.data_augment_flip as ref_data            # the reference's own module, found through __path__
assert os.path.abspath(ref_data.__file__).startswith(os.path.abspath(REF))

res = {}
cfgs = {"s": "cfgs/s_s50_onex_dfp_tal_flip.py", "m": "cfgs/m_s50_onex_dfp_tal_flip.py",
        "l": "cfgs/l_s50_onex_dfp_tal_filp.py", "l2x": "cfgs/l_s50_twox_dfp_tal_flip.py",
        "still": "cfgs/l_s50_still_dfp_flip.py"}
paths = {k: os.path.join(REF, v) for k, v in cfgs.items()}
paths["l_alias"] = os.path.join(ROOT, "compat", "cfgs", "l_s50_onex_dfp_tal_flip.py")
for name, path in paths.items():
    exp = get_exp(path, None)
    model = exp.get_model()
    assert type(model) is sy.YOLOX and type(model.backbone) is sy.DFPPAFPN, name
    assert type(model.head) is (sy.PIPEHead if name == "still" else sy.TALHead), name
    zoo = {"s": "s", "m": "m", "l": "l", "l2x": "l2x", "still": "l", "l_alias": "l"}[name]
    depth, width, thr, val = sy.MODEL_ZOO[zoo]
    assert (exp.depth, exp.width) == (depth, width), name
    if name != "still":
        assert (model.head.ignore_thr, model.head.ignore_value, model.head.gamma) == (thr, val, 1.0), name
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            assert m.eps == 1e-3 and m.momentum == 0.03
    assert abs(float(model.head.obj_preds[0].bias[0]) + float(np.log(99.0))) < 1e-5
    ocfg = O.OracleConfig.named(zoo)
    shapes = O.param_shapes(ocfg)
    sd = model.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == shapes, name
    gold = os.path.join(ROOT, "tests", "golden", "keys_%s.txt" % zoo)
    if os.path.exists(gold):
        assert sorted(sd) == sorted(l.split()[0] for l in open(gold)), name
    model.load_state_dict(synth_state_dict(shapes, seed=0), strict=True)
    res[name] = len(sd)
assert get_exp(paths["l_alias"], None).exp_name == "l_s50_onex_dfp_tal_flip"

# ---- the s cfg, narrowed through exp.merge (tools/train.py -o): golden eval + one training step on the kernels
_lib.use_library(os.path.join(ROOT, "tests", "emu", "_build", "libstreamyolo_emu.so"))
gd = os.path.join(ROOT, "tests", "golden")
exp = get_exp(paths["s"], None)
exp.merge(["width", "0.125"])
model = exp.get_model()
ocfg = O.OracleConfig.named("nano")
model.load_state_dict(synth_state_dict(O.param_shapes(ocfg), seed=0, bn_stats=load_bn_stats("nano")), strict=True)
model.eval().set_compute_dtype("fp32")
z = np.load(os.path.join(gd, "nano_eval_2x64x96.npz"))
B, H, W = [int(v) for v in z["shape"]]
with torch.no_grad():
    out = model(synth_frames(B, H, W, seed=2))
rel = lambda a, b: float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max() /
                         torch.as_tensor(b).double().abs().max().clamp_min(1e-30))
res["eval_rel"] = rel(out, z["decoded"])

exp = get_exp(paths["s"], None)
exp.merge(["width", "0.125"])
model = exp.get_model()
model.load_state_dict(synth_state_dict(O.param_shapes(ocfg), seed=0), strict=True)
model.train().set_compute_dtype("fp32")
model.head.use_l1 = True
z = np.load(os.path.join(gd, "nano_train_2x64x96.npz"))
B, H, W = [int(v) for v in z["shape"]]
lab, sup = synth_labels(B, H, W, ocfg.num_classes, num_gt=6, seed=3)
outs = model(synth_frames(B, H, W, seed=2), (lab, sup))                     # trainer call: model(inps, targets)
outs["total_loss"].backward()
names = ("total_loss", "iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg")
got = np.array([float(outs[k]) for k in names])
res["loss_rel"] = float(np.abs(got - z["losses"]).max() / np.abs(z["losses"]).max())
res["grad_rel"] = max(rel(p.grad, z["grad:" + n]) for n, p in model.named_parameters())
sd = model.state_dict()
res["stat_rel"] = max(rel(sd[k[5:]].float(), z[k]) for k in z.files if k.startswith("stat:") and "num_batches" not in k)
print("RESULT " + json.dumps(res))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "cfgs")), reason="needs the reference checkout (build container only)")
def test_reference_cfgs_build_the_native_model_unchanged():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "streamyolo_amd", "csrc"), "-j8", "emu"], check=True)
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "compat"), os.path.join(ROOT, "oracle", "ref_shim"), REF, ROOT])
    env["STREAMYOLO_REFERENCE"] = REF
    p = subprocess.run([sys.executable, "-c", CHILD, ROOT, REF], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["s"] == 480 and res["l"] == 768 and res["l_alias"] == 768 and res["l2x"] == 768 and res["still"] == 768
    assert res["eval_rel"] < 1e-3, res
    assert res["loss_rel"] < 1e-3 and res["grad_rel"] < 2e-3 and res["stat_rel"] < 1e-3, res
