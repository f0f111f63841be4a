"""The mixed-precision paths the reference's tools actually take:

  * training — `Trainer.train_one_iter` (exps/train_utils/double_trainer.py:95-116): fp16 inputs AND fp16 targets,
    `torch.cuda.amp.autocast`, `scaler.scale(loss).backward()`, `scaler.step(optimizer)`, `scaler.update()`;
  * eval / streaming — `model.half()` + half inputs (tools/eval.py --fp16, sAP/streamyolo/streamyolo_det.py:109).

Both are checked against the same model driven through the explicit `set_compute_dtype("fp16")` switch (whose parity
against the reference / oracle is pinned by test_model_eval.py / test_model_train.py): autocast must select the fp16
kernels, the loss scale must flow through the plan's backward (`_PlanFunction.backward` scales d_raw), the unscaled
update must equal the unscaled-path update, and fp16 targets must be accepted."""
import copy
import os

import numpy as np
import pytest
import torch

import streamyolo_amd as sy
from oracle import streamyolo_oracle as O
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels, load_bn_stats


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _nano(device, train, bn_stats=False):
    cfg = O.OracleConfig.named("nano")
    m = sy.build_model("nano")
    m.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats("nano") if bn_stats else None),
                      strict=True)
    m = m.to(device)
    return (m.train() if train else m.eval()), cfg


def test_train_one_iter_shaped_step_autocast_gradscaler(backend, monkeypatch):
    # 32 statistic replicas as in the fp32 mode: the 16-bit default of 4 (round 5) lets the arrival order of the statistics atomics
    # move a 16-bit loss by ~1e-4 between two runs of the SAME kernels — this test pins the AMP plumbing to 1e-5, not that noise
    from streamyolo_amd.train_engine import TrainPlan
    monkeypatch.setattr(TrainPlan, "STAT_COPIES", 32)
    dev = backend
    devtype = dev.type
    B, H, W = 2, 64, 96
    model, cfg = _nano(dev, True)
    model.head.use_l1 = True
    ref = copy.deepcopy(model).set_compute_dtype("fp16")
    x = synth_frames(B, H, W, seed=2).to(dev)
    lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=6, seed=3)
    # double_trainer.py:98-103: inputs and BOTH target tensors are cast to the AMP data type
    inps = x.to(torch.float16)
    targets = (lab.to(dev).to(torch.float16), sup.to(dev).to(torch.float16))
    targets[0].requires_grad = False
    targets[1].requires_grad = False
    lr = 0.01
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, nesterov=True)
    scaler = torch.amp.GradScaler(devtype, init_scale=128.0)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    with torch.autocast(devtype, dtype=torch.float16, enabled=True):
        outputs = model(inps, targets)
    loss = outputs["total_loss"]
    assert loss.dtype == torch.float32 and loss.requires_grad
    opt.zero_grad()
    scaler.scale(loss).backward()
    plan = next(iter(model._plans.plans.values()))
    assert plan.dtype == sy.ops.DT_F16, "autocast(fp16) must select the fp16 kernels"
    g_scaled = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    scaler.step(opt)
    scaler.update()
    assert scaler.get_scale() == 128.0, "the step must not have been skipped (no inf / nan gradients)"

    # the same step without AMP plumbing: explicit fp16 compute dtype, same (fp16-rounded) data, unscaled backward
    out_ref = ref(inps.float(), (targets[0].float(), targets[1].float()))
    out_ref["total_loss"].backward()
    assert abs(float(loss) - float(out_ref["total_loss"])) / abs(float(out_ref["total_loss"])) < 1e-5
    gref = dict(ref.named_parameters())
    worst = 0.0
    for n, p in model.named_parameters():
        gr = gref[n].grad
        # gradients carry the loss scale through the whole fp16 backward pass: equal to the unscaled ones up to fp16 rounding
        worst = max(worst, float((g_scaled[n].double() / 128.0 - gr.double()).norm() / gr.double().norm().clamp_min(1e-12)))
        # scaler.step unscaled the gradients it was handed; first SGD-nesterov step from zero momentum: p -= lr * (g + 0.9 g)
        want = before[n] - lr * 1.9 * (g_scaled[n] / 128.0)
        assert _rel(p.detach(), want) < 1e-6, n
    assert worst < 2e-2, "scaled vs unscaled fp16 gradients: worst per-parameter relative L2 error %.3e" % worst


def test_eval_model_half_matches_fp16_compute(backend):
    dev = backend
    model, cfg = _nano(dev, False, bn_stats=True)
    x = synth_frames(2, 64, 96, seed=2).to(dev)
    with torch.no_grad():
        want = copy.deepcopy(model).set_compute_dtype("fp16")(x)
        half = copy.deepcopy(model).half()                                  # tools/eval.py: model.half(); imgs.type(HalfTensor)
        got = half(x.half())
        assert got.dtype == torch.float16 and got.shape == want.shape
        plan = next(iter(half._plans.plans.values()))
        assert plan.dtype == sy.ops.DT_F16
        # same fp16 kernels; the only difference is parameters / BN statistics rounded to fp16 before folding
        assert _rel(got.float().cpu(), want.float().cpu()) < 3e-2
        o1, buf = half(x[:, 3:6].contiguous().half(), mode="on_pipe")        # streamyolo_det.py:109-185
        o2, _ = half(x[:, 0:3].contiguous().half(), buffer=buf, mode="on_pipe")
        assert o2.dtype == torch.float16 and _rel(o2.float().cpu(), got.float().cpu()) < 2e-3
    dets = sy.postprocess(got.float().clone(), cfg.num_classes, 0.01, 0.65)
    assert len(dets) == 2
