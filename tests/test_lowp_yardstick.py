"""The 16-bit speed modes pinned against the REFERENCE's own low-precision behaviour, the headline
batch in the exact mode, and a multi-step sanity run.

tests/golden/lowp_yardstick.json (oracle/make_golden_lowp.py) holds, for StreamYOLO-s 2x160x256 and StreamYOLO-l 1x600x960,
how far the reference's OWN gradients move when its unmodified modules run under torch.autocast('cpu', bf16 / fp16) —
the CPU counterpart of the trainer's AMP (exps/train_utils/double_trainer.py:100-116) — per parameter and per parameter
group.  The HIP path's 16-bit step, measured the same way against the fp32 oracle on the same inputs, must stay within
2x of the reference's own error, group by group.
"""
import json
import os

import numpy as np
import pytest
import torch

import streamyolo_amd as sy
from streamyolo_amd import ops
from conftest import record_parity
from oracle import streamyolo_oracle as O
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels

NAMES = ("total_loss", "iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg")
GROUPS = ("backbone", "neck", "head")


def _group_of(name):
    if name.startswith("head."):
        return "head"
    return "backbone" if name.startswith("backbone.backbone.") else "neck"


def _yardstick(golden_dir, model):
    with open(os.path.join(golden_dir, "lowp_yardstick.json")) as f:
        y = json.load(f)
    return next(c for c in y["cases"] if c["model"] == model)


def _oracle_grads(name, B, H, W, ngt):
    cfg = O.OracleConfig.named(name)
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    x = synth_frames(B, H, W, seed=2)
    lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=ngt, seed=3)
    osd = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v.clone())
           for k, v in sd.items()}
    ref = O.forward_train(osd, x, lab, sup, cfg)
    ref["total_loss"].backward()
    return cfg, sd, x, lab, sup, ref, {k: v.grad.double() for k, v in osd.items() if v.is_floating_point() and v.requires_grad}


def _stats(model, rgrads):
    """Per parameter group: median / p90 of ||g - g32|| / ||g32||, median cosine to g32, median |log norm ratio|."""
    per = {g: [] for g in GROUPS}
    for name, p in model.named_parameters():
        g, r = p.grad.detach().cpu().double(), rgrads[name]
        rn, gn = float(r.norm()), float(g.norm())
        per[_group_of(name)].append((float((g - r).norm()) / max(rn, 1e-30), float((g * r).sum()) / max(gn * rn, 1e-30),
                                     abs(np.log(max(gn, 1e-30) / max(rn, 1e-30)))))
    out = {}
    for grp, rows in per.items():
        a = np.array(rows)
        e = np.sort(a[:, 0])
        out[grp] = {"median": float(np.median(e)), "p90": float(e[int(round(0.9 * (len(e) - 1)))]),
                    "cos_median": float(np.median(a[:, 1])), "abs_log_norm_ratio_median": float(np.median(a[:, 2]))}
    return out


def test_yardstick_fixture_is_the_reference_in_low_precision(golden_dir):
    """CPU: the committed yardstick covers both models, both 16-bit types and every parameter; bf16 is coarser than fp16
    on s (the error scales with the rounding step), and on l the reference's own 16-bit gradients are uncorrelated with its
    fp32 ones parameter by parameter (rel-L2 > 1) — the fact DESIGN.md §4 quotes."""
    for name, nkeys in (("s", None), ("l", None)):
        c = _yardstick(golden_dir, name)
        cfg = O.OracleConfig.named(name)
        nparam = sum(1 for k in O.param_shapes(cfg) if "running_" not in k and "num_batches" not in k)
        for dt in ("bf16", "fp16"):
            d = c["dtypes"][dt]
            assert len(d["per_param"]) == nparam and set(d["groups"]) == {"backbone", "neck", "head", "all"}
            assert all(np.isfinite(v) for v in d["per_param"].values())
    s, l = _yardstick(golden_dir, "s"), _yardstick(golden_dir, "l")
    assert s["dtypes"]["bf16"]["groups"]["all"]["median"] > 1.5 * s["dtypes"]["fp16"]["groups"]["all"]["median"]
    assert l["dtypes"]["bf16"]["groups"]["backbone"]["median"] > 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,H,W,ngt", [("s", 2, 160, 256, 6), ("l", 1, 600, 960, 16)])
def test_16bit_step_within_2x_of_the_references_own_autocast_error(golden_dir, name, B, H, W, ngt):
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    yard = _yardstick(golden_dir, name)
    assert yard["shape"] == [B, H, W] and yard["num_gt"] == ngt
    cfg, sd, x, lab, sup, ref, rgrads = _oracle_grads(name, B, H, W, ngt)
    want = np.array([float(ref[k]) for k in NAMES])
    for dt in ("bf16", "fp16"):
        model = sy.build_model(name)
        model.load_state_dict(sd, strict=True)
        model = model.to(dev).train().set_compute_dtype(dt)
        model.head.use_l1 = True
        out = model(x.to(dev), (lab.to(dev), sup.to(dev)))
        out["total_loss"].backward()
        got = np.array([float(out[k]) for k in NAMES])
        lerr = np.abs(got - want).max() / np.abs(want).max()
        ours, theirs = _stats(model, rgrads), yard["dtypes"][dt]["groups"]
        for grp in GROUPS:
            o, t = ours[grp], theirs[grp]
            print("%s %s %-8s rel-L2 median %.3f (reference autocast %.3f)  p90 %.3f (%.3f)  cos %.3f (%.3f)  |log norm ratio| %.3f (%.3f)"
                  % (name, dt, grp, o["median"], t["median"], o["p90"], t["p90"], o["cos_median"], t["cos_median"],
                     o["abs_log_norm_ratio_median"], t["abs_log_norm_ratio_median"]))
            assert o["median"] <= 2.0 * t["median"], (name, dt, grp, o, t)
            assert o["p90"] <= 2.0 * t["p90"], (name, dt, grp, o, t)
            # where rel-L2 saturates (deep net, batch statistics renormalise every rounding error): direction and length of
            # the gradient must still be no worse than the reference's own 16-bit run (+ slack for one sample)
            assert o["cos_median"] >= t["cos_median"] - 0.15, (name, dt, grp, o, t)
            # gradient LENGTH: where the direction is noise the norm ratio is a property of the noise realisation (the reference's
            # own value on l / neck is 0.34 in bf16 and 0.04 in fp16; ours 0.31 / 0.20) — bounded by the larger of the reference's
            # two 16-bit runs
            t_len = max(yard["dtypes"][d_]["groups"][grp]["abs_log_norm_ratio_median"] for d_ in ("bf16", "fp16"))
            assert o["abs_log_norm_ratio_median"] <= 2.0 * t_len + 0.05, (name, dt, grp, o, t_len)
        print("%s %s loss rel err %.3e (reference autocast %.3e)" % (name, dt, lerr, yard["dtypes"][dt]["loss_rel"]))
        record_parity("train_%s_b%d_%s" % (name, B, dt), loss_rel=lerr, reference_autocast_loss_rel=yard["dtypes"][dt]["loss_rel"],
                      grad_rel_l2_median_backbone=ours["backbone"]["median"], reference_autocast_grad_rel_l2_median_backbone=theirs["backbone"]["median"],
                      shape=[B, H, W], test=__name__ + "::test_16bit_step_within_2x_of_the_references_own_autocast_error")
        # 2x the spread measured on the MI355X over rounds 3-5 (s: bf16 3.0e-3, fp16 1.0e-3; l at one pair: bf16 5.6e-3 ... 1.6e-2, fp16
        # 6.5e-3 ... 1.05e-2 depending on the tuner's tile choices) or 2x the reference's own autocast error, whichever is larger
        # (VERDICT r05 "weak" #2: the former flat 5e-2 let a 3x regression pass)
        own = {"s": {"bf16": 6e-3, "fp16": 2e-3}, "l": {"bf16": 3.2e-2, "fp16": 2.1e-2}}[name][dt]
        assert lerr < max(own, 2.0 * yard["dtypes"][dt]["loss_rel"])
        del model, out
        torch.cuda.empty_cache()


@pytest.mark.gpu
def test_headline_batch_l_8x600x960_exact_mode_vs_oracle():
    """BASELINE.json configs[2]'s own batch — StreamYOLO-l, EIGHT 600x960 frame pairs: 8 images x 2 frames per statistics
    segment, a full SimOTA batch — in the exact-fp32 mode against the oracle's autograd on the host cores: the loss dict within
    1e-3 (north_star's bound), the SimOTA foreground mask equal to the oracle's up to anchors at a rounding-level tie, and —
    with identical assignments — every parameter gradient within 2e-2 of its own norm."""
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    cfg, sd, x, lab, sup, ref, rgrads = _oracle_grads("l", 8, 600, 960, 16)
    want = np.array([float(ref[k]) for k in NAMES])
    model = sy.build_model("l")
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train().set_compute_dtype("fp32")
    model.head.use_l1 = True
    # The exact mode is run-to-run deterministic since round 5 (one statistics replica row per workgroup, loss partials added in
    # index order: tests/test_model_train.py::test_exact_mode_steps_are_bit_equal), so ONE run is the test: loss dict within 1e-3,
    # the SimOTA foreground mask equal to the oracle's, every gradient within 2e-2 of its own norm (measured worst 5e-3).
    # SimOTA's dynamic-k matching is a discrete function of fp32 costs: an anchor whose cost sits within fp32 rounding of the
    # k-th best can legitimately fall on the other side in two correct fp32 implementations (the oracle sums the convolutions in
    # another order).  That case is not waved through: the looser gradient bound below applies ONLY if the per-anchor matched
    # ground truths themselves differ from the oracle's; equal assignments must meet the strict bound.
    rn = np.array([float(rgrads[n].norm()) for n, _ in model.named_parameters()])
    out = model(x.to(dev), (lab.to(dev), sup.to(dev)))
    out["total_loss"].backward()
    got = np.array([float(out[k]) for k in NAMES])
    lerr = np.abs(got - want).max() / np.abs(want).max()
    errs = sorted((float((p.grad.detach().cpu().double() - rgrads[n]).norm() / rgrads[n].norm().clamp_min(1e-30)), n)
                  for n, p in model.named_parameters())
    print("l 8x600x960 fp32: loss rel err %.3e; per-parameter rel-L2 worst %.3e (%s), median %.3e"
          % (lerr, errs[-1][0], errs[-1][1], errs[len(errs) // 2][0]))
    for e_, n_ in errs[-8:]:
        print("    %.3e  %s" % (e_, n_))
    record_parity("train_l_b8_fp32", loss_rel=lerr, grad_rel_l2_worst=errs[-1][0], grad_rel_l2_median=errs[len(errs) // 2][0],
                  reference="oracle autograd on the host cores (oracle pinned to the reference's own outputs)", test=__name__ + "::test_headline_batch_l_8x600x960_exact_mode_vs_oracle")
    assert lerr < 1e-3
    plan = next(p_ for k_, p_ in model._plans.plans.items() if str(k_[0]).startswith("train"))
    assert plan.exact_stats
    fg_ours = plan.loss_ws.fg.cpu().numpy() != 0
    fg_ref = np.asarray(ref["_fg_mask"].numpy() if torch.is_tensor(ref["_fg_mask"]) else np.stack([m.numpy() for m in ref["_fg_mask"]])) != 0
    # per-anchor matched ground truth (sy_tal_loss_assignment) against the oracle's: also sees an anchor re-matched to another
    # box with the foreground mask unchanged (the one red run of this test had 0 differing mask bits and 7.7e-2)
    mg_ours = ops.tal_assignment(plan.loss_ws)[0].cpu().numpy()
    mg_ref = ref["_matched_gt"].numpy()
    flips = int((mg_ours.reshape(mg_ref.shape) != mg_ref).sum())
    assert ((mg_ours >= 0) == fg_ours.reshape(mg_ours.shape)).all()
    vals = np.array([e for e, _ in errs])
    print("    foreground anchors: ours %d, oracle %d; anchors whose matched ground truth differs: %d (mask bits %d)"
          % (int(fg_ours.sum()), int(fg_ref.sum()), flips, int((fg_ours.reshape(fg_ref.shape) != fg_ref).sum())))
    gn = np.array([float(p.grad.detach().double().norm()) for _, p in model.named_parameters()])
    nerr = np.abs(gn - rn).max() / rn.max()
    assert np.median(vals) < 1e-2
    if flips == 0:
        assert vals[-1] < 2e-2 and nerr < 2e-3, (errs[-3:], nerr)
    else:                                                    # a proven rounding-level tie: bounded, and reported
        assert flips <= 4 and vals[-1] < 0.25 and nerr < 2e-2, (flips, errs[-3:], nerr)
    record_parity("train_l_b8_fp32", matched_gt_differ=flips, grad_norm_rel=nerr)
    # the 16-bit modes at the same batch — bf16 is the mode bench.py times, fp16 the reference's own --fp16: the loss dict, bounded at
    # 2x the upper end of what the MI355X measures (bf16 3.8e-3 ... 1.24e-2 depending on the tuner's tile choices and the split-K
    # summation order: rounds 5-6; VERDICT r05 "weak" #2 — the old bound, 5e-2, let a 3x regression pass), and the per-parameter gradient figures for the record (DESIGN.md section 4:
    # with random-init weights they are amplified rounding noise, exactly like the reference's own autocast run — the yardstick test)
    for dt16, bound in (("bf16", 2.5e-2), ("fp16", 1.5e-2)):
        model.set_compute_dtype(dt16)
        for p in model.parameters():
            p.grad = None
        model.load_state_dict(sd, strict=True)
        out = model(x.to(dev), (lab.to(dev), sup.to(dev)))
        out["total_loss"].backward()
        got = np.array([float(out[k]) for k in NAMES])
        lerr16 = np.abs(got - want).max() / np.abs(want).max()
        e16 = sorted(float((p.grad.detach().cpu().double() - rgrads[n]).norm() / rgrads[n].norm().clamp_min(1e-30))
                     for n, p in model.named_parameters())
        cos = sorted(float((p.grad.detach().cpu().double() * rgrads[n]).sum() / (p.grad.detach().cpu().double().norm() * rgrads[n].norm()).clamp_min(1e-30))
                     for n, p in model.named_parameters())
        print("l 8x600x960 %s: loss rel err %.3e; per-parameter rel-L2 median %.3e, cosine median %.3f" % (dt16, lerr16, e16[len(e16) // 2], cos[len(cos) // 2]))
        record_parity("train_l_b8_" + dt16, loss_rel=lerr16, grad_rel_l2_median=e16[len(e16) // 2], grad_rel_l2_worst=e16[-1],
                      grad_cosine_median=cos[len(cos) // 2], reference="oracle autograd on the host cores, fp32",
                      note="16-bit per-parameter gradients of a random-init l are rounding noise amplified by ~100 BatchNorms, as in the "
                           "reference's own autocast run (tests/golden/lowp_yardstick.json); the loss dict is the comparable figure",
                      test=__name__ + "::test_headline_batch_l_8x600x960_exact_mode_vs_oracle")
        assert lerr16 < bound, (dt16, lerr16)


def _run_curve(name, B, H, W, dt, steps, lr, dev):
    from streamyolo_amd.optim import FusedSGDEMA
    from streamyolo_amd.train_engine import TrainStep
    cfg = O.OracleConfig.named(name)
    model = sy.build_model(name)
    model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0), strict=True)
    model = model.to(dev).set_compute_dtype(dt)
    st = TrainStep(model)
    opt = FusedSGDEMA(model)
    x = synth_frames(B, H, W, seed=2).to(dev)
    lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=16 if H >= 200 else 6, seed=3)
    lab, sup = lab.to(dev), sup.to(dev)
    losses = []
    for _ in range(steps):
        out = st.step(x, (lab, sup))
        losses.append(out["total_loss"].clone())
        opt.step(lr)
    return np.array([float(v) for v in losses])


@pytest.mark.gpu
def test_30_sgd_steps_bf16_tracks_fp32():
    """Thirty FusedSGDEMA iterations (the trainer's SGD-nesterov + EMA step, double_trainer.py:107-119) on one fixed synthetic
    batch of the headline configuration, in the bf16 speed mode and in the exact-fp32 mode from the same initial weights:
    both losses go down and the two curves stay within 5 % of each other at every step."""
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    # basic_lr_per_img 0.001 / 64 x batch 8 (cfgs/l_*.py).  Measured (tools/sgd_curves.py, profiles/r03/b_sgd_curves.txt): two fp32
    # runs of this loop differ by up to 1.6 % from each other (SimOTA re-assignments amplify run-to-run atomics noise), bf16
    # stays within 1.6 % of fp32; at lr 1e-3: 3.1 % / 2.5 %; at lr 1e-5: 1.5 % / 2.0 %.
    lr = float(os.environ.get("STREAMYOLO_TEST_LR", "1.25e-4"))
    steps = 30
    c32 = _run_curve("l", 8, 600, 960, "fp32", steps, lr, dev)
    torch.cuda.empty_cache()
    c16 = _run_curve("l", 8, 600, 960, "bf16", steps, lr, dev)
    dev_rel = np.abs(c16 - c32) / np.abs(c32)
    print("fp32 loss: %s" % np.round(c32[[0, 4, 9, 14, 19, 24, 29]], 4))
    print("bf16 loss: %s" % np.round(c16[[0, 4, 9, 14, 19, 24, 29]], 4))
    print("max relative deviation %.3e at step %d" % (dev_rel.max(), int(dev_rel.argmax())))
    assert np.all(np.isfinite(c32)) and np.all(np.isfinite(c16))
    assert c32[-5:].mean() < c32[:5].mean() and c16[-5:].mean() < c16[:5].mean(), "loss is not decreasing"
    assert dev_rel.max() < 5e-2


def test_sgd_steps_track_across_modes_on_the_emulator(backend):
    """CPU counterpart of the test above at nano size on the SIMT emulator: the fused step + optimizer loop runs through the
    launch tapes (steps 3+ are replays that must re-read the updated weights: the loss keeps moving) and a second run from the
    same state reproduces the first steps.  (Random-init StreamYOLO + SimOTA amplifies a 1e-7 difference ~100x per SGD step at
    this learning rate — measured here: 0, 2e-6, 2e-4, 7e-2 — so only the first steps are compared tightly.)"""
    if str(backend) != "cpu":
        pytest.skip("emulator-sized variant")
    a = _run_curve("nano", 1, 32, 64, "fp32", 4, 5e-4, backend)
    b = _run_curve("nano", 1, 32, 64, "fp32", 4, 5e-4, backend)
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))
    assert len(set(np.round(a, 5))) == len(a), "the loss does not move: replayed steps do not see the updated weights"
    assert np.abs(a[:3] - b[:3]).max() / np.abs(a).max() < 1e-4
