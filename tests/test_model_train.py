"""Training-step parity: HIP forward + TAL loss + HIP backward vs the reference's golden losses,
parameter gradients and BatchNorm running statistics (SURVEY.md §8(d) parity metric)."""
import os

import numpy as np
import pytest
import torch

import streamyolo_amd as sy
from conftest import record_parity
from oracle import streamyolo_oracle as O
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels

NAMES = ("total_loss", "iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg")


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _per_param_l2(model, ref_grads):
    """(worst, name, median) of ||g - g_ref|| / ||g_ref|| over the parameters, every parameter normalised by its OWN
    reference norm (a small-gradient tensor that is 100 % wrong cannot hide behind the largest one)."""
    errs = []
    for name, p in model.named_parameters():
        r = ref_grads[name].double()
        errs.append((float((p.grad.detach().cpu().double() - r).norm() / r.norm().clamp_min(1e-30)), name))
    errs.sort()
    return errs[-1][0], errs[-1][1], errs[len(errs) // 2][0]


def _oracle_step(name, B, H, W, ngt, sd=None):
    cfg = O.OracleConfig.named(name)
    sd = synth_state_dict(O.param_shapes(cfg), seed=0) if sd is None else sd
    x = synth_frames(B, H, W, seed=2)
    lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=ngt, seed=3)
    osd = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v.clone())
           for k, v in sd.items()}
    ref = O.forward_train(osd, x, lab, sup, cfg)
    ref["total_loss"].backward()
    return ref, {k: v.grad for k, v in osd.items() if v.is_floating_point() and v.requires_grad}


def _setup(name, tag, golden_dir, device, dt, ngt=6):
    z = np.load(os.path.join(golden_dir, tag + ".npz"))
    B, H, W = [int(v) for v in z["shape"]]
    cfg = O.OracleConfig.named(name)
    model = sy.build_model(name)
    model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0), strict=True)
    model = model.to(device).train().set_compute_dtype(dt)
    model.head.use_l1 = True
    x = synth_frames(B, H, W, seed=2).to(device)
    lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=ngt, seed=3)
    return z, model, x, (lab.to(device), sup.to(device))


@pytest.mark.parametrize("dt,ltol,gtol", [("fp32", 1e-3, 2e-3), ("bf16", 1e-1, None)])
def test_train_step_nano_dropin_autograd(backend, golden_dir, dt, ltol, gtol):
    z, model, x, targets = _setup("nano", "nano_train_2x64x96", golden_dir, backend, dt)
    out = model(x, targets)                                  # the reference's call: model(inps, targets)
    assert set(NAMES) == set(out.keys())
    out["total_loss"].backward()
    got = np.array([float(out[k]) for k in NAMES])
    lerr = np.abs(got - z["losses"]).max() / np.abs(z["losses"]).max()
    assert lerr < ltol, "loss rel err %.3e (%s vs %s)" % (lerr, got, z["losses"])
    sd = model.state_dict()
    # BN running statistics after one step; shared BNs were updated twice (trap T2)
    for k in z.files:
        if k.startswith("stat:") and "num_batches" not in k:
            assert _rel(sd[k[5:]].float().cpu(), z[k]) < max(ltol, 1e-3), k
    assert int(sd["backbone.backbone.stem.conv.bn.num_batches_tracked"]) == 2
    assert int(sd["backbone.jian0.bn.num_batches_tracked"]) == 2
    assert int(sd["head.stems.2.bn.num_batches_tracked"]) == 1
    if gtol is not None:
        worst = ("", 0.0)
        for name, p in model.named_parameters():
            assert p.grad is not None, name
            r = _rel(p.grad.cpu(), z["grad:" + name])
            if r > worst[1]:
                worst = (name, r)
        assert worst[1] < gtol, "worst grad rel err %.3e at %s" % (worst[1], worst[0])


def test_train_step_fast_path_matches_dropin(backend, golden_dir):
    from streamyolo_amd.train_engine import TrainStep
    z, model, x, targets = _setup("nano", "nano_train_2x64x96", golden_dir, backend, "fp32")
    st = TrainStep(model)
    out = st.step(x, targets)
    assert abs(float(out["total_loss"]) - z["losses"][0]) / z["losses"][0] < 1e-3
    for name, p in model.named_parameters():
        assert p.grad.data_ptr() == st.plan.gview[id(p)].data_ptr()         # .grad is a view of the arena
    g = dict(model.named_parameters())["backbone.backbone.dark3.1.m.0.conv2.conv.weight"].grad
    assert _rel(g.cpu(), z["grad:backbone.backbone.dark3.1.m.0.conv2.conv.weight"]) < 2e-3


def test_launch_tape_replay_matches_direct_step(backend, golden_dir):
    """Step 1 = Python wrappers, step 2 = the same under the tape recorder, step 3+ = tape replay: identical state
    and inputs must give the same loss, gradients and running statistics (and follow changed weights)."""
    from streamyolo_amd.train_engine import TrainStep
    z, model, x, targets = _setup("nano", "nano_train_2x64x96", golden_dir, backend, "fp32")
    st = TrainStep(model, graph=False)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    res = []
    for i in range(3):
        model.load_state_dict(state0)
        out = st.step(x, targets)
        res.append((float(out["total_loss"]), st.plan.arena.clone(),
                    {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
    assert set(st.plan.programs) == {"fwd", "bwd"}
    for l, g, r in res[1:]:
        assert abs(l - res[0][0]) / abs(res[0][0]) < 1e-6
        assert _rel(g, res[0][1]) < 1e-5
        for k in r:
            assert _rel(r[k], res[0][2][k]) < 1e-6
    # the tape re-reads the parameters every step: scaled weights change the loss
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(0.5)
    out = st.step(x, targets)
    assert abs(float(out["total_loss"]) - res[0][0]) / abs(res[0][0]) > 1e-3


# Per-parameter gradient error of the 16-bit modes, measured on MI355X (profiles/r02/b_pytest_gpu_all.log):
#   s 160x256   fp32 worst 9.8e-5 / median 6.7e-5      bf16 worst 0.97 / median 0.63
#   l 600x960   fp32 worst 3.3e-3 / median 2.3e-3      bf16 worst 1.53 / median 1.26
# With RANDOM-INIT weights the network amplifies rounding noise by ~3e3 (the exact-fp32 mode itself goes from 1e-6 per
# kernel to 3e-3 end to end), so at bf16's 4e-3 rounding step the per-parameter gradients are dominated by amplified
# rounding noise, whatever the kernels do — a bound there cannot tell a bug from noise.  What CAN be pinned, and is:
#   (1) the exact-fp32 mode, per parameter, each normalised by its own norm (the same kernels, templated on the type);
#   (2) the error of the 16-bit modes is rounding noise: it SCALES with the rounding step — fp16 (2^-11) must come out
#       ~8x below bf16 (2^-8), which a defect common to the 16-bit code paths (staging, epilogues, g-space fusion) would
#       break; bounds: fp16 median <= 3x measured, bf16 / fp16 ratio inside [2.5, 25];
#   (3) per-kernel 16-bit parity on bf16-rounded operands (tests/test_kernels_*.py, 2e-2).
# (loss, worst per-parameter rel L2); fp16 loss measured 1.0e-3 ... 3.5e-3 depending on the tuner's tile choices (summation order): 2x
S_TRAIN_TOL = {"fp32": (1e-3, 2e-3), "fp16": (7e-3, None), "bf16": (3e-2, None)}
S_FP16_MEDIAN_BOUND = 0.30         # 3x the measured fp16 median (profiles/r02 pytest log)


@pytest.mark.gpu
def test_train_step_s_160x256_per_parameter_gradients():
    """Every parameter gradient of the s step against the oracle's autograd, each normalised by its own norm: 2e-3 in the
    exact-fp32 mode; in the 16-bit modes the error must behave like amplified rounding noise (see the note above)."""
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    ref, rgrads = _oracle_step("s", 2, 160, 256, 6)
    cfg = O.OracleConfig.named("s")
    want = np.array([float(ref[k]) for k in NAMES])
    lab, sup = synth_labels(2, 160, 256, cfg.num_classes, num_gt=6, seed=3)
    med = {}
    for dt in ("fp32", "fp16", "bf16"):
        model = sy.build_model("s")
        model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0), strict=True)
        model = model.to(dev).train().set_compute_dtype(dt)
        model.head.use_l1 = True
        out = model(synth_frames(2, 160, 256, seed=2).to(dev), (lab.to(dev), sup.to(dev)))
        out["total_loss"].backward()
        got = np.array([float(out[k]) for k in NAMES])
        lerr = np.abs(got - want).max() / np.abs(want).max()
        worst, wname, med[dt] = _per_param_l2(model, rgrads)
        print("s 160x256 %s: loss rel err %.3e; per-parameter gradient rel L2 error worst %.3e (%s), median %.3e"
              % (dt, lerr, worst, wname, med[dt]))
        assert lerr < S_TRAIN_TOL[dt][0]
        if S_TRAIN_TOL[dt][1] is not None:
            assert worst < S_TRAIN_TOL[dt][1]
    assert med["fp16"] < S_FP16_MEDIAN_BOUND
    assert 2.5 < med["bf16"] / med["fp16"] < 25.0, "16-bit gradient error does not scale with the rounding step"


@pytest.mark.gpu
@pytest.mark.parametrize("dt,ltol", [("fp32", 1e-3), ("bf16", 3e-2)])
def test_train_step_s_160x256(golden_dir, dt, ltol):
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    z, model, x, targets = _setup("s", "s_train_2x160x256", golden_dir, dev, dt)
    out = model(x, targets)
    out["total_loss"].backward()
    got = np.array([float(out[k]) for k in NAMES])
    lerr = np.abs(got - z["losses"]).max() / np.abs(z["losses"]).max()
    print("s train %s: loss rel err %.3e" % (dt, lerr))
    assert lerr < ltol
    if dt == "fp32":
        worst = 0.0
        for name, p in model.named_parameters():
            if "grad:" + name in z.files:
                worst = max(worst, _rel(p.grad.cpu(), z["grad:" + name]))
        norms = np.array([float(dict(model.named_parameters())[k].grad.double().norm())
                          for k in sorted(n for n, _ in model.named_parameters())])
        nerr = np.abs(norms - z["grad_norms"]).max() / np.abs(z["grad_norms"]).max()
        print("s train fp32: worst small-grad rel err %.3e, grad-norm rel err %.3e" % (worst, nerr))
        assert worst < 5e-3 and nerr < 2e-3


@pytest.mark.gpu
def test_overlapped_step_matches_single_stream(golden_dir):
    """Step 1 runs on one stream (kernel tuning); later steps put the support frame, two head levels and every
    weight gradient on a side stream, and TrainStep finally replays the whole step from a hipGraph.  Same inputs, same weights: losses, gradients and BatchNorm running
    statistics must match the single-stream pass (fp32; BN partial sums are atomically folded, hence 1e-4)."""
    from streamyolo_amd import _lib
    from streamyolo_amd.train_engine import TrainStep
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    z, model, x, targets = _setup("s", "s_train_2x160x256", golden_dir, dev, "fp32")
    st = TrainStep(model)
    state0 = {k: v.clone() for k, v in model.state_dict().items()}
    res = []
    for mode in ("tune", "serial", "overlap", "overlap", "graph", "graph", "graph"):
        model.load_state_dict(state0)
        st.use_graph = (mode == "graph")
        if mode != "tune":
            assert st.plan.side is not None and st.plan.tuned
            st.plan.force_serial = (mode == "serial")
        out = st.step(x, targets)
        assert (st.graph is not None) == (mode == "graph")
        torch.cuda.synchronize()
        res.append((mode, float(out["total_loss"]), st.plan.arena.clone(),
                    {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
    st.plan.force_serial = False
    assert st.plan.tuned
    _, l_ref, g_ref, r_ref = res[1]
    for mode, l, g, r in res[2:]:
        assert abs(l - l_ref) / abs(l_ref) < 1e-5
        assert _rel(g, g_ref) < 1e-4
        for k in r_ref:
            assert _rel(r[k], r_ref[k]) < 1e-5, k
    # and the running statistics are the reference's (two momentum updates for the shared backbone modules)
    sd = model.state_dict()
    for k in z.files:
        if k.startswith("stat:") and "running" in k:
            assert _rel(sd[k[5:]].cpu(), z[k]) < 2e-3, k


@pytest.mark.parametrize("use_l1", [True, False])
def test_pipe_head_training_matches_reference(backend, golden_dir, use_l1):
    """YOLOX(DFPPAFPN, PIPEHead) — cfgs/l_s50_still_dfp_flip.py — against the reference's own pipe_head.py outputs
    (oracle/make_golden_pipe.py): one label tensor, no trend weights, L1 term only when `use_l1`."""
    z = np.load(os.path.join(golden_dir, "nano_pipe_train_2x64x96.npz"))
    B, H, W = [int(v) for v in z["shape"]]
    cfg = O.OracleConfig.named("nano")
    model = sy.build_model("nano", head="pipe")
    model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0), strict=True)
    model = model.to(backend).train().set_compute_dtype("fp32")
    model.head.use_l1 = use_l1
    x = synth_frames(B, H, W, seed=2).to(backend)
    lab, _ = synth_labels(B, H, W, cfg.num_classes, num_gt=6, seed=3)
    out = model(x, lab.to(backend))
    out["total_loss"].backward()
    tag = "l1" if use_l1 else "nol1"
    got = np.array([float(out[k]) for k in NAMES])
    assert np.abs(got - z["losses_" + tag]).max() / np.abs(z["losses_" + tag]).max() < 1e-3
    named = dict(model.named_parameters())
    for k in z.files:
        if k.startswith("grad_%s:" % tag):
            assert _rel(named[k.split(":", 1)[1]].grad.cpu(), z[k]) < 2e-3, k
    norms = np.array([float(named[k].grad.double().norm()) for k in sorted(named)])
    assert np.abs(norms - z["grad_norms_" + tag]).max() / np.abs(z["grad_norms_" + tag]).max() < 2e-3


# worst per-parameter relative L2 gradient error (each parameter normalised by its own norm); measured on MI355X: fp32 worst
# 3.3e-3 / median 2.3e-3 (bound 3x); the 16-bit modes are reported and checked for rounding-step scaling (note above S_TRAIN_TOL)
L_GRAD_TOL = {"fp32": 1e-2, "fp16": None, "bf16": None}


@pytest.mark.gpu
def test_train_step_l_600x960_full_size_vs_oracle():
    """BASELINE.json's configuration itself (StreamYOLO-l, 600x960; one frame pair so the CPU oracle finishes in
    seconds): the six loss-dict entries and every parameter-gradient norm of the HIP step against the oracle's autograd
    — 1e-3 in the exact-fp32 mode (north_star's bound), and the loss within 3.2e-2 (bf16, the speed mode bench.py times) / 2.1e-2 (fp16): twice the upper end of the measured spread."""
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    cfg = O.OracleConfig.named("l")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    x = synth_frames(1, 600, 960, seed=2)
    lab, sup = synth_labels(1, 600, 960, cfg.num_classes, num_gt=16, seed=3)
    osd = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v.clone())
           for k, v in sd.items()}
    ref = O.forward_train(osd, x, lab, sup, cfg)
    ref["total_loss"].backward()
    want = np.array([float(ref[k]) for k in NAMES])
    rgrads = {k: v.grad for k, v in osd.items() if v.is_floating_point() and v.requires_grad}
    # The same step through the oracle in FLOAT64 (VERDICT r05 "weak" #1 / item 5): how far the fp32 ORACLE itself is from exact
    # arithmetic, parameter by parameter — the yardstick for the exact-fp32 mode's own gradient error below.  (l at this size
    # amplifies one fp32 rounding ~3e4 x through ~100 BatchNorms; two correct fp32 implementations that add in different orders
    # differ by that much.)
    osd64 = {k: (v.clone().double().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else
                 (v.clone().double() if v.is_floating_point() else v.clone())) for k, v in sd.items()}
    ref64 = O.forward_train(osd64, x.double(), lab.double(), sup.double(), cfg)
    ref64["total_loss"].backward()
    g64 = {k: v.grad for k, v in osd64.items() if v.is_floating_point() and v.requires_grad}
    o_err = sorted(float((rgrads[k].double() - g64[k]).norm() / g64[k].norm().clamp_min(1e-300)) for k in g64)
    loss64 = np.array([float(ref64[k]) for k in NAMES])
    print("l 600x960 oracle fp32 vs oracle fp64: loss rel %.3e; per-parameter gradient rel L2 worst %.3e, median %.3e"
          % (np.abs(want - loss64).max() / np.abs(loss64).max(), o_err[-1], o_err[len(o_err) // 2]))
    med = {}
    for dt, ltol, gtol in (("fp32", 1e-3, L_GRAD_TOL["fp32"]), ("fp16", 2.1e-2, None), ("bf16", 3.2e-2, None)):   # measured fp16 6.5e-3 .. 1.05e-2, bf16 5.6e-3 .. 1.6e-2 (tuner-dependent): bounds = 2x the upper end
        model = sy.build_model("l")
        model.load_state_dict(sd, strict=True)
        model = model.to(dev).train().set_compute_dtype(dt)
        model.head.use_l1 = True
        out = model(x.to(dev), (lab.to(dev), sup.to(dev)))
        out["total_loss"].backward()
        got = np.array([float(out[k]) for k in NAMES])
        lerr = np.abs(got - want).max() / np.abs(want).max()
        print("l 600x960 %s: loss rel err %.3e" % (dt, lerr))
        record_parity("train_l_b1_" + dt, loss_rel=lerr, reference="oracle autograd, fp32", test=__name__ + "::test_train_step_l_600x960_full_size_vs_oracle")
        assert lerr < ltol
        worst, wname, med[dt] = _per_param_l2(model, rgrads)
        print("l 600x960 %s: per-parameter gradient rel L2 error worst %.3e (%s), median %.3e" % (dt, worst, wname, med[dt]))
        record_parity("train_l_b1_" + dt, grad_rel_l2_worst=worst, grad_rel_l2_median=med[dt])
        assert gtol is None or worst < gtol
        if dt == "fp32":
            # ours against EXACT arithmetic, next to the fp32 oracle against exact arithmetic: the exact mode must be no further
            # from the truth than a few times the oracle's own fp32 rounding distance (it is a different summation order of the
            # same fp32 products, not a coarser computation)
            w64, n64, m64 = _per_param_l2(model, g64)
            print("l 600x960 fp32 vs oracle fp64: per-parameter gradient rel L2 worst %.3e (%s), median %.3e  [fp32 oracle vs fp64: worst %.3e, median %.3e]"
                  % (w64, n64, m64, o_err[-1], o_err[len(o_err) // 2]))
            record_parity("train_l_b1_fp32", grad_rel_l2_worst_vs_fp64_oracle=w64, grad_rel_l2_median_vs_fp64_oracle=m64,
                          fp32_oracle_vs_fp64_oracle_worst=o_err[-1], fp32_oracle_vs_fp64_oracle_median=o_err[len(o_err) // 2],
                          loss_rel_vs_fp64_oracle=float(np.abs(got - loss64).max() / np.abs(loss64).max()))
            assert m64 < 4.0 * max(o_err[len(o_err) // 2], 1e-5) and w64 < 4.0 * max(o_err[-1], 1e-4), (w64, m64, o_err[-1])
            gn, rn = [], []
            for name, p in model.named_parameters():
                gn.append(float(p.grad.double().norm())); rn.append(float(osd[name].grad.double().norm()))
            gn, rn = np.array(gn), np.array(rn)
            nerr = np.abs(gn - rn).max() / rn.max()
            print("l 600x960 fp32: grad-norm rel err %.3e over %d parameters" % (nerr, len(gn)))
            assert nerr < 2e-3
        del model, out
        torch.cuda.empty_cache()
    # l at this size amplifies rounding ~3e4x (fp32 mode: 6e-8 per operation -> 2e-3 here), so BOTH 16-bit modes sit at the
    # level of uncorrelated vectors (~1.1-1.5, measured fp16 1.12 / bf16 1.29): reported, finite, and no worse than that.
    # The rounding-step scaling that separates noise from defects is asserted on the s model above.
    assert med["fp16"] < 2.0 and med["bf16"] < 2.0


@pytest.mark.gpu
def test_train_step_m_vs_reference_golden(golden_dir):
    """StreamYOLO-m training step (widths 48..768: Cin / Cout that are not multiples of the 64-byte K slab go through the
    generic loader, the scatter wgrad and ragged channel tiles) against the reference's own losses, gradients and running
    statistics; fp32 mode 1e-3 / 2e-3, bf16 speed mode loss bound."""
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    for dt, ltol in (("fp32", 1e-3), ("bf16", 3e-2)):
        z, model, x, targets = _setup("m", "m_train_2x128x192", golden_dir, dev, dt)
        out = model(x, targets)
        out["total_loss"].backward()
        got = np.array([float(out[k]) for k in NAMES])
        lerr = np.abs(got - z["losses"]).max() / np.abs(z["losses"]).max()
        print("m train %s: loss rel err %.3e" % (dt, lerr))
        assert lerr < ltol
        if dt == "fp32":
            named = dict(model.named_parameters())
            worst = max(_rel(named[k[5:]].grad.cpu(), z[k]) for k in z.files if k.startswith("grad:"))
            norms = np.array([float(named[k].grad.double().norm()) for k in sorted(named)])
            nerr = np.abs(norms - z["grad_norms"]).max() / np.abs(z["grad_norms"]).max()
            sd = model.state_dict()
            serr = max(_rel(sd[k[5:]].float().cpu(), z[k]) for k in z.files if k.startswith("stat:"))
            print("m train fp32: worst small-grad rel err %.3e, grad-norm rel err %.3e, running-stat rel err %.3e" % (worst, nerr, serr))
            assert worst < 5e-3 and nerr < 2e-3 and serr < 2e-3


def test_submodule_training_calls_compose_like_the_reference(backend, golden_dir):
    """The reference's YOLOX.forward in training mode is `fpn_outs = self.backbone(x)` then `self.head(fpn_outs, targets, x)`
    (exps/model/yolox.py:32-38).  Calling the two sub-modules separately in training mode must give the reference's golden
    losses, parameter gradients (autograd carries d(features) from the head plan into the backbone plan) and BatchNorm
    running statistics — as the fused single-plan YOLOX.forward does."""
    z, model, x, targets = _setup("nano", "nano_train_2x64x96", golden_dir, backend, "fp32")
    fpn_outs = model.backbone(x)                                             # training-mode DFPPAFPN.forward
    assert len(fpn_outs) == 3 and all(f.requires_grad for f in fpn_outs)
    loss, iou_loss, conf_loss, cls_loss, l1_loss, num_fg = model.head(fpn_outs, targets, x)
    loss.backward()
    got = np.array([float(v) for v in (loss, iou_loss, l1_loss, conf_loss, cls_loss, num_fg)])
    lerr = np.abs(got - z["losses"]).max() / np.abs(z["losses"]).max()
    assert lerr < 1e-3, "loss rel err %.3e (%s vs %s)" % (lerr, got, z["losses"])
    worst = ("", 0.0)
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        r = _rel(p.grad.cpu(), z["grad:" + name])
        if r > worst[1]:
            worst = (name, r)
    assert worst[1] < 2e-3, "worst grad rel err %.3e at %s" % (worst[1], worst[0])
    sd = model.state_dict()
    for k in z.files:
        if k.startswith("stat:") and "num_batches" not in k:
            assert _rel(sd[k[5:]].float().cpu(), z[k]) < 1e-3, k
    assert int(sd["backbone.backbone.stem.conv.bn.num_batches_tracked"]) == 2
    assert int(sd["head.stems.2.bn.num_batches_tracked"]) == 1


def test_submodule_training_steps_follow_changing_inputs(backend):
    """the stand-alone DFPPAFPN training plan must seed its backward with THIS call's feature gradients on
    every step — steps 3+ replay a launch tape, whose recorded torch snippets must not hold on to the recording call's
    tensors.  Three steps on different inputs: split path (backbone -> head through autograd) vs the fused YOLOX.forward."""
    cfg = O.OracleConfig.named("nano")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)

    def fresh():
        m = sy.build_model("nano")
        m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        m = m.to(backend).train().set_compute_dtype("fp32")
        m.head.use_l1 = True
        return m
    split, fused = fresh(), fresh()
    for step in range(3):                                        # call 3 is the first tape REPLAY of the split path
        x = synth_frames(2, 32, 64, seed=20 + step).to(backend)
        lab, sup = synth_labels(2, 32, 64, cfg.num_classes, num_gt=4, seed=30 + step)
        targets = (lab.to(backend), sup.to(backend))
        for m in (split, fused):
            for p in m.parameters():
                p.grad = None
            m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)      # same state every step
        loss = split.head(split.backbone(x), targets, x)[0]
        loss.backward()
        out = fused(x, targets)
        out["total_loss"].backward()
        assert abs(float(loss) - float(out["total_loss"])) / abs(float(out["total_loss"])) < 1e-5, step
        ref = {n: p.grad.detach().cpu() for n, p in fused.named_parameters()}
        worst, name, _ = _per_param_l2(split, ref)
        assert worst < 1e-3, "step %d: %s off by %.3e (split vs fused plan)" % (step, name, worst)


def test_frames_as_stream_parallel_chains_match_the_paired_launches(backend, golden_dir, monkeypatch):
    """The two frames of a pair as separate chains (forward: half-size launches on two streams; backward: BatchNorm backward +
    data gradient per frame on streams 0 / 2, ONE weight gradient per layer behind both) against one launch per layer over
    both frames: same loss, gradients and running statistics, through direct steps, the recorded step and tape replays."""
    from streamyolo_amd import train_engine
    from streamyolo_amd.train_engine import TrainStep
    res = {}
    cfg = O.OracleConfig.named("nano")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    # emulator: a tiny frame; GPU: a size whose deepest maps still hold a few dozen samples per channel (at 32x64 the stride-32
    # BatchNorms normalise over 4 values and amplify atomics-order noise to 1e-3)
    Hh, Ww = (32, 64) if str(backend) == "cpu" else (128, 192)
    x = synth_frames(2, Hh, Ww, seed=2).to(backend)
    lab, sup = synth_labels(2, Hh, Ww, cfg.num_classes, num_gt=4, seed=3)
    targets = (lab.to(backend), sup.to(backend))
    for mode, fwd, bwd in (("paired", False, "0"), ("chains", True, "1")):
        monkeypatch.setattr(train_engine, "FWD_SPLIT_FRAMES", fwd)
        monkeypatch.setattr(train_engine, "BWD_SPLIT_FRAMES", bwd)
        model = sy.build_model("nano")
        model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        model = model.to(backend).train().set_compute_dtype("fp32")
        model.head.use_l1 = True
        st = TrainStep(model, graph=False)
        assert st._ensure(x).bwd_split == (bwd == "1")
        state0 = {k: v.clone() for k, v in model.state_dict().items()}
        for _ in range(3):                                       # direct, recorded, replayed
            model.load_state_dict(state0)
            out = st.step(x, targets)
        res[mode] = (float(out["total_loss"]), st.plan.arena.clone(), {id(p): n for n, p in model.named_parameters()},
                     st.plan, {k: v.clone() for k, v in model.state_dict().items() if "running" in k})
        res[mode + "_fg"] = st.plan.loss_ws.fg.clone()
    (la, ga, na, pa, ra), (lb, gb, nb, pb, rb) = res["paired"], res["chains"]
    assert abs(la - lb) / abs(la) < 1e-5
    # SimOTA is a discrete decision: on the GPU an anchor at a rounding-level cost tie can be assigned differently by the two
    # schedules (atomics-order noise in the BatchNorm statistics); the gradients are then legitimately different by percents
    flips = int((res["paired_fg"] != res["chains_fg"]).sum())
    assert flips <= 2
    if flips:
        return
    by_name = lambda plan, names, arena: {names[id(p)]: plan.gview[id(p)] for p in plan.params}      # noqa: E731
    A, Bm = by_name(pa, na, ga), by_name(pb, nb, gb)
    # the emulator's float atomics are ordered; on the GPU the BatchNorm statistics of the two schedules are summed in different
    # orders and the difference is amplified down to the stem (1.7e-4 measured on its weight gradient, fp32 mode)
    tol = 1e-5 if str(backend) == "cpu" else 1e-3
    for k in A:
        assert _rel(Bm[k].cpu(), A[k].cpu()) < tol, k
    for k in ra:
        assert _rel(rb[k], ra[k]) < (1e-6 if str(backend) == "cpu" else 1e-5), k


@pytest.mark.parametrize("ring", [3, 9, 0])             # 0: no ring — a raw-gradient buffer per layer, no slot events at all (round 6)
def test_raw_gradient_ring_depth_does_not_change_the_step(backend, ring, monkeypatch):
    """TrainPlan.RING raw-gradient slots (9 by default, round 5): a frame chain that wants a slot back waits for the weight gradient
    and the data gradients that last read it ("acquire_cur" / "slot_done" in the launch tape).  With 3 slots every layer reuses a
    slot its predecessors' weight gradients may still be reading; the recorded and replayed steps must give the gradients of the
    single-stream step either way."""
    from streamyolo_amd import train_engine
    from streamyolo_amd.train_engine import TrainPlan, TrainStep
    monkeypatch.setattr(TrainPlan, "RING", ring)
    monkeypatch.setattr(train_engine, "BWD_SPLIT_FRAMES", "1")
    cfg = O.OracleConfig.named("nano")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    Hh, Ww = (32, 64) if str(backend) == "cpu" else (128, 192)
    x = synth_frames(2, Hh, Ww, seed=5).to(backend)
    lab, sup = synth_labels(2, Hh, Ww, cfg.num_classes, num_gt=3, seed=6)
    targets = (lab.to(backend), sup.to(backend))
    res = []
    for serial in (True, False):
        model = sy.build_model("nano")
        model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        model = model.to(backend).train().set_compute_dtype("fp32")
        model.head.use_l1 = True
        st = TrainStep(model, graph=False)
        plan = st._ensure(x)
        assert len(plan.dyraw_ring) == ring
        state0 = {k: v.clone() for k, v in model.state_dict().items()}
        for _ in range(3):                                       # direct, recorded, replayed
            model.load_state_dict(state0)
            plan.force_serial = serial
            out = st.step(x, targets)
        names = {id(p): n for n, p in model.named_parameters()}
        res.append((float(out["total_loss"]), {names[id(p)]: plan.gview[id(p)].clone().cpu() for p in plan.params},
                    plan.loss_ws.fg.clone()))
    (l0, g0, f0), (l1, g1, f1) = res
    assert abs(l0 - l1) / abs(l0) < 1e-5
    if int((f0 != f1).sum()):                                    # a SimOTA tie flipped by atomics-order noise (GPU only): see above
        assert str(backend) != "cpu"
        return
    tol = 1e-5 if str(backend) == "cpu" else 1e-3
    for k in g0:
        assert _rel(g1[k], g0[k]) < tol, k


@pytest.mark.parametrize("ring", [9])
def test_frame_chains_wait_for_the_weight_gradients_once_per_batch_of_layers(backend, ring, monkeypatch):
    """Round 6: in the split backward a frame chain records no "slot done" event of its own (its half of a raw-gradient slot is only
    rewritten by itself) and waits for the weight gradients once per SLOT_BATCH layers (they retire in order on their streams) — the
    recorded tape issues far fewer stream waits than the rounds 3-5 schedule (both events, every layer), and the replayed step's
    gradients equal the single-stream step's either way."""
    from streamyolo_amd import train_engine
    from streamyolo_amd.train_engine import TrainPlan, TrainStep
    monkeypatch.setattr(TrainPlan, "RING", ring)
    monkeypatch.setattr(train_engine, "BWD_SPLIT_FRAMES", "1")
    cfg = O.OracleConfig.named("nano")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    Hh, Ww = (32, 64) if str(backend) == "cpu" else (128, 192)
    x = synth_frames(2, Hh, Ww, seed=5).to(backend)
    lab, sup = synth_labels(2, Hh, Ww, cfg.num_classes, num_gt=3, seed=6)
    targets = (lab.to(backend), sup.to(backend))
    res, waits = {}, {}
    for name, chain_ev, batch, serial in (("serial", False, 3, True), ("old", True, 1, False), ("new", False, 3, False)):
        monkeypatch.setattr(train_engine, "CHAIN_SLOT_DONE", chain_ev)
        monkeypatch.setattr(train_engine, "SLOT_BATCH", batch)
        model = sy.build_model("nano")
        model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        model = model.to(backend).train().set_compute_dtype("fp32")
        model.head.use_l1 = True
        st = TrainStep(model, graph=False)
        plan = st._ensure(x)
        state0 = {k: v.clone() for k, v in model.state_dict().items()}
        for _ in range(3):                                       # direct, recorded, replayed
            model.load_state_dict(state0)
            plan.force_serial = serial
            out = st.step(x, targets)
        names = {id(p): n for n, p in model.named_parameters()}
        res[name] = (float(out["total_loss"]), {names[id(p)]: plan.gview[id(p)].clone().cpu() for p in plan.params})
        if not serial:
            tape = plan.programs["bwd"][1]
            if str(backend) == "cpu":                            # no streams on the emulator: count on a replay over stand-in handles
                import ctypes as C                               # (the kernels run again, synchronously; the results above are copies)
                tape.replay(C.c_void_p(1), C.c_void_p(2), more=[C.c_void_p(3), C.c_void_p(4), C.c_void_p(5)])
            waits[name] = tape.counters()[1]
    assert 0 < waits["new"] < 0.7 * waits["old"], waits
    tol = 1e-5 if str(backend) == "cpu" else 1e-3
    for name in ("old", "new"):
        assert abs(res[name][0] - res["serial"][0]) / abs(res["serial"][0]) < 1e-5
        for k in res["serial"][1]:
            assert _rel(res[name][1][k], res["serial"][1][k]) < tol, (name, k)


def test_backward_tape_joins_every_weight_gradient_stream(backend, monkeypatch):
    """The backward tape must end with main-stream waits for EVERY stream that carried a weight gradient (stream 1 through the
    "join" mark, streams 3+ through "dep" marks): the arena's next reader — optimizer, all-reduce of late buckets, the caller —
    works on the main stream."""
    from streamyolo_amd import _lib, train_engine
    from streamyolo_amd.train_engine import TrainStep
    marks = []
    real = _lib.NativeTape.mark

    def logged(self, name, arg=None):
        marks.append((name, arg))
        return real(self, name, arg)
    monkeypatch.setattr(_lib.NativeTape, "mark", logged)
    assert train_engine.WGRAD_STREAMS[:2] == [1, 3]                  # the default: weight gradients alternate between two streams
    cfg = O.OracleConfig.named("nano")
    model = sy.build_model("nano")
    model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0), strict=True)
    model = model.to(backend).train().set_compute_dtype("fp32")
    x = synth_frames(2, 32, 64, seed=2).to(backend)
    lab, sup = synth_labels(2, 32, 64, cfg.num_classes, num_gt=4, seed=3)
    st = TrainStep(model, graph=False)
    st.step(x, (lab.to(backend), sup.to(backend)))                   # direct
    marks.clear()
    st.step(x, (lab.to(backend), sup.to(backend)))                   # recorded: forward tape, then backward tape
    used = {a for n, a in marks if n == "cur" and a not in (None, 0, 2)}
    assert 3 in used and 1 in used
    last_use = max(i for i, (n, a) in enumerate(marks) if n == "cur" and a == 3)
    assert any(n == "dep" and a == (3, 0) for n, a in marks[last_use:]), "stream 3 is never joined after its last weight gradient"
    assert marks[-1][0] == "join"


def test_plan_on_the_stride2_window_kernels_matches_the_implicit_gemm_plan(backend, monkeypatch):
    """The stride-2 3x3 layers on the window-in-LDS kernels (tile 110 forward, 108 data gradient — tuner candidates since round 4)
    inside a whole training step: the tuner is replaced by a function that hands them out wherever their shape rules allow, and
    the step must reproduce the loss and gradients of the plan that keeps those layers on the implicit-GEMM kernel (exact-fp32 mode)."""
    from streamyolo_amd import ops as ops_mod
    from streamyolo_amd.train_engine import TrainStep
    cfg = O.OracleConfig.named("nano")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    Hh, Ww = (64, 96) if str(backend) == "cpu" else (128, 192)
    x = synth_frames(2, Hh, Ww, seed=2).to(backend)
    lab, sup = synth_labels(2, Hh, Ww, cfg.num_classes, num_gt=4, seed=3)
    targets = (lab.to(backend), sup.to(backend))
    handed = {"s2": 0, "s2d": 0}
    real_tile = ops_mod.tuned_tile

    def tile_for(window):
        def tile(mode, dtype, N, H, W, Cin, Cout, k, stride, device, with_stats=False, only=None):
            if only is not None or k != 3 or stride != 2 or Cin % 16:
                return real_tile(mode, dtype, N, H, W, Cin, Cout, k, stride, device, with_stats=with_stats, only=only)
            if not window:
                return 0
            handed["s2" if mode == ops_mod.CONV_FWD else "s2d"] += 1
            return 110 if mode == ops_mod.CONV_FWD else 108
        return tile
    res = {}
    for mode in ("default", "new_tiles"):
        monkeypatch.setattr(ops_mod, "tuned_tile", tile_for(mode == "new_tiles"))
        model = sy.build_model("nano")
        model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        model = model.to(backend).train().set_compute_dtype("fp32")
        model.head.use_l1 = True
        st = TrainStep(model, graph=False)
        state0 = {k: v.clone() for k, v in model.state_dict().items()}
        for _ in range(2):                                       # direct, recorded
            model.load_state_dict(state0)
            out = st.step(x, targets)
        res[mode] = (float(out["total_loss"]), {n: st.plan.gview[id(p)].clone() for n, p in model.named_parameters()},
                     st.plan.loss_ws.fg.clone())
    assert min(handed.values()) > 0, handed
    (la, ga, fa), (lb, gb, fb) = res["default"], res["new_tiles"]
    assert abs(la - lb) / abs(la) < 1e-5
    if int((fa != fb).sum()):
        return
    # other kernels = another fp32 summation order (1e-7 per layer), amplified by the BatchNorms of the deepest maps, which
    # normalise over a handful of values at this input size.  The level to expect is the one the GPU-verified halo tile 117 shows
    # against the same default plan on the emulator (median 8e-5, maximum 2.5e-4; tile 110 alone: 4e-5 ... 8e-5 median, the
    # data-gradient variant alone 3e-7)
    errs = sorted((_rel(gb[k].cpu(), ga[k].cpu()), k) for k in ga)
    assert errs[len(errs) // 2][0] < 3e-4, errs[len(errs) // 2]
    assert errs[-1][0] < 2e-3, errs[-5:]


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,H,W", [("s", 4, 320, 512), ("l", 2, 288, 480)])
def test_exact_mode_steps_are_bit_equal(name, B, H, W):
    """Two identical exact-fp32 steps produce the SAME bits — loss dict, SimOTA assignment, every parameter gradient and the
    BatchNorm running statistics — although the frames run as stream-parallel chains: in this mode every statistics workgroup
    owns a replica row (TrainPlan.exact_stats) and the loss partials are added in index order (tal_finish_kernel), so nothing
    depends on the arrival order of atomics.  The reference's step is deterministic for a given input (tal_head.py:679-712
    is a pure function of the predictions); until round 5 one SimOTA decision flipped in ~2 of 5 identical runs here."""
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    cfg = O.OracleConfig.named(name)
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    x = synth_frames(B, H, W, seed=2).to(dev)
    lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=12, seed=3)
    lab, sup = lab.to(dev), sup.to(dev)
    model = sy.build_model(name).to(dev).train().set_compute_dtype("fp32")
    model.head.use_l1 = True
    runs = []
    for _ in range(4):                                       # run 0 tunes, run 1 records the tape, runs 2-3 replay it
        model.load_state_dict(sd, strict=True)
        for p in model.parameters():
            p.grad = None
        out = model(x, (lab, sup))
        out["total_loss"].backward()
        torch.cuda.synchronize()
        plan = next(p_ for k_, p_ in model._plans.plans.items() if str(k_[0]).startswith("train"))
        assert plan.exact_stats
        runs.append(({k: out[k].detach().clone() for k in NAMES}, plan.loss_ws.fg.clone(),
                     {n: p.grad.detach().clone() for n, p in model.named_parameters()},
                     {n: b.detach().clone() for n, b in model.named_buffers() if "running" in n}))
    for r in runs[1:]:
        for k in NAMES:
            assert torch.equal(r[0][k], runs[0][0][k]), k
        assert torch.equal(r[1], runs[0][1])
        bad = [n for n in r[2] if not torch.equal(r[2][n], runs[0][2][n])]
        assert not bad, "%d gradients differ between identical steps, e.g. %s" % (len(bad), bad[:4])
        bad = [n for n in r[3] if not torch.equal(r[3][n], runs[0][3][n])]
        assert not bad, "running statistics differ: %s" % bad[:4]
