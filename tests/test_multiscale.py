"""Multi-scale training as the reference trainer drives it: a new (H, W) every 10 iterations
(cfgs/l_s50_onex_dfp_tal_flip.py:139-158 `random_resize`, exps/train_utils/double_trainer.py:276-279).  The plan cache must
stay bounded (LRU), plans of different sizes share their scratch, steps after a size switch reproduce the oracle, and memory
stays flat once every size has been seen."""
import numpy as np
import pytest
import torch

import streamyolo_amd as sy
from oracle import streamyolo_oracle as O
from streamyolo_amd.model.plan_cache import PlanCache
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels

SIZES = [(32, 64), (64, 64), (32, 96)]          # emulator-sized; the GPU test below cycles five real multi-scale sizes


def _oracle_loss(cfg, sd, x, lab, sup):
    osd = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        return float(O.forward_train(osd, x, lab, sup, cfg)["total_loss"])


def test_size_switches_reproduce_the_oracle_with_a_bounded_plan_cache(backend, monkeypatch):
    monkeypatch.setattr(PlanCache, "MAX_TRAIN_PLANS", 2)
    cfg = O.OracleConfig.named("nano")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    model = sy.build_model("nano")
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model = model.to(backend).train().set_compute_dtype("fp32")
    model.head.use_l1 = True
    want = {}
    seen_scratch = set()
    # three sizes with two plans alive (the third evicts the first), then back to the first: rebuilt, and run into its tape replay
    for i, steps in ((0, 1), (1, 1), (2, 1), (0, 3)):
        H, W = SIZES[i]
        x = synth_frames(2, H, W, seed=40 + i)
        lab, sup = synth_labels(2, H, W, cfg.num_classes, num_gt=4, seed=50 + i)
        if (H, W) not in want:
            want[(H, W)] = _oracle_loss(cfg, sd, x, lab, sup)
        for _ in range(steps):
            model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)      # same state for every step
            for p in model.parameters():
                p.grad = None
            out = model(x.to(backend), (lab.to(backend), sup.to(backend)))
            out["total_loss"].backward()
            got = float(out["total_loss"])
            assert abs(got - want[(H, W)]) / abs(want[(H, W)]) < 1e-3, (i, H, W, got, want[(H, W)])
        train_plans = [k for k in model._plans.plans if str(k[0]).startswith("train")]
        assert len(train_plans) <= 2
        assert train_plans[-1][2:4] == (H, W)                   # most recently used last
        seen_scratch.add(model._plans.scratch["wgrad_ws"].data_ptr())
    assert len(seen_scratch) == 1                                   # ONE split-K workspace for every size


def test_a_train_step_pins_its_plan_against_the_drop_in_paths_evictions(backend, monkeypatch):
    """TrainStep holds its plan across steps; the drop-in path at OTHER sizes (and the sub-module entry points between
    their forward and backward) goes through the same bounded cache.  A plan somebody still uses must never be released: here
    a TrainStep at one size survives the drop-in path cycling through three other sizes with two plans allowed."""
    from streamyolo_amd.train_engine import TrainStep
    monkeypatch.setattr(PlanCache, "MAX_TRAIN_PLANS", 2)
    cfg = O.OracleConfig.named("nano")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    model = sy.build_model("nano")
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model = model.to(backend).train().set_compute_dtype("fp32")
    model.head.use_l1 = True
    H0, W0 = 32, 64
    x0 = synth_frames(2, H0, W0, seed=40).to(backend)
    lab0, sup0 = synth_labels(2, H0, W0, cfg.num_classes, num_gt=4, seed=50)
    t0 = (lab0.to(backend), sup0.to(backend))
    st = TrainStep(model, graph=False)
    first = float(st.step(x0, t0)["total_loss"])
    pinned = st.plan
    assert pinned.pending == 1
    for i, (H, W) in enumerate([(64, 64), (32, 96), (64, 96)]):
        x = synth_frames(2, H, W, seed=41 + i).to(backend)
        lab, sup = synth_labels(2, H, W, cfg.num_classes, num_gt=4, seed=51 + i)
        out = model(x, (lab.to(backend), sup.to(backend)))
        assert any(pl is pinned for pl in model._plans.plans.values()), "the TrainStep's plan was evicted"
        out["total_loss"].backward()
    assert pinned.grads.py is not None                                    # not released
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    for _ in range(3):                                                     # direct, recorded, replayed: same loss as the first step
        model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
        again = float(st.step(x0, t0)["total_loss"])
        assert abs(again - first) / abs(first) < 1e-5
    assert st.plan is pinned and pinned.pending == 1
    # a size change moves the pin
    x1 = synth_frames(2, 64, 64, seed=41).to(backend)
    lab1, sup1 = synth_labels(2, 64, 64, cfg.num_classes, num_gt=4, seed=51)
    st.step(x1, (lab1.to(backend), sup1.to(backend)))
    assert st.plan is not pinned and pinned.pending == 0 and st.plan.pending == 1


def test_plan_pins_are_released_when_their_holder_goes_away(backend):
    """a TrainStep that is dropped, and a drop-in forward whose backward never runs (an exception, a validation
    forward outside no_grad), give their pin on the plan back — otherwise the bounded plan cache could never evict those plans
    under multi-scale training."""
    import gc
    from streamyolo_amd.train_engine import TrainStep
    cfg = O.OracleConfig.named("nano")
    model = sy.build_model("nano")
    model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0), strict=True)
    model = model.to(backend).train().set_compute_dtype("fp32")
    model.head.use_l1 = True
    x = synth_frames(2, 32, 64, seed=40).to(backend)
    lab, sup = synth_labels(2, 32, 64, cfg.num_classes, num_gt=4, seed=50)
    t = (lab.to(backend), sup.to(backend))
    st = TrainStep(model, graph=False)
    st.step(x, t)
    plan = st.plan
    assert plan.pending == 1
    del st
    gc.collect()
    assert plan.pending == 0
    out = model(x, t)                                        # drop-in forward with grad ...
    assert plan.pending == 1
    del out                                                  # ... whose backward never comes
    gc.collect()
    assert plan.pending == 0
    out = model(x, t)
    out["total_loss"].backward()
    assert plan.pending == 0


def test_backward_repacks_a_raw_gradient_that_was_scaled_in_place(backend):
    """loss() hands out the plan's persistent d_raw; a caller that scales it IN PLACE (loss scaling) and passes the
    same tensor must not get the stale unscaled packed copy in the prediction convolutions' gradients."""
    from streamyolo_amd.train_engine import get_train_plan
    cfg = O.OracleConfig.named("nano")
    model = sy.build_model("nano")
    model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0), strict=True)
    model = model.to(backend).train().set_compute_dtype("fp32")
    model.head.use_l1 = True
    x = synth_frames(2, 32, 64, seed=40).to(backend)
    lab, sup = synth_labels(2, 32, 64, cfg.num_classes, num_gt=4, seed=50)
    plan = get_train_plan(model, x)
    grads = []
    for scale in (1.0, 4.0):
        plan.forward(x)
        _, d_raw = plan.loss(lab.to(backend), sup.to(backend))
        if scale != 1.0:
            d_raw.mul_(scale)
        grads.append(plan.backward(d_raw).clone())
    assert float((grads[1] - 4.0 * grads[0]).abs().max() / (4.0 * grads[0]).abs().max()) < 1e-5


@pytest.mark.gpu
def test_multiscale_memory_stays_flat_and_later_plans_build_from_the_tuner_cache(tmp_path, monkeypatch):
    """GPU: cycle through five sizes twice with at most three training plans alive: allocated memory after the second cycle
    is not above the first cycle's peak, every step matches the oracle loss (bf16 bound), and a plan rebuilt for a size seen
    before (its tuner choices are persisted) is ready much faster than the first build."""
    import time
    from streamyolo_amd import _lib, ops
    _lib.use_library(_lib.DEFAULT_PATH)
    monkeypatch.setenv("STREAMYOLO_TUNE_CACHE", str(tmp_path / "tune.json"))
    ops._tune_store.__init__()
    dev = torch.device("cuda:0")
    cfg = O.OracleConfig.named("s")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    model = sy.build_model("s")
    model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    model = model.to(dev).train().set_compute_dtype("bf16")
    model.head.use_l1 = True
    sizes = [(416, 672), (480, 768), (544, 864), (608, 960), (448, 704)]     # multiples of 32, as random_resize draws them
    peak, build_s = [], {}
    for rnd in range(2):
        for i, (H, W) in enumerate(sizes):
            x = synth_frames(2, H, W, seed=60 + i)
            lab, sup = synth_labels(2, H, W, cfg.num_classes, num_gt=8, seed=70 + i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = model(x.to(dev), (lab.to(dev), sup.to(dev)))
            out["total_loss"].backward()
            torch.cuda.synchronize()
            build_s.setdefault((H, W), []).append(time.perf_counter() - t0)
            if rnd == 0:
                want = _oracle_loss(cfg, sd, x, lab, sup)
                assert abs(float(out["total_loss"]) - want) / abs(want) < 5e-2, (H, W)
            model.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
            assert len([k for k in model._plans.plans if str(k[0]).startswith("train")]) <= PlanCache.MAX_TRAIN_PLANS
        del out
        torch.cuda.synchronize()
        peak.append(torch.cuda.memory_allocated())
    print("allocated after cycle 1 / 2: %.1f / %.1f MB" % (peak[0] / 2**20, peak[1] / 2**20))
    print("first build vs rebuild (s):", {k: [round(t, 2) for t in v] for k, v in build_s.items()})
    assert peak[1] <= peak[0] * 1.02 + (8 << 20)
    # sizes[0] and sizes[1] were evicted during cycle 1 and rebuilt in cycle 2 from the persisted tuner choices: no re-tuning,
    # a rebuilt plan's first step (plan construction + one step through the wrappers) stays well under a second
    assert build_s[sizes[0]][1] < build_s[sizes[0]][0] and build_s[sizes[0]][1] < 1.0
    assert (tmp_path / "tune.json").exists()
