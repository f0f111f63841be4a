"""sy_sgd_ema_step (fused SGD-nesterov + EMA) against torch.optim.SGD + the ModelEMA
restatement in oracle/optim_oracle.py: parameters, momentum buffers and EMA state after several iterations."""
import copy

import pytest
import torch

import streamyolo_amd as sy
from oracle import streamyolo_oracle as O
from oracle.optim_oracle import ReferenceOptimEMA, param_groups
from streamyolo_amd.optim import FusedSGDEMA, yolox_param_groups
from streamyolo_amd.utils.synth import synth_state_dict


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def test_param_groups_match_yolox_walk():
    m = sy.build_model("nano")
    for a, b in zip(yolox_param_groups(m), param_groups(m)):
        assert [id(p) for p in a] == [id(p) for p in b]
    pg0, pg1, pg2 = yolox_param_groups(m)
    n = sum(1 for _ in m.parameters())
    assert len(pg0) + len(pg1) + len(pg2) == n and len({id(p) for p in pg0 + pg1 + pg2}) == n


def test_fused_sgd_ema_matches_torch_sgd_and_model_ema(backend):
    cfg = O.OracleConfig.named("nano")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    ref_m = sy.build_model("nano"); ref_m.load_state_dict(sd, strict=True); ref_m.train()
    our_m = copy.deepcopy(ref_m).to(backend)
    ref = ReferenceOptimEMA(ref_m, lr=0.01)
    g = torch.Generator().manual_seed(5)
    ours = None
    for it, (lr, scale) in enumerate([(0.01, 1.0), (0.02, 0.5), (0.005, 1.0), (0.005, 0.25)]):
        for pr, po in zip(ref_m.parameters(), our_m.parameters()):
            grad = torch.randn(pr.shape, generator=g) * 0.1
            pr.grad = grad.clone()
            if po.grad is None:
                po.grad = grad.clone().to(backend)
            else:
                po.grad.copy_(grad)
        with torch.no_grad():                                    # BatchNorm buffers move too (the EMA tracks them)
            for (k, br), (_, bo) in zip(ref_m.named_buffers(), our_m.named_buffers()):
                if br.dtype.is_floating_point:
                    delta = torch.randn(br.shape, generator=g) * 0.01
                    br.add_(delta); bo.add_(delta.to(backend))
        if ours is None:
            ours = FusedSGDEMA(our_m)
        ref.step(lr, grad_scale=scale)
        ours.step(lr, grad_scale=scale)
    for (k, pr), (_, po) in zip(ref_m.named_parameters(), our_m.named_parameters()):
        assert _rel(po.detach().cpu(), pr.detach()) < 1e-6, k
    o = 0
    for p, pr in zip(ours.params, [q for grp in param_groups(ref_m) for q in grp]):
        buf = ours.buf[o:o + p.numel()].view(p.shape).cpu()
        o += p.numel()
        assert _rel(buf, ref.opt.state[pr]["momentum_buffer"]) < 1e-6
    esd, rsd = ours.ema_state_dict(), ref.ema.state_dict()
    assert set(esd) == set(rsd)
    for k in rsd:
        if rsd[k].dtype.is_floating_point:
            assert _rel(esd[k].cpu(), rsd[k]) < 1e-6, k
        else:
            assert torch.equal(esd[k].cpu(), rsd[k]), k


@pytest.mark.gpu
def test_fused_optimizer_on_the_training_arena():
    """With the training plan: gradients live in the flat arena (p.grad are views); one fused step after
    forward + backward moves every parameter and the EMA, and the next forward sees the new weights."""
    from streamyolo_amd import _lib
    from streamyolo_amd.train_engine import TrainStep
    from streamyolo_amd.utils.synth import synth_frames, synth_labels
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    cfg = O.OracleConfig.named("nano")
    m = sy.build_model("nano"); m.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0), strict=True)
    m = m.to(dev).set_compute_dtype("fp32")
    x = synth_frames(2, 96, 160, seed=2).to(dev)
    lab, sup = synth_labels(2, 96, 160, cfg.num_classes, num_gt=6, seed=3)
    st = TrainStep(m)
    opt = None
    losses = []
    for _ in range(5):
        out = st.step(x, (lab.to(dev), sup.to(dev)))
        if opt is None:
            opt = FusedSGDEMA(m)
        before = [p.detach().clone() for p in m.parameters()]
        opt.step(1e-3)
        losses.append(float(out["total_loss"]))
        moved = [not torch.equal(a, p.detach()) for a, p in zip(before, m.parameters())]
        assert sum(moved) > 0.9 * len(moved)                     # (a zero-gradient tensor without decay stays put)
    assert losses[-1] < losses[0]                                # same batch, SGD on it: the loss goes down
    ema = opt.ema_state_dict()
    assert set(ema) == set(m.state_dict())


def test_cached_eval_plan_follows_raw_pointer_weight_updates(backend):
    """FusedSGDEMA.step and the BatchNorm running-statistics launch write parameters / buffers through raw pointers
    (tensor._version does not move): an inference plan built BEFORE the update must repack its weights and refold its
    BatchNorm afterwards (ops.weights_epoch), i.e. train -> eval -> optimizer step -> eval serves the new weights."""
    from streamyolo_amd.utils.synth import synth_frames, load_bn_stats
    cfg = O.OracleConfig.named("nano")
    m = sy.build_model("nano")
    m.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats("nano")), strict=True)
    m = m.to(backend).eval().set_compute_dtype("fp32")
    x = synth_frames(1, 64, 96, seed=2).to(backend)
    with torch.no_grad():
        out0 = m(x).clone()
    g = torch.Generator().manual_seed(7)
    for p in m.parameters():
        p.grad = (torch.randn(p.shape, generator=g) * 0.05).to(backend)
    versions = [p._version for p in m.parameters()]
    FusedSGDEMA(m).step(0.05)
    assert versions == [p._version for p in m.parameters()]      # the raw-pointer write is invisible to torch ...
    with torch.no_grad():
        out1 = m(x).clone()                                      # ... the cached plan must notice anyway
        fresh = copy.deepcopy(m)                                 # deepcopy: empty plan cache, same (updated) weights
        want = fresh(x)
    assert _rel(out1.cpu(), want.cpu()) < 1e-6
    assert _rel(out1.cpu(), out0.cpu()) > 1e-3                   # and the update was big enough to matter
