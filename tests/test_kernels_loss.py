"""sy_tal_loss (SimOTA + Trend-Aware loss + gradient) vs the oracle restatement of the reference's
TALHead.get_losses, on the golden raw head outputs whose assignment was captured from the
reference's own get_assignments (tests/golden/nano_simota_2x96x160.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import streamyolo_oracle as O
from streamyolo_amd import ops


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _oracle_loss_and_grad(raw, hw, lab, sup, cfg):
    r = raw.clone().requires_grad_(True)
    levels, a0 = [], 0
    for (h, w) in hw:
        levels.append(r[:, a0:a0 + h * w].permute(0, 2, 1).reshape(raw.shape[0], raw.shape[2], h, w))
        a0 += h * w
    out = O.tal_loss(levels, lab, sup, cfg)
    out["total_loss"].backward()
    return out, r.grad


@pytest.mark.parametrize("use_support", [True, False])
def test_tal_loss_matches_reference_assignment_and_autograd(backend, golden_dir, use_support):
    z = np.load(os.path.join(golden_dir, "nano_simota_2x96x160.npz"))
    cfg = O.OracleConfig.named("nano")
    raw = torch.from_numpy(z["raw"])
    lab, sup = torch.from_numpy(z["labels"]), torch.from_numpy(z["support"])
    if not use_support:
        sup = torch.zeros_like(sup)                 # no support GT: trend weights fall back to ones (:394-396)
        sup[:, 0] = torch.tensor([1.0, 50.0, 40.0, 30.0, 20.0])   # keep >=1 row so the CPU reference avoids .cuda()
    hw = [tuple(int(v) for v in r) for r in z["hw"]]
    B, A, nch = raw.shape
    ws = ops.TalLossWorkspace(B, A, nch, hw, cfg.strides, backend)
    losses, d_raw, fg = ops.tal_loss(raw.to(backend), lab, sup, cfg.num_classes, cfg.gamma, cfg.ignore_thr,
                                     cfg.ignore_value, True, ws)
    ref, gref = _oracle_loss_and_grad(raw, hw, lab, sup, cfg)
    # assignment: identical foreground mask to the REFERENCE's get_assignments
    for i in range(B):
        assert np.array_equal(fg[i].cpu().numpy().astype(bool), z["fg%d" % i])
    names = ("total_loss", "iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg")
    want = np.array([float(ref[k]) for k in names])
    got = losses[:6].cpu().numpy()
    assert np.abs(got - want).max() / np.abs(want).max() < 2e-5, (got, want)
    assert _rel(d_raw.cpu(), gref) < 2e-4


def test_tal_loss_empty_image_and_no_gt(backend):
    cfg = O.OracleConfig.named("nano")
    hw = [(4, 6), (2, 3), (1, 2)]
    A = sum(h * w for h, w in hw)
    g = torch.Generator().manual_seed(0)
    raw = torch.randn(2, A, 13, generator=g) * 0.5
    lab = torch.zeros(2, 120, 5)
    sup = torch.zeros(2, 120, 5)
    lab[0, 0] = torch.tensor([3.0, 20.0, 14.0, 18.0, 12.0])      # image 1 has no GT at all
    sup[0, 0] = torch.tensor([3.0, 22.0, 15.0, 17.0, 12.0])
    ws = ops.TalLossWorkspace(2, A, 13, hw, cfg.strides, backend)
    losses, d_raw, fg = ops.tal_loss(raw.to(backend), lab, sup, 8, 1.0, 0.5, 1.5, True, ws)
    ref, gref = _oracle_loss_and_grad(raw, hw, lab, sup, cfg)
    assert int(fg[1].sum()) == 0 and int(fg[0].sum()) == int(ref["_fg_mask"][0].sum())
    assert abs(float(losses[0]) - float(ref["total_loss"])) / abs(float(ref["total_loss"])) < 2e-5
    assert _rel(d_raw.cpu(), gref) < 2e-4


def test_tal_loss_more_anchors_than_one_mask_pass_covers(backend):
    """The matching / conflict kernels scan a thread's anchors 64 at a time (candidate flags as a bit mask, then the candidates): a
    1536x2560 image (80640 anchors) takes five passes in tal_match_kernel (256 x 64 anchors each) and two in tal_resolve_kernel
    (1024 x 64) — foreground mask and matched ground truths equal to the oracle's, loss and gradient as in the small cases."""
    cfg = O.OracleConfig.named("nano")
    hw = [(192, 320), (96, 160), (48, 80)]
    A = sum(h * w for h, w in hw)
    g = torch.Generator().manual_seed(3)
    raw = torch.randn(1, A, 13, generator=g) * 0.5
    lab = torch.zeros(1, 120, 5)
    sup = torch.zeros(1, 120, 5)
    n = 9
    cx, cy = torch.rand(n, generator=g) * 2300 + 100, torch.rand(n, generator=g) * 1300 + 100
    w, h = torch.rand(n, generator=g) * 300 + 40, torch.rand(n, generator=g) * 200 + 40
    cls = torch.randint(0, 8, (n,), generator=g).float()
    lab[0, :n] = torch.stack([cls, cx, cy, w, h], 1)
    sup[0, :n] = torch.stack([cls, cx + 6, cy - 4, w * 1.05, h * 0.97], 1)
    ws = ops.TalLossWorkspace(1, A, 13, hw, cfg.strides, backend)
    losses, d_raw, fg = ops.tal_loss(raw.to(backend), lab, sup, 8, cfg.gamma, cfg.ignore_thr, cfg.ignore_value, True, ws)
    ref, gref = _oracle_loss_and_grad(raw, hw, lab, sup, cfg)
    fg_ref = ref["_fg_mask"][0].numpy().astype(bool)
    assert fg_ref[256 * 64:].any() and fg_ref[1024 * 64:].any()          # later passes hold matches
    assert np.array_equal(fg[0].cpu().numpy().astype(bool), fg_ref)
    mg = ops.tal_assignment(ws)[0].cpu().numpy().reshape(-1)
    assert np.array_equal(mg, ref["_matched_gt"].numpy().reshape(-1))
    assert abs(float(losses[0]) - float(ref["total_loss"])) / abs(float(ref["total_loss"])) < 2e-5
    assert _rel(d_raw.cpu(), gref) < 2e-4
