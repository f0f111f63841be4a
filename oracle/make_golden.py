#!/usr/bin/env python3
"""Mint tests/golden/*.npz from the REFERENCE ITSELF and pin oracle/streamyolo_oracle.py to it.

Runs only where /root/reference exists (the build container).  It imports the reference's
unmodified exps/model/{yolox,dfp_pafpn,darknet,tal_head}.py through oracle/ref_shim (the
`yolox==0.3` symbols they need are not vendored — see oracle/ref_shim/README.md), builds the model
exactly as `Exp.get_model()` does (cfgs/s_s50_onex_dfp_tal_flip.py:34-55: BN eps 1e-3 /
momentum 0.03, `initialize_biases(1e-2)`), loads the deterministic synthetic weights from
streamyolo_amd.utils.synth, and records its outputs.  While doing so it asserts that the
functional CPU oracle reproduces the reference (the numbers printed at the end go into DESIGN.md).

Test infrastructure only.  Usage:  python oracle/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("STREAMYOLO_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "ref_shim"))
sys.path.insert(0, REF)

from oracle import streamyolo_oracle as O                      # noqa: E402
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels, load_bn_stats  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def build_reference(cfg: O.OracleConfig):
    import torch.nn as nn
    from exps.model.yolox import YOLOX
    from exps.model.dfp_pafpn import DFPPAFPN
    from exps.model.tal_head import TALHead
    ic = list(cfg.in_channels)
    model = YOLOX(DFPPAFPN(cfg.depth, cfg.width, in_channels=ic),
                  TALHead(cfg.num_classes, cfg.width, in_channels=ic, gamma=cfg.gamma,
                          ignore_thr=cfg.ignore_thr, ignore_value=cfg.ignore_value))
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = cfg.bn_eps, cfg.bn_momentum
    model.head.initialize_biases(1e-2)
    return model


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def clone_sd(sd, grad=False):
    out = {}
    for k, v in sd.items():
        t = v.clone()
        if grad and t.dtype.is_floating_point and "running_" not in k:
            t.requires_grad_(True)
        out[k] = t
    return out


def main():
    torch.set_num_threads(os.cpu_count())
    os.makedirs(GOLD, exist_ok=True)
    report = {}

    # ------------------------------------------------------------------ key inventory
    for name in ("nano", "s", "l"):
        cfg = O.OracleConfig.named(name)
        ref = build_reference(cfg)
        ref_shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        assert ref_shapes == O.param_shapes(cfg), "oracle param inventory != reference (%s)" % name
        report["keys_" + name] = len(ref_shapes)
        nparam = sum(p.numel() for p in ref.parameters())
        report["params_" + name] = nparam
        with open(os.path.join(GOLD, "keys_%s.txt" % name), "w") as f:
            for k in sorted(ref_shapes):
                f.write("%s %s\n" % (k, "x".join(map(str, ref_shapes[k])) or "scalar"))

    # ------------------------------------------------------------------ BN calibration (see synth.load_bn_stats)
    def calibrate(name, H, W):
        import torch.nn as nn
        cfg = O.OracleConfig.named(name)
        ref = build_reference(cfg)
        ref.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0), strict=True)
        ref.train()
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.momentum = 1.0
        with torch.no_grad():
            fused = ref.backbone(synth_frames(2, H, W, seed=11), mode="off_pipe")
            ref.head.training = False            # eval branch of the head, BN children still in train mode
            ref.head(fused)
        stats = {k: v.numpy() for k, v in ref.state_dict().items() if "running_" in k}
        np.savez_compressed(os.path.join(GOLD, "bnstats_%s.npz" % name), **stats)

    calibrate("nano", 128, 192)
    calibrate("s", 320, 512)
    calibrate("l", 320, 512)

    # ------------------------------------------------------------------ eval goldens
    def eval_case(name, B, H, W, tag):
        cfg = O.OracleConfig.named(name)
        ref = build_reference(cfg)
        sd = synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats(name, GOLD))
        ref.load_state_dict(sd, strict=True)
        ref.eval()
        x = synth_frames(B, H, W, seed=2)
        with torch.no_grad():
            fused_ref = ref.backbone(x.clone(), mode="off_pipe")
            dec_ref = ref(x.clone())
            o1, buf = ref(x[:, 3:6].clone(), mode="on_pipe")
            o2, buf2 = ref(x[:, 0:3].clone(), buffer=buf, mode="on_pipe")
        assert torch.equal(o2, dec_ref), "reference off_pipe != chained on_pipe"
        dec_or = O.forward_eval(clone_sd(sd), x, cfg)
        (f_or, cur_or) = O.forward_features(clone_sd(sd), x, cfg)
        oo1, obuf = O.forward_online(clone_sd(sd), x[:, 3:6], None, cfg)
        oo2, _ = O.forward_online(clone_sd(sd), x[:, 0:3], obuf, cfg)
        report["%s_eval_rel" % tag] = rel(dec_or, dec_ref)
        report["%s_online_rel" % tag] = max(rel(oo1, o1), rel(oo2, o2))
        for a, b in zip(f_or, fused_ref):
            assert rel(a, b) < 1e-5
        assert rel(dec_or, dec_ref) < 1e-5, report
        # post-processing on the reference's decoded tensor, through the shim postprocess
        from yolox.utils import postprocess as ref_post
        dets_ref = ref_post(dec_ref.clone(), cfg.num_classes, 0.01, 0.65)
        dets_or = O.postprocess(dec_ref, cfg.num_classes, 0.01, 0.65)
        keep_idx = []
        for dr, (do, idx) in zip(dets_ref, dets_or):
            n_ref = 0 if dr is None else dr.shape[0]
            assert n_ref == do.shape[0], (n_ref, do.shape)
            if n_ref:
                assert torch.equal(dr, do), "oracle postprocess != shim postprocess"
            keep_idx.append(idx.numpy().astype(np.int32))
        report["%s_ndet" % tag] = [int(k.shape[0]) for k in keep_idx]
        out = {"decoded": dec_ref.numpy(), "online_first": o1.numpy(),
               "shape": np.array([B, H, W], dtype=np.int32)}
        for i, f in enumerate(fused_ref):
            if f.numel() < 400000:
                out["fused%d" % i] = f.numpy()
        for i, k in enumerate(keep_idx):
            out["keep%d" % i] = k
        np.savez_compressed(os.path.join(GOLD, "%s.npz" % tag), **out)

    eval_case("nano", 2, 64, 96, "nano_eval_2x64x96")
    eval_case("nano", 1, 152, 200, "nano_eval_1x152x200")      # odd /8 map (19x25): non-integer upsample
    eval_case("s", 1, 600, 960, "s_eval_1x600x960")

    # ------------------------------------------------------------------ training goldens
    def train_case(name, B, H, W, tag, keep_grads):
        cfg = O.OracleConfig.named(name)
        ref = build_reference(cfg)
        sd = synth_state_dict(O.param_shapes(cfg), seed=0)
        ref.load_state_dict(sd, strict=True)
        ref.train()
        ref.head.use_l1 = True                         # double_trainer.py:209-216
        x = synth_frames(B, H, W, seed=2)
        lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=6 if H < 200 else 16, seed=3)
        out_ref = ref(x.clone(), (lab.clone(), sup.clone()))
        out_ref["total_loss"].backward()
        osd = clone_sd(sd, grad=True)
        out_or = O.forward_train(osd, x, lab, sup, cfg)
        out_or["total_loss"].backward()
        names = ("total_loss", "iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg")
        losses_ref = np.array([float(out_ref[k]) for k in names], dtype=np.float64)
        losses_or = np.array([float(out_or[k]) for k in names], dtype=np.float64)
        report["%s_loss_ref" % tag] = losses_ref.tolist()
        report["%s_loss_rel" % tag] = float(np.abs(losses_ref - losses_or).max() / np.abs(losses_ref).max())
        assert report["%s_loss_rel" % tag] < 1e-4, report
        gmax = 0.0
        grads = {}
        rsd = ref.state_dict()
        for k, p in ref.named_parameters():
            g_ref = p.grad
            g_or = osd[k].grad
            assert g_ref is not None and g_or is not None, k
            gmax = max(gmax, rel(g_or, g_ref))
            grads[k] = g_ref
        report["%s_grad_rel_max" % tag] = gmax
        stats = {}
        for k, v in rsd.items():
            if "running_" in k or "num_batches" in k:
                assert rel(osd[k], v) < 1e-4, k
                stats[k] = v
        out = {"losses": losses_ref, "fg_mask": out_or["_fg_mask"].numpy(),
               "shape": np.array([B, H, W], dtype=np.int32)}
        for k in keep_grads(grads):
            out["grad:" + k] = grads[k].numpy()
        for k, v in stats.items():
            if v.numel() <= 64:
                out["stat:" + k] = v.numpy()
        out["grad_norms"] = np.array([float(grads[k].double().norm()) for k in sorted(grads)])
        np.savez_compressed(os.path.join(GOLD, "%s.npz" % tag), **out)

    train_case("nano", 2, 64, 96, "nano_train_2x64x96", lambda g: sorted(g))
    train_case("s", 2, 160, 256, "s_train_2x160x256",
               lambda g: [k for k in sorted(g) if g[k].numel() <= 4096])

    # ------------------------------------------------------------------ SimOTA pinned on the reference's own get_assignments
    cfg = O.OracleConfig.named("nano")
    ref = build_reference(cfg)
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)
    ref.load_state_dict(sd, strict=True)
    ref.train()
    ref.head.use_l1 = True
    B, H, W = 2, 96, 160
    x = synth_frames(B, H, W, seed=5)
    lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=9, seed=6)
    captured = []
    orig = ref.head.get_assignments

    def spy(*a, **kw):
        r = orig(*a, **kw)
        captured.append([t.clone() if torch.is_tensor(t) else t for t in r])
        return r
    ref.head.get_assignments = spy
    ref(x.clone(), (lab.clone(), sup.clone()))
    osd = clone_sd(sd)
    out_or = O.forward_train(osd, x, lab, sup, cfg)
    sim = {}
    for i, (cls_m, fg, piou, mg, nfg) in enumerate(captured):
        assert torch.equal(fg, out_or["_fg_mask"][i]), "oracle SimOTA fg mask != reference"
        sim["fg%d" % i] = fg.numpy()
        sim["matched_gt%d" % i] = mg.numpy().astype(np.int32)
        sim["matched_iou%d" % i] = piou.numpy()
    raw = O.flatten_levels([r.detach() for r in out_or["_raw"]])
    sim["raw"] = raw.numpy()
    sim["labels"] = lab.numpy()
    sim["support"] = sup.numpy()
    sim["hw"] = np.array([tuple(r.shape[-2:]) for r in out_or["_raw"]], dtype=np.int32)
    np.savez_compressed(os.path.join(GOLD, "nano_simota_2x96x160.npz"), **sim)
    report["simota_fg"] = [int(c[4]) for c in captured]

    # ------------------------------------------------------------------ FLOP accounting cross-check
    for name, want in (("s", 61.43e9), ("l", 384.30e9)):
        got = O.conv_flops_per_pair(O.OracleConfig.named(name))
        report["gflop_pair_" + name] = got / 1e9
        assert abs(got - want) / want < 2e-3, (name, got)

    for k in sorted(report):
        print("%-32s %s" % (k, report[k]))
    with open(os.path.join(GOLD, "REPORT.txt"), "w") as f:
        f.write("# written by oracle/make_golden.py (torch %s); oracle-vs-reference agreement\n" % torch.__version__)
        for k in sorted(report):
            f.write("%-32s %s\n" % (k, report[k]))


if __name__ == "__main__":
    main()
