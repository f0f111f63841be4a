"""End-to-end inference parity: streamyolo_amd.YOLOX (HIP plan) vs the reference's golden outputs.

Parity metric: rel = max|y - y_ref| / max|y_ref| per compared tensor; the
north-star bound is 1e-3 in the fp32 (exact-f32 MFMA) mode; bf16 / fp16 speed modes are reported
against their own looser bounds.  NMS keep-indices must match bit-exactly when fed the reference's
decoded tensor.
"""
import os

import numpy as np
import pytest
import torch

import streamyolo_amd as sy
from oracle import streamyolo_oracle as O
from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, load_bn_stats


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _model(name, device):
    cfg = O.OracleConfig.named(name)
    m = sy.build_model(name)
    sd = synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats(name))
    missing = m.load_state_dict(sd, strict=True)          # strict: key set == reference's (480 / 768 keys)
    return m.to(device).eval(), sd, cfg


def test_state_dict_keys_match_reference(golden_dir):
    for name, n in (("nano", 480), ("s", 480), ("l", 768)):
        want = [l.split()[0] for l in open(os.path.join(golden_dir, "keys_%s.txt" % name))]
        got = list(sy.build_model(name).state_dict().keys())
        assert sorted(got) == sorted(want) and len(got) == n


# nano is 8..32 channels wide, so 16-bit rounding noise averages out far less than in s / l: loose speed-mode bounds
@pytest.mark.parametrize("dt,tol", [("fp32", 1e-3), ("fp16", 6e-2), ("bf16", 3e-1)])
@pytest.mark.parametrize("tag", ["nano_eval_2x64x96", "nano_eval_1x152x200"])
def test_eval_off_pipe_and_on_pipe_nano(backend, golden_dir, tag, dt, tol):
    z = np.load(os.path.join(golden_dir, tag + ".npz"))
    B, H, W = [int(v) for v in z["shape"]]
    model, sd, cfg = _model("nano", backend)
    model.set_compute_dtype(dt)
    x = synth_frames(B, H, W, seed=2).to(backend)
    with torch.no_grad():
        out = model(x)
        assert out.shape == z["decoded"].shape
        r = _rel(out.cpu(), z["decoded"])
        assert r < tol, "off_pipe rel err %.3e" % r
        # fused FPN features through the stand-alone backbone entry point
        if dt == "fp32":
            feats = model.backbone(x)
            for i, f in enumerate(feats):
                if "fused%d" % i in z.files:
                    assert _rel(f.float().cpu(), z["fused%d" % i]) < tol
            assert _rel(model.head(feats).cpu(), z["decoded"]) < tol
        # streaming: first frame fuses with itself, second frame with the buffer
        o1, buf = model(x[:, 3:6].contiguous(), mode="on_pipe")
        assert _rel(o1.cpu(), z["online_first"]) < tol
        o2, _ = model(x[:, 0:3].contiguous(), buffer=buf, mode="on_pipe")
        assert _rel(o2.cpu(), z["decoded"]) < tol
        if dt == "fp32":
            assert _rel(o2.cpu(), out.cpu()) < 1e-5        # off_pipe == chained on_pipe


def test_three_channel_input_is_duplicated(backend):
    """get_model_info / thop path: eval forward on a [1,3,64,64] zero image (double_trainer.py:140-142)."""
    model, sd, cfg = _model("nano", backend)
    import copy
    m2 = copy.deepcopy(model)
    x = torch.zeros(1, 3, 64, 64, device=backend)
    with torch.no_grad():
        out = m2(x)
    assert out.shape == (1, 84, 13)
    ref = O.forward_eval({k: v.clone() for k, v in sd.items()}, x.cpu(), cfg)
    assert _rel(out.cpu(), ref) < 1e-3


def test_postprocess_dropin_on_reference_decoded(backend, golden_dir):
    z = np.load(os.path.join(golden_dir, "nano_eval_2x64x96.npz"))
    dec = torch.from_numpy(z["decoded"]).to(backend)
    det, idx, cnt = sy.postprocess.__globals__["postprocess_device"](dec, 8, 0.01, 0.65)
    for i in range(dec.shape[0]):
        n = int(cnt[i])
        assert np.array_equal(idx[i, :n].cpu().numpy(), z["keep%d" % i])
    outs = sy.postprocess(dec.clone(), 8, 0.01, 0.65)
    ref = O.postprocess(torch.from_numpy(z["decoded"]), 8, 0.01, 0.65)
    for o, (rdet, _) in zip(outs, ref):
        assert torch.equal(o.cpu(), rdet)


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [("fp32", 1e-3), ("fp16", 3e-2), ("bf16", 1.5e-1)])
def test_eval_s_600x960_full_size(golden_dir, dt, tol):
    """BASELINE.json configs[1]: StreamYOLO-s 600x960 forward on one MI355X vs the reference output."""
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    z = np.load(os.path.join(golden_dir, "s_eval_1x600x960.npz"))
    model, sd, cfg = _model("s", dev)
    model.set_compute_dtype(dt)
    x = synth_frames(1, 600, 960, seed=2).to(dev)
    with torch.no_grad():
        out = model(x)
    r = _rel(out.cpu(), z["decoded"])
    print("s 600x960 %s rel err vs reference: %.3e" % (dt, r))
    assert out.shape == (1, 11850, 13) and r < tol
    # NMS on the REFERENCE's decoded tensor: bit-exact keep list at full size
    det, idx, cnt = sy.postprocess.__globals__["postprocess_device"](torch.from_numpy(z["decoded"]).to(dev), 8, 0.01, 0.65)
    n = int(cnt[0])
    assert np.array_equal(idx[0, :n].cpu().numpy(), z["keep0"])
    if dt == "fp32":
        # end-to-end detection-set agreement when NMS consumes OUR decoded tensor
        det2, idx2, cnt2 = sy.postprocess.__globals__["postprocess_device"](out.float().contiguous(), 8, 0.01, 0.65)
        mine = set(idx2[0, :int(cnt2[0])].cpu().tolist())
        ref = set(z["keep0"].tolist())
        agree = len(mine & ref) / max(len(ref), 1)
        print("end-to-end kept-set agreement: %.4f (%d vs %d)" % (agree, len(mine), len(ref)))
        assert agree > 0.97
