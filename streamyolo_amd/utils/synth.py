"""Deterministic synthetic weights / frames / labels (SURVEY.md §8(d) "configs -> concrete synthetic inputs").

There is no network for checkpoints or Argoverse-HD, so every test, golden vector and bench run
uses tensors minted here.  Values are a pure function of (key name, shape, seed) — NOT of module
construction order — so the reference modules (loaded through oracle/ref_shim), the CPU oracle
and the HIP path all see bit-identical parameters without storing tens of MB of weights in git.

Scaling is chosen so activations stay O(1) through ~130 convs in eval mode (default torch init
shrinks them by ~3x per layer, which would make parity vacuous): conv weights ~ N(0, 2.5/fan_in),
BN running statistics can be overridden by calibrated ones (load_bn_stats) and are non-trivial so
that folding is exercised, and the prediction biases follow `initialize_biases(1e-2)`
(cfgs/s_s50_onex_dfp_tal_flip.py:54 of the reference).
"""
import math
import zlib

import torch

__all__ = ["synth_state_dict", "synth_frames", "synth_labels"]


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) * 2654435761 + seed * 97 + 12345) % (2 ** 63 - 1))
    return g


def load_bn_stats(name, root=None):
    """Calibrated BN running statistics for model `name` ('nano' | 's' | 'm' | 'l'), minted once by
    oracle/make_golden.py / make_golden_m.py (one training-mode pass of the reference over synthetic frames) and
    committed as tests/golden/bnstats_<name>.npz.  With them the eval-mode network keeps O(1)
    activations through all ~130 convs, so output parity is sensitive to every layer."""
    import os
    import numpy as np
    if root is None:
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))),
                            "tests", "golden")
    z = np.load(os.path.join(root, "bnstats_%s.npz" % name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def synth_state_dict(shapes, seed=0, bn_stats=None):
    """shapes: dict key -> tuple shape (reference state_dict names).  Returns dict key -> fp32/long
    tensor.  `bn_stats` (see load_bn_stats) overrides the BN running_mean / running_var."""
    out = {}
    for key in sorted(shapes):
        shp = tuple(shapes[key])
        g = _gen(key, seed)
        if key.endswith("num_batches_tracked"):
            t = torch.zeros((), dtype=torch.long)
        elif key.endswith(".bn.weight"):
            t = torch.rand(shp, generator=g) + 0.5
        elif key.endswith(".bn.bias"):
            t = torch.randn(shp, generator=g) * 0.1
        elif key.endswith(".bn.running_mean"):
            t = torch.randn(shp, generator=g) * 0.1
        elif key.endswith(".bn.running_var"):
            t = torch.rand(shp, generator=g) + 0.5
        elif key.endswith(".weight") and len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            is_pred = "_preds." in key
            # prediction convs are kept small so box logits stay in a trained model's range (wh = exp(v)*stride)
            if "reg_preds" in key:
                std = 0.05 / math.sqrt(fan_in)
            elif is_pred:                    # obj / cls logits spread enough for some anchors to pass conf 0.01
                std = 0.8 / math.sqrt(fan_in)
            else:
                std = math.sqrt(2.5 / fan_in)
            t = torch.randn(shp, generator=g) * std
        elif key.endswith(".bias"):
            if "cls_preds" in key or "obj_preds" in key:
                t = torch.full(shp, -math.log((1 - 1e-2) / 1e-2))
                t = t + torch.randn(shp, generator=g) * 0.05
            else:
                t = torch.randn(shp, generator=g) * 0.1
        else:
            raise KeyError("synth_state_dict: unrecognised key %r" % key)
        if bn_stats is not None and key in bn_stats:
            t = bn_stats[key].clone().float().reshape(shp)
        out[key] = t.contiguous()
    return out


def synth_frames(batch, height, width, seed=2, channels=6):
    """Frame pairs: [B,6,H,W] fp32 in 0..255, channels 0-2 current frame, 3-5 support (previous) frame."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.rand((batch, channels, height, width), generator=g) * 255.0


def synth_labels(batch, height, width, num_classes=8, num_gt=16, max_labels=120, seed=3):
    """(labels_t+1, labels_t): two [B,120,5] tensors of (cls, cx, cy, w, h) px; zero padded (cfg :80)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    cur = torch.zeros(batch, max_labels, 5)
    sup = torch.zeros(batch, max_labels, 5)
    ngt = min(num_gt, max_labels)
    cls = torch.randint(0, num_classes, (batch, ngt), generator=g).float()
    cx = torch.rand(batch, ngt, generator=g) * width
    cy = torch.rand(batch, ngt, generator=g) * height
    bw = torch.rand(batch, ngt, generator=g) * (min(width, 960) * 0.15) + 16.0
    bh = torch.rand(batch, ngt, generator=g) * (min(height, 600) * 0.24) + 16.0
    cur[:, :ngt] = torch.stack([cls, cx, cy, bw, bh], dim=2)
    dxy = torch.randn(batch, ngt, 2, generator=g) * 4.0
    swh = torch.rand(batch, ngt, 2, generator=g) * 0.2 + 0.9
    sup[:, :ngt, 0] = cls
    sup[:, :ngt, 1:3] = cur[:, :ngt, 1:3] + dxy
    sup[:, :ngt, 3:5] = cur[:, :ngt, 3:5] * swh
    return cur, sup
