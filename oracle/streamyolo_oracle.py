"""CPU ORACLE for the StreamYOLO dual-frame hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product package `streamyolo_amd` never does (it fails loudly when the HIP library is missing).

What it is: a functional, torch-CPU-fp32 restatement of the reference algorithm on a plain
state_dict (reference key names), written independently of both the reference's nn.Module tree
and the product's execution plan.  Each function cites the reference lines it follows
(paths relative to /root/reference; `yolox==0.3` symbols are not vendored there — their semantics
are taken from SURVEY.md Appendix C and pinned through oracle/ref_shim, see below).

Parity pinning: the reference ships NO tests, golden vectors or fixtures for this path
 — "parity unpinned" by the reference itself.  This oracle is therefore
pinned against *outputs of the reference run here*: oracle/make_golden.py imports the reference's
unmodified exps/model/{yolox,dfp_pafpn,darknet,tal_head}.py (through oracle/ref_shim) in the build
container, checks this restatement against it and writes tests/golden/*.npz, which
tests/test_oracle_golden.py re-checks everywhere (no /root/reference needed at test time).
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# configuration
# ----------------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    depth: float = 0.33
    width: float = 0.50
    num_classes: int = 8
    in_channels: Tuple[int, int, int] = (256, 512, 1024)
    strides: Tuple[int, int, int] = (8, 16, 32)
    gamma: float = 1.0            # cfgs/*_tal_*.py:49-50 pass gamma=1.0
    ignore_thr: float = 0.5
    ignore_value: float = 1.5
    bn_eps: float = 1e-3          # init_yolo, cfgs/s_s50_onex_dfp_tal_flip.py:40-44
    bn_momentum: float = 0.03

    @staticmethod
    def named(name: str) -> "OracleConfig":
        table = {
            "s": dict(depth=0.33, width=0.50, ignore_thr=0.5, ignore_value=1.5),
            "m": dict(depth=0.67, width=0.75, ignore_thr=0.4, ignore_value=1.7),
            "l": dict(depth=1.0, width=1.0, ignore_thr=0.5, ignore_value=1.6),
            "l2x": dict(depth=1.0, width=1.0, ignore_thr=0.4, ignore_value=1.5),
            # tiny variant used only to keep CPU tests and fixtures small
            "nano": dict(depth=0.33, width=0.125, ignore_thr=0.5, ignore_value=1.5),
        }
        return OracleConfig(**table[name])


# ----------------------------------------------------------------------------------------------
# parameter inventory (reference state_dict key names)
# ----------------------------------------------------------------------------------------------
def _bc(shapes, pfx, cin, cout, k):
    """BaseConv = Conv2d(no bias) + BatchNorm2d (yolox network_blocks, Appendix C)."""
    shapes[pfx + ".conv.weight"] = (cout, cin, k, k)
    for nm in ("weight", "bias", "running_mean", "running_var"):
        shapes[pfx + ".bn." + nm] = (cout,)
    shapes[pfx + ".bn.num_batches_tracked"] = ()


def _csp(shapes, pfx, cin, cout, n):
    hid = int(cout * 0.5)
    _bc(shapes, pfx + ".conv1", cin, hid, 1)
    _bc(shapes, pfx + ".conv2", cin, hid, 1)
    _bc(shapes, pfx + ".conv3", 2 * hid, cout, 1)
    for i in range(n):
        _bc(shapes, "%s.m.%d.conv1" % (pfx, i), hid, hid, 1)
        _bc(shapes, "%s.m.%d.conv2" % (pfx, i), hid, hid, 3)


def param_shapes(cfg: OracleConfig) -> Dict[str, Tuple[int, ...]]:
    """Key -> shape for YOLOX(DFPPAFPN, TALHead).  Follows exps/model/darknet.py:111-165,
    dfp_pafpn.py:33-105, tal_head.py:55-131."""
    s: Dict[str, Tuple[int, ...]] = {}
    bc = int(cfg.width * 64)
    bd = max(round(cfg.depth * 3), 1)
    nd = round(3 * cfg.depth)
    c3, c4, c5 = [int(c * cfg.width) for c in cfg.in_channels]
    bb = "backbone.backbone."
    _bc(s, bb + "stem.conv", 12, bc, 3)
    _bc(s, bb + "dark2.0", bc, bc * 2, 3)
    _csp(s, bb + "dark2.1", bc * 2, bc * 2, bd)
    _bc(s, bb + "dark3.0", bc * 2, bc * 4, 3)
    _csp(s, bb + "dark3.1", bc * 4, bc * 4, bd * 3)
    _bc(s, bb + "dark4.0", bc * 4, bc * 8, 3)
    _csp(s, bb + "dark4.1", bc * 8, bc * 8, bd * 3)
    _bc(s, bb + "dark5.0", bc * 8, bc * 16, 3)
    _bc(s, bb + "dark5.1.conv1", bc * 16, bc * 8, 1)
    _bc(s, bb + "dark5.1.conv2", bc * 32, bc * 16, 1)
    _csp(s, bb + "dark5.2", bc * 16, bc * 16, bd)
    nk = "backbone."
    _bc(s, nk + "lateral_conv0", c5, c4, 1)
    _csp(s, nk + "C3_p4", 2 * c4, c4, nd)
    _bc(s, nk + "reduce_conv1", c4, c3, 1)
    _csp(s, nk + "C3_p3", 2 * c3, c3, nd)
    _bc(s, nk + "bu_conv2", c3, c3, 3)
    _csp(s, nk + "C3_n3", 2 * c3, c4, nd)
    _bc(s, nk + "bu_conv1", c4, c4, 3)
    _csp(s, nk + "C3_n4", 2 * c4, c5, nd)
    _bc(s, nk + "jian2", c3, c3 // 2, 1)
    _bc(s, nk + "jian1", c4, c4 // 2, 1)
    _bc(s, nk + "jian0", c5, c5 // 2, 1)
    hw = int(256 * cfg.width)
    for k, ck in enumerate((c3, c4, c5)):
        _bc(s, "head.stems.%d" % k, ck, hw, 1)
        for j in range(2):
            _bc(s, "head.cls_convs.%d.%d" % (k, j), hw, hw, 3)
            _bc(s, "head.reg_convs.%d.%d" % (k, j), hw, hw, 3)
        for nm, co in (("cls_preds", cfg.num_classes), ("reg_preds", 4), ("obj_preds", 1)):
            s["head.%s.%d.weight" % (nm, k)] = (co, hw, 1, 1)
            s["head.%s.%d.bias" % (nm, k)] = (co,)
    return s


# ----------------------------------------------------------------------------------------------
# blocks (yolox.models.network_blocks semantics)
# ----------------------------------------------------------------------------------------------
class _Ctx:
    """Carries the state_dict and the BN mode through the functional blocks."""

    def __init__(self, sd, cfg: OracleConfig, train: bool):
        self.sd, self.cfg, self.train = sd, cfg, train

    def conv_bn_silu(self, pfx, x, k, stride):
        sd = self.sd
        y = F.conv2d(x, sd[pfx + ".conv.weight"], None, stride, (k - 1) // 2)
        if self.train:
            sd[pfx + ".bn.num_batches_tracked"] += 1
        y = F.batch_norm(y, sd[pfx + ".bn.running_mean"], sd[pfx + ".bn.running_var"],
                         sd[pfx + ".bn.weight"], sd[pfx + ".bn.bias"], self.train,
                         self.cfg.bn_momentum, self.cfg.bn_eps)
        return F.silu(y)

    def csp(self, pfx, x, n, shortcut):
        a = self.conv_bn_silu(pfx + ".conv1", x, 1, 1)
        b = self.conv_bn_silu(pfx + ".conv2", x, 1, 1)
        for i in range(n):
            t = self.conv_bn_silu("%s.m.%d.conv1" % (pfx, i), a, 1, 1)
            t = self.conv_bn_silu("%s.m.%d.conv2" % (pfx, i), t, 3, 1)
            a = t + a if shortcut else t
        return self.conv_bn_silu(pfx + ".conv3", torch.cat([a, b], 1), 1, 1)


def focus_pack(x):
    """Space-to-depth, patch order TL, BL, TR, BR (trap T4)."""
    return torch.cat([x[..., 0::2, 0::2], x[..., 1::2, 0::2], x[..., 0::2, 1::2], x[..., 1::2, 1::2]], 1)


def backbone_pafpn(ctx: _Ctx, frame):
    """One frame through CSPDarknet (darknet.py:167-179) + PAFPN (dfp_pafpn.py:124-140).
    Returns the three PRE-fusion PAN outputs (pan2 /8, pan1 /16, pan0 /32)."""
    cfg = ctx.cfg
    bd = max(round(cfg.depth * 3), 1)
    nd = round(3 * cfg.depth)
    bb = "backbone.backbone."
    x = ctx.conv_bn_silu(bb + "stem.conv", focus_pack(frame), 3, 1)
    x = ctx.conv_bn_silu(bb + "dark2.0", x, 3, 2)
    x = ctx.csp(bb + "dark2.1", x, bd, True)
    x = ctx.conv_bn_silu(bb + "dark3.0", x, 3, 2)
    d3 = ctx.csp(bb + "dark3.1", x, bd * 3, True)
    x = ctx.conv_bn_silu(bb + "dark4.0", d3, 3, 2)
    d4 = ctx.csp(bb + "dark4.1", x, bd * 3, True)
    x = ctx.conv_bn_silu(bb + "dark5.0", d4, 3, 2)
    x = ctx.conv_bn_silu(bb + "dark5.1.conv1", x, 1, 1)                      # SPP (trap T5)
    x = torch.cat([x] + [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)], 1)
    x = ctx.conv_bn_silu(bb + "dark5.1.conv2", x, 1, 1)
    d5 = ctx.csp(bb + "dark5.2", x, bd, False)
    nk = "backbone."
    fpn0 = ctx.conv_bn_silu(nk + "lateral_conv0", d5, 1, 1)
    t = torch.cat([F.interpolate(fpn0, size=d4.shape[2:4], mode="nearest"), d4], 1)
    t = ctx.csp(nk + "C3_p4", t, nd, False)
    fpn1 = ctx.conv_bn_silu(nk + "reduce_conv1", t, 1, 1)
    t = torch.cat([F.interpolate(fpn1, size=d3.shape[2:4], mode="nearest"), d3], 1)
    pan2 = ctx.csp(nk + "C3_p3", t, nd, False)
    t = torch.cat([ctx.conv_bn_silu(nk + "bu_conv2", pan2, 3, 2), fpn1], 1)
    pan1 = ctx.csp(nk + "C3_n3", t, nd, False)
    t = torch.cat([ctx.conv_bn_silu(nk + "bu_conv1", pan1, 3, 2), fpn0], 1)
    pan0 = ctx.csp(nk + "C3_n4", t, nd, False)
    return pan2, pan1, pan0


def dfp_fuse(ctx: _Ctx, cur, sup):
    """out_K = cat[jianK(cur_K), jianK(sup_K)] + cur_K  (dfp_pafpn.py:168-170).  The current
    frame's jian call comes first (BN running-stat order, trap T2)."""
    outs = []
    for name, c, s in zip(("jian2", "jian1", "jian0"), cur, sup):
        a = ctx.conv_bn_silu("backbone." + name, c, 1, 1)
        b = ctx.conv_bn_silu("backbone." + name, s, 1, 1)
        outs.append(torch.cat([a, b], 1) + c)
    return tuple(outs)


def features_off_pipe(ctx: _Ctx, x):
    """dfp_pafpn.py:232-243 + off_forward :109-175.  3-channel input is duplicated."""
    if x.shape[1] == 3:
        x = torch.cat([x, x], 1)
    cur = backbone_pafpn(ctx, x[:, 0:3])
    sup = backbone_pafpn(ctx, x[:, 3:6])
    return dfp_fuse(ctx, cur, sup), cur


def head_raw(ctx: _Ctx, feats):
    """Per level: cat[reg(4), obj(1), cls(nc)] raw logits, [B, 5+nc, h, w] (tal_head.py:159-174)."""
    sd = ctx.sd
    outs = []
    for k, x in enumerate(feats):
        x = ctx.conv_bn_silu("head.stems.%d" % k, x, 1, 1)
        c = ctx.conv_bn_silu("head.cls_convs.%d.0" % k, x, 3, 1)
        c = ctx.conv_bn_silu("head.cls_convs.%d.1" % k, c, 3, 1)
        r = ctx.conv_bn_silu("head.reg_convs.%d.0" % k, x, 3, 1)
        r = ctx.conv_bn_silu("head.reg_convs.%d.1" % k, r, 3, 1)
        cls = F.conv2d(c, sd["head.cls_preds.%d.weight" % k], sd["head.cls_preds.%d.bias" % k])
        reg = F.conv2d(r, sd["head.reg_preds.%d.weight" % k], sd["head.reg_preds.%d.bias" % k])
        obj = F.conv2d(r, sd["head.obj_preds.%d.weight" % k], sd["head.obj_preds.%d.bias" % k])
        outs.append(torch.cat([reg, obj, cls], 1))
    return outs


def anchor_grid(hw_list, strides, dtype=torch.float32):
    """(x_shift, y_shift, stride) per anchor, level-major then row-major (trap T9;
    tal_head.py:231-240, 245-256)."""
    xs, ys, ss = [], [], []
    for (h, w), s in zip(hw_list, strides):
        yv, xv = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        xs.append(xv.reshape(-1).to(dtype))
        ys.append(yv.reshape(-1).to(dtype))
        ss.append(torch.full((h * w,), float(s), dtype=dtype))
    return torch.cat(xs), torch.cat(ys), torch.cat(ss)


def flatten_levels(raw_levels):
    """[B,C,h,w] x3 -> [B, A, C] anchor-major (tal_head.py:217-219 / :236-239)."""
    return torch.cat([r.flatten(2) for r in raw_levels], 2).permute(0, 2, 1).contiguous()


def decode(flat, hw_list, strides):
    """xy = (xy + grid) * stride ; wh = exp(wh) * stride (tal_head.py:258-259 / :241-242)."""
    gx, gy, gs = anchor_grid(hw_list, strides, flat.dtype)
    out = flat.clone()
    out[..., 0] = (flat[..., 0] + gx) * gs
    out[..., 1] = (flat[..., 1] + gy) * gs
    out[..., 2] = torch.exp(flat[..., 2]) * gs
    out[..., 3] = torch.exp(flat[..., 3]) * gs
    return out


def head_eval(ctx: _Ctx, feats):
    """Eval branch of TALHead.forward (tal_head.py:196-221): sigmoid on obj/cls, decode boxes."""
    raw = head_raw(ctx, feats)
    hw = [tuple(r.shape[-2:]) for r in raw]
    flat = flatten_levels(raw)
    flat[..., 4:] = torch.sigmoid(flat[..., 4:])
    return decode(flat, hw, ctx.cfg.strides)


# ----------------------------------------------------------------------------------------------
# public forward entry points
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def forward_eval(sd, x, cfg: OracleConfig):
    """YOLOX.forward(mode='off_pipe') in eval mode (yolox.py:31-50): decoded [B, A, 5+nc]."""
    ctx = _Ctx(sd, cfg, False)
    fused, _ = features_off_pipe(ctx, x)
    return head_eval(ctx, fused)


@torch.no_grad()
def forward_features(sd, x, cfg: OracleConfig):
    """Fused (P3,P4,P5) and the current frame's pre-fusion PAN outputs, eval mode."""
    ctx = _Ctx(sd, cfg, False)
    return features_off_pipe(ctx, x)


@torch.no_grad()
def forward_online(sd, frame, buffer, cfg: OracleConfig):
    """YOLOX.forward(mode='on_pipe') (yolox.py:51-55; dfp_pafpn.py:177-228, :245-255)."""
    assert frame.shape[1] == 3
    ctx = _Ctx(sd, cfg, False)
    cur = backbone_pafpn(ctx, frame)
    if buffer is None:
        sup = cur
    else:
        assert len(buffer) == 3
        sup = buffer
    fused = dfp_fuse(ctx, cur, sup)
    return head_eval(ctx, fused), cur


# ----------------------------------------------------------------------------------------------
# box utilities (yolox.utils.bboxes_iou / yolox.models.losses.IOUloss, Appendix C, trap T13)
# ----------------------------------------------------------------------------------------------
def pairwise_iou_cxcywh(a, b):
    """[N,4] x [M,4] cxcywh -> [N,M]; `en` mask, no epsilon (bboxes_iou(xyxy=False))."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[None, :, :2] - b[None, :, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[None, :, :2] + b[None, :, 2:] / 2)
    en = (tl < br).to(a.dtype).prod(2)
    inter = (br - tl).prod(2) * en
    return inter / (a[:, 2:].prod(1)[:, None] + b[:, 2:].prod(1)[None, :] - inter)


def iou_loss_cxcywh(p, t):
    """1 - IoU^2 with +1e-16 in the denominator (IOUloss, loss_type='iou')."""
    tl = torch.max(p[:, :2] - p[:, 2:] / 2, t[:, :2] - t[:, 2:] / 2)
    br = torch.min(p[:, :2] + p[:, 2:] / 2, t[:, :2] + t[:, 2:] / 2)
    en = (tl < br).to(p.dtype).prod(1)
    inter = (br - tl).prod(1) * en
    iou = inter / (p[:, 2:].prod(1) + t[:, 2:].prod(1) - inter + 1e-16)
    return 1 - iou ** 2


# ----------------------------------------------------------------------------------------------
# SimOTA assignment (tal_head.py:479-712) — one image
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def simota_assign(gt_boxes, gt_cls, boxes, obj_logit, cls_logit, gx, gy, gs, num_classes):
    """Returns (fg_mask[A] bool, matched_gt[F] long, matched_iou[F]) for one image.

    gt_boxes [G,4] cxcywh px, gt_cls [G]; boxes [A,4] decoded predictions; obj_logit [A];
    cls_logit [A,nc]; gx/gy/gs per-anchor grid shifts and stride.
    """
    G = gt_boxes.shape[0]
    # --- candidates (get_in_boxes_info :594-677)
    xc = gx * gs + 0.5 * gs
    yc = gy * gs + 0.5 * gs
    l = (gt_boxes[:, 0] - 0.5 * gt_boxes[:, 2])[:, None]
    r = (gt_boxes[:, 0] + 0.5 * gt_boxes[:, 2])[:, None]
    t = (gt_boxes[:, 1] - 0.5 * gt_boxes[:, 3])[:, None]
    b = (gt_boxes[:, 1] + 0.5 * gt_boxes[:, 3])[:, None]
    in_box = torch.stack([xc[None] - l, yc[None] - t, r - xc[None], b - yc[None]], 2).min(2).values > 0.0
    rad = 2.5 * gs[None]
    cl = gt_boxes[:, 0:1] - rad
    cr = gt_boxes[:, 0:1] + rad
    ct = gt_boxes[:, 1:2] - rad
    cb = gt_boxes[:, 1:2] + rad
    in_ctr = torch.stack([xc[None] - cl, yc[None] - ct, cr - xc[None], cb - yc[None]], 2).min(2).values > 0.0
    cand = (in_box.sum(0) > 0) | (in_ctr.sum(0) > 0)
    both = in_box[:, cand] & in_ctr[:, cand]
    # --- costs (:517-553)
    cb_boxes = boxes[cand]
    ious = pairwise_iou_cxcywh(gt_boxes, cb_boxes)                       # [G,C]
    iou_cost = -torch.log(ious + 1e-8)
    onehot = F.one_hot(gt_cls.to(torch.int64), num_classes).float()       # [G,nc]
    prob = (torch.sigmoid(cls_logit[cand].float()) * torch.sigmoid(obj_logit[cand].float())[:, None]).sqrt()
    cls_cost = F.binary_cross_entropy(prob[None].expand(G, -1, -1),
                                      onehot[:, None, :].expand(-1, prob.shape[0], -1),
                                      reduction="none").sum(-1)
    cost = cls_cost + 3.0 * iou_cost + 100000.0 * (~both)
    # --- dynamic-k (:679-712)
    C = cost.shape[1]
    topv, _ = torch.topk(ious, min(10, C), dim=1)
    ks = torch.clamp(topv.sum(1).int(), min=1)
    match = torch.zeros_like(cost)
    for g in range(G):
        _, idx = torch.topk(cost[g], k=int(ks[g]), largest=False)
        match[g, idx] = 1.0
    multi = match.sum(0) > 1
    if multi.any():
        amin = cost[:, multi].argmin(0)
        match[:, multi] = 0.0
        match[amin, multi] = 1.0
    fg_in = match.sum(0) > 0
    fg = torch.zeros_like(cand)
    fg[cand.nonzero().squeeze(1)[fg_in]] = True
    matched_gt = match[:, fg_in].argmax(0)
    matched_iou = (match * ious).sum(0)[fg_in]
    return fg, matched_gt, matched_iou


# ----------------------------------------------------------------------------------------------
# Trend-Aware loss (tal_head.py:262-470)
# ----------------------------------------------------------------------------------------------
def tal_loss(raw_levels, labels, support_labels, cfg: OracleConfig):
    """raw_levels: list of [B,5+nc,h,w] raw logits with autograd history.  Returns the reference's
    6-tuple packed as the YOLOX.forward dict (yolox.py:39-46).  `use_l1` is True (the trainer sets
    it from epoch 0 — double_trainer.py:209-216 — and get_losses :435 needs it)."""
    nc = cfg.num_classes
    hw = [tuple(r.shape[-2:]) for r in raw_levels]
    flat = flatten_levels(raw_levels)                                    # [B,A,13] raw
    gx, gy, gs = anchor_grid(hw, cfg.strides, flat.dtype)
    dec = decode(flat, hw, cfg.strides)
    boxes, obj, cls = dec[..., :4], dec[..., 4], dec[..., 5:]
    origin = flat[..., :4]
    B, A = flat.shape[:2]
    nlabel = (labels.sum(2) > 0).sum(1)
    nsup = (support_labels.sum(2) > 0).sum(1)

    fg_all = torch.zeros(B, A, dtype=torch.bool)
    mg_all = torch.full((B, A), -1, dtype=torch.int32)                   # matched GT index per anchor (test read-out)
    cls_t, reg_t, l1_t, trend = [], [], [], []
    num_fg, num_gt_total = 0.0, 0.0
    for i in range(B):
        G = int(nlabel[i])
        S = int(nsup[i])
        num_gt_total += G
        if G == 0:
            continue
        gtb = labels[i, :G, 1:5]
        gtc = labels[i, :G, 0]
        fg, mg, miou = simota_assign(gtb, gtc, boxes[i].detach(), obj[i].detach(), cls[i].detach(),
                                     gx, gy, gs, nc)
        fg_all[i] = fg
        mg_all[i, fg] = mg.to(torch.int32)
        num_fg += int(fg.sum())
        cls_t.append(F.one_hot(gtc[mg].to(torch.int64), nc) * miou[:, None])
        reg_t.append(gtb[mg])
        g = gtb[mg]
        s_ = gs[fg]
        l1_t.append(torch.stack([g[:, 0] / s_ - gx[fg], g[:, 1] / s_ - gy[fg],
                                 torch.log(g[:, 2] / s_ + 1e-8), torch.log(g[:, 3] / s_ + 1e-8)], 1))
        if S == 0:                                                       # :394-396
            tr = torch.ones(G)
        else:                                                            # :398-403
            tr, _ = pairwise_iou_cxcywh(gtb, support_labels[i, :S, 1:5]).max(1)
            tr = torch.where(tr < cfg.ignore_thr, torch.full_like(tr, cfg.ignore_value), tr)
        trend.append(tr[mg])

    def _cat(lst, width):
        return torch.cat(lst, 0) if lst else flat.new_zeros((0, width))

    cls_t, reg_t, l1_t = _cat(cls_t, nc), _cat(reg_t, 4), _cat(l1_t, 4)
    trend = torch.cat(trend, 0) if trend else flat.new_zeros((0,))
    fgm = fg_all.view(-1)
    w = 1 / (trend ** cfg.gamma + 1e-8)                                    # :429
    iou_l = iou_loss_cxcywh(boxes.reshape(-1, 4)[fgm], reg_t)
    w_iou = ((w * iou_l.sum()) / (w * iou_l).sum()).detach()              # :432-433
    l1_l = (origin.reshape(-1, 4)[fgm] - l1_t).abs()
    w4 = w[:, None].repeat(1, 4)
    w_l1 = ((w4 * l1_l.sum()) / (w4 * l1_l).sum()).detach()               # :436-438
    nf = max(num_fg, 1)
    loss_iou = (w_iou * iou_l).sum() / nf
    loss_obj = F.binary_cross_entropy_with_logits(obj.reshape(-1, 1), fgm.to(flat.dtype)[:, None],
                                                  reduction="none").sum() / nf
    loss_cls = F.binary_cross_entropy_with_logits(cls.reshape(-1, nc)[fgm], cls_t,
                                                  reduction="none").sum() / nf
    loss_l1 = (w_l1 * l1_l).sum() / nf
    total = 5.0 * loss_iou + loss_obj + loss_cls + loss_l1
    return {"total_loss": total, "iou_loss": 5.0 * loss_iou, "l1_loss": loss_l1,
            "conf_loss": loss_obj, "cls_loss": loss_cls, "num_fg": num_fg / max(num_gt_total, 1),
            "_fg_mask": fg_all, "_matched_gt": mg_all}


def forward_train(sd, x, labels, support_labels, cfg: OracleConfig):
    """Training forward (yolox.py:33-46): BN in batch-statistics mode, running stats updated in
    `sd` in place (current frame first, then support frame — trap T2), returns the loss dict and
    the raw per-level head outputs.  Tensors in `sd` that have requires_grad get .grad from
    `out['total_loss'].backward()`."""
    ctx = _Ctx(sd, cfg, True)
    fused, _ = features_off_pipe(ctx, x)
    raw = head_raw(ctx, fused)
    out = tal_loss(raw, labels, support_labels, cfg)
    out["_raw"] = raw
    out["_fused"] = fused
    return out


# ----------------------------------------------------------------------------------------------
# post-processing (yolox.utils.postprocess + torchvision batched_nms semantics, Appendix C;
# in-tree explicit restatement: sAP/streamyolo/streamyolo_det.py:62-83)
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def greedy_nms(boxes, scores, thr):
    """Indices kept, score-descending; suppress iff IoU > thr (strict)."""
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = b.shape[0]
    alive = torch.ones(n, dtype=torch.bool)
    keep = []
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            lt = torch.maximum(b[i, :2], b[i + 1:, :2])
            rb = torch.minimum(b[i, 2:], b[i + 1:, 2:])
            wh = (rb - lt).clamp(min=0)
            inter = wh[:, 0] * wh[:, 1]
            alive[i + 1:] &= ~(inter / (area[i] + area[i + 1:] - inter) > thr)
    return order[torch.tensor(keep, dtype=torch.long)] if keep else order[:0]


@torch.no_grad()
def postprocess(decoded, num_classes, conf_thre=0.01, nms_thre=0.65):
    """decoded [B,A,5+nc] (cxcywh, obj, cls...) -> per image (dets [n,7], anchor_index [n]).
    dets rows: x1,y1,x2,y2,obj,class_conf,class_pred, in kept (score-descending) order.
    Class-aware NMS through the per-class coordinate offset `cls * (max_coord + 1)` exactly as
    torchvision.ops.batched_nms does."""
    results = []
    for img in decoded:
        xyxy = torch.stack([img[:, 0] - img[:, 2] / 2, img[:, 1] - img[:, 3] / 2,
                            img[:, 0] + img[:, 2] / 2, img[:, 1] + img[:, 3] / 2], 1)
        cconf, cpred = img[:, 5:5 + num_classes].max(1)
        score = img[:, 4] * cconf
        sel = (score >= conf_thre).nonzero().squeeze(1)
        if sel.numel() == 0:
            results.append((img.new_zeros((0, 7)), sel))
            continue
        bx, sc, cp = xyxy[sel], score[sel], cpred[sel]
        off = cp.to(bx) * (bx.max() + 1)
        keep = greedy_nms(bx + off[:, None], sc, nms_thre)
        idx = sel[keep]
        det = torch.cat([xyxy[idx], img[idx, 4:5], cconf[idx, None], cpred[idx, None].to(img.dtype)], 1)
        results.append((det, idx))
    return results


# ----------------------------------------------------------------------------------------------
# work accounting (conv FLOPs = 2*Cin*Cout*k^2*Hout*Wout over executed convs)
# ----------------------------------------------------------------------------------------------
def conv_flops_per_pair(cfg: OracleConfig, height=600, width=960, mode="off_pipe"):
    """Algorithmic forward conv FLOPs for one frame pair (off_pipe) or one streamed frame."""
    total = [0.0]

    class _Count(_Ctx):
        def conv_bn_silu(self, pfx, x, k, stride):
            w = self.sd[pfx + ".conv.weight"]
            ho = (x.shape[2] + 2 * ((k - 1) // 2) - k) // stride + 1
            wo = (x.shape[3] + 2 * ((k - 1) // 2) - k) // stride + 1
            total[0] += 2.0 * w[1] * w[0] * k * k * ho * wo
            return torch.empty((x.shape[0], w[0], ho, wo), device="meta")

    shapes = param_shapes(cfg)
    ctx = _Count(shapes, cfg, False)
    fr = torch.empty((1, 3, height, width), device="meta")
    cur = backbone_pafpn(ctx, fr)
    if mode == "off_pipe":
        sup = backbone_pafpn(ctx, fr)
    else:
        sup = cur
    fused = dfp_fuse(ctx, cur, sup)
    hw = int(256 * cfg.width)
    for k, f in enumerate(fused):
        x = ctx.conv_bn_silu("head.stems.%d" % k, f, 1, 1)
        for br in ("cls_convs", "reg_convs"):
            t = ctx.conv_bn_silu("head.%s.%d.0" % (br, k), x, 3, 1)
            ctx.conv_bn_silu("head.%s.%d.1" % (br, k), t, 3, 1)
        total[0] += 2.0 * hw * (cfg.num_classes + 5) * f.shape[2] * f.shape[3]
    return total[0]
