"""Trend-Aware loss + SimOTA assignment on the raw head output (device-side torch glue).

Follows TALHead.get_losses / get_assignments / get_in_boxes_info / dynamic_k_matching of the
reference (exps/model/tal_head.py:262-470, :479-592, :594-677, :679-712) on the anchor-major raw
logits `raw` [B, A, 5+nc] that the training plan's prediction kernels emit (fp32).  The tensors here
are tiny (<= a few hundred foreground anchors per image); the reference's OOM-to-CPU fallback
(:345-374) and per-image `empty_cache()` (:376) have no equivalent because nothing here allocates
more than a [G, candidates] cost matrix.

`use_l1` is honoured the way the reference's trainer drives it: the reference computes the L1 branch
unconditionally at :435 (it only works with use_l1=True, which double_trainer.py:209-216 sets from
epoch 0); with use_l1=False this implementation returns l1_loss = 0.0 instead of failing.
"""
import torch
import torch.nn.functional as F


def anchor_grid(hw_list, strides, device, dtype=torch.float32):
    xs, ys, ss = [], [], []
    for (h, w), s in zip(hw_list, strides):
        yv, xv = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
        xs.append(xv.reshape(-1).to(dtype))
        ys.append(yv.reshape(-1).to(dtype))
        ss.append(torch.full((h * w,), float(s), dtype=dtype, device=device))
    return torch.cat(xs), torch.cat(ys), torch.cat(ss)


def pairwise_iou(a, b):
    """cxcywh [N,4] x [M,4] -> [N,M] (yolox bboxes_iou, xyxy=False; no epsilon — trap T13)."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[None, :, :2] - b[None, :, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[None, :, :2] + b[None, :, 2:] / 2)
    en = (tl < br).to(a.dtype).prod(2)
    inter = (br - tl).prod(2) * en
    return inter / (a[:, 2:].prod(1)[:, None] + b[:, 2:].prod(1)[None, :] - inter)


def iou_loss(p, t):
    """1 - IoU^2, +1e-16 in the denominator (yolox IOUloss 'iou')."""
    tl = torch.max(p[:, :2] - p[:, 2:] / 2, t[:, :2] - t[:, 2:] / 2)
    br = torch.min(p[:, :2] + p[:, 2:] / 2, t[:, :2] + t[:, 2:] / 2)
    en = (tl < br).to(p.dtype).prod(1)
    inter = (br - tl).prod(1) * en
    iou = inter / (p[:, 2:].prod(1) + t[:, 2:].prod(1) - inter + 1e-16)
    return 1 - iou ** 2


@torch.no_grad()
def simota(gt_boxes, gt_cls, boxes, obj_logit, cls_logit, gx, gy, gs, num_classes):
    """One image.  Returns (fg [A] bool, matched_gt [F], matched_iou [F])."""
    G = gt_boxes.shape[0]
    xc = gx * gs + 0.5 * gs
    yc = gy * gs + 0.5 * gs
    gl = (gt_boxes[:, 0] - 0.5 * gt_boxes[:, 2])[:, None]
    gr = (gt_boxes[:, 0] + 0.5 * gt_boxes[:, 2])[:, None]
    gt_ = (gt_boxes[:, 1] - 0.5 * gt_boxes[:, 3])[:, None]
    gb = (gt_boxes[:, 1] + 0.5 * gt_boxes[:, 3])[:, None]
    in_box = torch.stack([xc[None] - gl, yc[None] - gt_, gr - xc[None], gb - yc[None]], 2).min(2).values > 0.0
    rad = 2.5 * gs[None]
    in_ctr = torch.stack([xc[None] - (gt_boxes[:, 0:1] - rad), yc[None] - (gt_boxes[:, 1:2] - rad),
                          (gt_boxes[:, 0:1] + rad) - xc[None], (gt_boxes[:, 1:2] + rad) - yc[None]], 2).min(2).values > 0.0
    cand = (in_box.sum(0) > 0) | (in_ctr.sum(0) > 0)
    both = in_box[:, cand] & in_ctr[:, cand]
    ious = pairwise_iou(gt_boxes, boxes[cand])
    iou_cost = -torch.log(ious + 1e-8)
    onehot = F.one_hot(gt_cls.to(torch.int64), num_classes).float()
    prob = (torch.sigmoid(cls_logit[cand].float()) * torch.sigmoid(obj_logit[cand].float())[:, None]).sqrt()
    C = prob.shape[0]
    cls_cost = F.binary_cross_entropy(prob[None].expand(G, -1, -1), onehot[:, None, :].expand(-1, C, -1),
                                      reduction="none").sum(-1)
    cost = cls_cost + 3.0 * iou_cost + 100000.0 * (~both)
    topv, _ = torch.topk(ious, min(10, C), dim=1)
    ks = torch.clamp(topv.sum(1).int(), min=1).tolist()
    match = torch.zeros_like(cost)
    for g in range(G):
        _, idx = torch.topk(cost[g], k=ks[g], largest=False)
        match[g, idx] = 1.0
    multi = match.sum(0) > 1
    amin = cost.argmin(0)
    resolved = torch.zeros_like(match)
    resolved[amin, torch.arange(C, device=cost.device)] = 1.0
    match = torch.where(multi[None], resolved, match)
    fg_in = match.sum(0) > 0
    fg = torch.zeros_like(cand)
    fg[cand.nonzero().squeeze(1)[fg_in]] = True
    return fg, match[:, fg_in].argmax(0), (match * ious).sum(0)[fg_in]


def tal_loss(raw, hw_list, labels, support_labels, head):
    """raw [B,A,5+nc] fp32 (requires grad).  Returns the reference's loss dict (yolox.py:39-46)."""
    nc = head.num_classes
    dev = raw.device
    gx, gy, gs = anchor_grid(hw_list, head.strides, dev, raw.dtype)
    boxes = torch.cat([(raw[..., 0:1] + gx[None, :, None]) * gs[None, :, None],
                       (raw[..., 1:2] + gy[None, :, None]) * gs[None, :, None],
                       torch.exp(raw[..., 2:4]) * gs[None, :, None]], 2)
    obj, cls, origin = raw[..., 4], raw[..., 5:], raw[..., :4]
    B, A = raw.shape[:2]
    labels = labels.to(dev).float()
    support_labels = support_labels.to(dev).float()
    nlabel = (labels.sum(2) > 0).sum(1).tolist()
    nsup = (support_labels.sum(2) > 0).sum(1).tolist()
    fg_all = torch.zeros(B, A, dtype=torch.bool, device=dev)
    cls_t, reg_t, l1_t, trend = [], [], [], []
    num_fg, num_gt = 0.0, 0.0
    boxes_d, obj_d, cls_d = boxes.detach(), obj.detach(), cls.detach()
    for i in range(B):
        G, S = int(nlabel[i]), int(nsup[i])
        num_gt += G
        if G == 0:
            continue
        gtb, gtc = labels[i, :G, 1:5], labels[i, :G, 0]
        fg, mg, miou = simota(gtb, gtc, boxes_d[i], obj_d[i], cls_d[i], gx, gy, gs, nc)
        fg_all[i] = fg
        cls_t.append(F.one_hot(gtc[mg].to(torch.int64), nc) * miou[:, None])
        g = gtb[mg]
        reg_t.append(g)
        s_ = gs[fg]
        l1_t.append(torch.stack([g[:, 0] / s_ - gx[fg], g[:, 1] / s_ - gy[fg],
                                 torch.log(g[:, 2] / s_ + 1e-8), torch.log(g[:, 3] / s_ + 1e-8)], 1))
        if S == 0:
            tr = torch.ones(G, device=dev)
        else:
            tr, _ = pairwise_iou(gtb, support_labels[i, :S, 1:5]).max(1)
            tr = torch.where(tr < head.ignore_thr, torch.full_like(tr, head.ignore_value), tr)
        trend.append(tr[mg])

    def _cat(lst, width):
        return torch.cat(lst, 0) if lst else raw.new_zeros((0, width))
    cls_t, reg_t, l1_t = _cat(cls_t, nc), _cat(reg_t, 4), _cat(l1_t, 4)
    trend = torch.cat(trend, 0) if trend else raw.new_zeros((0,))
    fgm = fg_all.view(-1)
    num_fg = float(fgm.sum())
    w = 1 / (trend ** head.gamma + 1e-8)
    iou_l = iou_loss(boxes.reshape(-1, 4)[fgm], reg_t)
    w_iou = ((w * iou_l.sum()) / (w * iou_l).sum()).detach()
    nf = max(num_fg, 1)
    loss_iou = (w_iou * iou_l).sum() / nf
    loss_obj = F.binary_cross_entropy_with_logits(obj.reshape(-1, 1), fgm.to(raw.dtype)[:, None], reduction="none").sum() / nf
    loss_cls = F.binary_cross_entropy_with_logits(cls.reshape(-1, nc)[fgm], cls_t, reduction="none").sum() / nf
    if head.use_l1:
        l1_l = (origin.reshape(-1, 4)[fgm] - l1_t).abs()
        w4 = w[:, None].repeat(1, 4)
        w_l1 = ((w4 * l1_l.sum()) / (w4 * l1_l).sum()).detach()
        loss_l1 = (w_l1 * l1_l).sum() / nf
    else:
        loss_l1 = 0.0
    total = 5.0 * loss_iou + loss_obj + loss_cls + loss_l1
    return {"total_loss": total, "iou_loss": 5.0 * loss_iou, "l1_loss": loss_l1, "conf_loss": loss_obj,
            "cls_loss": loss_cls, "num_fg": num_fg / max(num_gt, 1)}
