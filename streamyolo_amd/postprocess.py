"""postprocess — drop-in for yolox.utils.postprocess as the reference calls it
(exps/evaluators/onex_stream_evaluator.py:148-150) and for the inline `inference()` of
sAP/streamyolo/streamyolo_det.py:62-83: confidence filter + class-aware NMS on the device
(sy_postprocess), one host read of the per-image counts at the end (the reference's `.cpu()`)."""
import torch

from . import ops

_ws_cache = {}


def postprocess_device(prediction, num_classes, conf_thre=0.7, nms_thre=0.45):
    """No host synchronisation: returns (det [B,A,7], anchor_index [B,A], count [B]) device tensors."""
    pred = prediction.float().contiguous()
    key = (pred.shape[0], pred.shape[1], str(pred.device))
    ws = _ws_cache.get(key)
    if ws is None:
        ws = ops.PostprocessWorkspace(pred.shape[0], pred.shape[1], pred.device)
        _ws_cache[key] = ws
    return ops.postprocess(pred, num_classes, conf_thre, nms_thre, ws)


def postprocess(prediction, num_classes, conf_thre=0.7, nms_thre=0.45, class_agnostic=False):
    """List (one entry per image) of [n,7] tensors (x1,y1,x2,y2,obj,class_conf,class_pred) in
    descending score order, or None when nothing survives — as yolox.utils.postprocess returns.
    Like the reference it also rewrites prediction[..., :4] to corner form in place."""
    if class_agnostic:
        raise NotImplementedError("class_agnostic NMS is not used on the StreamYOLO path")
    det, idx, cnt = postprocess_device(prediction, num_classes, conf_thre, nms_thre)
    counts = cnt.cpu().tolist()
    if prediction.dtype == torch.float32 and prediction.is_contiguous() and prediction.dim() == 3:
        ops.head_decode(prediction, boxes=False, corners=True)          # one launch (sy_head_decode, flag 4)
    else:                                                               # a caller's half / strided tensor: its own dtype's arithmetic
        cx, cy, w, h = prediction[..., 0].clone(), prediction[..., 1].clone(), prediction[..., 2].clone(), prediction[..., 3].clone()
        prediction[..., 0], prediction[..., 1] = cx - w / 2, cy - h / 2
        prediction[..., 2], prediction[..., 3] = cx + w / 2, cy + h / 2
    return [det[i, :n].clone().to(prediction.dtype) if n else None for i, n in enumerate(counts)]
