"""Streaming harness + evaluator glue: the callers on either side of the on_pipe hot path.

Mirrors, name for name where the reference has a name:
  inference()                 sAP/streamyolo/streamyolo_det.py:62-83     decoded output -> (boxes / in_scale, scores, labels, None)
  StreamingDetector           streamyolo_det.py:96-124, :176-181         model.half() on_pipe loop state (buffer), warm-up, one frame
  run_sequence()              streamyolo_det.py:138-199                  the real-time scheduling loop of one sequence
  runtime_summary()           streamyolo_det.py:201-229 + sAP/util print_stats   timing statistics
  pair_with_ground_truth()    sAP/streamyolo/streaming_eval.py:74-139    time-based prediction / ground-truth pairing -> COCO dicts
  convert_to_coco_format()    exps/evaluators/onex_stream_evaluator.py:167-209   offline evaluator rows with the +1 frame-id shift

Per-pixel and per-box arithmetic is on the device (sy_frames_u8_pack, the on_pipe plan, sy_postprocess); this file is the
Python control flow the reference also keeps in Python.  Dataset access (pycocotools) and image decoding (cv2.imread)
stay with the caller: every function takes plain lists / arrays.
"""
import os
import time

import numpy as np
import torch

from .data import FramePairsU8
from .postprocess import postprocess_device


def inference(outputs, num_classes=8, conf_thre=0.01, nms_thresh=0.65, in_scale=0.5):
    """`inference(result[0])` of streamyolo_det.py:62-83 for ONE image's decoded rows [A, 5+nc] (or [1, A, 5+nc]):
    returns (bboxes xyxy / in_scale, scores = obj * class_conf, labels int32, None) as numpy arrays in descending
    score order.  Filter + class-aware NMS run in sy_postprocess; one D2H copy of the kept rows."""
    pred = outputs if outputs.dim() == 3 else outputs.unsqueeze(0)
    det, _, cnt = postprocess_device(pred, num_classes, conf_thre, nms_thresh)
    return _parse(det, cnt, in_scale)


def _parse(det, cnt, in_scale):
    """Kept rows [n, 7] of image 0 -> the reference's (bboxes / in_scale, scores, labels, None) host arrays."""
    n = int(cnt[0])
    d = det[0, :n].cpu().numpy()
    return d[:, :4] / in_scale, d[:, 4] * d[:, 5], d[:, 6].astype(np.int32), None


class StreamingDetector:
    """The model-side state of streamyolo_det.py's loop: `model.eval().half()` (:104-109), ten warm-up calls (:113-120),
    then per camera frame `preproc` -> `model(frame, buffer=buffer, mode='on_pipe')` -> `inference` (:176-181), the
    buffer being reset at every sequence start (:150).

    Frames are the raw uint8 HWC arrays `cv2.imread` returns.  The reference resizes every frame to
    (h_img, w_img) = (int(H * in_scale), int(W * in_scale)) with cv2.resize(img, (w_img, h_img)) (:176-178: truncation, each
    axis stretched on its own): with in_scale = 0.5 on even frames that is the exact 2x decimation, with any other in_scale
    OpenCV's fixed-point bilinear resize — both done on the device together with the Focus packing (sy_frames_u8_pack,
    csrc/input_pipeline.hip)."""

    def __init__(self, model, frame_hw, in_scale=0.5, num_classes=8, conf_thre=0.01, nms_thresh=0.65, dtype="fp16",
                 device="cuda"):
        self.model = model.to(device).eval().set_compute_dtype(dtype)
        self.device = torch.device(device)
        self.in_scale = in_scale
        self.frame_hw = (int(frame_hw[0]), int(frame_hw[1]))
        # h_img, w_img = int(1200 * in_scale), int(1920 * in_scale) (streamyolo_det.py:177): truncation, per axis
        self.canvas = (int(self.frame_hw[0] * in_scale), int(self.frame_hw[1] * in_scale))
        if self.canvas[0] % 2 or self.canvas[1] % 2:
            raise ValueError("StreamingDetector: in_scale %r gives an odd %dx%d network input (the Focus stem needs even sizes)"
                             % (in_scale, self.canvas[0], self.canvas[1]))
        if self.canvas == self.frame_hw:
            self.decimate = 1
        elif (2 * self.canvas[0], 2 * self.canvas[1]) == self.frame_hw:
            self.decimate = 2
        else:
            self.decimate = -1                                    # per-axis fixed-point bilinear stretch to the canvas
        self.num_classes, self.conf_thre, self.nms_thresh = num_classes, conf_thre, nms_thresh
        self._slot = torch.empty((1,) + self.frame_hw + (3,), dtype=torch.uint8, device=self.device)
        self._in = FramePairsU8(self._slot, None, self.canvas, self.decimate)
        # 16-bit speed modes: the deep small-map layers may run as split-K (different fp32 summation order than the facade's
        # exact path — a plan object of its own); the fp32 parity mode keeps the single-pass kernels
        split_k = str(dtype) not in ("fp32", "torch.float32") and os.environ.get("STREAMYOLO_STREAM_SPLITK", "1") != "0"
        self.plan = model._plans.inference(model.backbone, model.head, "on_pipe", self._in, owner=model, split_k=split_k)
        self._first = True
        self._post = lambda out: postprocess_device(out, num_classes, conf_thre, nms_thresh)   # one object: part of the tape key

    def warm_up(self, n=10):
        """streamyolo_det.py:113-120 (ten calls on a dummy frame; here they also build / tune the plan)."""
        self._slot.fill_(1)
        with torch.no_grad():
            for i in range(n):
                self.plan.run_stream(self._in, first=(i == 0))
        torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
        self.reset()

    def reset(self):
        """`buffer = None` at the start of a sequence (:150): the next frame fuses with itself (node 'star')."""
        self._first = True

    def __call__(self, frame):
        """One camera frame (uint8 [H, W, 3] numpy array or tensor) -> (bboxes, scores, labels, None), host arrays.
        Ends with the device synchronised (the D2H copy of the detections), as the reference's loop does (:183)."""
        f = torch.as_tensor(frame)
        assert f.dtype == torch.uint8 and tuple(f.shape) == self.frame_hw + (3,)
        self._slot[0].copy_(f, non_blocking=True)
        with torch.no_grad():
            if self._first:                                   # node 'star': extra fan-in copies, through the wrappers
                self._first = False
                return inference(self.plan.run_stream(self._in, first=True), self.num_classes, self.conf_thre,
                                 self.nms_thresh, self.in_scale)
            # steady state: the ~140 launches of the frame (pack, network, decode, NMS) replayed from a launch tape
            det, _, cnt = self.plan.run_stream_taped(self._in, post=self._post, check_params=False)
            return _parse(det, cnt, self.in_scale)


def run_sequence(frames, detect, fps=30.0, det_stride=1, dynamic_schedule=False, clock=time.perf_counter, reset=None):
    """The scheduling loop of one sequence (streamyolo_det.py:138-199): real time runs at `fps`; whenever the detector is
    free it takes the LATEST frame (index floor(t * fps)), skipping it if already seen, if the stride counter says so, or
    — dynamic schedule — if more than half of that frame's interval has passed; results are stamped with their finish
    time.  `detect(frame)` returns the parsed result and must have finished its device work on return.  `clock` is
    injectable (tests drive it with a simulated clock; it is read exactly where the reference reads perf_counter).
    Returns the reference's per-sequence pickle dict (minus `results_raw`, the undecoded device tensors)."""
    duration = len(frames) / fps                     # the sequence's length in real time
    log = {"results_parsed": [], "timestamps": [], "input_fidx": [], "runtime": []}
    seen = None                                      # index of the frame looked at last
    since_det = 0                                    # fixed-stride schedule: frames since the last detection (0 = none yet)
    if reset is not None:
        reset()
    t0 = clock()
    while True:
        began = clock()
        now = began - t0
        if now >= duration:
            break
        pos = now * fps                              # fractional frame position of "now"
        latest = int(np.floor(pos))
        if latest == seen:
            continue                                 # nothing new yet: poll again
        seen = latest
        if dynamic_schedule:
            if pos - latest > 0.5:                   # too far into this frame's interval: wait for the next one
                continue
        else:
            take = since_det % det_stride == 0
            since_det = 1 if take else since_det + 1
            if not take:
                continue
        parsed = detect(frames[latest])
        ended = clock()
        finish = ended - t0
        if finish >= duration:
            break                                    # finished after the sequence ended: not recorded
        log["timestamps"].append(finish)
        log["results_parsed"].append(parsed)
        log["input_fidx"].append(latest)
        log["runtime"].append(ended - began)
    return log


def runtime_summary(runtime_all, n_total, fps=30.0):
    """streamyolo_det.py:201-229: the `time_info.pkl` dict plus print_stats' numbers (sAP/util/__init__.py:13-36) in ms."""
    r = np.asarray(runtime_all, dtype=np.float64)
    out = {"runtime_all": list(runtime_all), "n_processed": int(r.size), "n_total": int(n_total),
           "n_small_runtime": int((r < 1.0 / fps).sum())}
    if r.size > 1:
        out["stats_ms"] = {"mean": 1e3 * r.mean(), "std": 1e3 * r.std(ddof=1), "min": 1e3 * r.min(), "max": 1e3 * r.max()}
    elif r.size == 1:
        out["stats_ms"] = {"scalar": 1e3 * float(r[0])}
    return out


def ltrb2ltwh(bboxes):
    """sAP/util/bbox.py:15-21,74-76 (copying form)."""
    b = np.array(bboxes, copy=True)
    if len(b):
        if b.ndim == 1:
            b[2:] -= b[:2]
        else:
            b[:, 2:] -= b[:, :2]
    return b


def pair_with_ground_truth(results, image_ids, fps=30.0, eta=0.0):
    """streaming_eval.py:74-139 for one sequence: ground-truth frame number n (time (n - eta) / fps) is paired with the
    LAST result whose timestamp is <= that time.  `results` = run_sequence's dict, `image_ids` = the dataset ids of the
    sequence's frames in order.  Returns (coco_rows, {"in_time", "miss", "mismatch"})."""
    stamps, sources, parsed = results["timestamps"], results["input_fidx"], results["results_parsed"]
    rows = []
    stats = {"in_time": 0, "miss": 0, "mismatch": 0}
    ready = 0                                        # results finished by the current query time (monotone cursor)
    for n, image_id in enumerate(image_ids):
        query = (n - eta) / fps
        while ready < len(stamps) and stamps[ready] <= query:
            ready += 1
        if ready == 0:                               # nothing has been produced yet
            stats["miss"] += 1
            continue
        src = sources[ready - 1]
        stats["in_time"] += int(src == n)
        stats["mismatch"] += n - src
        boxes, scores, labels, masks = parsed[ready - 1][:4]
        if not len(boxes):
            continue
        ltwh = ltrb2ltwh(boxes)
        for j in range(len(boxes)):
            row = {"image_id": image_id, "bbox": ltwh[j], "score": scores[j], "category_id": labels[j]}
            if masks is not None:
                row["segmentation"] = masks[j]
            rows.append(row)
    return rows, stats


def convert_to_coco_format(outputs, info_imgs, ids, img_size, class_ids, images, skip_ids=(15060, 15061)):
    """ONEX_COCOEvaluator.convert_to_coco_format (exps/evaluators/onex_stream_evaluator.py:167-209), dataset access
    replaced by plain arguments: `class_ids` = dataset.class_ids, `images` = coco.dataset['images'] (each with 'fid').
    A detection made on frame t is scored against frame t+1 (`image_id = img_id + 1`); frames whose successor starts a
    new sequence are dropped, and so — exactly as in the reference, whose append sits inside the final `else` — are
    the detections of every sequence's first frame, and the two hard-coded ids."""
    rows = []
    for det, h, w, img_id in zip(outputs, info_imgs[0], info_imgs[1], ids):
        if det is None:
            continue
        t = int(img_id)
        if t in skip_ids or images[t + 1]["fid"] == 0 or images[t]["fid"] == 0:
            continue                                 # (reference :183-187: no row is ever emitted for these frames)
        det = det.cpu()
        r = min(img_size[0] / float(h), img_size[1] / float(w))      # undo the letterbox scale
        xywh = det[:, 0:4] / r
        xywh[:, 2:4] -= xywh[:, 0:2]                                  # yolox.utils.xyxy2xywh
        conf = det[:, 4] * det[:, 5]
        for k in range(det.shape[0]):
            rows.append({"image_id": t + 1, "category_id": class_ids[int(det[k, 6])], "bbox": xywh[k].numpy().tolist(),
                         "score": conf[k].numpy().item(), "segmentation": []})
    return rows
