"""Device-side input pipeline: the host mirror of what the reference does between
`cv2.imread` and `model(inps, targets)`, with the per-pixel work in ONE HIP launch (csrc/input_pipeline.hip).

Reference path (per sample, CPU workers, then the GPU):
    `_mirror` + `preproc` (exps/data/data_augment_flip.py:140-167)  ->  fp32 CHW on a 114 canvas
    `np.concatenate((img, support_img))` (exps/data/tal_flip_mosaicdetection.py:257)  ->  fp32 [6,H,W]
    `DataPrefetcher` H2D on a side stream (exps/train_utils/double_data_prefetcher.py)  ->  13.8 MB per 600x960 pair
    `Exp.preprocess(inps, targets, tsize)` (cfgs/l_s50_onex_dfp_tal_filp.py:161-172)  ->  bilinear multi-scale resize
Here the uint8 HWC frames are uploaded as they are (3.5 MB per pair) and `FramePairsU8` carries them, with the mirror
flags and the target size, to the plan's stem, where `sy_frames_u8_pack` writes the Focus-packed operand directly.
Box targets are scaled by the same torch arithmetic as the reference (tiny tensors — glue).

There is no CPU implementation here: every entry point needs the HIP library.
"""
import torch

from . import ops


class FramePairsU8:
    """A batch of (current, support) uint8 HWC frames on the device, standing in for the reference's fp32
    [B, 6, H, W] input tensor (`mode='off_pipe'`) or, with `sup=None`, its [B, 3, H, W] on_pipe frame.

    cur, sup : uint8 [B, Hs, Ws, 3] (BGR, as cv2.imread delivers them), same strides
    canvas   : (H, W) letterbox size = exp.input_size / exp.test_size; the frames sit top-left on a 114 canvas
    decimate : 1, or 2 for the exact 2x load-time resize (Argoverse 1200x1920 -> 600x960), or 0 for any other camera
               size: `preproc`'s r = min(H / h, W / w) bilinear letterbox resize (OpenCV fixed-point arithmetic), or -1
               for the streaming detector's per-axis stretch to exactly the canvas (streamyolo_det.py:176-178)
    mirror   : None or uint8 [B]; flag b flips BOTH frames of pair b horizontally (DoubleTrainTransform)
    out_size : (Ho, Wo) the plan runs at (Exp.preprocess's tsize); defaults to the canvas

    Quacks like the tensor it replaces where the model facades look at their input (`shape`, `size()`, `dtype`,
    `device`, `is_cuda`, `float()`, `contiguous()`)."""

    def __init__(self, cur, sup, canvas, decimate=1, mirror=None, out_size=None):
        assert cur.dtype == torch.uint8 and cur.dim() == 4 and cur.shape[-1] == 3
        self.cur, self.sup = cur.contiguous(), (None if sup is None else sup.contiguous())
        self.canvas = (int(canvas[0]), int(canvas[1]))
        self.decimate = int(decimate)
        self.mirror = None if mirror is None else mirror.to(device=cur.device, dtype=torch.uint8).contiguous()
        self.out_size = self.canvas if out_size is None else (int(out_size[0]), int(out_size[1]))

    # ---- tensor look-alike -------------------------------------------------------------------------------
    @property
    def shape(self):
        return torch.Size((self.cur.shape[0], 3 if self.sup is None else 6, self.out_size[0], self.out_size[1]))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    dtype = torch.float32

    @property
    def device(self):
        return self.cur.device

    @property
    def is_cuda(self):
        return self.cur.is_cuda

    def float(self):
        return self

    def contiguous(self):
        return self

    def resized(self, tsize):
        """Same frames, run at `tsize` (the bilinear resize happens inside the pack kernel)."""
        return FramePairsU8(self.cur, self.sup, self.canvas, self.decimate, self.mirror, tsize)

    def paired_with_self(self):
        """3-channel input on the off_pipe path: the reference duplicates it (dfp_pafpn.py:236-238)."""
        return self if self.sup is not None else FramePairsU8(self.cur, self.cur, self.canvas, self.decimate,
                                                             self.mirror, self.out_size)

    # ---- materialisation ---------------------------------------------------------------------------------
    def to_nchw(self):
        """The reference's own input tensor: fp32 [B, 3|6, Ho, Wo] (drop-in for any consumer of `inps`)."""
        out = torch.empty(tuple(self.shape), dtype=torch.float32, device=self.device)
        ops.frames_u8_pack(self.cur, self.sup, self.canvas, self.out_size, out, None, self.decimate, self.mirror)
        return out

    def pack_focus(self, cur_view, sup_view=None):
        """Write the stem operand(s) (Focus-packed NHWC16 Views of the plan) straight from the uint8 frames."""
        assert (sup_view is None) == (self.sup is None)
        ops.frames_u8_pack(self.cur, self.sup, self.canvas, self.out_size, cur_view, sup_view, self.decimate, self.mirror)


def preprocess(inputs, targets, tsize, input_size):
    """`Exp.preprocess(inputs, targets, tsize)` (cfgs/l_s50_onex_dfp_tal_filp.py:161-172) with `self.input_size`
    passed explicitly: resize the batch to `tsize` (bilinear, align_corners=False) and scale both target tensors
    IN PLACE (x columns 1::2 by scale_x, y columns 2::2 by scale_y), or return everything unchanged when
    tsize == input_size.  `inputs`: the reference's fp32 [B,6,H,W] tensor (resized by sy_resize_bilinear_nchw) or a
    FramePairsU8 (the resize is deferred into the pack kernel)."""
    scale_y = tsize[0] / input_size[0]
    scale_x = tsize[1] / input_size[1]
    if scale_x != 1 or scale_y != 1:
        if isinstance(inputs, FramePairsU8):
            inputs = inputs.resized(tsize)
        else:
            x = inputs.float().contiguous()
            out = torch.empty((x.shape[0], x.shape[1], int(tsize[0]), int(tsize[1])), dtype=torch.float32, device=x.device)
            ops.resize_bilinear_nchw(x, out)
            inputs = out.to(inputs.dtype)
        targets[0][..., 1::2] = targets[0][..., 1::2] * scale_x
        targets[0][..., 2::2] = targets[0][..., 2::2] * scale_y
        targets[1][..., 1::2] = targets[1][..., 1::2] * scale_x
        targets[1][..., 2::2] = targets[1][..., 2::2] * scale_y
    return inputs, targets


class DevicePrefetcher:
    """`DataPrefetcher` (exps/train_utils/double_data_prefetcher.py:8-56) for uint8 frame batches: the next batch's
    H2D copies run on a side stream while the current step computes; `next()` makes the current stream wait for
    them and pins their memory to it (`record_stream`), exactly as the reference hands its fp32 batch over.

    `loader` yields ((cur_u8 [B,Hs,Ws,3], sup_u8, mirror [B] or None), (labels, support_labels), ...)."""

    def __init__(self, loader, canvas, decimate=1, device="cuda"):
        self.loader = iter(loader)
        self.canvas, self.decimate, self.device = canvas, decimate, torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        self.preload()

    def preload(self):
        try:
            item = next(self.loader)
        except StopIteration:
            self.next_input = self.next_target = None
            return
        (cur, sup, mirror), target = item[0], item[1]
        with torch.cuda.stream(self.stream):
            cur = cur.to(self.device, non_blocking=True)
            sup = None if sup is None else sup.to(self.device, non_blocking=True)
            mirror = None if mirror is None else torch.as_tensor(mirror, dtype=torch.uint8).to(self.device, non_blocking=True)
            self.next_input = FramePairsU8(cur, sup, self.canvas, self.decimate, mirror)
            self.next_target = (target[0].to(self.device, non_blocking=True), target[1].to(self.device, non_blocking=True))

    def next(self):
        cs = torch.cuda.current_stream(self.device)
        cs.wait_stream(self.stream)
        inp, target = self.next_input, self.next_target
        if inp is not None:
            for t in (inp.cur, inp.sup, inp.mirror):
                if t is not None:
                    t.record_stream(cs)
        if target is not None:
            target[0].record_stream(cs)
            target[1].record_stream(cs)
        self.preload()
        return inp, target
