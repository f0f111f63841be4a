"""Thin typed wrappers over the C ABI: torch tensors in, raw pointers + the current HIP stream out.

PyTorch is plumbing here (device memory, streams); every numeric result comes from the kernels in
streamyolo_amd/csrc through libstreamyolo_hip.so.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (CONV_DGRAD, CONV_FWD, DT_BF16, DT_F16, DT_F32, EPI_BNR, EPI_DECODE, EPI_LINEAR, EPI_SIGMOID,
                   EPI_SILU, ConvDesc, WgradDesc, check)

TORCH_DTYPE = {DT_BF16: torch.bfloat16, DT_F16: torch.float16, DT_F32: torch.float32}
DTYPE_CODE = {v: k for k, v in TORCH_DTYPE.items()}
DTYPE_NAME = {"bf16": DT_BF16, "fp16": DT_F16, "f16": DT_F16, "fp32": DT_F32, "f32": DT_F32}


def dtype_code(d):
    if isinstance(d, str):
        return DTYPE_NAME[d]
    if isinstance(d, torch.dtype):
        return DTYPE_CODE[d]
    return int(d)


def stream_of(t):
    """Raw hipStream_t of torch's CURRENT stream on t's device (kernels must run on
    the stream the trainer/prefetcher handed the tensors over on)."""
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


class View:
    """NHWC activation view: a channel slice [c_off, c_off+C) of a dense [N,H,W,ld] buffer."""
    __slots__ = ("buf", "N", "H", "W", "C", "ld", "c_off", "bs_", "root")

    def __init__(self, buf, N, H, W, C_, ld=None, c_off=0, bs=None, root=None):
        self.buf, self.N, self.H, self.W, self.C = buf, N, H, W, C_
        self.ld = C_ if ld is None else ld
        self.c_off = c_off            # element offset of the view's first element inside `buf`
        self.bs_ = bs                 # explicit batch stride (elements) when images are not densely packed
        self.root = root              # (full tensor, first image) when `buf` is a batch slice of a frame-pair buffer

    @staticmethod
    def alloc(N, H, W, C_, dtype, device, zero=False):
        fn = torch.zeros if zero else torch.empty
        return View(fn((N, H, W, C_), dtype=TORCH_DTYPE[dtype_code(dtype)], device=device), N, H, W, C_)

    @property
    def bs(self):
        return self.H * self.W * self.ld if self.bs_ is None else self.bs_

    @property
    def dtype(self):
        return DTYPE_CODE[self.buf.dtype]

    @property
    def pixels(self):
        return self.N * self.H * self.W

    def ptr(self):
        return self.buf.data_ptr() + self.c_off * self.buf.element_size()

    def bytes_from_ptr(self):
        """Bytes of the underlying allocation addressable from ptr() (buffer bounds for the kernels)."""
        st = self.buf.untyped_storage()
        return st.nbytes() - (self.ptr() - st.data_ptr())

    def slice(self, c0, c):
        assert 0 <= c0 and c0 + c <= self.C
        return View(self.buf, self.N, self.H, self.W, c, self.ld, self.c_off + c0, self.bs_, self.root)

    def like(self, zero=False):
        """A fresh dense buffer with the same full layout (used for gradient mirrors)."""
        fn = torch.zeros_like if zero else torch.empty_like
        return View(fn(self.buf), self.N, self.H, self.W, self.C, self.ld, self.c_off)

    def pair(self):
        """Both frames of a frame-pair buffer as ONE view of 2N images (self must be the first frame's slice)."""
        full, n0 = self.root
        assert n0 == 0 and full.shape[0] == 2 * self.N
        return View(full, 2 * self.N, self.H, self.W, self.C, self.ld, self.c_off, self.bs_)

    def export(self):
        """A COPY of the view as an NCHW-shaped tensor in the storage dtype (channels-last memory): what the drop-in modules hand
        back to their caller.  Honours the view's channel slice and row pitch — a fused feature that lives inside a wider
        concatenation buffer exports only its own channels."""
        assert self.bs_ is None
        return self.buf.view(self.N, self.H, self.W, self.ld)[..., self.c_off:self.c_off + self.C] \
            .clone(memory_format=torch.contiguous_format).permute(0, 3, 1, 2)

    def nchw(self):
        """Float32 NCHW copy (tests / debugging only)."""
        return self.buf.view(self.N, self.H, self.W, self.ld)[..., self.c_off:self.c_off + self.C] \
            .permute(0, 3, 1, 2).float().contiguous()

    def set_nchw(self, t):
        self.buf.view(self.N, self.H, self.W, self.ld)[..., self.c_off:self.c_off + self.C] = \
            t.permute(0, 2, 3, 1).to(self.buf.dtype)


def _p(t):
    return None if t is None else t.data_ptr()


# ---- weights epoch ---------------------------------------------------------------------------------------------
# Kernels that rewrite parameters / BatchNorm buffers through raw pointers (sy_sgd_ema_step, sy_bn_running_update) do
# not bump torch's tensor._version, which is what the inference-side caches (engine.ParamCache, the streaming launch
# tape) watch.  They bump this counter instead and the caches include it in their signatures.
_weights_epoch = [0]


def weights_epoch():
    return _weights_epoch[0]


def bump_weights_epoch():
    _weights_epoch[0] += 1


def conv_out_size(h, k, stride):
    return (h + 2 * ((k - 1) // 2) - k) // stride + 1


def conv2d(x, w, y, ksize, stride, scale=None, shift=None, res=None, epilogue=EPI_LINEAR, mode=CONV_FWD,
           accumulate=False, stats=None, dec_stride=0.0, y_f32=False, y_ptr=None, y_ld=None, y_bs=None,
           cout=None, tile=0, wfrag=None, segments=1, k_splits=0, pre=None, bn_reduce=None):
    """One launch of sy_conv2d.  x, y, res: View;  w: packed weight tensor [Cout, k*k*Cin] in x's dtype.
    y_ptr/y_ld/y_bs/cout override the output addressing (head predictions write into [B,A,5+nc]).
    bn_reduce (data gradient, first write): (z View, scale, shift, sums) of the layer that produced this launch's output
    activation — its BatchNorm-backward reduce rides in the epilogue (SY_EPI_BNR; sums [copies][2][C], row 1 = the raw moment)."""
    if bn_reduce is not None:
        z, scale, shift, sums = bn_reduce
        assert mode == CONV_DGRAD and not accumulate and res is None and stats is None and z.ld == y.ld and z.bs == y.bs
        res, epilogue, stats = z, EPI_BNR, (sums, sums)
    d = ConvDesc()
    d.x, d.w = x.ptr(), w.data_ptr()
    d.scale, d.shift = _p(scale), _p(shift)
    d.res = None if res is None else res.ptr()
    d.stat_sum, d.stat_sqsum = (None, None) if stats is None else (stats[0].data_ptr(), stats[1].data_ptr())
    d.stat_copies = 1 if stats is None else max(1, stats[0].numel() // ((2 if bn_reduce is not None else 1) * segments * (y.C if cout is None else cout)))
    d.stat_segments = segments          # statistics arrays [segments][copies][Cout] (frame pairs: one segment per frame)
    d.N, d.H, d.W, d.Cin = x.N, x.H, x.W, x.C
    if y is not None:
        d.y, d.Ho, d.Wo, d.Cout = y.ptr(), y.H, y.W, y.C
        d.ldy, d.ybs = y.ld, y.bs
    if y_ptr is not None:
        d.y, d.ldy, d.ybs = y_ptr, y_ld, y_bs
        if y is None:
            d.Ho, d.Wo = conv_out_size(x.H, ksize, stride), conv_out_size(x.W, ksize, stride)
    if cout is not None:
        d.Cout = cout
    d.KH = d.KW = ksize
    d.stride, d.pad = stride, (ksize - 1) // 2
    d.ldx, d.xbs = x.ld, x.bs
    d.ldr, d.rbs = (0, 0) if res is None else (res.ld, res.bs)
    d.dtype = x.dtype
    d.y_f32 = 1 if y_f32 else 0
    d.mode, d.epilogue, d.accumulate = mode, epilogue, 1 if accumulate else 0
    d.dec_stride = float(dec_stride)
    d.tile = int(tile)
    d.x_bytes = x.bytes_from_ptr()
    d.w_bytes = w.numel() * w.element_size()
    if wfrag is not None:
        d.wfrag, d.wfrag_bytes = wfrag.data_ptr(), wfrag.numel() * wfrag.element_size()
    d.k_splits = int(k_splits)          # > 1: fp32 partial sums per channel-slab range (see sy_splitk_epilogue)
    if pre is not None:                 # fused Bottleneck (tile 119): (1x1 fragment-packed weights, scale, shift); x = the 1x1 conv's input
        pw, ps, ph = pre
        d.pre_w, d.pre_w_bytes, d.pre_scale, d.pre_shift = pw.data_ptr(), pw.numel() * pw.element_size(), ps.data_ptr(), ph.data_ptr()
        d.pre_cin = x.C
        d.Cin = ps.numel()              # the 3x3 convolution's input width = the hidden channels
    check(_lib.lib().sy_conv2d(C.byref(d), stream_of(x.buf)), "sy_conv2d")


def conv2d_wgrad(x, dy, dw, ksize, stride, oihw=False, workspace=None, tile=0, target_blocks=0):
    """dw fp32 += wgrad(x, dy): [Cout, k*k*Cin] packed layout, or the OIHW parameter layout.
    workspace: optional uint8/fp32 device tensor for the split-K slabs (more parallelism on big layers)."""
    d = WgradDesc()
    d.x, d.dy, d.dw = x.ptr(), dy.ptr(), dw.data_ptr()
    d.N, d.H, d.W, d.Cin = x.N, x.H, x.W, x.C
    d.Ho, d.Wo, d.Cout = dy.H, dy.W, dy.C
    d.KH = d.KW = ksize
    d.stride, d.pad = stride, (ksize - 1) // 2
    d.ldx, d.lddy, d.xbs, d.dybs = x.ld, dy.ld, x.bs, dy.bs
    d.dtype = x.dtype
    d.dw_oihw = 1 if oihw else 0
    d.tile, d.target_blocks = int(tile), int(target_blocks)
    d.x_bytes, d.dy_bytes = x.bytes_from_ptr(), dy.bytes_from_ptr()
    if workspace is not None:
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
    check(_lib.lib().sy_conv2d_wgrad(C.byref(d), stream_of(x.buf)), "sy_conv2d_wgrad")


def conv2d_splitk(x, w, y, ksize, stride, scale, shift, part, splits, res=None, epilogue=EPI_SILU, tile=118, wfrag=None):
    """3x3 stride-1 convolution as `splits` partial contractions (fp32 partials in `part`, >= splits * pixels * Cout floats) +
    sy_splitk_epilogue: for layers whose pixel x channel tiling yields far fewer workgroups than the chip has CUs."""
    assert part.dtype == torch.float32 and part.numel() >= splits * y.pixels * y.C
    pv = View(part, splits * y.N, y.H, y.W, y.C)             # partial z = "images" [z*N, (z+1)*N) of a dense fp32 tensor
    d = dict(y_f32=True, y_ptr=part.data_ptr(), y_ld=y.C, y_bs=y.H * y.W * y.C)
    conv2d(x, w, pv, ksize, stride, epilogue=EPI_LINEAR, tile=tile, wfrag=wfrag, k_splits=splits, **d)
    check(_lib.lib().sy_splitk_epilogue(part.data_ptr(), splits, y.pixels, y.C, _p(scale), _p(shift),
                                        None if res is None else res.ptr(), 0 if res is None else res.ld, y.ptr(), y.ld, y.dtype,
                                        epilogue, stream_of(y.buf)), "sy_splitk_epilogue")


def focus_pack(frames, c0, out):
    """frames: [N, Ctot, H, W] fp32 contiguous; out: View [N, H/2, W/2, 16]."""
    assert frames.dtype == torch.float32 and frames.is_contiguous()
    N, Ct, H, W = frames.shape
    assert out.C == 16 and out.ld == 16 and out.c_off == 0
    check(_lib.lib().sy_focus_pack(frames.data_ptr(), N, Ct, c0, H, W, out.ptr(), out.dtype, stream_of(frames)),
          "sy_focus_pack")


FRAMES_NCHW, FRAMES_FOCUS = 0, 1


def frames_u8_pack(cur, sup, canvas, out_size, out_cur, out_sup=None, decimate=1, mirror=None):
    """cur / sup: uint8 [B, Hs, Ws, 3] device tensors (sup None = single frame).  canvas = (H, W) letterbox size
    (exp.input_size / test_size), out_size = (Ho, Wo).  out_cur: fp32 tensor [B, 3|6, Ho, Wo] (NCHW layout) or a
    Focus-packed View [B, Ho/2, Wo/2, 16]; then out_sup is the support frame's View."""
    assert cur.dtype == torch.uint8 and cur.dim() == 4 and cur.shape[3] == 3 and cur.stride(3) == 1 and cur.stride(2) == 3
    B, Hs, Ws, _ = cur.shape
    if sup is not None:
        assert sup.dtype == torch.uint8 and sup.shape == cur.shape and sup.stride() == cur.stride()
    if mirror is not None:
        assert mirror.dtype == torch.uint8 and mirror.numel() == B and mirror.is_contiguous()
    H, W = canvas
    Ho, Wo = out_size
    if isinstance(out_cur, View):
        layout, dt = FRAMES_FOCUS, out_cur.dtype
        for v in (out_cur, out_sup):
            assert v is None or (v.C == 16 and v.ld == 16 and v.c_off == 0 and (v.N, v.H, v.W) == (B, Ho // 2, Wo // 2))
        po, ps = out_cur.ptr(), (out_sup.ptr() if out_sup is not None else None)
    else:
        layout, dt = FRAMES_NCHW, DT_F32
        assert out_cur.dtype == torch.float32 and out_cur.is_contiguous()
        assert tuple(out_cur.shape) == (B, 3 if sup is None else 6, Ho, Wo)
        po, ps = out_cur.data_ptr(), None
    check(_lib.lib().sy_frames_u8_pack(cur.data_ptr(), _p(sup), B, Hs, Ws, cur.stride(0), cur.stride(1), int(decimate),
                                       _p(mirror), H, W, Ho, Wo, layout, po, ps, dt, stream_of(cur)),
          "sy_frames_u8_pack")


def resize_bilinear_nchw(x, out):
    """fp32 NCHW [N, 3|6, H, W] -> out [N, C, Ho, Wo]: torch bilinear, align_corners=False (Exp.preprocess)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and out.dtype == torch.float32 and out.is_contiguous()
    N, Cc, H, W = x.shape
    assert out.shape[0] == N and out.shape[1] == Cc
    check(_lib.lib().sy_resize_bilinear_nchw(x.data_ptr(), N, Cc, H, W, out.data_ptr(), out.shape[2], out.shape[3],
                                             stream_of(x)), "sy_resize_bilinear_nchw")


def resize_nearest(src, dst):
    assert src.C == dst.C and src.N == dst.N
    check(_lib.lib().sy_resize_nearest(src.ptr(), src.N, src.H, src.W, src.C, src.ld, src.bs, dst.ptr(), dst.H,
                                       dst.W, dst.ld, dst.bs, src.dtype, stream_of(src.buf)), "sy_resize_nearest")


def resize_nearest_bwd(ddst, dsrc, accumulate):
    check(_lib.lib().sy_resize_nearest_bwd(ddst.ptr(), ddst.N, ddst.H, ddst.W, ddst.C, ddst.ld, ddst.bs, dsrc.ptr(),
                                           dsrc.H, dsrc.W, dsrc.ld, dsrc.bs, 1 if accumulate else 0, ddst.dtype,
                                           stream_of(ddst.buf)), "sy_resize_nearest_bwd")


def spp_pool(v, argmax=None):
    """v: View of the 4C-wide SPP concat buffer (C = v.C // 4 channels already hold x).
    argmax: optional uint8 tensor [N,H,W,3,C] recording the pooled maxima's window offsets (training)."""
    c = v.C // 4
    check(_lib.lib().sy_spp_pool(v.ptr(), v.N, v.H, v.W, c, v.ld, v.bs, _p(argmax), v.dtype, stream_of(v.buf)),
          "sy_spp_pool")


def spp_pool_bwd(dv, argmax):
    c = dv.C // 4
    check(_lib.lib().sy_spp_pool_bwd(dv.ptr(), argmax.data_ptr(), dv.N, dv.H, dv.W, c, dv.ld, dv.bs, dv.dtype,
                                     stream_of(dv.buf)), "sy_spp_pool_bwd")


def view_copy(src, dst, accumulate=False):
    assert src.C == dst.C and src.pixels == dst.pixels
    check(_lib.lib().sy_view_copy(src.ptr(), src.ld, dst.ptr(), dst.ld, src.pixels, src.C, src.dtype,
                                  1 if accumulate else 0, stream_of(src.buf)), "sy_view_copy")


def rows_add_f32(dst, src, rows, cols, ldd, lds, zero_src=False):
    """dst[r][:cols] += src[r][:cols] (fp32, row pitches ldd / lds in elements); zero_src: clear the source behind the read."""
    assert dst.dtype == torch.float32 and src.dtype == torch.float32
    check(_lib.lib().sy_rows_add_f32(dst.data_ptr(), ldd, src.data_ptr(), lds, rows, cols, 1 if zero_src else 0, stream_of(dst)),
          "sy_rows_add_f32")


def pred_grad_fold(d_raw, a0, rows, num_classes, scratch, cin, g_reg, g_obj, g_cls, gb_reg, gb_obj, gb_cls, workspace):
    """Parameter gradients of one head level's prediction convs from d_raw[:, a0:a0+rows] and the wgrad scratch (see the header)."""
    B, A, nch = d_raw.shape
    assert d_raw.dtype == torch.float32 and d_raw.is_contiguous() and nch == 5 + num_classes
    assert scratch.dim() == 3 and scratch.shape[0] == 2 and scratch.is_contiguous()
    assert workspace.numel() >= _lib.lib().sy_pred_grad_fold_workspace_floats(num_classes)
    check(_lib.lib().sy_pred_grad_fold(d_raw.data_ptr() + a0 * nch * 4, B, A * nch, rows, num_classes, scratch.data_ptr(),
                                       scratch.shape[1], scratch.shape[2], cin, g_reg.data_ptr(), g_obj.data_ptr(), g_cls.data_ptr(),
                                       gb_reg.data_ptr(), gb_obj.data_ptr(), gb_cls.data_ptr(), workspace.data_ptr(),
                                       stream_of(d_raw)), "sy_pred_grad_fold")


def bn_finalize(ssum, ssq, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd,
                nseg=1):
    """count = elements per channel PER SEGMENT; statistics [nseg][copies][C], outputs [nseg][C] (see the header)."""
    C_ = gamma.numel()
    check(_lib.lib().sy_bn_finalize(ssum.data_ptr(), ssq.data_ptr(), C_, ssum.numel() // (nseg * C_), float(count),
                                    gamma.data_ptr(), beta.data_ptr(), float(eps), float(momentum), _p(running_mean),
                                    _p(running_var), scale.data_ptr(), shift.data_ptr(), _p(mean), _p(invstd), nseg,
                                    stream_of(ssum)), "sy_bn_finalize")


def bn_finalize_apply(ssum, ssq, count, gamma, beta, eps, scale, shift, mean, invstd, y, out, res=None, nseg=1):
    """sy_bn_finalize + sy_bn_silu_apply in one launch (training forward); scale / shift / mean / invstd are outputs."""
    C_ = gamma.numel()
    check(_lib.lib().sy_bn_finalize_apply(ssum.data_ptr(), ssq.data_ptr(), ssum.numel() // (nseg * C_), float(count),
                                          gamma.data_ptr(), beta.data_ptr(), float(eps), scale.data_ptr(), shift.data_ptr(),
                                          mean.data_ptr(), invstd.data_ptr(), y.ptr(), y.ld,
                                          None if res is None else res.ptr(), 0 if res is None else res.ld, out.ptr(), out.ld,
                                          y.pixels // nseg, y.C, y.dtype, nseg, stream_of(y.buf)), "sy_bn_finalize_apply")


class BnRunningTable:
    """Device table for sy_bn_running_update.  `modules` = [(bn_module, [(sum, sqsum, count[, copies, ld]), ...calls in order])];
    sum / sqsum start at the module's first channel; ld = replica pitch when they are a channel slice of a wider array."""

    def __init__(self, modules, device, count_batches=False):
        arr = (_lib.BnRunningEntry * len(modules))()
        self.max_c, self.keep = 0, []
        for e, (bn, calls) in zip(arr, modules):
            assert 1 <= len(calls) <= 2
            e.running_mean, e.running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
            for j, call in enumerate(calls):
                ssum, ssq, count = call[:3]
                e.sum[j], e.sqsum[j], e.count[j] = ssum.data_ptr(), ssq.data_ptr(), float(count)
            e.C = bn.num_features
            e.copies = calls[0][3] if len(calls[0]) > 3 else calls[0][0].numel() // e.C
            e.ld = calls[0][4] if len(calls[0]) > 4 else 0
            e.calls = len(calls)
            e.momentum = bn.momentum if bn.momentum is not None else 0.1
            nbt = getattr(bn, "num_batches_tracked", None)           # += calls inside the launch (no torch._foreach_add_ per step)
            if nbt is not None and count_batches:
                assert nbt.dtype == torch.int64
                e.num_batches_tracked = nbt.data_ptr()
                self.keep.append((bn.running_mean, bn.running_var, nbt))
            else:
                self.keep.append((bn.running_mean, bn.running_var))
            self.max_c = max(self.max_c, e.C)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.table = raw.to(device)
        self.n = len(modules)
        self.sig = tuple(t.data_ptr() for pair in self.keep for t in pair)

    def valid(self):
        return self.sig == tuple(t.data_ptr() for pair in self.keep for t in pair)

    def run(self):
        check(_lib.lib().sy_bn_running_update(self.table.data_ptr(), self.n, self.max_c, stream_of(self.table)),
              "sy_bn_running_update")
        bump_weights_epoch()                       # running statistics changed behind torch's back


BN_CAP_NAMES = ("apply", "finalize_apply", "bwd_reduce", "bwd_apply")


def bn_grid_caps(**set_):
    """sy_bn_grid_caps: set (keywords of BN_CAP_NAMES, > 0) and return the workgroup caps of the BatchNorm row kernels in force."""
    assert all(k in BN_CAP_NAMES for k in set_), set_
    s4 = (C.c_int32 * 4)(*[int(set_.get(k, 0)) for k in BN_CAP_NAMES])
    g4 = (C.c_int32 * 4)()
    check(_lib.lib().sy_bn_grid_caps(s4, g4), "sy_bn_grid_caps")
    return dict(zip(BN_CAP_NAMES, [int(v) for v in g4]))


def zero(t):
    """t (a torch tensor: dense, or a 2-D strided view whose rows are dense) := 0 through sy_zero_rows — a launch the plan's tape
    records, not an ATen fill."""
    esz = t.element_size()
    if t.is_contiguous():
        rows, row_b, pitch_b = 1, t.numel() * esz, t.numel() * esz
    else:
        assert t.dim() == 2 and t.stride(1) == 1, "zero(): dense tensor or a row-strided 2-D view"
        rows, row_b, pitch_b = t.shape[0], t.shape[1] * esz, t.stride(0) * esz
    if rows == 0 or row_b == 0:
        return
    check(_lib.lib().sy_zero_rows(t.data_ptr(), rows, row_b, pitch_b, stream_of(t)), "sy_zero_rows")


def bn_silu_apply(y, scale, shift, out, res=None, nseg=1):
    check(_lib.lib().sy_bn_silu_apply(y.ptr(), y.ld, scale.data_ptr(), shift.data_ptr(),
                                      None if res is None else res.ptr(), 0 if res is None else res.ld, out.ptr(),
                                      out.ld, y.pixels // nseg, y.C, y.dtype, nseg, stream_of(y.buf)), "sy_bn_silu_apply")


def bn_silu_bwd_reduce(y, da, scale, shift, mean, invstd, sums, nseg=1):
    check(_lib.lib().sy_bn_silu_bwd_reduce(y.ptr(), y.ld, da.ptr(), da.ld, scale.data_ptr(), shift.data_ptr(),
                                           mean.data_ptr(), invstd.data_ptr(), sums.data_ptr(),
                                           sums.numel() // (2 * y.C * nseg), y.pixels // nseg, y.C, y.dtype, nseg,
                                           stream_of(y.buf)), "sy_bn_silu_bwd_reduce")


def bn_silu_bwd_apply(y, da, scale, shift, mean, invstd, gamma, sums, dy, dgamma=None, dbeta=None, nseg=1,
                      dres=None, dres_accumulate=False, atomic_param_grads=False, raw_moment=False):
    """dres: gradient View of the residual input (y = silu(bn(conv)) + res): written (or accumulated) with da in this pass.
    raw_moment: row 1 of `sums` is sum dz*y (left by a data gradient's fused reduce, conv2d(bn_reduce=...)), not sum dz*xhat."""
    assert dres is None or (dres.C == y.C and dres.pixels == y.pixels)
    check(_lib.lib().sy_bn_silu_bwd_apply(y.ptr(), y.ld, da.ptr(), da.ld, scale.data_ptr(), shift.data_ptr(),
                                          mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), sums.data_ptr(),
                                          sums.numel() // (2 * y.C * nseg), dy.ptr(), dy.ld, y.pixels // nseg, y.C,
                                          _p(dgamma), _p(dbeta), None if dres is None else dres.ptr(),
                                          0 if dres is None else dres.ld,
                                          (1 if dres_accumulate else 0) | (2 if atomic_param_grads else 0) | (4 if raw_moment else 0), y.dtype, nseg,
                                          stream_of(y.buf)), "sy_bn_silu_bwd_apply")


def head_decode(out, hw=None, strides=None, boxes=True, obj_sigmoid=False, corners=False):
    """sy_head_decode in place on out [B, A, 5+nc] fp32 contiguous: boxes -> (xy + grid) * stride, exp(wh) * stride over the
    levels hw = [(h, w)] with `strides`; obj_sigmoid -> sigmoid of column 4; corners -> (cx, cy, w, h) to (x1, y1, x2, y2)."""
    assert out.dtype == torch.float32 and out.is_contiguous() and out.dim() == 3
    B, A, nch = out.shape
    flags = (1 if boxes else 0) | (2 if obj_sigmoid else 0) | (4 if corners else 0)
    if boxes:
        n = len(hw)
        lh = (C.c_int32 * n)(*[int(h) for h, _ in hw])
        lw = (C.c_int32 * n)(*[int(w) for _, w in hw])
        ls = (C.c_float * n)(*[float(s) for s in strides])
    else:
        n, lh, lw, ls = 0, None, None, None
    check(_lib.lib().sy_head_decode(out.data_ptr(), B, A, nch, lh, lw, ls, n, flags, stream_of(out)), "sy_head_decode")
    return out


class PostprocessWorkspace:
    """Device buffers reused across sy_postprocess calls for a given (B, A)."""

    def __init__(self, B, A, device, max_det=None):
        self.B, self.A = B, A
        self.max_det = A if max_det is None else max_det
        nbytes = _lib.lib().sy_postprocess_workspace_bytes(B, A)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.det = torch.zeros((B, self.max_det, 7), dtype=torch.float32, device=device)
        self.index = torch.zeros((B, self.max_det), dtype=torch.int32, device=device)
        self.count = torch.zeros((B,), dtype=torch.int32, device=device)


def postprocess(pred, num_classes, conf_thre, nms_thre, ws=None):
    """pred: [B, A, 5+nc] fp32 decoded head output.  Returns (det [B,max,7], index [B,max], count [B])
    device tensors; no host synchronisation happens here."""
    assert pred.dtype == torch.float32 and pred.is_contiguous() and pred.shape[2] == 5 + num_classes
    B, A = pred.shape[:2]
    if ws is None or ws.B != B or ws.A != A:
        ws = PostprocessWorkspace(B, A, pred.device)
    check(_lib.lib().sy_postprocess(pred.data_ptr(), B, A, num_classes, float(conf_thre), float(nms_thre),
                                    ws.max_det, ws.det.data_ptr(), ws.index.data_ptr(), ws.count.data_ptr(),
                                    ws.ws.data_ptr(), stream_of(pred)), "sy_postprocess")
    return ws.det, ws.index, ws.count


class TalLossWorkspace:
    """Device buffers of sy_tal_loss for a given (B, A): workspace, d_raw, losses[8], fg mask, level tables."""

    def __init__(self, B, A, nch, hw_list, strides, device, max_labels=120):
        self.B, self.A, self.max_labels = B, A, max_labels
        self.ws = torch.empty(_lib.lib().sy_tal_loss_workspace_bytes(B, A, max_labels), dtype=torch.uint8, device=device)
        self.d_raw = torch.empty((B, A, nch), dtype=torch.float32, device=device)
        self.losses = torch.zeros(8, dtype=torch.float32, device=device)
        self.fg = torch.zeros((B, A), dtype=torch.int32, device=device)
        # the level tables are read on the HOST by the launcher (tiny, by-value kernel argument)
        self.lh = (C.c_int32 * len(hw_list))(*[int(h) for h, _ in hw_list])
        self.lw = (C.c_int32 * len(hw_list))(*[int(w) for _, w in hw_list])
        self.ls = (C.c_float * len(hw_list))(*[float(s) for s in strides])
        self.nlevels = len(hw_list)


def tal_loss(raw, labels, support, num_classes, gamma, ignore_thr, ignore_value, use_l1, ws, d_pad=None):
    """SimOTA + Trend-Aware loss forward and gradient (sy_tal_loss).  raw [B,A,5+nc] fp32 contiguous;
    labels/support [B, max_labels, 5] fp32.  Returns (losses[8], d_raw, fg_mask) device tensors.
    d_pad: optional [B, A, 16] tensor (bf16 / fp16 / fp32) that receives the gradient in the backward pass's operand layout."""
    assert raw.dtype == torch.float32 and raw.is_contiguous()
    # wider label rows (the reference's mixup branch, tal_head.py:277-283, slices [..., :5] itself): the kernel's row pitch is 5
    labels = labels[..., :5].to(raw.device, torch.float32).contiguous()
    support = support[..., :5].to(raw.device, torch.float32).contiguous()
    assert labels.shape[1] == ws.max_labels and support.shape == labels.shape
    check(_lib.lib().sy_tal_loss(raw.data_ptr(), ws.B, ws.A, num_classes, labels.data_ptr(), support.data_ptr(),
                                 ws.max_labels, C.cast(ws.lh, C.c_void_p), C.cast(ws.lw, C.c_void_p),
                                 C.cast(ws.ls, C.c_void_p), ws.nlevels, float(gamma), float(ignore_thr),
                                 float(ignore_value), 1 if use_l1 else 0, ws.d_raw.data_ptr(), ws.losses.data_ptr(),
                                 ws.fg.data_ptr(), ws.ws.data_ptr(), _p(d_pad), 0 if d_pad is None else DTYPE_CODE[d_pad.dtype],
                                 stream_of(raw)), "sy_tal_loss")
    return ws.losses, ws.d_raw, ws.fg


def tal_assignment(ws):
    """(matched_gt [B, A] int32, -1 = background; matched_iou [B, A] fp32) of the last tal_loss call on `ws` (sy_tal_loss_assignment)."""
    mg = torch.empty((ws.B, ws.A), dtype=torch.int32, device=ws.ws.device)
    mi = torch.empty((ws.B, ws.A), dtype=torch.float32, device=ws.ws.device)
    check(_lib.lib().sy_tal_loss_assignment(ws.ws.data_ptr(), ws.B, ws.A, ws.max_labels, mg.data_ptr(), mi.data_ptr(),
                                            stream_of(ws.ws)), "sy_tal_loss_assignment")
    return mg, mi


# ---------------------------------------------------------------------------------------------------
# per-shape kernel-variant selection for sy_conv2d (measured once per shape on the device, cached)
# ---------------------------------------------------------------------------------------------------
TILE_RS = 16
TILE_WR = 80           # codes >= 80: fragment-packed weights straight to VGPRs (need the `wfrag` operand)
_TILE_CANDIDATES = {            # workgroup tile (channels x pixels) + staging strategy codes, see include/streamyolo_hip.h
    # rs = register-staged, d2/d3 = 2-/3-deep LDS-DMA ring (codes: include/streamyolo_hip.h)
    # wr = register-staged pixels + fragment-packed weights straight to VGPRs
    "wide256": [19, 22, 23, 35, 38, 51, 54, 83, 86, 87, 99, 102, 103],        # Cout >= 256 adds the 256-channel weights-in-register tiles
    "wide": [19, 22, 23, 35, 38, 51, 54, 83, 86, 87, 99, 102, 103],   # 96+: WR with 3 register stages
    "c64": [20, 23, 39, 55, 36, 84, 87, 103],
    "c32": [21, 23, 39, 85, 87],
}
_tile_cache = {}
_wgrad_cache = {}
import os as _os


# ---- persisted tuner choices ------------------------------------------------------------------------------------------------
# The per-shape variant choice is a measurement (a few launches of every candidate): ~70 conv shapes + ~70 wgrad shapes per
# (batch, size) for StreamYOLO-l.  Multi-scale training meets a new (H, W) every 10 iterations (cfgs/l_s50_onex_dfp_tal_flip.py:
# 139-158), so the choices are written to a JSON file keyed by a hash of the kernel sources (another build re-tunes) and the
# device name; later processes and later plans start from it.  STREAMYOLO_TUNE_CACHE: path, or "0" to disable.  The in-tree
# default travels with the library; it also makes the variant choice — and with it the step time — reproducible run to run.
class _TuneStore:
    def __init__(self):
        self.path, self.loaded, self.dirty, self.key = None, False, False, None

    def _source_key(self, device):
        import hashlib
        h = hashlib.sha256()
        h.update(_lib.kernel_source_key().encode())
        h.update(_lib.lib().sy_version())
        h.update(repr((HALO_TILES, HALO_SMALL_TILES, HALO_KGROUP_TILES, HALO_S2_KGROUP_TILES, HALO_S2_TILES, TILE_1X1K, WGRAD_EXTRA)).encode())          # candidate-set switches (A/B runs)
        # (no device name in the key: this library is gfx950-only, and torch reports an empty name under rocprofv3 — a profiled
        #  run then overwrote the cache of the normal runs with its own)
        return h.hexdigest()[:16]

    def load(self, device):
        if self.loaded:
            return
        self.loaded = True
        env = _os.environ.get("STREAMYOLO_TUNE_CACHE", "")
        if env == "0" or _lib.is_emulator():
            return
        self.path = env or _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lib", "tune_cache.json")
        self.key = self._source_key(device)
        try:
            import json
            with open(self.path) as f:
                d = json.load(f)
            if d.get("key") != self.key:
                return
            dev = str(device)
            for k, v in d.get("tile", []):
                _tile_cache[tuple(k) + (dev,)] = int(v)
            for k, v in d.get("wgrad", []):
                _wgrad_cache[tuple(k) + (dev,)] = (int(v[0]), int(v[1]))
        except (OSError, ValueError, KeyError, TypeError):
            pass

    def save(self):
        if not self.dirty or self.path is None:
            return
        self.dirty = False
        import json
        d = {"key": self.key, "tile": [[list(k[:-1]), v] for k, v in _tile_cache.items()],
             "wgrad": [[list(k[:-1]), list(v)] for k, v in _wgrad_cache.items()]}
        try:
            tmp = self.path + ".tmp%d" % _os.getpid()
            with open(tmp, "w") as f:
                json.dump(d, f)
            _os.replace(tmp, self.path)
        except OSError:
            pass


_tune_store = _TuneStore()


def save_tuned():
    """Write tuner choices made since the last call to the cache file (plans call this after their tuning step)."""
    _tune_store.save()


# (112 / 113 — the first-generation 2x2-wave and 128-pixel tiles — were never chosen by the tuner on any StreamYOLO shape: removed)
# 107 / 104 = the software-pipelined tile on 3 / 5 output rows: a frame's 38 x 60 map is 13 x 3 or 8 x 5 rows, 416 / 256 workgroups
# instead of 117's 608 (one ragged round of 2.4 per CU) with 0.67 / 0.4 of its weight-fragment loads per MFMA: +15 % on the
# 256->256 @38x60 layers, +17 % on 512->512 @19x30 (profiles/r04 stage bd)
# 101 / 100 / 98 / 109 (round 5) = the third generation of the same four tiles (csrc/conv3x3_halo3.h): fragment reads with immediate
# offsets, hand-placed instruction stream, prefetch across the slab boundary; +7-17 % in the probes (profiles/r05 stages e-h); same
# accumulation order as the second generation (bit-identical results); 96 / 97 = the same kernel on 64 channels x (2 x 2 | 2 x 3) rows
HALO_TILES = [int(t) for t in _os.environ.get("STREAMYOLO_HALO_TILES", "114,115,116,117,118,107,104,101,100,98,109,96,97").replace("+", ",").split(",") if t]
# stride-2 3x3 layers on the window-in-LDS kernels (tile codes 110 = forward, 108 = data gradient).  Measured in round 4
# (profiles/r04/a_probe_s2_*.txt): forward 421 vs 395 (dark2.0) / 739 vs 585 TF/s (dark4.0) against the best implicit-GEMM
# variant, data gradient 245 vs 250 / 504 vs 484; l step 22.61 vs 22.70 ms (b_bench_s2 / b_bench_base) — candidates by default
# launches of few pixels (one streamed frame; the 19x30 maps of a training batch): one MFMA tile per wave, more workgroups
HALO_SMALL_TILES = [int(t) for t in _os.environ.get("STREAMYOLO_HALO_SMALL_TILES", "112,113").replace("+", ",").split(",") if t]
HALO_SMALL_PIXELS = 12000
# K groups inside the workgroup (csrc/conv3x3_halo.h, KS): eval / streaming plans that allow another fp32 summation order
HALO_KGROUP_TILES = [int(t) for t in _os.environ.get("STREAMYOLO_HALO_KGROUP_TILES", "111,106").replace("+", ",").split(",") if t]
HALO_S2_KGROUP_TILES = [int(t) for t in _os.environ.get("STREAMYOLO_HALO_S2_KGROUP_TILES", "105").replace("+", ",").split(",") if t]
# 125 / 126 / 127 (round 6, csrc/conv3x3_s2dgrad.h): the data gradient with ALL FOUR parity classes in one workgroup — the dy window
# parked once, 18 TP MFMAs per wave and slab instead of 4 ... 16, no light / heavy workgroups (126: 128 ch x 2 class rows; 125: one
# class row, two waves per SIMD; 127: the 64-channel tile)
HALO_S2_TILES = [int(t) for t in _os.environ.get("STREAMYOLO_HALO_S2_TILES", "110,108,125,126,127").replace("+", ",").split(",") if t]
S2_DGRAD_TILES = (108, 125, 126, 127)
TILE_1X1K = [int(t) for t in _os.environ.get("STREAMYOLO_TILE_1X1K", "121,122,123,124").replace("+", ",").split(",") if t]


def autotune_enabled(device):
    import os
    return device.type == "cuda" and os.environ.get("STREAMYOLO_AUTOTUNE", "1") != "0"


def _time_launches(run, device, launches=4, rounds=2):
    """Best of `rounds` timings of `launches` back-to-back launches (HIP events on the current stream), in ms.  (Timing a hipGraph of
    the launches instead — no host time between them — picks WORSE tiles for the few-microsecond launches of a streamed frame:
    1.55-1.56 vs 1.41-1.47 ms per frame over three tunings each on one box, profiles/r04 stage ao.  Through the wrappers every launch
    starts on an idle chip, as it does behind the ~3 us dependent-launch gap of the plan's tape.)"""
    best = float("inf")
    for _ in range(rounds):
        torch.cuda.synchronize(device)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(launches):
            run()
        e.record()
        torch.cuda.synchronize(device)
        best = min(best, s.elapsed_time(e))
    return best


def tuned_tile(mode, dtype, N, H, W, Cin, Cout, k, stride, device, with_stats=False, only=None):
    """Fastest sy_conv2d variant for this problem shape (H, W = INPUT size of the launch), or 0 (the
    library's static heuristic) when tuning is off.  Measured with HIP events on dummy tensors.
    only: restrict the choice to these tile codes (launches that need a particular kernel family)."""
    if not autotune_enabled(device):
        return 0 if only is None else only[-1]
    _tune_store.load(device)
    key = (mode, dtype_code(dtype), N, H, W, Cin, Cout, k, stride, bool(with_stats)) + \
        (() if only is None else (",".join(str(t) for t in only),)) + (str(device),)
    hit = _tile_cache.get(key)
    if hit is not None:
        return hit
    code = dtype_code(dtype)
    if mode == CONV_FWD:
        Ho, Wo = conv_out_size(H, k, stride), conv_out_size(W, k, stride)
        x = View.alloc(N, H, W, Cin, code, device, zero=True)
        y = View.alloc(N, Ho, Wo, Cout, code, device)
        w = torch.zeros((Cout, k * k * Cin), dtype=TORCH_DTYPE[code], device=device)
    else:       # data gradient: launch input = dy [N,H,W,Cin(=fwd Cout)], output = dx at the forward input size
        Ho, Wo = (H, W) if stride == 1 else (H * 2, W * 2)
        x = View.alloc(N, H, W, Cin, code, device, zero=True)
        y = View.alloc(N, Ho, Wo, Cout, code, device)
        w = torch.zeros((Cout, k * k * Cin), dtype=TORCH_DTYPE[code], device=device)
    scale = torch.ones(Cout, device=device)
    shift = torch.zeros(Cout, device=device)
    stats = (torch.zeros(32 * Cout, device=device), torch.zeros(32 * Cout, device=device)) if with_stats else None
    from .model.packing import pack_conv_weight_frag
    wf = pack_conv_weight_frag(w, k)
    cands = list(_TILE_CANDIDATES["c32" if Cout <= 32 else "c64" if Cout <= 64 else "wide" if Cout < 256 else "wide256"])
    if k == 3 and stride == 1 and wf is not None and HALO_TILES:
        # 3x3 stride-1 layers: the halo-resident kernel (csrc/conv3x3_halo.h), tiles of 64 / 128 / 256 channels
        cands += [t for t in HALO_TILES if not (t in (116, 96, 97) and Cout > 64)]      # 116 / 96 / 97: the 64-channel tiles
        if N * Ho * Wo <= HALO_SMALL_PIXELS:
            cands += HALO_SMALL_TILES
    if k == 3 and stride == 2 and wf is not None and Cin % (16 if code == DT_F32 else 32) == 0:
        cands += [t for t in HALO_S2_TILES if (t in S2_DGRAD_TILES) == (mode == CONV_DGRAD)       # 110: forward; 108 / 126 / 127: data gradient
                  and not (t == 127 and Cout > 64)]
    if k == 1 and stride == 1 and wf is not None and Cin in (64, 128, 256, 512, 1024, 2048) and code != DT_F32:
        # whole-K burst kernel (csrc/conv1x1_tile.h): 128 ch x 64 px | 64 ch x 128 px | 128 ch x 128 px (Cin <= 256)
        cands += [t for t in TILE_1X1K if not (t == 122 and Cout > 64) and not (t == 123 and Cin > 256) and not (t not in (121, 124) and Cin > 512)
                  and not (t == 124 and (Cin < 1024 or N * Ho * Wo > HALO_SMALL_PIXELS))]
    if only is not None:
        cands = list(only)
    best, best_t = (0 if only is None else only[-1]), float("inf")
    for t in cands:
        if t >= TILE_WR and wf is None:
            continue

        def run():
            if with_stats:
                conv2d(x, w, y, k, stride, stats=stats, mode=mode, tile=t, wfrag=wf)
            elif mode == CONV_FWD:
                conv2d(x, w, y, k, stride, scale, shift, epilogue=EPI_SILU, mode=mode, tile=t, wfrag=wf)
            else:                                # data gradient as the backward pass launches it: raw output, no affine
                conv2d(x, w, y, k, stride, epilogue=EPI_LINEAR, mode=mode, tile=t, wfrag=wf)
        try:
            run()
            # (the few-microsecond eval launches of a streamed frame: more repetitions, the choice among near-equal tiles is noisy)
            small = mode == CONV_FWD and not with_stats and N * Ho * Wo <= HALO_SMALL_PIXELS
            dt = _time_launches(run, device, launches=8, rounds=4) if small else _time_launches(run, device)
        except _lib.HipLibraryError:
            continue
        if dt < best_t:
            best, best_t = t, dt
    _tile_cache[key] = best
    _tune_store.dirty = True
    return best


def tuned_splitk(dtype, N, H, W, Cin, Cout, device, base_tile, stride=1):
    """(splits, tile) for a 3x3 stride-1 EVAL convolution of a small map: the plain kernel (`base_tile`, splits = 1) against (a) the
    K-group tiles (HALO_KGROUP_TILES: the contraction split over wave groups INSIDE the workgroup, one launch, splits = 1) and (b)
    2 / 4 channel-slab ranges + sy_splitk_epilogue (maps of <= 6000 pixels), timed on dummy tensors (cached, persisted as mode 2
    entries).  Every alternative sums the fp32 products in another order than the plain kernels — plans opt in (allow_split_k)."""
    code = dtype_code(dtype)
    epc = 4 if code == DT_F32 else 8
    Ho, Wo = conv_out_size(H, 3, stride), conv_out_size(W, 3, stride)
    if not autotune_enabled(device) or Cin % (4 * epc) or N * Ho * Wo > HALO_SMALL_PIXELS:
        return (1, base_tile)
    _tune_store.load(device)
    key = ("splitk", code, N, H, W, Cin, Cout, base_tile, str(device)) + (() if stride == 1 else (stride,))
    hit = _tile_cache.get(key)
    if hit is not None:
        return (hit // 1000, hit % 1000) if hit >= 1000 else (1, base_tile)
    from .model.packing import pack_conv_weight_frag
    x = View.alloc(N, H, W, Cin, code, device, zero=True)
    y = View.alloc(N, Ho, Wo, Cout, code, device)
    w = torch.zeros((Cout, 9 * Cin), dtype=TORCH_DTYPE[code], device=device)
    wf = pack_conv_weight_frag(w, 3)
    scale, shift = torch.ones(Cout, device=device), torch.zeros(Cout, device=device)
    part = torch.empty(4 * N * H * W * Cout, dtype=torch.float32, device=device) if stride == 1 else None
    best, best_t = (1, base_tile), _time_launches(
        lambda: conv2d(x, w, y, 3, stride, scale, shift, epilogue=EPI_SILU, tile=base_tile, wfrag=wf if base_tile >= TILE_WR else None),
        device, launches=8, rounds=4)
    for t in (HALO_KGROUP_TILES if stride == 1 else HALO_S2_KGROUP_TILES):
        try:
            run = lambda: conv2d(x, w, y, 3, stride, scale, shift, epilogue=EPI_SILU, tile=t, wfrag=wf)   # noqa: E731
            run()
            dt_ = _time_launches(run, device, launches=8, rounds=4)
        except _lib.HipLibraryError:
            continue
        if dt_ < best_t:
            best, best_t = (1, t), dt_
    for S in (2, 4):
        if S > Cin // (4 * epc) or N * H * W > 6000 or stride != 1:
            continue
        for t in (117, 118, 113, 112):
            try:
                run = lambda: conv2d_splitk(x, w, y, 3, 1, scale, shift, part, S, epilogue=EPI_SILU, tile=t, wfrag=wf)   # noqa: E731
                run()
                dt_ = _time_launches(run, device, launches=8, rounds=4)
            except _lib.HipLibraryError:
                continue
            if dt_ < best_t:
                best, best_t = (S, t), dt_
    _tile_cache[key] = best[0] * 1000 + best[1] if best != (1, base_tile) else 1
    _tune_store.dirty = True
    return best


def tuned_bottleneck(dtype, N, H, W, cin, hid, cout, shortcut, device, tile1, tile3):
    """True when the fused Bottleneck launch (tile 119: 1x1 -> 3x3, hidden activation in LDS) beats the two launches (tiles `tile1`,
    `tile3`, the plan's tuned choices) on this shape — measured on dummy tensors, each candidate replayed from a hipGraph of 8
    repetitions (no host time between the launches, as in the plans' graphs / tapes).  Cached and persisted like the tile choices."""
    import os
    from .engine import FUSE_BOTTLENECKS
    code = dtype_code(dtype)
    if code == DT_F32 or cin % 32 or hid % 32 or cin + hid > 544:
        return False
    # the fused launch does 2-3x the 1x1 layer's products: it can only pay where the two launches are latency-bound, i.e. where a
    # launch is well under one round of workgroups per CU (measured: profiles/r04/m_bottleneck_probe.txt — batch 1 wins 3-11 % at
    # 64 / 128 channels, batch 16 loses 7-39 % everywhere)
    if N * H * W > 40000 and FUSE_BOTTLENECKS != "force":
        return False
    if FUSE_BOTTLENECKS == "force":
        return True
    if not autotune_enabled(device) or torch.cuda.is_current_stream_capturing():
        return False
    _tune_store.load(device)
    key = ("bnk", code, N, H, W, cin, hid, cout, bool(shortcut), int(tile1), int(tile3), str(device))
    hit = _tile_cache.get(key)
    if hit is not None:
        return bool(hit)
    from .model.packing import pack_conv_weight_frag
    tdt = TORCH_DTYPE[code]
    g = torch.Generator(device="cpu").manual_seed(cin + hid)          # random operands: zero-filled ones clock the chip up
    x = View.alloc(N, H, W, cin, code, device)
    x.buf.copy_(torch.randn(x.buf.shape, generator=g).to(tdt))
    h = View.alloc(N, H, W, hid, code, device)
    y = View.alloc(N, H, W, cout, code, device)
    w1 = (torch.randn((hid, cin), generator=g) / cin ** 0.5).to(tdt).to(device)
    w2 = (torch.randn((cout, 9 * hid), generator=g) / (9 * hid) ** 0.5).to(tdt).to(device)
    w1f, w2f = pack_conv_weight_frag(w1, 1), pack_conv_weight_frag(w2, 3)
    s1, b1 = torch.ones(hid, device=device), torch.zeros(hid, device=device)
    s2, b2 = torch.ones(cout, device=device), torch.zeros(cout, device=device)
    res = x if (shortcut and cin == cout) else None

    def two():
        conv2d(x, w1, h, 1, 1, s1, b1, epilogue=EPI_SILU, tile=tile1, wfrag=w1f if tile1 >= TILE_WR else None)
        conv2d(h, w2, y, 3, 1, s2, b2, res=res, epilogue=EPI_SILU, tile=tile3, wfrag=w2f if tile3 >= TILE_WR else None)

    def one():
        conv2d(x, w2, y, 3, 1, s2, b2, res=res, epilogue=EPI_SILU, tile=119, wfrag=w2f, pre=(w1f, s1, b1))
    times = []
    try:
        for fn in (two, one):
            fn()
            torch.cuda.synchronize(device)
            side = torch.cuda.Stream(device=device)
            with torch.cuda.stream(side):                           # (capture on a side stream: the caller's stream may be the legacy one)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(8):
                        fn()
                g.replay()
                torch.cuda.synchronize(device)
                best = float("inf")
                for _ in range(3):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    g.replay()
                    e.record()
                    torch.cuda.synchronize(device)
                    best = min(best, s.elapsed_time(e))
            times.append(best)
        win = times[1] < 0.96 * times[0]                            # fused must win by a margin
    except (_lib.HipLibraryError, RuntimeError):
        win = False
    _tile_cache[key] = 1 if win else 0
    _tune_store.dirty = True
    return win


# tile codes 1-6: scatter-transposed staging; +16 / +32: LDS-DMA ring (3 / 4 slabs) + ds_read_b64_tr_b16 fragments
_WGRAD_CANDIDATES = [(0, 0), (1, 1024), (2, 512), (4, 1024), (17, 512), (17, 1024), (33, 1024), (18, 512), (18, 1024),
                     (20, 1024), (22, 1024), (21, 1024),
                     (17, 256), (33, 256), (18, 256),
                     # 3x3 stride 1: all nine taps per workgroup (conv_wgrad9_kernel); few splits: the split-K slabs + fold are a
                     # third of its time at 1024 workgroups (profiles/r02/f_wgrad_probe.txt).  52: four waves x 32 input channels in
                     # <= 256 registers per lane (rounds 2-4 ran it unbounded — 464 registers, one workgroup owned its CU; round 5
                     # stages j, q: as fast alone, the l step 21.37-21.44 vs 21.53-21.56 ms)
                     (52, 128), (52, 256),
                     # 59 / 60: the slab loop as one instruction stream (rendezvous, DMA issue and the next slab's first fragments
                     # behind MFMAs): +10-16 % on 128 workgroups, +2-5 % on 256 (profiles/r05 stage zf); 60: eight waves, 64 input
                     # channels per workgroup (Cin % 64 == 0; -27 % L2 -> LDS bytes per MFMA) — the tile the training plan schedules
                     # (train_engine.scheduled_wgrad)
                     (59, 128), (59, 256), (60, 128), (60, 256)]
# further candidates for A/B runs, "tile:blocks,tile:blocks"; part
# of the tuner-cache key, so such a run tunes by itself
# ("tile/blocks+tile/blocks" is accepted as well: tools/gpu.sh splits its task arguments at ":" and ",")
WGRAD_EXTRA = [tuple(int(v) for v in e.replace("/", ":").split(":"))
               for e in _os.environ.get("STREAMYOLO_WGRAD_EXTRA", "").replace("+", ",").split(",") if e]

def tuned_wgrad(dtype, N, H, W, Cin, Ho, Wo, Cout, k, stride, device, workspace):
    """(tile, target_blocks) of the fastest sy_conv2d_wgrad variant for this shape (cached), (0, 0) when off."""
    if not autotune_enabled(device):
        return (0, 0)
    _tune_store.load(device)
    key = (dtype_code(dtype), N, H, W, Cin, Ho, Wo, Cout, k, stride, str(device))
    hit = _wgrad_cache.get(key)
    if hit is not None:
        return hit
    code = dtype_code(dtype)
    x = View.alloc(N, H, W, Cin, code, device, zero=True)
    dy = View.alloc(N, Ho, Wo, Cout, code, device, zero=True)
    dw = torch.zeros((Cout, Cin, k, k), dtype=torch.float32, device=device)
    best, best_t = (0, 0), float("inf")
    for (t, tb) in _WGRAD_CANDIDATES + WGRAD_EXTRA:
        if t in (52, 59, 60):
            if k != 3 or stride != 1 or Cin % (64 if t == 60 else 32) or Cout % 16:
                continue
        elif (t & 15) in (1, 5, 6) and Cout < 128:
            continue
        elif (t & 15) == 2 and Cout < 64:
            continue
        try:
            def run():
                conv2d_wgrad(x, dy, dw, k, stride, oihw=True, workspace=workspace, tile=t, target_blocks=tb)
            run()
            dt = _time_launches(run, device)
        except _lib.HipLibraryError:
            continue
        if dt < best_t:
            best, best_t = (t, tb), dt
    _wgrad_cache[key] = best
    _tune_store.dirty = True
    return best
