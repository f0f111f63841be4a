"""Execution plans: the StreamYOLO network as a static list of HIP kernel launches over
pre-allocated NHWC buffers.

MI355X-first design (not a translation of the reference's eager module tree):
  * the nn.Module tree (streamyolo_amd/model/*) only owns parameters; a Plan is compiled once per
    (mode, batch, H, W, dtype) and replayed — fixed buffers, fixed launch order, capturable in a
    hipGraph (torch.cuda.graphs) because nothing allocates or synchronises while it runs;
  * every torch.cat of the reference (CSP, SPP, top-down, bottom-up, DFP — trap T6) is a channel
    slice of a wider buffer that the producing kernel writes directly;
  * eval / streaming: BN folded into the conv epilogue, both frames of a pair batched as 2B images
    (legal outside training — SURVEY.md R4), DFP "cat + add" is the residual epilogue of the two
    jian convs, sigmoid + box decode are the epilogue of the prediction convs;
  * training: per-frame passes (separate BN statistics, current frame first — trap T2), conv
    emits sum / sum-of-squares from its fp32 accumulators, backward = BN/SiLU reduce+apply,
    MFMA dgrad and MFMA wgrad writing into one flat fp32 gradient arena in parameter layout.

Reference lines each piece follows are cited in the builders below.
"""
import math
import os

import torch

from . import ops
from .data import FramePairsU8
from .ops import View, EPI_SILU, EPI_LINEAR, EPI_SIGMOID, EPI_DECODE, CONV_DGRAD
from .model.packing import pack_conv_weight, pack_conv_weight_frag, fold_bn


# ------------------------------------------------------------------------------------------------
# op records (pure data; executors below interpret them)
# ------------------------------------------------------------------------------------------------
class ConvOp:
    """BaseConv: conv -> BN -> SiLU (+ residual).  `mod` is the parameter-holding BaseConv module."""
    kind = "conv"

    def __init__(self, mod, x, y, res=None, need_dx=True, tag=""):
        self.mod, self.x, self.y, self.res, self.need_dx, self.tag = mod, x, y, res, need_dx, tag
        self.k, self.stride = mod.ksize, mod.stride
        # training state (allocated lazily by TrainState)
        self.yraw = None
        self.stat = None
        self._tiles = {}
        self.pre_op = None          # a Bottleneck's 3x3 conv: the 1x1 ConvOp in front of it (its ONLY reader) — inference plans may fuse
        self.fused_into = None      # ... and that 1x1 op: the 3x3 op that computes it inside its own launch when the tuner says so

    def tile(self, kind):
        """Tuned kernel variant for this op: kind = 'fwd' (eval epilogue), 'fwd_stats' (training) or 'dgrad'."""
        t = self._tiles.get(kind)
        if t is None:
            x, y = self.x, self.y
            if kind == "dgrad":
                t = ops.tuned_tile(CONV_DGRAD, y.dtype, y.N, y.H, y.W, y.C, x.C, self.k, self.stride, y.buf.device)
            else:
                t = ops.tuned_tile(ops.CONV_FWD, x.dtype, x.N, x.H, x.W, x.C, y.C, self.k, self.stride, x.buf.device,
                                   with_stats=(kind == "fwd_stats"))
            self._tiles[kind] = t
        return t


class PredOp:
    """The three 1x1 prediction convs of one level as two launches: (reg|obj) from reg_feat and cls
    from cls_feat (tal_head.py:167-171 of the reference)."""
    kind = "pred"

    def __init__(self, level, cls_mod, reg_mod, obj_mod, cls_x, reg_x, a0, stride):
        self.level, self.cls_mod, self.reg_mod, self.obj_mod = level, cls_mod, reg_mod, obj_mod
        self.cls_x, self.reg_x, self.a0, self.stride = cls_x, reg_x, a0, stride


class ResizeOp:
    kind = "resize"

    def __init__(self, src, dst):
        self.src, self.dst = src, dst


class SppOp:
    kind = "spp"

    def __init__(self, v):
        self.v = v


class FocusOp:
    kind = "focus"

    def __init__(self, c0, out):
        self.c0, self.out = c0, out


# ------------------------------------------------------------------------------------------------
# network builders
# ------------------------------------------------------------------------------------------------
# (streaming head level 0 beside levels 1-2 on a second stream: 1.883 vs 1.849 ms per frame on one stream — profiles/r03/c_* — at
#  batch 1 the frame is bound by dispatch, not by idle CUs; removed in round 4)
# Bottleneck 1x1 -> 3x3 as one launch where the plan's tuner finds it faster than the two launches (inference / streaming plans;
# csrc/bottleneck_fused.h).  "0": never; "force": wherever the kernel applies (tests, the emulator has no tuner).
FUSE_BOTTLENECKS = os.environ.get("STREAMYOLO_FUSE_BOTTLENECKS", "1")
FUSE_BOTTLENECKS = False if FUSE_BOTTLENECKS == "0" else FUSE_BOTTLENECKS
# Streaming step (launch tape, batch 1): the three detection levels — their DFP fusion convs and head towers — leave the main
# chain as soon as their PAN output exists and run on chains 1 / 2 beside the rest of the bottom-up path ("0": one chain).
STREAM_CHAINS = os.environ.get("STREAMYOLO_STREAM_CHAINS", "1") != "0"
# test hook: "S,tile" forces a split-K decision for every eligible layer (the emulator has no tuner)
FORCE_SPLIT_K = tuple(int(v) for v in os.environ["STREAMYOLO_FORCE_SPLIT_K"].split(",")) if os.environ.get("STREAMYOLO_FORCE_SPLIT_K") else None


class MergedConv:
    """Two BaseConvs that read the SAME input (CSPLayer conv2 / conv1, the first cls / reg tower conv of a head level) as ONE
    launch with their output channels stacked: one read of the input, twice the GEMM N on the narrowest layers, one launch
    less per kernel kind.  Eval plans fold BatchNorm into the epilogue's affine (ParamCache); training plans stack the
    operand layouts, BatchNorm parameters and gradient-arena slots of the parts (train_engine.StagedWeights / TrainPlan) —
    per-channel batch statistics do not care which module a channel belongs to."""

    def __init__(self, mods):
        self.mods = tuple(mods)
        m0 = self.mods[0]
        assert all(m.ksize == m0.ksize and m.stride == m0.stride and m.conv.in_channels == m0.conv.in_channels for m in self.mods)
        self.ksize, self.stride = m0.ksize, m0.stride
        self.out_channels = sum(m.conv.out_channels for m in self.mods)

    def parameters(self):
        for m in self.mods:
            yield from m.parameters()

    def buffers(self):
        for m in self.mods:
            yield from m.buffers()


class _Builder:
    merge_siblings = False                       # InferencePlan: True (training plans keep one op per BaseConv)

    def __init__(self, dtype, device):
        self.dtype, self.device = dtype, device
        self.ops = []

    def buf(self, N, H, W, C):
        return View.alloc(N, H, W, C, self.dtype, self.device)

    def conv(self, mod, x, y=None, res=None, need_dx=True, tag=""):
        if y is None:
            cout = mod.out_channels if isinstance(mod, MergedConv) else mod.conv.out_channels
            y = self.buf(x.N, ops.conv_out_size(x.H, mod.ksize, mod.stride), ops.conv_out_size(x.W, mod.ksize, mod.stride), cout)
        self.ops.append(ConvOp(mod, x, y, res, need_dx, tag))
        return y

    def csp(self, mod, x, out=None, tag=""):
        """CSPLayer: conv3(cat[m(conv1(x)), conv2(x)])  (yolox CSPLayer, Appendix C; trap T6)."""
        hid = mod.hidden
        n = mod.n
        if self.merge_siblings and n >= 1 and hid % 32 == 0:
            # one buffer [m(conv1) | conv2 | conv1]: the merged launch writes [conv2 | conv1], conv3 reads the first two thirds
            wide = self.buf(x.N, x.H, x.W, 3 * hid)
            cat = wide.slice(0, 2 * hid)
            self.conv(MergedConv((mod.conv2, mod.conv1)), x, wide.slice(hid, 2 * hid), tag=tag + ".conv2+conv1")
            a = wide.slice(2 * hid, hid)
        else:
            cat = self.buf(x.N, x.H, x.W, 2 * hid)
            a = self.conv(mod.conv1, x, cat.slice(0, hid) if n == 0 else None, tag=tag + ".conv1")
            self.conv(mod.conv2, x, cat.slice(hid, hid), tag=tag + ".conv2")
        for i, b in enumerate(mod.m):
            u = self.conv(b.conv1, a, tag="%s.m.%d.conv1" % (tag, i))
            dst = cat.slice(0, hid) if i == n - 1 else None
            a = self.conv(b.conv2, u, dst, res=a if b.use_add else None, tag="%s.m.%d.conv2" % (tag, i))
            self.ops[-1].pre_op, self.ops[-2].fused_into = self.ops[-2], self.ops[-1]       # u has no other reader
        return self.conv(mod.conv3, cat, out, tag=tag + ".conv3")


def base_convs(mod):
    """The BaseConv modules behind a ConvOp's `mod` (one, or the stacked parts of a MergedConv, in channel order)."""
    return mod.mods if isinstance(mod, MergedConv) else (mod,)


def build_frame_net(b, pafpn, N, H, W):
    """CSPDarknet (exps/model/darknet.py:167-179) + PAFPN (dfp_pafpn.py:124-140) for N images.
    Returns (focus_buffer, (pan2, pan1, pan0))."""
    bb = pafpn.backbone
    w = pafpn.width
    c3, c4, c5 = [int(c * w) for c in pafpn.in_channels]
    f0 = b.buf(N, H // 2, W // 2, 16)
    x = b.conv(bb.stem.conv, f0, need_dx=False, tag="stem")
    x = b.conv(bb.dark2[0], x, tag="dark2.0")
    x = b.csp(bb.dark2[1], x, tag="dark2.1")
    x = b.conv(bb.dark3[0], x, tag="dark3.0")
    h8, w8 = x.H, x.W
    catP3 = b.buf(N, h8, w8, 2 * c3)                     # cat[up(fpn_out1), dark3]  (:131)
    d3 = b.csp(bb.dark3[1], x, catP3.slice(c3, c3), tag="dark3.1")
    x = b.conv(bb.dark4[0], d3, tag="dark4.0")
    h16, w16 = x.H, x.W
    catP4 = b.buf(N, h16, w16, 2 * c4)                   # cat[up(fpn_out0), dark4]  (:126)
    d4 = b.csp(bb.dark4[1], x, catP4.slice(c4, c4), tag="dark4.1")
    x = b.conv(bb.dark5[0], d4, tag="dark5.0")
    h32, w32 = x.H, x.W
    spp = bb.dark5[1]
    hid = spp.conv1.conv.out_channels
    sppcat = b.buf(N, h32, w32, 4 * hid)                 # cat[x, pool5, pool9, pool13]  (trap T5)
    b.conv(spp.conv1, x, sppcat.slice(0, hid), tag="spp.conv1")
    b.ops.append(SppOp(sppcat))
    x = b.conv(spp.conv2, sppcat, tag="spp.conv2")
    d5 = b.csp(bb.dark5[2], x, tag="dark5.2")
    catN4 = b.buf(N, h32, w32, 2 * c4)                   # cat[bu_conv1(pan1), fpn_out0]  (:139)
    fpn0 = b.conv(pafpn.lateral_conv0, d5, catN4.slice(c4, c4), tag="lateral_conv0")
    b.ops.append(ResizeOp(fpn0, catP4.slice(0, c4)))     # F.interpolate(size=x1.shape[2:4]) (:125)
    f_out0 = b.csp(pafpn.C3_p4, catP4, tag="C3_p4")
    catN3 = b.buf(N, h16, w16, 2 * c3)                   # cat[bu_conv2(pan2), fpn_out1]  (:135)
    fpn1 = b.conv(pafpn.reduce_conv1, f_out0, catN3.slice(c3, c3), tag="reduce_conv1")
    b.ops.append(ResizeOp(fpn1, catP3.slice(0, c3)))     # (:130)
    pan2 = b.csp(pafpn.C3_p3, catP3, tag="C3_p3")
    b.conv(pafpn.bu_conv2, pan2, catN3.slice(0, c3), tag="bu_conv2")
    pan1 = b.csp(pafpn.C3_n3, catN3, tag="C3_n3")
    b.conv(pafpn.bu_conv1, pan1, catN4.slice(0, c4), tag="bu_conv1")
    pan0 = b.csp(pafpn.C3_n4, catN4, tag="C3_n4")
    return f0, (pan2, pan1, pan0)


def build_fuse_net(b, pafpn, cur, sup):
    """DFP: out_K = cat[jianK(cur_K), jianK(sup_K)] + cur_K (dfp_pafpn.py:168-170): the `+ cur_K`
    is the residual epilogue of the two jian launches, the cat is their channel slices."""
    fused = []
    for mod, c, s, name in zip((pafpn.jian2, pafpn.jian1, pafpn.jian0), cur, sup, ("jian2", "jian1", "jian0")):
        half = c.C // 2
        out = b.buf(c.N, c.H, c.W, c.C)
        b.conv(mod, c, out.slice(0, half), res=c.slice(0, half), tag=name + ".cur")
        b.conv(mod, s, out.slice(half, half), res=c.slice(half, half), tag=name + ".sup")
        fused.append(out)
    return tuple(fused)


def build_head_net(b, head, feats):
    """TALHead towers (tal_head.py:159-171).  Returns per-level PredOps appended to b.ops."""
    a0 = 0
    preds = []
    for k, x in enumerate(feats):
        st = b.conv(head.stems[k], x, tag="head.stem%d" % k)
        c0, r0 = head.cls_convs[k][0], head.reg_convs[k][0]
        if b.merge_siblings and c0.conv.out_channels % 32 == 0:
            both = b.conv(MergedConv((c0, r0)), st, tag="head.cls%d.0+reg%d.0" % (k, k))       # the towers' first convs share `st`
            c, r = both.slice(0, c0.conv.out_channels), both.slice(c0.conv.out_channels, r0.conv.out_channels)
        else:
            c = b.conv(c0, st, tag="head.cls%d.0" % k)
            r = b.conv(r0, st, tag="head.reg%d.0" % k)
        c = b.conv(head.cls_convs[k][1], c, tag="head.cls%d.1" % k)
        r = b.conv(head.reg_convs[k][1], r, tag="head.reg%d.1" % k)
        op = PredOp(k, head.cls_preds[k], head.reg_preds[k], head.obj_preds[k], c, r, a0, head.strides[k])
        b.ops.append(op)
        preds.append(op)
        a0 += x.H * x.W
    return preds, a0


# ------------------------------------------------------------------------------------------------
# parameter cache (packed K-contiguous weights + folded BN), refreshed when a parameter changes
# ------------------------------------------------------------------------------------------------
class ParamCache:
    def __init__(self, dtype, device):
        self.dtype, self.device = ops.dtype_code(dtype), device
        self.entries = {}

    @staticmethod
    def _ver(*ts):
        # + the weights epoch: FusedSGDEMA.step / BnRunningTable.run write through raw pointers (no _version bump)
        return tuple((t.data_ptr(), t._version) for t in ts) + (ops.weights_epoch(),)

    def conv_eval(self, mod):
        """(packed weight, scale, shift) with eval-mode BN folded (eps read at call time — trap T1)."""
        if isinstance(mod, MergedConv):
            key = ("eval", id(mod))
            parts = [self.conv_eval(m) for m in mod.mods]
            ver = tuple(self.entries[("eval", id(m))][0] for m in mod.mods)     # the parts' own versions
            e = self.entries.get(key)
            if e is None or e[0] != ver:
                e = (ver, torch.cat([p[0] for p in parts], 0).contiguous(), torch.cat([p[1] for p in parts]).contiguous(),
                     torch.cat([p[2] for p in parts]).contiguous())
                self.entries[key] = e
            return e[1:]
        bn = mod.bn
        key = ("eval", id(mod))
        ver = self._ver(mod.conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var) + (bn.eps,)
        e = self.entries.get(key)
        if e is None or e[0] != ver:
            w = self._pack(mod.conv.weight)
            scale, shift = fold_bn(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
            e = (ver, w, scale.to(self.device), shift.to(self.device))
            self.entries[key] = e
        return e[1:]

    def _pack(self, w, transpose=False):
        cin = w.shape[1]
        pad = 16 if cin == 12 else None                  # Focus stem: 12 -> 16 channels (zero weights)
        return pack_conv_weight(w.detach().float(), self.dtype, transpose=transpose, pad_cin_to=pad).to(self.device)

    def conv_weight(self, mod, transpose=False):
        key = ("w", id(mod), transpose)
        ver = self._ver(mod.conv.weight)
        e = self.entries.get(key)
        if e is None or e[0] != ver:
            e = (ver, self._pack(mod.conv.weight, transpose))
            self.entries[key] = e
        return e[1]

    def conv_weight_frag(self, mod, transpose=False):
        """Fragment-ordered copy of the packed weights for the SY_TILE_WR kernels (None if not applicable)."""
        key = ("wf", id(mod), transpose)
        if isinstance(mod, MergedConv):
            assert not transpose
            w = self.conv_eval(mod)[0]
            ver = self.entries[("eval", id(mod))][0]
            e = self.entries.get(key)
            if e is None or e[0] != ver:
                e = (ver, pack_conv_weight_frag(w, mod.ksize))
                self.entries[key] = e
            return e[1]
        ver = self._ver(mod.conv.weight)
        e = self.entries.get(key)
        if e is None or e[0] != ver:
            e = (ver, pack_conv_weight_frag(self.conv_weight(mod, transpose), mod.ksize))
            self.entries[key] = e
        return e[1]

    def pred(self, op):
        """Packed (reg|obj) [5 -> rows, Cin] and cls [nc, Cin] weights + fp32 biases."""
        key = ("pred", id(op.cls_mod))
        ver = self._ver(op.cls_mod.weight, op.cls_mod.bias, op.reg_mod.weight, op.reg_mod.bias,
                        op.obj_mod.weight, op.obj_mod.bias)
        e = self.entries.get(key)
        if e is None or e[0] != ver:
            w_ro = torch.cat([op.reg_mod.weight.detach().float(), op.obj_mod.weight.detach().float()], 0)
            b_ro = torch.cat([op.reg_mod.bias.detach().float(), op.obj_mod.bias.detach().float()], 0)
            e = (ver,
                 pack_conv_weight(w_ro, self.dtype).to(self.device), b_ro.contiguous().to(self.device),
                 pack_conv_weight(op.cls_mod.weight.detach().float(), self.dtype).to(self.device),
                 op.cls_mod.bias.detach().float().contiguous().to(self.device),
                 pack_conv_weight(w_ro, self.dtype, transpose=True, pad_cout_to=8).to(self.device),
                 pack_conv_weight(op.cls_mod.weight.detach().float(), self.dtype, transpose=True).to(self.device))
            self.entries[key] = e
        return e[1:]


# ------------------------------------------------------------------------------------------------
# inference plan (eval off_pipe / on_pipe)
# ------------------------------------------------------------------------------------------------
class InferencePlan:
    """mode 'off_pipe': frames [B,6,H,W] -> decoded [B,A,5+nc]   (yolox.py:31-50, dfp_pafpn.py:109-175)
       mode 'on_pipe' : frame  [B,3,H,W] + buffer -> (decoded, buffer')   (yolox.py:51-55, dfp_pafpn.py:177-228)
    Either part may be absent (pafpn only / head only) for the stand-alone module entry points."""

    def __init__(self, pafpn, head, mode, B, H, W, dtype, device, decode=True):
        self.pafpn, self.head, self.mode = pafpn, head, mode
        self.B, self.H, self.W, self.device = B, H, W, device
        self.dtype = ops.dtype_code(dtype)
        self.decode = decode
        self.cache = ParamCache(self.dtype, device)
        self._stream_tape, self._tape_param_list = None, None
        self._rec = None                             # open native tape (run_stream_taped)
        # split-K for the deep small-map 3x3 layers of the streaming step (36-72 workgroups at batch 1): opt-in, because the
        # fp32 summation order differs from the single-pass kernel (the facade's off_pipe == chained on_pipe bit-identity is
        # kept on the exact path); StreamingDetector / bench.py --workload stream turn it on for the 16-bit modes
        self.allow_split_k, self._in_stream, self._splitk_ws, self._ws_grew = False, False, None, False
        b = _Builder(self.dtype, device)
        b.merge_siblings = os.environ.get("STREAMYOLO_MERGE_SIBLINGS", "1") != "0"
        self.b = b
        self.pair = (mode == "off_pipe")
        self.fused = None
        if pafpn is not None:
            n_img = 2 * B if self.pair else B
            self.f0, pans = build_frame_net(b, pafpn, n_img, H, W)
            if self.pair:
                cur = tuple(_batch_slice(p, 0, B) for p in pans)
                sup = tuple(_batch_slice(p, B, B) for p in pans)
                self.cur_pans = cur
            else:
                cur = pans
                self.cur_pans = pans
                # the support-frame inputs of the fusion are re-pointed at the caller's buffer per call
                sup = tuple(View.alloc(p.N, p.H, p.W, p.C, self.dtype, device, zero=True) for p in pans)
            self.sup_in = sup
            self._sup_own = [v.buf for v in sup]                   # plan-owned support buffers (run_stream's state)
            self.fused = build_fuse_net(b, pafpn, cur, sup)
        self.n_backbone_ops = len(b.ops)
        self.preds, self.A, self.out = None, 0, None
        if head is not None:
            if self.fused is None:
                w = head.width
                self.fused = tuple(b.buf(B, math.ceil(H / s), math.ceil(W / s), int(c * w))
                                   for s, c in zip(head.strides, head.in_channels_))
            self.preds, self.A = build_head_net(b, head, self.fused)
            self.nc = head.num_classes
            self.out = torch.empty((B, self.A, 5 + self.nc), dtype=torch.float32, device=device)
        self.ops = b.ops
        lvl = 0
        for op in self.ops[self.n_backbone_ops:]:                 # head ops: per-level towers end with their PredOp
            op.level = lvl
            lvl += 1 if op.kind == "pred" else 0
        for op in self.ops:
            op.chain = 0
        self._chain_streams = None
        self._stream_order = self._chained_order() if (STREAM_CHAINS and not self.pair and pafpn is not None and head is not None) else None

    def _chained_order(self):
        """Issue order of the streaming step over three chains: a list of ("op", op, chain, part) and ("dep", from, to).  Level k's
        fusion convs and head ops (chain 1 for the stride-8 level, 2 for stride 16, the main chain for stride 32) follow the
        op that produces PAN output k instead of the end of the neck: the stride-8 head (the largest) runs beside C3_n3 /
        C3_n4, the stride-16 head beside C3_n4.  The stride-32 level is the tail of the frame: its cls tower (second conv +
        prediction launch) forks to chain 1, idle by then.  Every op keeps its own output buffer, so any topological order
        computes the same frame."""
        n_fuse = 6
        neck = self.ops[:self.n_backbone_ops - n_fuse]
        fuse = self.ops[self.n_backbone_ops - n_fuse:self.n_backbone_ops]
        head = self.ops[self.n_backbone_ops:]
        last = {}                                                   # level -> index of the neck op that writes its PAN output
        for k, pan in enumerate(self.cur_pans):
            idx = [i for i, op in enumerate(neck) if op.kind == "conv" and op.y.buf is pan.buf and op.y.c_off == pan.c_off]
            last[k] = idx[-1]
        order, at = [], 0
        for k in range(3):
            order += [("op", op, 0, None) for op in neck[at:last[k] + 1]]
            at = last[k] + 1
            chain = (1, 2, 0)[k]
            level = fuse[2 * k:2 * k + 2] + [op for op in head if op.level == k]
            if chain:
                order.append(("dep", 0, chain))
                order += [("op", op, chain, None) for op in level]
                continue
            cls1 = [op for op in level if op.kind == "conv" and op.tag == "head.cls%d.1" % k]
            pred = level[-1]
            assert len(cls1) == 1 and pred.kind == "pred" and pred.cls_x.buf is cls1[0].y.buf
            shared = [op for op in level[:-1] if op is not cls1[0] and not op.tag.startswith("head.reg%d.1" % k)]
            reg1 = [op for op in level[:-1] if op.tag.startswith("head.reg%d.1" % k)]
            order += [("op", op, 0, None) for op in shared]
            order += [("dep", 0, 1), ("op", cls1[0], 1, None), ("op", pred, 1, "cls")]
            order += [("op", op, 0, None) for op in reg1] + [("op", pred, 0, "reg")]
        assert at == len(neck)
        order += [("dep", 1, 0), ("dep", 2, 0)]
        for e in order:
            if e[0] == "op" and e[3] is None:
                e[1].chain = e[2]
        return order

    def _mark(self, kind, arg=None):
        if self._rec is not None:
            self._rec.mark(kind, arg)

    def _split_decision(self, op, t):
        """(splits, tile) of a streaming-step conv: (1, t) = the ordinary single-pass launch."""
        if not (self.allow_split_k and op.k == 3 and op.stride in (1, 2) and op.y.bs_ is None and (op.res is None or op.res.bs_ is None)):
            return (1, t)
        dec = op._tiles.get("splitk")
        if dec is None:
            dec = (FORCE_SPLIT_K if op.stride == 1 else None) or \
                ops.tuned_splitk(op.x.dtype, op.x.N, op.x.H, op.x.W, op.x.C, op.y.C, self.device, t, op.stride)
            if dec[0] > 1 and op.x.C // (16 if op.x.dtype == ops.DT_F32 else 32) < dec[0]:
                dec = (1, t)
            op._tiles["splitk"] = dec
        return dec

    def _ensure_tuned(self):
        """Make every tuner decision of the streaming step now: the tuners time dummy launches, which must not land on a tape
        that is being recorded."""
        need = 0
        for op in self.ops:
            if op.kind == "conv":
                dec = self._split_decision(op, op.tile("fwd"))
                if op.pre_op is not None:
                    self._fuse_decision(op)
                if dec[0] > 1:
                    need = max(need, dec[0] * op.y.pixels * op.y.C)
        # the fp32 partial-sum scratch is sized HERE, once, for the largest split decision of the plan: a tape that is being
        # recorded holds its raw pointer, so it must not be re-allocated in the middle of a recording
        if need:
            self._splitk_grow(need)

    def _splitk_grow(self, need):
        """fp32 partial-sum scratch, one row per chain (chains run concurrently).  True if it was (re-)allocated."""
        if self._splitk_ws is not None and self._splitk_ws.shape[1] >= need:
            return False
        self._splitk_ws = torch.empty((3 if self._stream_order is not None else 1, max(need, 4 << 20)), dtype=torch.float32,
                                      device=self.device)
        self._stream_tape = None                                     # recorded pointers are stale
        return True

    # -- execution --------------------------------------------------------------------------------
    def _fuse_decision(self, op):
        """True: this Bottleneck 3x3 conv computes its 1x1 predecessor inside its own launch (csrc/bottleneck_fused.h).  Decided once
        per op by measurement against the two launches (ops.tuned_bottleneck) — it pays only where the step is latency-bound."""
        pre = op.pre_op
        if pre is None or not FUSE_BOTTLENECKS or op.x.dtype == ops.DT_F32 or op.x.bs_ is not None or pre.x.bs_ is not None:
            return False
        dec = op._tiles.get("fuse")
        if dec is None:
            t3 = op.tile("fwd")
            sk = self._split_decision(op, t3)                        # the two-launch alternative runs the 3x3 on its streaming tile
            t3 = sk[1] if sk[0] == 1 else t3
            dec = op._tiles["fuse"] = bool(ops.tuned_bottleneck(op.x.dtype, op.x.N, op.x.H, op.x.W, pre.x.C, op.x.C, op.y.C,
                                                                op.res is not None, self.device, pre.tile("fwd"), t3))
        return dec

    def _run_op(self, op, part=None):
        """part: "reg" / "cls" = only that launch of a PredOp (the towers of a level on two chains)."""
        if op.kind == "conv":
            if op.fused_into is not None and self._fuse_decision(op.fused_into):
                return                                                  # computed inside the 3x3 launch that reads it
            if op.pre_op is not None and self._fuse_decision(op):
                pre = op.pre_op
                w, scale, shift = self.cache.conv_eval(op.mod)
                _, s1, b1 = self.cache.conv_eval(pre.mod)
                ops.conv2d(pre.x, w, op.y, 3, 1, scale, shift, res=op.res, epilogue=EPI_SILU, tile=119,
                           wfrag=self.cache.conv_weight_frag(op.mod), pre=(self.cache.conv_weight_frag(pre.mod), s1, b1))
                return
            w, scale, shift = self.cache.conv_eval(op.mod)
            t = op.tile("fwd")
            dec = self._split_decision(op, t) if self._in_stream else (1, t)
            if dec[0] > 1:
                if self._splitk_grow(dec[0] * op.y.pixels * op.y.C):   # (a decision made after _ensure_tuned ran)
                    self._ws_grew = True                                # the pointers of a recording that is open NOW are stale too
                ops.conv2d_splitk(op.x, w, op.y, op.k, op.stride, scale, shift, self._splitk_ws[op.chain], dec[0], res=op.res,
                                  epilogue=EPI_SILU, tile=dec[1], wfrag=self.cache.conv_weight_frag(op.mod))
                return
            t = dec[1]                                                  # (a K-group tile where the streaming tuner found one faster)
            ops.conv2d(op.x, w, op.y, op.k, op.stride, scale, shift, res=op.res, epilogue=EPI_SILU, tile=t,
                       wfrag=self.cache.conv_weight_frag(op.mod) if t >= ops.TILE_WR else None)
        elif op.kind == "resize":
            ops.resize_nearest(op.src, op.dst)
        elif op.kind == "spp":
            ops.spp_pool(op.v)
        elif op.kind == "pred":
            w_ro, b_ro, w_c, b_c = self.cache.pred(op)[:4]
            nch = 5 + self.nc
            base = self.out.data_ptr() + op.a0 * nch * 4
            ybs = self.A * nch
            dec = self.decode
            if part != "cls":
                ops.conv2d(op.reg_x, w_ro, None, 1, 1, None, b_ro, epilogue=EPI_DECODE if dec else EPI_LINEAR,
                           dec_stride=op.stride, y_f32=True, y_ptr=base, y_ld=nch, y_bs=ybs, cout=5)
            if part != "reg":
                ops.conv2d(op.cls_x, w_c, None, 1, 1, None, b_c, epilogue=EPI_SIGMOID, y_f32=True,
                           y_ptr=base + 5 * 4, y_ld=nch, y_bs=ybs, cout=self.nc)
        else:
            raise AssertionError(op.kind)

    def run_backbone(self, x, buffer=None):
        """x: [B,6,H,W] (off_pipe) or [B,3,H,W] (on_pipe) float tensor on self.device.
        on_pipe: `buffer` = the 3 pre-fusion PAN tensors returned by the previous call, or None for the
        first frame, which fuses with itself (node 'star', dfp_pafpn.py:211-214)."""
        x = x.float().contiguous()
        B = self.B
        n_fuse = 6
        if isinstance(x, FramePairsU8):                                # uint8 HWC frames straight to the stem operand
            if self.pair:
                x.pack_focus(_batch_slice(self.f0, 0, B), _batch_slice(self.f0, B, B))
            else:
                x.pack_focus(self.f0)
        elif self.pair:
            ops.focus_pack(x, 0, _batch_slice(self.f0, 0, B))          # current frame  (dfp_pafpn.py:120)
            ops.focus_pack(x, 3, _batch_slice(self.f0, B, B))          # support frame  (:145)
        else:
            ops.focus_pack(x, 0, self.f0)
        for op in self.ops[:self.n_backbone_ops - n_fuse]:
            self._run_op(op)
        if not self.pair:
            # the support inputs of the fusion point at the caller's buffer (or at the current frame: node 'star') for
            # THIS call only; run_stream / its launch tape keep using the plan-owned support buffers
            src = self.cur_pans if buffer is None else buffer
            for dst, s in zip(self.sup_in, src):
                dst.buf = s.buf if isinstance(s, View) else _nhwc_of(s, dst)
        try:
            for op in self.ops[self.n_backbone_ops - n_fuse:self.n_backbone_ops]:
                self._run_op(op)
        finally:
            if not self.pair:
                for dst, own in zip(self.sup_in, self._sup_own):
                    dst.buf = own
        return self.fused

    def run_stream(self, x, first=False):
        """Streaming step with plan-owned state (graph-capturable: no pointer swaps, no allocation):
        fuse the current frame with the PRE-fusion PAN outputs kept from the previous call, then keep the
        current ones for the next call.  `first=True` = node 'star' (fuse with itself)."""
        assert not self.pair
        if isinstance(x, FramePairsU8):
            x.pack_focus(self.f0)
        else:
            ops.focus_pack(x.float().contiguous(), 0, self.f0)
        n_fuse = 6
        self._in_stream = True
        if self._rec is not None and self._stream_order is not None and not first:
            # launch tape: the chained issue order (one stream while recording; sy_tape_replay_n spreads it over the chains)
            try:
                chain = 0
                for e in self._stream_order:
                    if e[0] == "dep":
                        self._mark("dep", (e[1], e[2]))
                        continue
                    if e[2] != chain:
                        chain = e[2]
                        self._mark("cur", chain)
                    self._run_op(e[1], e[3])
                self._mark("cur", 0)
                if not self.decode:
                    ops.head_decode(self.out, boxes=False, obj_sigmoid=True)
            finally:
                self._in_stream = False
            for dst, s in zip(self.sup_in, self.cur_pans):
                ops.view_copy(s, dst)
            return self.out
        try:
            for op in self.ops[:self.n_backbone_ops - n_fuse]:
                self._run_op(op)
            if first:
                for dst, s in zip(self.sup_in, self.cur_pans):
                    ops.view_copy(s, dst)
            for op in self.ops[self.n_backbone_ops - n_fuse:self.n_backbone_ops]:
                self._run_op(op)
            out = self.run_head()
        finally:
            self._in_stream = False
        for dst, s in zip(self.sup_in, self.cur_pans):
            ops.view_copy(s, dst)
        if self._rec is None:
            ops.save_tuned()
        return out

    # ---- launch tape for the streaming step ----------------------------------------------------------------
    # The ~145 launches of a streaming frame re-issued from a launch tape (streamyolo_amd/_lib.py: the C-ABI calls with
    # their marshalled arguments, recorded once) instead of the Python wrappers (~33 us of host time per launch): the host
    # side of a frame drops from 4.6 to 3.4 ms and stays below the 3.7 ms of kernels, so a caller that cannot hold a
    # hipGraph (StreamingDetector: first-frame variant, per-sequence reset) still runs at the graph's latency.
    def _tape_signature(self, x, check_params=True):
        src = (x.cur.data_ptr(), None if x.mirror is None else x.mirror.data_ptr(), x.canvas, x.out_size, x.decimate) \
            if isinstance(x, FramePairsU8) else (x.data_ptr(), x.dtype, tuple(x.shape))
        # check_params=False: the caller guarantees frozen weights (streaming inference) and saves ~0.1 ms of host time
        return src, ((ops.weights_epoch(),) + tuple((p.data_ptr(), p._version) for p in self._tape_params())
                     if check_params else None)

    def _tape_params(self):
        if self._tape_param_list is None:
            seen, lst = set(), []
            for op in self.ops:
                for name in ("mod", "reg_mod", "obj_mod", "cls_mod"):
                    m = getattr(op, name, None)
                    if m is None:
                        continue
                    for t in list(m.parameters()) + list(m.buffers()):
                        if id(t) not in seen:
                            seen.add(id(t)); lst.append(t)
            self._tape_param_list = lst
        return self._tape_param_list

    def run_stream_taped(self, x, post=None, check_params=True):
        """run_stream(x) [+ post(out), e.g. postprocess_device] from a launch tape: the first call with a given input
        buffer / parameter state records it (running the ordinary wrappers), later calls replay the recorded C-ABI
        calls on the current stream.  `x` must live in the SAME device buffer every call (the tape holds its pointer);
        anything `post` does must be C-ABI launches only.  Returns what run_stream / post returned at record time
        (plan-owned output buffers, overwritten by every call).  check_params=False skips the per-call scan of the
        parameter versions (weights must then not change while the tape lives)."""
        assert not self.pair and self.decode
        if not (isinstance(x, FramePairsU8) or (x.dtype == torch.float32 and x.is_contiguous())):
            raise ValueError("run_stream_taped needs a FramePairsU8 or a contiguous fp32 frame (no conversion kernels on the tape)")
        sig = self._tape_signature(x, check_params)
        prog = self._stream_tape
        from . import _lib
        if prog is None or prog[0] != sig or prog[1] is not post:
            # native tape (csrc/tape.hip): the launches are recorded inside the library and replayed by ONE C call
            self._ensure_tuned()
            tape = _lib.NativeTape()
            self._ws_grew = False
            with tape:
                self._rec = tape
                try:
                    out = self.run_stream(x)
                    res = post(out) if post is not None else out
                finally:
                    self._rec = None
            # a scratch buffer re-allocated DURING the recording left launches with the freed pointer on the tape: this call's
            # results are correct (the wrappers ran with the pointers of their moment), the tape is not kept
            self._stream_tape = None if self._ws_grew else (sig, post, tape, res)
            return res
        if self._stream_order is None or not self.out.is_cuda:
            prog[2].replay(ops.stream_of(self.out), None)
        else:
            if self._chain_streams is None:
                self._chain_streams = [torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)]
            prog[2].replay(ops.stream_of(self.out), self._chain_streams[0].cuda_stream, more=(self._chain_streams[1].cuda_stream,))
        return prog[3]

    def export_buffer(self):
        """The current frame's PRE-fusion PAN outputs as NCHW-shaped (channels-last memory) tensors:
        what the reference returns as `buffer_` (dfp_pafpn.py:226) and takes back next frame."""
        return tuple(p.export() for p in self.cur_pans)

    def run_head(self):
        head = self.ops[self.n_backbone_ops:]
        for op in head:
            self._run_op(op)
        if not self.decode:
            # decode_in_inference=False (tal_head.py:220-223): boxes stay raw, obj/cls are still sigmoids
            ops.head_decode(self.out, boxes=False, obj_sigmoid=True)
        return self.out

    def run(self, x, buffer=None):
        self.run_backbone(x, buffer)
        out = self.run_head()
        ops.save_tuned()                   # tuner choices made while this plan ran its first launches (no-op afterwards)
        return out

    def profile(self, x, iters=3):
        """Per-op-kind kernel time (ms per forward), HIP events recorded on the launch stream around
        every op (bench.py's roofline leg; adds host gaps, so not used for the throughput number)."""
        assert x.is_cuda
        x = x.float().contiguous()
        totals = {}
        evs = []

        def timed(kind, fn):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            evs.append((kind, a, b))
        for _ in range(iters):
            B = self.B
            if isinstance(x, FramePairsU8):
                timed("focus", lambda: x.pack_focus(*((_batch_slice(self.f0, 0, B), _batch_slice(self.f0, B, B))
                                                      if self.pair else (self.f0,))))
            elif self.pair:
                timed("focus", lambda: (ops.focus_pack(x, 0, _batch_slice(self.f0, 0, B)),
                                        ops.focus_pack(x, 3, _batch_slice(self.f0, B, B))))
            else:
                timed("focus", lambda: ops.focus_pack(x, 0, self.f0))
            for op in self.ops:
                timed(op.kind, lambda op=op: self._run_op(op))
        torch.cuda.synchronize()
        for kind, a, b in evs:
            totals[kind] = totals.get(kind, 0.0) + a.elapsed_time(b)
        return {k: v / iters for k, v in totals.items()}


def _nhwc_of(t, like):
    """NHWC storage of a caller-supplied NCHW-shaped feature tensor, in the plan's dtype."""
    u = t.permute(0, 2, 3, 1)
    if u.dtype != like.buf.dtype or not u.is_contiguous():
        u = u.to(like.buf.dtype).contiguous()
    assert tuple(u.shape) == (like.N, like.H, like.W, like.C), "on_pipe buffer has the wrong shape"
    return u


def _batch_slice(v, n0, n):
    """Images [n0, n0+n) of a View (same channel slice)."""
    sub = v.buf.view(v.N, v.H, v.W, v.ld)[n0:n0 + n]
    return View(sub, n, v.H, v.W, v.C, v.ld, v.c_off)
