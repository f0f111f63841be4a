"""Training plan: dual-frame forward, Trend-Aware loss and the full backward pass as a static list
of HIP kernel launches (the measured unit of BASELINE.json: StreamYOLO-l fwd+bwd at 600x960).

What the reference does for one `Trainer.train_one_iter` (exps/train_utils/double_trainer.py:95-131):
autocast forward through ~260 eager modules (conv, BN, SiLU, cat, interpolate as separate
kernels), a host-synchronising assignment loop, then autograd's generic backward.  Here:

  forward   per BaseConv: MFMA conv emitting raw output + per-channel sum / sum^2 (fp32 accumulators)
            -> sy_bn_finalize (batch statistics, running-stat update with the module's momentum/eps —
            trap T1) -> sy_bn_silu_apply (+ Bottleneck / DFP residual) into the consumer's channel
            slice.  Current frame first, then support frame, with SEPARATE statistics (trap T2).
  loss      raw [B, A, 5+nc] fp32 -> SimOTA + TAL loss and its closed-form gradient (csrc/tal_loss.hip) -> d_raw.
  backward  reverse walk: BN/SiLU backward (reduce + apply), MFMA wgrad straight into a flat fp32
            gradient arena laid out like the parameters (so .grad tensors are views of it and the
            RCCL all-reduce is ONE collective over one buffer), MFMA dgrad into gradient mirrors of
            the activation buffers (fan-in = accumulate epilogue; torch.cat backward = channel slices).
  DDP       one process per GPU; `TrainStep` all-reduces the arena over RCCL (xGMI) once per step.

`train_forward` is the drop-in path (YOLOX.forward in training mode returns the reference's loss
dict whose total_loss.backward() fills .grad through one autograd.Function); `TrainStep` is the
sync-free fast path bench.py and a native trainer use.
"""
import ctypes as C
import os

import torch

from . import ops
from .data import FramePairsU8
from .engine import _Builder, MergedConv, base_convs, build_frame_net, build_fuse_net, build_head_net
from .model.plan_cache import compute_dtype_for
from .ops import View, EPI_LINEAR, CONV_DGRAD


# one launch for BatchNorm finalize + BN.SiLU apply (sy_bn_finalize_apply); "0": the two separate launches (A/B timing)
# (default off: kernel time 2.77 vs 2.90 ms per l step, but the step itself is not faster at 32 statistic replicas, and the 8
# replicas that make it 0.1-0.26 ms faster break the 1e-5 run-to-run reproducibility of the fp16 step — profiles/r02/t_*, u_*,
# profiles/r04/d_bench_ff8.json)
# Round 5: ON for the 16-bit modes together with 4 statistic replicas (l step 20.46-20.50 vs 20.64-20.67 ms, profiles/r05 stage l:
# the replicas every workgroup of the apply pass re-reads shrink 8x, and -216 launches on the forward chains); the run-to-run
# reproducibility that argued for 32 replicas is the exact mode's business now (one replica row per workgroup, bit-equal steps).
# "auto" (default): 16-bit compute dtypes only; "1" / "0": everywhere / nowhere (A/B timing).
_FUSED_FINALIZE = __import__("os").environ.get("STREAMYOLO_FUSED_FINALIZE", "auto")
# Round 5: the BatchNorm-backward REDUCE of a Bottleneck's 1x1 layer can ride in the epilogue of the 3x3 layer's data gradient (the
# only producer of its activation gradient: engine.ConvOp.pre_op) — sy_conv2d's SY_EPI_BNR; 72 launches and one read of dA less per
# l step.  16-bit modes, frames as stream-parallel chains (per-frame launches).  Built, parity-green (kernel + model tests on the
# MI355X) and measured: OFF by default — the reduce pass leaves the chains (-0.87 ms of kernel time), but its sigmoid / product work
# (80 values per lane) then runs in the epilogue of a one-wave-per-SIMD MFMA kernel instead of in a memory-bound launch that other
# kernels overlap: data gradients +0.6 ms, the l step 20.78-20.82 vs 20.68-20.75 ms (profiles/r05 stage m).  "1": on (A/B timing).
BNR_FUSION = __import__("os").environ.get("STREAMYOLO_BNR_FUSION", "0") != "0"
BNR_TILES = (98, 100, 101, 109, 104, 107, 117, 118)     # data-gradient tiles whose staged epilogue carries it (csrc/conv_igemm.hip)
# exact (fp32) mode: one statistics replica row per workgroup -> run-to-run bit-equal steps (TrainPlan.__init__); "0" = the
# replica counts of the speed modes (A/B switch)
EXACT_STATS = __import__("os").environ.get("STREAMYOLO_EXACT_STATS", "1") != "0"
# The workgroup cap of the BatchNorm-backward APPLY pass AS SCHEDULED in the step: the library's own default (1024) is that kernel's
# optimum from round 2; beside round 5's frame chains and weight-gradient streams fewer, fatter workgroups leave the MFMA kernels
# their CUs: 512 -> l step 20.83-20.88 vs 21.01-21.03 ms and 20.92 vs 21.00-21.14 on another box (256: 21.40-21.44; the forward apply
# / backward reduce caps: 1024 / 512 instead of 2048 / 768 nothing on top — profiles/r05 stages zm, zn).  The pass is a grid-stride
# map over pixel rows (channel sums written by workgroup 0): its results do not depend on the grid.  Set EXPLICITLY through
# sy_bn_grid_caps when a plan is built (round 6; it was an environment write at import time, which a library that had already read
# its caps ignored and which leaked into child processes — ADVICE r05); an explicit SY_BN_BAPPLY_BLOCKS in the environment wins.
BN_BAPPLY_BLOCKS_SCHEDULED = 512


def _schedule_bn_caps():
    """Caps of the row kernels for a training plan -> the backward-reduce cap in force (the exact mode's replica-row count)."""
    if os.environ.get("SY_BN_BAPPLY_BLOCKS"):
        return ops.bn_grid_caps()["bwd_reduce"]
    return ops.bn_grid_caps(bwd_apply=BN_BAPPLY_BLOCKS_SCHEDULED)["bwd_reduce"]


# Measured and REMOVED in round 4 (profiles/r04/README.md): BatchNorm finalisation by the producing convolution's last workgroup
# (every statistics launch got 8-13 us slower — each workgroup waits for its own atomics and a ticket round trip — the l step
# 22.6-23.9 vs 22.3-23.0 ms at 2 ... 32 replicas) and BatchNorm backward as one resident launch whose workgroups wait for each
# other (parity-green on the MI355X, the step 24.58 vs 22.70 ms: spinning workgroups hold the CUs the other chains need).
# Round 6 built that launch AGAIN (commit 13db7f2, reverted: sy_bn_silu_bwd_fused — rows held in registers across a per-channel-slice
# rendezvous, the largest piece per thread that keeps a launch within half the chip's guaranteed residency, one lane per workgroup
# polling every ~0.5 us, bounded spin; kernel + model parity green on the MI355X) on the two frame chains: 20.83-20.88 vs 19.96 ms with
# the 38x60 / 19x30 layers fused (114 launches), 20.75-20.89 vs 20.76-20.81 ms with only the 19x30 layers (same-box alternating runs,
# profiles/r06 stages i, j) although tools/ablate_step.py prices the reduce launches at 1.77 ms of the step.  Removing launches from
# the chains does not pay here; a launch that holds CUs while it waits costs more than the second read it saves.

class _GradSpace:
    """Gradient mirrors of activation buffers + first-write / accumulate bookkeeping per channel range."""

    def __init__(self):
        self.mirror = {}          # id(buf tensor) -> grad tensor
        self.written = {}         # id(buf tensor) -> list of (c0, c1)
        self.py = lambda fn: fn() # torch-op hook (TrainPlan records these snippets on its launch tape)

    def reset(self):
        self.written = {}

    def view(self, v):
        if v.root is not None:                       # frame-pair buffer: one mirror for both frames, sliced
            full, n0 = v.root
            gf = self.mirror.get(id(full))
            if gf is None:
                gf = torch.empty_like(full)
                self.mirror[id(full)] = gf
            return View(gf[n0:n0 + v.N], v.N, v.H, v.W, v.C, v.ld, v.c_off, v.bs_, root=(gf, n0))
        g = self.mirror.get(id(v.buf))
        if g is None:
            g = torch.empty_like(v.buf)
            self.mirror[id(v.buf)] = g
        return View(g, v.N, v.H, v.W, v.C, v.ld, v.c_off, v.bs_)

    def target(self, v):
        """(grad view, accumulate?) for writing the gradient of activation view `v`."""
        key = (id(v.root[0]), v.root[1]) if v.root is not None else id(v.buf)
        iv = self.written.setdefault(key, [])                # sorted, merged, disjoint [a, b) channel ranges
        c0, c1 = v.c_off, v.c_off + v.C
        assert v.bs_ is None and c1 <= v.ld
        gaps, pos = [], c0
        for a, b in iv:
            if b <= pos or a >= c1:
                continue
            if a > pos:
                gaps.append((pos, a))
            pos = max(pos, b)
        if pos < c1:
            gaps.append((pos, c1))
        g = self.view(v)
        if len(gaps) == 1 and gaps[0] == (c0, c1):
            accumulate = False                                   # nothing written yet: plain store
        else:
            accumulate = True                                    # (partly) written: zero the holes, then +=
            flat = g.buf.view(-1, v.ld)
            for a, b in gaps:
                ops.zero(flat[:, a:b])                           # sy_zero_rows: a launch of the plan's tape (round 6: was an ATen fill
                                                                 # behind a return to Python at every such hole, ~30 per l step)
        if gaps:
            merged = []
            for a, b in sorted(iv + [(c0, c1)]):
                if merged and a <= merged[-1][1]:
                    merged[-1] = (merged[-1][0], max(merged[-1][1], b))
                else:
                    merged.append((a, b))
            iv[:] = merged
        return g, accumulate


class _PairBuilder(_Builder):
    """Allocates every buffer of the per-frame network for BOTH frames at once ([2N, H, W, C]): the first pass
    hands out the first halves, the replay pass the second halves in the same order.  The backward pass can
    then run the shared-weight dgrad / wgrad of a layer over both frames in ONE launch."""

    def __init__(self, dtype, device):
        super().__init__(dtype, device)
        self.fulls, self.replay, self.paired = [], None, True

    def buf(self, N, H, W, C):
        if not self.paired:
            return super().buf(N, H, W, C)
        if self.replay is None:
            full = torch.empty((2 * N, H, W, C), dtype=ops.TORCH_DTYPE[ops.dtype_code(self.dtype)], device=self.device)
            self.fulls.append(full)
            half = 0
        else:
            full = self.fulls[self.replay]
            self.replay += 1
            half = 1
            assert tuple(full.shape) == (2 * N, H, W, C)
        return View(full[half * N:(half + 1) * N], N, H, W, C, root=(full, half * N))


class StagedWeights:
    """Compute-dtype operand layouts of every convolution of the training plan, re-derived from the fp32 OIHW
    nn.Parameters by ONE sy_pack_weights launch per step (the optimizer rewrites the parameters every iteration —
    exps/train_utils/double_trainer.py:114-119 — so nothing here is cached across steps).  Same lookup surface as
    engine.ParamCache: conv_weight / conv_weight_frag / pred."""

    def __init__(self, plan_ops, dtype, device):
        from . import _lib
        self.dtype, self.device = dtype, device
        tdt = ops.TORCH_DTYPE[dtype]
        epc = 16 // torch.empty(0, dtype=tdt).element_size()
        bk = 4 * epc
        self.conv, self.preds, self.sources = {}, {}, []
        rows = []

        def z(n, dt=tdt):
            return torch.zeros(n, dtype=dt, device=device)

        def entry(w, packed, packed_t, frag, frag_t, co_n, ci_n, taps, r0, R, R_t, CI, dt=dtype):
            e = _lib.PackEntry()
            e.w = w.data_ptr()
            e.packed, e.packed_t = _ptr(packed), _ptr(packed_t)
            e.frag, e.frag_t = _ptr(frag), _ptr(frag_t)
            e.co_n, e.ci_n, e.taps, e.r0, e.R, e.R_t, e.CI, e.dtype = co_n, ci_n, taps, r0, R, R_t, CI, dt
            rows.append(e)
            self.sources.append(w)

        def _ptr(t):
            return None if t is None else t.data_ptr()

        self.bn = {}                     # id(MergedConv) -> (gamma stack, beta stack) fp32, refreshed with the weights
        for op in plan_ops:
            if op.kind == "conv" and id(op.mod) not in self.conv:
                parts = base_convs(op.mod)                                 # one BaseConv, or the stacked parts of a MergedConv
                w0 = parts[0].conv.weight
                ci, kh, kw = w0.shape[1:]
                co = sum(m.conv.weight.shape[0] for m in parts)
                taps, CI = kh * kw, (16 if ci == 12 else ci)           # Focus stem: 12 -> 16 channels (zero weights)
                packed, packed_t = z(co * taps * CI).view(co, taps * CI), z(CI * taps * co).view(CI, taps * co)
                frag = z(-(-co // 32) * 32 * taps * CI) if CI % bk == 0 else None
                frag_t = z(-(-CI // 32) * 32 * taps * co) if co % bk == 0 else None
                r0 = 0
                for m in parts:
                    w = m.conv.weight
                    assert w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape[1:]) == (ci, kh, kw)
                    entry(w, packed, packed_t, frag, frag_t, w.shape[0], ci, taps, r0, co, co, CI)
                    r0 += w.shape[0]
                self.conv[id(op.mod)] = (packed, packed_t, frag, frag_t)
                if len(parts) > 1:                                     # stacked BatchNorm affine parameters (fp32 copies)
                    gs, bs_ = z(co, torch.float32), z(co, torch.float32)
                    r0 = 0
                    for m in parts:
                        c = m.bn.weight.numel()
                        entry(m.bn.weight, gs, None, None, None, c, 1, 1, r0, co, co, 1, ops.DTYPE_NAME["fp32"])
                        entry(m.bn.bias, bs_, None, None, None, c, 1, 1, r0, co, co, 1, ops.DTYPE_NAME["fp32"])
                        r0 += c
                    self.bn[id(op.mod)] = (gs, bs_)
            elif op.kind == "pred":
                cin, nc = op.reg_mod.weight.shape[1], op.cls_mod.weight.shape[0]
                w_ro, w_ro_t = z(5 * cin).view(5, cin), z(cin * 8).view(cin, 8)
                w_c, w_c_t = z(nc * cin).view(nc, cin), z(cin * nc).view(cin, nc)
                b_ro = z(5, torch.float32)
                entry(op.reg_mod.weight, w_ro, w_ro_t, None, None, 4, cin, 1, 0, 5, 8, cin)
                entry(op.obj_mod.weight, w_ro, w_ro_t, None, None, 1, cin, 1, 4, 5, 8, cin)
                entry(op.cls_mod.weight, w_c, w_c_t, None, None, nc, cin, 1, 0, nc, nc, cin)
                entry(op.reg_mod.bias, b_ro, None, None, None, 4, 1, 1, 0, 5, 5, 1, ops.DTYPE_NAME["fp32"])
                entry(op.obj_mod.bias, b_ro, None, None, None, 1, 1, 1, 4, 5, 5, 1, ops.DTYPE_NAME["fp32"])
                self.preds[id(op)] = (w_ro, b_ro, w_c, op.cls_mod.bias, w_ro_t, w_c_t)
        arr, tiles = _lib.pack_table(rows)
        self.table = (torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(rows), tiles)
        self.n = len(rows)
        self.sig = tuple(t.data_ptr() for t in self.sources)

    def valid(self):
        return self.sig == tuple(t.data_ptr() for t in self.sources)

    def refresh(self):
        """Stage the weights (one launch).  (Staging stem + dark2 first and the rest beside them on an idle stream was measured
        neutral in round 3 — 22.45-22.55 vs 22.48-22.50 ms, profiles/r03/t_* — and removed.)"""
        table, n, tiles = self.table
        ops.check(ops._lib.lib().sy_pack_weights(table.data_ptr(), n, tiles, ops.stream_of(table)), "sy_pack_weights")

    def conv_weight(self, mod, transpose=False):
        return self.conv[id(mod)][1 if transpose else 0]

    def conv_weight_frag(self, mod, transpose=False):
        return self.conv[id(mod)][3 if transpose else 2]

    def pred(self, op):
        w_ro, b_ro, w_c, b_c, w_ro_t, w_c_t = self.preds[id(op)]
        return w_ro, b_ro, w_c, b_c.detach(), w_ro_t, w_c_t


CSP_FORK = os.environ.get("STREAMYOLO_CSP_FORK", "1") != "0"
MERGE_SIBLINGS_TRAIN = os.environ.get("STREAMYOLO_MERGE_TRAIN", "1") != "0"
# VERDICT r02 "next" #3, measured NEGATIVE (profiles/r03/e_*): with the producer's BatchNorm.SiLU applied by the 3x3 consumer in
# LDS the l step takes 24.90-24.93 ms against 24.10 ms on the same box — the normalising halo kernel costs 12-22 us more per
# launch than the plain one (+20-30 %: the extra LDS round trip and barrier per channel slab are not hidden), about what the
# apply pass it would replace costs (tools/norm_probe.py: -0.12 ms over the 48 candidate launches), and the apply pass is
# still needed by the backward pass.  Off by default; the kernel, its parity test and the probe stay.
# The two frames of a pair as two stream-parallel chains in the forward pass (current frame on the main stream, support frame on
# the side stream, half-size launches issued alternately): one chain's HBM-bound BatchNorm passes and kernel tails fill the
# other's MFMA-bound convolutions.  Measured (profiles/r03/f_*): 23.18-23.24 vs 24.02 ms per l step at 8 pairs, 14.75 vs 15.28
# at 4 — although the half-size kernels are individually slower (kernel sum 18.9 vs 18.0 ms).
FWD_SPLIT_FRAMES = os.environ.get("STREAMYOLO_FWD_SPLIT_FRAMES", "1") != "0"
# ... and in the backward pass: BatchNorm backward + data gradient of the two frames as chains on streams 0 and 2, the (paired)
# weight gradient of the layer on stream 1 behind both.
# Measured (profiles/r03/g_*, j_*): l at 8 pairs 22.7-23.0 vs 23.36 ms, l at 4 pairs 14.79 vs 14.75, m 15.45 (both) vs 15.83 (none),
# s 8.05 vs 7.96 — it pays once a frame's layers are big enough to keep the chip busy in half-size launches: "auto" turns it on
# for B * H * W * width^2 >= 2e6 (l from 4 pairs, m from 8 at 600x960; not s).  "1" / "0" force it.
BWD_SPLIT_FRAMES = os.environ.get("STREAMYOLO_BWD_SPLIT_FRAMES", "auto")
# On the same machinery: head levels 1-2 backward as a chain of their own (stream 2) beside level 0; weight gradients alternating
# between two streams (1 and 3, a split-K workspace each) so that one layer's fold runs beside the next layer's wgrad.  Measured
# on one box (profiles/r03/l_*): 22.66-22.77 (neither) / 22.73 (head chain) / 22.54 (two wgrad streams) / 22.27 ms (both).
HEAD_BWD_CHAINS = os.environ.get("STREAMYOLO_HEAD_BWD_CHAINS", "1") != "0"
# Round 5, forward: a head level (+ its two DFP fusion convs) starts as soon as its PAN output exists in both frames, on a third
# chain beside the rest of the frame chains (TrainPlan._early_levels).  "0": the levels behind the join of the frame chains.
HEAD_EARLY = os.environ.get("STREAMYOLO_HEAD_EARLY", "1") != "0"
# Round 6: stream-wait packets on the frame chains are not free.  A timing experiment that dropped every ring-slot event of the
# backward pass (results invalid, profiles/r06 stage u) ran the l step in 20.39 instead of 21.09 ms, dropping the chain -> weight-gradient
# dependencies on top 20.12-20.17: each hipStreamWaitEvent is a barrier packet the chain's queue stops at (~3 us), signalled or not,
# and a frame chain issued two of them per layer.  In the split backward a chain's half of a raw-gradient slot is only ever rewritten
# by the SAME chain (stream order) — its own "slot done" event (STREAMYOLO_CHAIN_SLOT_DONE=1 restores it) made the OTHER chain wait for
# nothing — and the weight gradients retire in order on their streams, so a chain needs to wait for them once per STREAMYOLO_SLOT_BATCH
# layers (on the newest slot of each weight-gradient stream), not once per layer.
CHAIN_SLOT_DONE = os.environ.get("STREAMYOLO_CHAIN_SLOT_DONE", "0") != "0"
# Measured (profiles/r06 stage v, same box): both events per layer (rounds 3-5) 20.61 / 20.64 ms; no chain-side event 20.20 / 20.18;
# + a wait every 2 / 3 / 4 layers 20.18 / 20.12, 20.11 / 20.14 (3 or 4 with a ring of 12: 20.12 / 20.10) -> default 3.
SLOT_BATCH = max(1, int(os.environ.get("STREAMYOLO_SLOT_BATCH", "3")))
# (The chain -> weight-gradient dependencies were sized the same way — 0.3 ms without them, stage x — but they are real dependencies,
#  not packet overhead: issuing the split backward's weight gradients in batches of 2 ... 6 layers behind ONE event per chain bought
#  nothing, 20.12-20.24 vs 20.11-20.12 ms, stage y; removed.)
DUAL_WGRAD = os.environ.get("STREAMYOLO_DUAL_WGRAD", "1") != "0"
WGRAD_STREAMS = [1, 3, 4, 5][:max(1, min(4, int(os.environ.get("STREAMYOLO_WGRAD_STREAMS", "2")) if DUAL_WGRAD else 1))]
# (Weight gradients on streams of their own instead of sharing stream 1 with the forward pass's support-frame chain, with or without
# a low hardware queue priority: 22.78-22.79 vs 22.70 ms — profiles/r04/b_bench_ownw*.json — removed.)


NINE_TAP_WGRAD_TILES = (52, 59, 60)     # conv_wgrad9_kernel (csrc/conv_wgrad.hip)


def scheduled_wgrad(wt, cin, env=None):
    """(tile, target workgroups) of a weight-gradient launch AS SCHEDULED in the training step, from the per-kernel tuner's choice
    `wt`.  The tuner times kernels alone on an idle chip; in the step the weight gradients run on side streams BESIDE the frame
    chains, and what counts is the CU time they take away from those chains, not their own duration:
      * split-K workgroups capped at 512 (the tuner likes ~1024): -0.3 ms per l step although the kernels themselves get ~2 % slower
        (round 2, tools/gpu_sweep2.sh);
      * the all-nine-taps kernel on 128 workgroups: it takes ~1.8x as long, the step is 0.6 ms shorter (22.4-22.8 -> 21.8-22.1 ms; 64 /
        192 / 384 worse — profiles/r04 stage s; re-measured with the <= 256-register variants in round 5, stage q: same optimum);
      * wherever the layer has a multiple of 64 input channels, the eight-wave variant (64 input channels per workgroup, the dy slab
        staged once for twice the MFMAs) on 96 workgroups: alone no faster than tile 52 (668 vs 671 TF/s at 256->256 @38x60 x 16), in
        the step it frees a quarter of the CUs tile 52 held — l 20.77-20.89 vs 20.93-21.00 ms, m 14.43-14.45 vs 14.52-14.56, 4 pairs
        13.43 vs 13.50, s 7.57 vs 7.59 (profiles/r05 stages zb-zd; 80 / 112 / 128 workgroups: 20.91-21.03, 64 / 256: slower than tile
        52).  Tile 60 = that variant with the pipelined slab loop: the kernel itself -4 % on these 96 workgroups, the step 20.88-20.89
        vs 20.94-20.96 ms (stages zf, zg) — the step does not wait for this kernel, it shares the L2 / HBM path with it."""
    env = os.environ if env is None else env
    tile, blocks = wt
    cap = int(env.get("STREAMYOLO_WGRAD_BLOCKS_CAP", "512"))
    if cap > 0 and blocks > cap:
        blocks = cap
    if tile in NINE_TAP_WGRAD_TILES:
        cap9 = int(env.get("STREAMYOLO_WGRAD9_BLOCKS", "128"))
        wide9 = int(env.get("STREAMYOLO_WGRAD9_WIDE", "60"))               # 0: off
        if wide9 and cin % 64 == 0:
            tile, cap9 = wide9, int(env.get("STREAMYOLO_WGRAD9_WIDE_BLOCKS", "96"))
        if cap9 > 0 and blocks > cap9:
            blocks = cap9
    return (tile, blocks)


def _csp_role(tag):
    """'conv2' / 'conv3' for the two CSPLayer convs named <csp>.conv2 / <csp>.conv3 by engine._Builder.csp (the
    Bottleneck convs are <csp>.m.<i>.conv1/2, SPP's are spp.conv1/2)."""
    head, _, leaf = tag.rpartition(".")
    if leaf not in ("conv2", "conv3") or ".m." in tag or head in ("", "spp"):
        return None
    return leaf


class TrainPlan:
    # replicas of each BN-backward reduction: 2, folded by every workgroup of the apply pass itself (no fold launch);
    # measured 28.3 vs 28.6 ms per l step against 16 replicas + fold kernel (profiles/r01/f_*)
    BWD_COPIES = int(os.environ.get("STREAMYOLO_BWD_COPIES", "2"))
    # replicas of each conv's sum / sum^2 arrays: 4 are as fast as 32 but the longer fp32 atomic chains make the batch
    # variance (E[y^2] - mean^2) visibly order-dependent (tape-replay test: 1e-4 instead of 1e-6 between identical steps)
    # (round 5: 4 in the 16-bit modes — there the order noise is far below the rounding step, and the fused finalize + apply launch
    #  reads every replica in every workgroup; 32 stays the fp32 default when the exact one-row-per-workgroup layout is switched off)
    STAT_COPIES = int(os.environ.get("STREAMYOLO_STAT_COPIES", "0"))     # 0 = by compute dtype: 4 (16-bit) / 32 (fp32)
    WGRAD_WS_BYTES = 256 << 20   # split-K slabs of sy_conv2d_wgrad
    # raw-gradient scratch: the weight gradient of layer i (side streams) overlaps BatchNorm backward / data gradient of the layers
    # behind it.  Rounds 2-5 recycled a RING of slots (5, then 9 of the largest raw gradient: a frame chain that wants a slot back waits
    # for the weight gradient that last read it) — round 6 found that the WAITING is not the cost, the stream-wait packets are: with a
    # buffer per layer (RING = 0: ~4 GB for l at 8 pairs of the GPU's 288, no slot events at all) the l step is 0.24 ms shorter than
    # with the ring and the cheapest event schedule, 0.7 ms (8 pairs) / 0.9 ms (4 pairs) shorter than rounds 3-5's (profiles/r06
    # stages u-w).  RING > 0 keeps the ring (memory-tight hosts, A/B timing).
    RING = int(os.environ.get("STREAMYOLO_RING", "0"))
    STREAMS = int(os.environ.get("STREAMYOLO_STREAMS", "2"))   # 1: everything on the caller's stream
    TAPE_ON_CPU = True           # the SIMT-emulator test runs replay launch tapes too (same code path as the GPU)

    def __init__(self, model, B, H, W, dtype, device, parts="full", feat_shapes=None, pool=None):
        """parts: "full" = YOLOX (backbone + head, the fused forward + loss + backward plan); "backbone" = a DFPPAFPN alone
        (frames -> fused FPN features, gradients of the features in); "head" = a TALHead alone on given fused features
        (`feat_shapes` = [(C, H, W)] per level) — the stand-alone training-mode entry points of the sub-modules, which the
        reference's YOLOX.forward composes (exps/model/yolox.py:32-38)."""
        self.model, self.B, self.H, self.W, self.device = model, B, H, W, device
        self.parts = parts
        self.pool = pool                     # model-level PlanCache: shared scratch across the plans of different input sizes
        self.dtype = ops.dtype_code(dtype)
        self.tdtype = ops.TORCH_DTYPE[self.dtype]
        pafpn = model.backbone if parts == "full" else (model if parts == "backbone" else None)
        head = model.head if parts == "full" else (model if parts == "head" else None)
        self.head = head
        self.cache = None                    # StagedWeights, built below once the plan's ops exist
        b = _PairBuilder(self.dtype, device)
        # sibling convolutions (CSP conv2 || conv1, the first cls / reg tower conv of a head level) as ONE conv / BatchNorm /
        # dgrad / wgrad launch each, as in the inference plans (engine.MergedConv): -88 launches per l step, no `+=` dgrad pass
        b.merge_siblings = MERGE_SIBLINGS_TRAIN
        if pafpn is not None:
            self.f0_cur, cur = build_frame_net(b, pafpn, B, H, W)          # current frame  (dfp_pafpn.py:120-140)
            self.n_frame_ops = len(b.ops)
            b.replay = 0
            self.f0_sup, sup = build_frame_net(b, pafpn, B, H, W)          # support frame  (:145-165), same weights
            assert len(b.ops) == 2 * self.n_frame_ops and b.replay == len(b.fulls)
            b.paired = False
            fused = build_fuse_net(b, pafpn, cur, sup)                      # (:168-170)
        else:
            self.n_frame_ops = 0
            b.paired = False
            fused = tuple(b.buf(B, h, w, c) for c, h, w in feat_shapes)
        self.fused = fused
        self.n_head_start = len(b.ops)
        self.preds, self.A = (build_head_net(b, head, fused) if head is not None else ([], 0))
        lvl = 0
        for op in b.ops[self.n_head_start:]:                             # per-level towers end with their PredOp
            op.level = lvl
            lvl += 1 if op.kind == "pred" else 0
        self.hw = [(f.H, f.W) for f in fused]
        self.ops = b.ops
        self.nc = head.num_classes if head is not None else 0
        nch = 5 + self.nc
        self.raw = torch.empty((B, self.A, nch), dtype=torch.float32, device=device)
        self.dpad = torch.zeros((B, self.A, 16), dtype=self.tdtype, device=device)

        # ---- per-op training state carved out of flat arenas (zeroed with one memset each) ---------
        convs = [op for op in self.ops if op.kind == "conv"]
        tot_c = sum(op.y.C for op in convs)
        # Replicas of the statistics arrays, per conv.  Speed modes: STAT_COPIES / BWD_COPIES everywhere (fp32 atomics of many
        # workgroups per replica: the sums depend on their arrival order in the last bits).  EXACT mode (fp32 compute, round 5):
        # every workgroup gets a replica row of its own — one `0 + x` atomic per row and channel — and the finalize / fold
        # launches add the rows in index order, so two identical steps are bit-equal and SimOTA's discrete decisions
        # (tal_head.py:679-712) no longer flip between runs.  Rows: forward = pixel tiles of a frame's launch (<= N * H *
        # ceil(W / 32), whatever tile the tuner picks); backward reduce = its workgroup cap (SY_BN_REDUCE_BLOCKS, 768).
        self.exact_stats = EXACT_STATS and self.dtype == ops.DT_F32
        self.stat_copies = self.STAT_COPIES if self.STAT_COPIES > 0 else (32 if self.dtype == ops.DT_F32 else 4)
        # (never with the exact mode's one-row-per-workgroup statistics: sy_bn_finalize_apply does not fold the thousands of rows —
        #  only sy_bn_finalize's fold_rows_kernel does, and the running-statistics launch reads the folded rows — ADVICE r05)
        self.fused_finalize = ((_FUSED_FINALIZE == "1") or (_FUSED_FINALIZE == "auto" and self.dtype != ops.DT_F32)) and not self.exact_stats
        reduce_cap = _schedule_bn_caps()       # the library's own figure: one replica row per reduce workgroup in the exact mode

        def copies_of(op):
            if not self.exact_stats:
                return self.stat_copies, self.BWD_COPIES
            return max(self.stat_copies, op.y.N * op.y.H * ((op.y.W + 31) // 32)), max(self.BWD_COPIES, reduce_cap)
        for op in convs:
            op.stat_copies, op.bwd_copies = copies_of(op)
        self.stat_arena = torch.zeros(2 * sum(op.stat_copies * op.y.C for op in convs), dtype=torch.float32, device=device)  # [sum | sumsq] x copies
        self.bwd_arena = torch.zeros(2 * sum(op.bwd_copies * op.y.C for op in convs), dtype=torch.float32, device=device)   # [copies][sum dz | sum dz*xhat]
        self.aff_arena = torch.empty(4 * tot_c, dtype=torch.float32, device=device)       # scale|shift|mean|invstd
        sq0 = self.stat_arena.numel() // 2                     # the sumsq half of the statistics arena
        off = 0
        max_raw = 0
        nf = self.n_frame_ops
        for i in range(nf):                                      # raw conv outputs of a frame pair share one tensor
            a, b2 = self.ops[i], self.ops[nf + i]
            if a.kind == "conv":
                full = torch.empty((2 * a.y.N, a.y.H, a.y.W, a.y.C), dtype=self.tdtype, device=device)
                a.yraw = View(full[:a.y.N], a.y.N, a.y.H, a.y.W, a.y.C, root=(full, 0))
                b2.yraw = View(full[a.y.N:], a.y.N, a.y.H, a.y.W, a.y.C, root=(full, a.y.N))
        # Units: (current-frame op, support-frame op) of the shared per-frame network, then the single fusion / head
        # convs.  A unit's state is laid out segment-major — statistics [seg][copies][C], backward sums
        # [seg][copies][2][C], affine [seg][C] — so ONE launch per layer serves both frames (nseg = 2) while each
        # frame keeps its own batch statistics, like the reference's two backbone passes (dfp_pafpn.py:120-165).
        units = [(self.ops[i], self.ops[nf + i]) for i in range(nf) if self.ops[i].kind == "conv"]
        units += [(op,) for op in self.ops[2 * nf:] if op.kind == "conv"]
        s_off = b_off = 0
        for unit in units:
            C, S = unit[0].y.C, len(unit)
            SC, BC = unit[0].stat_copies, unit[0].bwd_copies
            u_sum = self.stat_arena[s_off:s_off + SC * S * C]
            u_sq = self.stat_arena[sq0 + s_off:sq0 + s_off + SC * S * C]
            u_bsum = self.bwd_arena[b_off:b_off + 2 * BC * S * C]
            s_off += SC * S * C
            b_off += 2 * BC * S * C
            u_aff = tuple(self.aff_arena[k * tot_c + off:k * tot_c + off + S * C] for k in range(4))
            for s_, op in enumerate(unit):
                op.stat = (u_sum[s_ * SC * C:(s_ + 1) * SC * C], u_sq[s_ * SC * C:(s_ + 1) * SC * C])
                op.bsum = u_bsum[s_ * 2 * BC * C:(s_ + 1) * 2 * BC * C]
                op.aff = tuple(t[s_ * C:(s_ + 1) * C] for t in u_aff)
                if op.yraw is None:
                    op.yraw = View.alloc(op.y.N, op.y.H, op.y.W, C, self.dtype, device)
                max_raw = max(max_raw, 2 * op.y.pixels * C)
            unit[0].unit = (u_sum, u_sq, u_bsum, u_aff)             # the whole unit's arrays (paired launches)
            off += S * C
        # raw-gradient scratch ring: the weight-gradient kernels of layer i run on the side stream while the main
        # stream is already producing layer i-1's raw gradient, so a slot is reused only after its wgrad retired
        self.max_raw = max_raw
        self._bind_scratch()
        self.ring_i = 0
        self.side = torch.cuda.Stream(device=device) if (device.type == "cuda" and self.STREAMS > 1) else None
        w_ = float(getattr(pafpn, "width", 1.0)) if pafpn is not None else 0.0
        self.bwd_split = (BWD_SPLIT_FRAMES == "1") or (BWD_SPLIT_FRAMES == "auto" and B * H * W * w_ * w_ >= 2.0e6)
        self.side2 = torch.cuda.Stream(device=device) if (self.side is not None and (self.bwd_split or HEAD_BWD_CHAINS)) else None
        self.side_w = {k: torch.cuda.Stream(device=device) for k in WGRAD_STREAMS[1:]} if self.side is not None else {}
        self.side3 = self.side_w.get(3)
        self._chain, self._wg_stream, self._wg_flip = 0, 1, 0
        self.tuned = False                    # the first step (autotuning) runs on one stream
        self.force_serial = False             # profile(): per-kernel durations without overlap
        self._ev_pool = []
        for op in self.ops:
            if op.kind == "spp":
                op.argmax = torch.empty((op.v.N, op.v.H, op.v.W, 3, op.v.C // 4), dtype=torch.uint8, device=device)
        self.cache = StagedWeights(self.ops, self.dtype, device)
        self.loss_ws = None
        self._zeroed, self._zero_ev, self._packed_for = None, None, None
        self.run_table = None
        self.grads = _GradSpace()
        self.grads.py = self._py
        self.programs, self._rec, self._param_sig = {}, None, None
        self._side_region = False
        self.pending = 0                      # forwards of an autograd node whose backward has not run yet (+ 1 while a TrainStep
                                              # holds the plan): the plan cache never evicts (releases) such a plan

        # ---- flat gradient arena in parameter layout ------------------------------------------------
        # Parameter order of the arena = model.parameters() order, except that the parts of a merged convolution sit side by
        # side — [w_a | w_b], [gamma_a | gamma_b], [beta_a | beta_b] — so the stacked launch writes ONE contiguous gradient.
        group_of = {}
        for op in self.ops:
            if op.kind == "conv" and isinstance(op.mod, MergedConv):
                parts = base_convs(op.mod)
                block = [m.conv.weight for m in parts] + [m.bn.weight for m in parts] + [m.bn.bias for m in parts]
                for t in block:
                    group_of[id(t)] = block
        self.params, seen = [], set()
        for p in model.parameters():
            if not p.requires_grad or id(p) in seen:
                continue
            for t in group_of.get(id(p), (p,)):
                assert t.requires_grad and id(t) not in seen
                seen.add(id(t))
                self.params.append(t)
        n = sum(p.numel() for p in self.params)
        self.arena = torch.zeros(n, dtype=torch.float32, device=device)
        self.gview = {}
        o = 0
        for p in self.params:
            self.gview[id(p)] = self.arena[o:o + p.numel()].view(p.shape)
            o += p.numel()
        self._build_buckets()
        self.on_bucket = None                 # TrainStep: callable(k, main_stream, side_stream) -> starts the bucket's all-reduce
        self.bn_mods = [m.bn for op in convs for m in base_convs(op.mod)]
        self.stem_scratch = None
        # wgrad scratch of the prediction convs [reg 0-3 | obj 4] / [cls] (zero between uses: sy_pred_grad_fold clears it)
        self.pred_scratch = torch.zeros((2, max(8, self.nc), int(256 * head.width)), dtype=torch.float32, device=device) \
            if head is not None else None
        self.pred_ws = torch.empty(int(ops._lib.lib().sy_pred_grad_fold_workspace_floats(self.nc)), dtype=torch.float32,
                                   device=device) if head is not None else None

    def _bind_scratch(self):
        """Split-K workspace and raw-gradient ring: shared with the module's other plans (multi-scale training keeps several
        plans alive; these two are pure scratch, 256 MB + RING x the largest raw gradient) or plan-owned without a pool."""
        esz = torch.empty(0, dtype=self.tdtype).element_size()
        ring_bytes = -(-self.max_raw * esz // 256) * 256
        # one tensor PER SLOT: the kernels address an operand through a 32-bit buffer extent measured to the end of its
        # allocation, so a single flat ring (RING x ~300 MB for l at 8 pairs) must not grow past 2 GiB
        if self.pool is not None:
            self.wgrad_ws = self.pool.shared_scratch("wgrad_ws", self.WGRAD_WS_BYTES, self.device)
            slots = [self.pool.shared_scratch("dyraw_ring_%d" % i, ring_bytes, self.device) for i in range(max(self.RING, 0))]
            self._scratch_gen = self.pool.scratch_gen
        else:
            self.wgrad_ws = torch.empty(self.WGRAD_WS_BYTES, dtype=torch.uint8, device=self.device)
            slots = [torch.empty(ring_bytes, dtype=torch.uint8, device=self.device) for _ in range(max(self.RING, 0))]
            self._scratch_gen = 0
        self.dyraw_ring = [t[:self.max_raw * esz].view(self.tdtype) for t in slots]
        # RING == 0: no reuse at all — every layer's raw gradient gets a buffer of its own (allocated at its first use, ~4 GB for l at
        # 8 pairs of the GPU's 288), so no chain ever waits for a weight gradient and no ring-slot event exists
        self.dyraw_own = []
        self.wgrad_ws_by = {WGRAD_STREAMS[0]: self.wgrad_ws}       # WGRAD_STREAMS[0] == 1: the prediction convs' wgrads run there too
        for k in WGRAD_STREAMS[1:]:                              # one split-K workspace per weight-gradient stream
            self.wgrad_ws_by[k] = (self.pool.shared_scratch("wgrad_ws%d" % k, self.WGRAD_WS_BYTES, self.device)
                                   if self.pool is not None else torch.empty(self.WGRAD_WS_BYTES, dtype=torch.uint8, device=self.device))
        if self.pool is not None:
            self._scratch_gen = self.pool.scratch_gen

    def release(self):
        """Dropped from the plan cache (LRU; never while `pending`; calling it twice is harmless): free the recorded tapes (they
        hold raw pointers into buffers that go back to the allocator with this object) and break the plan <-> gradient-space reference cycle so that the buffers are freed NOW,
        not at some later cyclic-GC pass (measured: without this, cycling through five sizes doubled the allocated memory)."""
        self.programs.clear()
        self.dyraw_own = []
        self.grads.py = None
        self.grads.mirror.clear()
        self.on_bucket = None

    # ------------------------------------------------------------------------------------------------
    def forward(self, x):
        """x [B,6,H,W] float on device -> raw [B, A, 5+nc] fp32 (plan-owned).  parts == "backbone": returns the fused
        feature views instead; parts == "head": x = the three fused feature tensors."""
        if self.parts != "head":
            x = x.float().contiguous()
        if self.pool is not None and self._scratch_gen != self.pool.scratch_gen:
            self._bind_scratch()                                 # a larger plan re-allocated the shared scratch
            self.programs.clear()
        sig = tuple(p.data_ptr() for p in self.params)
        if sig != self._param_sig or not self.cache.valid():     # a parameter was re-allocated (.to(), load with assign)
            self._param_sig = sig
            self.programs.clear()                                # the tapes hold raw pointers
            if not self.cache.valid():
                self.cache = StagedWeights(self.ops, self.dtype, self.device)
        self.cache.refresh()                                     # this step's weights -> MFMA operand layouts (one launch)
        ops.zero(self.stat_arena)
        if self.parts == "head":                                 # x = the three fused FPN features (NCHW-shaped tensors)
            for v, t in zip(self.fused, x):
                assert tuple(t.shape) == (v.N, v.C, v.H, v.W)
                v.set_nchw(t.detach())
        elif isinstance(x, FramePairsU8):                        # uint8 HWC frames: mirror / letterbox / resize / pack in one launch
            x.pack_focus(self.f0_cur, self.f0_sup)
        else:
            ops.focus_pack(x, 0, self.f0_cur)
            ops.focus_pack(x, 3, self.f0_sup)
        self._run("fwd", self._forward_ops)
        if self.run_table is None or not self.run_table.valid():
            mods = {}
            for op in self.ops:                                      # plan order == the reference's call order
                if op.kind == "conv":
                    # (sy_bn_finalize folds more than 64 replica rows — the exact mode's one row per workgroup — into the first 32
                    #  in place: csrc/train_ops.hip kFoldAbove / kFoldTo; the running-statistics launch reads those)
                    c0, ctot, SC = 0, op.y.C, (32 if op.stat_copies > 64 else op.stat_copies)
                    for m in base_convs(op.mod):                     # a stacked launch: each module's channel slice of the arrays
                        mods.setdefault(id(m.bn), (m.bn, []))[1].append((op.stat[0][c0:], op.stat[1][c0:], op.y.pixels, SC, ctot))
                        c0 += m.bn.num_features
            # num_batches_tracked: +1 per BN call (shared backbone / neck / jian BNs are called twice — trap T2) inside the same
            # launch (sy_bn_running_entry::num_batches_tracked += calls; round 6: was a torch._foreach_add_ per step)
            self.run_table = ops.BnRunningTable(list(mods.values()), self.device, count_batches=True)
        self.run_table.run()
        return self.raw if self.head is not None else self.fused

    def _forward_ops(self):
        """The op loop in launch order.  The per-frame network runs ONCE over both frames (2B images per launch, one
        statistics segment per frame — the reference's two backbone passes, dfp_pafpn.py:120-165); after the DFP
        fusion the three head levels fan out over the two streams."""
        nf = self.n_frame_ops
        if FWD_SPLIT_FRAMES and nf:
            # the two frames as two independent chains, current frame on the main stream, support frame on the side stream,
            # issued alternately (half-size launches, but one chain's tails and BatchNorm passes fill the other's gaps)
            early = self._early_levels() if (HEAD_EARLY and self.head is not None and self.side2 is not None) else {}
            self._mark("fork")
            for i in range(nf):
                self._forward_op(self.ops[i])
                self._mark("side_nw")
                self._forward_op(self.ops[nf + i])
                self._mark("main", None)
                for lvl_ops in early.get(i, ()):
                    # PAN output k of BOTH frames is complete here: its DFP fusion convs and head level are a third chain (stream
                    # 2) beside the rest of the two frame chains (the stride-8 level — the largest head level — beside the whole
                    # bottom-up path, 18 BaseConvs per frame), instead of three small chains behind the join (round 5: the launch
                    # timeline had <= 1.2 launches resident for the 1.5 ms around the head and the loss)
                    self._mark("dep", (0, 2))
                    self._mark("dep", (1, 2))
                    self._mark("cur", 2)
                    for op in lvl_ops:
                        self._forward_op(op)
                    self._mark("cur", 0)
            self._mark("join")
            if early:
                done = {id(op) for v in early.values() for lvl_ops in v for op in lvl_ops}
                for op in self.ops[2 * nf:]:                      # the level whose PAN output is the frame network's last op: main stream
                    if id(op) not in done:
                        self._forward_op(op)
                self._mark("dep", (2, 0))
                return
        for i in range(nf if not (FWD_SPLIT_FRAMES and nf) else 0):
            a, b2 = self.ops[i], self.ops[nf + i]
            if a.kind == "conv":
                # CSPLayer: conv2(x) only meets the bottleneck chain conv1(x) -> m(...) again in conv3(cat[...]) —
                # it runs on the side stream (idle during forward) beside the chain and joins before conv3
                role = _csp_role(a.tag) if CSP_FORK else None
                if role == "conv2":
                    self._mark("side")
                    self._forward_pair(a)
                    self._mark("main", None)
                    continue
                if role == "conv3":
                    self._mark("join")
                self._forward_pair(a)
            else:
                self._forward_op(a)
                self._forward_op(b2)
        for op in self.ops[2 * nf:self.n_head_start]:
            self._forward_op(op)
        self._mark("fork")
        for op in self.ops[self.n_head_start:]:
            if op.level == 0:
                self._forward_op(op)
        self._mark("side_nw")
        self._side_region = True                                 # levels 1-2 already run on the side stream
        for op in self.ops[self.n_head_start:]:
            if op.level != 0:
                self._forward_op(op)
        self._side_region = False
        self._mark("main", None)
        self._mark("join")

    def _early_levels(self):
        """{index i of the per-frame op list: [ops of a level that can start behind op i of both frames]}: level k = its two DFP
        fusion convs (engine.build_fuse_net order: level-major, current / support) + the head ops of that level; it can start when
        the last per-frame op that writes the buffer its fusion reads has been issued.  The level that is ready only behind the
        last frame op stays where it was (after the join, on the main stream)."""
        nf = self.n_frame_ops
        fuse = self.ops[2 * nf:self.n_head_start]
        head_ops = self.ops[self.n_head_start:]
        nlev = len(self.preds)
        if len(fuse) != 2 * nlev:
            return {}
        out = {}
        for k in range(nlev):
            src = fuse[2 * k].x.buf.data_ptr()                   # (the tensor's own start: buffers may share one pooled storage)
            writers = [i for i in range(nf) if self.ops[i].kind == "conv" and self.ops[i].y.buf.data_ptr() == src]
            if not writers or max(writers) >= nf - 1:
                continue
            out.setdefault(max(writers), []).append([fuse[2 * k], fuse[2 * k + 1]] + [op for op in head_ops if op.level == k])
        return out

    # ---- parameters of a (possibly stacked) BaseConv op -------------------------------------------------------------------
    def _bn_params(self, op):
        """(gamma, beta, eps, momentum) of the op's BatchNorm; a MergedConv's are the stacked fp32 copies StagedWeights keeps."""
        parts = base_convs(op.mod)
        bn = parts[0].bn
        mom = bn.momentum if bn.momentum is not None else 0.1
        if len(parts) == 1:
            return bn.weight, bn.bias, bn.eps, mom
        assert all(m.bn.eps == bn.eps and m.bn.momentum == bn.momentum for m in parts)
        g, b_ = self.cache.bn[id(op.mod)]
        return g, b_, bn.eps, mom

    def _bn_bwd(self, op, y, da, aff, gamma, bsum, dy, dgamma, dbeta, nseg=1, dres=None, acc=False, atomic=False):
        """BatchNorm.SiLU backward of one launch unit: the reduce pass (unless the data gradient that produced `da` carried it:
        op.bnr_by), then the apply pass."""
        fused = getattr(op, "bnr_by", None) is not None
        if not fused:
            ops.bn_silu_bwd_reduce(y, da, *aff, bsum, nseg=nseg)
        ops.bn_silu_bwd_apply(y, da, *aff, gamma, bsum, dy, dgamma, dbeta, nseg=nseg, dres=dres, dres_accumulate=acc,
                              atomic_param_grads=atomic, raw_moment=fused)

    def _bnr_producer(self, op, dx, acc, t):
        """The 1x1 ConvOp whose BatchNorm-backward reduce this op's data gradient can carry (SY_EPI_BNR), or None.  A static
        property of the plan (the tapes record the decision): op reads ONLY that layer's activation, first write of its
        gradient, 16-bit compute, a tile with the staged 128-channel epilogue, the raw output laid out like the gradient view."""
        pre = op.pre_op
        if (not BNR_FUSION or pre is None or acc or self.dtype == ops.DT_F32 or t not in BNR_TILES or op.k != 3 or op.stride != 1
                or pre.res is not None or pre.yraw is None or pre.yraw.ld != dx.ld or pre.yraw.bs != dx.bs or pre.y.C % 8 or dx.ld % 8
                or (dx.ptr() | pre.yraw.ptr()) % 16 or dx.bs % 8):     # sy_conv2d's own precondition of the staged epilogue
            return None
        return pre

    def _bn_grads(self, op):
        """(dgamma, dbeta) arena views starting at the op's first part (the parts' slots are adjacent: see the arena order)."""
        bn = base_convs(op.mod)[0].bn
        return self.gview[id(bn.weight)], self.gview[id(bn.bias)]

    def _forward_pair(self, a):
        """BaseConv of the shared per-frame network on both frames: conv (+ per-frame statistics), finalize, BN+SiLU."""
        gamma, beta, eps, mom = self._bn_params(a)
        x2, raw2, y2 = a.x.pair(), a.yraw.pair(), a.y.pair()
        u_sum, u_sq, _, (scale, shift, mean, invstd) = a.unit
        t = a._tiles.get("fwd_stats2")
        if t is None:
            t = ops.tuned_tile(ops.CONV_FWD, x2.dtype, x2.N, x2.H, x2.W, x2.C, y2.C, a.k, a.stride, self.device,
                               with_stats=True)
            a._tiles["fwd_stats2"] = t
        ops.conv2d(x2, self.cache.conv_weight(a.mod), raw2, a.k, a.stride, stats=(u_sum, u_sq), tile=t,
                   wfrag=self.cache.conv_weight_frag(a.mod) if t >= ops.TILE_WR else None, segments=2)
        if self.fused_finalize:
            ops.bn_finalize_apply(u_sum, u_sq, a.y.pixels, gamma, beta, eps, scale, shift, mean, invstd, raw2, y2,
                                  res=None if a.res is None else a.res.pair(), nseg=2)
        else:
            ops.bn_finalize(u_sum, u_sq, a.y.pixels, gamma, beta, eps, mom, None, None, scale, shift, mean, invstd,
                            nseg=2)
            ops.bn_silu_apply(raw2, scale, shift, y2, res=None if a.res is None else a.res.pair(), nseg=2)

    # ---- launch programs ------------------------------------------------------------------------------------
    # Step 1 runs the Python wrappers directly (kernel variants get tuned).  Step 2 runs them again under
    # _lib.record(): every C-ABI call lands on a tape together with the torch ops in between ("py" entries) and
    # the stream marks emitted above.  From step 3 on the tape IS the step: the interpreter below re-issues the
    # recorded calls on explicit hipStream_t handles — main stream / side stream as the marks say — so the
    # per-launch host cost is one ctypes call, and no Python-side view / descriptor / bookkeeping code runs.
    def _mark(self, kind, arg=None):
        if self._rec is not None:
            self._rec.mark(kind, arg)

    def _py(self, fn):
        """A torch-op snippet inside a pass: runs now, and is replayed from the tape later."""
        fn()
        if self._rec is not None:
            self._rec.snippet(fn)

    def _run(self, name, body, key=None):
        from . import _lib
        if self.force_serial or not self.tuned or self.device.type != "cuda" and not self.TAPE_ON_CPU:
            self._rec = None
            body()                                               # direct (tuning step, profile(), CPU test runs)
            return
        prog = self.programs.get(name)
        if prog is None or prog[0] != key:
            tape = _lib.NativeTape()                             # records inside the library, at the kernel-launch level
            with tape:
                self._rec = tape
                try:
                    body()
                finally:
                    self._rec = None
            self.programs[name] = (key, tape)
            return
        self._interpret(prog[1])

    def _interpret(self, tape):
        """Replay a recorded pass: ONE library call walks the launches, stream switches and event pairs (csrc/tape.hip);
        Python is re-entered only for the recorded torch snippets and, in data-parallel runs, at the bucket marks."""
        side, side2 = self.side, self.side2
        extra = [self.side_w.get(k) for k in (3, 4, 5)]
        if self.device.type == "cuda":
            main = torch.cuda.current_stream(self.device)
            main_h = C.c_void_p(main.cuda_stream)
            side_h = C.c_void_p(side.cuda_stream) if side is not None else None
            more = [C.c_void_p(st_.cuda_stream) if st_ is not None else None for st_ in [side2] + extra]
            while more and more[-1] is None:
                more.pop()
            if torch.cuda.is_current_stream_capturing():
                # hipGraph capture (TrainStep(graph=True)): forks across more than two streams segfault inside capture_end on
                # this ROCm build — chain 2 folds back onto the main stream there (sy_tape_replay_n's fallback)
                more = []
        else:
            main, main_h, side_h, more = None, C.c_void_p(0), None, []
        chains = [main, side, side2] + extra

        def snippet(fn, k):
            if k and chains[k] is not None:
                with torch.cuda.stream(chains[k]):
                    fn()
            else:
                fn()
        on_bucket = None
        if self.on_bucket is not None:
            on_bucket = lambda k: self.on_bucket(k, main, [s_ for s_ in [side, side2] + extra if s_ is not None])      # noqa: E731
        tape.replay(main_h, side_h, snippet, on_bucket, more=more)

    def _forward_op(self, op):
        nch = 5 + self.nc
        k = op.kind
        if k == "conv":
            gamma, beta, eps, mom = self._bn_params(op)
            w = self.cache.conv_weight(op.mod)
            t = op.tile("fwd_stats")
            scale, shift, mean, invstd = op.aff
            ops.conv2d(op.x, w, op.yraw, op.k, op.stride, stats=op.stat, tile=t,
                       wfrag=self.cache.conv_weight_frag(op.mod) if t >= ops.TILE_WR else None)
            # running statistics: one batched launch at the end of the pass (the two frames' calls of a shared
            # module update them in call order there, whatever stream each frame ran on)
            if self.fused_finalize:
                ops.bn_finalize_apply(op.stat[0], op.stat[1], op.y.pixels, gamma, beta, eps, scale, shift, mean, invstd,
                                      op.yraw, op.y, res=op.res)
            else:
                ops.bn_finalize(op.stat[0], op.stat[1], op.y.pixels, gamma, beta, eps, mom,
                                None, None, scale, shift, mean, invstd)
                ops.bn_silu_apply(op.yraw, scale, shift, op.y, res=op.res)
        elif k == "resize":
            ops.resize_nearest(op.src, op.dst)
        elif k == "spp":
            ops.spp_pool(op.v, op.argmax)
        else:   # pred: raw logits (tal_head.py:174: cat[reg, obj, cls]), decoded later by the loss
            w_ro, b_ro, w_c, b_c = self.cache.pred(op)[:4]
            base = self.raw.data_ptr() + op.a0 * nch * 4
            ops.conv2d(op.reg_x, w_ro, None, 1, 1, None, b_ro, epilogue=EPI_LINEAR, y_f32=True, y_ptr=base,
                       y_ld=nch, y_bs=self.A * nch, cout=5)
            ops.conv2d(op.cls_x, w_c, None, 1, 1, None, b_c, epilogue=EPI_LINEAR, y_f32=True, y_ptr=base + 20,
                       y_ld=nch, y_bs=self.A * nch, cout=self.nc)

    # ------------------------------------------------------------------------------------------------
    def loss(self, labels, support):
        """SimOTA + Trend-Aware loss and d(total)/d(raw) in two HIP kernels, no host sync (sy_tal_loss).
        Returns (loss dict of 0-dim device tensors, d_raw [B, A, 5+nc])."""
        head = self.head
        if self.loss_ws is None or self.loss_ws.max_labels != labels.shape[1]:
            self.loss_ws = ops.TalLossWorkspace(self.B, self.A, 5 + self.nc, self.hw, head.strides, self.device,
                                                max_labels=labels.shape[1])
        # d(total)/d(raw) also lands in `dpad`, the operand layout of the prediction convs' gradients ([reg 4 | obj | 0 0 0 | cls nc] in
        # the compute dtype): backward() starts its tape without a repacking pass when it is handed this very d_raw (the fast path;
        # the drop-in path scales d_raw by the incoming gradient — GradScaler — and repacks)
        pad = self.dpad if self.nc <= 8 else None
        self._zero_gradient_arenas()
        losses, d_raw, _ = ops.tal_loss(self.raw, labels, support, self.nc, head.gamma, head.ignore_thr,
                                        head.ignore_value, head.use_l1, self.loss_ws, d_pad=pad)
        self._packed_for = (d_raw, d_raw._version) if pad is not None else None     # identity AND content: an in-place scaling repacks
        out = {"total_loss": losses[0], "iou_loss": losses[1], "l1_loss": losses[2], "conf_loss": losses[3],
               "cls_loss": losses[4], "num_fg": losses[5]}
        return out, d_raw

    # ------------------------------------------------------------------------------------------------
    def backward(self, d_raw, d_fused=None):
        """d_raw [B, A, 5+nc] fp32 -> parameter gradients accumulated into self.arena (zeroed here).
        parts == "backbone": d_raw is None and d_fused = gradients of the three fused features (NCHW-shaped tensors);
        parts == "head": afterwards self.fused_grads() holds the gradients of the fused inputs."""
        self._zero_gradient_arenas(wait=True)
        nc = self.nc
        pf = getattr(self, "_packed_for", None)
        if self.head is not None and not (pf is not None and pf[0] is d_raw and pf[1] == d_raw._version):
            # pack d_raw as [reg 4 | obj 1 | 0 0 0 | cls nc] in the compute dtype for the MFMA kernels (sy_tal_loss wrote it already
            # when d_raw is the loss's own, unscaled, tensor)
            self.dpad[..., 0:5] = d_raw[..., 0:5]
            self.dpad[..., 8:8 + nc] = d_raw[..., 5:]
        self._packed_for = None
        self._seed = d_fused
        self._run("bwd", lambda: self._backward_ops(d_raw), key=None if d_raw is None else d_raw.data_ptr())
        if not self.tuned:
            self.tuned = True                                        # kernels are tuned after the first full step
            ops.save_tuned()                                         # ... and the choices persisted (next plan / process)
        return self.arena

    def _zero_gradient_arenas(self, wait=False):
        """The flat gradient arena (l: 219 MB) and the BatchNorm-backward sums are cleared once per step.  On the GPU the two
        memsets are issued when the LOSS starts, on a stream that idles between the passes: the assignment kernel keeps eight
        workgroups busy for ~0.25 ms and nothing else can run beside it — the 60 us of memsets used to sit in front of the backward
        pass.  backward() (wait=True) orders its streams behind them, or clears the arenas itself when no loss call did (the
        sub-module plans, a caller that skips loss())."""
        st = self.side_w.get(3) or self.side2 or self.side
        if wait:
            if self._zeroed is None:
                ops.zero(self.arena)
                ops.zero(self.bwd_arena)
            elif self._zeroed is not True:
                torch.cuda.current_stream(self.device).wait_event(self._zeroed)
            self._zeroed = None
            return
        if self._zeroed is not None:
            return                                               # loss() called twice before a backward: already clear
        if st is None or self.device.type != "cuda" or torch.cuda.is_current_stream_capturing():
            ops.zero(self.arena)
            ops.zero(self.bwd_arena)
            self._zeroed = True
            return
        st.wait_stream(torch.cuda.current_stream(self.device))   # whoever read the gradients last (optimizer, all-reduce) is done
        with torch.cuda.stream(st):
            ops.zero(self.arena)
            ops.zero(self.bwd_arena)
            if self._zero_ev is None:
                self._zero_ev = torch.cuda.Event()
            self._zero_ev.record()
        self._zeroed = self._zero_ev

    def fused_grads(self):
        return tuple(self.grads.view(f).nchw() for f in self.fused)

    # ---- gradient buckets (data-parallel overlap) -----------------------------------------------------------
    # The arena is in parameter order = forward order, the backward walk runs it back to front, so contiguous
    # arena ranges complete one after the other.  _build_buckets cuts it into ranges of >= BUCKET_BYTES and notes,
    # for each, the position of the backward walk after which every gradient in the range is final; the walk emits
    # a "bucket" mark there and TrainStep starts that range's RCCL all-reduce while the rest of backward still runs.
    BUCKET_BYTES = 32 << 20
    # The bucket at the FRONT of the arena (stem, dark2, dark3: the parameters whose gradients the backward walk finishes last)
    # is the one whose all-reduce nothing can hide — everything behind it overlaps with the rest of backward — so it is kept
    # small: a ring all-reduce of a few MB over xGMI is latency-bound (tens of microseconds), one of 32 MB is not.
    FIRST_BUCKET_BYTES = 4 << 20

    def _backward_sequence(self):
        nf = self.n_frame_ops
        return list(reversed(self.ops[2 * nf:])) + [self.ops[i] for i in reversed(range(nf))]

    def _build_buckets(self):
        last = {}
        for pos, op in enumerate(self._backward_sequence()):
            if op.kind == "conv":
                prm = tuple(t for m in base_convs(op.mod) for t in (m.conv.weight, m.bn.weight, m.bn.bias))
            elif op.kind == "pred":
                prm = tuple(t for m in (op.reg_mod, op.obj_mod, op.cls_mod) for t in (m.weight, m.bias))
            else:
                prm = ()
            for t in prm:
                last[id(t)] = pos
        self.buckets, lo, o, ready = [], 0, 0, -1
        for p in self.params:
            o += p.numel()
            ready = max(ready, last.get(id(p), -1))
            if (o - lo) * 4 >= (self.BUCKET_BYTES if self.buckets else min(self.FIRST_BUCKET_BYTES, self.BUCKET_BYTES)):
                self.buckets.append((lo, o, ready))
                lo, ready = o, -1
        if o > lo:
            self.buckets.append((lo, o, ready))
        self.bucket_at = {}
        for k, (_, _, r) in enumerate(self.buckets):
            self.bucket_at.setdefault(r, []).append(k)

    def _bucket_marks(self, pos):
        for k in self.bucket_at.get(pos, ()):
            self._mark("bucket", k)

    def _backward_ops(self, d_raw):
        G = self.grads
        G.reset()
        for op in self.ops:                                      # fused-reduce marks are per pass (set by the consumer's data gradient)
            if op.kind == "conv":
                op.bnr_by = None
        self.ring_i = 0
        self._split_j = 0                                        # layers issued by _conv_pair_backward_split in this pass
        nf = self.n_frame_ops
        if self.head is None:                                    # backbone alone: the feature gradients come from the caller
            for i, f in enumerate(self.fused):
                gv, acc = G.target(f)
                assert not acc
                # read THIS call's seed at replay time: the recorded snippet must not bind the recording call's tensors
                self._py(lambda gv=gv, i=i: gv.set_nchw(self._seed[i]))
        self._bucket_marks(-1)                                   # ranges no kernel writes (unused parameters)
        split = self.bwd_split and nf > 0
        self._chain, self._wg_flip = 0, 0
        head_chain = HEAD_BWD_CHAINS and self.head is not None and self.side2 is not None
        n_head = len(self.ops) - self.n_head_start               # the backward walk starts with the head's ops, last level first
        if head_chain:
            self._mark("dep", (0, 2))                            # d_raw / dpad are ready on the main stream
        for pos, a in enumerate(self._backward_sequence()):      # head, DFP fusion, then the per-frame network
            if head_chain and pos == n_head:
                self._chain = 0
                self._mark("cur", 0)
                self._mark("dep", (2, 0))                        # the fusion backward reads every level's feature gradient
            if split and pos == len(self.ops) - 2 * nf:
                self._mark("dep", (0, 2))                        # the support-frame chain starts behind the head / fusion backward
            if pos < len(self.ops) - 2 * nf:
                if head_chain and pos < n_head:                  # levels 1-2 as a chain of their own beside level 0
                    k = 2 if a.level != 0 else 0
                    if k != self._chain:
                        self._chain = k
                        self._mark("cur", k)
                if a.kind == "pred":
                    self._pred_backward(a, d_raw)
                else:
                    self._conv_backward(a)
            else:                                                # layer i of both frames together
                b2 = self.ops[nf + (len(self.ops) - 2 * nf) + nf - 1 - pos]
                if a.kind == "conv":
                    if split:
                        self._conv_pair_backward_split(a, b2)
                    else:
                        self._conv_pair_backward(a, b2)
                else:
                    for op in (b2, a):
                        if split:
                            self._mark("cur", 2 if op is b2 else 0)
                        if op.kind == "resize":
                            dsrc, acc = G.target(op.src)
                            ops.resize_nearest_bwd(G.view(op.dst), dsrc, acc)
                        elif op.kind == "spp":
                            ops.spp_pool_bwd(G.view(op.v), op.argmax)
            self._bucket_marks(pos)
        if split:
            self._mark("cur", 0)
            self._mark("dep", (2, 0))
        for k in WGRAD_STREAMS:                                  # "join" covers stream 1; the further weight-gradient streams join
            if k != 1:                                           # here: whoever reads the arena next does so on the main stream
                self._mark("dep", (k, 0))
        self._mark("join")

    # ---- weight gradients off the critical path -------------------------------------------------------------
    # Nothing downstream in the backward pass reads a weight gradient, so every wgrad (+ its fold) runs on the side
    # stream, ordered after the main-stream kernels that produced its raw gradient, and the main stream goes
    # straight on to the data gradient of the same layer.
    def _scratch(self, numel):
        """Next raw-gradient slot of the ring.  A slot's raw gradient has up to three readers: the weight gradient (on a
        weight-gradient stream) and the data gradient(s) on the chain(s) that produced it; each marks "slot_done" on its own
        stream behind its launch, and the chain that overwrites the slot next waits ("acquire_cur") for every one of those
        events that lives on ANOTHER stream — several chains share the ring (head levels 1-2 / the support frame on chain 2),
        so the next writer is not necessarily on the stream of the previous data gradient."""
        if self.RING <= 0:
            self._slot = None
            return self._own_buffer(numel)
        self.ring_i = (self.ring_i + 1) % self.RING
        self._slot = self.ring_i
        self._mark("acquire_cur", self._slot)
        return self.dyraw_ring[self._slot][:numel]

    def _own_buffer(self, numel):
        """RING == 0: the raw-gradient buffer of the ring_i-th convolution of the backward walk (the walk is the same every pass)."""
        i = self.ring_i
        self.ring_i += 1
        if i == len(self.dyraw_own):
            self.dyraw_own.append(torch.empty(numel, dtype=self.tdtype, device=self.device))
        assert self.dyraw_own[i].numel() >= numel
        return self.dyraw_own[i][:numel]

    def _wgrad_stream(self, fixed=None):
        """Stream the next weight gradient goes to: round robin over WGRAD_STREAMS (1, 3, ...)."""
        if fixed is not None:
            return fixed
        self._wg_flip = (self._wg_flip + 1) % len(WGRAD_STREAMS)
        return WGRAD_STREAMS[self._wg_flip]

    def _on_side(self, fn, slot=None, chains=None, stream=None):
        """fn's launches (a weight gradient + fold) go to a weight-gradient stream, after everything issued so far on the
        chain(s) that produced its raw gradient; the ring slot is marked free behind them."""
        chains = (self._chain,) if chains is None else chains
        w = self._wgrad_stream(stream)
        for k in chains:
            self._mark("dep", (k, w))
        self._mark("cur", w)
        self._wg_stream = w
        fn()
        self._wg_stream = 1
        if slot is not None:
            self._mark("slot_done", slot)
        self._mark("cur", self._chain)

    def _ws(self):
        return self.wgrad_ws_by[self._wg_stream]

    def _pred_backward(self, op, d_raw):
        G = self.grads
        hwk = op.reg_x.H * op.reg_x.W
        B, A, nc = self.B, self.A, self.nc
        w_ro_t, w_c_t = self.cache.pred(op)[4:6]
        d_ro = View(self.dpad, B, op.reg_x.H, op.reg_x.W, 8, 16, op.a0 * 16, bs=A * 16)
        d_c = View(self.dpad, B, op.reg_x.H, op.reg_x.W, nc, 16, op.a0 * 16 + 8, bs=A * 16)
        g_r, acc_r = G.target(op.reg_x)
        ops.conv2d(d_ro, w_ro_t, g_r, 1, 1, mode=CONV_DGRAD, accumulate=acc_r)
        g_c, acc_c = G.target(op.cls_x)
        ops.conv2d(d_c, w_c_t, g_c, 1, 1, mode=CONV_DGRAD, accumulate=acc_c)
        cin = op.reg_x.C

        sc = self.pred_scratch
        g_reg, g_obj, g_cls = (self.gview[id(m.weight)] for m in (op.reg_mod, op.obj_mod, op.cls_mod))
        gb_r, gb_o, gb_c = (self.gview[id(m.bias)] for m in (op.reg_mod, op.obj_mod, op.cls_mod))

        def wg():
            # two MFMA weight-gradient launches into the (zero) scratch, then one fold: weight gradients += scratch rows,
            # bias gradients += column sums of this level's d_raw rows — all C-ABI launches, nothing for the host to do
            ops.conv2d_wgrad(op.reg_x, d_ro, sc[0], 1, 1, workspace=self._ws())
            ops.conv2d_wgrad(op.cls_x, d_c, sc[1], 1, 1, workspace=self._ws())
            ops.pred_grad_fold(d_raw, op.a0, hwk, nc, sc, cin, g_reg, g_obj, g_cls, gb_r, gb_o, gb_c, self.pred_ws)
        self._on_side(wg, stream=1)                                  # one scratch for the three levels: always stream 1

    def _bn_backward(self, op, dyraw):
        """Residual fan-in + BatchNorm/SiLU backward of one BaseConv call: fills `dyraw` (grad of the raw conv
        output) and accumulates dgamma / dbeta."""
        G = self.grads
        gamma = self._bn_params(op)[0]
        dgamma, dbeta = self._bn_grads(op)
        dY = G.view(op.y)
        dres, acc = (None, False) if op.res is None else G.target(op.res)    # y = silu(bn(conv)) + res: dres (+)= dY
        scale, shift, mean, invstd = op.aff
        self._bn_bwd(op, op.yraw, dY, (scale, shift, mean, invstd), gamma, op.bsum, dyraw, dgamma, dbeta, dres=dres, acc=acc)

    def _wgrad(self, op, x, dyraw):
        w = base_convs(op.mod)[0].conv.weight                        # stacked parts: their arena slots follow this one
        key = "wgrad%d" % x.N
        wt = op._tiles.get(key)
        if wt is None:
            wt = ops.tuned_wgrad(x.dtype, x.N, x.H, x.W, x.C, dyraw.H, dyraw.W, dyraw.C, op.k, op.stride,
                                 self.device, self.wgrad_ws)
            wt = scheduled_wgrad(wt, x.C)
            op._tiles[key] = wt
        if w.shape[1] == x.C:
            ops.conv2d_wgrad(x, dyraw, self.gview[id(w)], op.k, op.stride, oihw=True, workspace=self._ws(),
                             tile=wt[0], target_blocks=wt[1])
        else:                                                        # Focus stem: 12 real + 4 zero-padded channels
            if self.stem_scratch is None:
                self.stem_scratch = torch.zeros((w.shape[0], x.C, op.k, op.k), dtype=torch.float32, device=self.device)
            sc, gw = self.stem_scratch, self.gview[id(w)]                # scratch zero between uses (cleared by the fold)
            ops.conv2d_wgrad(x, dyraw, sc, op.k, op.stride, oihw=True, workspace=self._ws(),
                             tile=wt[0], target_blocks=wt[1])
            kk = op.k * op.k
            ops.rows_add_f32(gw, sc, w.shape[0], w.shape[1] * kk, w.shape[1] * kk, x.C * kk, zero_src=True)

    def _conv_pair_backward(self, a, b2):
        """Layer i of the current-frame and support-frame networks together: per-frame BN backward (separate
        statistics), then ONE wgrad and ONE dgrad launch over the 2B images (the weights are shared)."""
        G = self.grads
        N, H, W, C = a.y.N, a.y.H, a.y.W, a.y.C
        full = self._scratch(2 * N * H * W * C).view(2 * N, H, W, C)
        slot = self._slot
        dy2 = View(full, 2 * N, H, W, C)
        gamma = self._bn_params(a)[0]
        dgamma, dbeta = self._bn_grads(a)
        dYa, dYb = G.view(a.y), G.view(b2.y)
        assert dYa.root[0] is dYb.root[0]
        dY2 = dYa.pair()
        dres2, acca = None, False
        if a.res is not None:                                        # y = silu(bn(conv)) + res, both frames at once
            dra, acca = G.target(a.res)
            drb, accb = G.target(b2.res)
            assert acca == accb and dra.root[0] is drb.root[0]
            dres2 = dra.pair()                                       # written by the BN backward apply pass below
        _, _, u_bsum, (scale, shift, mean, invstd) = a.unit
        raw2 = a.yraw.pair()
        self._bn_bwd(a, raw2, dY2, (scale, shift, mean, invstd), gamma, u_bsum, dy2, dgamma, dbeta, nseg=2, dres=dres2, acc=acca)
        self._on_side(lambda: self._wgrad(a, a.x.pair(), dy2), slot)
        if a.need_dx:
            dxa, acca = G.target(a.x)
            dxb, accb = G.target(b2.x)
            assert acca == accb and dxa.root[0] is dxb.root[0]
            t = a._tiles.get("dgrad2")
            if t is None:
                t = ops.tuned_tile(CONV_DGRAD, dy2.dtype, 2 * N, H, W, C, a.x.C, a.k, a.stride, self.device)
                a._tiles["dgrad2"] = t
            ops.conv2d(dy2, self.cache.conv_weight(a.mod, transpose=True), dxa.pair(), a.k, a.stride,
                       mode=CONV_DGRAD, accumulate=acca, tile=t,
                       wfrag=self.cache.conv_weight_frag(a.mod, transpose=True) if t >= ops.TILE_WR else None)
            if slot is not None:
                self._mark("slot_done", slot)                        # ... and this chain's reader of the slot (see _scratch)

    def _conv_pair_backward_split(self, a, b2):
        """Layer i of the two frames as two chains: BatchNorm backward + data gradient of the current frame on stream 0, of the
        support frame on stream 2 (separate half-size launches: one chain's HBM-bound passes run beside the other's MFMA-bound
        data gradient), the weight gradient ONCE over both frames on stream 1 behind both chains' raw gradients."""
        G = self.grads
        N, H, W, C = a.y.N, a.y.H, a.y.W, a.y.C
        if self.RING <= 0:
            slot, full = None, self._own_buffer(2 * N * H * W * C).view(2 * N, H, W, C)
        else:
            self.ring_i = (self.ring_i + 1) % self.RING
            slot = self.ring_i
            full = self.dyraw_ring[slot][:2 * N * H * W * C].view(2 * N, H, W, C)
        dy2 = View(full, 2 * N, H, W, C)
        dys = (View(full[:N], N, H, W, C), View(full[N:], N, H, W, C))
        gamma = self._bn_params(a)[0]
        dgamma, dbeta = self._bn_grads(a)
        j, K = self._split_j, max(1, min(SLOT_BATCH, self.RING - 2))
        self._split_j += 1
        # the first RING layers of the split phase reuse slots of the head / fusion ops (other streams): per-layer waits; afterwards a
        # chain waits once per K layers, for the newest slot of each weight-gradient stream among the K it is about to overwrite (the
        # weight gradients of consecutive layers alternate between the streams and retire in order on each)
        batched = K > 1 and j >= self.RING
        for k, (op, dyr) in enumerate(zip((a, b2), dys)):            # BatchNorm / SiLU backward, frame by frame
            self._mark("cur", 2 * k)
            if slot is None:
                pass                                                 # a buffer of its own: nobody to wait for
            elif not batched:
                self._mark("acquire_cur", slot)                      # the wgrad that last read this slot has retired
            elif (j - self.RING) % K == 0:
                for m in range(min(K, len(WGRAD_STREAMS))):
                    self._mark("acquire_cur", (slot + K - 1 - m) % self.RING)
            dres, acc = (None, False) if op.res is None else G.target(op.res)
            scale, shift, mean, invstd = op.aff
            self._bn_bwd(op, op.yraw, G.view(op.y), (scale, shift, mean, invstd), gamma, op.bsum, dyr, dgamma, dbeta,
                         dres=dres, acc=acc, atomic=True)
        self._chain = 0
        self._on_side(lambda: self._wgrad(a, a.x.pair(), dy2), slot, chains=(0, 2))
        if a.need_dx:
            t = a.tile("dgrad")
            for k, (op, dyr) in enumerate(zip((a, b2), dys)):
                self._mark("cur", 2 * k)
                dx, acc = G.target(op.x)
                pre = self._bnr_producer(op, dx, acc, t)
                if pre is not None:                                  # ... + the BatchNorm-backward reduce of the layer that made op.x
                    pre.bnr_by = op
                ops.conv2d(dyr, self.cache.conv_weight(a.mod, transpose=True), dx, a.k, a.stride, mode=CONV_DGRAD,
                           accumulate=acc, tile=t,
                           wfrag=self.cache.conv_weight_frag(a.mod, transpose=True) if t >= ops.TILE_WR else None,
                           bn_reduce=None if pre is None else (pre.yraw, pre.aff[0], pre.aff[1], pre.bsum))
                if CHAIN_SLOT_DONE and slot is not None:
                    self._mark("slot_done", slot)                    # this chain's reader of its half of the slot
        self._mark("cur", 0)

    def _conv_backward(self, op):
        G = self.grads
        C = op.y.C
        dyraw = View(self._scratch(op.y.pixels * C).view(op.y.N, op.y.H, op.y.W, C), op.y.N, op.y.H, op.y.W, C)
        slot = self._slot
        self._bn_backward(op, dyraw)
        self._on_side(lambda: self._wgrad(op, op.x, dyraw), slot)
        if op.need_dx:
            dx, acc = G.target(op.x)
            t = op.tile("dgrad")
            ops.conv2d(dyraw, self.cache.conv_weight(op.mod, transpose=True), dx, op.k, op.stride,
                       mode=CONV_DGRAD, accumulate=acc, tile=t,
                       wfrag=self.cache.conv_weight_frag(op.mod, transpose=True) if t >= ops.TILE_WR else None)
            if slot is not None:
                self._mark("slot_done", slot)                        # ... and this chain's reader of the slot (see _scratch)

    # ------------------------------------------------------------------------------------------------
    def profile(self, x, targets, iters=2, detail=False):
        """Per-op-kind kernel time (ms / step) with HIP events on the launch stream (bench.py roofline).
        detail=True: per (kind, shape) rows [(kind, shape, launches / step, ms / step, flops / step)] instead."""
        evs = []
        real = {n: getattr(ops, n) for n in ("conv2d", "conv2d_wgrad", "bn_finalize", "bn_silu_apply", "bn_finalize_apply",
                                             "bn_silu_bwd_reduce", "bn_silu_bwd_apply", "resize_nearest",
                                             "resize_nearest_bwd", "spp_pool", "spp_pool_bwd", "view_copy", "focus_pack")}

        def wrap(name, fn):
            def inner(*a, **k):
                kind = name
                if name == "conv2d":
                    kind = "dgrad" if k.get("mode", 0) == CONV_DGRAD else "conv"
                elif name == "conv2d_wgrad":
                    kind = "wgrad"
                desc, fl = "", 0.0
                if detail and name in ("conv2d", "conv2d_wgrad"):
                    xi, yo = a[0], (a[2] if name == "conv2d" else a[1])
                    if xi is None or yo is None:            # prediction convs write / read a raw fp32 pointer
                        xi = yo = (xi or yo)
                        desc, kind = "pred N%d %dx%d c%d" % (xi.N, xi.H, xi.W, xi.C), kind + "(pred)"
                    if kind == "dgrad":
                        xi, yo = yo, xi
                    if not desc:
                        desc = "N%d %dx%d %d->%d k%d s%d" % (xi.N, yo.H, yo.W, xi.C, yo.C, a[3], a[4])
                        fl = 2.0 * xi.C * yo.C * a[3] * a[3] * yo.N * yo.H * yo.W
                elif detail and isinstance(a[0], View):
                    desc = "N%d %dx%d c%d" % (a[0].N, a[0].H, a[0].W, a[0].C)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = fn(*a, **k)
                e.record()
                evs.append((kind, s, e, desc, fl))
                return r
            return inner
        self.force_serial = True                                 # per-kernel durations: one stream, no overlap
        try:
            for n, fn in real.items():
                setattr(ops, n, wrap(n, fn))
            for _ in range(iters):
                self.forward(x)
                _, d_raw = self.loss(targets[0], targets[1])
                self.backward(d_raw)
        finally:
            for n, fn in real.items():
                setattr(ops, n, fn)
            self.force_serial = False
        torch.cuda.synchronize()
        tot, rows = {}, {}
        for kind, s, e, desc, fl in evs:
            ms = s.elapsed_time(e)
            tot[kind] = tot.get(kind, 0.0) + ms
            r = rows.setdefault((kind, desc), [0, 0.0, 0.0])
            r[0] += 1; r[1] += ms; r[2] += fl
        if detail:
            return [(k[0], k[1], r[0] / iters, r[1] / iters, r[2] / iters) for k, r in rows.items()]
        return {k: v / iters for k, v in tot.items()}


def get_train_plan(model, x):
    dt = compute_dtype_for(model, x)
    B, _, H, W = x.shape
    key = ("train", B, H, W, dt, str(x.device))
    return model._plans.get(key, lambda: TrainPlan(model, B, H, W, dt, x.device, pool=model._plans))


class _Pin:
    """One count of a plan's `pending`, given back exactly once — by release() or when the holder (an autograd node's ctx) dies."""

    def __init__(self, plan):
        self.plan = plan

    def release(self):
        if self.plan is not None:
            self.plan.pending = max(0, self.plan.pending - 1)
            self.plan = None

    def __del__(self):
        self.release()


class _PlanFunction(torch.autograd.Function):
    """total_loss = TAL(plan.forward(x)); the forward pass also produces d(total)/d(raw) (closed form in
    sy_tal_loss), so backward only scales it by the incoming gradient (GradScaler's loss scale) and
    runs the HIP backward plan, handing each parameter its slice of the gradient arena — DDP hooks,
    GradScaler and optimizers see ordinary .grad tensors."""

    @staticmethod
    def forward(ctx, plan, x, labels, support, *params):
        ctx.plan = plan
        ctx.counted = any(ctx.needs_input_grad)         # False under no_grad: no backward will come
        if ctx.counted:
            plan.pending += 1              # the plan cache must not evict (release) it before this node's backward ran
            ctx.pin = _Pin(plan)           # ... and is released when backward runs OR the node is dropped without one
        plan.forward(x)
        out, d_raw = plan.loss(labels, support)
        ctx.d_raw = d_raw
        stats = torch.stack([out[k] for k in ("iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg")]).clone()
        ctx.mark_non_differentiable(stats)
        return out["total_loss"].clone(), stats

    @staticmethod
    def backward(ctx, g_total, _g_stats):
        plan = ctx.plan
        if ctx.counted:
            ctx.pin.release()
            ctx.counted = False
        arena = plan.backward((ctx.d_raw * g_total.float()).contiguous()).clone()
        return (None, None, None, None) + _arena_views(arena, plan.params)


def _arena_views(arena, params):
    """One gradient tensor per parameter as a view of the (cloned) flat arena, in parameter order: a single C++ call when every
    parameter is fp32 (393 Python-level slice + view pairs cost ~1 ms of host time per step on the drop-in path)."""
    if all(p.dtype == arena.dtype for p in params):
        return tuple(torch._C._nn.unflatten_dense_tensors(arena, list(params)))
    outs, o = [], 0
    for p in params:
        outs.append(arena[o:o + p.numel()].view(p.shape).to(p.dtype))
        o += p.numel()
    return tuple(outs)


class _BackboneFunction(torch.autograd.Function):
    """DFPPAFPN.forward in training mode: frames -> the three fused FPN features, batch statistics (two BatchNorm calls per
    shared module, current frame first) and running-statistics updates as in the reference (dfp_pafpn.py:109-175); backward
    takes the feature gradients through the HIP backward plan."""

    @staticmethod
    def forward(ctx, plan, x, *params):
        ctx.plan = plan
        ctx.counted = any(ctx.needs_input_grad)         # False under no_grad: no backward will come
        if ctx.counted:
            plan.pending += 1
            ctx.pin = _Pin(plan)
        fused = plan.forward(x)
        return tuple(f.export() for f in fused)

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        if ctx.counted:
            ctx.pin.release()
            ctx.counted = False
        arena = plan.backward(None, d_fused=[g.float() for g in gouts]).clone()
        return (None, None) + _arena_views(arena, plan.params)


class _HeadFunction(torch.autograd.Function):
    """TALHead.forward(xin, labels, imgs) in training mode: towers + predictions + SimOTA / Trend-Aware loss on given fused
    features; backward returns the gradients of the three features and of the head's parameters."""

    @staticmethod
    def forward(ctx, plan, labels, support, f0, f1, f2, *params):
        ctx.plan = plan
        ctx.counted = any(ctx.needs_input_grad)
        if ctx.counted:
            plan.pending += 1
            ctx.pin = _Pin(plan)
        plan.forward((f0, f1, f2))
        out, d_raw = plan.loss(labels, support)
        ctx.d_raw = d_raw
        stats = torch.stack([out[k] for k in ("iou_loss", "l1_loss", "conf_loss", "cls_loss", "num_fg")]).clone()
        ctx.mark_non_differentiable(stats)
        return out["total_loss"].clone(), stats

    @staticmethod
    def backward(ctx, g_total, _g_stats):
        plan = ctx.plan
        if ctx.counted:
            ctx.pin.release()
            ctx.counted = False
        arena = plan.backward((ctx.d_raw * g_total.float()).contiguous()).clone()
        gf = tuple(g.to(plan.tdtype) for g in plan.fused_grads())
        return (None, None, None) + gf + _arena_views(arena, plan.params)


def backbone_train_forward(pafpn, x):
    """DFPPAFPN.forward(input, mode='off_pipe') with self.training (what YOLOX.forward calls first, yolox.py:32)."""
    if x.size()[1] == 3:
        x = torch.cat([x, x], dim=1)
    assert x.size()[1] == 6
    dt = compute_dtype_for(pafpn, x)
    B, _, H, W = x.shape
    key = ("train-backbone", B, H, W, dt, str(x.device))
    plan = pafpn._plans.get(key, lambda: TrainPlan(pafpn, B, H, W, dt, x.device, parts="backbone", pool=pafpn._plans))
    return _BackboneFunction.apply(plan, x, *plan.params)


def head_train_forward(head, xin, labels):
    """TALHead.forward(xin, labels, imgs) with self.training: the reference's 6-tuple (loss, 5 * iou, conf, cls, l1, num_fg /
    num_gt) in ITS order (tal_head.py:463-470; YOLOX.forward unpacks it at yolox.py:36-38)."""
    x0 = xin[0]
    dt = compute_dtype_for(head, x0)
    shapes = tuple((int(t.shape[1]), int(t.shape[2]), int(t.shape[3])) for t in xin)
    key = ("train-head", int(x0.shape[0]), shapes, dt, str(x0.device))
    plan = head._plans.get(key, lambda: TrainPlan(head, int(x0.shape[0]), 0, 0, dt, x0.device, parts="head", feat_shapes=shapes,
                                                  pool=head._plans))
    lab, sup = split_targets_head(head, labels)
    total, stats = _HeadFunction.apply(plan, lab, sup, xin[0], xin[1], xin[2], *plan.params)
    return total, stats[0], stats[2], stats[3], stats[1], stats[4]


def split_targets_head(head, targets):
    if getattr(head, "single_labels", False) and torch.is_tensor(targets):
        return targets, targets
    return targets


def split_targets(model, targets):
    """(labels, support): TALHead takes the pair `(t, st)` (yolox.py:37-38, tal_head.py:280-286); PIPEHead one tensor,
    which then serves as its own support (model/pipe_head.py)."""
    if getattr(model.head, "single_labels", False) and torch.is_tensor(targets):
        return targets, targets
    labels, support = targets
    return labels, support


def train_forward(model, x, targets):
    """YOLOX.forward in training mode (exps/model/yolox.py:33-46): returns the reference's loss dict."""
    if x.size()[1] == 3:
        x = x.paired_with_self() if isinstance(x, FramePairsU8) else torch.cat([x, x], dim=1)
    assert x.size()[1] == 6
    plan = get_train_plan(model, x)
    labels, support = split_targets(model, targets)
    total, stats = _PlanFunction.apply(plan, x, labels, support, *plan.params)
    return {"total_loss": total, "iou_loss": stats[0], "l1_loss": stats[1], "conf_loss": stats[2],
            "cls_loss": stats[3], "num_fg": stats[4]}


class TrainStep:
    """Sync-free training step for bench.py / a native trainer: forward + loss + backward (+ one RCCL
    all-reduce of the flat gradient arena when world_size > 1); parameters' .grad are arena views."""

    def __init__(self, model, world_size=1, process_group=None, graph=None, grad_comm_dtype=None):
        """grad_comm_dtype: None / "fp32" = all-reduce the fp32 gradient arena as it is (what the reference's DDP does);
        "bf16" (or STREAMYOLO_GRAD_COMM=bf16) = opt-in gradient compression: every bucket is rounded to bf16 into a persistent
        communication buffer, reduced there (half the bytes over xGMI) and widened back into the arena — SURVEY.md 8(e)."""
        self.model, self.world, self.dist = model, world_size, process_group
        gc_ = grad_comm_dtype if grad_comm_dtype is not None else os.environ.get("STREAMYOLO_GRAD_COMM", "fp32")
        assert gc_ in ("fp32", "bf16"), gc_
        self.comm_bf16 = gc_ == "bf16"
        self._comm16 = None                         # bf16 mirror of the arena (grad_comm_dtype="bf16")
        self._t_bwd_end, self._t_comm_end = None, None
        self.exposed_allreduce_ms = None            # of the last step: all-reduce time NOT hidden behind backward
        self.plan = None
        # optional hipGraph replay of forward + loss + backward (STREAMYOLO_GRAPH=1 / graph=True).  Off by default:
        # on ROCm 7 replaying this ~1500-node graph costs MORE host time than the launch tapes (measured)
        self.use_graph = (os.environ.get("STREAMYOLO_GRAPH", "0") != "0") if graph is None else bool(graph)
        self.graph = None
        self.eager_steps = 0
        self.comm = None                            # stream the bucket all-reduces are issued from
        model.train()
        if not getattr(model.head, "single_labels", False):
            model.head.use_l1 = True                # double_trainer.py:209-216 (no_aug_epochs == max_epoch); TALHead's L1
                                                    # branch is unguarded (tal_head.py:435), PIPEHead honours use_l1

    def close(self):
        """Release the step's pin on its plan (`pending`): the plan cache may evict it again.  Called by __del__; idempotent."""
        if getattr(self, "plan", None) is not None:
            self.plan.pending = max(0, self.plan.pending - 1)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:                                    # noqa: BLE001 (interpreter shutdown)
            pass

    def _ensure(self, x):
        """The plan of this input size, through the model's plan cache on EVERY step (LRU order stays right, a size change gets
        its own plan); the step's current plan is pinned (`pending`) so that other users of the cache — the drop-in path at
        other sizes, a second TrainStep — cannot evict and release it under this one."""
        plan = get_train_plan(self.model, x)
        if plan is not self.plan:
            if self.plan is not None:
                self.plan.pending = max(0, self.plan.pending - 1)
            plan.pending += 1
            self.plan = plan
            for p in plan.params:
                p.grad = plan.gview[id(p)]
        return plan

    def _eager(self, x, lab, sup):
        plan = self.plan
        plan.forward(x)
        out, d_raw = plan.loss(lab, sup)
        plan.backward(d_raw)
        return out

    def _capture(self, x, lab, sup):
        """Record the step into a hipGraph over static copies of the inputs.  The side-stream forks join back inside
        forward() / backward(), so the capture is one connected graph whose independent branches may run concurrently."""
        if isinstance(x, FramePairsU8):
            raise NotImplementedError("TrainStep(graph=True) takes the fp32 [B,6,H,W] tensor; uint8 frame pairs "
                                      "(FramePairsU8) run through the launch tapes (graph=False, the default)")
        self.gx, self.glab, self.gsup = x.float().contiguous().clone(), lab.clone(), sup.clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.gout = self._eager(self.gx, self.glab, self.gsup)
        self.graph = g

    def step(self, x, targets):
        plan = self._ensure(x)
        self._last = (x, targets)
        lab, sup = split_targets(self.model, targets)
        self._works, self._reduced = [], set()
        plan.on_bucket = self._reduce_bucket if (self.world > 1 and not self.use_graph) else None
        graphable = self.use_graph and x.is_cuda and plan.run_table is not None and plan.tuned
        if self.graph is not None and not (plan.cache.valid() and plan.run_table.valid()):
            self.graph = None                                   # a parameter / buffer moved: the recording is stale
        if graphable and self.eager_steps >= 2 and (self.graph is None or self.glab.shape != lab.shape):
            self._capture(x, lab, sup)
        if graphable and self.graph is not None:
            self.gx.copy_(x)
            self.glab.copy_(lab)
            self.gsup.copy_(sup)
            self.graph.replay()
            out = self.gout
        else:
            out = self._eager(x, lab, sup)
            self.eager_steps += 1
        self.last_overlapped = len(self._reduced)               # buckets whose all-reduce started DURING backward
        if self.world > 1:
            # buckets whose all-reduce was not started during backward (the first two steps run the Python
            # wrappers / record the tape) go out now; then wait for all of them and average
            cuda = plan.device.type == "cuda"
            if cuda:
                if self._t_bwd_end is None:
                    self._t_bwd_end, self._t_comm_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self._t_bwd_end.record()                        # backward (all chains joined) is complete here on the main stream
            t_host = __import__("time").perf_counter()
            for k in range(len(plan.buckets)):
                if k not in self._reduced:
                    self._reduce_bucket(k, None, None)
            for w in self._works:
                w.wait()                                        # (cuda: the current stream waits for the collective)
            if self.comm_bf16:
                for lo, hi, _ in plan.buckets:                  # widen the reduced bf16 buckets back into the fp32 arena, averaged
                    plan.arena[lo:hi].copy_(self._comm16[lo:hi]).div_(self.world)   # widen FIRST: no second bf16 rounding of the quotient
            else:
                plan.arena.div_(self.world)
            if cuda:
                self._t_comm_end.record()
                self._exposed_pending = True                    # read lazily (exposed_allreduce): no synchronisation in the step
            else:
                self.exposed_allreduce_ms = (__import__("time").perf_counter() - t_host) * 1e3
        return out

    def exposed_allreduce(self):
        """ms of the last step between the end of backward and the end of the gradient exchange (all-reduce tail + averaging): the
        part of the communication that backward did not hide.  Synchronises."""
        if getattr(self, "_exposed_pending", False):
            self._t_comm_end.synchronize()
            self.exposed_allreduce_ms = self._t_bwd_end.elapsed_time(self._t_comm_end)
            self._exposed_pending = False
        return self.exposed_allreduce_ms

    def _reduce_bucket(self, k, main, side):
        """Start the RCCL all-reduce of gradient bucket k (its gradients are final on `main` and `side`); the
        collective runs beside the rest of the backward pass."""
        lo, hi, _ = self.plan.buckets[k]
        view = self.plan.arena[lo:hi]
        if self.comm_bf16:
            if self._comm16 is None or self._comm16.numel() != self.plan.arena.numel():
                self._comm16 = torch.empty(self.plan.arena.numel(), dtype=torch.bfloat16, device=self.plan.device)
            src, view = view, self._comm16[lo:hi]
        if main is not None:
            if self.comm is None:
                self.comm = torch.cuda.Stream(device=self.plan.device)
            self.comm.wait_stream(main)
            for s_ in (side if isinstance(side, (list, tuple)) else [side]):
                if s_ is not None:
                    self.comm.wait_stream(s_)
            with torch.cuda.stream(self.comm):
                if self.comm_bf16:
                    view.copy_(src)                             # fp32 -> bf16 (round to nearest even) on the communication stream
                self._works.append(self.dist.all_reduce(view, async_op=True))
        else:
            if self.comm_bf16:
                view.copy_(src)
            self._works.append(self.dist.all_reduce(view, async_op=True))
        self._reduced.add(k)

    def profile(self, iters=2):
        x, targets = self._last
        return self.plan.profile(x, targets, iters)
