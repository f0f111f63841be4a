"""sy_conv2d / sy_conv2d_wgrad against torch fp32 references of the same op (per-kernel parity)."""
import os

import pytest
import torch
import torch.nn.functional as F

from streamyolo_amd import ops
from streamyolo_amd.ops import View
from streamyolo_amd.model.packing import pack_conv_weight, pack_conv_weight_frag

TOL = {"bf16": 2e-2, "fp16": 3e-3, "fp32": 2e-5}


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def _q(t, dt):
    """round a reference tensor through the storage dtype so only accumulation differs"""
    return t.to(ops.TORCH_DTYPE[ops.dtype_code(dt)]).float()


@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("cin,cout,k,stride,H,W,N", [
    (16, 32, 3, 1, 9, 11, 2),       # stem-like: Cin 16 (taps straddle the 64-byte slab), narrow Cout
    (32, 64, 3, 2, 11, 14, 1),      # stride 2, odd sizes
    (64, 136, 1, 1, 7, 9, 2),       # 1x1, Cout not a multiple of the tile
    (24, 40, 3, 1, 6, 5, 1),        # K = 216: partial last slab
])
def test_conv_fwd_silu_residual(backend, dt, cin, cout, k, stride, H, W, N):
    g = torch.Generator().manual_seed(cin * 1000 + cout + k)
    x = _q(torch.randn(N, cin, H, W, generator=g), dt)
    w = _q(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, dt)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.3
    Ho, Wo = ops.conv_out_size(H, k, stride), ops.conv_out_size(W, k, stride)
    res = _q(torch.randn(N, cout, Ho, Wo, generator=g), dt)
    ref = F.silu(F.conv2d(x, w, None, stride, (k - 1) // 2) * scale[None, :, None, None] + shift[None, :, None, None]) + res

    # views inside wider buffers: exercises ld / channel offsets (concat-free writes)
    xb = View.alloc(N, H, W, cin + 8, dt, backend, zero=True).slice(8, cin)
    xb.set_nchw(x.to(backend))
    yb = View.alloc(N, Ho, Wo, cout + 16, dt, backend, zero=True).slice(16, cout)
    rb = View.alloc(N, Ho, Wo, cout, dt, backend)
    rb.set_nchw(res.to(backend))
    wp = pack_conv_weight(w, ops.dtype_code(dt)).to(backend)
    ops.conv2d(xb, wp, yb, k, stride, scale.to(backend), shift.to(backend), res=rb, epilogue=ops.EPI_SILU)
    got = yb.nchw().cpu()
    assert _rel(got, ref) < TOL[dt]
    # the neighbouring channels of the wide buffer stay untouched
    assert float(yb.buf[..., :16].float().abs().max()) == 0.0


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 17, 18, 19, 20, 21, 22, 23, 35, 36, 38, 39, 51, 52, 54, 55, 83, 84, 85, 86, 87, 24, 88, 89, 99, 102, 103])
@pytest.mark.parametrize("dt,mode", [("bf16", "fwd"), ("fp32", "dgrad")])
def test_conv_every_tile_configuration(backend, tile, dt, mode):
    """Each workgroup tile (256x256 / 128x256 on 8 waves, 128x128 / 64x256 / 32x256 on 4) on a shape with
    ragged channel and pixel edges, forward (with BN statistics) and data-gradient gathers."""
    g = torch.Generator().manual_seed(tile)
    N, cin, cout, k, stride, H, W = 1, (24 if tile < 80 else 32), (264 if tile < 80 else 288), 3, 2, 21, 27
    code = ops.dtype_code(dt)
    x = _q(torch.randn(N, cin, H, W, generator=g), dt).requires_grad_(True)
    w = _q(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, dt)
    y = F.conv2d(x, w, None, stride, 1)
    Ho, Wo = y.shape[2:]
    xv = View.alloc(N, H, W, cin, dt, backend); xv.set_nchw(x.detach().to(backend))
    if mode == "fwd":
        yv = View.alloc(N, Ho, Wo, cout, dt, backend)
        ssum = torch.zeros(2 * cout, device=backend); ssq = torch.zeros(2 * cout, device=backend)
        wp = pack_conv_weight(w, code)
        wf = pack_conv_weight_frag(wp, k)
        ops.conv2d(xv, wp.to(backend), yv, k, stride, stats=(ssum, ssq), tile=tile, wfrag=None if wf is None else wf.to(backend))
        assert _rel(yv.nchw().cpu(), y.detach()) < TOL[dt]
        assert _rel(ssq.view(2, cout).sum(0).cpu(), (y.detach() ** 2).sum((0, 2, 3))) < 1e-3
    else:
        dy = _q(torch.randn(y.shape, generator=g), dt)
        y.backward(dy)
        dyv = View.alloc(N, Ho, Wo, cout, dt, backend); dyv.set_nchw(dy.to(backend))
        dxv = View.alloc(N, H, W, cin, dt, backend, zero=True)
        wt = pack_conv_weight(w, code, transpose=True)
        wf = pack_conv_weight_frag(wt, k)
        ops.conv2d(dyv, wt.to(backend), dxv, k, stride, mode=ops.CONV_DGRAD, tile=tile, wfrag=None if wf is None else wf.to(backend))
        assert _rel(dxv.nchw().cpu(), x.grad) < TOL[dt]


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
@pytest.mark.parametrize("cin,cout,k,stride,H,W,N", [(32, 48, 3, 1, 8, 7, 2), (32, 64, 3, 2, 9, 12, 1), (64, 32, 1, 1, 5, 6, 2)])
def test_conv_stats_dgrad_wgrad(backend, dt, cin, cout, k, stride, H, W, N):
    g = torch.Generator().manual_seed(7 + cin + cout)
    x = _q(torch.randn(N, cin, H, W, generator=g), dt).requires_grad_(True)
    w = _q(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, dt).requires_grad_(True)
    y = F.conv2d(x, w, None, stride, (k - 1) // 2)
    Ho, Wo = y.shape[2:]
    dy = _q(torch.randn(y.shape, generator=g), dt)
    y.backward(dy)
    code = ops.dtype_code(dt)

    xv = View.alloc(N, H, W, cin, dt, backend); xv.set_nchw(x.detach().to(backend))
    yv = View.alloc(N, Ho, Wo, cout, dt, backend)
    ssum = torch.zeros(cout, device=backend); ssq = torch.zeros(cout, device=backend)
    ops.conv2d(xv, pack_conv_weight(w.detach(), code).to(backend), yv, k, stride, stats=(ssum, ssq))
    assert _rel(yv.nchw().cpu(), y.detach()) < TOL[dt]
    assert _rel(ssum.cpu(), y.detach().sum((0, 2, 3))) < 1e-3 + TOL[dt] * 0.1 or float(y.detach().sum((0, 2, 3)).abs().max()) < 1e-2
    assert _rel(ssq.cpu(), (y.detach() ** 2).sum((0, 2, 3))) < 1e-3

    dyv = View.alloc(N, Ho, Wo, cout, dt, backend); dyv.set_nchw(dy.to(backend))
    dxv = View.alloc(N, H, W, cin, dt, backend, zero=True)
    wt = pack_conv_weight(w.detach(), code, transpose=True).to(backend)
    ops.conv2d(dyv, wt, dxv, k, stride, mode=ops.CONV_DGRAD)
    assert _rel(dxv.nchw().cpu(), x.grad) < TOL[dt]
    # accumulate: second launch doubles the result
    ops.conv2d(dyv, wt, dxv, k, stride, mode=ops.CONV_DGRAD, accumulate=True)
    assert _rel(dxv.nchw().cpu(), 2 * x.grad) < 2 * TOL[dt]

    dw = torch.zeros(cout, k * k * cin, device=backend)
    ops.conv2d_wgrad(xv, dyv, dw, k, stride)
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(cout, -1)
    assert _rel(dw.cpu(), ref_dw) < TOL[dt]
    # split-K slabs + fold, parameter (OIHW) layout, accumulating on top of the first result
    ws = torch.empty(1 << 22, dtype=torch.uint8, device=backend)
    dw2 = torch.zeros(cout, cin, k, k, device=backend)
    ops.conv2d_wgrad(xv, dyv, dw2, k, stride, oihw=True, workspace=ws)
    ops.conv2d_wgrad(xv, dyv, dw2, k, stride, oihw=True, workspace=ws)
    assert _rel(dw2.cpu(), 2 * w.grad) < TOL[dt]
    # every wgrad workgroup tile
    for t in (1, 2, 3, 4, 5, 6, 17, 18, 20, 21, 22, 33, 34):   # +16 / +32: transpose-read variants
        dw3 = torch.zeros(cout, k * k * cin, device=backend)
        ops.conv2d_wgrad(xv, dyv, dw3, k, stride, workspace=ws, tile=t, target_blocks=8)
        assert _rel(dw3.cpu(), ref_dw) < TOL[dt], "wgrad tile %d" % t
    # statistics spread over replicas
    s2 = torch.zeros(4 * cout, device=backend); q2 = torch.zeros(4 * cout, device=backend)
    ops.conv2d(xv, pack_conv_weight(w.detach(), code).to(backend), yv, k, stride, stats=(s2, q2))
    assert _rel(s2.view(4, cout).sum(0).cpu(), ssum.cpu()) < 1e-4 or float(ssum.abs().max()) < 1e-2
    assert _rel(q2.view(4, cout).sum(0).cpu(), ssq.cpu()) < 1e-5


def test_head_prediction_epilogues(backend):
    """reg+obj (decode) and cls (sigmoid) 1x1 convs writing into one [B, A, 5+nc] fp32 tensor."""
    g = torch.Generator().manual_seed(3)
    N, C, H, W, nc, A0, A = 2, 32, 5, 7, 8, 10, 60
    feat = torch.randn(N, C, H, W, generator=g)
    w_ro = torch.randn(5, C, 1, 1, generator=g) * 0.1
    b_ro = torch.randn(5, generator=g) * 0.1
    w_c = torch.randn(nc, C, 1, 1, generator=g) * 0.1
    b_c = torch.randn(nc, generator=g) * 0.1
    ro = F.conv2d(feat, w_ro, b_ro)
    cl = torch.sigmoid(F.conv2d(feat, w_c, b_c))
    yv, xv = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    ref = torch.cat([(ro[:, 0:1] + xv) * 16.0, (ro[:, 1:2] + yv) * 16.0, torch.exp(ro[:, 2:4]) * 16.0,
                     torch.sigmoid(ro[:, 4:5]), cl], 1).flatten(2).permute(0, 2, 1)
    fv = View.alloc(N, H, W, C, "fp32", backend); fv.set_nchw(feat.to(backend))
    out = torch.zeros(N, A, 5 + nc, device=backend)
    base = out.data_ptr() + A0 * (5 + nc) * 4
    ops.conv2d(fv, pack_conv_weight(w_ro, ops.DT_F32).to(backend), None, 1, 1, None, b_ro.to(backend),
               epilogue=ops.EPI_DECODE, dec_stride=16.0, y_f32=True, y_ptr=base, y_ld=5 + nc, y_bs=A * (5 + nc), cout=5)
    ops.conv2d(fv, pack_conv_weight(w_c, ops.DT_F32).to(backend), None, 1, 1, None, b_c.to(backend),
               epilogue=ops.EPI_SIGMOID, y_f32=True, y_ptr=base + 5 * 4, y_ld=5 + nc, y_bs=A * (5 + nc), cout=nc)
    got = out[:, A0:A0 + H * W].cpu()
    assert _rel(got, ref) < 1e-5
    assert float(out[:, :A0].abs().max()) == 0.0 and float(out[:, A0 + H * W:].abs().max()) == 0.0


def test_wgrad_many_splits_fold(backend):
    """>= 64 pixel splits: the ZL=16 fold path (small weight tensor, long pixel axis)."""
    g = torch.Generator().manual_seed(5)
    N, cin, cout, H, W = 1, 8, 8, 136, 128
    x = torch.randn(N, cin, H, W, generator=g).requires_grad_(False)
    dy = torch.randn(N, cout, H, W, generator=g)
    ref = torch.einsum("nchw,ndhw->dc", x, dy)                       # 1x1 conv weight gradient [cout, cin]
    xv = View.alloc(N, H, W, cin, "fp32", backend); xv.set_nchw(x.to(backend))
    dv = View.alloc(N, H, W, cout, "fp32", backend); dv.set_nchw(dy.to(backend))
    ws = torch.empty(1 << 22, dtype=torch.uint8, device=backend)
    dw = torch.zeros(cout, cin, device=backend)
    ops.conv2d_wgrad(xv, dv, dw, 1, 1, workspace=ws, tile=3, target_blocks=4096)
    assert _rel(dw.cpu(), ref) < 1e-4


@pytest.mark.parametrize("tile", [96, 97, 98, 100, 101, 104, 106, 107, 109, 111, 112, 113, 114, 115, 116, 117, 118])
@pytest.mark.parametrize("dt,mode", [("bf16", "fwd"), ("fp32", "fwd"), ("bf16", "dgrad"), ("fp32", "dgrad")])
def test_conv3x3_halo_kernel(backend, tile, dt, mode):
    """csrc/conv3x3_halo.h (tile codes 104, 106, 107, 111..118 and csrc/conv3x3_halo3.h's 96, 97 (64 channels), 98, 100, 101, 109 (third generation: immediate-offset fragment reads, hand-placed instruction stream, cross-slab prefetch); 104 / 107 / 112 / 113 / 117 / 118 = the in-wave software-pipelined generation, 106 / 111 = the same with K groups inside the workgroup): 3x3 stride-1 forward with per-frame BatchNorm statistics, and the data
    gradient (first write, accumulate, channel-slice output), on an image whose width and height are ragged against the
    32-pixel / TH-row tiles, with Cout ragged against the channel tile; against torch and against the implicit-GEMM kernel."""
    g = torch.Generator().manual_seed(tile + len(mode))
    N, cin, cout, H, W = 2, 64, 72, 11, 37
    code = ops.dtype_code(dt)
    x = _q(torch.randn(N, cin, H, W, generator=g), dt).requires_grad_(True)
    w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, dt)
    y = F.conv2d(x, w, None, 1, 1)
    xv = View.alloc(N, H, W, cin + 32, dt, backend, zero=True).slice(32, cin); xv.set_nchw(x.detach().to(backend))
    if mode == "fwd":
        wp = pack_conv_weight(w, code).to(backend)
        wf = pack_conv_weight_frag(wp, 3)
        yv = View.alloc(N, H, W, cout + 8, dt, backend, zero=True).slice(8, cout)
        ssum = torch.zeros(2 * 4 * cout, device=backend); ssq = torch.zeros(2 * 4 * cout, device=backend)
        ops.conv2d(xv, wp, yv, 3, 1, stats=(ssum, ssq), tile=tile, wfrag=wf, segments=2)
        assert _rel(yv.nchw().cpu(), y.detach()) < TOL[dt]
        assert float(yv.buf[..., :8].float().abs().max()) == 0.0
        for s_ in range(2):                                           # one statistics segment per frame
            ys = y.detach()[s_:s_ + 1]
            assert _rel(ssq.view(2, 4, cout)[s_].sum(0).cpu(), (ys ** 2).sum((0, 2, 3))) < 1e-3
            assert float((ssum.view(2, 4, cout)[s_].sum(0).cpu() - ys.sum((0, 2, 3))).abs().max()) < 1e-2 * float(ys.abs().sum((0, 2, 3)).max())
        # eval-style epilogue: affine + SiLU + residual
        scale, shift = (torch.rand(cout, generator=g) + 0.5), torch.randn(cout, generator=g) * 0.3
        res = _q(torch.randn(N, cout, H, W, generator=g), dt)
        rv = View.alloc(N, H, W, cout, dt, backend); rv.set_nchw(res.to(backend))
        ops.conv2d(xv, wp, yv, 3, 1, scale.to(backend), shift.to(backend), res=rv, epilogue=ops.EPI_SILU, tile=tile, wfrag=wf)
        ref = F.silu(y.detach() * scale[None, :, None, None] + shift[None, :, None, None]) + res
        assert _rel(yv.nchw().cpu(), ref) < TOL[dt]
    else:
        dy = _q(torch.randn(y.shape, generator=g), dt)
        y.backward(dy)
        dyv = View.alloc(N, H, W, cout + 24, dt, backend, zero=True).slice(24, cout)
        # the halo kernel walks whole 64-byte channel slabs of its INPUT: pad dy's channels (zero weights) to a slab multiple
        cpad = -(-cout // (32 if dt != "fp32" else 16)) * (32 if dt != "fp32" else 16)
        dyp = View.alloc(N, H, W, cpad, dt, backend, zero=True)
        dyp.slice(0, cout).set_nchw(dy.to(backend))
        wt = pack_conv_weight(torch.cat([w, w.new_zeros(cpad - cout, cin, 3, 3)], 0), code, transpose=True).to(backend)
        wf = pack_conv_weight_frag(wt, 3)
        dxv = View.alloc(N, H, W, cin, dt, backend, zero=True)
        ops.conv2d(dyp, wt, dxv, 3, 1, mode=ops.CONV_DGRAD, tile=tile, wfrag=wf)
        assert _rel(dxv.nchw().cpu(), x.grad) < TOL[dt]
        ops.conv2d(dyp, wt, dxv, 3, 1, mode=ops.CONV_DGRAD, tile=tile, wfrag=wf, accumulate=True)
        assert _rel(dxv.nchw().cpu(), 2 * x.grad) < 2 * TOL[dt]
        ref = View.alloc(N, H, W, cin, dt, backend, zero=True)
        ops.conv2d(dyp, wt, ref, 3, 1, mode=ops.CONV_DGRAD, tile=19)  # implicit-GEMM kernel, same operands
        ops.conv2d(dyp, wt, dxv, 3, 1, mode=ops.CONV_DGRAD, tile=tile, wfrag=wf)
        assert _rel(dxv.nchw().cpu(), ref.nchw().cpu()) < (1e-5 if dt == "fp32" else 1e-2)


@pytest.mark.parametrize("tile", [109, 100, 101, 98, 104, 117])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_dgrad_carries_the_producers_bn_backward_reduce(backend, tile, dt):
    """SY_EPI_BNR: the data gradient of a 3x3 layer also reduces the BatchNorm backward sums of the layer that produced its input
    (z -> BN -> SiLU -> a -> conv3x3): dx unchanged, and sy_bn_silu_bwd_apply fed with the fused sums (raw second moment) gives the
    same dz / dgamma / dbeta as behind the separate sy_bn_silu_bwd_reduce launch — and as torch autograd of the chain."""
    g = torch.Generator().manual_seed(tile)
    N, C, cout, H, W = 2, 64, 96, 9, 37
    code = ops.dtype_code(dt)
    z = _q(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3, dt).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    a = F.silu(F.batch_norm(z, None, None, gamma, beta, True, 0.03, 1e-3))
    w = _q(torch.randn(cout, C, 3, 3, generator=g) / (C * 9) ** 0.5, dt)
    y = F.conv2d(a, w, None, 1, 1)
    dy = _q(torch.randn(y.shape, generator=g), dt)
    y.backward(dy)
    dev = backend
    zv = View.alloc(N, H, W, C, dt, dev); zv.set_nchw(z.detach().to(dev))
    ssum = z.detach().sum((0, 2, 3)).to(dev); ssq = (z.detach() ** 2).sum((0, 2, 3)).to(dev)
    scale, shift, mean, invstd = [torch.empty(C, device=dev) for _ in range(4)]
    ops.bn_finalize(ssum, ssq, N * H * W, gamma.detach().to(dev), beta.detach().to(dev), 1e-3, 0.03, None, None, scale, shift, mean, invstd)
    dyp = View.alloc(N, H, W, cout, dt, dev); dyp.set_nchw(dy.to(dev))
    wt = pack_conv_weight(w, code, transpose=True).to(dev)
    wf = pack_conv_weight_frag(wt, 3)
    # separate launches: data gradient, reduce, apply
    da0 = View.alloc(N, H, W, C, dt, dev)
    ops.conv2d(dyp, wt, da0, 3, 1, mode=ops.CONV_DGRAD, tile=tile, wfrag=wf)
    sums0 = torch.zeros(2 * 2 * C, device=dev)
    ops.bn_silu_bwd_reduce(zv, da0, scale, shift, mean, invstd, sums0)
    dz0 = View.alloc(N, H, W, C, dt, dev)
    dg0, db0 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_silu_bwd_apply(zv, da0, scale, shift, mean, invstd, gamma.detach().to(dev), sums0, dz0, dg0, db0)
    # fused: the reduce rides in the data gradient's epilogue
    da1 = View.alloc(N, H, W, C, dt, dev)
    sums1 = torch.zeros(2 * 2 * C, device=dev)
    ops.conv2d(dyp, wt, da1, 3, 1, mode=ops.CONV_DGRAD, tile=tile, wfrag=wf, bn_reduce=(zv, scale, shift, sums1))
    assert torch.equal(da1.buf, da0.buf)
    dz1 = View.alloc(N, H, W, C, dt, dev)
    dg1, db1 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_silu_bwd_apply(zv, da1, scale, shift, mean, invstd, gamma.detach().to(dev), sums1, dz1, dg1, db1, raw_moment=True)
    # (the fused sums see the fp32 accumulators, the separate pass the rounded gradient: agreement to the rounding step)
    assert _rel(db1.cpu(), db0.cpu()) < TOL[dt] and _rel(dg1.cpu(), dg0.cpu()) < TOL[dt]
    assert _rel(dz1.nchw().cpu(), dz0.nchw().cpu()) < TOL[dt]
    assert _rel(db1.cpu(), beta.grad) < 2 * TOL[dt] and _rel(dg1.cpu(), gamma.grad) < 2 * TOL[dt]
    assert _rel(dz1.nchw().cpu(), z.grad) < 2 * TOL[dt]


@pytest.mark.parametrize("tile", [52, 59, 60])
@pytest.mark.parametrize("N,cin,cout,H,W", [(2, 64, 144, 7, 37), (1, 32, 48, 5, 70), (1, 128, 32, 9, 33)])
def test_wgrad_all_taps_kernel(backend, tile, N, cin, cout, H, W):
    """conv_wgrad9_kernel (tile codes 52; 59 / 60 = the slab loop as one instruction stream, 60 on eight waves x 64 input channels):
    3x3 stride-1 weight gradient with all nine taps per workgroup and the x halo
    window resident in LDS — ragged 32-pixel row segments, ragged Cout tile, one split and many splits (+ fold), packed and
    OIHW layouts, against torch and against the per-tap transpose-read kernel."""
    if tile == 60 and cin % 64:
        pytest.skip("64-input-channel workgroups")
    dt = "bf16"
    g = torch.Generator().manual_seed(tile + cin)
    x = _q(torch.randn(N, cin, H, W, generator=g), dt)
    w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, dt).requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1)
    dy = _q(torch.randn(y.shape, generator=g), dt)
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).reshape(cout, -1)
    xv = View.alloc(N, H, W, cin + 16, dt, backend, zero=True).slice(16, cin); xv.set_nchw(x.to(backend))
    dyv = View.alloc(N, H, W, cout, dt, backend); dyv.set_nchw(dy.to(backend))
    ws = torch.empty(1 << 24, dtype=torch.uint8, device=backend)
    for tb in (1, 8, 4096):
        dw = torch.zeros(cout, 9 * cin, device=backend)
        ops.conv2d_wgrad(xv, dyv, dw, 3, 1, workspace=ws, tile=tile, target_blocks=tb)
        assert _rel(dw.cpu(), ref) < TOL[dt], "splits target %d" % tb
    dw2 = torch.zeros(cout, cin, 3, 3, device=backend)
    ops.conv2d_wgrad(xv, dyv, dw2, 3, 1, oihw=True, workspace=ws, tile=tile, target_blocks=16)
    ops.conv2d_wgrad(xv, dyv, dw2, 3, 1, oihw=True, workspace=None, tile=tile)             # one split: += in place
    assert _rel(dw2.cpu(), 2 * w.grad) < TOL[dt]
    dw3 = torch.zeros(cout, 9 * cin, device=backend)
    ops.conv2d_wgrad(xv, dyv, dw3, 3, 1, workspace=ws, tile=17, target_blocks=8)
    dw4 = torch.zeros(cout, 9 * cin, device=backend)
    ops.conv2d_wgrad(xv, dyv, dw4, 3, 1, workspace=ws, tile=tile, target_blocks=8)
    # the two kernels cut the pixel range into different splits, and in bf16 mode the split-K slabs are stored as bf16 (fp32
    # fold): they agree to the slab rounding (2^-9 per partial sum), both within the bf16 bound of the torch reference above
    assert _rel(dw4.cpu(), dw3.cpu()) < 1e-2


@pytest.mark.parametrize("tile", [121, 122, 123, 124])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("cin,cout,N,H,W", [(64, 72, 2, 7, 9), (128, 160, 4, 5, 13), (256, 128, 2, 9, 11), (512, 64, 2, 6, 13),
                                           (1024, 128, 2, 5, 9), (2048, 64, 2, 4, 9)])
def test_conv1x1_tile_kernel(backend, tile, dt, cin, cout, N, H, W):
    """csrc/conv1x1_tile.h (tile codes 121..124): 1x1 stride-1 training forward (raw output + per-frame BatchNorm
    statistics), the eval epilogue (affine + SiLU + residual), and the data gradient (first write and accumulate), with pixel
    and channel tiles ragged against the image; against torch and against the implicit-GEMM kernel."""
    if (tile == 122 and cout > 64) or (tile == 123 and cin > 256) or (tile not in (121, 124) and cin > 512):
        pytest.skip("not a tuner candidate for this shape")
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(cin + cout + tile)
    x = _q(torch.randn(N, cin, H, W, generator=g), dt).requires_grad_(True)
    w = _q(torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5, dt)
    y = F.conv2d(x, w)
    xv = View.alloc(N, H, W, cin + 8, dt, backend, zero=True).slice(8, cin); xv.set_nchw(x.detach().to(backend))
    wp = pack_conv_weight(w, code).to(backend)
    wf = pack_conv_weight_frag(wp, 1)
    yv = View.alloc(N, H, W, cout + 16, dt, backend, zero=True).slice(16, cout)
    ssum = torch.zeros(2 * 4 * cout, device=backend); ssq = torch.zeros(2 * 4 * cout, device=backend)
    ops.conv2d(xv, wp, yv, 1, 1, stats=(ssum, ssq), tile=tile, wfrag=wf, segments=2)
    assert _rel(yv.nchw().cpu(), y.detach()) < TOL[dt]
    assert float(yv.buf[..., :16].float().abs().max()) == 0.0
    for s_ in range(2):
        ys = y.detach()[s_ * (N // 2):(s_ + 1) * (N // 2)]
        assert _rel(ssq.view(2, 4, cout)[s_].sum(0).cpu(), (ys ** 2).sum((0, 2, 3))) < 1e-3
        assert float((ssum.view(2, 4, cout)[s_].sum(0).cpu() - ys.sum((0, 2, 3))).abs().max()) < 1e-2 * float(ys.abs().sum((0, 2, 3)).max())
    ref = View.alloc(N, H, W, cout, dt, backend)
    ops.conv2d(xv, wp, ref, 1, 1, stats=(torch.zeros_like(ssum), torch.zeros_like(ssq)), tile=19, segments=2)
    assert _rel(yv.nchw().cpu(), ref.nchw().cpu()) < 1e-2
    # eval-style epilogue: affine + SiLU + residual
    scale, shift = (torch.rand(cout, generator=g) + 0.5), torch.randn(cout, generator=g) * 0.3
    res = _q(torch.randn(N, cout, H, W, generator=g), dt)
    rv = View.alloc(N, H, W, cout, dt, backend); rv.set_nchw(res.to(backend))
    ops.conv2d(xv, wp, yv, 1, 1, scale.to(backend), shift.to(backend), res=rv, epilogue=ops.EPI_SILU, tile=tile, wfrag=wf)
    assert _rel(yv.nchw().cpu(), F.silu(y.detach() * scale[None, :, None, None] + shift[None, :, None, None]) + res) < TOL[dt]
    # data gradient: dy [N,H,W,cout_pad] x W^T -> dx, then +=
    if cout in (64, 128, 256, 512):
        dy = _q(torch.randn(y.shape, generator=g), dt)
        y.backward(dy)
        dyv = View.alloc(N, H, W, cout, dt, backend); dyv.set_nchw(dy.to(backend))
        wt = pack_conv_weight(w, code, transpose=True).to(backend)
        wft = pack_conv_weight_frag(wt, 1)
        dxv = View.alloc(N, H, W, cin + 32, dt, backend, zero=True).slice(32, cin)
        if (tile == 122 and cin > 64) or (tile == 123 and cout > 256):
            return
        ops.conv2d(dyv, wt, dxv, 1, 1, mode=ops.CONV_DGRAD, tile=tile, wfrag=wft)
        assert _rel(dxv.nchw().cpu(), x.grad) < TOL[dt]
        ops.conv2d(dyv, wt, dxv, 1, 1, mode=ops.CONV_DGRAD, tile=tile, wfrag=wft, accumulate=True)
        assert _rel(dxv.nchw().cpu(), 2 * x.grad) < 2 * TOL[dt]
        assert float(dxv.buf[..., :32].float().abs().max()) == 0.0



@pytest.mark.parametrize("tile", [117, 118])
@pytest.mark.parametrize("dt,splits", [("fp16", 2), ("fp16", 4), ("bf16", 3), ("fp32", 2)])
def test_conv3x3_split_k_equals_the_single_pass_kernel(backend, tile, dt, splits):
    """sy_conv_desc::k_splits + sy_splitk_epilogue (the batch-1 streaming step's deep small-map layers): the channel slabs
    cut into `splits` ranges, fp32 partials summed in split order, then the same epilogue arithmetic (affine, SiLU, residual,
    one rounding) — against the single-pass halo kernel and against torch; ragged image edges, a residual view inside a
    wider buffer, Cin of six channel slabs (an uneven 6 / 4 split included)."""
    g = torch.Generator().manual_seed(tile + splits)
    epc = 4 if dt == "fp32" else 8
    N, cin, cout, H, W = 1, 6 * 4 * epc, 160, 7, 37
    x = _q(torch.randn(N, cin, H, W, generator=g), dt)
    w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, dt)
    scale = (torch.rand(cout, generator=g) + 0.5).to(backend)
    shift = (torch.randn(cout, generator=g) * 0.3).to(backend)
    res = _q(torch.randn(N, cout, H, W, generator=g), dt)
    code = ops.dtype_code(dt)
    xv = View.alloc(N, H, W, cin, dt, backend); xv.set_nchw(x.to(backend))
    rv = View.alloc(N, H, W, cout + 16, dt, backend, zero=True).slice(16, cout); rv.set_nchw(res.to(backend))
    wp = pack_conv_weight(w, code)
    wf = pack_conv_weight_frag(wp, 3).to(backend)
    wp = wp.to(backend)
    y0 = View.alloc(N, H, W, cout, dt, backend)
    ops.conv2d(xv, wp, y0, 3, 1, scale, shift, res=rv, epilogue=ops.EPI_SILU, tile=tile, wfrag=wf)
    y1 = View.alloc(N, H, W, cout, dt, backend)
    part = torch.full((splits * N * H * W * cout + 64,), float("nan"), device=backend)       # every partial must be written
    ops.conv2d_splitk(xv, wp, y1, 3, 1, scale, shift, part, splits, res=rv, epilogue=ops.EPI_SILU, tile=tile, wfrag=wf)
    ref = F.silu(F.conv2d(x, w, None, 1, 1) * scale.cpu()[None, :, None, None] + shift.cpu()[None, :, None, None]) + res
    assert torch.isfinite(y1.nchw()).all()
    assert _rel(y1.nchw().cpu(), ref) < TOL[dt]
    # vs the single-pass kernel: same products, a different fp32 summation order -> a few last-bit flips of the stored type
    assert _rel(y1.nchw().cpu(), y0.nchw().cpu()) < {"fp32": 1e-5, "fp16": 2e-3, "bf16": 1.6e-2}[dt]


@pytest.mark.parametrize("tile", [110, 105])
@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("N,cin,cout,H,W", [(2, 64, 72, 11, 37), (1, 32, 160, 8, 66), (2, 64, 128, 5, 130)])
def test_conv3x3_stride2_halo_kernel(backend, tile, dt, N, cin, cout, H, W):
    """conv3x3_halo2_kernel<..., S2> (tile code 110; 105 = the same with two K groups of waves): the 3x3 STRIDE-2 forward with the (2 TH + 1) x 65 input window resident in
    LDS, columns split by parity — odd and even input sizes, ragged output tiles and channel tiles, training epilogue (raw output
    + per-frame statistics) and eval epilogue (affine + SiLU), against torch and against the implicit-GEMM kernel."""
    code = ops.dtype_code(dt)
    g = torch.Generator().manual_seed(cin + cout + W)
    x = _q(torch.randn(N, cin, H, W, generator=g), dt)
    w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, dt)
    y = F.conv2d(x, w, None, 2, 1)
    Ho, Wo = y.shape[2], y.shape[3]
    xv = View.alloc(N, H, W, cin + 32, dt, backend, zero=True).slice(32, cin); xv.set_nchw(x.to(backend))
    wp = pack_conv_weight(w, code).to(backend)
    wf = pack_conv_weight_frag(wp, 3)
    yv = View.alloc(N, Ho, Wo, cout + 8, dt, backend, zero=True).slice(8, cout)
    segs = 2 if N % 2 == 0 else 1
    ssum = torch.zeros(segs * 4 * cout, device=backend); ssq = torch.zeros(segs * 4 * cout, device=backend)
    ops.conv2d(xv, wp, yv, 3, 2, stats=(ssum, ssq), tile=tile, wfrag=wf, segments=segs)
    assert _rel(yv.nchw().cpu(), y) < TOL[dt]
    assert float(yv.buf[..., :8].float().abs().max()) == 0.0
    for s_ in range(segs):
        ys = y[s_ * (N // segs):(s_ + 1) * (N // segs)]
        assert _rel(ssq.view(segs, 4, cout)[s_].sum(0).cpu(), (ys ** 2).sum((0, 2, 3))) < 1e-3
        assert float((ssum.view(segs, 4, cout)[s_].sum(0).cpu() - ys.sum((0, 2, 3))).abs().max()) < 1e-2 * float(ys.abs().sum((0, 2, 3)).max())
    scale, shift = (torch.rand(cout, generator=g) + 0.5), torch.randn(cout, generator=g) * 0.3
    ops.conv2d(xv, wp, yv, 3, 2, scale.to(backend), shift.to(backend), epilogue=ops.EPI_SILU, tile=tile, wfrag=wf)
    ref = F.silu(y * scale[None, :, None, None] + shift[None, :, None, None])
    assert _rel(yv.nchw().cpu(), ref) < TOL[dt]
    ref_v = View.alloc(N, Ho, Wo, cout, dt, backend)
    ops.conv2d(xv, wp, ref_v, 3, 2, scale.to(backend), shift.to(backend), epilogue=ops.EPI_SILU, tile=19)   # implicit GEMM, same operands
    assert _rel(yv.nchw().cpu(), ref_v.nchw().cpu()) < (1e-5 if dt == "fp32" else 1e-2)
    # the data gradient of a stride-2 layer stays on the implicit-GEMM kernel
    with pytest.raises(ops._lib.HipLibraryError):
        ops.conv2d(ref_v, wp, xv, 3, 2, mode=ops.CONV_DGRAD, tile=tile, wfrag=wf)


@pytest.mark.parametrize("dt", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("tile", [108, 125, 126, 127])
@pytest.mark.parametrize("N,cin,cout,H,W", [(2, 64, 72, 11, 37), (1, 32, 160, 8, 66), (2, 64, 128, 6, 129), (1, 160, 64, 9, 70)])
def test_conv3x3_stride2_data_gradient_kernel(backend, dt, tile, N, cin, cout, H, W):
    """conv3x3_s2dgrad_kernel (tile code 108): the data gradient of a 3x3 stride-2 convolution as four output-parity classes of 1 / 2 /
    2 / 4 taps read from a (TH + 1) x 34 window of dy in LDS; conv3x3_s2dgrad4_kernel (125 / 126 / 127, round 6): the same with all four
    classes in one workgroup (window parked once, four accumulator sets, one epilogue per class) — odd and even input sizes (ragged
    last class row / column), ragged channel tiles, first write and +=, against torch and against the implicit-GEMM kernel."""
    code = ops.dtype_code(dt)
    slab = 16 if dt == "fp32" else 32
    g = torch.Generator().manual_seed(cin + cout + W)
    x = _q(torch.randn(N, cin, H, W, generator=g), dt).requires_grad_(True)
    w = _q(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, dt)
    y = F.conv2d(x, w, None, 2, 1)
    dy = _q(torch.randn(y.shape, generator=g), dt)
    y.backward(dy)
    Hs, Ws = y.shape[2], y.shape[3]
    cpad = -(-cout // slab) * slab                                   # the window kernel walks whole channel slabs of dy
    dyp = View.alloc(N, Hs, Ws, cpad, dt, backend, zero=True)
    dyp.slice(0, cout).set_nchw(dy.to(backend))
    wt = pack_conv_weight(torch.cat([w, w.new_zeros(cpad - cout, cin, 3, 3)], 0), code, transpose=True).to(backend)
    wf = pack_conv_weight_frag(wt, 3)
    dxv = View.alloc(N, H, W, cin + 8, dt, backend, zero=True).slice(8, cin)
    ops.conv2d(dyp, wt, dxv, 3, 2, mode=ops.CONV_DGRAD, tile=tile, wfrag=wf)
    assert _rel(dxv.nchw().cpu(), x.grad) < TOL[dt]
    assert float(dxv.buf[..., :8].float().abs().max()) == 0.0
    ops.conv2d(dyp, wt, dxv, 3, 2, mode=ops.CONV_DGRAD, tile=tile, wfrag=wf, accumulate=True)
    assert _rel(dxv.nchw().cpu(), 2 * x.grad) < 2 * TOL[dt]
    ref = View.alloc(N, H, W, cin, dt, backend, zero=True)
    ops.conv2d(dyp, wt, ref, 3, 2, mode=ops.CONV_DGRAD, tile=19)      # implicit GEMM (four parity classes), same operands
    ops.conv2d(dyp, wt, dxv, 3, 2, mode=ops.CONV_DGRAD, tile=tile, wfrag=wf)
    assert _rel(dxv.nchw().cpu(), ref.nchw().cpu()) < (1e-5 if dt == "fp32" else 1e-2)
    with pytest.raises(ops._lib.HipLibraryError):                    # forward launches are not this kernel's
        ops.conv2d(dxv, wt, dyp, 3, 2, tile=tile, wfrag=wf)


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
@pytest.mark.parametrize("N,cin,hid,cout,H,W,shortcut", [(2, 64, 64, 64, 7, 37, True), (1, 128, 64, 160, 5, 70, False), (1, 256, 256, 256, 4, 33, True)])
def test_bottleneck_fused_equals_the_two_launches(backend, dt, N, cin, hid, cout, H, W, shortcut):
    """csrc/bottleneck_fused.h (tile code 119): yolox Bottleneck forward in eval mode — conv1 1x1 + BN + SiLU -> conv2 3x3 + BN +
    SiLU (+ x) — as ONE launch with the hidden activation kept in LDS, against the two launches it replaces (1x1 tile kernel, then
    the halo kernel with the residual epilogue) and against torch: image width / height ragged against the 2 x 32 tile, hidden
    tiles shared unevenly by the four waves, output channels ragged against the 128-channel tile, channel-slice views."""
    g = torch.Generator().manual_seed(cin + hid + W)
    code = ops.dtype_code(dt)
    x = _q(torch.randn(N, cin, H, W, generator=g), dt)
    w1 = _q(torch.randn(hid, cin, 1, 1, generator=g) / cin ** 0.5, dt)
    w2 = _q(torch.randn(cout, hid, 3, 3, generator=g) / (hid * 9) ** 0.5, dt)
    s1, b1 = torch.rand(hid, generator=g) + 0.5, torch.randn(hid, generator=g) * 0.3
    s2, b2 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    if shortcut:
        assert cin == cout
    xv = View.alloc(N, H, W, cin + 32, dt, backend, zero=True).slice(32, cin); xv.set_nchw(x.to(backend))
    w1p = pack_conv_weight(w1, code).to(backend); w1f = pack_conv_weight_frag(w1p, 1)
    w2p = pack_conv_weight(w2, code).to(backend); w2f = pack_conv_weight_frag(w2p, 3)
    dev = lambda t: t.to(backend)                                                     # noqa: E731
    # the two launches
    hv = View.alloc(N, H, W, hid, dt, backend)
    ops.conv2d(xv, w1p, hv, 1, 1, dev(s1), dev(b1), epilogue=ops.EPI_SILU, tile=121, wfrag=w1f)
    y2 = View.alloc(N, H, W, cout + 8, dt, backend, zero=True).slice(8, cout)
    ops.conv2d(hv, w2p, y2, 3, 1, dev(s2), dev(b2), res=xv if shortcut else None, epilogue=ops.EPI_SILU, tile=117, wfrag=w2f)
    # one launch
    y1 = View.alloc(N, H, W, cout + 8, dt, backend, zero=True).slice(8, cout)
    ops.conv2d(xv, w2p, y1, 3, 1, dev(s2), dev(b2), res=xv if shortcut else None, epilogue=ops.EPI_SILU, tile=119, wfrag=w2f,
               pre=(w1f, dev(s1), dev(b1)))
    a, b = y1.nchw().cpu(), y2.nchw().cpu()
    assert float(y1.buf[..., :8].float().abs().max()) == 0.0
    # same products, same fp32 accumulation order over K in both stages, the hidden activation rounded to the storage type in both:
    # the two paths agree to the last bit or to one rounding of the output type
    assert float((a - b).abs().max()) <= 2.0 ** (-7 if dt == "bf16" else -10) * float(b.abs().max()) * 1.01
    hid_ref = F.silu(F.conv2d(x, w1) * s1[None, :, None, None] + b1[None, :, None, None])
    ref = F.silu(F.conv2d(_q(hid_ref, dt), w2, None, 1, 1) * s2[None, :, None, None] + b2[None, :, None, None]) + (x if shortcut else 0)
    assert _rel(a, ref) < TOL[dt]
    # a launch without the pre_* operands on tile 119 (or with them on another tile) is an argument error
    with pytest.raises(Exception):
        ops.conv2d(hv, w2p, y1, 3, 1, dev(s2), dev(b2), epilogue=ops.EPI_SILU, tile=119, wfrag=w2f)
