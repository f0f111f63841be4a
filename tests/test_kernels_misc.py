"""Glue kernels (focus, resize, SPP, BN-train, NMS) against torch fp32 references / the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import streamyolo_oracle as O
from streamyolo_amd import ops
from streamyolo_amd.ops import View


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_focus_pack(backend, dt):
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 6, 12, 20, generator=g) * 255
    out = View.alloc(2, 6, 10, 16, dt, backend)
    ops.focus_pack(x.to(backend), 3, out)
    ref = O.focus_pack(x[:, 3:6])
    got = out.nchw().cpu()
    tol = 1e-2 if dt == "bf16" else 0.0
    assert _rel(got[:, :12], ref) <= tol
    assert float(got[:, 12:].abs().max()) == 0.0


@pytest.mark.parametrize("hi,wi,ho,wo", [(19, 30, 38, 60), (38, 60, 75, 120), (10, 13, 19, 25), (5, 5, 5, 5)])
def test_resize_nearest_fwd_bwd(backend, hi, wi, ho, wo):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8, hi, wi, generator=g, requires_grad=True)
    ref = F.interpolate(x, size=(ho, wo), mode="nearest")
    dref = torch.randn(ref.shape, generator=g)
    ref.backward(dref)
    src = View.alloc(2, hi, wi, 8, "fp32", backend); src.set_nchw(x.detach().to(backend))
    wide = View.alloc(2, ho, wo, 24, "fp32", backend, zero=True)
    dst = wide.slice(8, 8)
    ops.resize_nearest(src, dst)
    assert torch.equal(dst.nchw().cpu(), ref.detach())
    dd = View.alloc(2, ho, wo, 8, "fp32", backend); dd.set_nchw(dref.to(backend))
    ds = View.alloc(2, hi, wi, 8, "fp32", backend, zero=True)
    ops.resize_nearest_bwd(dd, ds, accumulate=False)
    assert _rel(ds.nchw().cpu(), x.grad) < 1e-6
    ops.resize_nearest_bwd(dd, ds, accumulate=True)
    assert _rel(ds.nchw().cpu(), 2 * x.grad) < 1e-6


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_spp_pool_fwd_bwd(backend, dt):
    g = torch.Generator().manual_seed(2)
    C, H, W = 8, 9, 14
    x = torch.randn(1, C, H, W, generator=g).to(ops.TORCH_DTYPE[ops.dtype_code(dt)]).float().requires_grad_(True)
    pools = [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)]
    ref = torch.cat([x] + pools, 1)
    v = View.alloc(1, H, W, 4 * C, dt, backend, zero=True)
    v.slice(0, C).set_nchw(x.detach().to(backend))
    am = torch.zeros((1, H, W, 3, C), dtype=torch.uint8, device=backend)
    ops.spp_pool(v, am)
    assert torch.equal(v.nchw().cpu(), ref.detach())
    if dt == "fp32":        # tie-free inputs: arg-max routing is well defined
        dref = torch.randn(ref.shape, generator=g)
        ref.backward(dref)
        dv = View.alloc(1, H, W, 4 * C, dt, backend); dv.set_nchw(dref.to(backend))
        ops.spp_pool_bwd(dv, am)
        assert _rel(dv.slice(0, C).nchw().cpu(), x.grad) < 1e-6


@pytest.mark.parametrize("dt,H,W,C", [("bf16", 19, 30, 16), ("fp32", 7, 5, 16), ("bf16", 3, 5, 16), ("bf16", 6, 7, 1024)])
def test_spp_tile_kernels_match_scan_kernels(backend, dt, H, W, C, monkeypatch):
    """The LDS-tiled (separable) SPP kernels must reproduce the scan kernels bit for bit — pooled values, the
    arg-max bytes (ties included: inputs quantised to a few levels); routed gradients to rounding.  Launches of < 256 channel
    chunks run the three pooling levels as three workgroups per chunk (C = 16), larger ones one workgroup per chunk (C = 1024)."""
    g = torch.Generator().manual_seed(5)
    N = 2
    x = (torch.randn(N, C, H, W, generator=g) * 2).round() / 2          # many exact ties
    dy = torch.randn(N, 4 * C, H, W, generator=g)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("SY_SPP_SCAN", mode)
        v = View.alloc(N, H, W, 4 * C, dt, backend, zero=True)
        v.slice(0, C).set_nchw(x.to(backend))
        am = torch.zeros((N, H, W, 3, C), dtype=torch.uint8, device=backend)
        ops.spp_pool(v, am)
        dv = View.alloc(N, H, W, 4 * C, dt, backend); dv.set_nchw(dy.to(backend))
        ops.spp_pool_bwd(dv, am)
        res[mode] = (v.nchw().cpu().clone(), am.cpu().clone(), dv.nchw().cpu().clone())
    pools = [F.max_pool2d(x, k, 1, k // 2) for k in (5, 9, 13)]
    assert torch.equal(res["0"][0].float(), torch.cat([x] + pools, 1))
    assert torch.equal(res["0"][0], res["1"][0]) and torch.equal(res["0"][1], res["1"][1])
    # the tiled backward groups its additions by window row: same terms, fp32 sums may differ in the last bit
    ga, gb = res["0"][2].float(), res["1"][2].float()
    assert float((ga - gb).abs().max()) <= (1e-5 if dt == "fp32" else 2e-2) * float(gb.abs().max())


@pytest.mark.parametrize("copies,nseg", [(200, 1), (777, 2), (64, 2)])
def test_bn_finalize_folds_many_replica_rows(backend, copies, nseg):
    """sy_bn_finalize over hundreds of replica rows (the exact mode's one row per convolution workgroup): the rows are folded into
    32 by a fixed tree (fold_rows_kernel) and then finalized — same statistics as the plain sum, the same bits on a second call
    over the same rows, segment by segment; and the backward replicas' fold (sy_bn_silu_bwd_apply over 768 rows) likewise."""
    g = torch.Generator().manual_seed(copies)
    C, M = 96, 4096
    rows = torch.randn(nseg, copies, C, generator=g)
    sq = torch.rand(nseg, copies, C, generator=g) * 3 + 1.0
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    dev = backend
    outs = []
    for _ in range(2):
        ssum, ssq = rows.clone().to(dev).reshape(-1), (sq * M / copies).clone().to(dev).reshape(-1)
        scale, shift, mean, invstd = [torch.empty(nseg * C, device=dev) for _ in range(4)]
        ops.bn_finalize(ssum, ssq, M, gamma.to(dev), beta.to(dev), 1e-3, 0.03, None, None, scale, shift, mean, invstd, nseg=nseg)
        outs.append((mean.cpu().clone(), invstd.cpu().clone(), scale.cpu().clone()))
    m_ref = rows.double().sum(1) / M
    v_ref = (sq.double() * M / copies).sum(1) / M - m_ref ** 2
    assert _rel(outs[0][0].view(nseg, C), m_ref.float()) < 1e-5
    assert _rel(outs[0][1].view(nseg, C), (1.0 / (v_ref + 1e-3).sqrt()).float()) < 1e-5
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))


@pytest.mark.parametrize("C", [16, 96, 192])          # 96 / 192: the backward reduce runs in channel slices of 48 / 64
@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_bn_train_silu_fwd_bwd(backend, dt, C):
    g = torch.Generator().manual_seed(4)
    N, H, W = 2, 6, 5
    tdt = ops.TORCH_DTYPE[ops.dtype_code(dt)]
    y = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(tdt).float().requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    res = torch.randn(N, C, H, W, generator=g).to(tdt).float()
    a = F.silu(F.batch_norm(y, rm_ref, rv_ref, gamma, beta, True, 0.03, 1e-3)) + res
    da = torch.randn(a.shape, generator=g).to(tdt).float()
    a.backward(da)

    dev = backend
    yv = View.alloc(N, H, W, C, dt, dev); yv.set_nchw(y.detach().to(dev))
    ssum = y.detach().sum((0, 2, 3)).to(dev); ssq = (y.detach() ** 2).sum((0, 2, 3)).to(dev)
    scale, shift, mean, invstd = [torch.empty(C, device=dev) for _ in range(4)]
    rmd, rvd = rm.to(dev), rv.to(dev)
    ops.bn_finalize(ssum, ssq, N * H * W, gamma.detach().to(dev), beta.detach().to(dev), 1e-3, 0.03, rmd, rvd,
                    scale, shift, mean, invstd)
    assert _rel(rmd.cpu(), rm_ref) < 1e-5 and _rel(rvd.cpu(), rv_ref) < 1e-5
    rv_ = View.alloc(N, H, W, C, dt, dev); rv_.set_nchw(res.to(dev))
    av = View.alloc(N, H, W, C, dt, dev)
    ops.bn_silu_apply(yv, scale, shift, av, res=rv_)
    tol = 2e-2 if dt == "bf16" else 1e-5
    assert _rel(av.nchw().cpu(), a.detach()) < tol
    dav = View.alloc(N, H, W, C, dt, dev); dav.set_nchw(da.to(dev))
    sums = torch.zeros(2 * C, device=dev)
    ops.bn_silu_bwd_reduce(yv, dav, scale, shift, mean, invstd, sums)
    dyv = View.alloc(N, H, W, C, dt, dev)
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_silu_bwd_apply(yv, dav, scale, shift, mean, invstd, gamma.detach().to(dev), sums, dyv, dgam, dbet)
    assert _rel(dbet.cpu(), beta.grad) < 1e-4 and _rel(dgam.cpu(), gamma.grad) < 1e-4
    assert _rel(dyv.nchw().cpu(), y.grad) < (3e-2 if dt == "bf16" else 1e-4)
    # residual branch gradient written by the same pass (a = silu(bn(y)) + res  =>  dres = da), then accumulated
    wide = View.alloc(N, H, W, 2 * C, dt, dev, zero=True)
    drv, dy2 = wide.slice(C, C), View.alloc(N, H, W, C, dt, dev)
    ops.bn_silu_bwd_apply(yv, dav, scale, shift, mean, invstd, gamma.detach().to(dev), sums, dy2, dres=drv)
    assert torch.equal(drv.nchw(), dav.nchw()) and torch.equal(dy2.buf, dyv.buf)
    assert torch.equal(wide.slice(0, C).nchw(), torch.zeros_like(dav.nchw()))
    ops.bn_silu_bwd_apply(yv, dav, scale, shift, mean, invstd, gamma.detach().to(dev), sums, dy2, dres=drv, dres_accumulate=True)
    want = (dav.nchw().float() * 2).to(dav.buf.dtype)
    assert torch.equal(drv.nchw(), want)


@pytest.mark.parametrize("dt,C,copies,nseg", [("bf16", 192, 32, 2), ("bf16", 16, 3, 1), ("fp16", 96, 8, 2), ("fp32", 64, 32, 1), ("bf16", 512, 8, 2)])
def test_bn_finalize_apply_fused_equals_the_two_launches(backend, dt, C, copies, nseg):
    """sy_bn_finalize_apply (one launch, channel-sliced workgroups folding their own replicas) == sy_bn_finalize followed by
    sy_bn_silu_apply, bit for bit: same fold order, same double-precision finalize, same apply arithmetic; with a residual and
    per-frame statistics segments."""
    g = torch.Generator().manual_seed(C + copies)
    N, H, W = 2 * nseg, 5, 7
    tdt = ops.TORCH_DTYPE[ops.dtype_code(dt)]
    dev = backend
    y = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(tdt).float()
    res = torch.randn(N, C, H, W, generator=g).to(tdt).float()
    yv = View.alloc(N, H, W, C, dt, dev); yv.set_nchw(y.to(dev))
    rv = View.alloc(N, H, W, C + 8, dt, dev).slice(8, C); rv.set_nchw(res.to(dev))
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.2).to(dev)
    # replica arrays [nseg][copies][C] with the per-segment sums spread unevenly over the replicas
    per = N // nseg
    parts = torch.rand(nseg, copies, C, generator=g) + 0.1
    parts = parts / parts.sum(1, keepdim=True)
    ssum = torch.stack([y[s * per:(s + 1) * per].sum((0, 2, 3)) for s in range(nseg)])[:, None, :] * parts
    ssq = torch.stack([(y[s * per:(s + 1) * per] ** 2).sum((0, 2, 3)) for s in range(nseg)])[:, None, :] * parts
    ssum, ssq = ssum.reshape(-1).contiguous().to(dev), ssq.reshape(-1).contiguous().to(dev)
    count = per * H * W
    a = [torch.empty(nseg * C, device=dev) for _ in range(4)]
    b = [torch.empty(nseg * C, device=dev) for _ in range(4)]
    o1, o2 = View.alloc(N, H, W, C, dt, dev), View.alloc(N, H, W, C + 16, dt, dev, zero=True).slice(16, C)
    ops.bn_finalize(ssum, ssq, count, gamma, beta, 1e-3, 0.03, None, None, *a, nseg=nseg)
    ops.bn_silu_apply(yv, a[0], a[1], o1, res=rv, nseg=nseg)
    ops.bn_finalize_apply(ssum, ssq, count, gamma, beta, 1e-3, *b, yv, o2, res=rv, nseg=nseg)
    for t1, t2 in zip(a, b):
        assert torch.equal(t1, t2)
    assert torch.equal(o1.nchw(), o2.nchw())
    assert float(o2.buf[..., :16].float().abs().max()) == 0.0
    ref = torch.cat([F.silu(F.batch_norm(y[s * per:(s + 1) * per], None, None, gamma.cpu(), beta.cpu(), True, 0.03, 1e-3))
                     for s in range(nseg)]) + res
    assert _rel(o2.nchw().cpu(), ref) < (2e-2 if dt == "bf16" else 2e-3 if dt == "fp16" else 1e-5)


def test_bn_running_update_batched(backend):
    """sy_bn_running_update: several modules in one launch; a module called twice applies both momentum updates in
    call order (dfp_pafpn.py:120-165 calls the shared backbone per frame, current frame first)."""
    import torch.nn as nn
    g = torch.Generator().manual_seed(5)
    mods, calls, ref = [], [], []
    for C, ncalls, copies in ((40, 2, 3), (8, 1, 1), (72, 2, 8)):
        bn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.03)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.1); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
        rbn = nn.BatchNorm2d(C, eps=1e-3, momentum=0.03).train()
        rbn.load_state_dict(bn.state_dict())
        bn = bn.to(backend)
        cl = []
        for j in range(ncalls):
            y = torch.randn(2 + j, C, 5, 7, generator=g) * (1.0 + j) + 0.3 * j
            rbn(y)
            parts = y.permute(1, 0, 2, 3).reshape(C, -1).chunk(copies, dim=1)      # replicas hold disjoint partial sums
            parts = list(parts) + [torch.zeros(C, 0)] * (copies - len(parts))
            ssum = torch.stack([p.sum(1) for p in parts]).contiguous().to(backend)
            ssq = torch.stack([(p ** 2).sum(1) for p in parts]).contiguous().to(backend)
            cl.append((ssum, ssq, y.numel() // C))
        mods.append((bn, cl)); ref.append(rbn)
    tab = ops.BnRunningTable(mods, torch.device(backend) if not isinstance(backend, torch.device) else backend, count_batches=True)
    assert tab.valid()
    tab.run()
    for (bn, _), rbn in zip(mods, ref):
        assert _rel(bn.running_mean.cpu(), rbn.running_mean) < 1e-5
        assert _rel(bn.running_var.cpu(), rbn.running_var) < 1e-5
        # num_batches_tracked += one per training-mode call, inside the same launch (nn.BatchNorm2d.forward does it on the host)
        assert int(bn.num_batches_tracked) == int(rbn.num_batches_tracked) and int(rbn.num_batches_tracked) in (1, 2)
    tab.run()
    assert [int(bn.num_batches_tracked) for bn, _ in mods] == [4, 2, 4]
    # without count_batches the counters are left alone (a caller that counts on the host)
    tab2 = ops.BnRunningTable(mods, torch.device(backend) if not isinstance(backend, torch.device) else backend)
    tab2.run()
    assert [int(bn.num_batches_tracked) for bn, _ in mods] == [4, 2, 4]


@pytest.mark.parametrize("shape,sl", [((7, 96), (16, 80)), ((5, 40), (3, 11)), ((1000,), None), ((33, 8), (0, 8)), ((4, 6, 24), None)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_zero_rows(backend, shape, sl, dt):
    """ops.zero (sy_zero_rows): dense tensors and row-strided column ranges are cleared, their surroundings untouched — what the
    training plan uses for its arenas and for the unwritten channel ranges of a gradient buffer (16-byte and 4-byte store paths)."""
    if dt == torch.bfloat16 and sl is not None and ((sl[1] - sl[0]) % 2 or sl[0] % 2):
        pytest.skip("2-byte elements: ranges are whole 4-byte words on the training path")
    full = torch.randn(shape).to(dt).to(backend) + 3.0
    keep = full.clone()
    t = full if sl is None else full[:, sl[0]:sl[1]]
    ops.zero(t)
    assert float(t.float().abs().max()) == 0.0
    if sl is not None:
        assert torch.equal(full[:, :sl[0]], keep[:, :sl[0]]) and torch.equal(full[:, sl[1]:], keep[:, sl[1]:])


def _random_preds(B, A, nc, seed, img=(600.0, 960.0), tie_free=True):
    g = torch.Generator().manual_seed(seed)
    p = torch.zeros(B, A, 5 + nc)
    p[..., 0] = torch.rand(B, A, generator=g) * img[1]
    p[..., 1] = torch.rand(B, A, generator=g) * img[0]
    p[..., 2] = torch.rand(B, A, generator=g) * 200 + 8
    p[..., 3] = torch.rand(B, A, generator=g) * 200 + 8
    p[..., 4] = torch.rand(B, A, generator=g)
    p[..., 5:] = torch.rand(B, A, nc, generator=g)
    if tie_free:
        p[..., 4] = p[..., 4] * 0.5 + 0.25 + torch.arange(A)[None] * 2.0 ** -20
    return p


@pytest.mark.parametrize("A", [0 + 1, 64, 257, 1500])
def test_postprocess_matches_oracle_bit_exact(backend, A):
    nc = 8
    pred = _random_preds(2, A, nc, seed=A)
    pred[1, :, 4] *= 0.02                       # second image: most anchors fall under the threshold
    det, idx, cnt = ops.postprocess(pred.to(backend), nc, 0.01, 0.65)
    ref = O.postprocess(pred, nc, 0.01, 0.65)
    for i, (rdet, ridx) in enumerate(ref):
        n = int(cnt[i])
        assert n == ridx.numel()
        assert np.array_equal(idx[i, :n].cpu().numpy(), ridx.numpy().astype(np.int32))     # order included
        assert torch.equal(det[i, :n].cpu(), rdet)


def test_postprocess_empty_and_ties(backend):
    nc = 8
    pred = _random_preds(1, 100, nc, seed=9)
    pred[..., 4] = 0.0                          # nothing survives the confidence filter
    det, idx, cnt = ops.postprocess(pred.to(backend), nc, 0.01, 0.65)
    assert int(cnt[0]) == 0
    # exact score ties: stable order = ascending anchor index
    pred = _random_preds(1, 200, nc, seed=10, tie_free=False)
    pred[..., 4] = 0.5
    pred[..., 5:] = 0.0
    pred[..., 5] = 0.8
    det, idx, cnt = ops.postprocess(pred.to(backend), nc, 0.01, 0.65)
    rdet, ridx = O.postprocess(pred, nc, 0.01, 0.65)[0]
    n = int(cnt[0])
    assert n == ridx.numel() and np.array_equal(idx[0, :n].cpu().numpy(), ridx.numpy().astype(np.int32))


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_pack_weights_matches_host_packing(backend, dt):
    """sy_pack_weights (one launch, all layers) against the host-side packers used by the inference cache."""
    from streamyolo_amd import _lib
    from streamyolo_amd.model.packing import pack_conv_weight, pack_conv_weight_frag
    code = ops.dtype_code(dt)
    tdt = ops.TORCH_DTYPE[code]
    g = torch.Generator().manual_seed(11)
    shapes = [(40, 32, 3), (64, 64, 1), (24, 12, 3), (5, 48, 1), (96, 80, 3)]
    rows, keep, want = [], [], []
    for co, ci, k in shapes:
        w = torch.randn(co, ci, k, k, generator=g)
        CI = 16 if ci == 12 else ci
        p = pack_conv_weight(w, code, pad_cin_to=CI)
        pt = pack_conv_weight(w, code, transpose=True, pad_cin_to=CI)
        f, ft = pack_conv_weight_frag(p, k), pack_conv_weight_frag(pt, k)
        wd = w.to(backend)
        bufs = [torch.zeros(t.numel(), dtype=tdt, device=backend) if t is not None else None for t in (p, pt, f, ft)]
        e = _lib.PackEntry()
        e.w = wd.data_ptr()
        e.packed, e.packed_t, e.frag, e.frag_t = [None if b is None else b.data_ptr() for b in bufs]
        e.co_n, e.ci_n, e.taps, e.r0, e.R, e.R_t, e.CI, e.dtype = co, ci, k * k, 0, co, co, CI, code
        rows.append(e); keep.append((wd, bufs)); want.append((p, pt, f, ft))
    # a stacked pair (MergedConv: two modules' rows r0 = 0 / 64 of ONE set of layouts, R = R_t = 128) and a 3x3 whose rows start at
    # r0 = 8: full 32 x 32 tiles off the fast path's row origin
    for parts, ci, k in (((64, 64), 64, 3), ((8, 32), 32, 3)):
        ws = [torch.randn(co, ci, k, k, generator=g) for co in parts]
        wcat = torch.cat(ws, 0)
        R = wcat.shape[0]
        p = pack_conv_weight(wcat, code)
        pt = pack_conv_weight(wcat, code, transpose=True)
        f = pack_conv_weight_frag(p, k)
        ft = pack_conv_weight_frag(pt, k) if R % 32 == 0 else None
        bufs = [torch.zeros(t.numel(), dtype=tdt, device=backend) if t is not None else None for t in (p, pt, f, ft)]
        r0 = 0
        for w in ws:
            wd = w.to(backend)
            e = _lib.PackEntry()
            e.w = wd.data_ptr()
            e.packed, e.packed_t, e.frag, e.frag_t = [None if b is None else b.data_ptr() for b in bufs]
            e.co_n, e.ci_n, e.taps, e.r0, e.R, e.R_t, e.CI, e.dtype = w.shape[0], ci, k * k, r0, R, R, ci, code
            rows.append(e); keep.append((wd, bufs if r0 == 0 else [None] * 4)); want.append((p, pt, f, ft) if r0 == 0 else (None,) * 4)
            r0 += w.shape[0]
    arr, total = _lib.pack_table(rows)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(backend)
    ops.check(_lib.lib().sy_pack_weights(table.data_ptr(), len(rows), total, ops.stream_of(table)), "sy_pack_weights")
    for (_, bufs), exp in zip(keep, want):
        for b, t in zip(bufs, exp):
            if t is not None:
                assert torch.equal(b.cpu().view(-1), t.reshape(-1))


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
def test_segmented_conv_stats_and_batchnorm(backend, dt):
    """stat_segments / nseg = 2: one launch per kernel over a 2-frame batch must equal two independent single-frame
    BatchNorm passes (separate batch statistics per frame, shared gamma / beta whose gradients add up)."""
    from streamyolo_amd.model.packing import pack_conv_weight
    g = torch.Generator().manual_seed(21)
    code = ops.dtype_code(dt)
    tdt = ops.TORCH_DTYPE[code]
    N, cin, C, H, W, copies = 2, 16, 24, 7, 9, 3            # 2 segments of N images each; 63 pixels per image (ragged tiles)
    x = (torch.randn(2 * N, cin, H, W, generator=g) * torch.tensor([1.0, 1.0, 3.0, 3.0]).view(4, 1, 1, 1)).to(tdt).float()
    w = (torch.randn(C, cin, 3, 3, generator=g) / 12.0).to(tdt).float()
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.2).requires_grad_(True)
    da = torch.randn(2 * N, C, H, W, generator=g).to(tdt).float()
    yref = F.conv2d(x, w, None, 1, 1)
    yq = yref.to(tdt).float().requires_grad_(True)
    a_ref = torch.cat([F.silu(F.batch_norm(yq[s * N:(s + 1) * N], None, None, gamma, beta, True, 0.03, 1e-3)) for s in range(2)])
    a_ref.backward(da)
    dev = backend
    xv = View.alloc(2 * N, H, W, cin, dt, dev); xv.set_nchw(x.to(dev))
    yv = View.alloc(2 * N, H, W, C, dt, dev)
    ssum = torch.zeros(2 * copies * C, device=dev); ssq = torch.zeros(2 * copies * C, device=dev)
    ops.conv2d(xv, pack_conv_weight(w, code).to(dev), yv, 3, 1, stats=(ssum, ssq), segments=2)
    for s in range(2):
        got = ssum.view(2, copies, C)[s].sum(0).cpu()
        assert _rel(got, yref[s * N:(s + 1) * N].sum((0, 2, 3))) < (2e-2 if dt == "bf16" else 1e-4)
    yv.set_nchw(yq.detach().to(dev))                       # continue from the exactly representable activations
    ssum.copy_(torch.stack([torch.cat([yq[s * N:(s + 1) * N].detach().sum((0, 2, 3)), torch.zeros((copies - 1) * C)]) for s in range(2)]).view(-1))
    ssq.copy_(torch.stack([torch.cat([(yq[s * N:(s + 1) * N].detach() ** 2).sum((0, 2, 3)), torch.zeros((copies - 1) * C)]) for s in range(2)]).view(-1))
    scale, shift, mean, invstd = [torch.empty(2 * C, device=dev) for _ in range(4)]
    gd, bd = gamma.detach().to(dev), beta.detach().to(dev)
    ops.bn_finalize(ssum, ssq, N * H * W, gd, bd, 1e-3, 0.03, None, None, scale, shift, mean, invstd, nseg=2)
    av = View.alloc(2 * N, H, W, C, dt, dev)
    ops.bn_silu_apply(yv, scale, shift, av, nseg=2)
    assert _rel(av.nchw().cpu(), a_ref.detach()) < (2e-2 if dt == "bf16" else 1e-5)
    dav = View.alloc(2 * N, H, W, C, dt, dev); dav.set_nchw(da.to(dev))
    sums = torch.zeros(2 * copies * 2 * C, device=dev)
    ops.bn_silu_bwd_reduce(yv, dav, scale, shift, mean, invstd, sums, nseg=2)
    dyv = View.alloc(2 * N, H, W, C, dt, dev)
    dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ops.bn_silu_bwd_apply(yv, dav, scale, shift, mean, invstd, gd, sums, dyv, dgam, dbet, nseg=2)
    assert _rel(dbet.cpu(), beta.grad) < 1e-4 and _rel(dgam.cpu(), gamma.grad) < 1e-4
    assert _rel(dyv.nchw().cpu(), yq.grad) < (3e-2 if dt == "bf16" else 1e-4)


# ---- property tests (hypothesis; NMS keep-set vs an independent greedy walk; decode invertibility) ----
def _greedy_nms_reference(pred, nc, conf, thr):
    """Independent restatement (numpy, one pass over score-sorted candidates) of yolox postprocess + torchvision
    batched_nms semantics: keep list of anchor indices in descending score order, suppress when IoU > thr (strict),
    class-aware, IoU = inter / (a + b - inter)."""
    p = pred.numpy().astype(np.float32)
    box = np.stack([p[:, 0] - p[:, 2] / 2, p[:, 1] - p[:, 3] / 2, p[:, 0] + p[:, 2] / 2, p[:, 1] + p[:, 3] / 2], 1)
    cls = p[:, 5:5 + nc].argmax(1)
    cconf = p[:, 5:5 + nc].max(1)
    score = (p[:, 4] * cconf).astype(np.float32)
    cand = np.nonzero(score >= np.float32(conf))[0]
    cand = cand[np.argsort(-score[cand], kind="stable")]
    keep = []
    for i in cand:
        ok = True
        for j in keep:
            if cls[j] != cls[i]:
                continue
            lt = np.maximum(box[i, :2], box[j, :2]); rb = np.minimum(box[i, 2:], box[j, 2:])
            wh = np.maximum(rb - lt, np.float32(0))
            inter = wh[0] * wh[1]
            a = (box[i, 2] - box[i, 0]) * (box[i, 3] - box[i, 1]); b = (box[j, 2] - box[j, 0]) * (box[j, 3] - box[j, 1])
            if inter / (a + b - inter) > np.float32(thr):
                ok = False
                break
        if ok:
            keep.append(int(i))
    return keep


def test_nms_keep_set_property_random_boxes(backend):
    """hypothesis-driven: for random box sets (clustered so that suppression actually happens) the device keep list
    equals the independent greedy walk — set AND order — and is idempotent: NMS of the kept boxes keeps all of them."""
    from hypothesis import given, settings, strategies as st, HealthCheck
    nc = 4

    @settings(max_examples=12, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(st.integers(0, 2 ** 31 - 1), st.sampled_from([1, 7, 64, 129, 400]), st.floats(0.3, 0.8))
    def run(seed, A, thr):
        g = torch.Generator().manual_seed(seed)
        centres = torch.rand(max(1, A // 6), 2, generator=g) * 200 + 50
        which = torch.randint(0, centres.shape[0], (A,), generator=g)
        p = torch.zeros(1, A, 5 + nc)
        p[0, :, 0:2] = centres[which] + torch.randn(A, 2, generator=g) * 4
        p[0, :, 2:4] = torch.rand(A, 2, generator=g) * 40 + 10
        p[0, :, 4] = torch.rand(A, generator=g) * 0.5 + 0.25 + torch.arange(A) * 2.0 ** -20      # tie-free
        p[0, :, 5:] = torch.rand(A, nc, generator=g)
        det, idx, cnt = ops.postprocess(p.to(backend), nc, 0.2, float(thr))
        n = int(cnt[0])
        got = idx[0, :n].cpu().tolist()
        assert got == _greedy_nms_reference(p[0], nc, 0.2, float(thr))
        if n:
            again, idx2, cnt2 = ops.postprocess(p[:, got].contiguous().to(backend), nc, 0.2, float(thr))
            assert int(cnt2[0]) == n and idx2[0, :n].cpu().tolist() == list(range(n))
    run()


@pytest.mark.gpu
@pytest.mark.parametrize("A", [0, 4096, 11850])
def test_nms_keep_set_at_survey_sizes(A):
    """SURVEY.md §8(d) NMS unit inputs at N in {0, 4096, 11850} (1 and 64 run above on both back ends): bit-exact keep
    lists, order included, against the oracle's torchvision-semantics restatement."""
    from streamyolo_amd import _lib
    _lib.use_library(_lib.DEFAULT_PATH)
    dev = torch.device("cuda:0")
    nc = 8
    pred = _random_preds(2, max(A, 1), nc, seed=100 + A)
    if A == 0:
        pred[..., 4] = 0.0
    det, idx, cnt = ops.postprocess(pred.to(dev), nc, 0.01, 0.65)
    for i, (rdet, ridx) in enumerate(O.postprocess(pred, nc, 0.01, 0.65)):
        n = int(cnt[i])
        assert n == ridx.numel() and np.array_equal(idx[i, :n].cpu().numpy(), ridx.numpy().astype(np.int32))
        assert torch.equal(det[i, :n].cpu(), rdet)


def test_decode_is_invertible(backend):
    """The decode epilogue of the prediction convs: (xy + grid) * stride, exp(wh) * stride, sigmoid(obj / cls).  Inverting
    it on the device output — raw = logit / log / (x / stride - grid) — must return the undecoded (EPI_LINEAR) output of
    the same launch configuration, for every level geometry of a 96 x 160 input."""
    from streamyolo_amd.ops import View, conv2d, EPI_DECODE, EPI_LINEAR
    g = torch.Generator().manual_seed(3)
    for (H, W, stride) in ((12, 20, 8), (6, 10, 16), (3, 5, 32)):
        B, cin, nch = 2, 32, 13
        x = View.alloc(B, H, W, cin, "fp32", backend)
        x.buf.copy_(torch.randn(B, H, W, cin, generator=g).to(backend) * 0.3)
        w = (torch.randn(5, cin, generator=g) * 0.2).to(backend)
        b = (torch.randn(5, generator=g) * 0.1).to(backend)
        outs = []
        for epi in (EPI_LINEAR, EPI_DECODE):
            out = torch.zeros(B, H * W, nch, device=backend)
            conv2d(x, w, None, 1, 1, None, b, epilogue=epi, dec_stride=stride, y_f32=True, y_ptr=out.data_ptr(), y_ld=nch,
                   y_bs=H * W * nch, cout=5)
            outs.append(out.cpu())
        raw, dec = outs
        gy, gx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        grid = torch.stack([gx, gy], -1).reshape(1, H * W, 2).float()
        inv = torch.cat([dec[..., 0:2] / stride - grid, (dec[..., 2:4] / stride).log(), torch.logit(dec[..., 4:5])], -1)
        assert (inv - raw[..., :5]).abs().max() < 2e-4 * max(1.0, float(raw[..., :5].abs().max()))

