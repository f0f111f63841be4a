"""Device input pipeline: sy_frames_u8_pack / sy_resize_bilinear_nchw against golden vectors
minted from the reference's own `preproc` / `_mirror` / `DoubleTrainTransform` / `Exp.preprocess`
(oracle/make_golden_input.py) and against the oracle restatement on larger seeded inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import input_oracle as IO
from streamyolo_amd import ops
from streamyolo_amd.data import FramePairsU8, preprocess
from streamyolo_amd.ops import View


@pytest.fixture(scope="module")
def gold(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "input_pipeline.npz")))


def test_oracle_matches_reference_golden(gold):
    g = gold
    H, W = g["a_ref"].shape[2:]
    assert np.array_equal(IO.pair_tensor(g["a_cur"], g["a_sup"], (H, W), 1, g["a_mirror"]).numpy(), g["a_ref"])
    assert np.array_equal(IO.pair_tensor(g["b_cur"], None, (H, W)).numpy(), g["b_ref"])
    assert np.array_equal(IO.pair_tensor(g["c_cur"], None, (H, W), 2).numpy(), g["c_ref"])
    assert np.array_equal(IO.pair_tensor(g["a_cur"][:1], g["a_sup"][:1], (H, W), 1, g["d_flag"]).numpy()[0], g["d_img"])
    t = (torch.tensor([[[1.0, 10.0, 6.0, 8.0, 4.0]]]).repeat(2, 1, 1), torch.tensor([[[1.0, 11.0, 7.0, 8.0, 4.0]]]).repeat(2, 1, 1))
    x, t = IO.exp_preprocess(torch.from_numpy(g["a_ref"]).clone(), t, tuple(g["e_tsize"]), (H, W))
    assert np.array_equal(x.numpy(), g["e_ref"]) and np.array_equal(t[0].numpy(), g["e_t0"]) and np.array_equal(t[1].numpy(), g["e_t1"])


def _dev(a, backend):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(backend)


def test_frames_to_nchw_bit_exact_vs_reference(backend, gold):
    """uint8 HWC -> the reference's fp32 [B,6,H,W]: integer-valued, so bit-exact (mirror, letterbox, 2x decimation)."""
    g = gold
    H, W = g["a_ref"].shape[2:]
    a = FramePairsU8(_dev(g["a_cur"], backend), _dev(g["a_sup"], backend), (H, W), mirror=_dev(g["a_mirror"], backend))
    assert tuple(a.shape) == (2, 6, H, W)
    assert np.array_equal(a.to_nchw().cpu().numpy(), g["a_ref"])
    b = FramePairsU8(_dev(g["b_cur"], backend), None, (H, W))
    assert np.array_equal(b.to_nchw().cpu().numpy(), g["b_ref"])
    c = FramePairsU8(_dev(g["c_cur"], backend), None, (H, W), decimate=2)
    assert np.array_equal(c.to_nchw().cpu().numpy(), g["c_ref"])
    d = FramePairsU8(_dev(g["a_cur"][:1], backend), _dev(g["a_sup"][:1], backend), (H, W), mirror=_dev(g["d_flag"], backend))
    assert np.array_equal(d.to_nchw().cpu().numpy()[0], g["d_img"])


def test_exp_preprocess_matches_reference(backend, gold):
    """Exp.preprocess: HIP bilinear (torch align_corners=False arithmetic) within 1e-5 relative of the reference's
    F.interpolate output (fp32, 0-255 range), targets scaled in place exactly."""
    g = gold
    H, W = g["a_ref"].shape[2:]
    tsize = tuple(int(v) for v in g["e_tsize"])
    t = (torch.tensor([[[1.0, 10.0, 6.0, 8.0, 4.0]]]).repeat(2, 1, 1).to(backend),
         torch.tensor([[[1.0, 11.0, 7.0, 8.0, 4.0]]]).repeat(2, 1, 1).to(backend))
    x, t2 = preprocess(_dev(g["a_ref"], backend), t, tsize, (H, W))
    assert tuple(x.shape) == g["e_ref"].shape
    err = float((x.cpu() - torch.from_numpy(g["e_ref"])).abs().max())
    assert err <= 1e-5 * 255.0, err
    assert np.array_equal(t2[0].cpu().numpy(), g["e_t0"]) and np.array_equal(t2[1].cpu().numpy(), g["e_t1"])
    # unchanged size: identity, same objects back (cfg :164)
    x0 = _dev(g["a_ref"], backend)
    y0, _ = preprocess(x0, t, (H, W), (H, W))
    assert y0 is x0
    # the uint8 route defers the resize into the pack kernel and must agree with resizing the fp32 tensor
    f = FramePairsU8(_dev(g["a_cur"], backend), _dev(g["a_sup"], backend), (H, W), mirror=_dev(g["a_mirror"], backend))
    t3 = (torch.zeros(2, 1, 5, device=backend), torch.zeros(2, 1, 5, device=backend))
    f2, _ = preprocess(f, t3, tsize, (H, W))
    assert float((f2.to_nchw().cpu() - torch.from_numpy(g["e_ref"])).abs().max()) <= 1e-5 * 255.0


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
@pytest.mark.parametrize("case", ["same", "letterbox_mirror", "decimate", "resize"])
def test_frames_to_focus_matches_oracle(backend, dt, case):
    """Focus-packed stem operand straight from uint8 == oracle pipeline -> Focus slicing -> storage dtype."""
    rng = np.random.RandomState(11)
    B, H, W = 2, 20, 36
    hs, ws, dec, mirror, tsize = H, W, 1, None, (H, W)
    if case == "letterbox_mirror":
        hs, mirror = H - 4, np.array([0, 1], dtype=np.uint8)
    elif case == "decimate":
        hs, ws, dec = 2 * H, 2 * W, 2
    elif case == "resize":
        tsize, mirror = (28, 44), np.array([1, 1], dtype=np.uint8)
    cur = rng.randint(0, 256, (B, hs, ws, 3)).astype(np.uint8)
    sup = rng.randint(0, 256, (B, hs, ws, 3)).astype(np.uint8)
    ref = IO.pair_tensor(cur, sup, (H, W), dec, mirror)
    tz = (torch.zeros(B, 1, 5), torch.zeros(B, 1, 5))
    ref, _ = IO.exp_preprocess(ref, tz, tsize, (H, W))
    f = FramePairsU8(_dev(cur, backend), _dev(sup, backend), (H, W), dec, _dev(mirror, backend), tsize)
    vc = View.alloc(B, tsize[0] // 2, tsize[1] // 2, 16, dt, backend)
    vs = View.alloc(B, tsize[0] // 2, tsize[1] // 2, 16, dt, backend)
    f.pack_focus(vc, vs)
    tdt = ops.TORCH_DTYPE[ops.dtype_code(dt)]
    for v, c0 in ((vc, 0), (vs, 3)):
        got = v.buf.cpu().float()
        want = IO.focus_pack(ref[:, c0:c0 + 3])
        assert torch.equal(got[..., 12:], torch.zeros_like(got[..., 12:]))
        if case == "resize":
            assert float((got[..., :12] - want.to(tdt).float()).abs().max()) <= (2.0 if dt == "bf16" else 1e-5 * 255.0)
        else:
            assert torch.equal(got[..., :12], want.to(tdt).float())


def test_unsupported_ratio_is_an_error(backend):
    """Ratios that would need cv2's interpolation tables are refused, not approximated."""
    from streamyolo_amd._lib import HipLibraryError
    cur = torch.zeros((1, 30, 50, 3), dtype=torch.uint8, device=backend)
    with pytest.raises(HipLibraryError):
        FramePairsU8(cur, None, (20, 36)).to_nchw()
    with pytest.raises(HipLibraryError):
        FramePairsU8(cur, None, (20, 36), decimate=3).to_nchw()


def test_model_accepts_uint8_frames(backend):
    """A FramePairsU8 through the eval facade == the reference-style fp32 tensor through it (same plan, same kernels)."""
    import streamyolo_amd as sy
    from oracle import streamyolo_oracle as O
    from streamyolo_amd.utils.synth import synth_state_dict, load_bn_stats
    cfg = O.OracleConfig.named("nano")
    model = sy.build_model("nano")
    model.load_state_dict(synth_state_dict(O.param_shapes(cfg), seed=0, bn_stats=load_bn_stats("nano")), strict=True)
    model = model.to(backend).eval().set_compute_dtype("fp32")
    rng = np.random.RandomState(5)
    B, H, W = 1, 64, 96
    cur = rng.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    sup = rng.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    f = FramePairsU8(_dev(cur, backend), _dev(sup, backend), (H, W))
    with torch.no_grad():
        a = model(f)
        b = model(IO.pair_tensor(cur, sup, (H, W)).to(backend))
        on_u8, _ = model(FramePairsU8(_dev(cur, backend), None, (H, W)), mode="on_pipe")
        on_f, _ = model(IO.pair_tensor(cur, None, (H, W)).to(backend), mode="on_pipe")
    assert torch.equal(a, b) and torch.equal(on_u8, on_f)


@pytest.mark.gpu
def test_device_prefetcher_hands_over_uint8_batches():
    """DevicePrefetcher (the reference's DataPrefetcher for uint8 frames): side-stream H2D, wait_stream + record_stream
    hand-over, batches in loader order, (None, None) when the loader is exhausted."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from streamyolo_amd import _lib
    from streamyolo_amd.data import DevicePrefetcher
    _lib.use_library(_lib.DEFAULT_PATH)
    rng = np.random.RandomState(2)
    H, W = 32, 64
    batches = []
    for i in range(3):
        cur = torch.from_numpy(rng.randint(0, 256, (2, H, W, 3)).astype(np.uint8)).pin_memory()
        sup = torch.from_numpy(rng.randint(0, 256, (2, H, W, 3)).astype(np.uint8)).pin_memory()
        lab = torch.full((2, 4, 5), float(i))
        batches.append(((cur, sup, np.array([i % 2, 1], dtype=np.uint8)), (lab, lab + 1), None, None))
    pf = DevicePrefetcher(batches, canvas=(H, W), device="cuda:0")
    for i in range(3):
        inp, tgt = pf.next()
        want = IO.pair_tensor(batches[i][0][0].numpy(), batches[i][0][1].numpy(), (H, W), 1, batches[i][0][2])
        assert torch.equal(inp.to_nchw().cpu(), want)
        assert tgt[0].is_cuda and float(tgt[0][0, 0, 0]) == float(i) and float(tgt[1][0, 0, 0]) == float(i + 1)
    inp, tgt = pf.next()
    assert inp is None and tgt is None


def test_cv2_linear_restatement_properties():
    """oracle.input_oracle.cv2_resize_linear_u8 (OpenCV's fixed-point INTER_LINEAR, restated — no cv2 binary here): the
    properties any correct restatement has — same size is the identity, constants stay constant, results stay inside the
    source range, a horizontal ramp stays monotone, exact 2x takes the INTER_AREA fast path, and the tap weights of every
    output pixel sum to 2048."""
    g = np.random.default_rng(3)
    img = g.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(IO.cv2_resize_linear_u8(img, (53, 37)), img)
    const = np.full((30, 44, 3), 171, dtype=np.uint8)
    assert np.all(IO.cv2_resize_linear_u8(const, (33, 21)) == 171)
    for dsize in ((40, 25), (80, 60), (17, 9)):
        out = IO.cv2_resize_linear_u8(img, dsize)
        assert out.shape == (dsize[1], dsize[0], 3) and out.min() >= img.min() and out.max() <= img.max()
    ramp = np.repeat(np.linspace(0, 255, 64).astype(np.uint8)[None, :, None], 20, 0).repeat(3, 2)
    up = IO.cv2_resize_linear_u8(ramp, (150, 31)).astype(np.int32)
    assert np.all(np.diff(up[:, :, 0], axis=1) >= 0)
    big = g.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    assert np.array_equal(IO.cv2_resize_linear_u8(big, (20, 12)), IO.load_time_resize(big, 2))


@pytest.mark.parametrize("hs,ws,canvas", [(45, 80, (40, 64)), (27, 64, (40, 64)), (50, 41, (40, 64)), (80, 128, (40, 64)),
                                          (40, 64, (40, 64))])
def test_general_ratio_letterbox_matches_the_oracle_bit_exact(backend, hs, ws, canvas):
    """decimate=0: a camera size that is neither the canvas nor twice it — `preproc`'s r = min(H / h, W / w) followed by
    OpenCV's fixed-point bilinear resize, mirror, letterbox on 114 — the device kernel against the numpy restatement,
    integer arithmetic on both sides: every pixel equal, for shrinking, enlarging, exact-2x and same-size cameras."""
    g = torch.Generator().manual_seed(hs * 131 + ws)
    cur = torch.randint(0, 256, (2, hs, ws, 3), generator=g, dtype=torch.uint8)
    sup = torch.randint(0, 256, (2, hs, ws, 3), generator=g, dtype=torch.uint8)
    mirror = torch.tensor([1, 0], dtype=torch.uint8)
    want = IO.pair_tensor(cur.numpy(), sup.numpy(), canvas, decimate=0, mirror=mirror.numpy())
    got = FramePairsU8(cur.to(backend), sup.to(backend), canvas, decimate=0, mirror=mirror.to(backend)).to_nchw()
    assert got.shape == want.shape and torch.equal(got.cpu(), want)


@pytest.mark.parametrize("hs,ws,canvas", [(45, 80, (30, 52)), (40, 64, (52, 84)), (61, 97, (20, 32)), (64, 96, (32, 48))])
def test_streaming_per_axis_stretch_matches_the_oracle_bit_exact(backend, hs, ws, canvas):
    """decimate=-1: the streaming detector's cv2.resize(img, (w_img, h_img)) — each axis stretched on its own to
    exactly the canvas, no letterbox (sAP/streamyolo/streamyolo_det.py:176-178) — shrinking, enlarging (the vertical border
    rows keep their fraction: OpenCV clamps only the row indices there) and the exact-2x shortcut."""
    g = torch.Generator().manual_seed(hs * 17 + ws)
    cur = torch.randint(0, 256, (2, hs, ws, 3), generator=g, dtype=torch.uint8)
    want = np.stack([IO.preproc(IO.cv2_resize_linear_u8(f, (canvas[1], canvas[0])), canvas) for f in cur.numpy()])
    got = FramePairsU8(cur.to(backend), None, canvas, decimate=-1).to_nchw()
    assert tuple(got.shape) == want.shape and np.array_equal(got.cpu().numpy(), want)


def test_vertical_border_rows_keep_their_fraction():
    """OpenCV's vertical pass clamps the two ROW INDICES at the top / bottom border but keeps the fraction (horizontally it zeroes
    the fraction): when enlarging, border rows go through two separate `>> 16` floors with b0 + b1 = 2048 and may land 1 LSB
    below the source value, never above; interior behaviour is unchanged (constant images stay within 1 LSB)."""
    img = np.full((5, 7, 3), 201, dtype=np.uint8)
    out = IO.cv2_resize_linear_u8(img, (21, 15)).astype(np.int32)
    assert out.max() <= 201 and out.min() >= 200
    assert np.all(out[2:-2] == 201)                                  # rows whose two taps are distinct rows: exact
