"""tools/tune_in_situ.py --workload stream: the streaming plan's per-layer decisions tuned in the taped frame."""
import numpy as np
import torch


def tune_stream(a, model, cfg, dev):
    from streamyolo_amd import ops
    from streamyolo_amd.postprocess import postprocess_device
    from streamyolo_amd.utils.synth import synth_frames
    model.eval()
    frame = synth_frames(1, 600, 960, seed=2).to(dev)[:, 0:3].contiguous()
    plan = model._plans.inference(model.backbone, model.head, "on_pipe", frame, owner=model, split_k=a.dtype != "fp32")
    post = lambda out: postprocess_device(out, cfg.num_classes, 0.01, 0.65)        # noqa: E731

    def run():
        with torch.no_grad():
            return plan.run_stream_taped(frame, post=post, check_params=False)
    with torch.no_grad():
        # a trained checkpoint's candidate count (bench.py --candidates realistic): one constant on the objectness biases
        dec = plan.run_stream(frame, first=True).clone()
        obj, cls = dec[0, :, 4].double().clamp(1e-12, 1 - 1e-12), dec[0, :, 5:].double().max(dim=1).values
        logit = torch.log(obj / (1 - obj))
        lo, hi = -40.0, 0.0
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            lo, hi = (mid, hi) if float(((torch.sigmoid(logit + mid) * cls) >= 0.01).double().mean()) < 0.01 else (lo, mid)
        for conv in model.head.obj_preds:
            conv.bias.add_(0.5 * (lo + hi))
        plan.run_stream(frame, first=True)
    for _ in range(5):
        run()
    torch.cuda.synchronize()

    def measure():
        """us per frame: best of three timings of --frames taped frames"""
        for _ in range(4):
            run()                                                        # (re-records the tape after a change)
        best = float("inf")
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.frames):
                run()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / a.frames * 1e3)
        return best

    def apply(group, what, value):
        for op in group:
            op._tiles[what] = value
        plan._stream_tape = None

    convs = [op for op in plan.ops if op.kind == "conv"]
    for op in convs:                                                     # make every decision exist
        op.tile("fwd")
    plan._ensure_tuned()
    groups = {}
    for op in convs:
        sig = (op.x.H, op.x.W, op.x.C, op.y.C, op.k, op.stride, op.res is not None, op.pre_op is not None, op.fused_into is not None)
        groups.setdefault(sig, []).append(op)
    print("%d shape groups over %d convolutions" % (len(groups), len(convs)))
    base = best = measure()
    print("frame %.1f us with the per-kernel tuner's choices" % base)
    gain = a.gain * 1e3 if a.gain < 0.05 else 4.0                        # us (the default --gain is the training step's, in ms)
    kept = []
    for sig, group in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        H, W, cin, cout, k, stride, has_res, has_pre, is_pre = sig
        op0 = group[0]
        trials = []                                                      # (what, value)
        if has_pre and op0._tiles.get("fuse"):
            trials.append(("fuse", False))                               # (its own 3x3 tile does not run while the pair is fused)
        elif k == 3 and stride == 1 and cin % 32 == 0 and "splitk" in op0._tiles:
            cur = tuple(op0._tiles["splitk"])
            for t in (117, 118, 112, 113, 111, 106):
                if (1, t) != cur and not (t == 106 and cout % 32):
                    trials.append(("splitk", (1, t)))
            if H * W <= 6000 and cin >= 128:
                trials += [("splitk", (2, t)) for t in (117, 112) if (2, t) != cur]
            if has_pre and "fuse" in op0._tiles:
                trials.append(("fuse", not op0._tiles["fuse"]))
        elif k == 3 and stride == 2 and cin % 32 == 0 and "splitk" in op0._tiles:
            cur = tuple(op0._tiles["splitk"])
            trials += [("splitk", (1, t)) for t in (110, 105) if (1, t) != cur]
        elif k == 1 and cin in (64, 128, 256, 512, 1024, 2048) and not is_pre:
            cur = int(op0._tiles["fwd"])
            for t in (121, 122, 123, 124):
                if t != cur and not (t == 122 and cout > 64) and not (t == 123 and cin > 256) and not (t == 124 and cin < 1024) \
                        and not (t in (122, 123) and cin > 512):
                    trials.append(("fwd", t))
        for what, c in trials:
            won = op0._tiles[what]
            try:
                apply(group, what, c)
                t1 = measure()
            except ops._lib.HipLibraryError:
                apply(group, what, won)
                continue
            if t1 < best - gain:
                apply(group, what, won)
                t0 = measure()
                apply(group, what, c)
                t2 = measure()
                if t2 < t0 - gain:
                    print("  %3dx%-3d %4d->%-4d k%d s%d x%-2d  %-6s %s -> %s   %.1f / %.1f vs %.1f us" %
                          (H, W, cin, cout, k, stride, len(group), what, won, c, t1, t2, t0))
                    best = min(t1, t2)
                    kept.append((sig, what, won, c))
                    continue
            apply(group, what, won)
    final = measure()
    print("frame %.1f us with %d in-situ choices (was %.1f)" % (final, len(kept), base))
    if a.save and kept and final < base - gain:
        code = ops.dtype_code(a.dtype)
        for sig, what, old, c in kept:
            H, W, cin, cout, k, stride = sig[:6]
            op = groups[sig][0]
            if what == "fwd":
                ops._tile_cache[(ops.CONV_FWD, code, 1, H, W, cin, cout, k, stride, False, str(dev))] = int(c)
            elif what == "splitk":
                key = ("splitk", code, 1, H, W, cin, cout, int(op.tile("fwd")), str(dev)) + (() if stride == 1 else (stride,))
                ops._tile_cache[key] = c[0] * 1000 + c[1] if tuple(c) != (1, int(op.tile("fwd"))) else 1
            else:
                pre = op.pre_op
                sk = op._tiles.get("splitk", (1, op.tile("fwd")))
                t3 = sk[1] if sk[0] == 1 else op.tile("fwd")
                ops._tile_cache[("bnk", code, 1, H, W, pre.x.C, cin, cout, bool(op.res is not None), int(pre.tile("fwd")), int(t3), str(dev))] = 1 if c else 0
        ops._tune_store.dirty = True
        ops.save_tuned()
        print("saved to the tuner cache")
