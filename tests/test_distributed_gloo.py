"""N>1 path on CPU: world_size-2 `gloo` processes (one per "GPU"), SIMT-emulator build of the kernels.

Checks (a) TrainStep's single all-reduce over the flat gradient arena and (b) the drop-in path under
torch's DistributedDataParallel wrapper (what the reference trainer does, double_trainer.py:171):
after one step both ranks hold the MEAN of the two ranks' gradients, and `.grad` of shared
backbone weights (used by both frames) fires its DDP hook exactly once."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, emu_lib, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SY_EMU_THREADS="4")
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import streamyolo_amd as sy
    from streamyolo_amd import _lib
    from streamyolo_amd.train_engine import TrainStep
    from oracle import streamyolo_oracle as O
    from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels
    _lib.use_library(emu_lib)
    cfg = O.OracleConfig.named("nano")
    sd = synth_state_dict(O.param_shapes(cfg), seed=0)

    def fresh():
        m = sy.build_model("nano")
        m.load_state_dict(sd, strict=True)
        m.train().set_compute_dtype("fp32")
        m.head.use_l1 = True
        return m
    x = synth_frames(1, 64, 96, seed=20 + rank)
    lab, sup = synth_labels(1, 64, 96, 8, num_gt=4, seed=30 + rank)

    # (a) fast path: all-reduce over the flat arena.  Step 1 runs the wrappers, step 2 records the launch tape
    # (both reduce after backward); step 3 replays the tape and starts each gradient bucket's all-reduce from
    # inside the backward walk (small buckets here so that several are in flight).
    from streamyolo_amd import train_engine
    train_engine.TrainPlan.BUCKET_BYTES = 256 << 10
    train_engine.TrainPlan.FIRST_BUCKET_BYTES = 32 << 10
    m = fresh()
    st = TrainStep(m, world_size=world, process_group=dist)
    fasts = []
    for _ in range(3):
        m.load_state_dict(sd, strict=True)
        st.step(x, (lab, sup))
        fasts.append((st.plan.arena.clone(), st.last_overlapped))
    fast = fasts[0][0]
    assert len(st.plan.buckets) >= 3 and fasts[0][1] == 0 and fasts[2][1] == len(st.plan.buckets)
    # bucket layout: contiguous cover of the arena in parameter order; the FRONT bucket — stem / dark2, final at the very end of
    # the backward walk, the only all-reduce nothing hides — is the small one; readiness moves towards the end of the walk as the
    # buckets move towards the front of the arena
    bk = st.plan.buckets
    assert bk[0][0] == 0 and bk[-1][1] == st.plan.arena.numel() and all(bk[i][1] == bk[i + 1][0] for i in range(len(bk) - 1))
    sizes = [(hi - lo) * 4 for lo, hi, _ in bk]
    assert sizes[0] < train_engine.TrainPlan.BUCKET_BYTES and max(sizes[1:-1]) >= train_engine.TrainPlan.BUCKET_BYTES
    assert sizes[0] <= min(sizes[1:-1])
    ready = [r for _, _, r in bk]
    assert ready[0] == max(ready) and all(ready[i] >= ready[i + 1] for i in range(len(ready) - 1))
    assert st.exposed_allreduce() is not None and st.exposed_allreduce() >= 0.0

    # local (un-reduced) gradients of this rank, for the expected mean
    m1 = fresh()
    s1 = TrainStep(m1)
    s1.step(x, (lab, sup))
    local = s1.plan.arena.clone()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean = sum(gathered) / world
    err_fast = max(float((f - mean).abs().max() / mean.abs().max()) for f, _ in fasts)

    # (a') opt-in gradient compression: buckets rounded to bf16 for the exchange (half the bytes), widened back, averaged
    m3 = fresh()
    st16 = TrainStep(m3, world_size=world, process_group=dist, grad_comm_dtype="bf16")
    errs16 = []
    for _ in range(3):
        m3.load_state_dict(sd, strict=True)
        st16.step(x, (lab, sup))
        errs16.append(float((st16.plan.arena - mean).abs().max() / mean.abs().max()))
    assert st16._comm16 is not None and st16._comm16.dtype == torch.bfloat16 and st16.last_overlapped == len(st16.plan.buckets)
    err_bf16 = max(errs16)

    # (b) drop-in path under DDP
    m2 = fresh()
    ddp = torch.nn.parallel.DistributedDataParallel(m2, broadcast_buffers=False)
    out = ddp(x, (lab, sup))
    out["total_loss"].backward()
    # `mean` is in ARENA order (the parts of stacked sibling convolutions sit side by side there): same order for the DDP grads
    name_of = {id(p): n for n, p in m1.named_parameters()}
    named2 = dict(m2.named_parameters())
    flat = torch.cat([named2[name_of[id(p)]].grad.reshape(-1) for p in s1.plan.params])
    err_ddp = float((flat - mean).abs().max() / mean.abs().max())
    torch.save({"err_fast": err_fast, "err_ddp": err_ddp, "err_bf16": err_bf16, "nonzero": bool(mean.abs().max() > 0)},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_ddp(tmp_path):
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "streamyolo_amd", "csrc"), "-j8", "emu"], check=True)
    emu_lib = os.path.join(ROOT, "tests", "emu", "_build", "libstreamyolo_emu.so")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, emu_lib, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        res = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r))
        assert res["nonzero"]
        assert res["err_fast"] < 1e-5, res
        assert res["err_ddp"] < 1e-5, res
        assert 1e-6 < res["err_bf16"] < 1.5e-2, res            # bf16 rounding of each rank's gradient (2^-9 relative), not more


@pytest.mark.gpu
def test_bucketed_allreduce_streams_on_gpu_single_rank_rccl(golden_dir):
    """The overlapped-bucket machinery on real streams with RCCL: a 1-rank `nccl` group makes every all-reduce the
    identity, so with world_size=2 declared to TrainStep the taped step must return exactly half of the local
    gradients — any missing stream dependency (communication stream vs main / weight-gradient stream) or a bucket
    reduced before its gradients are final shows up as a mismatch."""
    import numpy as np
    import streamyolo_amd as sy
    from streamyolo_amd import _lib, train_engine
    from streamyolo_amd.train_engine import TrainStep
    from oracle import streamyolo_oracle as O
    from streamyolo_amd.utils.synth import synth_state_dict, synth_frames, synth_labels
    _lib.use_library(_lib.DEFAULT_PATH)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda:0")
        z = np.load(os.path.join(golden_dir, "s_train_2x160x256.npz"))
        B, H, W = [int(v) for v in z["shape"]]
        cfg = O.OracleConfig.named("s")
        sd = synth_state_dict(O.param_shapes(cfg), seed=0)
        x = synth_frames(B, H, W, seed=2).to(dev)
        lab, sup = synth_labels(B, H, W, cfg.num_classes, num_gt=6, seed=3)
        lab, sup = lab.to(dev), sup.to(dev)

        def fresh():
            m = sy.build_model("s")
            m.load_state_dict(sd, strict=True)
            m = m.to(dev).train().set_compute_dtype("fp32")
            m.head.use_l1 = True
            return m
        old = train_engine.TrainPlan.BUCKET_BYTES
        train_engine.TrainPlan.BUCKET_BYTES = 4 << 20
        try:
            m = fresh()
            ref = TrainStep(m)
            ref.step(x, (lab, sup))
            local = ref.plan.arena.clone()
            state0 = {k: v.clone() for k, v in m.state_dict().items()}
            m2 = fresh()
            st = TrainStep(m2, world_size=2, process_group=dist)
            for i in range(4):
                m2.load_state_dict(state0)
                st.step(x, (lab, sup))
                torch.cuda.synchronize()
                err = float((st.plan.arena - 0.5 * local).abs().max() / local.abs().max())
                assert err < 1e-4, (i, err)
            assert len(st.plan.buckets) >= 3 and st.last_overlapped == len(st.plan.buckets)
        finally:
            train_engine.TrainPlan.BUCKET_BYTES = old
    finally:
        dist.destroy_process_group()
