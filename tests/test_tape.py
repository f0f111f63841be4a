"""Native launch tapes (sy_tape_*, csrc/tape.hip): what is recorded, what a replay re-executes, where it returns to the caller.
Stream marks are dependencies between hipStreams and cannot be observed on the emulator (one host thread); their effect is
pinned on the GPU by tests/test_model_train.py::test_overlapped_step_matches_single_stream and the 1-rank RCCL bucket test."""
import ctypes as C

import pytest
import torch

from streamyolo_amd import _lib, ops
from streamyolo_amd.ops import View


def _views(backend, n=3):
    return [View.alloc(1, 4, 8, 16, "fp32", backend, zero=True) for _ in range(n)]


def _stream(backend):
    return ops.stream_of(torch.zeros(1, device=backend))


def test_tape_records_launches_and_replays_them_on_new_data(backend):
    a, b, c = _views(backend)
    a.buf.fill_(1.0)
    tape = _lib.NativeTape()
    with tape:
        ops.view_copy(a, b)                       # b = a
        tape.mark("side")
        ops.view_copy(b, c, accumulate=True)      # c += b
        tape.mark("main", 0)
        tape.mark("join")
    n_entries, n_launches = tape.size()
    assert n_launches == 2 and n_entries == 5
    assert float(b.buf.sum()) == 512.0 and float(c.buf.sum()) == 512.0          # recording executes as well
    a.buf.fill_(2.0)
    tape.replay(_stream(backend), None)                                           # one-stream replay: marks are ignored
    assert float(b.buf.sum()) == 1024.0 and float(c.buf.sum()) == 512.0 + 1024.0
    tape.replay(_stream(backend), None)
    assert float(c.buf.sum()) == 512.0 + 2048.0


def test_tape_returns_to_the_caller_at_snippets_and_bucket_marks(backend):
    a, b, c = _views(backend)
    a.buf.fill_(1.0)
    seen = []
    tape = _lib.NativeTape()
    with tape:
        ops.view_copy(a, b)
        tape.snippet(lambda: seen.append(("snippet", float(b.buf.sum()))))
        tape.mark("bucket", 7)
        ops.view_copy(b, c)
        tape.mark("bucket", 8)
    seen.clear()
    a.buf.fill_(3.0)
    c.buf.zero_()
    tape.replay(_stream(backend), None, on_bucket=lambda k: seen.append(("bucket", k, float(c.buf.sum()))))
    # the snippet runs after the first launch, bucket 7 before the second launch, bucket 8 after it
    assert seen == [("snippet", 3.0 * 512), ("bucket", 7, 0.0), ("bucket", 8, 3.0 * 512)]
    seen.clear()
    tape.replay(_stream(backend), None)           # no bucket handler: the marks are skipped inside the library
    assert seen == [("snippet", 3.0 * 512)]


def test_tape_api_rejects_misuse(backend):
    lib = _lib.lib()
    assert lib.sy_tape_mark(_lib.TAPE_MARKS["join"], -1) != 0                     # no recording open
    t1 = _lib.NativeTape()
    with t1:
        assert not lib.sy_tape_begin()                                            # recordings do not nest
        assert lib.sy_tape_mark(99, 0) != 0                                       # unknown kind
        assert lib.sy_tape_mark(_lib.TAPE_MARKS["acquire"], 1000) != 0            # ring slot out of range
    pos, kind, arg, k = C.c_int(5), C.c_int(0), C.c_int(0), C.c_int(0)
    arr = (C.c_void_p * 1)(None)
    assert lib.sy_tape_replay_n(t1.handle, arr, 1, C.byref(pos), 0, C.byref(kind), C.byref(arg), C.byref(k)) != 0   # position beyond the end
    assert lib.sy_tape_replay_n(None, arr, 1, C.byref(pos), 0, C.byref(kind), C.byref(arg), C.byref(k)) != 0
    with pytest.raises(_lib.HipLibraryError):
        with _lib.NativeTape():
            with _lib.NativeTape():
                pass
