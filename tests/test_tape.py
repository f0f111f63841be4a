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


def _fake_streams(n):
    """n distinct non-null 'stream handles': the emulator issues nothing on them, the library only compares them."""
    return [C.c_void_p(0x1000 + 16 * i) for i in range(n)]


def test_ring_slot_waits_for_every_reader_on_another_stream(backend):
    """A raw-gradient ring slot is read by the weight gradient (its own stream) AND by the data gradient(s) on the producing
    chain(s); the chain that overwrites the slot next must wait for all of them that live on other streams (chains 0 and
    2 share the ring, and only the weight gradient's event used to guard a slot)."""
    if not _lib.is_emulator():
        pytest.skip("event bookkeeping is counted without a device (fake stream handles)")
    a, b = _views(backend, 2)
    tape = _lib.NativeTape()
    with tape:
        # layer L on chain 2: acquire, produce, wgrad on stream 1, dgrad on chain 2
        tape.mark("cur", 2); tape.mark("acquire_cur", 3)
        ops.view_copy(a, b)
        tape.mark("dep", (2, 1)); tape.mark("cur", 1)
        ops.view_copy(a, b)
        tape.mark("slot_done", 3); tape.mark("cur", 2)
        ops.view_copy(a, b)
        tape.mark("slot_done", 3)
        # five layers later chain 0 takes the slot: it must wait for stream 1 AND stream 2
        tape.mark("cur", 0); tape.mark("acquire_cur", 3)
        ops.view_copy(a, b)
        # ... and chain 2 acquiring its part of the same slot waits for stream 1 only (its own dgrad is stream-ordered)
        tape.mark("cur", 2); tape.mark("acquire_cur", 3)
        # the next layer's readers start a new set: one event, on stream 1
        tape.mark("cur", 1); tape.mark("slot_done", 3)
        tape.mark("cur", 0); tape.mark("acquire_cur", 3)
    main, side, s2 = _fake_streams(3)
    tape.replay(main, side, more=[s2])
    events, waits = tape.counters()
    # events: dep, slot_done x 2, slot_done = 4; waits: dep 1 + first acquire 0 + chain 0's acquire 2 + chain 2's 1 + last 1
    assert (events, waits) == (4, 5)
    tape.replay(main, side, more=[])              # chain 2 folded onto main: its events are same-stream for chain 0
    events, waits = tape.counters()
    assert (events, waits) == (4, 1 + 1 + 1 + 1)  # dep, then each acquisition waits for stream 1 only
    tape.replay(main, None)                       # one stream: nothing to order
    assert tape.counters() == (0, 0)


def test_break_reports_the_stream_the_cursor_really_issues_on(backend):
    """A snippet recorded while the cursor is on chain 2 runs on the MAIN stream when the replay was given no stream for chain 2
    (hipGraph capture folds it back): the break must report stream 0 then, not 2."""
    if not _lib.is_emulator():
        pytest.skip("needs fake stream handles (emulator)")
    seen = []
    tape = _lib.NativeTape()
    with tape:
        tape.mark("cur", 2)
        tape.snippet(lambda: None)
    main, side, s2 = _fake_streams(3)
    tape.replay(main, side, on_snippet=lambda f, k: seen.append(k), more=[s2])
    tape.replay(main, side, on_snippet=lambda f, k: seen.append(k), more=[])
    assert seen == [2, 0]
    lib = _lib.lib()
    t2 = _lib.NativeTape()
    with t2:
        assert lib.sy_tape_mark(_lib.TAPE_MARKS["acquire"], -1) != 0              # negative ring slot
        assert lib.sy_tape_mark(_lib.TAPE_MARKS["acquire_cur"], -1) != 0
