"""ctypes binding of libstreamyolo_hip.so (C ABI: include/streamyolo_hip.h).

The product path has exactly one implementation: the hand-written gfx950 kernels in this shared
library.  If it is missing the package fails loudly — there is no eager / CPU fallback
(build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C streamyolo_amd/csrc`).

`use_library(path)` exists for the test-suite only: tests/emu builds the SAME kernel sources for
the host with a lock-step SIMT emulator so that indexing logic can be checked in a GPU-less
container.  Nothing in the package ever selects it on its own.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PATH = os.path.join(_HERE, "lib", "libstreamyolo_hip.so")

DT_BF16, DT_F16, DT_F32 = 0, 1, 2
EPI_LINEAR, EPI_SILU, EPI_SIGMOID, EPI_DECODE, EPI_BNR = 0, 1, 2, 3, 4
CONV_FWD, CONV_DGRAD = 0, 1

ABI_VERSION = 7          # SY_ABI_VERSION of include/streamyolo_hip.h this binding was written against
_ERR = {1: "bad argument", 2: "kernel launch failed", 3: "unsupported shape"}
SY_ERR_UNSUPPORTED = 3


class HipLibraryError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("res", C.c_void_p), ("y", C.c_void_p), ("stat_sum", C.c_void_p), ("stat_sqsum", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("ldx", C.c_int32), ("ldy", C.c_int32), ("ldr", C.c_int32),
        ("xbs", C.c_int64), ("ybs", C.c_int64), ("rbs", C.c_int64),
        ("dtype", C.c_int32), ("y_f32", C.c_int32), ("mode", C.c_int32), ("epilogue", C.c_int32),
        ("accumulate", C.c_int32), ("dec_stride", C.c_float), ("stat_copies", C.c_int32), ("stat_segments", C.c_int32), ("tile", C.c_int32), ("x_bytes", C.c_int64), ("w_bytes", C.c_int64), ("wfrag", C.c_void_p), ("wfrag_bytes", C.c_int64),
        ("k_splits", C.c_int32), ("pre_cin", C.c_int32), ("pre_w", C.c_void_p), ("pre_w_bytes", C.c_int64),
        ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p),
    ]


class BnRunningEntry(C.Structure):
    """sy_bn_running_entry (include/streamyolo_hip.h)."""
    _fields_ = [("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("sum", C.c_void_p * 2),
                ("sqsum", C.c_void_p * 2), ("count", C.c_double * 2), ("C", C.c_int32), ("copies", C.c_int32),
                ("calls", C.c_int32), ("momentum", C.c_float), ("ld", C.c_int32), ("reserved", C.c_int32),
                ("num_batches_tracked", C.c_void_p)]


class PackEntry(C.Structure):
    """sy_pack_entry (include/streamyolo_hip.h)."""
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p), ("packed_t", C.c_void_p), ("frag", C.c_void_p),
                ("frag_t", C.c_void_p), ("co_n", C.c_int32), ("ci_n", C.c_int32), ("taps", C.c_int32),
                ("r0", C.c_int32), ("R", C.c_int32), ("R_t", C.c_int32), ("CI", C.c_int32), ("dtype", C.c_int32),
                ("tile0", C.c_int32), ("reserved", C.c_int32)]


class OptimEntry(C.Structure):
    """sy_optim_entry (include/streamyolo_hip.h)."""
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("buf", C.c_void_p), ("ema", C.c_void_p), ("n", C.c_int64),
                ("weight_decay", C.c_float), ("lr_mult", C.c_float), ("chunk0", C.c_int32), ("reserved", C.c_int32)]


def pack_table(rows):
    """ctypes array of PackEntry with the tile0 prefix filled in -> (array, total_tiles)."""
    t = 0
    for e in rows:
        e.tile0 = t
        t += -(-e.co_n // 32) * -(-e.ci_n // 32)
    return (PackEntry * len(rows))(*rows), t


class WgradDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("ldx", C.c_int32), ("lddy", C.c_int32), ("xbs", C.c_int64), ("dybs", C.c_int64),
        ("dtype", C.c_int32), ("dw_oihw", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("tile", C.c_int32), ("target_blocks", C.c_int32), ("x_bytes", C.c_int64), ("dy_bytes", C.c_int64),
    ]


_P, _I, _L, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes).  Must list every symbol include/streamyolo_hip.h declares
# (tests/test_abi.py parses the header and checks both directions).
SIGNATURES = {
    "sy_conv2d": (_I, [C.POINTER(ConvDesc), _P]),
    "sy_conv2d_wgrad": (_I, [C.POINTER(WgradDesc), _P]),
    "sy_focus_pack": (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P]),
    "sy_frames_u8_pack": (_I, [_P, _P, _I, _I, _I, _L, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "sy_resize_bilinear_nchw": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _P]),
    "sy_resize_nearest": (_I, [_P, _I, _I, _I, _I, _I, _L, _P, _I, _I, _I, _L, _I, _P]),
    "sy_resize_nearest_bwd": (_I, [_P, _I, _I, _I, _I, _I, _L, _P, _I, _I, _I, _L, _I, _I, _P]),
    "sy_spp_pool": (_I, [_P, _I, _I, _I, _I, _I, _L, _P, _I, _P]),
    "sy_spp_pool_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _L, _I, _P]),
    "sy_postprocess_workspace_bytes": (_L, [_I, _I]),
    "sy_head_decode": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _P]),
    "sy_postprocess": (_I, [_P, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P]),
    "sy_sgd_ema_step": (_I, [_P, _I, _I, _F, _F, _F, _F, _I, _P]),
    "sy_pack_weights": (_I, [_P, _I, _I, _P]),
    "sy_bn_running_update": (_I, [_P, _I, _I, _P]),
    "sy_bn_finalize": (_I, [_P, _P, _I, _I, _D, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _I, _P]),
    "sy_bn_silu_apply": (_I, [_P, _I, _P, _P, _P, _I, _P, _I, _L, _I, _I, _I, _P]),
    "sy_bn_finalize_apply": (_I, [_P, _P, _I, _D, _P, _P, _F, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _L, _I, _I, _I, _P]),
    "sy_bn_silu_bwd_reduce": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _P]),
    "sy_bn_silu_bwd_apply": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _L, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sy_tal_loss_workspace_bytes": (_L, [_I, _I, _I]),
    "sy_tal_loss": (_I, [_P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _I, _F, _F, _F, _I, _P, _P, _P, _P, _P, _I, _P]),
    "sy_tal_loss_assignment": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "sy_zero_rows": (_I, [_P, _L, _L, _L, _P]),
    "sy_bn_grid_caps": (_I, [C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "sy_view_copy": (_I, [_P, _I, _P, _I, _L, _I, _I, _I, _P]),
    "sy_splitk_epilogue": (_I, [_P, _I, _L, _I, _P, _P, _P, _I, _P, _I, _I, _I, _P]),
    "sy_rows_add_f32": (_I, [_P, _L, _P, _L, _I, _I, _I, _P]),
    "sy_pred_grad_fold_workspace_floats": (_L, [_I]),
    "sy_pred_grad_fold": (_I, [_P, _I, _L, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sy_tape_begin": (_P, []),
    "sy_tape_mark": (_I, [_I, _I]),
    "sy_tape_end": (_P, []),
    "sy_tape_size": (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sy_tape_counters": (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sy_tape_replay": (_I, [_P, _P, _P, C.POINTER(C.c_int), _I, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sy_tape_replay_n": (_I, [_P, C.POINTER(C.c_void_p), _I, C.POINTER(C.c_int), _I, C.POINTER(C.c_int), C.POINTER(C.c_int),
                               C.POINTER(C.c_int)]),
    "sy_tape_free": (None, [_P]),
    "sy_version": (C.c_char_p, []),
    "sy_abi_version": (_I, []),
}

_lib = None
_lib_path = None


def _bind(path):
    if not os.path.exists(path):
        raise HipLibraryError(
            "streamyolo_amd: %s not found. The MI355X kernels are the ONLY implementation of this "
            "package (no CPU/eager fallback). Build them: make -C streamyolo_amd/csrc" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError("streamyolo_amd: %s lacks symbol %s (stale build?)" % (path, name)) from e
        fn.restype = res
        fn.argtypes = args
    if lib.sy_abi_version() != ABI_VERSION:
        raise HipLibraryError("streamyolo_amd: ABI version mismatch in %s" % path)
    return lib


def lib():
    """The bound library (loads libstreamyolo_hip.so on first use; raises HipLibraryError if absent)."""
    global _lib, _lib_path
    if _lib is None:
        # STREAMYOLO_HIP_LIB: another build of the SAME library (A/B timing of two kernel versions on one box, tools/)
        path = os.environ.get("STREAMYOLO_HIP_LIB") or DEFAULT_PATH
        _lib = _bind(path)
        _lib_path = path
    return _lib if _tape is None else _TapeProxy(_lib, _tape)


# ---- launch tapes --------------------------------------------------------------------------------------
# A plan's launch list is static: the same C-ABI calls with the same descriptors every step.  `record()` executes
# the ordinary Python wrappers once while noting every library call (function + marshalled arguments, minus the
# trailing stream); `replay()` re-issues them on an explicitly given hipStream_t — a few ctypes calls per
# layer instead of re-deriving views, strides and descriptors in Python.
_tape = None


class _TapeProxy:
    __slots__ = ("real", "tape")

    def __init__(self, real, tape):
        self.real, self.tape = real, tape

    def __getattr__(self, name):
        fn = getattr(self.real, name)
        tape = self.tape
        if name.endswith("_supported") or name.endswith("_bytes") or name.endswith("_floats") or name.endswith("version") or name.endswith("_caps") or name.startswith("sy_tape_"):
            return fn                                           # queries: no stream argument, nothing to replay

        def call(*args):
            tape.append((fn, args[:-1], name))                  # every entry point ends with `void* stream`
            return fn(*args)
        return call


class record:
    """with record() as tape: ...wrappers...   -> tape = [(cfunc, args_without_stream, name)]"""

    def __enter__(self):
        global _tape
        assert _tape is None, "launch tapes do not nest"
        lib()
        _tape = []
        return _tape

    def __exit__(self, *exc):
        global _tape
        _tape = None
        return False


def replay(tape, stream):
    for fn, args, name in tape:
        rc = fn(*args, stream)
        if rc != 0:
            check(rc, name)


# ---- native launch tapes (sy_tape_*, csrc/tape.hip) ------------------------------------------------------------------
# The Python tape above still costs one ctypes call (~12 us with argument conversion) per launch.  A NativeTape records
# at the SY_LAUNCH level inside the library — kernel, grid, block and argument values of every launch the wrappers make
# while it is open — and sy_tape_replay re-issues the whole list from C, including the plan's stream switches and
# event record / wait pairs.  Python is re-entered only at "snippets" (torch ops between launches) and, in data-parallel
# runs, at gradient-bucket marks.
TAPE_END, TAPE_LAUNCH, TAPE_SIDE, TAPE_FORK, TAPE_SIDE_NW, TAPE_MAIN, TAPE_ACQUIRE, TAPE_JOIN, TAPE_BREAK, TAPE_BUCKET = \
    -1, 0, 1, 2, 3, 4, 5, 6, 7, 8
TAPE_CUR, TAPE_DEP, TAPE_SLOT_DONE, TAPE_ACQUIRE_CUR = 9, 10, 11, 12
TAPE_MARKS = {"side": TAPE_SIDE, "fork": TAPE_FORK, "side_nw": TAPE_SIDE_NW, "main": TAPE_MAIN, "acquire": TAPE_ACQUIRE,
              "join": TAPE_JOIN, "bucket": TAPE_BUCKET, "cur": TAPE_CUR, "dep": TAPE_DEP, "slot_done": TAPE_SLOT_DONE,
              "acquire_cur": TAPE_ACQUIRE_CUR}


class NativeTape:
    """with NativeTape() as t: <ops wrappers>, t.mark("side"), t.snippet(fn) ...   then t.replay(main, side, ...)."""

    def __init__(self):
        self.handle, self.snippets, self._lib = None, [], lib()
        self._pos, self._kind, self._arg, self._side = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)

    def __enter__(self):
        assert _tape is None, "a Python launch tape is open"
        if not self._lib.sy_tape_begin():
            raise HipLibraryError("streamyolo_amd: sy_tape_begin failed (a recording is already open on this thread)")
        return self

    def __exit__(self, et, ev, tb):
        h = self._lib.sy_tape_end()
        if et is not None:
            self._lib.sy_tape_free(h)
        else:
            self.handle = h
        return False

    def mark(self, name, arg=None):
        if name == "dep":                                         # arg = (from, to)
            arg = arg[0] * 16 + arg[1]
        check(self._lib.sy_tape_mark(TAPE_MARKS[name], -1 if arg is None else int(arg)), "sy_tape_mark")

    def snippet(self, fn):
        """A host-side piece of the step (torch ops) at this position: replay() returns to Python here and runs it."""
        self.snippets.append(fn)
        check(self._lib.sy_tape_mark(TAPE_BREAK, len(self.snippets) - 1), "sy_tape_mark")

    def size(self):
        n, l = C.c_int(0), C.c_int(0)
        check(self._lib.sy_tape_size(self.handle, C.byref(n), C.byref(l)), "sy_tape_size")
        return n.value, l.value

    def counters(self):
        """(events recorded, stream-waits issued) by the last replay pass."""
        e, w = C.c_int(0), C.c_int(0)
        check(self._lib.sy_tape_counters(self.handle, C.byref(e), C.byref(w)), "sy_tape_counters")
        return e.value, w.value

    def replay(self, main, side=None, on_snippet=None, on_bucket=None, more=()):
        """main / side (/ more...): raw hipStream_t (ctypes c_void_p or int) of chains 0 / 1 (/ 2...).  on_snippet(fn, k) runs a
        recorded snippet (default: fn()) — k = index of the cursor stream; on_bucket(k): called at gradient-bucket marks
        (None: the marks are skipped inside the library)."""
        fn, h = self._lib.sy_tape_replay_n, self.handle
        pos, kind, arg, ons = self._pos, self._kind, self._arg, self._side
        pos.value = 0
        stop_b = 0 if on_bucket is None else 1
        lst = [main] + ([side] if side is not None else []) + (list(more) if side is not None else [])
        arr = (C.c_void_p * len(lst))(*[s.value if isinstance(s, C.c_void_p) else s for s in lst])
        nst = len(lst)
        while True:
            rc = fn(h, arr, nst, C.byref(pos), stop_b, C.byref(kind), C.byref(arg), C.byref(ons))
            if rc != 0:
                check(rc, "sy_tape_replay")
            k = kind.value
            if k == TAPE_END:
                return
            if k == TAPE_BREAK:
                f = self.snippets[arg.value]
                if on_snippet is None:
                    f()
                else:
                    on_snippet(f, ons.value)
            else:
                on_bucket(arg.value)

    def __del__(self):
        try:
            if self.handle:
                self._lib.sy_tape_free(self.handle)
                self.handle = None
        except Exception:
            pass


def kernel_source_key():
    """Hash of the kernel sources (csrc/*.hip, *.h) of this checkout: stamps everything measured on a particular build — the
    tuner's persisted choices (ops._TuneStore) and the PMC traffic files bench.py quotes (tools/pmc_traffic.py)."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    for fn in sorted(os.listdir(src)) if os.path.isdir(src) else []:
        if fn.endswith((".hip", ".h")):
            with open(os.path.join(src, fn), "rb") as f:
                h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def use_library(path):
    """TEST-SUITE ONLY: bind an explicitly named build of the kernel sources (e.g. the SIMT-emulator
    build under tests/emu/_build).  Never called by the package itself."""
    global _lib, _lib_path
    _lib = _bind(path)
    _lib_path = path
    return _lib


def library_path():
    lib()
    return _lib_path


def is_emulator():
    return b"EMULATOR" in lib().sy_version()


def check(status, what):
    if status != 0:
        raise HipLibraryError("streamyolo_amd: %s failed: %s (status %d)" % (what, _ERR.get(status, "?"), status))
