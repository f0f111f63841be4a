"""The C-ABI library loads and exports every symbol include/streamyolo_hip.h declares (and the
Python binding lists exactly those).  No compute calls: runs without a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

from streamyolo_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "streamyolo_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sy_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def hip_lib():
    if not os.path.exists(_lib.DEFAULT_PATH):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "streamyolo_amd", "csrc"), "-j8"], check=True)
    return _lib.DEFAULT_PATH


def test_header_symbols_match_binding():
    assert _declared() == sorted(_lib.SIGNATURES)


def test_gfx950_library_exports_every_symbol(hip_lib):
    out = subprocess.run(["nm", "-D", "--defined-only", hip_lib], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (sy_[a-z0-9_]+)", out))
    assert exported == set(_declared())                      # nothing else leaks (-fvisibility=hidden)
    # the code object inside is gfx950
    raw = open(hip_lib, "rb").read()
    assert b"gfx950" in raw


def test_library_loads_and_reports_abi(hip_lib):
    lib = ctypes.CDLL(hip_lib)
    lib.sy_abi_version.restype = ctypes.c_int
    lib.sy_version.restype = ctypes.c_char_p
    assert lib.sy_abi_version() == _lib.ABI_VERSION == 7
    assert b"gfx950" in lib.sy_version()
    lib.sy_postprocess_workspace_bytes.restype = ctypes.c_int64
    assert lib.sy_postprocess_workspace_bytes(1, 11850) > 11850 * 11850 // 8


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.HipLibraryError):
        _lib._bind(str(tmp_path / "nope.so"))


def test_desc_struct_sizes_match_header(tmp_path):
    """sizeof(sy_conv_desc) / sizeof(sy_wgrad_desc) as gcc sees the header == the ctypes mirrors."""
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu\\n", sizeof(sy_conv_desc), sizeof(sy_wgrad_desc));return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    a, b = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(a) == ctypes.sizeof(_lib.ConvDesc) and int(b) == ctypes.sizeof(_lib.WgradDesc)
