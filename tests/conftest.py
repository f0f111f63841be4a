import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver through gpurun)")


def pytest_collection_modifyitems(config, items):
    import torch
    # no single test may eat the GPU box's budget: a hang becomes a failure after 5 minutes (pytest-timeout); the SIMT-emulator
    # runs of the CPU suite are slower by nature
    limit = 300 if torch.cuda.is_available() else 1200
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(limit))
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


GOLDEN = os.path.join(ROOT, "tests", "golden")


def record_parity(key, **fields):
    """When STREAMYOLO_PARITY_OUT names a JSON file, the GPU parity tests also WRITE the figures they assert on (loss / output /
    gradient errors against the reference's goldens and the oracle) into it: rows[key] = fields.  The file of a full GPU run is
    committed as profiles/rNN/parity_table.json and bench.py quotes the row of the benchmarked dtype and configuration
    (`parity` of the driver line) instead of a hand-written string."""
    import json
    out = os.environ.get("STREAMYOLO_PARITY_OUT")
    if not out:
        return
    try:
        with open(out) as fh:
            d = json.load(fh)
    except (OSError, ValueError):
        d = {"rows": {}}
    try:
        d["commit"] = open(os.path.join(ROOT, "tools", ".head_commit")).read().strip()
    except OSError:
        pass
    d["metric"] = "rel = max|ours - ref| / max|ref| (loss dict: over its six entries; gradients: per parameter ||g - g_ref|| / ||g_ref||)"
    d["rows"].setdefault(key, {}).update({k: (float(v) if isinstance(v, (int, float)) or hasattr(v, "__float__") else v) for k, v in fields.items()})
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as fh:
        json.dump(d, fh, indent=1, sort_keys=True)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "libstreamyolo_emu.so")


def _build_emulator():
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "streamyolo_amd", "csrc"), "-j8", "emu"], check=True)
    return EMU_LIB


@pytest.fixture(scope="session", params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """Device the kernel tests run on.

    "emu": the SAME kernel sources compiled for the host with tests/emu (lock-step SIMT emulator),
           CPU tensors — checks indexing / predication / epilogues without a GPU (test infra only).
    "gpu": the real gfx950 library on cuda:0 — the parity tests proper (pytest -m gpu)."""
    import torch
    from streamyolo_amd import _lib
    if request.param == "emu":
        _lib.use_library(_build_emulator())
        assert _lib.is_emulator()
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _lib.use_library(_lib.DEFAULT_PATH)
    assert not _lib.is_emulator()
    return torch.device("cuda:0")
