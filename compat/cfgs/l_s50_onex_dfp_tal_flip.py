"""`l_s50_onex_dfp_tal_flip.py` — the name BASELINE.json and the reference's README use — as an alias of the file the
reference actually ships, `cfgs/l_s50_onex_dfp_tal_filp.py` ("filp").  The reference cfg is located on
`sys.path` / `$STREAMYOLO_REFERENCE` and executed unchanged; `Exp` is re-exported, so
`python tools/train.py -f compat/cfgs/l_s50_onex_dfp_tal_flip.py ...` and `get_exp(<this file>)` build the same
experiment (with `compat/` ahead of the reference on PYTHONPATH its `get_model()` returns the MI355X-native model)."""
import importlib.util
import os
import sys

_NAME = "l_s50_onex_dfp_tal_filp.py"


def _locate():
    roots = [os.environ.get("STREAMYOLO_REFERENCE", "")] + list(sys.path)
    for r in roots:
        for cand in (os.path.join(r, "cfgs", _NAME), os.path.join(r, _NAME)):
            if r and os.path.isfile(cand) and os.path.abspath(cand) != os.path.abspath(__file__):
                return cand
    raise ImportError("cfgs/%s of the StreamYOLO checkout is not on sys.path (set STREAMYOLO_REFERENCE)" % _NAME)


_spec = importlib.util.spec_from_file_location("l_s50_onex_dfp_tal_filp", _locate())
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)


class Exp(_mod.Exp):
    def __init__(self):
        super().__init__()
        self.exp_name = "l_s50_onex_dfp_tal_flip"
