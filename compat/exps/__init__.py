"""Alias package: put `compat/` ahead of the reference checkout on PYTHONPATH and the reference's
cfgs (`from exps.model.yolox import YOLOX`, cfgs/*.py:35-37) build the MI355X-native model unchanged.
Only `exps.model.{yolox,dfp_pafpn,tal_head,darknet}` are provided; the reference's own
`exps.dataset / exps.data / exps.evaluators / exps.train_utils` are found through `__path__` extension."""
import os
import sys

# let `exps.dataset`, `exps.data`, ... resolve to the reference checkout when it is on sys.path too
for _p in sys.path:
    _cand = os.path.join(_p, "exps")
    if os.path.isdir(_cand) and os.path.abspath(_cand) != os.path.dirname(os.path.abspath(__file__)):
        __path__.append(_cand)
