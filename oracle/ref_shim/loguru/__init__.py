"""Stub of `loguru.logger` (reference import: exps/model/tal_head.py:5). Test infrastructure."""
import logging as _l


class _Logger:
    def __getattr__(self, name):
        return getattr(_l.getLogger("ref_shim"), name if name != "success" else "info")


logger = _Logger()
