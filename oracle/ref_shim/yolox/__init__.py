"""Minimal stand-in for the un-vendored `yolox==0.3` package (test infrastructure only)."""
