"""cfg aliases shipped beside the `exps` alias package (put `compat/` on PYTHONPATH)."""
