from streamyolo_amd import TALHead  # noqa: F401  (drop-in for exps/model/tal_head.py of the reference)
