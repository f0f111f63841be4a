from streamyolo_amd import PIPEHead  # noqa: F401  (drop-in for exps/model/pipe_head.py of the reference)
