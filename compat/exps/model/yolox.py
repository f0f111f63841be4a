from streamyolo_amd import YOLOX  # noqa: F401  (drop-in for exps/model/yolox.py of the reference)
