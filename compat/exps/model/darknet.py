from streamyolo_amd import CSPDarknet  # noqa: F401  (drop-in for exps/model/darknet.py of the reference)
