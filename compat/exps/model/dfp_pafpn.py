from streamyolo_amd import DFPPAFPN  # noqa: F401  (drop-in for exps/model/dfp_pafpn.py of the reference)
