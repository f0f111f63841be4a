"""CSPDarknet parameter tree (mirrors exps/model/darknet.py:97-165 of the reference: same
constructor, attribute names and state_dict keys).  The legacy `Darknet` (v3) class of that file
is referenced by no cfg and is out of scope."""
import torch.nn as nn

from .network_blocks import BaseConv, CSPLayer, Focus, SPPBottleneck, _PlanOnly


class CSPDarknet(_PlanOnly):
    def __init__(self, dep_mul, wid_mul, out_features=("dark3", "dark4", "dark5"), depthwise=False, act="silu"):
        super().__init__()
        assert out_features, "please provide output features of Darknet"
        if depthwise:
            raise NotImplementedError("depthwise=True is not used by any StreamYOLO cfg")
        self.out_features = out_features
        bc = int(wid_mul * 64)
        bd = max(round(dep_mul * 3), 1)
        self.base_channels, self.base_depth = bc, bd
        self.stem = Focus(3, bc, ksize=3, act=act)
        self.dark2 = nn.Sequential(BaseConv(bc, bc * 2, 3, 2, act=act),
                                   CSPLayer(bc * 2, bc * 2, n=bd, depthwise=depthwise, act=act))
        self.dark3 = nn.Sequential(BaseConv(bc * 2, bc * 4, 3, 2, act=act),
                                   CSPLayer(bc * 4, bc * 4, n=bd * 3, depthwise=depthwise, act=act))
        self.dark4 = nn.Sequential(BaseConv(bc * 4, bc * 8, 3, 2, act=act),
                                   CSPLayer(bc * 8, bc * 8, n=bd * 3, depthwise=depthwise, act=act))
        self.dark5 = nn.Sequential(BaseConv(bc * 8, bc * 16, 3, 2, act=act),
                                   SPPBottleneck(bc * 16, bc * 16, activation=act),
                                   CSPLayer(bc * 16, bc * 16, n=bd, shortcut=False, depthwise=depthwise, act=act))
