"""Parameter containers with the checkpoint layout of yolox.models.network_blocks (yolox==0.3).

The reference imports these blocks from an un-vendored pip package (exps/model/darknet.py:7,
dfp_pafpn.py:10, tal_head.py:16 of the reference); their attribute names are the state_dict keys
(SURVEY.md §8(b), Appendix C).  Here they only OWN parameters — real nn.Conv2d / nn.BatchNorm2d
children so that `init_yolo`, `all_reduce_norm`, EMA, DDP, SGD param-grouping and strict
`load_state_dict` keep working (trap T1) — while all arithmetic runs in the HIP execution plan
(streamyolo_amd/engine.py).  Calling a block directly is a usage error, not a slow path.
"""
import torch.nn as nn


class _PlanOnly(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(
            "%s holds parameters only; StreamYOLO-AMD evaluates the network through its HIP execution "
            "plan (call the enclosing YOLOX / DFPPAFPN / TALHead module)" % type(self).__name__)


def get_activation(name="silu", inplace=True):
    if name != "silu":
        raise AttributeError("StreamYOLO-AMD kernels fuse SiLU only (every reference cfg uses act='silu'); got %r" % name)
    return nn.SiLU(inplace=inplace)


class BaseConv(_PlanOnly):
    """Conv2d(bias=False, pad=(k-1)//2) -> BatchNorm2d -> SiLU."""

    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu"):
        super().__init__()
        if groups != 1 or bias:
            raise NotImplementedError("depthwise / biased BaseConv is not used by any StreamYOLO cfg")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ksize, stride=stride,
                              padding=(ksize - 1) // 2, groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = get_activation(act, inplace=True)
        self.ksize, self.stride = ksize, stride


class Bottleneck(_PlanOnly):
    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(hidden, out_channels, 3, stride=1, act=act)
        self.use_add = shortcut and in_channels == out_channels


class CSPLayer(_PlanOnly):
    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5, depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, stride=1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act) for _ in range(n)])
        self.n, self.shortcut, self.hidden = n, shortcut, hidden


class SPPBottleneck(_PlanOnly):
    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), activation="silu"):
        super().__init__()
        if tuple(kernel_sizes) != (5, 9, 13):
            raise NotImplementedError("the SPP kernel is specialised for pools (5, 9, 13)")
        hidden = in_channels // 2
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=activation)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2) for k in kernel_sizes])
        self.conv2 = BaseConv(hidden * 4, out_channels, 1, stride=1, act=activation)


class Focus(_PlanOnly):
    def __init__(self, in_channels, out_channels, ksize=1, stride=1, act="silu"):
        super().__init__()
        self.conv = BaseConv(in_channels * 4, out_channels, ksize, stride, act=act)
