"""Shim of yolox.models.network_blocks (yolox==0.3, not vendored in the reference).

Written from SURVEY.md Appendix C.  Attribute names are the checkpoint keys the reference
relies on (exps/model/darknet.py:7,115-165; dfp_pafpn.py:10,33-105; tal_head.py:16,55-104).
"""
import torch
import torch.nn as nn


def get_activation(name="silu", inplace=True):
    table = {"silu": lambda: nn.SiLU(inplace=inplace),
             "relu": lambda: nn.ReLU(inplace=inplace),
             "lrelu": lambda: nn.LeakyReLU(0.1, inplace=inplace)}
    if name not in table:
        raise AttributeError("Unsupported act type: {}".format(name))
    return table[name]()


class BaseConv(nn.Module):
    def __init__(self, in_channels, out_channels, ksize, stride, groups=1, bias=False, act="silu"):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=ksize, stride=stride,
                              padding=(ksize - 1) // 2, groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = get_activation(act, inplace=True)

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))

    def fuseforward(self, x):
        return self.act(self.conv(x))


class DWConv(nn.Module):
    def __init__(self, in_channels, out_channels, ksize, stride=1, act="silu"):
        super().__init__()
        self.dconv = BaseConv(in_channels, in_channels, ksize=ksize, stride=stride,
                              groups=in_channels, act=act)
        self.pconv = BaseConv(in_channels, out_channels, ksize=1, stride=1, groups=1, act=act)

    def forward(self, x):
        return self.pconv(self.dconv(x))


class Bottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, shortcut=True, expansion=0.5,
                 depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        Conv = DWConv if depthwise else BaseConv
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = Conv(hidden, out_channels, 3, stride=1, act=act)
        self.use_add = shortcut and in_channels == out_channels

    def forward(self, x):
        y = self.conv2(self.conv1(x))
        return y + x if self.use_add else y


class ResLayer(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        mid = in_channels // 2
        self.layer1 = BaseConv(in_channels, mid, ksize=1, stride=1, act="lrelu")
        self.layer2 = BaseConv(mid, in_channels, ksize=3, stride=1, act="lrelu")

    def forward(self, x):
        return x + self.layer2(self.layer1(x))


class SPPBottleneck(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), activation="silu"):
        super().__init__()
        hidden = in_channels // 2
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=activation)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)
                                for k in kernel_sizes])
        self.conv2 = BaseConv(hidden * (len(kernel_sizes) + 1), out_channels, 1, stride=1,
                              act=activation)

    def forward(self, x):
        x = self.conv1(x)
        return self.conv2(torch.cat([x] + [m(x) for m in self.m], dim=1))


class CSPLayer(nn.Module):
    def __init__(self, in_channels, out_channels, n=1, shortcut=True, expansion=0.5,
                 depthwise=False, act="silu"):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv2 = BaseConv(in_channels, hidden, 1, stride=1, act=act)
        self.conv3 = BaseConv(2 * hidden, out_channels, 1, stride=1, act=act)
        self.m = nn.Sequential(*[Bottleneck(hidden, hidden, shortcut, 1.0, depthwise, act=act)
                                 for _ in range(n)])

    def forward(self, x):
        a = self.m(self.conv1(x))
        b = self.conv2(x)
        return self.conv3(torch.cat((a, b), dim=1))


class Focus(nn.Module):
    def __init__(self, in_channels, out_channels, ksize=1, stride=1, act="silu"):
        super().__init__()
        self.conv = BaseConv(in_channels * 4, out_channels, ksize, stride, act=act)

    def forward(self, x):
        tl = x[..., ::2, ::2]
        tr = x[..., ::2, 1::2]
        bl = x[..., 1::2, ::2]
        br = x[..., 1::2, 1::2]
        return self.conv(torch.cat((tl, bl, tr, br), dim=1))
