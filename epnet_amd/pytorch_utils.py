"""Shared-MLP building blocks (stock torch.nn; not part of the accelerated path).

``pointnet2_modules`` and the reference's callers (lib/net/rpn.py:27-47, lib/net/rcnn_net.py:23-88)
build their 1x1-conv stacks through these names, and reference checkpoints address parameters by
the module names they create (``mlps.0.layer0.conv.weight``, ``mlps.0.layer0.bn.bn.running_mean``,
...; reference: pointnet2_lib/pointnet2/pytorch_utils.py:5-235). This file re-creates the same
public names, constructor keywords and sub-module naming on top of two small generic builders; the
dense math itself is left to PyTorch-ROCm (MIOpen / hipBLASLt), as BASELINE.json's north_star says.
"""
from typing import List, Tuple

import torch.nn as nn

_RELU = nn.ReLU(inplace=True)

# per dimensionality: (convolution, batch norm, instance norm)
_ND = {1: (nn.Conv1d, nn.BatchNorm1d, nn.InstanceNorm1d), 2: (nn.Conv2d, nn.BatchNorm2d, nn.InstanceNorm2d)}


class _Norm(nn.Sequential):
    """a Sequential holding one batch-norm called ``<name>bn`` with weight 1 / bias 0"""

    def __init__(self, width: int, nd: int, name: str = ""):
        super().__init__()
        norm = _ND[nd][1](width)
        nn.init.constant_(norm.weight, 1.0)
        nn.init.constant_(norm.bias, 0)
        self.add_module(name + "bn", norm)


class BatchNorm1d(_Norm):
    def __init__(self, in_size: int, *, name: str = ""):
        super().__init__(in_size, 1, name)


class BatchNorm2d(_Norm):
    def __init__(self, in_size: int, name: str = ""):
        super().__init__(in_size, 2, name)


class _ConvUnit(nn.Sequential):
    """[bn, act, in]? -> conv  (preact)   or   conv -> [bn, act, in]?  (default)"""

    def __init__(self, nd, in_size, out_size, kernel_size, stride, padding, activation, bn, init, bias, preact, name,
                 instance_norm):
        super().__init__()
        conv = _ND[nd][0](in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                          bias=bias and not bn)
        init(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        width = in_size if preact else out_size
        extras = []
        if bn:
            extras.append((name + "bn", (BatchNorm1d if nd == 1 else BatchNorm2d)(width)))
        if activation is not None:
            extras.append((name + "activation", activation))
        if not bn and instance_norm:
            extras.append((name + "in", _ND[nd][2](width, affine=False, track_running_stats=False)))
        ordered = extras + [(name + "conv", conv)] if preact else [(name + "conv", conv)] + extras
        for key, mod in ordered:
            self.add_module(key, mod)


class Conv1d(_ConvUnit):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: int = 1, stride: int = 1, padding: int = 0,
                 activation=_RELU, bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True,
                 preact: bool = False, name: str = "", instance_norm=False):
        super().__init__(1, in_size, out_size, kernel_size, stride, padding, activation, bn, init, bias, preact, name,
                         instance_norm)


class Conv2d(_ConvUnit):
    def __init__(self, in_size: int, out_size: int, *, kernel_size: Tuple[int, int] = (1, 1),
                 stride: Tuple[int, int] = (1, 1), padding: Tuple[int, int] = (0, 0), activation=_RELU,
                 bn: bool = False, init=nn.init.kaiming_normal_, bias: bool = True, preact: bool = False,
                 name: str = "", instance_norm=False):
        super().__init__(2, in_size, out_size, kernel_size, stride, padding, activation, bn, init, bias, preact, name,
                         instance_norm)


class SharedMLP(nn.Sequential):
    """stack of 1x1 Conv2d units named ``<name>layer{i}``"""

    def __init__(self, args: List[int], *, bn: bool = False, activation=_RELU, preact: bool = False,
                 first: bool = False, name: str = "", instance_norm: bool = False):
        super().__init__()
        for i, (cin, cout) in enumerate(zip(args[:-1], args[1:])):
            plain = not (first and preact and i == 0)  # a pre-activated first layer gets no bn / act
            self.add_module(name + "layer{}".format(i),
                            Conv2d(cin, cout, bn=plain and bn, activation=activation if plain else None,
                                   preact=preact, instance_norm=instance_norm))


class FC(nn.Sequential):
    def __init__(self, in_size: int, out_size: int, *, activation=_RELU, bn: bool = False, init=None,
                 preact: bool = False, name: str = ""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        extras = []
        if bn:
            extras.append((name + "bn", BatchNorm1d(in_size if preact else out_size)))
        if activation is not None:
            extras.append((name + "activation", activation))
        ordered = extras + [(name + "fc", fc)] if preact else [(name + "fc", fc)] + extras
        for key, mod in ordered:
            self.add_module(key, mod)
