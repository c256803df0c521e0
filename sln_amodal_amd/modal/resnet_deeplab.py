"""Dilated ResNet pieces of the frozen DeepLab-v2 global layer module, with the
reference's module names / state-dict keys (modal/resnet_deeplab.py:26-110):
`<block>.{reduce,conv3x3,increase,shortcut}.{conv,bn}`.  Bias-free conv + BN
(eps 1e-5) + ReLU are fused through nn_ops.conv_bn_act; the GLM always runs in
eval mode without gradients (model.py:537-538)."""
import torch.nn as nn
import torch.nn.functional as F

from .. import nn_ops

_BATCH_NORM = nn.BatchNorm2d
_BOTTLENECK_EXPANSION = 4


class _ConvBnReLU(nn.Sequential):
    BATCH_NORM = _BATCH_NORM

    def __init__(self, in_ch, out_ch, kernel_size, stride, padding, dilation, relu=True):
        super(_ConvBnReLU, self).__init__()
        self.add_module("conv", nn.Conv2d(in_ch, out_ch, kernel_size, stride, padding, dilation,
                                          bias=False))
        self.add_module("bn", _BATCH_NORM(out_ch, eps=1e-5, momentum=0.999))
        self.has_relu = relu
        if relu:
            self.add_module("relu", nn.ReLU())

    def forward(self, x, residual=None, relu=None, parts_only=False):
        return nn_ops.conv_bn_act(x, self.conv, self.bn,
                                  relu=self.has_relu if relu is None else relu, residual=residual,
                                  parts_only=parts_only)


class _Bottleneck(nn.Module):
    def __init__(self, in_ch, out_ch, stride, dilation, downsample):
        super(_Bottleneck, self).__init__()
        mid_ch = out_ch // _BOTTLENECK_EXPANSION
        self.reduce = _ConvBnReLU(in_ch, mid_ch, 1, stride, 0, 1, True)
        self.conv3x3 = _ConvBnReLU(mid_ch, mid_ch, 3, 1, dilation, dilation, True)
        self.increase = _ConvBnReLU(mid_ch, out_ch, 1, 1, 0, 1, False)
        self.has_shortcut = bool(downsample)
        if downsample:
            self.shortcut = _ConvBnReLU(in_ch, out_ch, 1, stride, 0, 1, False)

    def forward(self, x):
        # Frozen net: every tensor of a block is read by convolutions and by the next shortcut add only (the
        # last block's output by the four ASPP branches), so on the packed HIP path no fp32 copy of any of
        # them is written -- the shortcut is added from the parts (nn_ops.conv_bn_act parts_only)
        sc = self.shortcut(x, parts_only=True) if self.has_shortcut else x
        h = self.conv3x3(self.reduce(x, parts_only=True), parts_only=True)
        return self.increase(h, residual=sc, relu=True, parts_only=True)  # relu(increase(h) + shortcut)


class _ResLayer(nn.Sequential):
    def __init__(self, n_layers, in_ch, out_ch, stride, dilation, multi_grids=None):
        super(_ResLayer, self).__init__()
        if multi_grids is None:
            multi_grids = [1 for _ in range(n_layers)]
        else:
            assert n_layers == len(multi_grids)
        for i in range(n_layers):
            self.add_module("block{}".format(i + 1), _Bottleneck(
                in_ch=(in_ch if i == 0 else out_ch), out_ch=out_ch,
                stride=(stride if i == 0 else 1), dilation=dilation * multi_grids[i],
                downsample=(i == 0)))


class _Stem(nn.Sequential):
    """7x7/2 conv-bn-relu + MaxPool(3, 2, 1, ceil_mode=True)."""

    def __init__(self, out_ch):
        super(_Stem, self).__init__()
        self.add_module("conv1", _ConvBnReLU(3, out_ch, 7, 2, 3, 1))
        self.add_module("pool", nn.MaxPool2d(3, 2, 1, ceil_mode=True))

    def forward(self, x):
        p = self.pool
        return nn_ops.max_pool_ceil(self.conv1(x), p.kernel_size, p.stride, p.padding)
