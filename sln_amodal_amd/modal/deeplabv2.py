"""DeepLab-v2 (dilated ResNet-101, output stride 8, ASPP rates 6/12/18/24) with
the reference's names and state-dict keys (modal/deeplabv2.py:16-65):
`base.layer{1..5}....`, `base.aspp.c{0..3}.{weight,bias}`."""
import torch.nn as nn

from .. import nn_ops
from .msc_deeplab import MSC
from .resnet_deeplab import _ConvBnReLU, _ResLayer, _Stem


def DeepLabV2_ResNet101_MSC(n_classes):
    return MSC(base=DeepLabV2(n_classes=n_classes, n_blocks=[3, 4, 23, 3],
                              atrous_rates=[6, 12, 18, 24]), scales=[0.5, 0.75])


class _ASPP(nn.Module):
    """Sum of four dilated 3x3 convs (with bias) on the 2048-channel map."""

    def __init__(self, in_ch, out_ch, rates):
        super(_ASPP, self).__init__()
        for i, rate in enumerate(rates):
            self.add_module("c{}".format(i), nn.Conv2d(in_ch, out_ch, 3, 1, padding=rate,
                                                       dilation=rate, bias=True))
        for m in self.children():
            nn.init.normal_(m.weight, mean=0, std=0.01)
            nn.init.constant_(m.bias, 0)

    def forward(self, x):
        out = None
        for stage in self.children():
            out = nn_ops.conv_bn_act(x, stage, residual=out)  # running sum in the epilogue
        return out


class DeepLabV2(nn.Sequential):
    supports_packed_scales = True      # every layer takes a conv_hip.MultiScale (msc_deeplab.MSC._forward_packed)

    def __init__(self, n_classes, n_blocks, atrous_rates):
        super(DeepLabV2, self).__init__()
        ch = [64 * 2 ** p for p in range(6)]
        self.add_module("layer1", _Stem(ch[0]))
        self.add_module("layer2", _ResLayer(n_blocks[0], ch[0], ch[2], 1, 1))
        self.add_module("layer3", _ResLayer(n_blocks[1], ch[2], ch[3], 2, 1))
        self.add_module("layer4", _ResLayer(n_blocks[2], ch[3], ch[4], 1, 2))
        self.add_module("layer5", _ResLayer(n_blocks[3], ch[4], ch[5], 1, 4))
        self.add_module("aspp", _ASPP(ch[5], n_classes, atrous_rates))

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, _ConvBnReLU.BATCH_NORM):
                m.eval()
