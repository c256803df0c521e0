"""ResNeXt-101 (32 groups) backbone of BASELINE.json's configs[4], with the names and state-dict keys of the
reference's (dead-code) classes so that its checkpoints load: `modal/resnext.py:31-157` (GroupBottleneck,
ResNeXt: three 3x3 stem convolutions, layers of 128/256/512/1024 planes with expansion 2, grouped 3x3 in the
middle of every block) and the encoder wrapper `modal/models_BCE.py:194-230` (`Resnet`: everything up to layer4,
optionally all four stage outputs).  Every convolution goes through nn_ops.conv_bn_act: the dense ones on the
split-operand MFMA kernels, the grouped 3x3 on csrc/grouped_conv.hip (forward only on the GPU).  BatchNorm is
plain `nn.BatchNorm2d` with the reference's SynchronizedBatchNorm2d hyper-parameters (eps 1e-5, momentum 0.001,
`modal/lib/nn/modules/batchnorm.py:39`): in eval mode -- the only mode the path has -- they are the same function.
`DeepLabV2_ResNeXt101_MSC` puts the ASPP of `modal/deeplabv2.py:24-42` on the 2048-channel output and wraps it in
the multi-scale maximum of `modal/msc_deeplab.py:13-48`."""
import math

import torch.nn as nn

from .. import nn_ops


def _bn(c):
    return nn.BatchNorm2d(c, eps=1e-5, momentum=0.001)


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class GroupBottleneck(nn.Module):
    expansion = 2

    def __init__(self, inplanes, planes, stride=1, groups=1, downsample=None):
        super(GroupBottleneck, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, groups=groups, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = nn.Conv2d(planes, planes * 2, kernel_size=1, bias=False)
        self.bn3 = _bn(planes * 2)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x
        if self.downsample is not None:
            residual = nn_ops.conv_bn_act(x, self.downsample[0], self.downsample[1])
        out = nn_ops.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        out = nn_ops.conv_bn_act(out, self.conv2, self.bn2, relu=True)
        return nn_ops.conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=residual)


class ResNeXt(nn.Module):
    def __init__(self, block, layers, groups=32, num_classes=1000):
        self.inplanes = 128
        super(ResNeXt, self).__init__()
        self.conv1 = conv3x3(3, 64, stride=2)
        self.bn1 = _bn(64)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(64, 64)
        self.bn2 = _bn(64)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv3 = conv3x3(64, 128)
        self.bn3 = _bn(128)
        self.relu3 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 128, layers[0], groups=groups)
        self.layer2 = self._make_layer(block, 256, layers[1], stride=2, groups=groups)
        self.layer3 = self._make_layer(block, 512, layers[2], stride=2, groups=groups)
        self.layer4 = self._make_layer(block, 1024, layers[3], stride=2, groups=groups)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(1024 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, groups=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                _bn(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, groups, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=groups))
        return nn.Sequential(*layers)

    def stem(self, x):
        x = nn_ops.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        x = nn_ops.conv_bn_act(x, self.conv2, self.bn2, relu=True)
        x = nn_ops.conv_bn_act(x, self.conv3, self.bn3, relu=True)
        return nn_ops.max_pool_pad(x, 3, 2, 1)            # (post-ReLU input)

    def forward(self, x):
        x = self.layer4(self.layer3(self.layer2(self.layer1(self.stem(x)))))
        x = self.avgpool(x)
        return self.fc(x.reshape(x.size(0), -1))


def resnext101(**kwargs):
    """[3, 4, 23, 3] GroupBottlenecks, 32 groups (modal/resnext.py:139-148; no download here)."""
    return ResNeXt(GroupBottleneck, [3, 4, 23, 3], **kwargs)


class ResNeXtEncoder(nn.Module):
    """Everything of a ResNeXt up to layer4 (the reference's `Resnet(orig_resnext)` wrapper,
    modal/models_BCE.py:134-136, 194-230): -> [layer4 output], or all four stage outputs."""

    def __init__(self, orig):
        super(ResNeXtEncoder, self).__init__()
        for name in ("conv1", "bn1", "relu1", "conv2", "bn2", "relu2", "conv3", "bn3", "relu3", "maxpool",
                     "layer1", "layer2", "layer3", "layer4"):
            setattr(self, name, getattr(orig, name))

    def forward(self, x, return_feature_maps=False):
        x = ResNeXt.stem(self, x)
        out = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            x = layer(x)
            out.append(x)
        return out if return_feature_maps else [x]


class _EncoderASPP(nn.Module):
    """base of the multi-scale wrapper: encoder -> ASPP logits at stride 32."""

    def __init__(self, enc, n_classes, rates):
        super(_EncoderASPP, self).__init__()
        from .deeplabv2 import _ASPP
        self.enc = enc
        self.aspp = _ASPP(2048, n_classes, rates)

    def forward(self, x):
        return self.aspp(self.enc(x)[0])


def DeepLabV2_ResNeXt101_MSC(n_classes, layers=(3, 4, 23, 3)):
    """configs[4]: ResNeXt-101 (32 groups) encoder + ASPP(6, 12, 18, 24) under the multi-scale maximum
    (scales 1, 0.5, 0.75)."""
    from .msc_deeplab import MSC
    enc = ResNeXtEncoder(ResNeXt(GroupBottleneck, list(layers)))
    return MSC(base=_EncoderASPP(enc, n_classes, [6, 12, 18, 24]), scales=[0.5, 0.75])


def load_reference_state_dict(module, state_dict):
    """Load a checkpoint written by the reference's classes: the same keys, minus SynchronizedBatchNorm2d's
    training-time accumulators (`_tmp_running_mean`, `_tmp_running_var`, `_running_iter`:
    modal/lib/nn/modules/batchnorm.py:44-49), which have no role in eval mode, and the classifier (`fc`) when the
    module is an encoder."""
    own = module.state_dict()
    keep = {k: v for k, v in state_dict.items()
            if not any(t in k for t in ("_tmp_running", "_running_iter")) and k in own}
    missing = [k for k in own if k not in keep and "num_batches_tracked" not in k]
    if missing:
        raise KeyError("reference checkpoint lacks %d keys, e.g. %s" % (len(missing), missing[:3]))
    module.load_state_dict(keep, strict=False)
    return module
