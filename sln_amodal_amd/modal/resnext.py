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


def _fp16_storage():
    """conv_hip.PARTS == 1 (configs[4] as stated: fp16 MFMA, fp16 storage): the tensors inside a block -- read by the
    next convolution and as ReLU patterns only -- and the block outputs -- read by convolutions and as the next
    shortcut -- then exist as their scaled fp16 part alone (conv_bn_act(parts_only=True)): no fp32 copy is written."""
    if nn_ops.BACKEND == "torch":
        return False
    try:
        from .. import conv_hip
    except ImportError:          # pragma: no cover
        return False
    return conv_hip.PARTS == 1


def _cba(x, conv, bn, residual=None, parts_only=False):
    """conv -> frozen BN -> (+ shortcut) -> ReLU, one fused launch on the HIP path."""
    return nn_ops.conv_bn_act(x, conv, bn, relu=True, residual=residual, parts_only=parts_only)


class GroupBottleneck(nn.Module):
    """1x1 reduce -> grouped 3x3 (carries the stride) -> 1x1 expand to 2 x planes, shortcut added before the last
    ReLU (modal/resnext.py:31-66).  Attribute names are the reference's state-dict keys."""
    expansion = 2

    def __init__(self, inplanes, planes, stride=1, groups=1, downsample=None):
        super(GroupBottleneck, self).__init__()
        wide = planes * self.expansion
        shapes = ((inplanes, planes, 1, 1, 0, 1), (planes, planes, 3, stride, 1, groups), (planes, wide, 1, 1, 0, 1))
        for i, (cin, cout, k, s, pad, g) in enumerate(shapes, 1):
            setattr(self, "conv%d" % i, nn.Conv2d(cin, cout, kernel_size=k, stride=s, padding=pad, groups=g, bias=False))
            setattr(self, "bn%d" % i, _bn(cout))
        self.relu = nn.ReLU(inplace=True)       # (kept for module-tree parity; the ReLUs run inside the fused convs)
        self.downsample = downsample
        self.stride = stride
        self.materialize_output = False      # True on a stage's last block when something other than a conv reads it

    def forward(self, x):
        po = _fp16_storage() and x.is_cuda
        conv = nn_ops.conv_bn_act
        # the backward fusions of modal.modals.Bottleneck (conv_hip._ConvFn): an identity shortcut's gradient is added
        # inside conv1's data-gradient epilogue (`link`) instead of by an autograd accumulation pass, and the block
        # output's gradient preparation (ReLU mask, BN scale -> operand parts) is done by the next identity block's
        # conv1 data gradient (`chain`): inside a stage that conv1 -- with its shortcut through `link` -- is the
        # output's only reader.  Inside the block conv1 -> grouped conv2 -> conv3 chain the same way on the fp16 path
        # (conv_hip._GroupedF16Fn is both a reader and a producer of the protocol; ignored on the fp32 grouped kernels).
        shortcut, link = x, {}
        if self.downsample is not None:
            shortcut, link = conv(x, self.downsample[0], self.downsample[1], parts_only=po), None
        cx_in = getattr(x, "_sln_chain", None) if link is not None else None
        cx_out = {}
        c12, c23 = ({}, {}) if po else (None, None)
        h = conv(x, self.conv1, self.bn1, relu=True, link=link, chain_in=cx_in, chain_out=c12, parts_only=po)
        h = conv(h, self.conv2, self.bn2, relu=True, chain_in=c12, chain_out=c23, parts_only=po)
        out = conv(h, self.conv3, self.bn3, relu=True, residual=shortcut, link=link, chain_in=c23, chain_out=cx_out,
                   parts_only=po and not self.materialize_output)
        if cx_out.get("active"):
            out._sln_chain = cx_out
        return out


_STEM = ((3, 64, 2), (64, 64, 1), (64, 128, 1))                  # three 3x3 convolutions (modal/resnext.py:73-81)
_STAGES = ((128, 1), (256, 2), (512, 2), (1024, 2))              # planes, stride of layer1..4 (:85-88)


def _stem_forward(m, x):
    for i in (1, 2, 3):
        x = _cba(x, getattr(m, "conv%d" % i), getattr(m, "bn%d" % i))
    return nn_ops.max_pool_pad(x, 3, 2, 1)                        # MaxPool2d(3, 2, 1) on a post-ReLU map


class ResNeXt(nn.Module):
    def __init__(self, block, layers, groups=32, num_classes=1000):
        super(ResNeXt, self).__init__()
        for i, (cin, cout, s) in enumerate(_STEM, 1):
            setattr(self, "conv%d" % i, nn.Conv2d(cin, cout, kernel_size=3, stride=s, padding=1, bias=False))
            setattr(self, "bn%d" % i, _bn(cout))
            setattr(self, "relu%d" % i, nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        width = _STEM[-1][1]
        for i, ((planes, stride), n) in enumerate(zip(_STAGES, layers), 1):
            setattr(self, "layer%d" % i, self._stage(block, width, planes, n, stride, groups))
            width = planes * block.expansion
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(width, num_classes)
        for m in self.modules():                                   # He initialisation over the fan-out (:92-98)
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    @staticmethod
    def _stage(block, width, planes, n, stride, groups):
        """n blocks; the first one carries the stride and, when the width changes, a 1x1 + BN shortcut."""
        out = planes * block.expansion
        down = None
        if stride != 1 or width != out:
            down = nn.Sequential(nn.Conv2d(width, out, kernel_size=1, stride=stride, bias=False), _bn(out))
        return nn.Sequential(block(width, planes, stride, groups, down),
                             *[block(out, planes, groups=groups) for _ in range(n - 1)])

    def forward(self, x):
        x = _stem_forward(self, x)
        for i in (1, 2, 3, 4):
            layer = getattr(self, "layer%d" % i)
            layer[-1].materialize_output = i == 4        # the pooling below is not a convolution: fp32 copy wanted
            x = layer(x)
        return self.fc(self.avgpool(x).flatten(1))


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
        x = _stem_forward(self, x)
        maps = []
        for i in (1, 2, 3, 4):
            layer = getattr(self, "layer%d" % i)
            # (fp16 storage: a stage output handed to the caller gets its fp32 copy; layer4's sole reader otherwise is
            # the ASPP's convolutions, which take the fp16 part)
            layer[-1].materialize_output = bool(return_feature_maps)
            x = layer(x)
            maps.append(x)
        return maps if return_feature_maps else maps[-1:]


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
