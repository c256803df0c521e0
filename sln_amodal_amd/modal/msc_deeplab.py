"""Multi-scale wrapper of the GLM (modal/msc_deeplab.py:13-48): run the base net
at scale 1 and at int(size*p) for p in scales, bilinearly resize the logits back
(align_corners=False) and take the element-wise maximum."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# run the scales of one layer as a single launch on the HIP backend (SLN_PACK_SCALES=0: A/B switch)
PACK_SCALES = os.environ.get("SLN_PACK_SCALES", "1") != "0"


class MSC(nn.Module):
    def __init__(self, base, scales=None):
        super(MSC, self).__init__()
        self.base = base
        self.scales = scales if scales else [0.5, 0.75]

    def _forward_packed(self, x):
        """All scales through the residual layers and the ASPP as ONE launch per layer
        (conv_hip.MultiScale); the 3-channel stems stay per scale.  Same arithmetic per
        output element as the sequential loop below (tests/test_conv_gpu.py)."""
        from .. import conv_hip
        xs = [x] + [F.interpolate(x, size=(int(x.size(2) * p), int(x.size(3) * p)), mode="bilinear",
                                  align_corners=False) for p in self.scales]
        mods = list(self.base.children())
        h = conv_hip.MultiScale.pack([mods[0](xi) for xi in xs])
        for m in mods[1:]:
            h = m(h)
        outs = h.tensors()
        return outs[0], outs[1:]

    def _packable(self, x):
        from .. import nn_ops
        # (a base declares that its layers -- children in order, the first one the per-scale stem -- take a
        # conv_hip.MultiScale: DeepLabV2 does, the ResNeXt encoder + ASPP base does not)
        return (x.is_cuda and not torch.is_grad_enabled() and nn_ops.BACKEND in ("auto", "hip") and
                nn_ops._hip_conv() is not None and PACK_SCALES and
                getattr(self.base, "supports_packed_scales", False))

    def softmax_tail(self, x):
        """Inference tail of the reference's glue (model.py:537-541) fused with the scale maximum below:
        -> ([B, C+1, H, W] = softmax probabilities of the maximum logits | argmax / 255, argmax [B, H, W]) -- one
        HIP pass over the three scales' logits instead of resizes, maxima, softmax, argmax and a concatenation."""
        from .. import ops
        logits, pyramid = self._forward_packed(x)
        return ops.msc_softmax_tail(logits, pyramid)

    def forward(self, x):
        if self._packable(x):
            logits, pyramid = self._forward_packed(x)
            H, W = logits.shape[2], logits.shape[3]
            logits_max = logits
            for l in pyramid:
                logits_max = torch.max(logits_max, F.interpolate(l, size=(H, W), mode="bilinear",
                                                                 align_corners=False))
            if self.training:
                return [logits] + pyramid + [logits_max]
            return logits_max
        logits = self.base(x)
        H, W = logits.shape[2], logits.shape[3]
        pyramid = []
        for p in self.scales:
            size = (int(x.size(2) * p), int(x.size(3) * p))
            pyramid.append(self.base(F.interpolate(x, size=size, mode="bilinear",
                                                   align_corners=False)))
        logits_max = logits
        for l in pyramid:
            logits_max = torch.max(logits_max, F.interpolate(l, size=(H, W), mode="bilinear",
                                                             align_corners=False))
        if self.training:
            return [logits] + pyramid + [logits_max]
        return logits_max
