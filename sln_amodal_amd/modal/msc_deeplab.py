"""Multi-scale wrapper of the GLM (modal/msc_deeplab.py:13-48): run the base net
at scale 1 and at int(size*p) for p in scales, bilinearly resize the logits back
(align_corners=False) and take the element-wise maximum."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class MSC(nn.Module):
    def __init__(self, base, scales=None):
        super(MSC, self).__init__()
        self.base = base
        self.scales = scales if scales else [0.5, 0.75]

    def forward(self, x):
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
