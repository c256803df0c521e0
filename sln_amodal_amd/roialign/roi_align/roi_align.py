"""`RoIAlign` module surface (roialign/roi_align/roi_align.py:7-48; unused by the
model, kept for API completeness).  Pixel-space (x1,y1,x2,y2) boxes are mapped to
the normalised (y1,x1,y2,x2) convention of crop_and_resize; with
`transform_fpcoor` the samples sit at bin centres of a continuous-coordinate box
(tensorpack's convention)."""
from torch import nn
import torch

from .crop_and_resize import CropAndResizeFunction


class RoIAlign(nn.Module):
    def __init__(self, crop_height, crop_width, extrapolation_value=0, transform_fpcoor=True):
        super(RoIAlign, self).__init__()
        self.crop_height = crop_height
        self.crop_width = crop_width
        self.extrapolation_value = extrapolation_value
        self.transform_fpcoor = transform_fpcoor

    def forward(self, featuremap, boxes, box_ind):
        """featuremap [N,C,H,W]; boxes [M,4] (x1,y1,x2,y2) in pixels; box_ind [M]."""
        h, w = featuremap.shape[2], featuremap.shape[3]
        x1, y1, x2, y2 = boxes[:, 0:1], boxes[:, 1:2], boxes[:, 2:3], boxes[:, 3:4]
        if self.transform_fpcoor:
            bin_w = (x2 - x1) / float(self.crop_width)
            bin_h = (y2 - y1) / float(self.crop_height)
            nx0 = (x1 + bin_w / 2 - 0.5) / float(w - 1)
            ny0 = (y1 + bin_h / 2 - 0.5) / float(h - 1)
            nw = bin_w * float(self.crop_width - 1) / float(w - 1)
            nh = bin_h * float(self.crop_height - 1) / float(h - 1)
            norm = torch.cat((ny0, nx0, ny0 + nh, nx0 + nw), 1)
        else:
            norm = torch.cat((y1 / float(h - 1), x1 / float(w - 1),
                              y2 / float(h - 1), x2 / float(w - 1)), 1)
        fn = CropAndResizeFunction(self.crop_height, self.crop_width, self.extrapolation_value)
        return fn(featuremap, norm.detach().contiguous(), box_ind.detach())
