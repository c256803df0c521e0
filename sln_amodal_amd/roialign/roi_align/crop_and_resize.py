"""crop_and_resize on MI355X behind the reference's Python surface
(roialign/roi_align/crop_and_resize.py:10-67).

`CropAndResizeFunction(crop_h, crop_w, extrapolation_value=0)(image, boxes, box_ind)`
keeps the legacy call shape; underneath is a new-style autograd Function that
calls sln_crop_and_resize_{fwd,bwd}_f32 (csrc/crop_and_resize.hip).  Channels-last
images run the NHWC kernels and return channels-last crops; contiguous NCHW
images run the reference-layout kernels.  backward returns (grad_image, None,
None): no gradient reaches the boxes (crop_and_resize.py:35-50)."""
import torch
import torch.nn as nn

from ... import ops

# When True every call checks the device-side bad-box-index flag (one host sync)
# and raises, where the reference CPU path exit(-1)s (crop_and_resize.c:39-42).
VALIDATE = False


class _CropAndResize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, boxes, box_ind, crop_height, crop_width, extrapolation_value):
        err = torch.zeros(1, dtype=torch.int32, device=image.device) if VALIDATE else None
        layout = ops.layout_of(image)
        if layout is None:
            image = image.contiguous()
            layout = ops.LAYOUT_NCHW
        boxes = boxes.detach().contiguous()
        box_ind = box_ind.contiguous()
        crops = ops.crop_and_resize_fwd(image, boxes, box_ind, crop_height, crop_width,
                                        extrapolation_value, err)
        if err is not None and int(err.item()) != 0:
            raise RuntimeError("crop_and_resize: box index out of range [0, %d)" % image.shape[0])
        ctx.im_size = tuple(image.shape)
        ctx.layout = layout
        ctx.save_for_backward(boxes, box_ind)
        return crops

    @staticmethod
    def backward(ctx, grad_outputs):
        boxes, box_ind = ctx.saved_tensors
        grad_image = ops.crop_and_resize_bwd(grad_outputs, boxes, box_ind, ctx.im_size, ctx.layout)
        return grad_image, None, None, None, None, None


class CropAndResizeFunction(object):
    """Legacy call shape: CropAndResizeFunction(ch, cw, ev)(image, boxes, box_ind)."""

    def __init__(self, crop_height, crop_width, extrapolation_value=0):
        self.crop_height = crop_height
        self.crop_width = crop_width
        self.extrapolation_value = extrapolation_value

    def __call__(self, image, boxes, box_ind):
        return _CropAndResize.apply(image, boxes, box_ind, self.crop_height, self.crop_width,
                                    float(self.extrapolation_value))


class CropAndResize(nn.Module):
    """nn.Module form (crop_and_resize.py:53-67)."""

    def __init__(self, crop_height, crop_width, extrapolation_value=0):
        super(CropAndResize, self).__init__()
        self.crop_height = crop_height
        self.crop_width = crop_width
        self.extrapolation_value = extrapolation_value

    def forward(self, image, boxes, box_ind):
        return CropAndResizeFunction(self.crop_height, self.crop_width,
                                     self.extrapolation_value)(image, boxes, box_ind)
