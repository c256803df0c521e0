"""Import harness for the Python reference -- runs ONLY in the build container.

It puts /root/reference on sys.path and registers stand-in modules for
third-party packages the image lacks (skimage, tensorboardX, pycocotools,
scipy.misc, cv2) so that the reference's *pure Python* functions can be
imported and executed to generate golden vectors (tools/gen_golden.py).  None
of the stubbed packages takes part in the arithmetic of any captured function.

The reference's two native extensions (nms/_ext, roialign/roi_align/_ext) cannot
be built here (they need PyTorch 0.4's TH headers and torch.utils.ffi).  The
harness installs fake `_ext` modules that forward to THIS repo's C oracle, and
a new-style autograd wrapper for the legacy `CropAndResizeFunction`.  Goldens of
callers that go through them (proposal_layer, pyramid_roi_align,
detection_target_layer) therefore pin the reference's Python graph logic around
the native ops, not the native arithmetic itself -- gen_golden.py tags those
fixtures `native: "oracle"`.

Contains no reference code.  Never shipped to, or run on, the GPU box.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    if ROOT not in sys.path:
        sys.path.insert(1, ROOT)
    import matplotlib
    matplotlib.use("Agg")

    def _na(*a, **k):
        raise RuntimeError("stubbed third-party function called")

    # --- absent third-party packages, none on a captured arithmetic path ---
    sk = _stub("skimage")
    sk.color = _stub("skimage.color")
    sk.io = _stub("skimage.io", imread=_na)
    sk.measure = _stub("skimage.measure", label=_na, regionprops=_na)
    sk.morphology = _stub("skimage.morphology", remove_small_objects=_na)
    _stub("tensorboardX", SummaryWriter=lambda *a, **k: None)
    pc = _stub("pycocotools")
    pc.coco = _stub("pycocotools.coco", COCO=type("COCO", (), {}))
    pc.cocoeval = _stub("pycocotools.cocoeval", COCOeval=type("COCOeval", (), {}))
    pc.mask = _stub("pycocotools.mask")
    _stub("cv2")
    try:
        import scipy.misc  # noqa: F401
    except Exception:
        import scipy
        scipy.misc = _stub("scipy.misc", imresize=_na)

    # --- native extensions -> this repo's C oracle ---
    from oracle import oracle as orc

    def cpu_nms(keep, num_out, boxes, order, areas, thresh):
        # nms/src/nms.c contract: boxes [N,5], order = indices sorted by score desc.
        b = boxes.numpy()
        # Re-express through the oracle: it sorts (stably) itself; the reference
        # callers pass scores already sorted, so `order` is the identity up to ties.
        k = orc.nms(b, float(thresh))
        keep[: len(k)] = torch.from_numpy(k)
        num_out[0] = len(k)
        return 1

    ext_nms = _stub("nms._ext.nms", cpu_nms=cpu_nms)
    import nms as _nms_pkg  # reference package (its __init__ is empty)
    ext_pkg = _stub("nms._ext", nms=ext_nms)
    _nms_pkg._ext = ext_pkg

    def car_fwd(image, boxes, box_ind, extrap, ch, cw, crops):
        out = orc.crop_and_resize_fwd(image.detach().numpy(), boxes.detach().numpy(),
                                      box_ind.numpy(), ch, cw, extrap)
        crops.resize_(*out.shape)
        crops.copy_(torch.from_numpy(out))

    def car_bwd(grads, boxes, box_ind, grads_image):
        gi = orc.crop_and_resize_bwd(grads.numpy(), boxes.detach().numpy(), box_ind.numpy(),
                                     tuple(grads_image.shape))
        grads_image.copy_(torch.from_numpy(gi))

    car = _stub("roialign.roi_align._ext.crop_and_resize",
                crop_and_resize_forward=car_fwd, crop_and_resize_backward=car_bwd)
    _stub("roialign.roi_align._ext", crop_and_resize=car)

    import roialign.roi_align.crop_and_resize as ref_car  # reference shim module

    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, image, boxes, box_ind, ch, cw, ev):
            crops = torch.zeros(1)
            car_fwd(image, boxes, box_ind, ev, ch, cw, crops)
            ctx.im_size = image.size()
            ctx.save_for_backward(boxes, box_ind)
            return crops

        @staticmethod
        def backward(ctx, g):
            boxes, box_ind = ctx.saved_tensors
            gi = torch.zeros(*ctx.im_size)
            car_bwd(g.contiguous(), boxes, box_ind, gi)
            return gi, None, None, None, None, None

    class CropAndResizeFunction:  # legacy call shape: F(ch, cw, ev)(image, boxes, ind)
        def __init__(self, ch, cw, ev=0):
            self.a = (ch, cw, ev)

        def __call__(self, image, boxes, box_ind):
            return _Fn.apply(image, boxes, box_ind, *self.a)

    ref_car.CropAndResizeFunction = CropAndResizeFunction
    import modal.modals as ref_modals
    import modal.Functions as ref_functions
    ref_modals.CropAndResizeFunction = CropAndResizeFunction
    ref_functions.CropAndResizeFunction = CropAndResizeFunction
    return ref_modals, ref_functions


from tests._util import key_init_  # noqa: E402,F401  (shared with the parity tests)
