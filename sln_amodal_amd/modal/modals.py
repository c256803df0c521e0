"""Detector modules with the reference's names and state-dict keys
(modal/modals.py): SamePad2d, Bottleneck, ResNet, FPN, RPN, Classifier, Mask,
pyramid_roi_align, pyramid_roi_align_image.

Differences in construction, not in arithmetic:
  * batched (B images) instead of batch 1; rois carry their image index;
  * activations are channels-last; every conv goes through nn_ops.conv_bn_act
    (conv + frozen-BN affine + residual + ReLU fused); SamePad2d is folded into
    the convolution's padding; nn.Conv2d / nn.BatchNorm2d objects are kept as
    parameter holders so reference checkpoints load by key
    (fpn.C1.0.weight, fpn.C2.0.conv1.weight, rpn.conv_shared.weight, ...);
  * pyramid_roi_align is one kernel launch over all levels (csrc/pyramid_crop.hip).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import nn_ops, ops
from ..roialign.roi_align.crop_and_resize import CropAndResizeFunction


def log2(x):
    """log(x)/log(2) in fp32, like the reference helper (modals.py:8-13)."""
    from ..utils import const_tensor
    ln2 = torch.log(const_tensor([2.0], torch.float32, x.device))
    return torch.log(x) / ln2


############################################################
#  ROIAlign Layer
############################################################

def roi_levels(boxes, image_shape):
    """FPN level of each roi, boxes [K,4] normalised (modals.py:51-64):
    4 + log2(sqrt(h*w) / (224/sqrt(image_area))), round (half-to-even), clamp 2..5."""
    y1, x1, y2, x2 = boxes.chunk(4, dim=1)
    h = y2 - y1
    w = x2 - x1
    from ..utils import const_tensor
    image_area = const_tensor([float(image_shape[0] * image_shape[1])], torch.float32, boxes.device)
    lvl = 4 + log2(torch.sqrt(h * w) / (224.0 / torch.sqrt(image_area)))
    lvl = torch.nan_to_num(lvl, nan=2.0, posinf=5.0, neginf=2.0)
    return lvl.round().int().clamp(2, 5).view(-1)


import os
CHAIN_TWO_READERS = os.environ.get("SLN_CHAIN_TWO_READERS", "1") != "0"   # RPN heads (A/B switch)
FUSE_RPN_HEADS = os.environ.get("SLN_FUSE_RPN_HEADS", "1") != "0"          # the two heads as one layer (round 5; A/B switch)

class CropGradPool(object):
    """One set of P2..P5 gradient maps shared by several pyramid crops of the same maps (classifier
    7x7 and mask 16x16, modals.py:438, 479).  Each crop's backward scatters into the shared maps; all
    but the last to run hand autograd `None` (= zero) for the maps, the last one returns the sum -- one
    memset and no accumulation pass per extra crop (1.4 GB each at 16 x 1024^2).  Only for graphs in
    which EVERY registered crop is differentiated (the train step): a crop whose output does not reach
    the loss would hold the others' gradients back."""

    def __init__(self):
        self.registered = self.pending = 0
        self.bufs = None
        self.sources = []       # gather backward: the crops' gradients wait here for ONE launch over all of them
        # conv_hip.GradInbox per map (or None): the maps' other reader (the RPN's shared conv) adds the crops'
        # gradient in its data-gradient epilogue instead of autograd's accumulation pass
        self.inboxes = None

    def hand_over(self, grads):
        """The finished gradient maps: left in the inboxes that take them, returned to autograd otherwise."""
        if not self.inboxes:
            return tuple(grads)
        return tuple(None if (box is not None and box.offer(g)) else g for g, box in zip(grads, self.inboxes))

    def register(self):
        self.registered += 1
        self.pending += 1


# 1 = the write-once gather backward (no atomics: the same bits on every run; 1.75 ms at 16 x 1024^2, 1600 rois),
# default = one-launch zero fill + atomic scatter (0.7 ms; fp32 atomic order varies in the last bit between runs)
GATHER_BACKWARD = os.environ.get("SLN_CROP_GATHER", "0") == "1"
CHAIN_FPN_LATERAL = os.environ.get("SLN_CHAIN_FPN_LATERAL", "1") != "0"      # A/B switch (FPN.forward)
STEM_POOL_HANDOFF = os.environ.get("SLN_STEM_POOL_HANDOFF", "1") != "0"      # A/B switch (_Stem.forward)
CHAIN_FPN_OUTPUTS = os.environ.get("SLN_CHAIN_FPN_OUTPUTS", "1") != "0"      # A/B switch (FPN.forward)


def _gather_backward(sources, shapes, device):
    """One write-once gather over all crop sets of the same four maps (csrc/pyramid_crop.hip): no zero fill, no
    atomics, the same bits on every run.  sources: [(g NHWC, cstride, coffset, boxes, box_ind, level, pool)]."""
    import ctypes as C
    from .. import _lib
    n = len(sources)
    B, Cc = shapes[0][0], shapes[0][1]
    grads = [torch.empty(s, dtype=torch.float32, device=device, memory_format=torch.channels_last) for s in shapes]
    vp, ia = (C.c_void_p * n), (C.c_int * n)
    total = sum(int(s[3].shape[0]) for s in sources)
    nbytes = _lib.lib().sln_pyramid_crop_bwd_gather_workspace_bytes(total, B)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.int32, device=device)
    ptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in grads])
    hw = (C.c_int * 8)(*[d for s in shapes for d in (s[2], s[3])])
    _lib.check(_lib.lib().sln_pyramid_crop_bwd_gather_f32(
        n, vp(*[s[0].data_ptr() for s in sources]), ia(*[int(s[1]) for s in sources]),
        ia(*[int(s[2]) for s in sources]), vp(*[s[3].data_ptr() for s in sources]),
        vp(*[s[4].data_ptr() for s in sources]), vp(*[s[5].data_ptr() for s in sources]),
        ia(*[int(s[3].shape[0]) for s in sources]), ia(*[int(s[6]) for s in sources]),
        ia(*[int(s[6]) for s in sources]), B, Cc, ptrs, hw, ops._ptr(ws), nbytes, ops._stream()),
        "sln_pyramid_crop_bwd_gather_f32")
    return tuple(grads)


def _pyramid_backward(ctx, g, cstride, coff):
    import ctypes as C
    from .. import _lib
    boxes, box_ind, level = ctx.saved_tensors
    g = g.contiguous(memory_format=torch.channels_last)
    B, Cc = ctx.shapes[0][0], ctx.shapes[0][1]
    pool = ctx.pool
    if GATHER_BACKWARD and ctx.pool_size <= 32:
        src = (g, cstride if cstride else Cc, coff, boxes, box_ind, level, ctx.pool_size)
        if pool is None:
            return _gather_backward([src], ctx.shapes, g.device)
        pool.sources.append(src)
        pool.pending -= 1
        if pool.pending > 0 and len(pool.sources) < 4:
            return (None,) * len(ctx.shapes)     # a later crop's backward launches the gather over all of them
        srcs, pool.sources = pool.sources, []
        if pool.pending <= 0:
            pool.pending = pool.registered
        return pool.hand_over(_gather_backward(srcs, ctx.shapes, g.device))
    first = pool is None or pool.bufs is None
    if first:
        grads = [torch.empty(s, dtype=torch.float32, device=g.device,
                             memory_format=torch.channels_last) for s in ctx.shapes]
        if pool is not None:
            pool.bufs = grads
    else:
        grads = pool.bufs
    ptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in grads])
    hw = (C.c_int * 8)(*[d for s in ctx.shapes for d in (s[2], s[3])])
    _lib.check(_lib.lib().sln_pyramid_crop_bwd_f32(
        ops._ptr(g), cstride if cstride else Cc, coff, ops._ptr(boxes), ops._ptr(box_ind), ops._ptr(level),
        boxes.shape[0], ctx.pool_size, ctx.pool_size, B, Cc, ptrs, hw, 0 if first else 1, ops._stream()),
        "sln_pyramid_crop_bwd_f32")
    if pool is not None:
        pool.pending -= 1
        if pool.pending > 0:
            return (None,) * len(grads)          # a later crop's backward returns the shared maps
        pool.bufs, pool.pending = None, pool.registered
        return pool.hand_over(grads)
    return tuple(grads)


class _PyramidCrop(torch.autograd.Function):
    """All-level crop in one launch; gradient flows to the four maps only."""

    @staticmethod
    def forward(ctx, boxes, box_ind, level, pool, grad_pool, *maps):
        import ctypes as C
        from .. import _lib
        maps = [m if m.is_contiguous(memory_format=torch.channels_last) else
                m.contiguous(memory_format=torch.channels_last) for m in maps]
        B, Cc = maps[0].shape[0], maps[0].shape[1]
        K = boxes.shape[0]
        out = torch.empty((K, Cc, pool, pool), dtype=torch.float32, device=boxes.device,
                          memory_format=torch.channels_last)
        ptrs = (C.c_void_p * 4)(*[m.data_ptr() for m in maps])
        hw = (C.c_int * 8)(*[d for m in maps for d in (m.shape[2], m.shape[3])])
        _lib.check(_lib.lib().sln_pyramid_crop_fwd_f32(
            ptrs, hw, B, Cc, ops._ptr(boxes), ops._ptr(box_ind), ops._ptr(level), K, pool, pool,
            0.0, ops._ptr(out), Cc, 0, ops._stream()), "sln_pyramid_crop_fwd_f32")
        ctx.save_for_backward(boxes, box_ind, level)
        ctx.shapes = [tuple(m.shape) for m in maps]
        ctx.pool_size, ctx.pool = pool, grad_pool
        if grad_pool is not None:
            grad_pool.register()
        return out

    @staticmethod
    def backward(ctx, g):
        return (None, None, None, None, None) + _pyramid_backward(ctx, g, 0, 0)


class _PyramidCropInto(torch.autograd.Function):
    """The mask head's torch.cat((glm_crop, roi_features), 1) (modals.py:481) without the copy: the
    roi features are cropped straight into channels [coff, coff+C) of the wide NHWC buffer whose
    first channels already hold the GLM crop (pyramid_roi_align_image(..., cat_extra=C))."""

    @staticmethod
    def forward(ctx, buf, coff, boxes, box_ind, level, pool, grad_pool, *maps):
        import ctypes as C
        from .. import _lib
        maps = [m if m.is_contiguous(memory_format=torch.channels_last) else
                m.contiguous(memory_format=torch.channels_last) for m in maps]
        B, Cc = maps[0].shape[0], maps[0].shape[1]
        ptrs = (C.c_void_p * 4)(*[m.data_ptr() for m in maps])
        hw = (C.c_int * 8)(*[d for m in maps for d in (m.shape[2], m.shape[3])])
        _lib.check(_lib.lib().sln_pyramid_crop_fwd_f32(
            ptrs, hw, B, Cc, ops._ptr(boxes), ops._ptr(box_ind), ops._ptr(level), boxes.shape[0], pool,
            pool, 0.0, ops._ptr(buf), buf.shape[1], coff, ops._stream()), "sln_pyramid_crop_fwd_f32")
        ctx.save_for_backward(boxes, box_ind, level)
        ctx.shapes = [tuple(m.shape) for m in maps]
        ctx.pool_size, ctx.coff, ctx.pool = pool, coff, grad_pool
        if grad_pool is not None:
            grad_pool.register()
        ctx.mark_dirty(buf)
        return buf

    @staticmethod
    def backward(ctx, g):
        return (None, None, None, None, None, None, None) + _pyramid_backward(ctx, g, g.shape[1], ctx.coff)


def pyramid_roi_align(inputs, pool_size, image_shape, box_ind=None, into=None, grad_pool=None):
    """inputs = [boxes] + [P2, P3, P4, P5].
    boxes: [B,R,4] (or the reference's [1,R,4]) normalised (y1,x1,y2,x2); image b's
    rois index feature-map batch entry b.  Padded roi slots may be marked by
    box_ind = -1 ([B*R] int32); their output is zero.
    Returns pooled [B*R, C, pool, pool] in roi order (reference: modals.py:20-110)."""
    boxes = inputs[0]
    feature_maps = list(inputs[1:])
    B = feature_maps[0].shape[0]
    if boxes.dim() == 3:
        R = boxes.shape[1]
        flat = boxes.reshape(-1, 4)
        if box_ind is None:
            box_ind = torch.arange(boxes.shape[0], dtype=torch.int32,
                                   device=boxes.device).repeat_interleave(R)
    else:
        flat = boxes.view(-1, 4)
        if box_ind is None:
            box_ind = torch.zeros(flat.shape[0], dtype=torch.int32, device=boxes.device)
    flat = flat.detach().contiguous()  # stop gradient into the proposals (modals.py:81)
    level = roi_levels(flat, image_shape)
    chlast = all(m.is_contiguous(memory_format=torch.channels_last) and not m.is_contiguous()
                 for m in feature_maps)
    if chlast and len(feature_maps) == 4 and into is not None:   # (buffer, channel offset): fused cat
        return _PyramidCropInto.apply(into[0], int(into[1]), flat, box_ind.contiguous(), level.contiguous(),
                                      int(pool_size), grad_pool, *feature_maps)
    if into is not None:
        raise RuntimeError("pyramid_roi_align(into=...) needs four channels-last feature maps")
    if chlast and len(feature_maps) == 4:
        return _PyramidCrop.apply(flat, box_ind.contiguous(), level.contiguous(), int(pool_size), grad_pool,
                                  *feature_maps)
    # Reference-layout (NCHW) maps: per-level crops, like the reference's loop.
    pooled = torch.zeros((flat.shape[0], feature_maps[0].shape[1], pool_size, pool_size),
                         dtype=torch.float32, device=flat.device)
    for i, lv in enumerate(range(2, 6)):
        ix = torch.nonzero((level == lv) & (box_ind >= 0))[:, 0]
        if ix.numel() == 0:
            continue
        crops = CropAndResizeFunction(pool_size, pool_size, 0)(feature_maps[i], flat[ix],
                                                               box_ind[ix].contiguous())
        pooled = pooled.index_add(0, ix, crops.contiguous())
    return pooled


def pyramid_roi_align_image(inputs, pool_size, image_shape, istrain=False, box_ind=None, cat_extra=0):
    """Single-map crop used for the GLM probabilities and the raw image
    (modals.py:112-157): ignores pyramid levels and `image_shape`.
    inputs = [boxes [B,R,4] | [R,4], image [B,C,H,W]].
    cat_extra = E > 0 (channels-last image, no gradient): the crop is written into the first C
    channels of a [K, C+E, pool, pool] NHWC buffer and returned as a view of it that carries the
    buffer (`_sln_cat_buf`) -- Mask.forward then crops its E roi-feature channels in behind it
    instead of concatenating two tensors."""
    boxes, image = inputs[0], inputs[1]
    if boxes.dim() == 3:
        R = boxes.shape[1]
        if box_ind is None:
            box_ind = torch.arange(boxes.shape[0], dtype=torch.int32,
                                   device=boxes.device).repeat_interleave(R)
        boxes = boxes.reshape(-1, 4)
    elif box_ind is None:
        box_ind = torch.zeros(boxes.shape[0], dtype=torch.int32, device=boxes.device)
    if cat_extra and image.is_cuda and not image.is_contiguous() and \
            image.is_contiguous(memory_format=torch.channels_last):
        import ctypes as C
        from .. import _lib
        # (allocated and sliced with grad mode as it is: autograd forbids the later in-place fill of a
        # view that was created under no_grad; the crop itself is not an autograd op)
        K, Cc = boxes.shape[0], image.shape[1]
        # (not a permuted view: marking a VIEW dirty in _PyramidCropInto makes autograd wrap it in
        # CopySlices, whose backward clones the [K, C+E, pool, pool] gradient three times)
        wide = torch.empty((K, Cc + cat_extra, pool_size, pool_size), dtype=torch.float32,
                           device=image.device, memory_format=torch.channels_last)
        ptrs = (C.c_void_p * 4)(*[image.data_ptr()] * 4)
        hw = (C.c_int * 8)(*[image.shape[2], image.shape[3]] * 4)
        level = torch.full((K,), 2, dtype=torch.int32, device=image.device)
        bx = boxes.detach().contiguous()
        bi = box_ind.contiguous()
        _lib.check(_lib.lib().sln_pyramid_crop_fwd_f32(
            ptrs, hw, image.shape[0], Cc, ops._ptr(bx), ops._ptr(bi), ops._ptr(level), K, pool_size,
            pool_size, 0.0, ops._ptr(wide), Cc + cat_extra, 0, ops._stream()), "sln_pyramid_crop_fwd_f32")
        view = wide.detach()[:, :Cc]
        view._sln_cat_buf = wide
        return view
    return CropAndResizeFunction(pool_size, pool_size, 0)(image, boxes.contiguous(),
                                                          box_ind.contiguous())


class SamePad2d(nn.Module):
    """TensorFlow 'SAME' padding (modals.py:159-184).  Kept as a module for
    Sequential index compatibility; the conv dispatch folds it into the conv."""

    def __init__(self, kernel_size, stride):
        super(SamePad2d, self).__init__()
        self.kernel_size = torch.nn.modules.utils._pair(kernel_size)
        self.stride = torch.nn.modules.utils._pair(stride)

    def forward(self, input):
        pt, pb = nn_ops.same_pad(input.shape[2], self.kernel_size[0], self.stride[0])
        pl, pr = nn_ops.same_pad(input.shape[3], self.kernel_size[1], self.stride[1])
        return F.pad(input, (pl, pr, pt, pb), "constant", 0)

    def __repr__(self):
        return self.__class__.__name__


############################################################
#  FPN Graph
############################################################

class FPN(nn.Module):
    def __init__(self, C1, C2, C3, C4, C5, out_channels):
        super(FPN, self).__init__()
        self.out_channels = out_channels
        self.C1, self.C2, self.C3, self.C4, self.C5 = C1, C2, C3, C4, C5
        self.P6 = nn.MaxPool2d(kernel_size=1, stride=2)
        for name, cin in (("P5", 2048), ("P4", 1024), ("P3", 512), ("P2", 256)):
            setattr(self, name + "_conv1", nn.Conv2d(cin, out_channels, kernel_size=1, stride=1))
            setattr(self, name + "_conv2", nn.Sequential(
                SamePad2d(kernel_size=3, stride=1),
                nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1)))

    def forward(self, x):
        x = x.contiguous(memory_format=torch.channels_last)
        x = self.C1(x)
        c2 = self.C2(x)
        c3 = self.C3(c2)
        c4 = self.C4(c3)
        c5 = self.C5(c4)
        conv = nn_ops.conv_bn_act
        fan = lambda c: getattr(c, "_sln_pair", None)     # c2..c4 also feed the next stage's strided pair
        p5 = conv(c5, self.P5_conv1)
        p4 = nn_ops.upsample2x_add(conv(c4, self.P4_conv1, pair=fan(c4)), p5)
        p3 = nn_ops.upsample2x_add(conv(c3, self.P3_conv1, pair=fan(c3)), p4)
        # the finest merged map is read by its 3x3 conv only, and the merge's gradient w.r.t. the lateral is the
        # identity: that conv's data gradient (kept as fp32 for the upsampled addend) also prepares the lateral
        # conv's gradient in its epilogue (conv_hip keep_dx chain: no stand-alone pass over the 1-GB map).  The
        # coarser merged maps have a second reader (the next merge) and stay on the ordinary path.
        ch2 = {"keep_dx": True} if CHAIN_FPN_LATERAL else None
        p2 = nn_ops.upsample2x_add(conv(c2, self.P2_conv1, pair=fan(c2), chain_out=ch2), p3)
        # P2..P4 are read by the RPN's shared conv and by the heads' crops, nothing else: when the crops' gradient
        # reaches the RPN conv through its inbox (model.rpn_forward), that conv's data gradient is the map's WHOLE
        # gradient and prepares it for the 3x3 conv here (SOFT chain: taken only then; P5 also feeds P6)
        soft = [({"soft": True} if CHAIN_FPN_OUTPUTS else None) for _ in range(3)]
        p5 = conv(p5, self.P5_conv2[1], same=True)
        p4 = conv(p4, self.P4_conv2[1], same=True, chain_out=soft[2])
        p3 = conv(p3, self.P3_conv2[1], same=True, chain_out=soft[1])
        p2 = conv(p2, self.P2_conv2[1], same=True, chain_in=ch2, chain_out=soft[0])
        for t, c in zip((p2, p3, p4), soft):
            if c is not None:
                try:
                    t._sln_chain = c
                except Exception:
                    pass
        p6 = p5[:, :, ::2, ::2]  # MaxPool2d(kernel 1, stride 2) == strided subsample
        p6 = p6.contiguous(memory_format=torch.channels_last)
        if x.is_cuda and torch.is_grad_enabled():
            hip = nn_ops._hip_conv() if nn_ops.BACKEND in ("auto", "hip") else None
            if hip is not None:
                # the levels' gradients rise and fall together (whichever level the positive rois fall on): one
                # scale per family (conv_hip.link_gradient_scales)
                hip.link_gradient_scales([self.P5_conv2[1].weight, self.P4_conv2[1].weight, self.P3_conv2[1].weight,
                                          self.P2_conv2[1].weight])
                hip.link_gradient_scales([self.P5_conv1.weight, self.P4_conv1.weight, self.P3_conv1.weight,
                                          self.P2_conv1.weight])
        return [p2, p3, p4, p5, p6]


############################################################
#  Resnet Graph
############################################################

class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        # the stride sits on the first 1x1 (modals.py:269)
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride)
        self.bn1 = nn.BatchNorm2d(planes, eps=0.001, momentum=0.01)
        self.padding2 = SamePad2d(kernel_size=3, stride=1)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3)
        self.bn2 = nn.BatchNorm2d(planes, eps=0.001, momentum=0.01)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1)
        self.bn3 = nn.BatchNorm2d(planes * 4, eps=0.001, momentum=0.01)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        conv = nn_ops.conv_bn_act
        residual, link, pair = x, {}, None
        # Every tensor this block produces is read by convolutions, by a shortcut add and as a ReLU mask only
        # (the block output too: the next block, the next stage's strided pair, an FPN lateral), so none of
        # them needs an fp32 copy in HBM (nn_ops.conv_bn_act parts_only); `parts_only_output = False` on the
        # module makes the block hand out an ordinary tensor (stand-alone use)
        po = True
        if self.downsample is not None:
            pair = {} if self.stride != 1 else None      # conv1 and the downsample: same x, same stride lattice
            if pair is not None:
                try:
                    x._sln_pair = pair                    # a later stride-1 reader of x (the FPN lateral) joins in
                except Exception:
                    pass
            residual, link = conv(x, self.downsample[0], self.downsample[1], pair=pair, parts_only=po), None
        c12, c23 = {}, {}   # conv1 -> conv2 -> conv3: each output has exactly one reader
        # block output -> next block: inside a stage the next identity block (its conv1, and its
        # shortcut through `link`) is the only reader; the dict travels on the tensor object and a
        # second reader is detected in backward (conv_hip._ConvFn)
        cx_in = getattr(x, "_sln_chain", None) if link is not None else None
        cx_out = {}
        out = conv(x, self.conv1, self.bn1, relu=True, link=link, chain_in=cx_in, chain_out=c12, pair=pair,
                   parts_only=po)
        out = conv(out, self.conv2, self.bn2, relu=True, same=True, chain_in=c12, chain_out=c23, parts_only=po)
        out = conv(out, self.conv3, self.bn3, relu=True, residual=residual, link=link, chain_in=c23,
                   chain_out=cx_out, parts_only=po and getattr(self, "parts_only_output", False))
        if cx_out.get("active"):
            out._sln_chain = cx_out
        return out


class _Stem(nn.Sequential):
    """C1: 7x7/2 conv + BN + ReLU + SamePad + 3x3/2 max-pool (modals.py:311-317)."""

    def forward(self, x):
        # the pool is the conv output's only reader: in backward it hands its incoming gradient to the conv's
        # gradient preparation, which gathers from it (no pool-backward pass over the 1-GB map)
        h = {} if STEM_POOL_HANDOFF else None
        x = nn_ops.conv_bn_act(x, self[0], self[1], relu=True, pool_handoff=h)
        return nn_ops.max_pool_same(x, 3, 2, handoff=h)


class ResNet(nn.Module):
    def __init__(self, architecture, stage5=False):
        super(ResNet, self).__init__()
        assert architecture in ["resnet50", "resnet101"]
        self.inplanes = 64
        self.layers = [3, 4, {"resnet50": 6, "resnet101": 23}[architecture], 3]
        self.block = Bottleneck
        self.stage5 = stage5
        self.C1 = _Stem(
            nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3),
            nn.BatchNorm2d(64, eps=0.001, momentum=0.01),
            nn.ReLU(inplace=True),
            SamePad2d(kernel_size=3, stride=2),
            nn.MaxPool2d(kernel_size=3, stride=2),
        )
        self.C2 = self.make_layer(self.block, 64, self.layers[0])
        self.C3 = self.make_layer(self.block, 128, self.layers[1], stride=2)
        self.C4 = self.make_layer(self.block, 256, self.layers[2], stride=2)
        self.C5 = self.make_layer(self.block, 512, self.layers[3], stride=2) if stage5 else None

    def forward(self, x):
        for stage in self.stages():
            x = stage(x)
        if x.is_cuda:      # stand-alone use: a real tensor (inside FPN the stages are called one by one)
            from .. import conv_hip
            x = conv_hip.materialize(x)
        return x

    def stages(self):
        return [self.C1, self.C2, self.C3, self.C4, self.C5]

    def make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride),
                nn.BatchNorm2d(planes * block.expansion, eps=0.001, momentum=0.01))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        for b in layers:       # inside a stage every block output has known readers (Bottleneck.forward)
            b.parts_only_output = True
        return nn.Sequential(*layers)


############################################################
#  Region Proposal Network
############################################################

class RPN(nn.Module):
    """Shared 3x3 + ReLU, then 1x1 class (2A) and 1x1 box (4A) heads
    (modals.py:361-412).  Returns [logits [B,HWA,2], probs, deltas [B,HWA,4]]."""

    def __init__(self, anchors_per_location, anchor_stride, depth):
        super(RPN, self).__init__()
        self.anchors_per_location = anchors_per_location
        self.anchor_stride = anchor_stride
        self.depth = depth
        self.padding = SamePad2d(kernel_size=3, stride=self.anchor_stride)
        self.conv_shared = nn.Conv2d(self.depth, 512, kernel_size=3, stride=self.anchor_stride)
        self.relu = nn.ReLU(inplace=True)
        self.conv_class = nn.Conv2d(512, 2 * anchors_per_location, kernel_size=1, stride=1)
        self.softmax = nn.Softmax(dim=2)
        self.conv_bbox = nn.Conv2d(512, 4 * anchors_per_location, kernel_size=1, stride=1)

    def forward(self, x, grad_inbox=None, chain_in=None):
        B = x.shape[0]
        if FUSE_RPN_HEADS and nn_ops.conv_pair_supported(x, self.conv_class, self.conv_bbox):
            # (round 5) the two heads as ONE pointwise layer over the concatenated weights: the 512-channel map is read
            # once per pass instead of twice, and its gradient is ONE data gradient that the shared convolution's
            # chained preparation takes as it stands -- at the 256^2 level the two-launch form moved 2 GB of fp32
            # partial gradient out and back in (profiles/HISTORY_r5.md)
            ch = {}
            x = nn_ops.conv_bn_act(x, self.conv_shared, relu=True, same=True, chain_in=chain_in, chain_out=ch,
                                   parts_only=True, grad_inbox=grad_inbox)
            y = nn_ops.conv_pair(x, self.conv_class, self.conv_bbox, chain_in=ch)
            if torch.is_grad_enabled():
                # the same weights on every pyramid level: their gradient roles share a scale (a level that draws no
                # anchors in a step has a gradient of ~1e-13)
                hip = nn_ops._hip_conv()
                hip.link_gradient_scales([nn_ops.pair_owner(self.conv_class)])
                hip.link_gradient_scales([self.conv_shared.weight])
            na = self.conv_class.out_channels
            logits = y[:, :na].permute(0, 2, 3, 1).reshape(B, -1, 2)
            bbox = y[:, na:].permute(0, 2, 3, 1).reshape(B, -1, 4)
            return [logits, self.softmax(logits), bbox]
        # the shared map has exactly two readers (the two 1x1 heads): whichever data gradient runs second
        # adds the first one, applies the ReLU mask and hands conv_shared its prepared gradient
        ch = {"readers": 2} if CHAIN_TWO_READERS else None
        # (its only readers are the two 1x1 heads and their ReLU mask: no fp32 copy of the 512-channel map)
        x = nn_ops.conv_bn_act(x, self.conv_shared, relu=True, same=True, chain_in=chain_in, chain_out=ch,
                               parts_only=True, grad_inbox=grad_inbox)
        # NHWC output == the reference's permute(0,2,3,1).contiguous()
        logits = nn_ops.conv_bn_act(x, self.conv_class, chain_in=ch).permute(0, 2, 3, 1).reshape(B, -1, 2)
        probs = self.softmax(logits)
        bbox = nn_ops.conv_bn_act(x, self.conv_bbox, chain_in=ch).permute(0, 2, 3, 1).reshape(B, -1, 4)
        return [logits, probs, bbox]


############################################################
#  Feature Pyramid Network Heads
############################################################

class Classifier(nn.Module):
    def __init__(self, depth, pool_size, image_shape, num_classes):
        super(Classifier, self).__init__()
        self.depth = depth
        self.pool_size = pool_size
        self.image_shape = image_shape
        self.num_classes = num_classes
        self.conv1 = nn.Conv2d(self.depth, 1024, kernel_size=self.pool_size, stride=1)
        self.bn1 = nn.BatchNorm2d(1024, eps=0.001, momentum=0.01)
        self.conv2 = nn.Conv2d(1024, 1024, kernel_size=1, stride=1)
        self.bn2 = nn.BatchNorm2d(1024, eps=0.001, momentum=0.01)
        self.relu = nn.ReLU(inplace=True)
        self.linear_class = nn.Linear(1024, num_classes)
        self.softmax = nn.Softmax(dim=1)
        self.linear_bbox = nn.Linear(1024, num_classes * 4)

    def forward(self, x, rois, box_ind=None, grad_pool=None):
        x = pyramid_roi_align([rois] + list(x), self.pool_size, self.image_shape, box_ind, grad_pool=grad_pool)
        x = nn_ops.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        x = nn_ops.conv_bn_act(x, self.conv2, self.bn2, relu=True)
        x = x.reshape(-1, 1024)
        mrcnn_class_logits = nn_ops.linear(x, self.linear_class)
        mrcnn_probs = self.softmax(mrcnn_class_logits)
        mrcnn_bbox = nn_ops.linear(x, self.linear_bbox)
        mrcnn_bbox = mrcnn_bbox.view(mrcnn_bbox.size()[0], -1, 4)
        return [mrcnn_class_logits, mrcnn_probs, mrcnn_bbox]


class Mask(nn.Module):
    def __init__(self, depth, pool_size, image_shape, num_classes):
        super(Mask, self).__init__()
        self.depth = depth
        self.pool_size = pool_size
        self.image_shape = image_shape
        self.num_classes = num_classes
        self.padding = SamePad2d(kernel_size=3, stride=1)
        self.conv1 = nn.Conv2d(self.depth, 256, kernel_size=3, stride=1)
        self.bn1 = nn.BatchNorm2d(256, eps=0.001)
        self.conv2 = nn.Conv2d(256, 256, kernel_size=3, stride=1)
        self.bn2 = nn.BatchNorm2d(256, eps=0.001)
        self.conv3 = nn.Conv2d(256, 256, kernel_size=3, stride=1)
        self.bn3 = nn.BatchNorm2d(256, eps=0.001)
        self.conv4 = nn.Conv2d(256, 256, kernel_size=3, stride=1)
        self.bn4 = nn.BatchNorm2d(256, eps=0.001)
        self.deconv = nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2)
        self.conv5 = nn.Conv2d(256, num_classes, kernel_size=1, stride=1)
        self.sigmoid = nn.Sigmoid()
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x, rois, cls_feature, box_ind=None, grad_pool=None):
        wide = getattr(cls_feature, "_sln_cat_buf", None)
        if wide is not None and wide.shape[1] == cls_feature.shape[1] + self.depth:
            # GLM channels first (modals.py:481): they are already in `wide`; crop the roi features in
            # behind them (one launch, no concatenation copy)
            x = pyramid_roi_align([rois] + list(x), self.pool_size, self.image_shape, box_ind,
                                  into=(wide, cls_feature.shape[1]), grad_pool=grad_pool)
        else:
            x = pyramid_roi_align([rois] + list(x), self.pool_size, self.image_shape, box_ind,
                                  grad_pool=grad_pool)
            x = torch.cat((cls_feature, x), dim=1)  # GLM channels first (modals.py:481)
            x = x.contiguous(memory_format=torch.channels_last)
        conv = nn_ops.conv_bn_act
        c12, c23, c34 = {}, {}, {}   # conv1 -> conv2 -> conv3 -> conv4: one reader each (chained gradients)
        # conv1..conv3 feed one convolution each (and its ReLU mask): parts only, no fp32 copy
        x = conv(x, self.conv1, self.bn1, relu=True, same=True, chain_out=c12, parts_only=True)
        x = conv(x, self.conv2, self.bn2, relu=True, same=True, chain_in=c12, chain_out=c23, parts_only=True)
        x = conv(x, self.conv3, self.bn3, relu=True, same=True, chain_in=c23, chain_out=c34, parts_only=True)
        feat = conv(x, self.conv4, self.bn4, relu=True, same=True, chain_in=c34)   # feat is also returned
        # deconv + ReLU + 1x1 logits; the sigmoid lives in the losses (modals.py:494-497)
        x = nn_ops.deconv2x2_relu_conv1x1(feat, self.deconv, self.conv5)
        return x, feat
