"""Tensor-level bindings of the C ABI (include/sln_amodal.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; all
arithmetic happens in libsln_amodal_hip.so.  Every function enqueues on
torch's current stream and returns without synchronising.
"""
import ctypes as C

import torch

from . import _lib

LAYOUT_NCHW, LAYOUT_NHWC = 0, 1


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (no CPU fallback in sln_amodal_amd)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def layout_of(t):
    """NHWC if the 4-d tensor is channels-last in memory (and not also plain
    contiguous), else NCHW."""
    if t.is_contiguous():
        return LAYOUT_NCHW
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return LAYOUT_NHWC
    return None


# ---------------------------------------------------------------------------- NMS
_ws_cache = {}


def _workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def nms_sorted(dets, thresh, max_out, n_valid=None):
    """dets [B,N,5] f32 (y1,x1,y2,x2,score), score-descending per image.
    Returns keep [B,max_out] int64 (-1 padded) and num_keep [B] int32."""
    _need(dets, torch.float32, "dets")
    if dets.dim() != 3 or dets.shape[2] != 5:
        raise ValueError("dets must be [B,N,5]")
    dets = dets.contiguous()
    B, N = dets.shape[0], dets.shape[1]
    keep = torch.empty((B, max_out), dtype=torch.int64, device=dets.device)
    num = torch.empty((B,), dtype=torch.int32, device=dets.device)
    L = _lib.lib()
    nbytes = L.sln_nms_workspace_bytes(B, N)
    ws = _workspace(nbytes, dets.device)
    if n_valid is not None:
        n_valid = _need(n_valid, torch.int32, "n_valid").contiguous()
    _lib.check(L.sln_nms_f32(_ptr(dets), B, N, _ptr(n_valid), float(thresh), int(max_out),
                             _ptr(keep), _ptr(num), _ptr(ws), nbytes, _stream()), "sln_nms_f32")
    return keep, num


# ---------------------------------------------------------------- crop_and_resize
def crop_and_resize_fwd(image, boxes, box_ind, ch, cw, extrap=0.0, err_flag=None):
    _need(image, torch.float32, "image")
    _need(boxes, torch.float32, "boxes")
    _need(box_ind, torch.int32, "box_ind")
    if image.dim() != 4:
        raise ValueError("image must be [B,C,H,W]")
    lay = layout_of(image)
    if lay is None:
        image = image.contiguous()
        lay = LAYOUT_NCHW
    boxes = boxes.contiguous().view(-1, 4)
    box_ind = box_ind.contiguous()
    B, Cc, H, W = image.shape
    K = boxes.shape[0]
    if box_ind.numel() != K:
        raise ValueError("box_ind must have one entry per box")
    fmt = torch.channels_last if lay == LAYOUT_NHWC else torch.contiguous_format
    crops = torch.empty((K, Cc, ch, cw), dtype=torch.float32, device=image.device,
                        memory_format=fmt)
    _lib.check(_lib.lib().sln_crop_and_resize_fwd_f32(
        _ptr(image), B, Cc, H, W, lay, _ptr(boxes), _ptr(box_ind), K, int(ch), int(cw),
        float(extrap), _ptr(crops), _ptr(err_flag), _stream()), "sln_crop_and_resize_fwd_f32")
    return crops


def crop_and_resize_bwd(grads, boxes, box_ind, image_shape, layout, err_flag=None):
    _need(grads, torch.float32, "grads")
    B, Cc, H, W = image_shape
    K, _, ch, cw = grads.shape
    if layout == LAYOUT_NHWC:
        grads = grads.contiguous(memory_format=torch.channels_last)
        fmt = torch.channels_last
    else:
        grads = grads.contiguous()
        fmt = torch.contiguous_format
    gimg = torch.empty((B, Cc, H, W), dtype=torch.float32, device=grads.device, memory_format=fmt)
    _lib.check(_lib.lib().sln_crop_and_resize_bwd_f32(
        _ptr(grads), _ptr(boxes), _ptr(box_ind), K, ch, cw, B, Cc, H, W, layout, _ptr(gimg),
        _ptr(err_flag), _stream()), "sln_crop_and_resize_bwd_f32")
    return gimg


# ------------------------------------------------------------------- label decode
def label_num_objects(label):
    """label [B,H,W] int64/uint64 bit pattern -> n_obj [B] int32 (device)."""
    label = _as_u64(label)
    B = label.shape[0]
    n = torch.empty((B,), dtype=torch.int32, device=label.device)
    _lib.check(_lib.lib().sln_label_num_objects_u64(_ptr(label), B, label[0].numel(), _ptr(n),
                                                    _stream()), "sln_label_num_objects_u64")
    return n


def label_zoom(src, src_hw, ys, xs):
    """src [B, stride] int64/uint64 (image b: src_hw[b] = (H0, W0) valid rows x columns, row pitch W0), ys [B,OH] /
    xs [B,OW] int32 index maps (-1 = fill 0) -> [B,OH,OW] int64: scipy.ndimage.zoom(order=0) of every label
    (utils.resize_layer, utils.py:358-362) on the device; a flipped image passes xs reversed."""
    if not src.is_cuda:
        raise RuntimeError("label must live on the GPU")
    if src.dtype not in (torch.int64, torch.uint64) or src.dim() != 2 or not src.is_contiguous():
        raise TypeError("src must be a contiguous [B, stride] tensor of 64-bit patterns")
    B, OH, OW = src.shape[0], ys.shape[1], xs.shape[1]
    for t, nm in ((src_hw, "src_hw"), (ys, "ys"), (xs, "xs")):
        _need(t, torch.int32, nm)
        if not t.is_contiguous() or t.shape[0] != B:
            raise ValueError("%s must be contiguous with one row per image" % nm)
    out = torch.empty((B, OH, OW), dtype=torch.int64, device=src.device)
    _lib.check(_lib.lib().sln_label_zoom_u64(_ptr(src), src.shape[1], _ptr(src_hw), _ptr(ys), _ptr(xs), B, OH, OW,
                                             _ptr(out), _stream()), "sln_label_zoom_u64")
    return out


def label_num_objects_ragged(src, src_hw):
    """Object counts [B] int32 of labels of different sizes packed as src [B, stride] (see label_zoom)."""
    if not src.is_cuda or src.dtype not in (torch.int64, torch.uint64) or src.dim() != 2 or not src.is_contiguous():
        raise TypeError("src must be a contiguous [B, stride] device tensor of 64-bit patterns")
    _need(src_hw, torch.int32, "src_hw")
    n = torch.empty((src.shape[0],), dtype=torch.int32, device=src.device)
    _lib.check(_lib.lib().sln_label_num_objects_ragged_u64(_ptr(src), src.shape[0], src.shape[1], _ptr(src_hw),
                                                           _ptr(n), _stream()), "sln_label_num_objects_ragged_u64")
    return n


def _as_u64(label):
    if not label.is_cuda:
        raise RuntimeError("label must live on the GPU")
    if label.dtype not in (torch.int64, torch.uint64):
        raise TypeError("label must hold 64-bit patterns (int64 or uint64)")
    if label.dim() == 2:
        label = label.unsqueeze(0)
    return label.contiguous()


def label_decode(label, L, N):
    """label [B,H,W] -> planes [B,L,N,H,W] uint8."""
    label = _as_u64(label)
    B, H, W = label.shape
    planes = torch.empty((B, L, N, H, W), dtype=torch.uint8, device=label.device)
    _lib.check(_lib.lib().sln_label_decode_u64(_ptr(label), B, H, W, int(L), int(N), _ptr(planes),
                                               _stream()), "sln_label_decode_u64")
    return planes


def mask_targets(label, L, rois, roi_img, roi_obj, mh, mw):
    """Fused decode+crop+round: -> masks [K,L,mh,mw] f32 in {0,1}."""
    label = _as_u64(label)
    B, H, W = label.shape
    rois = _need(rois, torch.float32, "rois").contiguous().view(-1, 4)
    roi_img = _need(roi_img, torch.int32, "roi_img").contiguous()
    roi_obj = _need(roi_obj, torch.int32, "roi_obj").contiguous()
    K = rois.shape[0]
    masks = torch.empty((K, L, mh, mw), dtype=torch.float32, device=label.device)
    _lib.check(_lib.lib().sln_mask_targets_u64(_ptr(label), B, H, W, int(L), _ptr(rois),
                                               _ptr(roi_img), _ptr(roi_obj), K, int(mh), int(mw),
                                               _ptr(masks), _stream()), "sln_mask_targets_u64")
    return masks


# --------------------------------------------------------------------- grouped convolution (forward only)
def grouped_conv3x3(x, weight, groups, stride=1, scale=None, shift=None, relu=False):
    """x [N,C,H,W] fp32 (channels-last in memory, else copied), weight [C, C/groups, 3, 3], padding 1 -> y
    [N,C,OH,OW] channels-last = relu?(conv * scale[c] + shift[c]).  No autograd here: GroupedConv3x3 below
    (csrc/grouped_conv.hip; reference modal/resnext.py:31-41 GroupBottleneck.conv2)."""
    _need(x, torch.float32, "x")
    N, C, H, W = x.shape
    if tuple(weight.shape) != (C, C // groups, 3, 3):
        raise ValueError("grouped_conv3x3: weight %s for C=%d groups=%d" % (tuple(weight.shape), C, groups))
    xc = x if x.permute(0, 2, 3, 1).is_contiguous() else x.contiguous(memory_format=torch.channels_last)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((N, OH, OW, C), dtype=torch.float32, device=x.device)
    w = weight.detach().contiguous()
    from . import conv_hip                       # (the live per-launch profile of bench.py: the forward entries)
    e0 = conv_hip._prof_begin()
    _lib.check(_lib.lib().sln_grouped_conv3x3_f32(_ptr(xc), N, H, W, C, int(groups), _ptr(w), int(stride),
                                                  _ptr(scale.detach().contiguous()) if scale is not None else None,
                                                  _ptr(shift.detach().contiguous()) if shift is not None else None,
                                                  1 if relu else 0, _ptr(y), _stream()), "sln_grouped_conv3x3_f32")
    conv_hip._prof_end(e0, 2.0 * N * OH * OW * C * 9 * (C // groups), "grouped_conv3x3_kernel",
                       "fwd grouped N%d %dx%d C%d g%d s%d" % (N, H, W, C, groups, stride),
                       4 * (xc.numel() + w.numel()), 4 * y.numel())
    return y.permute(0, 3, 1, 2)


class GroupedConv3x3(torch.autograd.Function):
    """conv (groups, 3x3, padding 1) -> frozen BN affine -> ReLU as one node: forward csrc/grouped_conv.hip, backward
    its data- and weight-gradient kernels with the ReLU mask and the BN scale applied while the gradient is read
    (what autograd does for the reference's nn.Conv2d(groups=32) + BN + ReLU, modal/resnext.py:50-52)."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, relu, groups, stride):
        y = grouped_conv3x3(x, weight, groups, stride, scale, shift, relu)
        ctx.save_for_backward(x, weight, scale, y if relu else None)
        ctx.cfg = (bool(relu), int(groups), int(stride))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, scale, y = ctx.saved_tensors
        relu, groups, stride = ctx.cfg
        N, C, H, W = x.shape
        nhwc = lambda t: t if t.permute(0, 2, 3, 1).is_contiguous() else t.contiguous(memory_format=torch.channels_last)
        gy, xc = nhwc(gy), nhwc(x)
        if not relu and scale is not None:          # no mask to carry the scale: fold it into the gradient
            gy, sc = gy * scale.view(1, -1, 1, 1), None
        else:
            sc = scale
        w = weight.detach().contiguous()
        gx = gw = None
        L = _lib.lib()
        if ctx.needs_input_grad[0]:
            gxb = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device)
            _lib.check(L.sln_grouped_conv3x3_dgrad_f32(_ptr(gy), _ptr(y) if relu else None, _ptr(sc), N, H, W, C, groups,
                                                       _ptr(w), stride, _ptr(gxb), _stream()),
                       "sln_grouped_conv3x3_dgrad_f32")
            gx = gxb.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            nbytes = L.sln_grouped_conv3x3_wgrad_workspace_bytes(N, H, W, C, groups, stride)
            ws = _workspace(nbytes, x.device)
            _lib.check(L.sln_grouped_conv3x3_wgrad_f32(_ptr(xc), _ptr(gy), _ptr(y) if relu else None, _ptr(sc), N, H, W,
                                                       C, groups, stride, _ptr(gw), _ptr(ws), nbytes, _stream()),
                       "sln_grouped_conv3x3_wgrad_f32")
        return gx, gw, None, None, None, None, None


# --------------------------------------------------------------------- global layer module tail
def _nhwc_stride(t):
    """Pixel stride (floats) of a logical [B,C,H,W] tensor that is NHWC in memory with rows of equal stride, else None."""
    B, C, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if sc == 1 and sw >= C and sh == W * sw and (B == 1 or sb == H * sh):
        return sw
    return None


def msc_softmax_tail(logits, pyramid):
    """logits [B,C,H,W] (scale 1) and the coarser scales' logits -> probs [B,C+1,H,W] (channels-last: the C softmax
    probabilities of the element-wise maximum over the resized scales, then argmax/255) and label [B,H,W] int64.
    Reference model.py:537-541 + modal/msc_deeplab.py:42-48 in one pass (csrc/glm_tail.hip)."""
    import ctypes as C
    ts = []
    for t in [logits] + list(pyramid):
        _need(t, torch.float32, "logits")
        if _nhwc_stride(t) is None:
            t = t.contiguous(memory_format=torch.channels_last)
        ts.append(t)
    B, Cc, H, W = ts[0].shape
    n = len(ts) - 1
    probs = torch.empty((B, H, W, Cc + 1), dtype=torch.float32, device=logits.device)
    label = torch.empty((B, H, W), dtype=torch.int64, device=logits.device)
    ptrs = (C.c_void_p * max(n, 1))(*[t.data_ptr() for t in ts[1:]])
    hw = (C.c_int32 * max(2 * n, 1))(*[d for t in ts[1:] for d in (t.shape[2], t.shape[3])])
    ps = (C.c_int64 * max(n, 1))(*[_nhwc_stride(t) for t in ts[1:]])
    _lib.check(_lib.lib().sln_msc_softmax_tail_f32(_ptr(ts[0]), _nhwc_stride(ts[0]), ptrs, hw, ps, n, B, Cc, H, W,
                                                   _ptr(probs), _ptr(label), _stream()), "sln_msc_softmax_tail_f32")
    return probs.permute(0, 3, 1, 2), label


# --------------------------------------------------------------------- proposals
def topk_order(scores, k):
    """scores [B,A] f32 (any strides) -> order [B,k] int64: the k best per row, score descending, ties
    by the lower index -- torch.sort(scores, 1, descending=True, stable=True)[1][:, :k] without
    sorting the other A - k elements."""
    _need(scores, torch.float32, "scores")
    if scores.dim() != 2:
        raise ValueError("scores must be [B,A]")
    B, A = scores.shape
    order = torch.empty((B, k), dtype=torch.int64, device=scores.device)
    nbytes = _lib.lib().sln_topk_workspace_bytes(B, A, int(k))
    ws = _workspace(nbytes, scores.device)
    _lib.check(_lib.lib().sln_topk_order_f32(_ptr(scores), B, A, scores.stride(0), scores.stride(1), int(k),
                                             _ptr(order), _ptr(ws), nbytes, _stream()), "sln_topk_order_f32")
    return order


def proposal_decode(probs, deltas, anchors, order, std_dev, win_h, win_w):
    """probs [B,A,2], deltas [B,A,4], anchors [A,4], order [B,n] int64 ->
    dets [B,n,5] (y1,x1,y2,x2,score) decoded + clipped."""
    probs = _need(probs, torch.float32, "probs").contiguous()
    deltas = _need(deltas, torch.float32, "deltas").contiguous()
    anchors = _need(anchors, torch.float32, "anchors").contiguous()
    order = _need(order, torch.int64, "order").contiguous()
    B, A = probs.shape[0], probs.shape[1]
    n = order.shape[1]
    dets = torch.empty((B, n, 5), dtype=torch.float32, device=probs.device)
    std = (C.c_float * 4)(*[float(s) for s in std_dev])
    _lib.check(_lib.lib().sln_proposal_decode_f32(_ptr(probs), _ptr(deltas), _ptr(anchors),
                                                  _ptr(order), B, A, n, std, float(win_h),
                                                  float(win_w), _ptr(dets), _stream()),
               "sln_proposal_decode_f32")
    return dets


def gather_rois(dets, keep, num_keep, norm_h, norm_w):
    B, N = dets.shape[0], dets.shape[1]
    max_out = keep.shape[1]
    rois = torch.empty((B, max_out, 4), dtype=torch.float32, device=dets.device)
    _lib.check(_lib.lib().sln_gather_rois_f32(_ptr(dets), _ptr(keep), _ptr(num_keep), B, N,
                                              max_out, float(norm_h), float(norm_w), _ptr(rois),
                                              _stream()), "sln_gather_rois_f32")
    return rois


# ------------------------------------------------------------------ inference tail
def unmold_masks(masks, class_ids, boxes, H, W):
    """masks [N,C,mh,mw] f32, class_ids [N] int32 (or None), boxes [N,4] int32 image pixels ->
    full [N,W,H] uint8 in {0,1}, column-major per mask (utils.py:447-465 for all detections)."""
    masks = _need(masks, torch.float32, "masks").contiguous()
    boxes = _need(boxes, torch.int32, "boxes").contiguous()
    if class_ids is not None:
        class_ids = _need(class_ids, torch.int32, "class_ids").contiguous()
    if masks.dim() != 4 or boxes.shape != (masks.shape[0], 4):
        raise ValueError("masks must be [N,C,mh,mw] and boxes [N,4]")
    N, Cc, mh, mw = masks.shape
    full = torch.empty((N, int(W), int(H)), dtype=torch.uint8, device=masks.device)
    _lib.check(_lib.lib().sln_unmold_masks_u8(_ptr(masks), _ptr(class_ids), _ptr(boxes), N, Cc, mh, mw,
                                              int(H), int(W), _ptr(full), _stream()), "sln_unmold_masks_u8")
    return full


def rle_encode(masks, max_runs):
    """masks [N,a] (or [N,W,H]) uint8, column-major bytes of each mask -> counts [N,max_runs] uint32
    (as int32 storage) and num_runs [N] int32 (maskApi.c:33-42).  Rows whose num_runs exceeds
    max_runs are unspecified."""
    masks = _need(masks, torch.uint8, "masks").contiguous()
    N = masks.shape[0]
    a = masks[0].numel() if N else 0
    counts = torch.empty((N, int(max_runs)), dtype=torch.int32, device=masks.device)
    num = torch.empty((N,), dtype=torch.int32, device=masks.device)
    _lib.check(_lib.lib().sln_rle_encode_u8(_ptr(masks), N, a, int(max_runs), _ptr(counts), _ptr(num),
                                            _stream()), "sln_rle_encode_u8")
    return counts, num
