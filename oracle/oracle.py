"""CPU oracle for the SLN-Amodal hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (sln_amodal_amd/) never does.

Two layers:
  * ctypes bindings of oracle/libsln_oracle.so (C restatement of the native
    arithmetic: NMS, crop_and_resize fwd/bwd, label decode, box decode, IoU);
  * numpy / torch-CPU restatements of the reference's per-image Python graph
    functions (anchors, proposal_layer, pyramid level assignment,
    detection_target_layer, build_rpn_targets, losses), each citing the
    reference file:line it follows.

Pinning status is stated per function (see also oracle/sln_oracle.c header and
DESIGN.md "Oracle").
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/libsln_oracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "libsln_oracle.so")
    src = os.path.join(_HERE, "sln_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libsln_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


# --------------------------------------------------------------------------
# native arithmetic (C restatement)
# --------------------------------------------------------------------------
def nms(dets, thresh):
    """dets [N,5] f32 (y1,x1,y2,x2,score) -> int64 keep indices, score order.
    reference: nms/pth_nms.py:5-24 + nms/src/nms.c:4-69.  parity unpinned."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    num = np.zeros(1, dtype=np.int64)
    rc = lib().orc_nms_f32(_p(dets, C.c_float), C.c_int64(n), C.c_float(thresh),
                           _p(keep, C.c_int64), _p(num, C.c_int64))
    assert rc == 0, rc
    return keep[: num[0]].copy()


def crop_and_resize_fwd(image, boxes, box_ind, ch, cw, extrap=0.0):
    """image [B,C,H,W] f32, boxes [K,4] f32 normalised, box_ind [K] i32.
    reference: roialign/roi_align/src/crop_and_resize.c:6-154. parity unpinned."""
    image = np.ascontiguousarray(image, dtype=np.float32)
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    box_ind = np.ascontiguousarray(box_ind, dtype=np.int32)
    B, Cc, H, W = image.shape
    K = boxes.shape[0]
    crops = np.empty((K, Cc, ch, cw), dtype=np.float32)
    rc = lib().orc_crop_and_resize_fwd_f32(
        _p(image, C.c_float), B, Cc, H, W, _p(boxes, C.c_float), _p(box_ind, C.c_int32),
        K, ch, cw, C.c_float(extrap), _p(crops, C.c_float))
    if rc != 0:
        raise RuntimeError("box index out of range")
    return crops


def crop_and_resize_bwd(grads, boxes, box_ind, image_shape):
    """reference: crop_and_resize.c:157-252 (serial, deterministic order)."""
    grads = np.ascontiguousarray(grads, dtype=np.float32)
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    box_ind = np.ascontiguousarray(box_ind, dtype=np.int32)
    B, Cc, H, W = image_shape
    K, _, ch, cw = grads.shape
    gi = np.empty((B, Cc, H, W), dtype=np.float32)
    rc = lib().orc_crop_and_resize_bwd_f32(
        _p(grads, C.c_float), _p(boxes, C.c_float), _p(box_ind, C.c_int32), K, ch, cw,
        B, Cc, H, W, _p(gi, C.c_float))
    if rc != 0:
        raise RuntimeError("box index out of range")
    return gi


def crop_and_resize_taps(H, W, boxes, ch, cw):
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 4)
    K = boxes.shape[0]
    taps = np.empty((K, ch, cw), dtype=np.int32)
    lib().orc_crop_and_resize_taps(H, W, _p(boxes, C.c_float), K, ch, cw,
                                   _p(taps, C.c_int32))
    return taps


def label_num_objects(label):
    label = np.ascontiguousarray(label, dtype=np.uint64)
    return int(lib().orc_label_num_objects(_p(label, C.c_uint64), C.c_int64(label.size)))


def label_decode(label, L, N=None):
    """uint64 label [H,W] -> uint8 [L,N,H,W].
    reference: amodal_train.py:236-271 + modal/Functions.py:1012-1095 (+ axis
    shuffle Functions.py:735, model.py:114).  Pinned by tests/golden/label_*."""
    label = np.ascontiguousarray(label, dtype=np.uint64)
    H, W = label.shape
    if N is None:
        N = label_num_objects(label)
    planes = np.empty((L, N, H, W), dtype=np.uint8)
    rc = lib().orc_label_decode_u64(_p(label, C.c_uint64), H, W, L, N,
                                    _p(planes, C.c_uint8))
    assert rc == 0, rc
    return planes


def box_decode_clip(anchors, deltas, std_dev, win_h, win_w):
    anchors = np.ascontiguousarray(anchors, dtype=np.float32)
    deltas = np.ascontiguousarray(deltas, dtype=np.float32)
    std = np.ascontiguousarray(std_dev, dtype=np.float32)
    out = np.empty_like(anchors)
    lib().orc_box_decode_clip_f32(_p(anchors, C.c_float), _p(deltas, C.c_float),
                                  C.c_int64(anchors.shape[0]), _p(std, C.c_float),
                                  C.c_float(win_h), C.c_float(win_w), _p(out, C.c_float))
    return out


def bbox_overlaps(b1, b2):
    b1 = np.ascontiguousarray(b1, dtype=np.float32).reshape(-1, 4)
    b2 = np.ascontiguousarray(b2, dtype=np.float32).reshape(-1, 4)
    out = np.empty((b1.shape[0], b2.shape[0]), dtype=np.float32)
    lib().orc_bbox_overlaps_f32(_p(b1, C.c_float), C.c_int64(b1.shape[0]),
                                _p(b2, C.c_float), C.c_int64(b2.shape[0]),
                                _p(out, C.c_float))
    return out


# --------------------------------------------------------------------------
# per-image graph functions (numpy restatements)
# --------------------------------------------------------------------------
def generate_pyramid_anchors(scales, ratios, feature_shapes, feature_strides,
                             anchor_stride):
    """reference: utils.py:472-528 (float64 numpy).  Pinned by golden anchors."""
    out = []
    for scale, shape, fstride in zip(scales, feature_shapes, feature_strides):
        r = np.asarray(ratios, dtype=np.float64)
        heights = scale / np.sqrt(r)
        widths = scale * np.sqrt(r)
        ys = np.arange(0, shape[0], anchor_stride) * fstride
        xs = np.arange(0, shape[1], anchor_stride) * fstride
        # order: (y, x, ratio) with ratio fastest  (utils.py:491-504)
        cy = np.repeat(ys, len(xs) * len(r)).astype(np.float64)
        cx = np.tile(np.repeat(xs, len(r)), len(ys)).astype(np.float64)
        hh = np.tile(heights, len(ys) * len(xs))
        ww = np.tile(widths, len(ys) * len(xs))
        out.append(np.stack([cy - 0.5 * hh, cx - 0.5 * ww, cy + 0.5 * hh, cx + 0.5 * ww],
                            axis=1))
    return np.concatenate(out, axis=0)


def proposal_layer(probs, deltas, anchors, proposal_count, nms_threshold,
                   std_dev=(0.1, 0.1, 0.2, 0.2), image_hw=(1024, 1024),
                   pre_nms_limit=6000):
    """One image.  probs [A,2], deltas [A,4], anchors [A,4] (pixels), fp32.
    reference: modal/Functions.py:114-178.  Score ties: stable (index
    ascending) -- see orc_nms_f32.  Returns normalised rois [<=count,4]."""
    probs = np.asarray(probs, dtype=np.float32)
    scores = probs[:, 1]
    order = np.argsort(-scores, kind="stable")[: min(pre_nms_limit, anchors.shape[0])]
    boxes = box_decode_clip(np.asarray(anchors, np.float32)[order],
                            np.asarray(deltas, np.float32)[order], std_dev,
                            float(image_hw[0]), float(image_hw[1]))
    dets = np.concatenate([boxes, scores[order][:, None]], axis=1)
    keep = nms(dets, nms_threshold)[:proposal_count]
    norm = np.array([image_hw[0], image_hw[1], image_hw[0], image_hw[1]], np.float32)
    return boxes[keep] / norm


def roi_levels(boxes, image_area):
    """FPN level per roi, reference modal/modals.py:51-64: fp32
    4 + log(sqrt(h*w) / (224/sqrt(area))) / log(2), round half-to-even (torch
    >= 1.1 semantics), clamp [2,5]."""
    import torch
    b = torch.as_tensor(np.asarray(boxes, np.float32)).view(-1, 4)
    y1, x1, y2, x2 = b.chunk(4, dim=1)
    h, w = y2 - y1, x2 - x1
    area = torch.tensor([float(image_area)], dtype=torch.float32)
    ln2 = torch.log(torch.tensor([2.0], dtype=torch.float32))
    lvl = 4 + torch.log(torch.sqrt(h * w) / (224.0 / torch.sqrt(area))) / ln2
    return lvl.round().int().clamp(2, 5).view(-1).numpy()


def pyramid_roi_align(boxes, feature_maps, pool, image_area):
    """boxes [R,4] normalised; feature_maps: list of 4 arrays [C,H,W] (P2..P5).
    reference: modal/modals.py:20-110.  Output [R,C,pool,pool] in roi order."""
    boxes = np.asarray(boxes, np.float32).reshape(-1, 4)
    lv = roi_levels(boxes, image_area)
    out = np.zeros((boxes.shape[0], feature_maps[0].shape[0], pool, pool), np.float32)
    for i, level in enumerate(range(2, 6)):
        ix = np.nonzero(lv == level)[0]
        if ix.size == 0:
            continue
        out[ix] = crop_and_resize_fwd(feature_maps[i][None], boxes[ix],
                                      np.zeros(ix.size, np.int32), pool, pool, 0.0)
    return out


def box_refinement(box, gt_box):
    """reference utils.py:96-117, fp32 (torch tensors there)."""
    box = np.asarray(box, np.float32)
    gt_box = np.asarray(gt_box, np.float32)
    h = box[:, 2] - box[:, 0]
    w = box[:, 3] - box[:, 1]
    cy = box[:, 0] + np.float32(0.5) * h
    cx = box[:, 1] + np.float32(0.5) * w
    gh = gt_box[:, 2] - gt_box[:, 0]
    gw = gt_box[:, 3] - gt_box[:, 1]
    gcy = gt_box[:, 0] + np.float32(0.5) * gh
    gcx = gt_box[:, 1] + np.float32(0.5) * gw
    return np.stack([(gcy - cy) / h, (gcx - cx) / w, np.log(gh / h), np.log(gw / w)],
                    axis=1).astype(np.float32)


def detection_target_layer(proposals, gt_class_ids, gt_boxes, gt_masks, perm_pos,
                           perm_neg, rois_per_image=100, positive_ratio=0.7,
                           bbox_std=(0.1, 0.1, 0.2, 0.2), mask_shape=(32, 32)):
    """One image, no crowds.  proposals [P,4] / gt_boxes [N,4] normalised fp32,
    gt_masks [L,N,H,W] u8.  perm_pos / perm_neg are the recorded
    torch.randperm draws (full permutations of the candidate sets).
    reference: modal/Functions.py:223-416.  Rounding: half-to-even."""
    proposals = np.asarray(proposals, np.float32).reshape(-1, 4)
    gt_boxes = np.asarray(gt_boxes, np.float32).reshape(-1, 4)
    ov = bbox_overlaps(proposals, gt_boxes)
    iou_max = ov.max(axis=1)
    pos_idx = np.nonzero(iou_max >= 0.5)[0]
    L = gt_masks.shape[0]
    empty = (np.zeros((0, 4), np.float32), np.zeros((0,), np.int32),
             np.zeros((0, 4), np.float32), np.zeros((0, L) + tuple(mask_shape), np.float32))
    pos_count = 0
    if pos_idx.size:
        want = int(rois_per_image * positive_ratio)
        perm_pos = np.asarray(perm_pos)
        pos_idx = pos_idx[perm_pos[perm_pos < pos_idx.size][:want]]   # (a longer permutation is restricted)
        pos_count = pos_idx.size
        pos_rois = proposals[pos_idx]
        assign = ov[pos_idx].argmax(axis=1)
        roi_gt = gt_boxes[assign]
        cls = np.asarray(gt_class_ids)[assign].astype(np.int32)
        deltas = box_refinement(pos_rois, roi_gt) / np.asarray(bbox_std, np.float32)
        masks = np.empty((pos_count, L) + tuple(mask_shape), np.float32)
        for l in range(L):
            planes = gt_masks[l][assign][:, None].astype(np.float32)  # [P,1,H,W]
            masks[:, l] = crop_and_resize_fwd(planes, pos_rois,
                                              np.arange(pos_count, dtype=np.int32),
                                              mask_shape[0], mask_shape[1], 0.0)[:, 0]
        masks = np.round(masks)  # numpy rounds half to even, like torch >= 1.1
    neg_idx = np.nonzero(iou_max < 0.5)[0]
    neg_count = 0
    if neg_idx.size and pos_count > 0:
        want = int((1.0 / positive_ratio) * pos_count - pos_count)
        perm_neg = np.asarray(perm_neg)
        neg_idx = neg_idx[perm_neg[perm_neg < neg_idx.size][:want]]
        neg_count = neg_idx.size
        neg_rois = proposals[neg_idx]
    if pos_count and neg_count:
        rois = np.concatenate([pos_rois, neg_rois])
        cls = np.concatenate([cls, np.zeros(neg_count, np.int32)])
        deltas = np.concatenate([deltas, np.zeros((neg_count, 4), np.float32)])
        masks = np.concatenate([masks, np.zeros((neg_count,) + masks.shape[1:], np.float32)])
        return rois, cls, deltas, masks
    if pos_count:
        return pos_rois, cls, deltas, masks
    return empty


def compute_overlaps_f64(boxes1, boxes2):
    """reference utils.py:58-94 (numpy float64, no +1)."""
    b1 = np.asarray(boxes1, np.float64)
    b2 = np.asarray(boxes2, np.float64)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    ov = np.zeros((b1.shape[0], b2.shape[0]))
    for i in range(b2.shape[0]):
        y1 = np.maximum(b2[i, 0], b1[:, 0]); y2 = np.minimum(b2[i, 2], b1[:, 2])
        x1 = np.maximum(b2[i, 1], b1[:, 1]); x2 = np.minimum(b2[i, 3], b1[:, 3])
        inter = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0)
        ov[:, i] = inter / (a2[i] + a1 - inter)
    return ov


def build_rpn_targets(anchors, gt_boxes, rng_choice, anchors_per_image=256,
                      std_dev=(0.1, 0.1, 0.2, 0.2)):
    """One image, no crowds.  reference: modal/Functions.py:739-847.
    rng_choice(ids, extra) must return the `extra` ids to reset (the recorded
    np.random.choice draws).  Returns rpn_match [A] i32, rpn_bbox [256,4] f64."""
    anchors = np.asarray(anchors, np.float64)
    gt_boxes = np.asarray(gt_boxes)
    rpn_match = np.zeros(anchors.shape[0], np.int32)
    rpn_bbox = np.zeros((anchors_per_image, 4))
    ov = compute_overlaps_f64(anchors, gt_boxes)
    amax_i = ov.argmax(axis=1)
    amax = ov[np.arange(ov.shape[0]), amax_i]
    rpn_match[amax < 0.3] = -1
    rpn_match[ov.argmax(axis=0)] = 1
    rpn_match[amax >= 0.7] = 1
    ids = np.where(rpn_match == 1)[0]
    extra = len(ids) - anchors_per_image // 2
    if extra > 0:
        rpn_match[rng_choice(ids, extra)] = 0
    ids = np.where(rpn_match == -1)[0]
    extra = len(ids) - (anchors_per_image - np.sum(rpn_match == 1))
    if extra > 0:
        rpn_match[rng_choice(ids, extra)] = 0
    ids = np.where(rpn_match == 1)[0]
    std = np.asarray(std_dev, np.float64)
    for ix, i in enumerate(ids):
        a = anchors[i]
        gt = gt_boxes[amax_i[i]]
        gh, gw = gt[2] - gt[0], gt[3] - gt[1]
        gcy, gcx = gt[0] + 0.5 * gh, gt[1] + 0.5 * gw
        ah, aw = a[2] - a[0], a[3] - a[1]
        acy, acx = a[0] + 0.5 * ah, a[1] + 0.5 * aw
        rpn_bbox[ix] = np.array([(gcy - acy) / ah, (gcx - acx) / aw,
                                 np.log(gh / ah), np.log(gw / aw)]) / std
    return rpn_match, rpn_bbox


def encode_labels(amodal_masks):
    """Synthetic-data helper: painter's order (object 0 on top) ->
    uint64 label, SURVEY.md section 8(d).  amodal_masks: [N,H,W] bool."""
    N = amodal_masks.shape[0]
    label = np.zeros(amodal_masks.shape[1:], np.uint64)
    covered = np.zeros(amodal_masks.shape[1:], bool)
    for i in range(N):
        m = amodal_masks[i]
        vis = m & ~covered
        label[vis] |= np.uint64(1) << np.uint64(i)
        label[m & covered] |= np.uint64(1) << np.uint64(32 + i)
        covered |= m
    return label


# --------------------------------------------------------------------------
# inference tail
# --------------------------------------------------------------------------
def build_ref(reference="/root/reference"):
    """oracle/_ref/libmaskapi_ref.so from the reference's own maskApi.c (plain C); returns the
    path, or None when neither the sources nor a prebuilt library are present."""
    so = os.path.join(_HERE, "_ref", "libmaskapi_ref.so")
    src = os.path.join(reference, "cocoapi", "common", "maskApi.c")
    if os.path.exists(src) and (not os.path.exists(so)
                                or os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "REFERENCE=" + reference, "ref"],
                              stdout=subprocess.DEVNULL)
    return so if os.path.exists(so) else None


class _RefRLE(C.Structure):
    # maskApi.h:13  typedef struct { siz h, w, m; uint *cnts; } RLE;
    _fields_ = [("h", C.c_ulong), ("w", C.c_ulong), ("m", C.c_ulong),
                ("cnts", C.POINTER(C.c_uint))]


_REF = None


def ref_maskapi():
    """ctypes handle of the compiled reference maskApi.c (None if unavailable)."""
    global _REF
    if _REF is None:
        so = build_ref()
        if so is None:
            return None
        _REF = C.CDLL(so)
        _REF.rleToString.restype = C.c_void_p
    return _REF


def ref_rle_encode(mask_hw):
    """The reference itself: rleEncode + rleToString on one [h,w] uint8 mask, through the same
    np.asfortranarray hand-off as amodal_train.py:397.  Returns (counts uint32[m], bytes)."""
    ref = ref_maskapi()
    m = np.asfortranarray(mask_hw.astype(np.uint8))
    h, w = m.shape
    R = _RefRLE()
    ref.rleEncode(C.byref(R), m.ctypes.data_as(C.POINTER(C.c_ubyte)), C.c_ulong(h),
                  C.c_ulong(w), C.c_ulong(1))
    cnts = np.ctypeslib.as_array(R.cnts, shape=(R.m,)).astype(np.uint32).copy() if R.m else \
        np.zeros(0, np.uint32)
    sp = ref.rleToString(C.byref(R))
    s = C.string_at(sp)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(sp)
    ref.rleFree(C.byref(R))
    return cnts, s


def rle_encode(mask_hw):
    """[h,w] mask -> run counts uint32[m], column-major scan (maskApi.c:33-42)."""
    m = np.asfortranarray(np.asarray(mask_hw).astype(np.uint8))
    flat = m.reshape(-1, order="F")
    cnts = np.empty(flat.size + 1, np.uint32)
    k = lib().orc_rle_encode_u8
    k.restype = C.c_int64
    n = k(_p(np.ascontiguousarray(flat), C.c_uint8), C.c_int64(flat.size), _p(cnts, C.c_uint32))
    return cnts[:n].copy()


def rle_decode(cnts, h, w):
    cnts = np.ascontiguousarray(cnts, np.uint32)
    assert int(cnts.astype(np.int64).sum()) == h * w
    flat = np.empty(h * w, np.uint8)
    lib().orc_rle_decode_u8(_p(cnts, C.c_uint32), C.c_int64(cnts.size), _p(flat, C.c_uint8))
    return flat.reshape(h, w, order="F")


def rle_to_string(cnts):
    """maskApi.c:204-216."""
    cnts = np.ascontiguousarray(cnts, np.uint32)
    buf = C.create_string_buffer(6 * cnts.size + 1)
    k = lib().orc_rle_to_string
    k.restype = C.c_int64
    n = k(_p(cnts, C.c_uint32), C.c_int64(cnts.size), buf)
    return buf.raw[:n]


def rle_from_string(s):
    """maskApi.c:218-231."""
    cnts = np.empty(max(len(s), 1), np.uint32)
    k = lib().orc_rle_from_string
    k.restype = C.c_int64
    n = k(C.c_char_p(bytes(s)), _p(cnts, C.c_uint32))
    return cnts[:n].copy()


def bytescale(data):
    """scipy.misc.bytescale of a float32 array (pilutil.py; float32 arithmetic)."""
    d = np.ascontiguousarray(data, np.float32)
    out = np.empty(d.shape, np.uint8)
    lib().orc_bytescale_f32(_p(d, C.c_float), C.c_int64(d.size), _p(out, C.c_uint8))
    return out


def pil_resize_bilinear(img_u8, oh, ow):
    """PIL Image.fromarray(img).resize((ow, oh), BILINEAR) restated (Resample.c)."""
    a = np.ascontiguousarray(img_u8, np.uint8)
    out = np.empty((oh, ow), np.uint8)
    lib().orc_pil_resize_bilinear_u8(_p(a, C.c_uint8), C.c_int(a.shape[0]), C.c_int(a.shape[1]),
                                     C.c_int(oh), C.c_int(ow), _p(out, C.c_uint8))
    return out


def unmold_mask(mask, bbox, image_shape):
    """utils.py:447-465.  mask [mh,mw] f32, bbox (y1,x1,y2,x2) ints -> [H,W] uint8."""
    m = np.ascontiguousarray(mask, np.float32)
    H, W = int(image_shape[0]), int(image_shape[1])
    full = np.empty((H, W), np.uint8)
    y1, x1, y2, x2 = (int(v) for v in bbox)
    rc = lib().orc_unmold_mask_f32(_p(m, C.c_float), C.c_int(m.shape[0]), C.c_int(m.shape[1]),
                                   C.c_int(y1), C.c_int(x1), C.c_int(y2), C.c_int(x2),
                                   C.c_int(H), C.c_int(W), _p(full, C.c_uint8))
    if rc != 0:
        raise ValueError("box outside the image")
    return full
