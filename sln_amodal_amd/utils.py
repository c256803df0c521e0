"""Host-side helpers with the reference's names (utils.py): anchors, box maths,
the label encoder.  numpy float64 where the reference is (anchors, rpn targets)."""
import numpy as np
import torch


def generate_anchors(scales, ratios, shape, feature_stride, anchor_stride):
    """Anchors of one pyramid level, (y1,x1,y2,x2) float64, ordered (y, x, ratio)
    with ratio fastest -- the order utils.py:472-508 produces."""
    scales = np.atleast_1d(np.asarray(scales, dtype=np.float64))
    r = np.asarray(ratios, dtype=np.float64)
    sc, rt = np.meshgrid(scales, r)
    sc, rt = sc.flatten(), rt.flatten()
    heights = sc / np.sqrt(rt)
    widths = sc * np.sqrt(rt)
    ys = np.arange(0, shape[0], anchor_stride) * feature_stride
    xs = np.arange(0, shape[1], anchor_stride) * feature_stride
    na = len(heights)
    cy = np.repeat(ys, len(xs) * na).astype(np.float64)
    cx = np.tile(np.repeat(xs, na), len(ys)).astype(np.float64)
    hh = np.tile(heights, len(ys) * len(xs))
    ww = np.tile(widths, len(ys) * len(xs))
    return np.stack([cy - 0.5 * hh, cx - 0.5 * ww, cy + 0.5 * hh, cx + 0.5 * ww], axis=1)


def generate_pyramid_anchors(scales, ratios, feature_shapes, feature_strides, anchor_stride):
    """All levels concatenated, scale[0]'s anchors first (utils.py:511-528)."""
    return np.concatenate([generate_anchors(scales[i], ratios, feature_shapes[i],
                                            feature_strides[i], anchor_stride)
                           for i in range(len(scales))], axis=0)


def box_refinement(box, gt_box):
    """Deltas that move `box` onto `gt_box`, torch tensors [N,4] (utils.py:96-117)."""
    h = box[:, 2] - box[:, 0]
    w = box[:, 3] - box[:, 1]
    cy = box[:, 0] + 0.5 * h
    cx = box[:, 1] + 0.5 * w
    gh = gt_box[:, 2] - gt_box[:, 0]
    gw = gt_box[:, 3] - gt_box[:, 1]
    gcy = gt_box[:, 0] + 0.5 * gh
    gcx = gt_box[:, 1] + 0.5 * gw
    return torch.stack([(gcy - cy) / h, (gcx - cx) / w, torch.log(gh / h), torch.log(gw / w)], dim=1)


def compute_overlaps(boxes1, boxes2):
    """IoU matrix [len(boxes1), len(boxes2)], numpy float64, no +1 (utils.py:58-94)."""
    b1 = np.asarray(boxes1, np.float64)
    b2 = np.asarray(boxes2, np.float64)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    y1 = np.maximum(b1[:, None, 0], b2[None, :, 0])
    x1 = np.maximum(b1[:, None, 1], b2[None, :, 1])
    y2 = np.minimum(b1[:, None, 2], b2[None, :, 2])
    x2 = np.minimum(b1[:, None, 3], b2[None, :, 3])
    inter = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0)
    return inter / (a2[None, :] + a1[:, None] - inter)


def reLayerMask(mask_amodal, mask_invis, min_size=64):
    """Encoder of the on-disk uint64 'layer' label (utils.py:531-547): low word bit i = object i
    visible, high word bit i = object i present but occluded; at most 32 objects; followed by the
    reference's small-region pruning (remove_small_path).  mask_invis[i] may be empty (no occlusion)."""
    label = np.zeros(np.asarray(mask_amodal[0]).shape, dtype=np.uint64)
    for i in range(min(len(mask_amodal), 32)):
        am = np.asarray(mask_amodal[i])
        if len(mask_invis[i]):
            inv = np.asarray(mask_invis[i])
            label[inv > 0] |= np.uint64(1) << np.uint64(i + 32)
            vis = (am - inv) > 0           # arithmetic difference, like the reference (utils.py:540)
        else:
            vis = am > 0
        label[vis] |= np.uint64(1) << np.uint64(i)
    return remove_small_path(label, min_size=min_size)


def remove_small_path(label, min_size=64):
    """utils.py:550-557: a label VALUE whose regions are all smaller than min_size pixels
    (4-connected components, skimage.morphology.remove_small_objects semantics) is erased everywhere;
    values with at least one large enough region are kept whole."""
    from scipy import ndimage
    for color in np.unique(label):
        mask = label == color
        comp, n = ndimage.label(mask)                      # connectivity 1, like skimage's default
        sizes = np.bincount(comp.ravel())[1:]
        if n == 0 or not (sizes >= min_size).any():
            label[mask] = 0
    return label


def labels_from_painter_order(amodal_masks):
    """Synthetic-data encoder: object 0 on top.  [N,H,W] bool."""
    label = np.zeros(amodal_masks.shape[1:], np.uint64)
    covered = np.zeros(amodal_masks.shape[1:], bool)
    for i in range(amodal_masks.shape[0]):
        m = amodal_masks[i]
        label[m & ~covered] |= np.uint64(1) << np.uint64(i)
        label[m & covered] |= np.uint64(1) << np.uint64(32 + i)
        covered |= m
    return label


_CONST_TENSORS = {}


def const_tensor(values, dtype, device):
    """A small constant as a device tensor, created once per (values, dtype, device).  torch.tensor(list,
    device="cuda") is a pageable host-to-device copy: the host waits for it, loses its lead over the GPU, and
    the queue runs dry (two 300-us holes per train step before this cache).  Do not modify the result."""
    import torch
    flat = tuple(np.asarray(values, dtype=np.float64).reshape(-1).tolist())
    key = (flat, tuple(np.shape(values)), dtype, str(device))
    t = _CONST_TENSORS.get(key)
    if t is None:
        t = _CONST_TENSORS[key] = torch.tensor(values, dtype=dtype, device=device)
    return t
