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


############################################################
#  Dataset side: image / label resize, jittered boxes (utils.py:28-54, 170-362)
############################################################

def zoom_nearest_index(n_in, n_out):
    """Source index of every output sample of scipy.ndimage.zoom(order=0, mode='constant') along one axis
    -- what utils.resize_layer (utils.py:358-362) applies to the label planes: output o reads input
    floor(o * step + 0.5) with step = (n_in - 1) / (n_out - 1) formed in float64 first (scipy computes it
    once and multiplies).  -1 = the constant fill (0): where the float64 product o * step lands above
    n_in - 1 -- it can, by one ulp, at the LAST sample -- scipy's 'constant' mode treats the coordinate
    as outside the array, so that row / column of the resized planes is empty.  Kept: the planes must
    equal the reference's, quirk included."""
    if n_out <= 0:
        return np.zeros(0, dtype=np.int64)
    if n_out == 1 or n_in <= 1:
        return np.zeros(n_out, dtype=np.int64)
    step = np.float64(n_in - 1) / np.float64(n_out - 1)
    cc = np.arange(n_out, dtype=np.float64) * step
    idx = np.floor(cc + 0.5).astype(np.int64)
    return np.where(cc > np.float64(n_in - 1), -1, np.clip(idx, 0, n_in - 1))


def take_zoom(a, ys, xs):
    """a[ys][:, xs] with -1 entries of the index maps reading the constant fill 0."""
    out = a[np.maximum(ys, 0)][:, np.maximum(xs, 0)]
    if (ys < 0).any():
        out[ys < 0] = 0
    if (xs < 0).any():
        out[:, xs < 0] = 0
    return out


def zoom_output_size(n_in, factor):
    """Length scipy.ndimage.zoom gives an axis of n_in samples: round(n_in * factor), half to even."""
    return int(round(n_in * factor))


def resize_image(image, min_dim=None, max_dim=None, padding=False):
    """utils.py:351-356: the image is SQUASHED to max_dim x max_dim by scipy.misc.imresize -- for a uint8
    image that is Pillow's BILINEAR resize (reducing-aware, 8-bit fixed point) -- whatever its aspect
    ratio; window = the whole square, scale = (max_dim / h, max_dim / w), no padding."""
    from PIL import Image
    h, w = image.shape[:2]
    u8 = np.ascontiguousarray(image)
    if u8.dtype != np.uint8:
        raise TypeError("resize_image expects the uint8 image the dataset loads")
    if (h, w) != (max_dim, max_dim):
        u8 = np.asarray(Image.fromarray(u8).resize((max_dim, max_dim), Image.BILINEAR))
    return u8, (0, 0, max_dim, max_dim), (max_dim / h, max_dim / w), [(0, 0), (0, 0), (0, 0)]


def resize_layer(mask, scale, padding=None):
    """utils.py:358-362: nearest-neighbour zoom of [H, W, ...] planes (or of the uint64 label itself: the
    decode is per pixel, so decoding the resized label equals resizing the decoded planes)."""
    h, w = mask.shape[:2]
    ys = zoom_nearest_index(h, zoom_output_size(h, scale[0]))
    xs = zoom_nearest_index(w, zoom_output_size(w, scale[1]))
    return take_zoom(mask, ys, xs)


def extract_bboxes(mask, jitter=None):
    """utils.py:28-54: tight boxes of [H, W, N] masks (y2 / x2 exclusive), each perturbed by
    (u * 2 - 1) * (h, w, h, w) / 15 with u = np.random.rand(4) drawn per instance IN ORDER (also for empty
    instances), negatives clamped to 0, truncated to int32.  jitter [N, 4]: replayed draws."""
    n = mask.shape[-1]
    boxes = np.zeros([n, 4], dtype=np.int32)
    for i in range(n):
        m = mask[:, :, i]
        cols = np.where(np.any(m, axis=0))[0]
        rows = np.where(np.any(m, axis=1))[0]
        if cols.shape[0]:
            x1, x2 = cols[[0, -1]]
            y1, y2 = rows[[0, -1]]
            x2 += 1
            y2 += 1
        else:
            x1, x2, y1, y2 = 0, 0, 0, 0
        u = np.random.rand(4) if jitter is None else np.asarray(jitter[i], np.float64)
        box = np.array([y1, x1, y2, x2]) + (u * 2 - 1) * (y2 - y1, x2 - x1, y2 - y1, x2 - x1) / 15
        box[box < 0] = 0
        boxes[i] = box
    return boxes.astype(np.int32)


def jitter_boxes(tight, u):
    """The same perturbation on tensors: tight [..., 4] integer boxes, u [..., 4] uniform draws (float64)
    -> int32 boxes; float64 arithmetic in the reference's order, so a replayed draw gives its box."""
    t = tight.to(torch.float64)
    hw = torch.stack([t[..., 2] - t[..., 0], t[..., 3] - t[..., 1]] * 2, dim=-1)
    box = t + (u.to(torch.float64) * 2 - 1) * hw / 15
    return torch.where(box < 0, torch.zeros_like(box), box).to(torch.int32)


class Dataset(object):
    """The slice of utils.Dataset (utils.py:170-300) the training path touches: an image list with
    `image_info` / `image_ids`, `load_image` (RGB uint8) and an empty `load_mask`."""

    def __init__(self, class_map=None):
        self._image_ids = []
        self.image_info = []
        self.class_info = [{"source": "", "id": 0, "name": "BG"}]

    def add_class(self, source, class_id, class_name):
        for info in self.class_info:
            if info["source"] == source and info["id"] == class_id:
                return
        self.class_info.append({"source": source, "id": class_id, "name": class_name})

    def add_image(self, source, image_id, path, **kwargs):
        info = {"id": image_id, "source": source, "path": path}
        info.update(kwargs)
        self.image_info.append(info)

    def prepare(self, class_map=None):
        self.num_classes = len(self.class_info)
        self.class_ids = np.arange(self.num_classes)
        self.num_images = len(self.image_info)
        self._image_ids = np.arange(self.num_images)

    @property
    def image_ids(self):
        return self._image_ids

    def source_image_link(self, image_id):
        return self.image_info[image_id]["path"]

    def load_image(self, image_id):
        from PIL import Image
        return np.asarray(Image.open(self.image_info[image_id]["path"]).convert("RGB"))

    def load_mask(self, image_id):
        return np.empty([0, 0, 0]), np.empty([0], np.int32)


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
