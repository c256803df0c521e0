"""Graph functions of the hot path with the reference's names
(modal/Functions.py): proposal_layer, bbox_overlaps, detection_target_layer,
refine_detections / detection_layer, build_rpn_targets, box helpers and the
sem-dist label decoders.

The reference versions are batch-1 and synchronise the host many times per image
(`.any()`, `nonzero`, `len`, `randperm` on the host).  These are batched over B
images, fixed-capacity and mask-based: a training step issues no device->host
copy.  Random sub-sampling takes explicit *priorities* (uniform random by
default), so the parity tests can replay the reference's recorded randperm /
np.random.choice draws through the very same code path.
"""
import numpy as np
import torch

from .. import ops, utils
from ..nms.nms_wrapper import nms  # noqa: F401  (reference import surface)


############################################################
#  Logging helpers
############################################################

def log(text, array=None):
    if array is not None:
        text = text.ljust(25) + "shape: {:20}  min: {:10.5f}  max: {:10.5f}".format(
            str(array.shape), array.min() if array.size else 0.0,
            array.max() if array.size else 0.0)
    print(text)


############################################################
#  Box helpers
############################################################

def apply_box_deltas(boxes, deltas):
    """boxes [N,4] (y1,x1,y2,x2), deltas [N,4] (dy,dx,log dh,log dw)
    (Functions.py:77-98)."""
    height = boxes[:, 2] - boxes[:, 0]
    width = boxes[:, 3] - boxes[:, 1]
    center_y = boxes[:, 0] + 0.5 * height
    center_x = boxes[:, 1] + 0.5 * width
    center_y = center_y + deltas[:, 0] * height
    center_x = center_x + deltas[:, 1] * width
    height = height * torch.exp(deltas[:, 2])
    width = width * torch.exp(deltas[:, 3])
    y1 = center_y - 0.5 * height
    x1 = center_x - 0.5 * width
    return torch.stack([y1, x1, y1 + height, x1 + width], dim=1)


def clip_boxes(boxes, window):
    """Clamp to window (y1,x1,y2,x2) (Functions.py:101-111)."""
    return torch.stack([boxes[:, 0].clamp(float(window[0]), float(window[2])),
                        boxes[:, 1].clamp(float(window[1]), float(window[3])),
                        boxes[:, 2].clamp(float(window[0]), float(window[2])),
                        boxes[:, 3].clamp(float(window[1]), float(window[3]))], 1)


def bbox_overlaps(boxes1, boxes2):
    """IoU without the +1 (Functions.py:184-218).  boxes1 [...,P,4], boxes2 [...,N,4]
    -> [...,P,N]; same operation order as the reference (fp32)."""
    b1 = boxes1.unsqueeze(-2)
    b2 = boxes2.unsqueeze(-3)
    y1 = torch.max(b1[..., 0], b2[..., 0])
    x1 = torch.max(b1[..., 1], b2[..., 1])
    y2 = torch.min(b1[..., 2], b2[..., 2])
    x2 = torch.min(b1[..., 3], b2[..., 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    return inter / (a1 + a2 - inter)


############################################################
#  Proposal Layer
############################################################

def proposal_layer(inputs, proposal_count, nms_threshold, anchors, config=None,
                   return_counts=False):
    """inputs = [rpn_probs [B,A,2], rpn_bbox [B,A,4]]; anchors [A,4] in pixels.
    Top PRE_NMS_LIMIT anchors by fg score -> decode + clip (HIP) -> NMS (HIP) ->
    first `proposal_count` -> normalise.  Returns rois [B,proposal_count,4], zero
    padded, like the reference's docstring promises (Functions.py:114-178), and
    optionally the per-image counts [B] int32."""
    probs, deltas = inputs[0], inputs[1]
    if probs.dim() == 2:
        probs, deltas = probs.unsqueeze(0), deltas.unsqueeze(0)
    A = anchors.shape[0]
    n = min(getattr(config, "PRE_NMS_LIMIT", 6000), A)
    # the reference's scores.sort(descending=True)[:6000] (Functions.py:133-147) with a defined
    # tie-break (lower anchor index first): a batched select + sort of the 6000 survivors only,
    # reading the foreground column in place
    fg = probs.detach()[:, :, 1]
    if fg.is_cuda and n <= 8192:
        order = ops.topk_order(fg, n)
    else:
        order = torch.sort(fg, dim=1, descending=True, stable=True)[1][:, :n].contiguous()
    height, width = config.IMAGE_SHAPE[:2]
    dets = ops.proposal_decode(probs.detach(), deltas.detach(), anchors, order,
                               config.RPN_BBOX_STD_DEV, float(height), float(width))
    keep, num = ops.nms_sorted(dets, nms_threshold, proposal_count)
    rois = ops.gather_rois(dets, keep, num, float(height), float(width))
    if return_counts:
        return rois, num
    return rois


############################################################
#  Detection Target Layer
############################################################

def _neg_count_table(rois_per_image, positive_ratio, device):
    """negative_count = int(r * p - p), r = 1/ratio, evaluated in Python floats
    exactly as Functions.py:356-357 does, for p = 0..max positives."""
    max_pos = int(rois_per_image * positive_ratio)
    r = 1.0 / positive_ratio
    return utils.const_tensor([int(r * p - p) for p in range(max_pos + 1)], torch.int64, device)


def priorities_from_draws(candidates, draws):
    """Recorded reference draws -> priorities.  candidates [B,P] bool (the set the reference
    permuted, in index order); draws[b] = the recorded torch.randperm(len(set)) of image b
    (Functions.py:291, 359).  The k-th drawn candidate gets priority -k, everything else -inf-like,
    so that descending priority visits the candidates in the reference's order.  Host-side helper for
    parity tests (it synchronises)."""
    B, P = candidates.shape
    pr = torch.full((B, P), -1e9, device=candidates.device)
    for b in range(B):
        idx = torch.nonzero(candidates[b])[:, 0]
        d = torch.as_tensor(draws[b], dtype=torch.long, device=candidates.device)
        if d.numel() != idx.numel():
            raise ValueError("image %d: %d recorded draws for %d candidates" % (b, d.numel(), idx.numel()))
        pr[b, idx[d]] = -torch.arange(d.numel(), device=candidates.device, dtype=torch.float32)
    return pr


def detection_target_layer(proposals, gt_class_ids, gt_boxes, gt_masks, config,
                           num_proposals=None, labels=None, priority_pos=None,
                           priority_neg=None, replay=None):
    """Sub-sample proposals and build their targets, batched and sync-free.

    proposals    [B,P,4] normalised, zero padded; num_proposals [B] or None (=P)
    gt_class_ids [B,N] int (0 = padding; crowds (<0) are not supported)
    gt_boxes     [B,N,4] normalised
    gt_masks     [B,L,N,H,W] uint8 planes, or None when `labels` is given
    labels       [B,H,W] int64 bit patterns: masks are produced by the fused
                 decode+crop+round kernel (never materialising float planes)
    priority_*   [B,P] float: candidates are taken in descending priority
                 (default uniform random == the reference's randperm prefix)
    replay       (draws_pos, draws_neg): per-image recorded torch.randperm draws of the
                 reference, replayed through the same priority path (parity tests)

    Returns dict(rois [B,R,4], class_ids [B,R] int32, deltas [B,R,4],
    masks [B,R,L,h,w], roi_valid [B,R] bool, gt_assign [B,R]).  R =
    TRAIN_ROIS_PER_IMAGE slots: positives first (<= R*ratio), then negatives; the
    valid slots, in order, are exactly the reference's output rows
    (Functions.py:223-416)."""
    B, P = proposals.shape[0], proposals.shape[1]
    dev = proposals.device
    R = config.TRAIN_ROIS_PER_IMAGE
    max_pos = int(R * config.ROI_POSITIVE_RATIO)
    max_neg = R - max_pos
    if num_proposals is None:
        prop_valid = torch.ones((B, P), dtype=torch.bool, device=dev)
    else:
        prop_valid = torch.arange(P, device=dev)[None, :] < num_proposals[:, None]
    gt_valid = gt_class_ids > 0
    ov = bbox_overlaps(proposals, gt_boxes)                              # [B,P,N]
    ov = torch.where(gt_valid[:, None, :] & prop_valid[:, :, None], ov, torch.full_like(ov, -1.0))
    iou_max, assign = ov.max(dim=2)
    pos = (iou_max >= 0.5) & prop_valid
    neg = (iou_max < 0.5) & prop_valid & gt_valid.any(dim=1, keepdim=True)
    if replay is not None:
        priority_pos = priorities_from_draws(pos, replay[0])
        priority_neg = priorities_from_draws(neg, replay[1])
    if priority_pos is None:
        priority_pos = torch.rand((B, P), device=dev)
    if priority_neg is None:
        priority_neg = torch.rand((B, P), device=dev)
    ninf = torch.full((B, P), float("-inf"), device=dev)
    kp = min(max_pos, P)
    kn = min(max_neg, P)
    pv, pidx = torch.topk(torch.where(pos, priority_pos, ninf), kp, dim=1)
    nv, nidx = torch.topk(torch.where(neg, priority_neg, ninf), kn, dim=1)
    pos_count = pos.sum(dim=1).clamp(max=max_pos)
    neg_want = _neg_count_table(R, config.ROI_POSITIVE_RATIO, dev)[pos_count]
    neg_count = torch.minimum(neg.sum(dim=1), neg_want)
    pos_slot = torch.arange(kp, device=dev)[None, :] < pos_count[:, None]
    neg_slot = torch.arange(kn, device=dev)[None, :] < neg_count[:, None]

    pos_rois = torch.gather(proposals, 1, pidx.unsqueeze(2).expand(-1, -1, 4))
    neg_rois = torch.gather(proposals, 1, nidx.unsqueeze(2).expand(-1, -1, 4))
    pos_assign = torch.gather(assign, 1, pidx)
    roi_gt = torch.gather(gt_boxes, 1, pos_assign.unsqueeze(2).expand(-1, -1, 4))
    roi_cls = torch.gather(gt_class_ids, 1, pos_assign).to(torch.int32)
    std = utils.const_tensor(np.asarray(config.BBOX_STD_DEV), torch.float32, dev)
    unit = utils.const_tensor([0., 0., 1., 1.], torch.float32, dev)
    safe_rois = torch.where(pos_slot.unsqueeze(2), pos_rois, unit)
    safe_gt = torch.where(pos_slot.unsqueeze(2), roi_gt, unit)
    deltas = utils.box_refinement(safe_rois.reshape(-1, 4), safe_gt.reshape(-1, 4)).view(B, kp, 4) / std

    mh, mw = config.MASK_SHAPE[0], config.MASK_SHAPE[1]
    roi_img = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(kp)
    roi_img = torch.where(pos_slot.reshape(-1), roi_img, torch.full_like(roi_img, -1))
    if labels is not None:
        L = config.NUM_CLASSES - 1
        masks = ops.mask_targets(labels, L, pos_rois.reshape(-1, 4), roi_img,
                                 pos_assign.reshape(-1).to(torch.int32), mh, mw)
    else:
        L, N = gt_masks.shape[1], gt_masks.shape[2]
        from ..roialign.roi_align.crop_and_resize import CropAndResizeFunction
        flat_idx = (torch.arange(B, device=dev)[:, None] * N + pos_assign).reshape(-1)
        safe_ind = torch.where(pos_slot.reshape(-1), flat_idx,
                               torch.full_like(flat_idx, -1)).to(torch.int32)
        planes = []
        for l in range(L):
            img = gt_masks[:, l].reshape(B * N, 1, gt_masks.shape[3], gt_masks.shape[4]).float()
            planes.append(CropAndResizeFunction(mh, mw, 0)(img, pos_rois.reshape(-1, 4).contiguous(),
                                                           safe_ind.contiguous())[:, 0])
        masks = torch.round(torch.stack(planes, dim=1))
    masks = masks.view(B, kp, L, mh, mw) * pos_slot[:, :, None, None, None].float()

    # pack: positives first, then negatives, contiguous per image (reference row order)
    slot = torch.arange(R, device=dev)[None, :]
    is_pos = slot < pos_count[:, None]
    is_neg = (slot >= pos_count[:, None]) & (slot < (pos_count + neg_count)[:, None])
    src_pos = slot.clamp(max=kp - 1).expand(B, R)
    src_neg = (slot - pos_count[:, None]).clamp(min=0, max=kn - 1)
    g4 = lambda t, i: torch.gather(t, 1, i.unsqueeze(2).expand(-1, -1, 4))
    zero4 = torch.zeros((B, R, 4), device=dev)
    rois = torch.where(is_pos.unsqueeze(2), g4(pos_rois, src_pos),
                       torch.where(is_neg.unsqueeze(2), g4(neg_rois, src_neg), zero4))
    out_deltas = torch.where(is_pos.unsqueeze(2), g4(deltas, src_pos), zero4)
    cls = torch.where(is_pos, torch.gather(roi_cls, 1, src_pos), torch.zeros((B, R), dtype=torch.int32, device=dev))
    m_src = src_pos[:, :, None, None, None].expand(-1, -1, L, mh, mw)
    out_masks = torch.gather(masks, 1, m_src) * is_pos[:, :, None, None, None].float()
    gt_assign = torch.where(is_pos, torch.gather(pos_assign, 1, src_pos), torch.full((B, R), -1, device=dev))
    return {"rois": rois, "class_ids": cls, "deltas": out_deltas, "masks": out_masks,
            "roi_valid": is_pos | is_neg, "gt_assign": gt_assign}


############################################################
#  Detection Layer (inference)
############################################################

def clip_to_window(window, boxes):
    return clip_boxes(boxes, window)


def coordinate_convert(rois, deltas_specific, config, use_cuda=False):
    """Apply class-specific deltas (scaled by RPN_BBOX_STD_DEV, as the reference
    does at inference, Functions.py:436-450) and go to pixel coordinates."""
    std = utils.const_tensor(np.reshape(config.RPN_BBOX_STD_DEV, [1, 4]), torch.float32, rois.device)
    refined = apply_box_deltas(rois, deltas_specific * std)
    height, width = config.IMAGE_SHAPE[:2]
    scale = utils.const_tensor([height, width, height, width], torch.float32, rois.device)
    return refined * scale


def refine_detections(rois, probs, deltas, window, config):
    """rois [N,4] normalised, probs [N,C], deltas [N,C,4] -> detections
    [M,(y1,x1,y2,x2,class_id,score)] in pixels + kept roi indices
    (Functions.py:453-557).  USE_NMS=False (the reference default): foreground
    rois, top 100 by score."""
    _, class_ids = torch.max(probs, dim=1)
    idx = torch.arange(class_ids.shape[0], device=rois.device)
    class_scores = probs[idx, class_ids]
    deltas_specific = deltas[idx, class_ids]
    refined = coordinate_convert(rois, deltas_specific, config)
    refined = clip_to_window(window, refined)
    refined = torch.round(refined)
    keep_bool = class_ids > 0
    if config.USE_NMS:
        if config.DETECTION_MIN_CONFIDENCE:
            keep_bool = keep_bool & (class_scores >= config.DETECTION_MIN_CONFIDENCE)
        keep = torch.nonzero(keep_bool)[:, 0]
        if keep.numel() == 0:
            return [], []
        kept = []
        for class_id in torch.unique(class_ids[keep]):
            ixs = keep[class_ids[keep] == class_id]
            sc, order = class_scores[ixs].sort(descending=True, stable=True)
            ck = nms(torch.cat((refined[ixs][order], sc.unsqueeze(1)), dim=1),
                     config.DETECTION_NMS_THRESHOLD)
            kept.append(ixs[order[ck]])
        keep = torch.unique(torch.cat(kept))
    else:
        keep = torch.nonzero(keep_bool).view(-1)
        if keep.numel() > 100:
            order = class_scores[keep].sort(descending=True, stable=True)[1]
            keep = keep[order[:100]]
    if keep.numel() == 0:
        return [], []
    order = class_scores[keep].sort(descending=True, stable=True)[1]
    keep = keep[order]
    result = torch.cat((refined[keep], class_ids[keep].unsqueeze(1).float(),
                        class_scores[keep].unsqueeze(1)), dim=1)
    return result, keep


def refine_detections_batched(rois, valid, probs, deltas, window, config, max_instances=100):
    """refine_detections (Functions.py:453-557, USE_NMS = False: the reference default) for B images at once,
    fixed capacity, no host sync: rois [B,R,4] normalised with `valid` [B,R] bool (proposal slots beyond an
    image's count are not detections), probs [B,R,C], deltas [B,R,C,4], window (y1,x1,y2,x2) pixels -- one
    for all images or [B,4], one per image.
    Per image exactly the reference's choice -- foreground rois, the 100 best scores, descending, ties in roi
    order (its two stable sorts) -- as rows [B,100,(y1,x1,y2,x2,class_id,score)], ZERO rows behind the
    per-image count (unmold_detections stops at the first class id 0, model.py:762-764), and counts [B]."""
    B, R = rois.shape[0], rois.shape[1]
    class_scores, class_ids = torch.max(probs, dim=2)
    idx = class_ids.unsqueeze(2).unsqueeze(3).expand(B, R, 1, 4)
    deltas_specific = torch.gather(deltas, 2, idx).squeeze(2)
    refined = coordinate_convert(rois.reshape(-1, 4), deltas_specific.reshape(-1, 4), config).view(B, R, 4)
    # every image is clipped to ITS OWN window (the reference runs one image per call, Functions.py:483)
    wn = np.asarray(window, dtype=np.float32).reshape(-1, 4)
    if (wn == wn[:1]).all():       # one window for all images (what mold_inputs produces): a cached device constant
        win = utils.const_tensor(wn[:1].tolist(), torch.float32, rois.device)
    else:                          # (distinct windows per image: not cached -- the cache would grow with the data)
        win = torch.as_tensor(wn, device=rois.device)
    if win.shape[0] not in (1, B):
        raise ValueError("refine_detections_batched: %d windows for %d images" % (win.shape[0], B))
    win = win.expand(B, 4).unsqueeze(1)                      # [B,1,4]
    lo = torch.cat((win[..., 0:2], win[..., 0:2]), dim=2)
    hi = torch.cat((win[..., 2:4], win[..., 2:4]), dim=2)
    refined = torch.round(torch.minimum(torch.maximum(refined, lo), hi))
    keep = (class_ids > 0) & valid
    if config.DETECTION_MIN_CONFIDENCE and config.USE_NMS:
        keep = keep & (class_scores >= config.DETECTION_MIN_CONFIDENCE)
    key = torch.where(keep, class_scores, torch.full_like(class_scores, float("-inf")))
    order = torch.sort(key, dim=1, descending=True, stable=True)[1][:, :max_instances]
    count = keep.sum(dim=1).clamp(max=max_instances).to(torch.int32)
    live = torch.arange(order.shape[1], device=rois.device)[None, :] < count[:, None]
    g = lambda t: torch.gather(t, 1, order)
    det = torch.cat((torch.gather(refined, 1, order.unsqueeze(2).expand(-1, -1, 4)),
                     g(class_ids).unsqueeze(2).float(), g(class_scores).unsqueeze(2)), dim=2)
    det = torch.where(live.unsqueeze(2), det, torch.zeros_like(det))
    if det.shape[1] < max_instances:
        det = torch.cat((det, det.new_zeros(B, max_instances - det.shape[1], 6)), dim=1)
        live = torch.cat((live, live.new_zeros(B, max_instances - live.shape[1])), dim=1)
    return det, live, count


def parse_image_meta(meta):
    return meta[:, 0], meta[:, 1:4], meta[:, 4:8], meta[:, 8:]


def compose_image_meta(image_id, image_shape, window, active_class_ids):
    return np.array([image_id] + list(image_shape) + list(window) + list(active_class_ids))


def detection_layer(config, rois, mrcnn_class, mrcnn_bbox, image_meta):
    """Batch-1 inference tail (Functions.py:560-575)."""
    rois = rois.squeeze(0)
    _, _, window, _ = parse_image_meta(image_meta)
    return refine_detections(rois, mrcnn_class, mrcnn_bbox, window[0], config)


def mold_image(images, config):
    return images.astype(np.float32) - config.MEAN_PIXEL


def unmold_image(normalized_images, config):
    return (normalized_images + config.MEAN_PIXEL).astype(np.uint8)


############################################################
#  Data generator (host side, one image: Functions.py:675-736)
############################################################

def label_planes_host(layer, num_layers):
    """AmodalDataset.load_layer2 (amodal_train.py:236-271) on the host, closed form per pixel: object i
    (count = max_objectID, Functions.py:1074-1079: the first shift s at which no label's low word has its
    HIGHEST set bit at s -- the highest low-word bit + 1 unless the tops have a gap) is visible where bit i
    of the label is set -> plane 0; where bit 32 + i is set it is occluded, at depth
    rank = popcount(high word & ((1 << i) - 1)) -> plane min(rank + 1, L - 1).
    layer [H,W] uint64 -> ([H,W,L,N] bool, class_ids [N] int32 all 1) like the reference."""
    layer = np.ascontiguousarray(layer).astype(np.uint64)
    lo = (layer & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (layer >> np.uint64(32)).astype(np.uint32)
    tops = set(int(v).bit_length() - 1 for v in np.unique(lo) if v)
    n = 0
    while n in tops:
        n += 1
    H, W = layer.shape
    L = max(int(num_layers), 1)
    planes = np.zeros((H, W, L, n), dtype=bool)
    for i in range(n):
        planes[:, :, 0, i] |= ((lo >> np.uint32(i)) & np.uint32(1)).astype(bool)
        occ = ((hi >> np.uint32(i)) & np.uint32(1)).astype(bool)
        if occ.any():
            below = hi & np.uint32((1 << i) - 1)
            rank = np.zeros(below.shape, dtype=np.int64)
            for b in range(i):
                rank += (below >> np.uint32(b)) & np.uint32(1)
            tgt = np.minimum(rank + 1, L - 1)
            for l in range(L):
                planes[:, :, l, i] |= occ & (tgt == l)
    return planes, np.ones([n], dtype=np.int32)


def load_image_gt(dataset, config, image_id, augment=False, use_mini_mask=False, draws=None):
    """Functions.py:675-736 for one image: dataset.load_image + dataset.load_layer2, the squash to
    IMAGE_MAX_DIM^2 (utils.resize_image / resize_layer), the random horizontal flip
    (`random.randint(0, 1)`, Functions.py:713), jittered boxes of the amodal masks (utils.extract_bboxes,
    np.random.rand(4) per instance) and the image meta.  draws = {"flip": 0|1, "jitter": [N,4]} replays
    recorded draws.  Returns (image uint8 [D,D,3], image_meta, class_ids [N], bbox int32 [N,4],
    mask_layers uint8 [D,D,N,L]) -- what the reference returns."""
    import random
    from .. import utils
    if use_mini_mask:
        raise NotImplementedError("USE_MINI_MASK is False on the amodal path (config.py)")
    image = dataset.load_image(image_id)
    mask_layers, class_ids = dataset.load_layer2(image_id, config)
    shape = image.shape
    image, window, scale, padding = utils.resize_image(image, min_dim=config.IMAGE_MIN_DIM,
                                                       max_dim=config.IMAGE_MAX_DIM,
                                                       padding=config.IMAGE_PADDING)
    mask_layers = utils.resize_layer(mask_layers, scale, padding)
    if augment:
        flip = random.randint(0, 1) if draws is None else int(draws["flip"])
        if flip:
            image = np.fliplr(image)
            mask_layers = np.fliplr(mask_layers)
    amodal_mask = np.sum(mask_layers, axis=2)
    bbox = utils.extract_bboxes(amodal_mask, None if draws is None else draws["jitter"])
    active_class_ids = np.ones([128], dtype=np.int32)
    image_meta = compose_image_meta(image_id, shape, window, active_class_ids)
    mask_layers = (np.swapaxes(mask_layers, 2, 3) > 0).astype("uint8")
    return image, image_meta, class_ids, bbox, mask_layers


############################################################
#  RPN targets
############################################################

def build_rpn_targets(image_shape, anchors, gt_class_ids, gt_boxes, config, priority=None):
    """Anchor matching on the device, float64 like the reference's numpy
    (Functions.py:739-847).  anchors [A,4] float64 pixels; gt_boxes [B,N,4] pixels;
    gt_class_ids [B,N] (0 = padding).  priority [B,A] float: anchors are *kept* in
    descending priority when positives / negatives are sub-sampled (default
    uniform random == np.random.choice without replacement).
    Returns rpn_match [B,A] int32 and rpn_bbox [B,T,4] float32 (row k = k-th positive
    anchor in anchor order)."""
    dev = gt_boxes.device
    T = config.RPN_TRAIN_ANCHORS_PER_IMAGE
    a = anchors.to(torch.float64)
    g = gt_boxes.to(torch.float64)
    B, A = g.shape[0], a.shape[0]
    gt_valid = gt_class_ids > 0
    a_area = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    g_area = (g[..., 2] - g[..., 0]) * (g[..., 3] - g[..., 1])
    y1 = torch.maximum(g[:, None, :, 0], a[None, :, None, 0])
    y2 = torch.minimum(g[:, None, :, 2], a[None, :, None, 2])
    x1 = torch.maximum(g[:, None, :, 1], a[None, :, None, 1])
    x2 = torch.minimum(g[:, None, :, 3], a[None, :, None, 3])
    inter = (x2 - x1).clamp(min=0) * (y2 - y1).clamp(min=0)
    ov = inter / (g_area[:, None, :] + a_area[None, :, None] - inter)          # [B,A,N]
    ov = torch.where(gt_valid[:, None, :], ov, torch.full_like(ov, -1.0))
    amax, amax_i = ov.max(dim=2)
    match = torch.zeros((B, A), dtype=torch.int32, device=dev)
    match[amax < 0.3] = -1
    best_anchor = ov.argmax(dim=1)                                              # [B,N]
    bidx = torch.arange(B, device=dev)[:, None].expand_as(best_anchor)
    match[bidx[gt_valid], best_anchor[gt_valid]] = 1
    match[amax >= 0.7] = 1
    if priority is None:
        priority = torch.rand((B, A), device=dev)
    ninf = torch.full((B, A), float("-inf"), device=dev, dtype=priority.dtype)

    def keep_top(mask, limit):
        # rank of each candidate by descending priority; keep rank < limit
        pr = torch.where(mask, priority, ninf)
        order = torch.argsort(pr, dim=1, descending=True, stable=True)
        rank = torch.empty_like(order)
        rank.scatter_(1, order, torch.arange(A, device=dev)[None, :].expand(B, A))
        return mask & (rank < limit[:, None])

    pos = match == 1
    pos_keep = keep_top(pos, torch.full((B,), T // 2, device=dev))
    match = torch.where(pos & ~pos_keep, torch.zeros_like(match), match)
    neg = match == -1
    neg_keep = keep_top(neg, T - (match == 1).sum(dim=1))
    match = torch.where(neg & ~neg_keep, torch.zeros_like(match), match)

    pos = match == 1
    rank = torch.cumsum(pos.long(), dim=1) - 1
    gsel = torch.gather(g, 1, amax_i.unsqueeze(2).expand(-1, -1, 4))            # [B,A,4]
    gh, gw = gsel[..., 2] - gsel[..., 0], gsel[..., 3] - gsel[..., 1]
    gcy, gcx = gsel[..., 0] + 0.5 * gh, gsel[..., 1] + 0.5 * gw
    ah, aw = (a[:, 2] - a[:, 0])[None], (a[:, 3] - a[:, 1])[None]
    acy, acx = (a[:, 0] + 0.5 * (a[:, 2] - a[:, 0]))[None], (a[:, 1] + 0.5 * (a[:, 3] - a[:, 1]))[None]
    std = torch.as_tensor(np.asarray(config.RPN_BBOX_STD_DEV), dtype=torch.float64, device=dev)
    d = torch.stack([(gcy - acy) / ah, (gcx - acx) / aw, torch.log(gh / ah), torch.log(gw / aw)],
                    dim=2) / std
    rpn_bbox = torch.zeros((B, T + 1, 4), dtype=torch.float64, device=dev)
    dst = torch.where(pos & (rank < T), rank, torch.full_like(rank, T))          # T = spill row
    rpn_bbox.scatter_(1, dst.unsqueeze(2).expand(-1, -1, 4), torch.where(pos.unsqueeze(2), d, torch.zeros_like(d)))
    return match, rpn_bbox[:, :T].float()


############################################################
#  Sem-dist label decoders (data decoder section of the reference)
############################################################

def max_objectID(labels):
    """labels [B,H,W] int64 bit patterns on the GPU -> object count per image."""
    return ops.label_num_objects(labels)


def decode_layers(labels, num_layers, num_objects):
    """labels [B,H,W] -> [B,L,N,H,W] uint8 planes (load_layer2 + axis shuffle)."""
    return ops.label_decode(labels, num_layers, num_objects)


def extract_bboxes_from_labels(labels, num_objects):
    """Tight amodal boxes [B,N,4] (y1,x1,y2,x2; x2,y2 exclusive) straight from the
    label bits, without the reference's random jitter (utils.py:28-54 adds
    +-1/15 noise; SURVEY.md 8(d) specifies jitter-free boxes for the benchmark).
    Objects absent from the image get zeros."""
    B, H, W = labels.shape
    dev = labels.device
    bits = torch.arange(num_objects, device=dev, dtype=torch.int64)
    present = (((labels.unsqueeze(1) >> bits[None, :, None, None]) |
                (labels.unsqueeze(1) >> (bits[None, :, None, None] + 32))) & 1).bool()  # [B,N,H,W]
    rows = present.any(dim=3)
    cols = present.any(dim=2)
    ar_h = torch.arange(H, device=dev)
    ar_w = torch.arange(W, device=dev)
    big = 1 << 30
    y1 = torch.where(rows, ar_h, torch.full_like(ar_h, big)).amin(dim=2)
    y2 = torch.where(rows, ar_h, torch.full_like(ar_h, -1)).amax(dim=2) + 1
    x1 = torch.where(cols, ar_w, torch.full_like(ar_w, big)).amin(dim=2)
    x2 = torch.where(cols, ar_w, torch.full_like(ar_w, -1)).amax(dim=2) + 1
    boxes = torch.stack([y1, x1, y2, x2], dim=2)
    return torch.where(rows.any(dim=2, keepdim=True), boxes, torch.zeros_like(boxes))
