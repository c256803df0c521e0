"""Losses of the hot path with the reference's names (modal/loss.py:10-152).

The reference is batch-1 and selects rows with torch.nonzero (a host sync per
loss).  These versions take a batch of B images with fixed-capacity, masked
tensors and compute, per image, exactly the reference's mean over the selected
rows; the step loss is the mean over images that have at least one positive roi
(the reference skips such images, model.py:423-434).  With B = 1 and all-valid
inputs each function equals its reference counterpart (tests/test_model_cpu.py
pins them against golden vectors produced by the reference code).
"""
import torch
import torch.nn.functional as F


def _masked_mean(values, mask, dims):
    """mean of `values` over `dims` restricted to mask==True; count clamped to 1."""
    m = mask.to(values.dtype)
    cnt = m.sum(dim=dims)
    return (values * m).sum(dim=dims) / cnt.clamp(min=1.0), cnt


def compute_rpn_class_loss(rpn_match, rpn_class_logits, reduce=True):
    """rpn_match [B,A,1] (1 pos, -1 neg, 0 neutral); logits [B,A,2].
    Cross-entropy over the non-neutral anchors of each image (loss.py:10-35)."""
    match = rpn_match.squeeze(2)
    target = (match == 1).long()
    ce = F.cross_entropy(rpn_class_logits.reshape(-1, 2), target.reshape(-1),
                         reduction="none").view_as(match)
    per_image, _ = _masked_mean(ce, match != 0, dims=1)
    return per_image.mean() if reduce else per_image


def compute_rpn_bbox_loss(target_bbox, rpn_match, rpn_bbox, reduce=True):
    """target_bbox [B,T,4]: row k belongs to the k-th positive anchor of that image
    (anchor order); rpn_bbox [B,A,4].  Smooth-L1, mean over positives x 4
    (loss.py:37-63)."""
    match = rpn_match.squeeze(2)
    pos = match == 1
    T = target_bbox.shape[1]
    if rpn_bbox.is_cuda and T <= 8192 and T <= match.shape[1]:
        # The positives are at most T of ~262 k anchors: select them (anchor order) with the top-k kernel
        # -- ties of the 0/1 score break by the lower index -- and evaluate the loss on [B,T,4] instead
        # of scanning and masking every anchor (two int64 cumsums + a [B,A,4] smooth-L1 otherwise).
        from .. import ops
        idx = ops.topk_order(pos.to(torch.float32), T)                     # [B,T] positives first
        npos = pos.sum(dim=1, keepdim=True).clamp(max=T)
        valid = torch.arange(T, device=match.device).unsqueeze(0) < npos   # [B,T]
        own = torch.gather(rpn_bbox, 1, idx.unsqueeze(2).expand(-1, -1, 4))
        sl1 = F.smooth_l1_loss(own, target_bbox, reduction="none")
        per_image, _ = _masked_mean(sl1, valid.unsqueeze(2).expand_as(sl1), dims=(1, 2))
        return per_image.mean() if reduce else per_image
    csum = torch.cumsum(pos.int(), dim=1) - 1
    rank = csum.clamp(min=0, max=T - 1).long()
    tgt = torch.gather(target_bbox, 1, rank.unsqueeze(2).expand(-1, -1, 4))
    sl1 = F.smooth_l1_loss(rpn_bbox, tgt, reduction="none")
    pos = pos & (csum < T)
    per_image, _ = _masked_mean(sl1, pos.unsqueeze(2).expand_as(sl1), dims=(1, 2))
    return per_image.mean() if reduce else per_image


def compute_mrcnn_class_loss(target_class_ids, pred_class_logits, roi_valid=None, reduce=True):
    """target_class_ids [B,R]; logits [B,R,num_classes]; CE mean over the valid rois
    of each image (loss.py:66-82)."""
    if target_class_ids.dim() == 1:
        target_class_ids = target_class_ids.unsqueeze(0)
        pred_class_logits = pred_class_logits.unsqueeze(0)
    if roi_valid is None:
        roi_valid = torch.ones_like(target_class_ids, dtype=torch.bool)
    B, R = target_class_ids.shape
    ce = F.cross_entropy(pred_class_logits.reshape(B * R, -1), target_class_ids.reshape(-1).long(),
                         reduction="none").view(B, R)
    per_image, _ = _masked_mean(ce, roi_valid, dims=1)
    return per_image.mean() if reduce else per_image


def compute_mrcnn_bbox_loss(target_bbox, target_class_ids, pred_bbox, reduce=True):
    """target_bbox [B,R,4]; pred_bbox [B,R,num_classes,4]; smooth-L1 on the
    positive rois' own-class deltas, mean over positives x 4 (loss.py:85-111)."""
    if target_class_ids.dim() == 1:
        target_class_ids = target_class_ids.unsqueeze(0)
        target_bbox = target_bbox.unsqueeze(0)
        pred_bbox = pred_bbox.unsqueeze(0)
    pos = target_class_ids > 0
    cls = target_class_ids.clamp(min=0).long()
    own = torch.gather(pred_bbox, 2, cls[:, :, None, None].expand(-1, -1, 1, 4)).squeeze(2)
    sl1 = F.smooth_l1_loss(own, target_bbox, reduction="none")
    per_image, _ = _masked_mean(sl1, pos.unsqueeze(2).expand_as(sl1), dims=(1, 2))
    return per_image.mean() if reduce else per_image


def _bce(pred_prob, target):
    return F.binary_cross_entropy(pred_prob, target, reduction="none")


def compute_amodal_loss(target_masks, target_class_ids, pred_masks, reduce=True):
    """BCE(sigmoid(sum_l pred[:,1:]), sum_l target), mean over positives x h x w
    (loss.py:113-124).  target [B,R,L,h,w]; pred [B,R,1+L,h,w]."""
    if target_class_ids.dim() == 1:
        target_class_ids = target_class_ids.unsqueeze(0)
        target_masks = target_masks.unsqueeze(0)
        pred_masks = pred_masks.unsqueeze(0)
    pos = target_class_ids > 0
    y_true = target_masks.sum(dim=2)
    y_pred = torch.sigmoid(pred_masks[:, :, 1:].sum(dim=2))
    bce = _bce(y_pred, y_true)
    per_image, _ = _masked_mean(bce, pos[:, :, None, None].expand_as(bce), dims=(1, 2, 3))
    if reduce:
        return per_image.mean(), y_pred
    return per_image, y_pred


def compute_layer_loss(target_masks, target_class_ids, pred_masks, reduce=True):
    """BCE(sigmoid(pred[:,1:]), target), mean over positives x L x h x w
    (loss.py:129-152).  Returns (loss, y_pred, y_true) like the reference's
    positive branch; with no positive roi the loss is 0 (the reference returns a
    2-tuple there, which its caller cannot unpack -- SURVEY.md appendix A)."""
    if target_class_ids.dim() == 1:
        target_class_ids = target_class_ids.unsqueeze(0)
        target_masks = target_masks.unsqueeze(0)
        pred_masks = pred_masks.unsqueeze(0)
    pos = target_class_ids > 0
    y_pred = torch.sigmoid(pred_masks[:, :, 1:])
    bce = _bce(y_pred, target_masks)
    per_image, _ = _masked_mean(bce, pos[:, :, None, None, None].expand_as(bce), dims=(1, 2, 3, 4))
    if reduce:
        return per_image.mean(), y_pred, target_masks
    return per_image, y_pred, target_masks


def total_loss(rpn_match, rpn_bbox, rpn_class_logits, rpn_pred_bbox, target_class_ids,
               mrcnn_class_logits, target_deltas, mrcnn_bbox, target_mask, mrcnn_mask, roi_valid):
    """Sum of the six losses (model.py:436), per image, averaged over the images
    that have a positive roi.  Returns (loss, dict of the six batch means)."""
    has_pos = ((target_class_ids > 0) & roi_valid).any(dim=1)
    w = has_pos.float()
    n = w.sum().clamp(min=1.0)
    parts = {
        "layer": compute_layer_loss(target_mask, target_class_ids, mrcnn_mask, reduce=False)[0],
        "rpn_bbox": compute_rpn_bbox_loss(rpn_bbox, rpn_match, rpn_pred_bbox, reduce=False),
        "mrcnn_bbox": compute_mrcnn_bbox_loss(target_deltas, target_class_ids, mrcnn_bbox,
                                              reduce=False),
        "mrcnn_class": compute_mrcnn_class_loss(target_class_ids, mrcnn_class_logits, roi_valid,
                                                reduce=False),
        "amodal": compute_amodal_loss(target_mask, target_class_ids, mrcnn_mask, reduce=False)[0],
        "rpn_class": compute_rpn_class_loss(rpn_match, rpn_class_logits, reduce=False),
    }
    per_image = sum(parts.values())
    loss = (per_image * w).sum() / n
    return loss, {k: (v * w).sum() / n for k, v in parts.items()}
