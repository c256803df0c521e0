"""Synthetic "COCOA-shape" training batches, generated on the device
(SURVEY.md section 8(d)): uint8-uniform images minus the mean pixel; per image
N axis-aligned ellipses in painter's order (object 0 on top) encoded as the
dataset's uint64 occlusion label; tight boxes; class id 1; RPN targets from
build_rpn_targets with seeded draws.  Seed = base + rank."""
import numpy as np
import torch

from .modal.Functions import build_rpn_targets, extract_bboxes_from_labels


def make_labels(B, H, W, n_obj, gen, device):
    """[B,H,W] int64 bit patterns (low word visible bit, high word occluded bits)."""
    s = H / 1024.0
    cy = (128 + 768 * torch.rand((B, n_obj), generator=gen, device=device)) * s
    cx = (128 + 768 * torch.rand((B, n_obj), generator=gen, device=device)) * (W / 1024.0)
    ry = (48 + 208 * torch.rand((B, n_obj), generator=gen, device=device)) * s
    rx = (48 + 208 * torch.rand((B, n_obj), generator=gen, device=device)) * (W / 1024.0)
    yy = torch.arange(H, device=device, dtype=torch.float32)[None, None, :, None]
    xx = torch.arange(W, device=device, dtype=torch.float32)[None, None, None, :]
    label = torch.zeros((B, H, W), dtype=torch.int64, device=device)
    covered = torch.zeros((B, H, W), dtype=torch.bool, device=device)
    for i in range(n_obj):
        m = (((yy[:, 0] - cy[:, i, None, None]) / ry[:, i, None, None]) ** 2 +
             ((xx[:, 0] - cx[:, i, None, None]) / rx[:, i, None, None]) ** 2) <= 1.0
        label |= (m & ~covered).long() << i
        label |= (m & covered).long() << (32 + i)
        covered |= m
    return label


def make_batch(config, B, H, W, n_obj=8, seed=1234, device="cuda", anchors_f64=None):
    gen = torch.Generator(device=device).manual_seed(seed)
    mean = torch.tensor(np.asarray(config.MEAN_PIXEL), dtype=torch.float32, device=device)
    img = torch.randint(0, 256, (B, H, W, 3), generator=gen, device=device, dtype=torch.uint8)
    images = (img.float() - mean).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
    labels = make_labels(B, H, W, n_obj, gen, device)
    gt_boxes = extract_bboxes_from_labels(labels, n_obj).float()          # [B,N,4] pixels
    present = (gt_boxes[..., 2] > gt_boxes[..., 0]) & (gt_boxes[..., 3] > gt_boxes[..., 1])
    gt_class_ids = present.to(torch.int32)
    pr = torch.rand((B, anchors_f64.shape[0]), generator=gen, device=device)
    rpn_match, rpn_bbox = build_rpn_targets((H, W, 3), anchors_f64, gt_class_ids, gt_boxes, config,
                                            priority=pr)
    return {"images": images, "image_metas": None, "gt_class_ids": gt_class_ids,
            "gt_boxes": gt_boxes, "gt_layer": labels, "rpn_match": rpn_match.unsqueeze(2),
            "rpn_bbox": rpn_bbox}


def _calibrate(forward):
    """Run `forward()` while every conv + frozen-BN pair first overwrites the BN's running statistics
    with the statistics of the raw convolution output (so layer k is calibrated on activations
    normalised by layers < k).  Convolutions stay on the active backend (nn_ops.CALIBRATING: conv_bn_act
    refreshes the statistics itself and flags its own F.batch_norm call); BN layers reached through
    F.batch_norm directly -- not through conv_bn_act -- are calibrated by the patch below."""
    import torch.nn.functional as F
    from . import nn_ops
    orig = F.batch_norm
    count = [0]

    def patched(x, rm, rv, w=None, b=None, training=False, momentum=0.1, eps=1e-5):
        if not training and not nn_ops.IN_CALIBRATED_BN:
            with torch.no_grad():
                rm.copy_(x.mean(dim=(0, 2, 3)))
                rv.copy_(x.var(dim=(0, 2, 3), unbiased=False).clamp(min=1e-6))
            count[0] += 1
        return orig(x, rm, rv, w, b, training, momentum, eps)

    nn_ops.CALIBRATING = count
    F.batch_norm = patched
    try:
        with torch.no_grad():
            forward()
    finally:
        F.batch_norm = orig
        nn_ops.CALIBRATING = None
    return count[0]


def calibrate_batchnorm(model, images):
    """Emulate pretrained BatchNorm statistics for a randomly initialised detector
    (the reference always starts from a COCO checkpoint whose BN statistics
    normalise the activations; with identity statistics and Xavier weights a
    100-layer backbone's activations explode and every proposal degenerates).
    Benchmark / test setup only; the timed step keeps BN frozen."""
    return _calibrate(lambda: model.fpn(images))


def calibrate_glm(model, images):
    """Same for the frozen DeepLab-v2 GLM (no checkpoint is available offline)."""
    import torch.nn.functional as F
    s = model.config.GLM_SIZE

    def fwd():
        x = F.interpolate(images, size=(s, s), mode="bilinear", align_corners=False)
        model.GLM_modual.eval()
        model.GLM_modual.base(x.contiguous(memory_format=torch.channels_last))
    return _calibrate(fwd)


def warm_start_rpn(model, batches, iters=60, lr=0.02):
    """Emulate a trained RPN on the (fixed) synthetic batches: FPN features are
    computed once without gradients, then only the RPN head is fitted to the
    batches' rpn_match / rpn_bbox targets for a few iterations.  Gives the timed
    training step realistic proposals (tens of positive rois per image, NMS with
    realistic survivor counts, non-zero gradients in every head).  Setup only."""
    from .modal import loss as L
    params = list(model.rpn.parameters())
    opt = torch.optim.SGD(params, lr=lr, momentum=0.9)
    feats = []
    with torch.no_grad():
        for b in batches:
            feats.append([p.detach() for p in model.fpn(b["images"])])
    for it in range(iters):
        for b, maps in zip(batches, feats):
            outs = [model.rpn(p) for p in maps]
            logits, _, bbox = [torch.cat(list(o), dim=1) for o in zip(*outs)]
            loss = L.compute_rpn_class_loss(b["rpn_match"], logits) + \
                L.compute_rpn_bbox_loss(b["rpn_bbox"], b["rpn_match"], bbox)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(params, 5.0)
            opt.step()
    return float(loss)
