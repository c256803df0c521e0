"""End-to-end parity of the assembled path on the GPU against the reference's OWN
MaskRCNN.train_model / predict / detect (tools/gen_golden_e2e.py: ResNet-101 + DeepLab-v2 GLM,
two synthetic 128x128 scenes, recorded RNG draws, name-keyed weights).  SURVEY.md 8(c), last row:
six losses <= 1e-4, rois / class ids exact, masks and scores within tolerance; plus one reference
optimizer step, the module-level gradients, and the loss / RPN-target / box-op fixtures on the device."""
import numpy as np
import pytest
import torch

from tests._parity import check_box_ops, check_losses, check_optimizer_step, check_rpn_targets, e2e_model
from tests._util import golden, key_init_

pytestmark = pytest.mark.gpu

LOSS_KEYS = {"compute_layer_loss": "layer", "compute_mrcnn_class_loss": "mrcnn_class",
             "compute_rpn_class_loss": "rpn_class", "compute_rpn_bbox_loss": "rpn_bbox",
             "compute_mrcnn_bbox_loss": "mrcnn_bbox", "compute_amodal_loss": "amodal"}


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def rel(got, want):
    """max |got - want| / max(|want|)."""
    want = np.asarray(want, np.float64)
    got = got.detach().double().cpu().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-12))


@pytest.fixture(autouse=True)
def _hip_backend():
    from sln_amodal_amd import nn_ops
    old = nn_ops.BACKEND
    nn_ops.BACKEND = "hip"
    yield
    nn_ops.BACKEND = old


def _inputs(gs):
    """Batch of the fixtures' scenes: [images, metas, gt_class_ids, gt_boxes (pixels), labels]."""
    n_max = max(g["gt_boxes"].shape[1] for g in gs)
    ids = torch.zeros((len(gs), n_max), dtype=torch.int32)
    boxes = torch.zeros((len(gs), n_max, 4), dtype=torch.float32)
    for b, g in enumerate(gs):
        n = g["gt_boxes"].shape[1]
        ids[b, :n] = torch.from_numpy(g["gt_class_ids"][0].astype(np.int32))
        boxes[b, :n] = torch.from_numpy(g["gt_boxes"][0])
    images = torch.cat([torch.from_numpy(g["images"]) for g in gs]).cuda()
    labels = torch.stack([torch.from_numpy(g["label"].view(np.int64)) for g in gs]).cuda()
    replay = ([g["perm_pos"] for g in gs], [g["perm_neg"] for g in gs])
    return [images, None, ids.cuda(), boxes.cuda(), labels], {"replay": replay}


def _check_forward(out, g, b=0):
    """One image of a training-mode predict() against its reference fixture."""
    A = g["rpn_class_logits"].shape[1]
    assert out["rpn_class_logits"].shape[1] == A
    assert rel(out["rpn_class_logits"][b], g["rpn_class_logits"][0]) < 1e-4
    assert rel(out["rpn_bbox"][b], g["rpn_bbox"][0]) < 1e-4
    k = int(out["num_rois"][b])
    assert k == g["rpn_rois"].shape[1], (k, g["rpn_rois"].shape)
    assert np.allclose(out["rpn_rois"][b, :k].cpu().numpy(), g["rpn_rois"][0], rtol=0, atol=1e-5)
    v = out["roi_valid"][b]
    n = int(v.sum())
    assert n == g["rois"].shape[0] and bool(v[:n].all())
    # sampled rois are gathered proposals: the reference's rows, in its order
    assert np.allclose(out["rois"][b, :n].cpu().numpy(), g["rois"], rtol=0, atol=1e-5)
    assert np.array_equal(out["target_class_ids"][b, :n].cpu().numpy(), g["target_class_ids"].reshape(-1))
    assert np.allclose(out["target_deltas"][b, :n].cpu().numpy(), g["target_deltas"], rtol=1e-4, atol=1e-4)
    assert np.array_equal(out["target_mask"][b, :n].cpu().numpy().astype(np.uint8), g["target_mask"])
    assert rel(out["mrcnn_class_logits"][b, :n], g["mrcnn_class_logits"]) < 1e-4
    assert rel(out["mrcnn_bbox"][b, :n], g["mrcnn_bbox"]) < 1e-4
    assert rel(out["mrcnn_mask"][b, :n], g["mrcnn_mask"]) < 1e-4
    assert rel(out["gloable_lab"][b:b + 1], g["gloable_lab"]) < 1e-4


@pytest.mark.parametrize("scene", [0, 1])
def test_training_predict_and_losses_match_reference(scene):
    g = golden("e2e_train_%d" % scene)
    m, cfg = e2e_model("cuda")
    inp, pr = _inputs([g])
    with torch.no_grad():
        out = m.predict(inp, mode="training", priorities=pr)
        loss, parts = m.compute_losses(out, dev(g["rpn_match"]), dev(g["rpn_bbox_target"]))
    _check_forward(out, g)
    for name, want in zip([str(n) for n in g["loss_names"]], g["losses"]):
        got = float(parts[LOSS_KEYS[name]])
        assert abs(got - want) <= 1e-4, (name, got, want)
    assert abs(float(loss) - g["losses"].sum()) <= 2e-4


def test_batched_step_equals_the_mean_of_the_reference_per_image_losses():
    """B = 2 (the reference is batch 1): per-image outputs equal each scene's fixture, the step loss
    is the mean of the two reference totals."""
    gs = [golden("e2e_train_0"), golden("e2e_train_1")]
    m, cfg = e2e_model("cuda")
    inp, pr = _inputs(gs)
    with torch.no_grad():
        out = m.predict(inp, mode="training", priorities=pr)
        loss, parts = m.compute_losses(out, torch.cat([dev(g["rpn_match"]) for g in gs]),
                                       torch.cat([dev(g["rpn_bbox_target"]) for g in gs]))
    for b, g in enumerate(gs):
        _check_forward(out, g, b)
    want = np.mean([g["losses"].sum() for g in gs])
    assert abs(float(loss) - want) <= 2e-4, (float(loss), want)


@pytest.mark.parametrize("scene", [0, 1])
def test_one_train_step_matches_the_reference_optimizer_step(scene):
    """predict -> losses -> backward -> clip 5.0 -> SGD (lr .01, momentum .9, wd 1e-4) on one image:
    the global gradient norm and the updates of 22 watched parameter tensors against the step the
    reference's train_epoch took (model.py:415-444)."""
    g = golden("e2e_train_%d" % scene)
    m, cfg = e2e_model("cuda")
    params = dict(m.named_parameters())
    names = [str(n) for n in g["names"]]
    for n in names:
        assert np.array_equal(params[n].detach().reshape(-1)[:256].cpu().numpy(), g["before/" + n]), n
    inp, pr = _inputs([g])
    batch = {"images": inp[0], "gt_class_ids": inp[2], "gt_boxes": inp[3], "gt_layer": inp[4],
             "rpn_match": dev(g["rpn_match"]), "rpn_bbox": dev(g["rpn_bbox_target"])}
    opt = m.make_optimizer(float(g["lr"]))
    loss, _ = m.train_step(batch, opt, priorities=pr)
    assert abs(float(loss) - g["losses"].sum()) <= 2e-4
    want_norm = float(g["total_norm"])
    got_norm = float(m.last_grad_norm)
    assert abs(got_norm - want_norm) <= 2e-3 * want_norm, (got_norm, want_norm)
    worst = 0.0
    for n in names:
        d_ref = (g["after/" + n].astype(np.float64) - g["before/" + n])
        d_got = params[n].detach().reshape(-1)[:256].double().cpu().numpy() - g["before/" + n]
        scale = np.abs(d_ref).max()
        assert scale > 0, n
        err = np.abs(d_got - d_ref).max() / scale
        worst = max(worst, err)
        assert err <= 2e-2, (n, err)
    print("worst relative update error", worst, "grad norm", got_norm, want_norm)


def test_optimizer_step_matches_reference_optimizer_on_device():
    check_optimizer_step("cuda")


def test_losses_match_reference_on_device():
    check_losses("cuda")


@pytest.mark.parametrize("dim", [128, 256])
def test_build_rpn_targets_replays_reference_draws_on_device(dim):
    check_rpn_targets(dim, "cuda")


def test_box_ops_match_reference_on_device():
    check_box_ops("cuda")


@pytest.mark.parametrize("scene", [0, 1])
def test_detect_matches_reference(scene):
    """MaskRCNN.detect: predict(mode='inference') detections (boxes / class ids exact, scores 1e-5),
    32x32 mask logits 1e-4, and the unmolded full-size masks (model.py:464-514, 576-628, 747-806)."""
    g = golden("e2e_detect_%d" % scene)
    m, cfg = e2e_model("cuda")
    cfg.DETECTION_MIN_CONFIDENCE = 0
    molded = dev(g["molded"])
    with torch.no_grad():
        detections, mrcnn_mask = m.predict([molded, g["image_metas"]], mode="inference")
    det = detections[0].cpu().numpy()
    want = g["detections"]
    assert det.shape == want.shape, (det.shape, want.shape)
    assert np.array_equal(det[:, :5], want[:, :5])
    assert np.allclose(det[:, 5], want[:, 5], rtol=0, atol=1e-5)
    assert rel(mrcnn_mask[0], g["mrcnn_mask"]) < 1e-4
    res = m.detect([g["image_u8"]])
    assert len(res) == 1
    r = res[0]
    assert np.array_equal(r["rois"], g["final_rois"])
    assert np.array_equal(r["class_ids"], g["final_class_ids"])
    assert np.allclose(r["scores"], g["final_scores"], rtol=0, atol=1e-5)
    shape = tuple(int(v) for v in g["final_masks_shape"])
    want_masks = np.unpackbits(g["final_masks"])[:int(np.prod(shape))].reshape(shape)
    assert r["masks"].shape == shape
    # a 1e-5 difference in a mask logit can move a bytescaled pixel across the 0.5 threshold
    assert (r["masks"] != want_masks).mean() < 1e-4


# ------------------------------------------------------------------ module-level gradients
def _freeze_bn(*mods):
    for mod in mods:
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                for p in m.parameters():
                    p.requires_grad = False


def _grad_close(got, want_slice, want_norm, name, tol=1e-4):
    got = got.reshape(-1)[:want_slice.size].double().cpu().numpy()
    assert np.abs(got - want_slice).max() <= tol * max(np.abs(want_slice).max(), 1e-3 * want_norm, 1e-12), name


def test_fpn_rpn_gradients_match_reference_modules():
    from sln_amodal_amd.modal.modals import FPN, RPN, ResNet
    g = golden("module_grads_fpn_rpn")
    resnet = ResNet("resnet50", stage5=True)
    fpn = FPN(*resnet.stages(), out_channels=256).eval().cuda()
    rpn = RPN(3, 1, 256).eval().cuda()
    key_init_(fpn); key_init_(rpn)
    _freeze_bn(fpn, rpn)
    x = dev(g["x"]).requires_grad_(True)
    p = fpn(x)
    outs = [rpn(t) for t in p]
    logits = torch.cat([o[0] for o in outs], 1)
    bbox = torch.cat([o[2] for o in outs], 1)
    loss = (logits * dev(g["up_logits"])).sum() + (bbox * dev(g["up_bbox"])).sum() + \
        sum((t * dev(g["up_p%d" % i])).sum() for i, t in enumerate(p[:4]))
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * max(abs(float(g["loss"])), 1.0)
    loss.backward()
    assert rel(x.grad, g["gx"]) < 1e-4
    fp, rp = dict(fpn.named_parameters()), dict(rpn.named_parameters())
    for n in [str(s) for s in g["fpn_names"]]:
        _grad_close(fp[n].grad, g["fpn_g/" + n], float(g["fpn_gn/" + n]), n)
    for n in [str(s) for s in g["rpn_names"]]:
        _grad_close(rp[n].grad, g["rpn_g/" + n], float(g["rpn_gn/" + n]), n)


def test_head_gradients_match_reference_modules():
    from sln_amodal_amd.modal.modals import Classifier, Mask
    g = golden("module_grads_heads")
    cls = Classifier(256, 7, (128, 128, 3), 2).eval()
    msk = Mask(256, 16, (128, 128, 3), 2).eval()
    msk.conv1 = torch.nn.Conv2d(439, 256, kernel_size=3, stride=1)
    key_init_(cls); key_init_(msk)
    cls, msk = cls.cuda(), msk.cuda()
    _freeze_bn(cls, msk)
    maps = [dev(g["map%d" % i]).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            for i in range(4)]
    rois = dev(g["rois"])
    c_out = cls(maps, rois)
    m_out, _ = msk(maps, rois, dev(g["glm_feat"]).contiguous(memory_format=torch.channels_last))
    loss = (c_out[0] * dev(g["up_cls"])).sum() + (c_out[2] * dev(g["up_bbox"])).sum() + \
        (m_out * dev(g["up_mask"])).sum()
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * max(abs(float(g["loss"])), 1.0)
    loss.backward()
    for i, mp in enumerate(maps):
        want = g["gmap%d" % i]
        got = mp.grad if mp.grad is not None else torch.zeros_like(mp)
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-6), i
    cp, mpar = dict(cls.named_parameters()), dict(msk.named_parameters())
    for n in [str(s) for s in g["cls_names"]]:
        _grad_close(cp[n].grad, g["cls_g/" + n], float(g["cls_gn/" + n]), "classifier." + n)
    for n in [str(s) for s in g["mask_names"]]:
        _grad_close(mpar[n].grad, g["mask_g/" + n], float(g["mask_gn/" + n]), "mask." + n)


def test_mask_head_fused_concat_equals_torch_cat():
    """pyramid_roi_align_image(cat_extra=256) + the roi-feature crop written in behind it is the
    reference's torch.cat((glm_crop, roi_features), 1) (modals.py:481): outputs and gradients equal."""
    from sln_amodal_amd.modal.modals import Mask, pyramid_roi_align_image
    gen = torch.Generator().manual_seed(3)
    msk = Mask(256, 16, (128, 128, 3), 2).eval()
    msk.conv1 = torch.nn.Conv2d(439, 256, kernel_size=3, stride=1)
    key_init_(msk)
    msk = msk.cuda()
    _freeze_bn(msk)
    B, R = 2, 9
    probs = torch.rand(B, 183, 65, 65, generator=gen).cuda().contiguous(memory_format=torch.channels_last)
    ctr = torch.rand(B, R, 2, generator=gen) * 0.6 + 0.2
    size = torch.exp(torch.rand(B, R, 2, generator=gen) * 3.0 - 3.2)
    rois = torch.cat([ctr - size / 2, ctr + size / 2], 2).clamp(0, 1).cuda()
    box_ind = torch.arange(B, dtype=torch.int32).repeat_interleave(R).cuda()
    box_ind[5] = -1                                        # a padded roi slot
    res = []
    for fused in (False, True):
        maps = [(torch.randn(B, 256, s, s, generator=torch.Generator().manual_seed(10 + s)) * 0.5).cuda()
                .contiguous(memory_format=torch.channels_last).requires_grad_(True) for s in (32, 16, 8, 4)]
        feat = pyramid_roi_align_image([rois, probs], 16, (65, 65), istrain=True, box_ind=box_ind,
                                       cat_extra=256 if fused else 0)
        assert (getattr(feat, "_sln_cat_buf", None) is not None) == fused
        out, _ = msk(maps, rois, feat.detach() if not fused else feat, box_ind)
        out.square().sum().backward()
        res.append((out.detach(), [m.grad for m in maps], msk.conv1.weight.grad.clone()))
        msk.zero_grad(set_to_none=True)
    assert torch.equal(res[0][0], res[1][0])
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(a.abs().max()))
    assert torch.allclose(res[0][2], res[1][2], rtol=1e-4, atol=1e-5 * float(res[0][2].abs().max()))
