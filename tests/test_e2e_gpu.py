"""End-to-end parity of the assembled path on the GPU against the reference's OWN
MaskRCNN.train_model / predict / detect (tools/gen_golden_e2e.py: ResNet-101 + DeepLab-v2 GLM,
two synthetic 128x128 scenes, recorded RNG draws, name-keyed weights).  SURVEY.md 8(c), last row:
six losses <= 1e-4, rois / class ids exact, masks and scores within tolerance; plus one reference
optimizer step, the module-level gradients, and the loss / RPN-target / box-op fixtures on the device."""
import numpy as np
import pytest
import torch

from tests._parity import (check_box_ops, check_fpn_rpn_grads, check_losses, check_optimizer_step,
                            check_rpn_targets, e2e_model)
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


def _check_proposals(out, g, b=0):
    """Stage A -- this path's OWN proposals against the reference's: RPN outputs <= 1e-4, the same
    number of proposals, the same boxes (1e-5).  Visiting order: two anchors whose foreground scores
    agree to ~1e-7 may come out in either order (the backbones differ by ~2e-6 and the reference's
    sort is unstable anyway), so rows may be swapped with a neighbour; the multiset must be equal
    and at most 1 % of the rows may sit at a different index."""
    A = g["rpn_class_logits"].shape[1]
    assert out["rpn_class_logits"].shape[1] == A
    assert rel(out["rpn_class_logits"][b], g["rpn_class_logits"][0]) < 1e-4
    assert rel(out["rpn_bbox"][b], g["rpn_bbox"][0]) < 1e-4
    k = int(out["num_rois"][b])
    assert k == g["rpn_rois"].shape[1], (k, g["rpn_rois"].shape)
    mine, ref = out["rpn_rois"][b, :k].cpu().numpy(), g["rpn_rois"][0]
    off = np.nonzero(np.abs(mine - ref).max(axis=1) > 1e-5)[0]
    assert off.size <= 0.01 * k, off[:20]
    key = lambda a: a[np.lexsort(np.round(a * 1e4).T[::-1])]
    assert np.abs(key(mine) - key(ref)).max() <= 1e-5
    assert rel(out["gloable_lab"][b:b + 1], g["gloable_lab"]) < 1e-4


def _with_reference_proposals(gs, pr):
    """Stage B input: the reference's proposals, bit for bit.  A roi clipped to the image border
    (x2 == 1.0) puts its last RoIAlign sample exactly on the map's edge, where one ulp of x1 decides
    between a real sample and the extrapolation value (crop_and_resize.c:60, 75) -- the reference has
    the same cliff; everything downstream is therefore compared on identical proposals."""
    rr = torch.zeros((len(gs), 1000, 4), device="cuda")
    num = torch.zeros((len(gs),), dtype=torch.int32, device="cuda")
    for b, g in enumerate(gs):
        k = g["rpn_rois"].shape[1]
        rr[b, :k] = torch.from_numpy(g["rpn_rois"][0]).cuda()
        num[b] = k
    return dict(pr, rpn_rois=rr, num_rois=num)


def _check_forward(out, g, b=0):
    """Stage B -- one image of a training-mode predict() on the reference's proposals."""
    v = out["roi_valid"][b]
    n = int(v.sum())
    assert n == g["rois"].shape[0] and bool(v[:n].all())
    # sampled rois are gathered proposals: the reference's rows, in its order
    assert np.array_equal(out["rois"][b, :n].cpu().numpy(), g["rois"])
    assert np.array_equal(out["target_class_ids"][b, :n].cpu().numpy(), g["target_class_ids"].reshape(-1))
    assert np.allclose(out["target_deltas"][b, :n].cpu().numpy(), g["target_deltas"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(out["target_mask"][b, :n].cpu().numpy().astype(np.uint8), g["target_mask"])
    for key in ("mrcnn_class_logits", "mrcnn_bbox", "mrcnn_mask"):
        got = out[key][b, :n].detach().double().cpu().numpy().reshape(n, -1)
        want = g[key].astype(np.float64).reshape(n, -1)
        e = np.abs(got - want).max(axis=1) / np.abs(want).max()
        assert e.max() < 1e-4, (key, "rows off", np.nonzero(e >= 1e-4)[0].tolist(), e.max())


# (name of the fixture, image size): two 128^2 scenes, scene 0 at 256^2 -- maps of 64^2 .. 4^2, i.e. whole 256-row
# tiles of the convolution kernels with their predicate-free epilogue (tools/gen_golden_e2e.py --dim 256) -- and at
# 512^2 (round 5, --dim 512: C2 at 128 columns, where the small-K 3x3 layers run per kernel ROW -- conv_fwd_kernel's /
# conv_wgrad_kernel's ROW3 instances -- and C3 / C4 at 64 / 32 columns on the tap-row instances of the 256^2 kernel)
TRAIN_FIXTURES = [("e2e_train_0", 128), ("e2e_train_1", 128), ("e2e_train_256_0", 256), ("e2e_train_512_0", 512)]


@pytest.mark.parametrize("name,dim", TRAIN_FIXTURES)
def test_training_predict_and_losses_match_reference(name, dim):
    g = golden(name)
    assert int(g["dim"]) == dim
    m, cfg = e2e_model("cuda", dim)
    inp, pr = _inputs([g])
    with torch.no_grad():
        own = m.predict(inp, mode="training", priorities=pr)
        _check_proposals(own, g)
        out = m.predict(inp, mode="training", priorities=_with_reference_proposals([g], pr))
        loss, parts = m.compute_losses(out, dev(g["rpn_match"]), dev(g["rpn_bbox_target"]))
    _check_forward(out, g)
    for name, want in zip([str(n) for n in g["loss_names"]], g["losses"]):
        got = float(parts[LOSS_KEYS[name]])
        assert abs(got - want) <= 1e-4, (name, got, want)
    assert abs(float(loss) - g["losses"].sum()) <= 1e-4


def test_batched_step_equals_the_mean_of_the_reference_per_image_losses():
    """B = 2 (the reference is batch 1): per-image outputs equal each scene's fixture, the step loss
    is the mean of the two reference totals."""
    gs = [golden("e2e_train_0"), golden("e2e_train_1")]
    m, cfg = e2e_model("cuda")
    inp, pr = _inputs(gs)
    with torch.no_grad():
        own = m.predict(inp, mode="training", priorities=pr)
        out = m.predict(inp, mode="training", priorities=_with_reference_proposals(gs, pr))
        loss, parts = m.compute_losses(out, torch.cat([dev(g["rpn_match"]) for g in gs]),
                                       torch.cat([dev(g["rpn_bbox_target"]) for g in gs]))
    for b, g in enumerate(gs):
        _check_proposals(own, g, b)
        _check_forward(out, g, b)
    want = np.mean([g["losses"].sum() for g in gs])
    assert abs(float(loss) - want) <= 1e-4, (float(loss), want)


@pytest.mark.parametrize("name,dim", TRAIN_FIXTURES)
def test_one_train_step_matches_the_reference_optimizer_step(name, dim):
    """predict -> losses -> backward -> clip 5.0 -> SGD (lr .01, momentum .9, wd 1e-4) on one image:
    the global gradient norm and the updates of 22 watched parameter tensors against the step the
    reference's train_epoch took (model.py:415-444)."""
    g = golden(name)
    m, cfg = e2e_model("cuda", dim)
    params = dict(m.named_parameters())
    names = [str(n) for n in g["names"]]
    for n in names:
        assert np.array_equal(params[n].detach().reshape(-1)[:256].cpu().numpy(), g["before/" + n]), n
    inp, pr = _inputs([g])
    batch = {"images": inp[0], "gt_class_ids": inp[2], "gt_boxes": inp[3], "gt_layer": inp[4],
             "rpn_match": dev(g["rpn_match"]), "rpn_bbox": dev(g["rpn_bbox_target"])}
    opt = m.make_optimizer(float(g["lr"]))
    loss, _ = m.train_step(batch, opt, priorities=_with_reference_proposals([g], pr))
    assert abs(float(loss) - g["losses"].sum()) <= 1e-4
    want_norm = float(g["total_norm"])
    got_norm = float(m.last_grad_norm)
    assert abs(got_norm - want_norm) <= 2e-3 * want_norm, (got_norm, want_norm)
    # Updates of the watched tensors.  Gradients below a ReLU depend on which units are active, and a
    # unit whose pre-activation is within the two backends' ~1e-6 forward difference of zero switches:
    # a fraction f of switched units shows up as a relative L2 error of ~sqrt(f) in everything
    # upstream (measured: 1e-6 for the heads / FPN / RPN layers, up to 2e-3 deep in the backbone; an
    # fp64 model sees the same from torch's own fp32).  Hence L2, not max, and a
    # tolerance per depth.
    worst = {}
    for n in names:
        d_ref = (g["after/" + n].astype(np.float64) - g["before/" + n])
        d_got = params[n].detach().reshape(-1)[:256].double().cpu().numpy() - g["before/" + n]
        assert np.abs(d_ref).max() > 0, n
        err = np.linalg.norm(d_got - d_ref) / np.linalg.norm(d_ref)
        worst[n] = err
        deep = n.startswith(("fpn.C1", "fpn.C2", "fpn.C3", "fpn.C4"))
        # (512^2: the two clip norms differ by 2.3e-4, which scales EVERY update -- the heads' whole error there -- and
        # C5 sits under four times the ReLU units of the 256^2 scene: measured 1.5e-3 on fpn.C5.2.conv3.weight, the
        # same in six runs)
        assert err <= (2e-2 if deep else 4e-3 if dim >= 512 else 2e-3), (n, err)
    print("relative update errors", {k: "%.1e" % v for k, v in worst.items()}, "grad norm", got_norm, want_norm)


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
    32x32 mask logits 1e-4, and the unmolded full-size masks (model.py:464-514, 576-628, 747-806).
    Stage A runs on this path's own proposals (at least 95 of the 100 detections must be the
    reference's: a border-clipped roi or a near-tie at the top-100 cut may differ, see
    _with_reference_proposals); stage B on the reference's proposals is exact."""
    g = golden("e2e_detect_%d" % scene)
    m, cfg = e2e_model("cuda")
    cfg.DETECTION_MIN_CONFIDENCE = 0
    molded = dev(g["molded"])
    want = g["detections"]
    with torch.no_grad():
        own, _ = m.predict([molded, g["image_metas"]], mode="inference")
    own = own[0].cpu().numpy()
    assert own.shape == want.shape
    same = sum(1 for r in own if (np.abs(want[:, :5] - r[:5]).max(axis=1) == 0).any())
    assert same >= 95, same
    # the reference's proposals = its classifier input rois: re-run its proposal stage on the fixture's
    # RPN outputs is not stored here; the train fixtures of the same scene hold them for mode='training'
    # only, so stage B takes them from the inference fixture's own record
    k = g["rpn_rois"].shape[1]
    rr = torch.zeros((1, 1000, 4), device="cuda")
    rr[0, :k] = torch.from_numpy(g["rpn_rois"][0]).cuda()
    pr = {"rpn_rois": rr, "num_rois": torch.tensor([k], dtype=torch.int32, device="cuda")}
    with torch.no_grad():
        detections, mrcnn_mask = m.predict([molded, g["image_metas"]], mode="inference", priorities=pr)
    det = detections[0].cpu().numpy()
    assert det.shape == want.shape, (det.shape, want.shape)
    assert np.array_equal(det[:, :5], want[:, :5])
    assert np.allclose(det[:, 5], want[:, 5], rtol=0, atol=1e-5)
    assert rel(mrcnn_mask[0], g["mrcnn_mask"]) < 1e-4
    res = m.detect([g["image_u8"]], priorities=[pr])
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
    # the device tail alone, fed the reference's own detections and mask logits, is bit-exact, and its
    # RLE hand-off decodes back to the reference's masks
    from sln_amodal_amd import mask_rle
    from sln_amodal_amd.amodal_train import build_coco_results
    dim = g["image_u8"].shape[0]
    out = m.unmold_detections_device(torch.from_numpy(want).cuda(), torch.from_numpy(g["mrcnn_mask"]).cuda(),
                                     g["image_u8"].shape, (0, 0, dim, dim), keep_device=True)
    assert out["masks"] is None and np.array_equal(out["rois"], g["final_rois"])
    assert np.array_equal(out["masks_device"].permute(2, 1, 0).cpu().numpy(), want_masks)
    coco = build_coco_results(None, [7], out["rois"], out["class_ids"], out["scores"], out["masks_device"])
    assert len(coco) == shape[2] and coco[0]["image_id"] == 7 and coco[0]["category_id"] == 1
    for i, c in enumerate(coco):
        seg = c["segmentation"]
        assert seg["size"] == [shape[0], shape[1]]
        assert np.array_equal(mask_rle.decode_counts(mask_rle.from_string(seg["counts"]), shape[0], shape[1]),
                              want_masks[:, :, i])
        y1, x1, y2, x2 = g["final_rois"][i]
        assert c["bbox"] == [x1, y1, x2 - x1, y2 - y1]


# ------------------------------------------------------------------ module-level gradients
from tests._parity import freeze_bn as _freeze_bn, grad_close as _grad_close  # noqa: E402


def test_fpn_rpn_gradients_match_reference_modules():
    """Gradients of the reference's FPN + RPN modules (ResNet-50, 64x64) on the product path: 1e-4
    for everything above C5, and the ReLU-switch tolerance (see
    test_one_train_step_matches_the_reference_optimizer_step) for what lies below 40+ layers.  The
    same module graph on aten convolutions is pinned at 1e-5 by the CPU suite
    (tests/test_model_cpu.py::test_fpn_rpn_gradients_match_reference_modules)."""
    check_fpn_rpn_grads("cuda", 1e-4, 5e-3)


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
        got = (mp.grad if mp.grad is not None else torch.zeros_like(mp)).cpu().numpy()
        assert np.linalg.norm(got - want) <= 5e-3 * max(np.linalg.norm(want), 1e-6), i
    cp, mpar = dict(cls.named_parameters()), dict(msk.named_parameters())
    # everything with a ReLU between it and the loss carries the ReLU-switch tolerance (12 rois x 256
    # channels x 16^2 units per mask-head layer: one switched unit is ~1e-3 in the layers below it)
    for n in [str(s) for s in g["cls_names"]]:
        deep = n.startswith(("conv1", "bn1"))
        _grad_close(cp[n].grad, g["cls_g/" + n], float(g["cls_gn/" + n]), "classifier." + n, 5e-3 if deep else 1e-4)
    for n in [str(s) for s in g["mask_names"]]:
        deep = n.startswith(("conv1", "conv2", "conv3", "conv4", "deconv"))
        _grad_close(mpar[n].grad, g["mask_g/" + n], float(g["mask_gn/" + n]), "mask." + n, 5e-3 if deep else 1e-4)


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


# ------------------------------------------------------------------ deep gradients, discrete choices forced
def _reference_switches(m, gm):
    """Providers for conv_hip.FORCE_RELU / nn_ops.FORCE_POOL_ARG from the e2e_relu_masks_0 fixture: the
    reference's ReLU sign bitmap of every detector layer (call order: RPN levels P2..P6) and the winning
    taps of C1's max-pool."""
    import re
    name_of = {id(p): n for n, p in m.named_parameters()}
    calls = {}

    def bitmap(key):
        shape = tuple(int(v) for v in gm["relu_shape/" + key])
        bits = np.unpackbits(gm["relu/" + key])[:int(np.prod(shape))].reshape(shape)
        return torch.from_numpy(bits.astype(bool)).cuda()

    def relu_provider(own, y):
        n = name_of.get(id(own))
        if n is None or n.startswith("GLM_modual"):
            return None
        k = calls.get(n, 0)
        calls[n] = k + 1
        mt = re.fullmatch(r"fpn\.(C\d\.\d+)\.conv(\d)\.weight", n)
        if n == "fpn.C1.0.weight":
            key = "fpn.C1.2#0"
        elif mt:
            key = "fpn.%s.relu#%d" % (mt.group(1), int(mt.group(2)) - 1)
        elif n == "rpn.conv_shared.weight":
            key = "rpn.relu#%d" % k
        elif re.fullmatch(r"(classifier|mask)\.conv\d\.weight", n):
            key = "%s.relu#%d" % (n.split(".")[0], int(n.split(".")[1][4:]) - 1)
        elif n == "mask.deconv.weight":
            key = "mask.relu#4"
        else:
            raise AssertionError("a ReLU layer the fixture does not know: " + n)
        b = bitmap(key)
        if n == "mask.deconv.weight":      # ours: [n, (a, b, c), i, j] before the depth-to-space shuffle
            nn_, C, H2, W2 = b.shape
            b = b.view(nn_, C, H2 // 2, 2, W2 // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(nn_, 4 * C, H2 // 2, W2 // 2)
        assert tuple(b.shape[1:]) == tuple(y.shape[1:]) and b.shape[0] <= y.shape[0], (n, b.shape, y.shape)
        return b

    def pool_provider(arg):
        tap = torch.from_numpy(gm["pool_tap"]).cuda().permute(0, 2, 3, 1).contiguous()     # [N,OH,OW,C]
        return tap if tuple(tap.shape) == tuple(arg.shape) else None
    return relu_provider, pool_provider, calls


@pytest.mark.parametrize("steps", [1, 2])
def test_deep_gradients_with_reference_relu_masks(steps):
    """SURVEY.md 8(c) "grads <= 1e-4" for the 101-layer graph.  A unit whose pre-activation lies within
    the two backends' ~1e-6 forward difference of zero switches state and moves every gradient below it by
    sqrt(fraction switched) -- that, not arithmetic, is what the depth-aware bounds of
    test_one_train_step_matches_the_reference_optimizer_step absorb.  Here the reference's own discrete
    choices are forced (the sign bitmap of all 112 ReLU outputs and the winning tap of every max-pool
    window, recorded by tools/gen_golden_e2e.py --masks from the reference's train step), so what is
    compared is the arithmetic of the backward pass alone: every watched weight gradient C1..C5, FPN, RPN,
    heads, and the gradient of the input image, at 1e-4 relative L2.  steps = 1 runs the bootstrap path
    (exact amax passes, un-chained gradient preparation), steps = 2 the steady-state one (delayed scales,
    gradient preparation chained into the data-gradient epilogues, shortcut links)."""
    from sln_amodal_amd import conv_hip, nn_ops
    g, gm = golden("e2e_train_0"), golden("e2e_relu_masks_0")
    m, cfg = e2e_model("cuda")
    params = dict(m.named_parameters())
    inp, pr = _inputs([g])
    pr = _with_reference_proposals([g], pr)
    stats0 = list(conv_hip.CHAIN_STATS)
    try:
        for it in range(steps):
            relu_p, pool_p, calls = _reference_switches(m, gm)
            conv_hip.FORCE_RELU, nn_ops.FORCE_POOL_ARG = relu_p, pool_p
            for p in params.values():
                p.grad = None
            images = inp[0].clone().requires_grad_(True)
            out = m.predict([images] + inp[1:], mode="training", priorities=pr)
            loss, parts = m.compute_losses(out, dev(g["rpn_match"]), dev(g["rpn_bbox_target"]))
            loss.backward()
    finally:
        conv_hip.FORCE_RELU = nn_ops.FORCE_POOL_ARG = None
    assert sum(calls.values()) == len(gm["relu_order"]), (sum(calls.values()), len(gm["relu_order"]))
    if steps == 2:
        assert conv_hip.CHAIN_STATS[1] > stats0[1]          # the chained path really ran
    assert abs(float(loss) - g["losses"].sum()) <= 1e-4
    worst = {}
    for n, want_norm in zip([str(s) for s in gm["names"]], gm["grad_norms"]):
        worst[n] = _grad_close(params[n].grad, gm["grad/" + n].astype(np.float64), float(want_norm), n, 1e-4)
    gi = images.grad.double().cpu().numpy()
    want = gm["grad_images"].astype(np.float64)
    worst["images"] = float(np.linalg.norm(gi - want) / np.linalg.norm(want))
    print("relative L2 gradient errors with forced switches:", {k: "%.1e" % v for k, v in worst.items()})
    assert worst["images"] <= 1e-4, worst["images"]


# ------------------------------------------------------------------ f4: on-disk formats on the device
def test_reference_layout_checkpoint_loads_into_the_gpu_model_and_reproduces_the_reference_step(tmp_path):
    """SURVEY 8(f4) / model.py:287-302, 366: a checkpoint in the reference's layout -- a plain state_dict
    with exactly the reference's 1432 keys and shapes (tests/golden/state_dict_keys.json), values
    name-keyed like the e2e fixtures' -- written with torch.save, found by find_last, loaded by
    load_weights into a DIFFERENTLY initialised model that already lives on the GPU: every key is
    consumed, none is missing, and the loaded model reproduces the reference's train step of
    e2e_train_0 (six losses at 1e-4, RPN outputs 1e-4)."""
    import json
    import os
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.model import MaskRCNN
    from tests._util import GOLDEN, e2e_init_
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))

    class Holder(object):                    # just enough of a module for the name-keyed initialiser
        def __init__(self):
            self.sd = {k: (torch.zeros(shape, dtype=torch.int64) if k.endswith("num_batches_tracked")
                           else torch.zeros(shape)) for k, shape in keys.items()}

        def state_dict(self):
            return self.sd

    h = Holder()
    e2e_init_(h)

    class C(Config):
        NAME = "e2e"
        IMAGE_MAX_DIM = 128
        IMAGE_MIN_DIM = 128
        STRICT_IMAGE_DIVISIBILITY = True

    ckpt_dir = tmp_path / "e2e"
    os.makedirs(ckpt_dir)
    path = str(ckpt_dir / "mask_rcnn_e2e_0005.pth")
    torch.save(h.sd, path)
    torch.manual_seed(123)
    m = MaskRCNN(C(), str(tmp_path)).apply_amodal_heads().cuda()
    assert m.find_last()[1] == path
    before = m.fpn.C4[5].conv3.weight.detach().clone()
    res = m.load_state_dict(torch.load(path, map_location="cpu"), strict=True)     # every key, both ways
    assert not res.missing_keys and not res.unexpected_keys
    torch.manual_seed(124)
    m = MaskRCNN(C(), str(tmp_path)).apply_amodal_heads().cuda()
    m.load_weights(path)                                                           # the reference's entry point
    assert not torch.equal(before, m.fpn.C4[5].conv3.weight.detach())
    assert all(p.is_cuda for p in m.parameters())
    m.set_trainable(".*", exclusive_off=False)
    for p in m.GLM_modual.parameters():
        p.requires_grad = False
    g = golden("e2e_train_0")
    params = dict(m.named_parameters())
    for n in [str(s) for s in g["names"]]:
        assert np.array_equal(params[n].detach().reshape(-1)[:256].cpu().numpy(), g["before/" + n]), n
    inp, pr = _inputs([g])
    with torch.no_grad():
        out = m.predict(inp, mode="training", priorities=_with_reference_proposals([g], pr))
        loss, parts = m.compute_losses(out, dev(g["rpn_match"]), dev(g["rpn_bbox_target"]))
    assert rel(out["rpn_class_logits"][0], g["rpn_class_logits"][0]) < 1e-4
    _check_forward(out, g)
    for name, want in zip([str(n) for n in g["loss_names"]], g["losses"]):
        assert abs(float(parts[LOSS_KEYS[name]]) - want) <= 1e-4, name


@pytest.mark.parametrize("scene", [0, 1])
def test_real_data_loader_batch_matches_reference_load_image_gt(scene, tmp_path):
    """AmodalDataset's device-resident batch (what train_model consumes) against the reference's
    load_image_gt / model.Dataset item (Functions.py:675-736, model.py:80-116; tools/gen_golden_loader.py)
    for a non-square uint8 image + `.npz` label, the recorded flip / jitter / anchor draws replayed:
    the molded image equal (uint8 Pillow squash, then the mean), the label planes -- decoded on the device
    from the resized uint64 label -- equal, boxes / class ids / RPN targets equal."""
    from sln_amodal_amd import amodal_train, ops
    from sln_amodal_amd.model import MaskRCNN
    from tests._parity import loader_config, loader_draws, unpack, write_loader_scene
    g = golden("loader_%d" % scene)
    cfg = loader_config(int(g["dim"]))
    cfg.ARCHITECTURE = "resnet50"
    write_loader_scene(tmp_path, g)
    m = MaskRCNN(cfg, str(tmp_path)).cuda()
    ds = amodal_train.AmodalDataset(cfg, m, root=str(tmp_path), device="cuda")
    batch = ds._load_real([0], draws=[loader_draws(g, m.anchors.shape[0])])
    assert batch["flipped"] == [int(g["flip"])]
    assert np.array_equal(batch["images"][0].cpu().numpy(), g["images"])
    n = g["gt_boxes"].shape[0]
    assert np.array_equal(batch["gt_class_ids"][0, :n].cpu().numpy(), g["gt_class_ids"])
    assert int(batch["gt_class_ids"][0, n:].sum()) == 0
    assert np.array_equal(batch["gt_boxes"][0, :n].cpu().numpy(), g["gt_boxes"])
    planes = ops.label_decode(batch["gt_layer"], cfg.NUM_CLASSES - 1, n)[0]          # [L,N,H,W]
    assert np.array_equal(planes.cpu().numpy(), unpack(g, "gt_layer"))
    assert np.array_equal(batch["rpn_match"][0].cpu().numpy(), g["rpn_match"])
    assert np.allclose(batch["rpn_bbox"][0].cpu().numpy(), g["rpn_bbox"], rtol=1e-6, atol=1e-6)
    # the iterator draws its own flips / jitter and yields the same structure
    cfg.BATCH_SIZE = 2
    it = iter(ds)
    b2 = next(it)
    assert b2["images"].shape == (2, 3, 128, 128) and b2["gt_layer"].dtype == torch.int64


# ------------------------------------------------------------------ batched inference (configs[1], evaluate)
def test_batched_inference_reproduces_each_reference_detect_fixture():
    """predict(mode='inference') over B = 2 images (the reference is batch 1, model.py:576-628): each image's
    rows equal its own e2e_detect fixture exactly as the batch-1 path's do -- boxes / class ids exact, scores
    1e-5, mask logits 1e-4 -- rows behind the per-image count are zero, and detect() on the two images gives
    the reference's final boxes and full-size masks per image."""
    gs = [golden("e2e_detect_0"), golden("e2e_detect_1")]
    m, cfg = e2e_model("cuda")
    cfg.DETECTION_MIN_CONFIDENCE = 0
    molded = torch.cat([dev(g["molded"]) for g in gs])
    rr = torch.zeros((2, 1000, 4), device="cuda")
    num = torch.zeros((2,), dtype=torch.int32, device="cuda")
    for b, g in enumerate(gs):
        k = g["rpn_rois"].shape[1]
        rr[b, :k] = torch.from_numpy(g["rpn_rois"][0]).cuda()
        num[b] = k
    pr = {"rpn_rois": rr, "num_rois": num}
    metas = np.concatenate([g["image_metas"] for g in gs])
    with torch.no_grad():
        detections, mrcnn_mask = m.predict([molded, metas], mode="inference", priorities=pr)
    counts = m.last_num_detections.cpu().numpy()
    assert detections.shape == (2, 100, 6) and mrcnn_mask.shape[:2] == (2, 100)
    for b, g in enumerate(gs):
        want = g["detections"]
        n = want.shape[0]
        assert counts[b] == n
        det = detections[b].cpu().numpy()
        assert np.array_equal(det[:n, :5], want[:, :5])
        assert np.allclose(det[:n, 5], want[:, 5], rtol=0, atol=1e-5)
        assert not det[n:].any() and not mrcnn_mask[b, n:].any()
        assert rel(mrcnn_mask[b, :n], g["mrcnn_mask"]) < 1e-4
    res = m.detect([g["image_u8"] for g in gs], priorities=pr)
    assert len(res) == 2
    for r, g in zip(res, gs):
        assert np.array_equal(r["rois"], g["final_rois"]) and np.array_equal(r["class_ids"], g["final_class_ids"])
        assert np.allclose(r["scores"], g["final_scores"], rtol=0, atol=1e-5)
        shape = tuple(int(v) for v in g["final_masks_shape"])
        want_masks = np.unpackbits(g["final_masks"])[:int(np.prod(shape))].reshape(shape)
        assert r["masks"].shape == shape and (r["masks"] != want_masks).mean() < 1e-4


def test_config2_batched_inference_resnet50_8x800_without_host_sync():
    """BASELINE configs[1]: ResNet-50 + FPN SLN forward-only, 8 x 800x800 (800 is not a multiple of 64: the
    reference's build() rejects it, model.py:153-157; the shapes are consistent).  The whole
    predict(mode='inference') -- GLM, backbone, RPN, proposals + NMS, classifier over 8 x 1000 proposal slots,
    top-100 detections, mask head -- runs under torch's sync debug mode "error": no device -> host copy, no
    `int(num_rois[0])`.  Then the hand-off: unmold_detections_device + RLE per image.  An image run alone
    gives the rows it has inside the batch (boxes / ids equal; scores 1e-5: the per-tensor fp16 scales of
    a batch of one differ from a batch of eight)."""
    from sln_amodal_amd import mask_rle
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.model import MaskRCNN
    from tests._util import e2e_init_

    class C(Config):
        NAME = "c2"
        IMAGE_MAX_DIM = 800
        IMAGE_MIN_DIM = 800
        ARCHITECTURE = "resnet50"
        DETECTION_MIN_CONFIDENCE = 0

    cfg = C()
    m = MaskRCNN(cfg, "/tmp/sln_logs").apply_amodal_heads()
    e2e_init_(m)
    m = m.cuda()
    gen = torch.Generator().manual_seed(5)
    imgs = [torch.randint(0, 256, (800, 800, 3), generator=gen, dtype=torch.uint8).numpy() for _ in range(8)]
    molded, metas, windows = m.mold_inputs(imgs)
    x = torch.from_numpy(molded.transpose(0, 3, 1, 2)).float().cuda()
    with torch.no_grad():
        m.predict([x, metas], mode="inference")            # (first call: scale bootstraps, constant caches)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            detections, mrcnn_mask = m.predict([x, metas], mode="inference")
        finally:
            torch.cuda.set_sync_debug_mode("default")
    counts = m.last_num_detections.cpu().numpy()
    assert detections.shape == (8, 100, 6) and mrcnn_mask.shape == (8, 100, 2, 32, 32)
    det = detections.cpu().numpy()
    assert (counts > 0).all() and (counts <= 100).all()
    for b in range(8):
        n = int(counts[b])
        assert not det[b, n:].any()
        assert (det[b, :n, 4] == 1).all() and (det[b, :n, :4] >= 0).all() and (det[b, :n, :4] <= 800).all()
        assert (np.diff(det[b, :n, 5]) <= 0).all()                    # descending scores
    # the hand-off per image: full-size masks on the device + their RLE
    total = 0
    for b in range(8):
        n = int(counts[b])
        out = m.unmold_detections_device(detections[b, :n], mrcnn_mask[b, :n], imgs[b].shape, windows[b],
                                         keep_device=True)
        k = out["rois"].shape[0]
        assert out["masks_device"].shape == (k, 800, 800)
        if k:
            rles = mask_rle.encode(out["masks_device"])
            i = k // 2
            back = mask_rle.decode_counts(mask_rle.from_string(rles[i]["counts"]), 800, 800)
            assert np.array_equal(back, out["masks_device"][i].cpu().numpy().T)
        total += k
    assert total > 0
    with torch.no_grad():
        d1, _ = m.predict([x[3:4], metas[3:4]], mode="inference")
    n = int(counts[3])
    assert int(m.last_num_detections[0]) == n
    one = d1[0].cpu().numpy()
    key = lambda a: a[np.lexsort(a[:, :4].T[::-1])]
    assert np.array_equal(key(one[:n, :5]), key(det[3, :n, :5]))
    assert np.allclose(np.sort(one[:n, 5]), np.sort(det[3, :n, 5]), rtol=0, atol=1e-5)
