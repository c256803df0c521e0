"""GPU parity of the assembled hot path (HIP backends) against goldens produced by
the reference's own Python / nn.Modules (tools/gen_golden.py)."""
import numpy as np
import pytest
import torch

from tests._util import golden, key_init_

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def close(got, want, tol=1e-4):
    want = np.asarray(want)
    return np.allclose(got.detach().cpu().numpy(), want, rtol=tol, atol=tol * max(np.abs(want).max(), 1e-6))


@pytest.fixture(autouse=True)
def _hip_backend():
    from sln_amodal_amd import nn_ops
    old = nn_ops.BACKEND
    nn_ops.BACKEND = "hip"       # raise instead of silently falling back
    yield
    nn_ops.BACKEND = old


def test_fpn_rpn_hip_match_reference_modules():
    from sln_amodal_amd.modal.modals import FPN, RPN, ResNet
    g = golden("module_fpn_rpn")
    resnet = ResNet("resnet50", stage5=True)
    fpn = FPN(*resnet.stages(), out_channels=256).eval().cuda()
    rpn = RPN(3, 1, 256).eval().cuda()
    key_init_(fpn); key_init_(rpn)
    with torch.no_grad():
        p = fpn(dev(g["x"]))
        r = rpn(p[0])
    for got, name in ((p[0], "p2"), (p[1], "p3"), (p[3], "p5"), (p[4], "p6"), (r[0], "rpn_logits"),
                      (r[1], "rpn_probs"), (r[2], "rpn_bbox")):
        assert tuple(got.shape) == g[name].shape, name
        assert close(got, g[name]), name


def test_glm_hip_matches_reference_module():
    from sln_amodal_amd.modal.deeplabv2 import DeepLabV2_ResNet101_MSC
    g = golden("module_glm")
    glm = DeepLabV2_ResNet101_MSC(182).eval().cuda()
    key_init_(glm)
    with torch.no_grad():
        lg = glm(dev(g["x"]).contiguous(memory_format=torch.channels_last))
    assert close(lg, g["logits"])


def test_heads_hip_match_reference_modules():
    from sln_amodal_amd.modal.modals import Classifier, Mask
    g = golden("module_heads")
    cls = Classifier(256, 7, (128, 128, 3), 2).eval().cuda()
    msk = Mask(256, 16, (128, 128, 3), 2).eval()
    msk.conv1 = torch.nn.Conv2d(439, 256, kernel_size=3, stride=1)
    msk = msk.cuda()
    key_init_(cls); key_init_(msk)
    maps = [dev(g["map%d" % i]).contiguous(memory_format=torch.channels_last) for i in range(4)]
    rois = dev(g["rois"])
    with torch.no_grad():
        c = cls(maps, rois)
        m, _ = msk(maps, rois, dev(g["glm_feat"]).contiguous(memory_format=torch.channels_last))
    assert close(c[0], g["cls_logits"]) and close(c[1], g["cls_probs"]) and close(c[2], g["cls_bbox"])
    assert close(m, g["mask_logits"])


@pytest.mark.parametrize("fmt", ["nhwc", "nchw"])
def test_pyramid_roi_align_matches_reference_graph(fmt):
    from sln_amodal_amd.modal.modals import pyramid_roi_align, roi_levels
    g = golden("pyramid_roi_align")
    boxes = dev(g["boxes"])
    assert np.array_equal(roi_levels(boxes, (1024, 1024, 3)).cpu().numpy(), g["levels"])
    C = g["map0"].shape[1]
    maps = []
    for i in range(4):
        m = dev(g["map%d" % i])
        if fmt == "nhwc":       # C=8 maps: force a genuinely channels-last buffer
            m = m.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        maps.append(m.requires_grad_(True))
    pooled = pyramid_roi_align([boxes.unsqueeze(0)] + maps, 7, (1024, 1024, 3))
    assert np.array_equal(pooled.detach().cpu().numpy(), g["pooled"])
    pooled.backward(dev(g["upstream"]))
    for i in range(4):
        want = g["grad%d" % i]
        got = maps[i].grad.cpu().numpy() if maps[i].grad is not None else np.zeros_like(want)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-5), i
    assert C == 8


@pytest.mark.parametrize("dim", [128, 256])
def test_proposal_layer_matches_reference_graph(dim):
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.modal.Functions import proposal_layer

    class C(Config):
        IMAGE_MAX_DIM = dim

    g = golden("proposal_layer_%d" % dim)
    rois, num = proposal_layer([dev(g["probs"]), dev(g["deltas"])], 1000, 0.7, dev(g["anchors"]), C(),
                               return_counts=True)
    k = int(num[0])
    assert k == g["rois"].shape[1]
    assert np.allclose(rois[0, :k].cpu().numpy(), g["rois"][0], rtol=0, atol=1e-6)


@pytest.mark.parametrize("case", ["a", "b"])
@pytest.mark.parametrize("path", ["labels", "planes"])
def test_detection_target_layer_replays_reference_draws(case, path):
    """The recorded torch.randperm draws are replayed as priorities; the valid slots,
    in order, must be the reference's rows (rois / class ids exact, masks exact)."""
    from sln_amodal_amd import ops
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.modal.Functions import bbox_overlaps, detection_target_layer
    g = golden("detection_target_%s" % case)
    L = int(g["L"])

    class C(Config):
        NUM_CLASSES = L + 1
        IMAGE_MAX_DIM = 128

    cfg = C()
    props, gtb = dev(g["proposals"]), dev(g["gt_boxes"])
    P, N = props.shape[0], gtb.shape[0]
    iou_max = bbox_overlaps(props, gtb).max(dim=1)[0]
    pos_idx = torch.nonzero(iou_max >= 0.5)[:, 0]
    neg_idx = torch.nonzero(iou_max < 0.5)[:, 0]
    pr_pos = torch.full((1, P), -1e9, device="cuda")
    pr_neg = torch.full((1, P), -1e9, device="cuda")
    pp, pn = dev(g["perm_pos"]).long(), dev(g["perm_neg"]).long()
    pr_pos[0, pos_idx[pp]] = -torch.arange(len(pp), device="cuda", dtype=torch.float32)
    pr_neg[0, neg_idx[pn]] = -torch.arange(len(pn), device="cuda", dtype=torch.float32)
    lab = dev(g["label"].view(np.int64)).unsqueeze(0)
    kw = {"labels": lab} if path == "labels" else {}
    planes = None if path == "labels" else ops.label_decode(lab, L, N)
    out = detection_target_layer(props.unsqueeze(0), torch.ones(1, N, dtype=torch.int32, device="cuda"),
                                 gtb.unsqueeze(0), planes, cfg, priority_pos=pr_pos, priority_neg=pr_neg,
                                 **kw)
    v = out["roi_valid"][0]
    n = int(v.sum())
    assert n == g["rois"].shape[0] and bool(v[:n].all())
    assert np.array_equal(out["rois"][0, :n].cpu().numpy(), g["rois"])
    assert np.array_equal(out["class_ids"][0, :n].cpu().numpy(), g["class_ids"])
    assert np.allclose(out["deltas"][0, :n].cpu().numpy(), g["deltas"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(out["masks"][0, :n].cpu().numpy(), g["masks"])


def _small_model(arch="resnet50", dim=256):
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.model import MaskRCNN

    class C(Config):
        NAME = "t"
        IMAGE_MAX_DIM = dim
        ARCHITECTURE = arch

    torch.manual_seed(0)
    cfg = C()
    m = MaskRCNN(cfg, "/tmp/sln_logs").apply_amodal_heads().cuda()
    m.set_trainable(".*", exclusive_off=False)
    for p in m.GLM_modual.parameters():
        p.requires_grad = False
    return m, cfg


def test_train_step_runs_updates_and_stays_finite():
    from sln_amodal_amd import synthetic
    m, cfg = _small_model()
    batch = synthetic.make_batch(cfg, 2, 256, 256, seed=1234, anchors_f64=m.anchors_f64)
    synthetic.calibrate_batchnorm(m, batch["images"])
    synthetic.calibrate_glm(m, batch["images"])
    synthetic.warm_start_rpn(m, [batch], iters=30)
    opt = m.make_optimizer(cfg.LEARNING_RATE)
    w0 = m.mask.conv2.weight.detach().clone()
    c0 = m.fpn.C3[0].conv1.weight.detach().clone()
    losses = []
    for _ in range(3):
        loss, parts = m.train_step(batch, opt)
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[0] > 0
    assert not torch.equal(w0, m.mask.conv2.weight) and not torch.equal(c0, m.fpn.C3[0].conv1.weight)
    assert set(parts) == {"layer", "rpn_bbox", "mrcnn_bbox", "mrcnn_class", "amodal", "rpn_class"}


def test_a_train_step_with_a_clamped_operand_block_is_skipped_not_applied():
    """a step whose operands clamped to +-65504 computed with under-estimated values; it must not reach
    the weights.  Provoked like a loss spike does it: one layer's weights are multiplied by 2^12 between two steps, so
    its output outgrows the 2^5 of head room its delayed scale has.  The step is vetoed on the device (no host sync in
    train_step), counted, weights and momentum stay bit-identical; the scale follows, and the next step applies.
    With the guard off (SLN_SKIP_CLAMPED_STEPS=0 semantics) the same step IS applied -- the round-5 behaviour."""
    from sln_amodal_amd import conv_hip, synthetic
    m, cfg = _small_model()
    batch = synthetic.make_batch(cfg, 2, 256, 256, seed=1234, anchors_f64=m.anchors_f64)
    synthetic.calibrate_batchnorm(m, batch["images"])
    synthetic.calibrate_glm(m, batch["images"])
    synthetic.warm_start_rpn(m, [batch], iters=10)
    opt = m.make_optimizer(1e-4)
    gen = torch.Generator(device="cuda").manual_seed(5)
    pr = {"pos": torch.rand(2, 1000, device="cuda", generator=gen), "neg": torch.rand(2, 1000, device="cuda", generator=gen)}
    for _ in range(3):
        m.train_step(batch, opt, priorities=pr)
    assert opt.skipped_steps() == 0
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    P = dict(m.named_parameters())
    sat0 = conv_hip.saturation_count()
    with torch.no_grad():
        m.fpn.P3_conv2[1].weight.mul_(4096.0)       # the spike: this layer's output (and what follows) is x 4096
    snap = {n: P[n].detach().clone() for n in names}
    bufs = {n: opt.state[P[n]].clone() for n in names if P[n] in opt.state}
    loss, _ = m.train_step(batch, opt, priorities=pr)
    assert conv_hip.saturation_count() > sat0                           # blocks clamped in that step ...
    assert opt.skipped_clamped_steps() == 1 and opt.skipped_steps() == 1    # ... so it was vetoed, as a clamp
    assert all(torch.equal(P[n], snap[n]) for n in names)               # nothing reached the weights
    assert all(torch.equal(opt.state[P[n]], b) for n, b in bufs.items())
    # the scales follow (a tensor computed FROM clamped operands under-reports its own maximum once more, so the
    # layers downstream may need a second vetoed step): within three steps a step runs clamp-free and is applied,
    # and no step that clamped was
    applied_at = None
    for k in range(3):
        sat1, skip1 = conv_hip.saturation_count(), opt.skipped_steps()
        before = {n: P[n].detach().clone() for n in names}
        m.train_step(batch, opt, priorities=pr)
        clamped = conv_hip.saturation_count() > sat1
        changed = sum(not torch.equal(P[n], before[n]) for n in names)
        assert (opt.skipped_steps() == skip1 + 1) == clamped
        assert (changed == 0) == clamped
        if not clamped:
            applied_at = k
            assert changed > len(names) // 2
            break
    assert applied_at is not None
    assert opt.skipped_steps() == opt.skipped_clamped_steps()
    sat1, skips = conv_hip.saturation_count(), opt.skipped_clamped_steps()
    # guard off: the clamped step is applied (what round 5 did)
    with torch.no_grad():
        m.fpn.P3_conv2[1].weight.mul_(4096.0)
    snap = {n: P[n].detach().clone() for n in names}
    old = conv_hip.SKIP_CLAMPED_STEPS
    conv_hip.SKIP_CLAMPED_STEPS = False
    try:
        m.train_step(batch, opt, priorities=pr)
    finally:
        conv_hip.SKIP_CLAMPED_STEPS = old
    assert conv_hip.saturation_count() > sat1 and opt.skipped_clamped_steps() == skips
    assert sum(not torch.equal(P[n], snap[n]) for n in names) > len(names) // 2


def test_loss_parity_hip_conv_vs_aten_conv_same_proposals():
    """Six losses with the HIP split-bf16 conv stack vs aten fp32 convs, same weights,
    batch, proposals and sampling priorities: within 1e-4 (north-star tolerance)."""
    from sln_amodal_amd import nn_ops, synthetic
    m, cfg = _small_model()
    batch = synthetic.make_batch(cfg, 2, 256, 256, seed=7, anchors_f64=m.anchors_f64)
    synthetic.calibrate_batchnorm(m, batch["images"])
    synthetic.calibrate_glm(m, batch["images"])
    synthetic.warm_start_rpn(m, [batch], iters=30)
    gen = torch.Generator(device="cuda").manual_seed(5)
    pr = {"pos": torch.rand(2, 1000, device="cuda", generator=gen),
          "neg": torch.rand(2, 1000, device="cuda", generator=gen)}
    inp = [batch["images"], None, batch["gt_class_ids"], batch["gt_boxes"], batch["gt_layer"]]
    res = {}
    for be in ("torch", "hip"):
        nn_ops.BACKEND = be
        with torch.no_grad():
            out = m.predict(inp, mode="training", priorities=pr)
            if be == "torch":
                pr = dict(pr, rpn_rois=out["rpn_rois"], num_rois=out["num_rois"])
            _, parts = m.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
        res[be] = {k: float(v) for k, v in parts.items()}
    assert int(out["roi_valid"].sum()) > 20
    for k in res["hip"]:
        assert abs(res["hip"][k] - res["torch"][k]) < 1e-4, (k, res)


def test_detect_inference_path_runs():
    from sln_amodal_amd import synthetic
    m, cfg = _small_model()
    batch = synthetic.make_batch(cfg, 1, 256, 256, seed=3, anchors_f64=m.anchors_f64)
    synthetic.calibrate_batchnorm(m, batch["images"])
    synthetic.calibrate_glm(m, batch["images"])
    img = (np.random.RandomState(0).rand(200, 180, 3) * 255).astype(np.uint8)
    res = m.detect([img])
    assert isinstance(res, list)
    for r in res:
        n = r["rois"].shape[0]
        assert r["rois"].shape == (n, 4) and r["scores"].shape == (n,) and r["class_ids"].shape == (n,)
        if n:   # like the reference, an image without surviving boxes yields an empty mask stack
            assert r["masks"].shape == (200, 180, n) and r["masks"].dtype == np.uint8
            assert (r["class_ids"] == 1).all()


def test_config2_resnet50_fpn_forward_800x800_bs8():
    """BASELINE config #2: ResNet-50 + FPN forward-only, 8 x 800x800 (not a multiple of
    64: the reference's build() rejects it, model.py:153-157; shapes are consistent)."""
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.model import MaskRCNN
    from sln_amodal_amd.modal.Functions import proposal_layer

    class C(Config):
        NAME = "c2"
        IMAGE_MAX_DIM = 800
        ARCHITECTURE = "resnet50"

    torch.manual_seed(0)
    cfg = C()
    m = MaskRCNN(cfg, "/tmp/sln_logs").cuda()
    assert cfg.BACKBONE_SHAPES.tolist() == [[200, 200], [100, 100], [50, 50], [25, 25], [13, 13]]
    x = torch.randn(8, 3, 800, 800, device="cuda")
    with torch.no_grad():
        maps, logits, probs, bbox = m.rpn_forward(x)
        rois, num = proposal_layer([probs, bbox], 1000, 0.7, m.anchors, cfg, return_counts=True)
    assert [tuple(t.shape[2:]) for t in maps] == [(200, 200), (100, 100), (50, 50), (25, 25), (13, 13)]
    A = m.anchors.shape[0]
    assert A == 3 * (200 * 200 + 100 * 100 + 50 * 50 + 25 * 25 + 13 * 13)
    assert probs.shape == (8, A, 2) and bbox.shape == (8, A, 4)
    assert rois.shape == (8, 1000, 4) and bool((num > 0).all())
    assert float(rois.min()) >= 0.0 and float(rois.max()) <= 1.0


def test_cli_evaluate_synthetic_runs(tmp_path):
    """BASELINE config #1 plumbing: `amodal_train.py evaluate` on 2 synthetic 512x512
    images with ResNet-50 (here on the GPU product path)."""
    from sln_amodal_amd import amodal_train
    amodal_train.main(["evaluate", "--synthetic", "--arch", "resnet50", "--image-dim", "512",
                       "--limit", "2", "--logs", str(tmp_path)])


def test_stage_transition_gradients_merge_into_the_lateral_map():
    """c2..c4 have three readers: the FPN lateral (stride 1) and the next stage's strided conv1 + downsample.
    The pair's lattice sum is added in place into the lateral's data gradient (conv_hip PAIR_STATS); the
    backbone gradients equal the unmerged path's up to the order of the additions."""
    from sln_amodal_amd import conv_hip
    from sln_amodal_amd.modal.modals import FPN, ResNet
    from tests._util import key_init_
    torch.manual_seed(0)
    resnet = ResNet("resnet50", stage5=True)
    fpn = FPN(*resnet.stages(), out_channels=256).eval()
    key_init_(fpn)
    fpn = fpn.cuda()
    for m in fpn.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            for p in m.parameters():
                p.requires_grad = False
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(2, 3, 128, 128, device="cuda", generator=g)
    ups = None
    grads = []
    for merged in (True, False):
        conv_hip.PAIR_STRIDED = merged
        try:
            for p in fpn.parameters():
                p.grad = None
            outs = fpn(x)
            if ups is None:
                ups = [torch.randn(o.shape, device="cuda", generator=g) for o in outs]
            before = list(conv_hip.PAIR_STATS)
            sum((o * u).sum() for o, u in zip(outs, ups)).backward()
            if merged:
                assert conv_hip.PAIR_STATS[0] == before[0] + 3 and conv_hip.PAIR_STATS[1] == before[1] + 3
            else:
                assert conv_hip.PAIR_STATS == before
        finally:
            conv_hip.PAIR_STRIDED = True
        grads.append({n: p.grad.clone() for n, p in fpn.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 100
    for n in grads[0]:
        a, b = grads[0][n], grads[1][n]
        assert (a - b).norm().item() <= 2e-5 * max(b.norm().item(), 1e-12), n


def test_crop_gradients_are_added_inside_the_rpn_data_gradient(monkeypatch):
    """P2..P5 are read by the RPN's shared conv and by the heads' crops: the crops' gradient maps are left in a
    conv_hip.GradInbox and added in the epilogue of the RPN conv's data gradient (model.FUSE_CROP_GRADS) instead of
    autograd's accumulation pass.  Same proposals and sampling priorities with the switch on and off: every
    parameter gradient agrees up to the order of the additions (the crops' own atomics already vary in the last
    bit); all four maps are handed over and consumed; a backward pass without the RPN losses fails loudly."""
    from sln_amodal_amd import conv_hip, model as model_mod, synthetic
    m, cfg = _small_model()
    batch = synthetic.make_batch(cfg, 2, 256, 256, seed=11, anchors_f64=m.anchors_f64)
    synthetic.calibrate_batchnorm(m, batch["images"])
    synthetic.calibrate_glm(m, batch["images"])
    synthetic.warm_start_rpn(m, [batch], iters=30)
    gen = torch.Generator(device="cuda").manual_seed(5)
    pr = {"pos": torch.rand(2, 1000, device="cuda", generator=gen),
          "neg": torch.rand(2, 1000, device="cuda", generator=gen)}
    inp = [batch["images"], None, batch["gt_class_ids"], batch["gt_boxes"], batch["gt_layer"]]
    with torch.no_grad():
        out = m.predict(inp, mode="training", priorities=pr)
    pr = dict(pr, rpn_rois=out["rpn_rois"], num_rois=out["num_rois"])
    grads = []
    for fused in (False, True, True):
        monkeypatch.setattr(model_mod, "FUSE_CROP_GRADS", fused)
        for p in m.parameters():
            p.grad = None
        before = list(conv_hip.GradInbox.STATS)
        out = m.predict(inp, mode="training", priorities=pr)
        loss, _ = m.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
        loss.backward()
        got = [a - b for a, b in zip(conv_hip.GradInbox.STATS, before)]
        assert got == ([4, 4] if fused else [0, 0]), got
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert int(out["roi_valid"].sum()) > 20
    assert grads[0].keys() == grads[1].keys() and len(grads[0]) > 100
    for n in grads[0]:
        a, b = grads[1][n], grads[0][n]
        assert (a - b).norm().item() <= 2e-5 * max(b.norm().item(), 1e-12), n
    # P2..P4: with the deposit in its inbox the RPN conv's data gradient is the map's whole gradient and prepares it
    # for the FPN output conv (modals.CHAIN_FPN_OUTPUTS, a soft chain): three more hand-overs, the same gradients
    from sln_amodal_amd.modal import modals
    used = []
    for chained in (False, True):
        monkeypatch.setattr(modals, "CHAIN_FPN_OUTPUTS", chained)
        for p in m.parameters():
            p.grad = None
        before = conv_hip.CHAIN_STATS[1]
        out = m.predict(inp, mode="training", priorities=pr)
        loss, _ = m.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
        loss.backward()
        used.append(conv_hip.CHAIN_STATS[1] - before)
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    assert used[1] == used[0] + 3, used
    for n in grads[-1]:
        a, b = grads[-1][n], grads[-2][n]
        assert (a - b).norm().item() <= 2e-5 * max(b.norm().item(), 1e-12), n
    # without the RPN losses the conv that should add the deposit never runs: an error, not a silent loss
    out = m.predict(inp, mode="training", priorities=pr)
    _, parts = m.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
    with pytest.raises(RuntimeError, match="never consumed"):
        (parts["layer"] + parts["mrcnn_class"]).backward()
    # (and nothing is left behind for the next pass)
    assert not conv_hip.GradInbox.pending
    out = m.predict(inp, mode="training", priorities=pr)
    loss, _ = m.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
    loss.backward()


def test_p2_lateral_gradient_prepared_by_the_finest_fpn_conv(monkeypatch):
    """modals.CHAIN_FPN_LATERAL: the finest 3x3 FPN conv's data gradient also prepares the P2 lateral's gradient
    (identity through the merge).  Chains work from the second pass on (the first one bootstraps the gradient's
    scale slot), so two passes per mode; every FPN / backbone gradient of the second pass agrees with the unchained
    path's, and the chained mode really hands one more gradient over."""
    from sln_amodal_amd import conv_hip
    from sln_amodal_amd.modal import modals
    from sln_amodal_amd.modal.modals import FPN, ResNet
    torch.manual_seed(0)
    resnet = ResNet("resnet50", stage5=True)
    fpn = FPN(*resnet.stages(), out_channels=256).eval()
    key_init_(fpn)
    fpn = fpn.cuda()
    for m in fpn.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            for p in m.parameters():
                p.requires_grad = False
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(2, 3, 128, 128, device="cuda", generator=g)
    ups, grads, used = None, [], []
    for chained in (False, True):
        monkeypatch.setattr(modals, "CHAIN_FPN_LATERAL", chained)
        for _ in range(2):
            conv_hip.update_scales()
            for p in fpn.parameters():
                p.grad = None
            outs = fpn(x)
            if ups is None:
                ups = [torch.randn(o.shape, device="cuda", generator=g) for o in outs]
            before = conv_hip.CHAIN_STATS[1]
            sum((o * u).sum() for o, u in zip(outs, ups)).backward()
        used.append(conv_hip.CHAIN_STATS[1] - before)
        grads.append({n: p.grad.clone() for n, p in fpn.named_parameters() if p.grad is not None})
    assert used[1] == used[0] + 1, used
    assert grads[0].keys() == grads[1].keys() and "P2_conv1.weight" in grads[0]
    for n in grads[0]:
        a, b = grads[1][n], grads[0][n]
        assert (a - b).norm().item() <= 2e-5 * max(b.norm().item(), 1e-12), n


def test_mask_head_tail_gradients_with_the_chained_deconv(monkeypatch):
    """nn_ops.CHAIN_DECONV: the logits conv's data gradient prepares the deconv's gradient (re-viewed [4M, Cout]
    parts, the bias sum credited to the first of the four bias copies).  Second pass of each mode (chains need a
    scale history): input, deconv and logits gradients agree with the unchained path's."""
    from sln_amodal_amd import conv_hip, nn_ops
    torch.manual_seed(3)
    deconv = torch.nn.ConvTranspose2d(256, 256, kernel_size=2, stride=2).cuda()
    conv5 = torch.nn.Conv2d(256, 2, kernel_size=1).cuda()
    g = torch.Generator(device="cuda").manual_seed(2)
    feat = torch.randn(24, 256, 16, 16, device="cuda", generator=g).relu_().contiguous(memory_format=torch.channels_last)
    up = torch.randn(24, 2, 32, 32, device="cuda", generator=g)
    params = list(deconv.parameters()) + list(conv5.parameters())
    res, used = [], []
    for chained in (False, True):
        monkeypatch.setattr(nn_ops, "CHAIN_DECONV", chained)
        for _ in range(2):
            conv_hip.update_scales()
            for p in params:
                p.grad = None
            x = feat.clone().requires_grad_(True)
            before = conv_hip.CHAIN_STATS[1]
            z = nn_ops.deconv2x2_relu_conv1x1(x, deconv, conv5)
            (z * up).sum().backward()
        used.append(conv_hip.CHAIN_STATS[1] - before)
        res.append([x.grad.clone()] + [p.grad.clone() for p in params])
    assert used == [0, 1], used
    for a, b in zip(res[1], res[0]):
        assert a.shape == b.shape
        assert (a - b).norm().item() <= 2e-5 * max(b.norm().item(), 1e-12)


def test_bench_config_detect_prints_the_contract_line():
    """`bench.py --config detect` (BASELINE.json configs[1]: ResNet-50 + FPN SLN forward-only; default 8 x 800^2) at a
    small shape, with the evaluation hand-off (`--tail`: unmold + RLE on the device): one JSON line with the contract's
    keys, a roofline for the dominant kernel, `traffic: null` (no counter file belongs to this problem)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "detect", "--batch", "2", "--dim", "256",
                        "--steps", "2", "--warmup", "1", "--tail"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "step_roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["images_per_gpu"] == 2
    assert out["config"]["arch"] == "resnet50" and out["config"]["tail"] is True
    assert out["config"]["hip_graph"] is True, out["config"].get("hip_graph_error")      # (round 6: replayed from one HIP graph)
    assert out["config"]["rle_masks_encoded"] > 0
    assert out["roofline"]["traffic"] is None and out["step_roofline"]["algorithmic_tflop_per_step"] > 0
    assert len(out["config"]["detections_last_batch"]) == 2


def test_mask_head_on_the_positive_slots_gives_the_same_train_step():
    """MaskRCNN.mask_train_slots (opt-in; default None = the reference's graph, the mask branch on all 100 sampled rois,
    model.py:664-700): with positive_slots() = int(R * ROI_POSITIVE_RATIO) the mask head runs on the first 70 slots
    only -- detection_target_layer returns the positives first (Functions.py:223-416) and both mask losses read
    positives only (loss.py:113-152).  From the same weights the first train step gives the SAME six losses (1e-6),
    the same clip norm (1e-4) and the same update of the mask head (1e-4; the RoIAlign backward's fp32 atomics land in
    another order); the following steps -- on delayed scales of the new roi count -- stay finite and near (a
    gross-divergence bound: they amplify the first step's difference like any two runs in this clipped regime)."""
    from sln_amodal_amd import nn_ops, synthetic
    old = nn_ops.BACKEND
    nn_ops.BACKEND = "hip"
    try:
        m, cfg = _small_model()
        batch = synthetic.make_batch(cfg, 2, 256, 256, seed=3, anchors_f64=m.anchors_f64)
        synthetic.calibrate_batchnorm(m, batch["images"])
        synthetic.calibrate_glm(m, batch["images"])
        synthetic.warm_start_rpn(m, [batch], iters=40)
        gen = torch.Generator(device="cuda").manual_seed(5)
        pr = {"pos": torch.rand(2, 1000, device="cuda", generator=gen),
              "neg": torch.rand(2, 1000, device="cuda", generator=gen)}
        assert m.mask_train_slots is None and m.positive_slots() == 70
        start = {k: v.detach().clone() for k, v in m.state_dict().items()}
        res = {}
        for tag, mode in (("boot", None), (None, None), ("again", None), (70, m.positive_slots())):
            m.load_state_dict(start)
            m.mask_train_slots = mode
            opt = m.make_optimizer(0.001)
            rows, first = [], None
            for it in range(3):
                loss, parts = m.train_step(batch, opt, priorities=pr)
                rows.append((float(loss), {k: float(v) for k, v in parts.items()}, float(m.last_grad_norm)))
                if it == 0:
                    first = {k: v.detach().clone() for k, v in m.state_dict().items()
                             if v.dtype == torch.float32 and k.startswith(("mask.", "classifier."))}
            res[tag] = (rows, first)
        m.mask_train_slots = None
        # ("boot": the same weights once before -- it bootstraps every scale slot, so that the two runs compared below
        # both start from slots with a history; its distance from the second all-slots run is printed as the noise floor)
        (a, wa), (b, wb) = res[None], res[70]
        noise = {}
        for other in ("boot", "again"):
            w0 = res[other][1]
            d = {k: float((wa[k] - w0[k]).norm() / (wa[k] - start[k]).norm().clamp_min(1e-30)) for k in wa}
            print("all slots vs all slots (%s):" % other, sorted(((v, k) for k, v in d.items()), reverse=True)[:4])
            if other == "again":
                noise = d
        (la, pa, na), (lb, pb, nb) = a[0], b[0]
        assert abs(la - lb) <= 2e-6 * max(1.0, abs(la)), (a[0], b[0])
        for k in pa:
            assert abs(pa[k] - pb[k]) <= 2e-6 * max(1.0, abs(pa[k])), (k, pa[k], pb[k])
        assert abs(na - nb) <= 1e-4 * na, (na, nb)
        moved = [k for k in wa if float((wa[k] - start[k]).norm()) > 0]
        assert any(k.startswith("mask.") for k in moved)
        rel = sorted(((float((wa[k] - wb[k]).norm() / (wa[k] - start[k]).norm()), k) for k in moved), reverse=True)
        print("first-step update, all slots vs positive slots, worst tensors:", rel[:6])
        # per tensor: what two runs of the SAME mode differ by (x 3; measured 1.6e-4 on mask.conv1.weight, <= 3.4e-5 on
        # every other tensor, some runs bit-identical -- delayed scales move with the slots' history), with floors of
        # 1e-3 for weights (inside the whole suite `mask.deconv.weight` came out 2.4e-4 apart with a same-mode spread of
        # 1e-8: the mode changes the roi count and with it the scale slots' maxima, and the mask head answers a changed
        # scale with ReLU switches -- 1-3e-3 between the bootstrap pass and any later pass of ONE mode, printed above)
        # and 1e-2 for the per-channel tensors (sums of 400 k terms by fp32 atomics)
        for r, k in rel:
            assert r <= max(3.0 * noise.get(k, 0.0), 1e-3 if wa[k].dim() >= 2 else 1e-2), (k, r, noise.get(k))
        # the following steps run on delayed scales of another roi count and amplify the first step's 1e-4 like any two
        # runs in this clipped regime do (tests/test_multistep_gpu.py): a gross-divergence check only (mostly < 1e-3; one
        # suite run in about twelve showed 1.7e-2 at the third step, profiles/r5_r_gpu_suite_run1_red_positive_slots.log)
        for (la, pa, na), (lb, pb, nb) in zip(a[1:], b[1:]):
            assert np.isfinite(lb) and abs(la - lb) <= 0.1 * max(1.0, abs(la)), (a, b)
    finally:
        nn_ops.BACKEND = old
