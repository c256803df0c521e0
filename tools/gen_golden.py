"""Generate golden vectors by RUNNING the reference's Python in the build container.

    python tools/gen_golden.py            # writes tests/golden/*.npz

Every fixture holds seeded inputs + the reference's outputs (data only).  The
script needs /root/reference and is never run on the GPU box.  Fixtures whose
outputs went through the reference's native extensions carry
`native = "oracle"`: those extensions are unbuildable here, so this repo's C
oracle stood in for them (tools/ref_harness.py) and the fixture pins only the
Python graph logic around them.
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_harness  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


def small_config(ref_config, dim, num_classes=2):
    class Cfg(ref_config.Config):
        NAME = "golden"
        GPU_COUNT = 0
        IMAGE_MAX_DIM = dim
        IMAGE_MIN_DIM = dim
        NUM_CLASSES = num_classes
        EXPERIMENT_DIR = tempfile.mkdtemp()
    return Cfg()


def synth_label(rng, H, W, n_obj, hole=None):
    """Painter's-order ellipses -> uint64 label."""
    yy, xx = np.mgrid[0:H, 0:W]
    masks = []
    for i in range(n_obj):
        cy, cx = rng.uniform(0.2, 0.8) * H, rng.uniform(0.2, 0.8) * W
        ry, rx = rng.uniform(0.1, 0.35) * H, rng.uniform(0.1, 0.35) * W
        masks.append(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0)
    from oracle.oracle import encode_labels
    return encode_labels(np.stack(masks)), np.stack(masks)


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_modals, ref_F = ref_harness.install()
    import utils as ref_utils
    import config as ref_config
    import modal.loss as ref_loss
    import amodal_train as ref_train

    # ---------------------------------------------------------------- anchors
    for dim in (128, 256):
        cfg = small_config(ref_config, dim)
        a = ref_utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS,
                                               cfg.BACKBONE_SHAPES, cfg.BACKBONE_STRIDES,
                                               cfg.RPN_ANCHOR_STRIDE)
        save("anchors_%d" % dim, anchors=a, shapes=cfg.BACKBONE_SHAPES)
    cfg = small_config(ref_config, 1024)
    a = ref_utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS,
                                           cfg.BACKBONE_SHAPES, cfg.BACKBONE_STRIDES, 1)
    save("anchors_1024_digest", count=np.array(a.shape[0]), colsum=a.sum(axis=0),
         first=a[:7], last=a[-7:], stride_sample=a[::9973])

    # ----------------------------------------------------------- label decode
    rng = np.random.RandomState(7)
    cases = {}
    specs = [(48, 64, 5, 1), (48, 64, 5, 2), (48, 64, 6, 4), (40, 40, 8, 4), (33, 47, 3, 3),
             (32, 32, 1, 1)]
    for ci, (H, W, n, L) in enumerate(specs):
        label, _ = synth_label(rng, H, W, n)
        if ci == 3:  # stress: hand-made deep occlusion stack + stray bits
            label[5:20, 5:20] |= np.uint64(0b1011) << np.uint64(32)
            label[25:30, 25:30] = (np.uint64(1) << np.uint64(2)) | (np.uint64(0b11110011) << np.uint64(32))
        tmp = tempfile.mkdtemp()
        np.savez(os.path.join(tmp, "img.npz"), layer=label)

        class FakeSelf:
            image_info = [{"path": os.path.join(tmp, "img.jpg"), "height": H, "width": W}]

        cfgL = small_config(ref_config, 128, num_classes=L + 1)
        out = ref_train.AmodalDataset.load_layer2(FakeSelf(), 0, cfgL)
        if not isinstance(out, tuple) or out[0].ndim != 4:
            raise RuntimeError("unexpected load_layer2 output")
        mask_layers, class_ids = out  # [H,W,L,N] bool
        # load_image_gt tail (Functions.py:735) + Dataset.__getitem__ (model.py:114)
        ml = (np.swapaxes(mask_layers, 2, 3) > 0).astype("uint8")
        planes = ml.transpose(3, 2, 0, 1)  # [L,N,H,W]
        cases["label_%d" % ci] = label
        cases["planes_%d" % ci] = planes
        cases["L_%d" % ci] = np.array(L)
        cases["class_ids_%d" % ci] = class_ids
    save("label_decode", n_cases=np.array(len(specs)), **cases)

    # ------------------------------------------------------------- box ops
    g = torch.Generator().manual_seed(11)
    n = 257
    anc = torch.rand(n, 2, generator=g) * 200
    anc = torch.cat([anc, anc + torch.rand(n, 2, generator=g) * 150 + 1], 1)
    dl = torch.randn(n, 4, generator=g) * 0.7
    dec = ref_F.apply_box_deltas(anc.clone(), dl * torch.tensor([[0.1, 0.1, 0.2, 0.2]]))
    clip = ref_F.clip_boxes(dec, np.array([0, 0, 256, 256]).astype(np.float32))
    b1 = torch.rand(40, 2, generator=g) * 0.6
    b1 = torch.cat([b1, b1 + torch.rand(40, 2, generator=g) * 0.4], 1)
    b2 = torch.rand(9, 2, generator=g) * 0.6
    b2 = torch.cat([b2, b2 + torch.rand(9, 2, generator=g) * 0.4], 1)
    ov = ref_F.bbox_overlaps(b1, b2)
    refine = ref_utils.box_refinement(b1[:9], b2)
    save("box_ops", anchors=anc.numpy(), deltas=dl.numpy(), decoded=dec.numpy(),
         clipped=clip.numpy(), b1=b1.numpy(), b2=b2.numpy(), overlaps=ov.numpy(),
         refine=refine.numpy())

    # ------------------------------------------------------- proposal_layer
    for dim in (128, 256):
        cfg = small_config(ref_config, dim)
        anchors = torch.from_numpy(ref_utils.generate_pyramid_anchors(
            cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS, cfg.BACKBONE_SHAPES,
            cfg.BACKBONE_STRIDES, 1)).float()
        A = anchors.shape[0]
        # torch's sort is not stable: keep the fixture free of fg-score ties so the
        # visiting order is well defined ("Sort ties").
        g = torch.Generator().manual_seed(100 + dim)
        fg = (torch.randperm(A, generator=g).float() + 0.5) / A
        assert np.unique(fg.numpy()).size == A
        probs = torch.stack([1 - fg, fg], dim=1).unsqueeze(0)
        deltas = torch.randn(1, A, 4, generator=g) * 0.8
        rois = ref_F.proposal_layer([probs.clone(), deltas.clone()], proposal_count=1000,
                                    nms_threshold=0.7, anchors=anchors, config=cfg)
        save("proposal_layer_%d" % dim, native=np.array("oracle"), probs=probs.numpy(),
             deltas=deltas.numpy(), anchors=anchors.numpy(), rois=rois.numpy())

    # --------------------------------------------------- pyramid_roi_align
    g = torch.Generator().manual_seed(5)
    C = 8
    maps = [torch.randn(1, C, s, s, generator=g, requires_grad=True) for s in (64, 32, 16, 8)]
    R = 60
    ctr = torch.rand(R, 2, generator=g)
    size = torch.exp(torch.rand(R, 2, generator=g) * 4.5 - 4.6)
    boxes_big = None  # spans levels 2..5 + clamps
    boxes = torch.cat([ctr - size / 2, ctr + size / 2], 1).clamp(0, 1)
    boxes[0] = torch.tensor([0.1, 0.1, 0.1 + 112 / 1024, 0.1 + 112 / 1024])  # exact level edges
    boxes[1] = torch.tensor([0.2, 0.2, 0.2 + 224 / 1024, 0.2 + 224 / 1024]).clamp(0, 1)
    boxes[2] = torch.tensor([0.3, 0.3, 0.3 + 56 / 1024, 0.3 + 56 / 1024])
    pooled = ref_modals.pyramid_roi_align([boxes.unsqueeze(0)] + list(maps), 7, (1024, 1024, 3))
    up = torch.randn(pooled.shape, generator=g)
    pooled.backward(up)
    h, w = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    lvl = 4 + ref_modals.log2(torch.sqrt(h * w) / (224.0 / torch.sqrt(torch.tensor([1024.0 * 1024]))))
    lvl = lvl.round().int().clamp(2, 5)
    save("pyramid_roi_align", native=np.array("oracle"), boxes=boxes.numpy(),
         levels=lvl.numpy(), pooled=pooled.detach().numpy(), upstream=up.numpy(),
         **{"map%d" % i: m.detach().numpy() for i, m in enumerate(maps)},
         **{"grad%d" % i: (m.grad.numpy() if m.grad is not None else np.zeros(m.shape, np.float32))
            for i, m in enumerate(maps)})

    # ----------------------------------------------- detection_target_layer
    cfg = small_config(ref_config, 128)
    rng = np.random.RandomState(21)
    H = W = 128
    label, amodal = synth_label(rng, H, W, 5)
    from oracle import oracle as orc
    for case, L in (("a", 1), ("b", 3)):
        planes = orc.label_decode(label, L)  # [L,N,H,W]; decode itself pinned above
        N = planes.shape[1]
        gt_boxes = []
        am = planes.sum(axis=0) > 0
        for i in range(N):
            ys, xs = np.where(am[i])
            gt_boxes.append([ys.min(), xs.min(), ys.max() + 1, xs.max() + 1])
        gt_boxes = torch.tensor(gt_boxes, dtype=torch.float32) / 128.0
        g = torch.Generator().manual_seed(31)
        P = 300
        jit = torch.randn(P, 4, generator=g) * 0.04
        props = gt_boxes[torch.randint(0, N, (P,), generator=g)] + jit
        props[200:] = torch.rand(100, 4, generator=g).sort(dim=1)[0][:, [0, 1, 2, 3]]
        props = torch.stack([torch.minimum(props[:, 0], props[:, 2]), torch.minimum(props[:, 1], props[:, 3]),
                             torch.maximum(props[:, 0], props[:, 2]) + 0.02,
                             torch.maximum(props[:, 1], props[:, 3]) + 0.02], 1).clamp(0, 1)
        draws = []
        real = torch.randperm

        def rec(n, *a, **k):
            p = real(n, *a, **k)
            draws.append(p.numpy().copy())
            return p

        torch.randperm = rec
        try:
            rois, cls, dl, masks = ref_F.detection_target_layer(
                props.unsqueeze(0), torch.ones(1, N, dtype=torch.int32), gt_boxes.unsqueeze(0),
                torch.from_numpy(planes).unsqueeze(0), cfg)
        finally:
            torch.randperm = real
        save("detection_target_%s" % case, native=np.array("oracle"), proposals=props.numpy(),
             gt_boxes=gt_boxes.numpy(), label=label, L=np.array(L), perm_pos=draws[0],
             perm_neg=draws[1], rois=rois.numpy(), class_ids=cls.numpy(),
             deltas=dl.numpy(), masks=masks.numpy())

    # ---------------------------------------------------- build_rpn_targets
    for dim in (128, 256):
        cfg = small_config(ref_config, dim)
        anchors = ref_utils.generate_pyramid_anchors(cfg.RPN_ANCHOR_SCALES, cfg.RPN_ANCHOR_RATIOS,
                                                     cfg.BACKBONE_SHAPES, cfg.BACKBONE_STRIDES, 1)
        rng = np.random.RandomState(dim)
        N = 6
        tl = rng.randint(0, dim // 2, size=(N, 2))
        wh = rng.randint(dim // 10, dim // 2, size=(N, 2))
        gt = np.concatenate([tl, np.minimum(tl + wh, dim)], 1).astype(np.int32)
        draws = []
        real_choice = np.random.choice

        def rec_choice(ids, extra, replace=False):
            r = real_choice(ids, extra, replace=replace)
            draws.append(np.asarray(r).copy())
            return r

        np.random.seed(dim)
        np.random.choice = rec_choice
        try:
            match, bbox = ref_F.build_rpn_targets((dim, dim, 3), anchors,
                                                  np.ones(N, np.int32), gt, cfg)
        finally:
            np.random.choice = real_choice
        save("rpn_targets_%d" % dim, gt_boxes=gt, rpn_match=match, rpn_bbox=bbox,
             n_draws=np.array(len(draws)),
             **{"draw%d" % i: d for i, d in enumerate(draws)})

    # ----------------------------------------------------------------- losses
    g = torch.Generator().manual_seed(77)
    A, R, P, L = 500, 40, 25, 1
    rpn_match = torch.zeros(1, A, 1, dtype=torch.int32)
    idx = torch.randperm(A, generator=g)
    rpn_match[0, idx[:30], 0] = 1
    rpn_match[0, idx[30:130], 0] = -1
    rpn_logits = torch.randn(1, A, 2, generator=g, requires_grad=True)
    rpn_bbox_t = torch.zeros(1, 256, 4)
    rpn_bbox_t[0, :30] = torch.randn(30, 4, generator=g)
    rpn_bbox_p = torch.randn(1, A, 4, generator=g, requires_grad=True)
    tcls = torch.cat([torch.ones(P), torch.zeros(R - P)]).int()
    cls_logits = torch.randn(R, 2, generator=g, requires_grad=True)
    tdl = torch.cat([torch.randn(P, 4, generator=g), torch.zeros(R - P, 4)])
    pdl = torch.randn(R, 2, 4, generator=g, requires_grad=True)
    tmask = torch.cat([(torch.rand(P, L, 32, 32, generator=g) > 0.5).float(),
                       torch.zeros(R - P, L, 32, 32)])
    pmask = torch.randn(R, 1 + L, 32, 32, generator=g, requires_grad=True)
    l_rc = ref_loss.compute_rpn_class_loss(rpn_match, rpn_logits)
    l_rb = ref_loss.compute_rpn_bbox_loss(rpn_bbox_t, rpn_match, rpn_bbox_p)
    l_mc = ref_loss.compute_mrcnn_class_loss(tcls, cls_logits)
    l_mb = ref_loss.compute_mrcnn_bbox_loss(tdl, tcls, pdl)
    l_ly, _, _ = ref_loss.compute_layer_loss(tmask, tcls, pmask)
    l_am, _ = ref_loss.compute_amodal_loss(tmask, tcls, pmask)
    total = l_ly + l_rb + l_mb + l_mc + l_am + l_rc
    total.backward()
    save("losses", rpn_match=rpn_match.numpy(), rpn_logits=rpn_logits.detach().numpy(),
         rpn_bbox_t=rpn_bbox_t.numpy(), rpn_bbox_p=rpn_bbox_p.detach().numpy(),
         tcls=tcls.numpy(), cls_logits=cls_logits.detach().numpy(), tdl=tdl.numpy(),
         pdl=pdl.detach().numpy(), tmask=tmask.numpy(), pmask=pmask.detach().numpy(),
         losses=np.array([l_rc.item(), l_rb.item(), l_mc.item(), l_mb.item(), l_ly.item(),
                          l_am.item()], dtype=np.float64),
         g_rpn_logits=rpn_logits.grad.numpy(), g_rpn_bbox=rpn_bbox_p.grad.numpy(),
         g_cls_logits=cls_logits.grad.numpy(), g_pdl=pdl.grad.numpy(),
         g_pmask=pmask.grad.numpy())


if __name__ == "__main__" and not any(a in sys.argv for a in ("--modules", "--module-grads", "--relayer", "--tail")):
    main()


def module_goldens():
    """Module-level goldens: the reference's nn.Modules (CPU, torch 2.10) with the
    name-keyed deterministic initialisation of tests/_util.key_init_."""
    import json
    ref_modals, ref_F = ref_harness.install()
    from tests._util import key_init_
    import config as ref_config
    import model as ref_model
    import modal.deeplabv2 as ref_dl
    torch.manual_seed(0)
    # ---- FPN (ResNet-50) + RPN on a 64x64 image ----
    resnet = ref_modals.ResNet("resnet50", stage5=True)
    fpn = ref_modals.FPN(*resnet.stages(), out_channels=256).eval()
    rpn = ref_modals.RPN(3, 1, 256).eval()
    key_init_(fpn); key_init_(rpn)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 64, 64, generator=g)
    with torch.no_grad():
        p = fpn(x)
        r = rpn(p[0])
    print("fpn out abs mean", [float(t.abs().mean()) for t in p], "rpn", float(r[0].abs().mean()))
    save("module_fpn_rpn", x=x.numpy(), p2=p[0].numpy(), p3=p[1].numpy(), p5=p[3].numpy(),
         p6=p[4].numpy(), rpn_logits=r[0].numpy(), rpn_probs=r[1].numpy(), rpn_bbox=r[2].numpy())
    # ---- heads on fixed maps / rois (native ops: oracle) ----
    C = 256
    maps = [torch.randn(1, C, s, s, generator=g) * 0.5 for s in (32, 16, 8, 4)]
    R = 12
    ctr = torch.rand(R, 2, generator=g) * 0.6 + 0.2
    size = torch.exp(torch.rand(R, 2, generator=g) * 3.0 - 3.2)
    rois = torch.cat([ctr - size / 2, ctr + size / 2], 1).clamp(0, 1).unsqueeze(0)
    cls = ref_modals.Classifier(256, 7, (128, 128, 3), 2).eval()
    msk = ref_modals.Mask(256, 16, (128, 128, 3), 2).eval()
    msk.conv1 = torch.nn.Conv2d(439, 256, kernel_size=3, stride=1)
    key_init_(cls); key_init_(msk)
    glm_feat = torch.randn(R, 183, 16, 16, generator=g) * 0.3
    with torch.no_grad():
        c_out = cls([m.clone() for m in maps], rois.clone())
        m_out, m_feat = msk([m.clone() for m in maps], rois.clone(), glm_feat)
    save("module_heads", native=np.array("oracle"), rois=rois.numpy(), glm_feat=glm_feat.numpy(),
         cls_logits=c_out[0].numpy(), cls_probs=c_out[1].numpy(), cls_bbox=c_out[2].numpy(),
         mask_logits=m_out.numpy(), **{"map%d" % i: m.numpy() for i, m in enumerate(maps)})
    # ---- DeepLab-v2 MSC (ResNet-101) on a 97x97 image ----
    glm = ref_dl.DeepLabV2_ResNet101_MSC(182).eval()
    key_init_(glm)
    xg = torch.randn(1, 3, 97, 97, generator=g)
    with torch.no_grad():
        lg = glm(xg)
    print("glm logits", tuple(lg.shape), float(lg.abs().mean()))
    save("module_glm", x=xg.numpy(), logits=lg.numpy())
    # ---- state-dict keys and shapes of the model after the amodal head surgery ----
    cfg = small_config(ref_config, 128, num_classes=81)
    m = ref_model.MaskRCNN(cfg, tempfile.mkdtemp())
    cfg.NUM_CLASSES = 2
    m.mask.conv1 = torch.nn.Conv2d(439, 256, kernel_size=3, stride=1)
    m.mask.conv5 = torch.nn.Conv2d(256, 2, kernel_size=1, stride=1)
    m.classifier.linear_class = torch.nn.Linear(1024, 2)
    m.classifier.linear_bbox = torch.nn.Linear(1024, 8)
    m.GLM_modual = glm
    keys = {k: list(v.shape) for k, v in m.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f)
    print("state dict keys", len(keys))


if __name__ == "__main__" and "--modules" in sys.argv:
    module_goldens()


def module_grad_goldens():
    """Backward goldens of the reference's nn.Modules ("outputs/grads <= 1e-4"):
    FPN + RPN (ResNet-50, 64x64 input), Classifier and Mask on fixed maps / rois.  Loss = sum of
    <output, seeded upstream>; recorded: the input gradient(s) and leading slices of weight gradients."""
    ref_modals, ref_F = ref_harness.install()
    from tests._util import key_init_
    n_keep = 512
    g = torch.Generator().manual_seed(41)
    # ---- FPN + RPN ----
    resnet = ref_modals.ResNet("resnet50", stage5=True)
    fpn = ref_modals.FPN(*resnet.stages(), out_channels=256).eval()
    rpn = ref_modals.RPN(3, 1, 256).eval()
    key_init_(fpn); key_init_(rpn)
    for m in list(fpn.modules()) + list(rpn.modules()):     # model.py:192-197: BN never trains
        if isinstance(m, torch.nn.BatchNorm2d):
            for p in m.parameters():
                p.requires_grad = False
    x = torch.randn(2, 3, 64, 64, generator=g, requires_grad=True)
    p = fpn(x)
    outs = [rpn(t) for t in p]
    logits = torch.cat([o[0] for o in outs], 1)
    bbox = torch.cat([o[2] for o in outs], 1)
    up_l = torch.randn(logits.shape, generator=g)
    up_b = torch.randn(bbox.shape, generator=g)
    up_p = [torch.randn(t.shape, generator=g) * 0.1 for t in p[:4]]
    loss = (logits * up_l).sum() + (bbox * up_b).sum() + sum((t * u).sum() for t, u in zip(p[:4], up_p))
    loss.backward()
    names = ["C1.0.weight", "C2.0.conv1.weight", "C2.2.conv2.weight", "C3.0.downsample.0.weight",
             "C4.3.conv2.weight", "C5.2.conv3.weight", "C5.2.conv3.bias", "P5_conv1.weight",
             "P3_conv1.weight", "P2_conv2.1.weight", "P2_conv2.1.bias"]
    fp = dict(fpn.named_parameters()); rp = dict(rpn.named_parameters())
    arrs = {"x": x.detach().numpy(), "gx": x.grad.numpy(), "up_logits": up_l.numpy(), "up_bbox": up_b.numpy(),
            "loss": np.array(loss.item()), "fpn_names": np.array(names), "rpn_names": np.array(list(rp))}
    for i, u in enumerate(up_p):
        arrs["up_p%d" % i] = u.numpy()
    for n in names:
        arrs["fpn_g/" + n] = fp[n].grad.reshape(-1)[:n_keep].numpy()
        arrs["fpn_gn/" + n] = np.array(float(fp[n].grad.norm()))
    for n in rp:
        arrs["rpn_g/" + n] = rp[n].grad.reshape(-1)[:n_keep].numpy()
        arrs["rpn_gn/" + n] = np.array(float(rp[n].grad.norm()))
    print("fpn/rpn loss", loss.item(), "gx", float(x.grad.abs().mean()))
    save("module_grads_fpn_rpn", **arrs)
    # ---- heads ----
    C = 256
    maps = [(torch.randn(1, C, s, s, generator=g) * 0.5).requires_grad_(True) for s in (32, 16, 8, 4)]
    R = 12
    ctr = torch.rand(R, 2, generator=g) * 0.6 + 0.2
    size = torch.exp(torch.rand(R, 2, generator=g) * 3.0 - 3.2)
    rois = torch.cat([ctr - size / 2, ctr + size / 2], 1).clamp(0, 1).unsqueeze(0)
    cls = ref_modals.Classifier(256, 7, (128, 128, 3), 2).eval()
    msk = ref_modals.Mask(256, 16, (128, 128, 3), 2).eval()
    msk.conv1 = torch.nn.Conv2d(439, 256, kernel_size=3, stride=1)
    key_init_(cls); key_init_(msk)
    for m in list(cls.modules()) + list(msk.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):
            for p_ in m.parameters():
                p_.requires_grad = False
    glm_feat = torch.randn(R, 183, 16, 16, generator=g) * 0.3
    c_out = cls(list(maps), rois.clone())
    m_out, _ = msk(list(maps), rois.clone(), glm_feat)
    up_c = torch.randn(c_out[0].shape, generator=g)
    up_bb = torch.randn(c_out[2].shape, generator=g)
    up_m = torch.randn(m_out.shape, generator=g) * 0.05
    loss = (c_out[0] * up_c).sum() + (c_out[2] * up_bb).sum() + (m_out * up_m).sum()
    loss.backward()
    cp = dict(cls.named_parameters()); mp = dict(msk.named_parameters())
    arrs = {"native": np.array("oracle"), "rois": rois.numpy(), "glm_feat": glm_feat.numpy(),
            "up_cls": up_c.numpy(), "up_bbox": up_bb.numpy(), "up_mask": up_m.numpy(),
            "loss": np.array(loss.item()),
            "cls_names": np.array([n for n, p_ in cp.items() if p_.grad is not None]),
            "mask_names": np.array([n for n, p_ in mp.items() if p_.grad is not None])}
    for i, m in enumerate(maps):
        arrs["map%d" % i] = m.detach().numpy()
        arrs["gmap%d" % i] = m.grad.numpy() if m.grad is not None else np.zeros(m.shape, np.float32)
    for n, p_ in cp.items():
        if p_.grad is not None:
            arrs["cls_g/" + n] = p_.grad.reshape(-1)[:n_keep].numpy()
            arrs["cls_gn/" + n] = np.array(float(p_.grad.norm()))
    for n, p_ in mp.items():
        if p_.grad is not None:
            arrs["mask_g/" + n] = p_.grad.reshape(-1)[:n_keep].numpy()
            arrs["mask_gn/" + n] = np.array(float(p_.grad.norm()))
    print("heads loss", loss.item())
    save("module_grads_heads", **arrs)


if __name__ == "__main__" and "--module-grads" in sys.argv:
    module_grad_goldens()


def relayer_goldens():
    """utils.reLayerMask (+ remove_small_path) of the reference on synthetic amodal / invisible masks.
    skimage is absent: the harness supplies morphology.remove_small_objects restated from skimage's
    documented behaviour with scipy.ndimage (connectivity 1, components smaller than min_size removed)."""
    ref_harness.install()
    import sys as _sys
    from scipy import ndimage

    def remove_small_objects(ar, min_size=64, connectivity=1, **_):
        comp, n = ndimage.label(ar)
        sizes = np.bincount(comp.ravel())
        small = sizes < min_size
        small[0] = False
        out = ar.copy()
        out[small[comp]] = False
        return out

    _sys.modules["skimage.morphology"].remove_small_objects = remove_small_objects
    import utils as ref_utils
    ref_utils.morphology.remove_small_objects = remove_small_objects
    rng = np.random.RandomState(13)
    cases = {}
    for ci, (H, W, n) in enumerate([(64, 80, 4), (96, 96, 6), (48, 48, 3)]):
        yy, xx = np.mgrid[0:H, 0:W]
        amodal, invis = [], []
        covered = np.zeros((H, W), bool)
        for i in range(n):
            cy, cx = rng.uniform(0.2, 0.8) * H, rng.uniform(0.2, 0.8) * W
            ry, rx = rng.uniform(0.08, 0.3) * H, rng.uniform(0.08, 0.3) * W
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            if i == n - 1:                      # a speck smaller than min_size: its colour must vanish
                m = np.zeros((H, W), bool); m[2:6, 3:8] = True
            amodal.append(m.astype(np.uint8))
            inv = (m & covered).astype(np.uint8)
            invis.append(inv if inv.any() else np.zeros((0,), np.uint8))
            covered |= m
        lab = ref_utils.reLayerMask([a.copy() for a in amodal], [v.copy() for v in invis])
        cases["amodal_%d" % ci] = np.stack(amodal)
        cases["invis_%d" % ci] = np.stack([v if v.size else np.zeros((H, W), np.uint8) for v in invis])
        cases["has_invis_%d" % ci] = np.array([v.size > 0 for v in invis])
        cases["label_%d" % ci] = lab
    save("relayer_mask", n_cases=np.array(3), **cases)


if __name__ == "__main__" and "--relayer" in sys.argv:
    relayer_goldens()


def tail_goldens():
    """Inference tail.
    rle.npz: the reference's own cocoapi/common/maskApi.c (compiled as oracle/_ref/libmaskapi_ref.so by
    oracle/Makefile) run on masks handed over exactly like amodal_train.py:397 (np.asfortranarray) ->
    run counts + compressed strings.
    unmold.npz: the reference's utils.unmold_mask (utils.py:447-465) on float32 head outputs; its
    scipy.misc.imresize (removed from scipy >= 1.3) is the harness stand-in of tools/gen_golden_e2e.py
    (scipy 1.0's published bytescale -> PIL resize), so the resampling itself is the installed Pillow's."""
    from oracle import oracle as orc
    assert orc.ref_maskapi() is not None, "oracle/_ref/libmaskapi_ref.so missing"
    rng = np.random.RandomState(77)
    yy, xx = np.mgrid[0:96, 0:128]
    masks = {
        "empty": np.zeros((6, 9), np.uint8),
        "full": np.ones((6, 9), np.uint8),                       # first run (zeros) is empty
        "one_pixel": np.eye(1, 40, 17, dtype=np.uint8).reshape(5, 8),
        "ellipse": (((yy - 40) / 30.0) ** 2 + ((xx - 70) / 45.0) ** 2 <= 1).astype(np.uint8),
        "noise": (rng.rand(37, 23) > 0.5).astype(np.uint8),
        "checker": ((yy[:16, :16] + xx[:16, :16]) & 1).astype(np.uint8),
        "sparse": (rng.rand(64, 64) > 0.98).astype(np.uint8),
        "last_pixel": np.eye(1, 35, 34, dtype=np.uint8).reshape(5, 7),
    }
    big = np.zeros((300, 400), np.uint8)
    big[:, :250] = 1                                              # 75000-long run: 4-char groups
    big[10:20, 300:310] = 1
    masks["long_runs"] = big
    shr = np.zeros((200, 3), np.uint8)                            # shrinking runs: negative deltas
    shr[:150, 0] = 1; shr[100:, 1] = 1; shr[5:8, 2] = 1
    masks["negative_delta"] = shr
    arrs = {"names": np.array(sorted(masks))}
    for name in sorted(masks):
        cnts, s = orc.ref_rle_encode(masks[name])
        arrs["mask/" + name] = masks[name]
        arrs["counts/" + name] = cnts
        arrs["string/" + name] = np.frombuffer(s, np.uint8)
    save("rle", **arrs)

    ref_harness.install()
    import scipy.misc
    from tools.gen_golden_e2e import imresize
    scipy.misc.imresize = imresize
    import utils as ref_utils
    ref_utils.scipy.misc.imresize = imresize
    H, W = 160, 192
    boxes = [(10, 20, 100, 150), (0, 0, 160, 192), (50, 60, 82, 92), (30, 40, 37, 190), (5, 5, 6, 6),
             (100, 3, 159, 20), (20, 100, 140, 111), (64, 64, 96, 160), (0, 170, 33, 192), (90, 90, 93, 95)]
    ms, fulls = [], []
    for i, b in enumerate(boxes):
        m = rng.randn(32, 32).astype(np.float32)
        if i % 3 == 0:
            m = (1 / (1 + np.exp(-3 * m))).astype(np.float32)     # sigmoid-like head output
        if i == 4:
            m[:] = 0.25                                            # constant mask: cscale == 0 branch
        if i % 3 == 1:                                             # smooth blob like a trained head
            gy, gx = np.mgrid[0:32, 0:32]
            m = np.exp(-(((gy - 15.5) / 9.0) ** 2 + ((gx - 14) / 11.0) ** 2)).astype(np.float32)
        ms.append(m)
        fulls.append(ref_utils.unmold_mask(m.copy(), np.array(b, np.int32), (H, W, 3)))
    save("unmold", masks=np.stack(ms), boxes=np.array(boxes, np.int32), image_shape=np.array([H, W, 3]),
         full=np.packbits(np.stack(fulls), axis=-1), pillow=np.array(__import__("PIL").__version__))


if __name__ == "__main__" and "--tail" in sys.argv:
    tail_goldens()
