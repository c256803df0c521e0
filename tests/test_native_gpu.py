"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through
the C ABI, against the CPU oracle on the same seeded inputs and against the
committed golden vectors.  Integer / index outputs must be bit-exact."""
import numpy as np
import pytest
import torch

from tests._util import golden, seeded_dets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


# ---------------------------------------------------------------------------- NMS
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 6000])
@pytest.mark.parametrize("thr", [0.3, 0.7])
def test_nms_bit_exact_vs_oracle(orc, n, thr):
    from sln_amodal_amd.nms.nms_wrapper import nms
    dets = seeded_dets(n, seed=1000 + n, span=1024.0)
    want = orc.nms(dets, thr)
    got = nms(dev(dets), thr)
    assert got.dtype == torch.int64 and got.is_cuda
    assert np.array_equal(got.cpu().numpy(), want)


def test_nms_empty_edge_ties_and_degenerate(orc):
    from sln_amodal_amd.nms.nms_wrapper import nms
    assert nms(torch.zeros(0, 5).cuda(), 0.5).numel() == 0
    d = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8], [50, 50, 50, 50, 0.7],
                  [50, 50, 50, 50, 0.6], [0, 0, 9, 9, 0.9]], np.float32)
    for thr in (0.5, 0.5000001, 0.0, 1.0):
        assert np.array_equal(nms(dev(d), thr).cpu().numpy(), orc.nms(d, thr)), thr
    with pytest.raises(RuntimeError):
        nms(torch.zeros(4, 4).cuda(), 0.5)


def test_nms_batched_fixed_capacity(orc):
    from sln_amodal_amd import ops
    B, N, cap = 5, 777, 200
    d = np.stack([seeded_dets(N, seed=50 + b, span=300.0) for b in range(B)])
    for b in range(B):  # API contract: rows sorted by score, descending
        d[b] = d[b][np.argsort(-d[b][:, 4], kind="stable")]
    nv = np.array([N, 0, 1, 64, 500], np.int32)
    keep, num = ops.nms_sorted(dev(d), 0.6, cap, dev(nv))
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for b in range(B):
        want = orc.nms(d[b][: nv[b]], 0.6)[:cap]
        assert num[b] == len(want)
        assert np.array_equal(keep[b, : num[b]], want)
        assert (keep[b, num[b]:] == -1).all()


def test_nms_full_size_properties():
    """BASELINE size (16 x 6000): idempotence + pairwise IoU of survivors < thr."""
    from sln_amodal_amd import ops
    B, N, thr = 16, 6000, 0.7
    d = np.stack([seeded_dets(N, seed=900 + b, span=1024.0) for b in range(B)])
    for b in range(B):
        d[b] = d[b][np.argsort(-d[b][:, 4], kind="stable")]
    dd = dev(d)
    keep, num = ops.nms_sorted(dd, thr, N)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for b in (0, 7, 15):
        k = keep[b, : num[b]]
        assert (np.diff(k) > 0).all()              # visiting order == sorted order
        surv = d[b][k]
        keep2, num2 = ops.nms_sorted(dev(surv[None]), thr, len(k))
        assert int(num2[0]) == len(k)              # NMS of survivors keeps them all
        assert np.array_equal(keep2[0].cpu().numpy(), np.arange(len(k)))


# ---------------------------------------------------------------- crop_and_resize
def _rand_boxes(g, K, oob=True):
    tl = torch.rand(K, 2, generator=g) * (0.9 if oob else 0.6) - (0.15 if oob else 0.0)
    wh = torch.rand(K, 2, generator=g) * 0.6 + 0.01
    return torch.cat([tl, tl + wh], 1)


@pytest.mark.parametrize("C", [1, 3, 183, 256])
@pytest.mark.parametrize("pool", [7, 16, 32])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_crop_forward_bit_exact_vs_oracle(orc, C, pool, layout):
    from sln_amodal_amd.roialign.roi_align.crop_and_resize import CropAndResizeFunction
    g = torch.Generator().manual_seed(C * 100 + pool)
    img = torch.randn(2, C, 37, 41, generator=g)
    K = 23
    boxes = _rand_boxes(g, K)
    boxes[0] = torch.tensor([0.0, 0.0, 1.0, 1.0])
    boxes[1] = torch.tensor([0.3, 0.3, 0.3, 0.3])       # degenerate
    boxes[2] = torch.tensor([0.5, 0.5, 0.25, 0.25])     # inverted
    ind = torch.randint(0, 2, (K,), generator=g).int()
    want = orc.crop_and_resize_fwd(img.numpy(), boxes.numpy(), ind.numpy(), pool, pool, 0.0)
    x = img.cuda()
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    got = CropAndResizeFunction(pool, pool, 0)(x, boxes.cuda(), ind.cuda())
    assert tuple(got.shape) == want.shape
    if layout == "nhwc" and C > 1:
        assert got.is_contiguous(memory_format=torch.channels_last)
    assert np.array_equal(got.cpu().numpy(), want)


def test_crop_single_bin_and_extrapolation_value(orc):
    from sln_amodal_amd.roialign.roi_align.crop_and_resize import CropAndResize
    g = torch.Generator().manual_seed(4)
    img = torch.randn(3, 8, 20, 20, generator=g)
    boxes = _rand_boxes(g, 11)
    ind = torch.randint(0, 3, (11,), generator=g).int()
    for (ch, cw, ev) in [(1, 1, 0.0), (1, 5, -3.5), (4, 1, 2.0)]:
        want = orc.crop_and_resize_fwd(img.numpy(), boxes.numpy(), ind.numpy(), ch, cw, ev)
        for x in (img.cuda(), img.cuda().contiguous(memory_format=torch.channels_last)):
            got = CropAndResize(ch, cw, ev)(x, boxes.cuda(), ind.cuda())
            assert np.array_equal(got.cpu().numpy(), want)


def test_crop_bad_box_index_raises_when_validated():
    import sln_amodal_amd.roialign.roi_align.crop_and_resize as car
    img = torch.randn(2, 4, 8, 8).cuda()
    boxes = torch.tensor([[0, 0, 1, 1.0], [0, 0, 1, 1.0]]).cuda()
    car.VALIDATE = True
    try:
        with pytest.raises(RuntimeError):
            car.CropAndResizeFunction(3, 3, 0)(img, boxes, torch.tensor([0, 2], dtype=torch.int32).cuda())
        car.CropAndResizeFunction(3, 3, 0)(img, boxes, torch.tensor([0, 1], dtype=torch.int32).cuda())
    finally:
        car.VALIDATE = False


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_crop_backward_vs_oracle(orc, layout):
    from sln_amodal_amd.roialign.roi_align.crop_and_resize import CropAndResizeFunction
    g = torch.Generator().manual_seed(12)
    img = torch.randn(2, 19, 21, 18, generator=g)
    K = 40
    boxes = _rand_boxes(g, K)
    ind = torch.randint(0, 2, (K,), generator=g).int()
    up = torch.randn(K, 19, 7, 7, generator=g)
    want = orc.crop_and_resize_bwd(up.numpy(), boxes.numpy(), ind.numpy(), tuple(img.shape))
    x = img.cuda()
    if layout == "nhwc":
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    out = CropAndResizeFunction(7, 7, 0)(x, boxes.cuda(), ind.cuda())
    out.backward(up.cuda())
    got = x.grad.cpu().numpy()
    # fp32 atomics: summation order differs from the serial CPU loop -> 1e-5
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)


def test_pyramid_golden_through_hip_crop():
    """Golden produced by the reference's pyramid_roi_align graph: per-level crops
    of the HIP op reproduce pooled outputs and feature-map gradients."""
    from sln_amodal_amd.roialign.roi_align.crop_and_resize import CropAndResizeFunction
    g = golden("pyramid_roi_align")
    lv = g["levels"]
    for i, level in enumerate(range(2, 6)):
        ix = np.nonzero(lv == level)[0]
        fm = dev(g["map%d" % i]).requires_grad_(True)
        out = CropAndResizeFunction(7, 7, 0)(fm, dev(g["boxes"][ix]),
                                             torch.zeros(len(ix), dtype=torch.int32).cuda())
        assert np.array_equal(out.detach().cpu().numpy(), g["pooled"][ix])
        out.backward(dev(g["upstream"][ix]))
        assert np.allclose(fm.grad.cpu().numpy(), g["grad%d" % i], rtol=1e-5, atol=1e-5)


def test_pyramid_crops_sharing_gradient_maps_equal_separate_backward():
    """CropGradPool: the classifier (7x7) and mask (16x16, cropped into a wider buffer) crops of the same
    maps accumulate into one set of gradient maps; the sum autograd sees equals the sum of the two
    separately computed gradients, for either backward order and for a second backward (retain_graph)."""
    from sln_amodal_amd.modal.modals import CropGradPool, pyramid_roi_align
    gen = torch.Generator().manual_seed(3)
    B, C, R = 2, 64, 40
    sizes = [(64, 64), (32, 32), (16, 16), (8, 8)]
    ctr = torch.rand(B, R, 2, generator=gen) * 0.6 + 0.2
    half = torch.rand(B, R, 2, generator=gen) * 0.25 + 0.01
    rois = torch.cat([ctr - half, ctr + half], dim=2).clamp(0, 1).cuda()
    base = [torch.randn(B, C, h, w, generator=gen).cuda().contiguous(memory_format=torch.channels_last)
            for h, w in sizes]
    up7 = torch.randn(B * R, C, 7, 7, generator=gen).cuda()
    up16 = torch.randn(B * R, C + 8, 16, 16, generator=gen).cuda()

    def run(pool, order):
        maps = [m.clone().requires_grad_(True) for m in base]
        a = pyramid_roi_align([rois] + maps, 7, (256, 256), grad_pool=pool)
        wide = torch.zeros((B * R, C + 8, 16, 16), device="cuda").contiguous(memory_format=torch.channels_last)
        b = pyramid_roi_align([rois] + maps, 16, (256, 256), into=(wide, 8), grad_pool=pool)
        terms = [(a * up7).sum(), (b * up16).sum()]
        loss = terms[order[0]] + terms[order[1]]
        loss.backward(retain_graph=True)
        first = [m.grad.clone() for m in maps]
        for m in maps:
            m.grad = None
        loss.backward()
        return first, [m.grad for m in maps]

    want, _ = run(None, (0, 1))
    for order in ((0, 1), (1, 0)):
        pool = CropGradPool()
        got, again = run(pool, order)
        assert pool.registered == 2 and pool.pending == 2 and pool.bufs is None
        for w, g, g2 in zip(want, got, again):
            scale = w.abs().max().item()
            assert (w - g).abs().max().item() <= 1e-5 * scale
            assert (w - g2).abs().max().item() <= 1e-5 * scale


@pytest.mark.parametrize("pool", [16, 8, 14, 7])
def test_pyramid_crop_backward_vs_oracle_all_regimes(orc, pool):
    """sln_pyramid_crop_bwd_f32 against the oracle's serial crop_and_resize backward, one
    level at a time: tiny rois (many bins per pixel), level-sized rois, whole-map rois
    (footprint > crop -> scatter branch), rois partly outside the map (skipped bins),
    padded slots, a channel count that is not a multiple of the 64-channel block."""
    from sln_amodal_amd.modal.modals import _PyramidCrop
    gen = torch.Generator().manual_seed(pool)
    B, C = 2, 72
    sizes = [(64, 64), (32, 32), (16, 16), (8, 8)]
    K = 96
    ctr = torch.rand(K, 2, generator=gen)
    half = torch.cat([torch.rand(24, 2, generator=gen) * 0.02,          # tiny
                      torch.rand(40, 2, generator=gen) * 0.15 + 0.03,   # level sized
                      torch.rand(16, 2, generator=gen) * 0.5 + 0.3,     # huge / out of bounds
                      torch.rand(16, 2, generator=gen) * 0.1])
    boxes = torch.cat([ctr - half, ctr + half], dim=1)[:, [0, 1, 2, 3]].float()
    boxes[90] = torch.tensor([0.0, 0.0, 1.0, 1.0])
    boxes[91] = torch.tensor([0.25, 0.25, 0.25, 0.25])                 # zero extent
    level = torch.randint(2, 6, (K,), generator=gen).int()
    ind = torch.randint(0, B, (K,), generator=gen).int()
    ind[5] = -1                                                         # padded slot
    maps = [torch.randn(B, C, h, w, generator=gen).cuda().contiguous(memory_format=torch.channels_last)
            .requires_grad_(True) for h, w in sizes]
    out = _PyramidCrop.apply(boxes.cuda(), ind.cuda(), level.cuda(), pool, None, *maps)
    up = torch.randn(K, C, pool, pool, generator=gen)
    out.backward(up.cuda().contiguous(memory_format=torch.channels_last))
    for i, (h, w) in enumerate(sizes):
        sel = np.nonzero((level.numpy() == i + 2) & (ind.numpy() >= 0))[0]
        want = orc.crop_and_resize_bwd(up.numpy()[sel], boxes.numpy()[sel], ind.numpy()[sel], (B, C, h, w))
        got = maps[i].grad.cpu().numpy()
        assert np.allclose(got, want, rtol=1e-5, atol=1e-5), (i, np.abs(got - want).max())
        fwd = orc.crop_and_resize_fwd(maps[i].detach().cpu().numpy(), boxes.numpy()[sel], ind.numpy()[sel],
                                      pool, pool)
        assert np.array_equal(out.detach().cpu().numpy()[sel], fwd)


def test_crop_full_size_linearity_and_adjoint():
    """BASELINE size: [16,256,256,256] P2 map, 1600 rois, 16x16 bins (NHWC)."""
    from sln_amodal_amd.roialign.roi_align.crop_and_resize import CropAndResizeFunction
    g = torch.Generator().manual_seed(1)
    B, C, H, W, K = 16, 256, 256, 256, 1600
    a = torch.randn(B, C, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(B, C, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    boxes = _rand_boxes(g, K, oob=False).cuda()
    ind = torch.randint(0, B, (K,), generator=g).int().cuda()
    f = CropAndResizeFunction(16, 16, 0)
    fa, fb, fab = f(a, boxes, ind), f(b, boxes, ind), f(a + 2 * b, boxes, ind)
    assert torch.allclose(fab, fa + 2 * fb, rtol=1e-4, atol=1e-4)
    a.requires_grad_(True)
    up = torch.randn(fa.shape, generator=g).cuda()
    out = f(a, boxes, ind)
    out.backward(up)
    lhs = (out.detach().double() * up.double()).sum()
    rhs = (a.grad.double() * a.detach().double()).sum()
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)


# ------------------------------------------------------------------- label decode
def test_label_decode_matches_reference_golden():
    from sln_amodal_amd import ops
    g = golden("label_decode")
    for ci in range(int(g["n_cases"])):
        label, planes, L = g["label_%d" % ci], g["planes_%d" % ci], int(g["L_%d" % ci])
        lab = dev(label.view(np.int64))
        n = ops.label_num_objects(lab)
        assert int(n[0]) == planes.shape[1]
        out = ops.label_decode(lab, L, planes.shape[1])
        assert np.array_equal(out[0].cpu().numpy(), planes), ci


def _synth_labels(orc, B, H, W, n_obj, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    labels = []
    for _ in range(B):
        masks = []
        for _i in range(n_obj):
            cy, cx = rng.uniform(0.125, 0.875) * H, rng.uniform(0.125, 0.875) * W
            ry, rx = rng.uniform(0.05, 0.25) * H, rng.uniform(0.05, 0.25) * W
            masks.append(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0)
        labels.append(orc.encode_labels(np.stack(masks)))
    return np.stack(labels)


@pytest.mark.parametrize("L", [1, 2, 4])
def test_label_decode_batched_vs_oracle(orc, L):
    from sln_amodal_amd import ops
    labels = _synth_labels(orc, 3, 61, 67, 7, seed=L)      # ragged: npix % 4 != 0
    labels[1][:] = 0                                       # empty image
    lab = dev(labels.view(np.int64))
    n = ops.label_num_objects(lab).cpu().numpy()
    want_n = [orc.label_num_objects(labels[b]) for b in range(3)]
    assert n.tolist() == want_n and want_n[1] == 0
    N = 8
    out = ops.label_decode(lab, L, N).cpu().numpy()
    for b in range(3):
        want = orc.label_decode(labels[b], L, N)
        assert np.array_equal(out[b], want)


def test_label_decode_full_size_checksum(orc):
    """BASELINE size: 16 x 1024 x 1024 labels, 8 objects.  Per-plane popcounts
    equal the popcounts computed from the label bits directly (integer exact)."""
    from sln_amodal_amd import ops
    labels = _synth_labels(orc, 2, 1024, 1024, 8, seed=3)
    labels = np.concatenate([labels] * 8)
    lab = dev(labels.view(np.int64))
    out = ops.label_decode(lab, 1, 8)
    sums = out.sum(dim=(3, 4), dtype=torch.int64).cpu().numpy()     # [B,1,N]
    for b in (0, 1, 15):
        for i in range(8):
            lo = (labels[b] >> np.uint64(i)) & np.uint64(1)
            hi = (labels[b] >> np.uint64(32 + i)) & np.uint64(1)
            assert sums[b, 0, i] == int((lo | hi).sum())


def test_fused_mask_targets_match_decode_then_crop(orc):
    from sln_amodal_amd import ops
    for case in ("a", "b"):
        g = golden("detection_target_%s" % case)
        L = int(g["L"])
        npos = int((g["class_ids"] > 0).sum())
        rois = g["rois"][:npos]
        planes = orc.label_decode(g["label"], L)
        ov = orc.bbox_overlaps(rois, g["gt_boxes"])
        assign = ov.argmax(axis=1).astype(np.int32)
        lab = dev(g["label"].view(np.int64))
        got = ops.mask_targets(lab, L, dev(rois), torch.zeros(npos, dtype=torch.int32).cuda(),
                               dev(assign), 32, 32)
        assert np.array_equal(got.cpu().numpy(), g["masks"][:npos])
        assert planes.shape[0] == L


# ---------------------------------------------------------------------- proposals
@pytest.mark.parametrize("dim", [128, 256])
def test_proposal_pipeline_matches_reference_golden(dim):
    from sln_amodal_amd import ops
    g = golden("proposal_layer_%d" % dim)
    probs, deltas, anchors = dev(g["probs"]), dev(g["deltas"]), dev(g["anchors"])
    A = anchors.shape[0]
    n = min(6000, A)
    order = torch.sort(probs[:, :, 1], dim=1, descending=True, stable=True)[1][:, :n].contiguous()
    dets = ops.proposal_decode(probs, deltas, anchors, order, (0.1, 0.1, 0.2, 0.2), dim, dim)
    keep, num = ops.nms_sorted(dets, 0.7, 1000)
    rois = ops.gather_rois(dets, keep, num, dim, dim)
    k = int(num[0])
    want = g["rois"][0]
    assert k == want.shape[0]
    assert np.allclose(rois[0, :k].cpu().numpy(), want, rtol=0, atol=1e-6)
    assert (rois[0, k:] == 0).all()


@pytest.mark.parametrize("case", ["random", "ties", "all_equal", "negative", "small", "nan_inf"])
def test_topk_order_equals_a_stable_descending_sort(case):
    """sln_topk_order_f32 == torch.sort(descending, stable)[:k], index for index: distinct scores,
    heavy ties at the cut (quantised scores), all-equal rows, negative values, A < 1024, NaN / inf."""
    from sln_amodal_amd import ops
    g = torch.Generator().manual_seed(len(case))
    B, A, k = 5, 70000, 6000
    if case == "random":
        s = torch.rand(B, A, generator=g)
    elif case == "ties":
        s = torch.round(torch.rand(B, A, generator=g) * 50) / 50        # ~1400 ties per value
    elif case == "all_equal":
        s = torch.full((B, A), 0.25)
    elif case == "negative":
        s = torch.randn(B, A, generator=g) - 3.0
    elif case == "small":
        B, A, k = 3, 700, 700
        s = torch.round(torch.randn(B, A, generator=g) * 4) / 4
    else:
        s = torch.randn(B, A, generator=g)
        s[:, ::97] = float("inf"); s[:, 5::1013] = float("-inf"); s[0, 3::5000] = float("nan")
    probs = torch.stack([1 - s, s], dim=2).cuda()                      # the strided foreground column
    got = ops.topk_order(probs[:, :, 1], k).cpu()
    want = torch.sort(s, dim=1, descending=True, stable=True)[1][:, :k]
    if case == "nan_inf":      # torch puts NaN first in a descending sort; so does the key order
        assert torch.equal(got, want)
    else:
        assert torch.equal(got, want)


@pytest.mark.parametrize("npos", [0, 3, 200, 256, 1000])
def test_topk_order_selects_positive_anchors_in_anchor_order(npos):
    """The RPN box loss's use (modal/loss.py): a 0/1 mask over 261 888 anchors, k = 256 -- the positives in
    anchor order first, then the lowest-index zeros; ties span every segment of the image."""
    from sln_amodal_amd import ops
    g = torch.Generator().manual_seed(npos)
    B, A, k = 4, 261888, 256
    s = torch.zeros(B, A)
    for b in range(B):
        s[b, torch.randperm(A, generator=g)[:npos]] = 1.0
    got = ops.topk_order(s.cuda(), k).cpu()
    want = torch.sort(s, dim=1, descending=True, stable=True)[1][:, :k]
    assert torch.equal(got, want)


def test_topk_order_is_reproducible_and_validates_its_workspace():
    from sln_amodal_amd import _lib, ops
    g = torch.Generator(device="cuda").manual_seed(5)
    s = torch.round(torch.rand(3, 50001, device="cuda", generator=g) * 100) / 100
    a, b = ops.topk_order(s, 3000), ops.topk_order(s, 3000)
    assert torch.equal(a, b)
    L = _lib.lib()
    assert L.sln_topk_workspace_bytes(3, 50001, 3000) > 3 * 3000 * 8
    assert L.sln_topk_order_f32(ops._ptr(s), 3, 50001, s.stride(0), s.stride(1), 3000, ops._ptr(a), None, 0,
                                None) == 2          # SLN_ERR_WORKSPACE


def test_topk_order_full_size_batch():
    from sln_amodal_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    s = torch.rand(16, 261888, device="cuda", generator=g)
    s[:, 1000:3000] = 0.999          # a plateau across the head of the ranking
    got = ops.topk_order(s, 6000)
    want = torch.sort(s, dim=1, descending=True, stable=True)[1][:, :6000]
    assert torch.equal(got, want)


# ------------------------------------------------------------------ FPN top-down merge
@pytest.mark.parametrize("shape", [(2, 256, 16, 16), (1, 64, 5, 7), (3, 8, 1, 1)])
def test_fpn_merge_equals_upsample_then_add(shape):
    """nn_ops.upsample2x_add (modal/modals.py:243-246): forward bit-identical to F.interpolate(nearest) + add,
    gradients: the lateral's is the incoming one, the coarse map's the 2x2 sums (1 ulp: summation order)."""
    import torch.nn.functional as F
    from sln_amodal_amd import nn_ops
    N, C, h, w = shape
    g = torch.Generator(device="cuda").manual_seed(h * 10 + w)
    lat = torch.randn(N, C, 2 * h, 2 * w, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    top = torch.randn(N, C, h, w, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    up = torch.randn(N, C, 2 * h, 2 * w, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    a, b = lat.clone().requires_grad_(True), top.clone().requires_grad_(True)
    out = nn_ops.upsample2x_add(a, b)
    assert type(out.grad_fn).__name__ == "_FpnMergeBackward"
    ra, rb = lat.clone().requires_grad_(True), top.clone().requires_grad_(True)
    ref = ra + F.interpolate(rb, scale_factor=2, mode="nearest")
    assert torch.equal(out, ref)
    out.backward(up)
    ref.backward(up)
    assert torch.equal(a.grad, ra.grad)
    assert torch.allclose(b.grad, rb.grad, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ stem max-pools
@pytest.mark.parametrize("shape", [(2, 64, 128, 128), (1, 64, 67, 45), (2, 8, 6, 9), (1, 64, 257, 257)])
@pytest.mark.parametrize("kind", ["same", "ceil"])
def test_max_pool_matches_torch_forward_and_argmax(shape, kind):
    """nn_ops.max_pool_same (modals.py:316-317) / max_pool_ceil (resnet_deeplab.py stem) on the HIP kernels:
    values equal to F.max_pool2d, and the gradient lands on the same element -- post-ReLU maps are full of
    exact ties (zeros), where torch keeps the first tap in kh-major order."""
    import torch.nn.functional as F
    from sln_amodal_amd import nn_ops
    N, C, H, W = shape
    g = torch.Generator(device="cuda").manual_seed(H + W)
    x = torch.relu(torch.randn(N, C, H, W, device="cuda", generator=g)).contiguous(memory_format=torch.channels_last)
    a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    if kind == "same":
        y = nn_ops.max_pool_same(a, 3, 2)
        pt, pb = nn_ops.same_pad(H, 3, 2)
        pl, pr = nn_ops.same_pad(W, 3, 2)
        ref = F.max_pool2d(F.pad(b, (pl, pr, pt, pb)), 3, 2)
    else:
        y = nn_ops.max_pool_ceil(a, 3, 2, 1)
        ref = F.max_pool2d(b, 3, 2, 1, ceil_mode=True)
    assert type(y.grad_fn).__name__ == "_MaxPoolFnBackward"
    assert y.shape == ref.shape and torch.equal(y, ref)
    up = torch.randn(y.shape, device="cuda", generator=g)
    y.backward(up)
    ref.backward(up)
    if kind == "ceil":
        assert torch.equal(a.grad, b.grad)
    else:
        # the reference pads with zeros: where a border window's maximum is 0 its first zero may be a padding
        # element (gradient dropped there); away from ties the two agree exactly
        same = a.grad == b.grad
        assert same.float().mean() > 0.999 or (x > 0).float().mean() < 0.5
        pos = x > 0
        assert torch.equal(a.grad[pos], b.grad[pos])


def test_pyramid_gather_backward_is_bit_reproducible_and_equals_the_scatter():
    """The write-once gather (no zero fill, no atomics) gives the same bits on every run and the atomic
    scatter's sums up to the order of the additions -- for one crop set and for two sets of the same maps
    in one launch."""
    from sln_amodal_amd.modal import modals
    gen = torch.Generator().manual_seed(11)
    B, C, R = 3, 96, 50
    sizes = [(64, 64), (32, 32), (16, 16), (8, 8)]
    ctr = torch.rand(B, R, 2, generator=gen) * 0.7 + 0.15
    half = torch.exp(torch.rand(B, R, 2, generator=gen) * 3.0 - 4.0)
    rois = torch.cat([ctr - half, ctr + half], dim=2).clamp(0, 1).cuda()
    base = [torch.randn(B, C, h, w, generator=gen).cuda().contiguous(memory_format=torch.channels_last)
            for h, w in sizes]
    up7 = torch.randn(B * R, C, 7, 7, generator=gen).cuda()
    up16 = torch.randn(B * R, C, 16, 16, generator=gen).cuda()

    def run(gather, pooled):
        saved = modals.GATHER_BACKWARD
        modals.GATHER_BACKWARD = gather
        try:
            maps = [m.clone().requires_grad_(True) for m in base]
            pool = modals.CropGradPool() if pooled else None
            a = modals.pyramid_roi_align([rois] + maps, 7, (256, 256), grad_pool=pool)
            b = modals.pyramid_roi_align([rois] + maps, 16, (256, 256), grad_pool=pool)
            ((a * up7).sum() + (b * up16).sum()).backward()
            return [m.grad.clone() for m in maps]
        finally:
            modals.GATHER_BACKWARD = saved

    for pooled in (False, True):
        g1, g2, s1 = run(True, pooled), run(True, pooled), run(False, pooled)
        for a, b, c in zip(g1, g2, s1):
            assert torch.equal(a, b)
            assert (a - c).abs().max().item() <= 1e-5 * max(c.abs().max().item(), 1e-6)


@pytest.mark.parametrize("shape", [(2, 182, 65, 65, ((33, 33), (49, 49))), (3, 21, 17, 23, ((9, 12), (13, 17))),
                                   (1, 200, 8, 8, ())])
def test_msc_softmax_tail_equals_the_unfused_ops(shape):
    """ops.msc_softmax_tail (csrc/glm_tail.hip) against the reference's sequence (msc_deeplab.py:42-48,
    model.py:537-541): bilinear resizes (align_corners False) -> maxima -> softmax -> argmax -> cat, run with
    torch ops on the same device.  Probabilities within 5e-6 absolute (a last-bit difference in a resized logit of
    magnitude ~10 moves exp() by 1e-6 relative; the path's bound is 1e-4); the label exact wherever the best two
    probabilities are not within rounding of each other; inputs are strided NHWC views like MultiScale's."""
    import torch.nn.functional as F
    from sln_amodal_amd import ops
    B, C, H, W, pyr = shape
    g = torch.Generator(device="cuda").manual_seed(C + H)

    def nhwc(b, c, h, w, pad):          # [b, c, h, w] view of an NHWC buffer whose pixel stride is c + pad
        buf = torch.randn(b, h, w, c + pad, device="cuda", generator=g) * 3.0
        return buf[..., :c].permute(0, 3, 1, 2)

    logits = nhwc(B, C, H, W, 2)
    pyramid = [nhwc(B, C, h, w, 0) for (h, w) in pyr]
    want = logits
    for l in pyramid:
        want = torch.max(want, F.interpolate(l, size=(H, W), mode="bilinear", align_corners=False))
    wp = F.softmax(want, dim=1)
    wl = torch.argmax(wp, dim=1)
    probs, label = ops.msc_softmax_tail(logits, pyramid)
    assert probs.shape == (B, C + 1, H, W) and probs.is_contiguous(memory_format=torch.channels_last)
    assert (probs[:, :C] - wp).abs().max().item() < 5e-6
    top2 = wp.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert clear.float().mean().item() > 0.99
    assert torch.equal(label[clear], wl[clear])
    # (true division like the reference's CPU path; ATen's CUDA division by a scalar multiplies by 1/255)
    assert torch.equal(probs[:, C].cpu(), label.cpu().float() / 255)
