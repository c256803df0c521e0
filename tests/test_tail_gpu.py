"""GPU parity tests for the inference tail (run with -m gpu on an MI355X): the HIP unmold and
run-length kernels, through the C ABI, against the CPU oracle and the reference-generated vectors.
Everything here is byte / integer work: bit-exact or failing."""
import numpy as np
import pytest
import torch

from tests._util import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def _col_major(mask_hw):
    """[h,w] -> the [w,h] array whose bytes are np.asfortranarray(mask)'s."""
    return np.ascontiguousarray(np.asarray(mask_hw, np.uint8).T)


def _encode(masks_hw, max_runs=None):
    from sln_amodal_amd import mask_rle
    dev = torch.from_numpy(np.stack([_col_major(m) for m in masks_hw])).cuda()
    return mask_rle.encode_counts(dev, max_runs)


def test_rle_matches_reference_vectors():
    from sln_amodal_amd import mask_rle
    g = golden("rle")
    for n in (str(x) for x in g["names"]):
        mask = g["mask/" + n]
        (got,) = _encode([mask])
        assert np.array_equal(got, g["counts/" + n]), n
        assert mask_rle.to_string(got) == bytes(g["string/" + n]), n
        d = mask_rle.encode(torch.from_numpy(_col_major(mask)[None]).cuda())[0]
        assert d["size"] == list(mask.shape) and d["counts"] == bytes(g["string/" + n])


@pytest.mark.parametrize("shape", [(1, 1), (1, 17), (17, 1), (64, 64), (97, 131), (128, 128), (333, 500)])
def test_rle_bit_exact_vs_oracle_batched(orc, shape):
    rng = np.random.RandomState(shape[0] * 1000 + shape[1])
    h, w = shape
    masks = [(rng.rand(h, w) < p).astype(np.uint8) for p in (0.0, 1.0, 0.5, 0.03, 0.97)]
    yy, xx = np.mgrid[0:h, 0:w]
    masks.append((((yy - h / 2) / (h / 3 + 1)) ** 2 + ((xx - w / 2) / (w / 4 + 1)) ** 2 <= 1).astype(np.uint8))
    got = _encode(masks)
    for m, c in zip(masks, got):
        assert np.array_equal(c, orc.rle_encode(m))


def test_rle_capacity_retry_and_worst_case(orc):
    from sln_amodal_amd import ops
    yy, xx = np.mgrid[0:96, 0:80]
    checker = ((yy + xx) & 1).astype(np.uint8)                # a + 1 runs in column-major order? no:
    stripes = (yy & 1).astype(np.uint8) * np.ones_like(checker)   # alternates every byte within a column
    dev = torch.from_numpy(np.stack([_col_major(stripes), _col_major(checker)])).cuda()
    counts, num = ops.rle_encode(dev, 16)                     # far too small: counts unspecified,
    want = [orc.rle_encode(stripes), orc.rle_encode(checker)]
    assert num.cpu().tolist() == [len(w) for w in want]       # but the true run count is reported
    got = _encode([stripes, checker], max_runs=16)            # the wrapper grows the capacity
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert len(want[0]) > 96 * 80 - 80


def test_rle_non_binary_bytes_and_unaligned_rows(orc):
    """rleEncode breaks a run wherever the byte changes (maskApi.c:38), not only 0 <-> 1; rows whose
    length is not a multiple of 16 start unaligned."""
    rng = np.random.RandomState(2)
    masks = [rng.randint(0, 3, (7, 9)).astype(np.uint8) for _ in range(5)]
    got = _encode(masks)
    for m, c in zip(masks, got):
        assert np.array_equal(c, orc.rle_encode(m))


def test_rle_full_size_round_trip_and_checksum(orc):
    """BASELINE-size masks (1024x1024, 100 detections): decode(encode(m)) == m, the counts sum to h*w,
    and a sample agrees with the oracle."""
    from sln_amodal_amd import mask_rle
    g = torch.Generator(device="cuda").manual_seed(0)
    N, H, W = 100, 1024, 1024
    yy = torch.arange(H, device="cuda").view(1, 1, H)
    xx = torch.arange(W, device="cuda").view(1, W, 1)
    cy = torch.rand(N, 1, 1, generator=g, device="cuda") * H
    cx = torch.rand(N, 1, 1, generator=g, device="cuda") * W
    ry = torch.rand(N, 1, 1, generator=g, device="cuda") * 300 + 5
    rx = torch.rand(N, 1, 1, generator=g, device="cuda") * 300 + 5
    masks = ((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1).to(torch.uint8)      # [N,W,H]
    counts = mask_rle.encode_counts(masks)
    host = masks.cpu().numpy()
    for i in range(N):
        assert int(counts[i].astype(np.int64).sum()) == H * W
    for i in (0, 17, 99):
        m_hw = host[i].T
        assert np.array_equal(counts[i], orc.rle_encode(m_hw))
        assert np.array_equal(mask_rle.decode_counts(counts[i], H, W), m_hw)


def _unmold(masks, boxes, H, W, class_ids=None, planes=None):
    from sln_amodal_amd import ops
    m = torch.from_numpy(np.ascontiguousarray(masks)).cuda()
    if planes is None:
        m = m[:, None]
    cid = None if class_ids is None else torch.from_numpy(np.asarray(class_ids, np.int32)).cuda()
    full = ops.unmold_masks(m, cid, torch.from_numpy(np.asarray(boxes, np.int32)).cuda(), H, W)
    assert full.shape == (len(boxes), W, H) and full.dtype == torch.uint8
    return full.permute(0, 2, 1).cpu().numpy()                 # [N,H,W]


def test_unmold_matches_reference_vectors():
    g = golden("unmold")
    H, W = (int(v) for v in g["image_shape"][:2])
    want = np.unpackbits(g["full"], axis=-1)[..., :W]
    got = _unmold(g["masks"], g["boxes"], H, W)
    for i in range(len(want)):
        assert np.array_equal(got[i], want[i]), i


@pytest.mark.parametrize("H,W", [(160, 192), (101, 67), (1024, 1024), (640, 1500)])
def test_unmold_bit_exact_vs_oracle(orc, H, W):
    rng = np.random.RandomState(H + W)
    n = 24
    boxes = []
    for i in range(n):
        y1, x1 = rng.randint(0, H - 1), rng.randint(0, W - 1)
        y2, x2 = rng.randint(y1 + 1, H + 1), rng.randint(x1 + 1, W + 1)
        if i % 6 == 0:                                        # thin / tiny boxes: the downscale branch
            y2, x2 = min(H, y1 + rng.randint(1, 9)), min(W, x1 + rng.randint(1, 40))
        boxes.append((y1, x1, y2, x2))
    boxes[0] = (0, 0, H, W)
    boxes[1] = (H - 1, W - 1, H, W)
    boxes[2] = (3, 5, 3, 50)                                  # empty: zero mask
    boxes[3] = (0, 0, H + 1, W)                               # outside the image: zero mask
    masks = rng.randn(n, 32, 32).astype(np.float32)
    gy, gx = np.mgrid[0:32, 0:32]
    for i in range(0, n, 2):                                  # blobs like a trained head's
        masks[i] = 1 / (1 + np.exp(-(6 - np.hypot(gy - rng.uniform(8, 24), gx - rng.uniform(8, 24)) / 1.7)))
    got = _unmold(masks, boxes, H, W)
    for i, b in enumerate(boxes):
        if i in (2, 3):
            assert got[i].sum() == 0
            continue
        assert np.array_equal(got[i], orc.unmold_mask(masks[i], b, (H, W))), (i, b)


def test_unmold_class_planes_and_other_mask_sizes(orc):
    rng = np.random.RandomState(8)
    planes = rng.randn(6, 3, 28, 20).astype(np.float32)       # [N,C,mh,mw], non-square head
    cid = np.array([0, 1, 2, 1, 0, 2], np.int32)
    boxes = [(2, 3, 90, 70), (10, 10, 24, 25), (0, 0, 100, 120), (50, 60, 51, 119), (7, 1, 99, 9),
             (30, 30, 60, 50)]
    got = _unmold(planes, boxes, 100, 120, class_ids=cid, planes=True)
    for i, b in enumerate(boxes):
        assert np.array_equal(got[i], orc.unmold_mask(planes[i, cid[i]], b, (100, 120))), i


def test_unmold_then_rle_equals_reference_pipeline(orc):
    """The evaluate hand-off: device masks -> COCO RLE strings == unmold_mask + rleEncode + rleToString
    of the oracle (pinned to the reference's maskApi.c)."""
    from sln_amodal_amd import mask_rle, ops
    g = golden("unmold")
    H, W = (int(v) for v in g["image_shape"][:2])
    full = ops.unmold_masks(torch.from_numpy(g["masks"]).cuda()[:, None], None,
                            torch.from_numpy(g["boxes"]).cuda(), H, W)
    rles = mask_rle.encode(full)
    for i, r in enumerate(rles):
        want = orc.rle_to_string(orc.rle_encode(orc.unmold_mask(g["masks"][i], g["boxes"][i], (H, W))))
        assert r == {"size": [H, W], "counts": want}


def _random_batch(B, S, seed, shapes):
    """Detections like predict(mode='inference') returns them ([B,S,6]: normalised-to-window boxes in pixels, class,
    score; zero rows behind each image's count), mask logits [B,S,2,28,28], windows."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    dim = 256
    counts = torch.tensor([S, 0, 7, 1, S // 2, 3][:B], dtype=torch.int32, device="cuda")
    tl = torch.rand(B, S, 2, device="cuda", generator=g) * (dim - 40)
    wh = torch.rand(B, S, 2, device="cuda", generator=g) * 120 + 3
    det = torch.cat([tl, (tl + wh).clamp(max=dim), torch.ones(B, S, 1, device="cuda"),
                     torch.rand(B, S, 1, device="cuda", generator=g)], 2)
    det[0, 3, 2:4] = det[0, 3, 0:2]            # a zero-area box inside the count: dropped (model.py:786-793)
    det[2, 5, 4] = 0                           # a class-0 row inside the count: the image ends there (model.py:762-764)
    det = det * (torch.arange(S, device="cuda").view(1, S, 1) < counts.view(B, 1, 1))
    msk = torch.randn(B, S, 2, 28, 28, device="cuda", generator=g)
    windows = [(0, 0, dim, dim)] * B
    return det.contiguous(), msk, counts, windows, [shapes[b % len(shapes)] for b in range(B)]


@pytest.mark.parametrize("shapes", [[(200, 320, 3)], [(200, 320, 3), (97, 131, 3)]])
def test_batched_tail_equals_the_per_image_hand_off(shapes):
    """tail.unmold_batch (one unmold + one run-length launch per image size for ALL detections of a batch, one C call
    for the strings) against the per-image path it replaces on the evaluation loop (MaskRCNN.unmold_detections_device +
    mask_rle.encode, themselves bit-identical to model.py:747-806 / utils.py:447-465 / maskApi.c:33-49, 204-216 --
    test_unmold_then_rle_equals_reference_pipeline): the same boxes, class ids, scores, masks and RLE strings."""
    from sln_amodal_amd import mask_rle, tail
    from sln_amodal_amd.model import MaskRCNN
    B, S = 6, 20
    det, msk, counts, windows, shp = _random_batch(B, S, 3, shapes)
    got = tail.unmold_batch(det, msk, counts, shp, windows, rle=True, keep_masks=True)
    seen = 0
    for b in range(B):
        n = int(counts[b])
        want = MaskRCNN.unmold_detections_device(None, det[b, :n], msk[b, :n], shp[b], windows[b], keep_device=True) if n else None
        if want is None or want["rois"].shape[0] == 0:
            assert b not in got
            continue
        seen += 1
        r = got[b]
        assert np.array_equal(r["rois"], want["rois"]) and np.array_equal(r["class_ids"], want["class_ids"])
        assert np.array_equal(r["scores"], want["scores"])
        assert torch.equal(r["masks_device"], want["masks_device"])
        assert r["rles"] == mask_rle.encode(want["masks_device"])
    assert seen >= 4
    assert got[2]["rois"].shape[0] == 5 and got[0]["rois"].shape[0] == S - 1       # the two filters above


def test_inference_tail_worker_delivers_every_batch_in_order():
    """tail.InferenceTail: submit() returns at once (an event record); a worker thread runs the batched hand-off on
    its own stream while the submitting thread goes on; results() returns every image of every batch under its key,
    equal to the synchronous call."""
    from sln_amodal_amd import tail
    B, S = 4, 12
    batches = [_random_batch(B, S, 10 + k, [(160, 192, 3)]) for k in range(5)]
    want = {}
    for k, (det, msk, counts, windows, shp) in enumerate(batches):
        for b, r in tail.unmold_batch(det, msk, counts, shp, windows).items():
            want[k * B + b] = r
    t = tail.InferenceTail("cuda", depth=2)
    try:
        for k, (det, msk, counts, windows, shp) in enumerate(batches):
            y = det * 2.0                                  # (work enqueued on the main stream behind the event)
            t.submit(det, msk, counts, shp, windows)
            del y
        got = t.results()
    finally:
        t.close()
    assert sorted(got) == sorted(want)
    for key in want:
        assert np.array_equal(got[key]["rois"], want[key]["rois"]) and got[key]["rles"] == want[key]["rles"]
        assert np.array_equal(got[key]["scores"], want[key]["scores"])
    assert t.results() == {}


def test_rle_to_strings_equals_the_per_mask_codec(orc):
    """sln_rle_to_strings (ABI 12) == sln_rle_to_string per row == the oracle's rleToString (maskApi.c:204-216)."""
    from sln_amodal_amd import mask_rle, tail
    rng = np.random.RandomState(4)
    rows = [rng.randint(0, 5000, size=n).astype(np.uint32) for n in (1, 2, 3, 40, 0, 977)]
    width = max(r.size for r in rows)
    counts = np.zeros((len(rows), width), np.uint32)
    for i, r in enumerate(rows):
        counts[i, :r.size] = r
    num = np.array([r.size for r in rows], np.int32)
    got = tail.to_strings(counts, num)
    assert got == [mask_rle.to_string(r) for r in rows]
    assert got[3] == orc.rle_to_string(rows[3])
