"""Input pipeline on the device: the label zoom / object count kernels against the host restatement of the
reference's scipy path, and the worker pipeline's batches against the same images loaded on the training thread
(`AmodalDataset._load_real`, which tests/test_e2e_gpu.py pins to the reference's load_image_gt fixtures)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _label(rng, h, w, n_obj):
    lo = np.uint64(1) << rng.randint(0, max(n_obj, 1), (h, w)).astype(np.uint64)
    lo[rng.rand(h, w) < 0.3] = 0
    hi = rng.randint(0, 1 << max(n_obj, 1), (h, w)).astype(np.uint64)
    return (lo | (hi << np.uint64(32))).astype(np.uint64)


def test_label_zoom_and_ragged_object_count_match_the_host_path():
    """sln_label_zoom_u64 == utils.resize_layer (scipy.ndimage.zoom order 0 of utils.py:358-362, the constant-fill
    quirk of a last sample included) followed by the flip of Functions.py:713-716; sln_label_num_objects_ragged_u64
    == the object count of the ORIGINAL label (Functions.py:1074-1079)."""
    from sln_amodal_amd import ops, utils
    rng = np.random.RandomState(0)
    for dim in (64, 128, 513):
        sizes = [(int(rng.randint(1, 700)), int(rng.randint(1, 700))) for _ in range(5)] + [(dim, dim), (1, 1), (641, 3)]
        B = len(sizes)
        labs = [_label(rng, h, w, int(rng.randint(0, 9))) for h, w in sizes]
        flips = rng.randint(0, 2, B)
        stride = (max(l.size for l in labs) + 7) // 8 * 8
        flat = np.zeros((B, stride), np.int64)
        flat[:] = -1                                   # garbage behind every label: must not be read
        ys, xs = np.empty((B, dim), np.int32), np.empty((B, dim), np.int32)
        fills = 0
        for b, l in enumerate(labs):
            flat[b, :l.size] = l.reshape(-1).view(np.int64)
            ys[b] = utils.zoom_nearest_index(l.shape[0], dim)
            x = utils.zoom_nearest_index(l.shape[1], dim)
            xs[b] = x[::-1] if flips[b] else x
            fills += int((ys[b] < 0).sum() + (x < 0).sum())
        hw = torch.tensor(sizes, dtype=torch.int32, device="cuda")
        src = torch.from_numpy(flat).cuda()
        out = ops.label_zoom(src, hw, torch.from_numpy(ys).cuda(), torch.from_numpy(xs).cuda()).cpu().numpy()
        cnt = ops.label_num_objects_ragged(src, hw).cpu().numpy()
        for b, l in enumerate(labs):
            want = utils.resize_layer(l, (dim / l.shape[0], dim / l.shape[1]))
            want = want[:, ::-1] if flips[b] else want
            assert np.array_equal(out[b].view(np.uint64), want), (dim, sizes[b], int(flips[b]))
            tops = set(int(v).bit_length() - 1 for v in np.unique(l & np.uint64(0xFFFFFFFF)) if v)
            n = 0
            while n in tops:
                n += 1
            assert cnt[b] == n, (sizes[b], cnt[b], n)


def _write_dataset(root, sizes, rng):
    from PIL import Image
    for k, (h, w) in enumerate(sizes):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(yy * 3 + xx) % 256, (xx * 2 + k * 17) % 256, (yy + xx * k) % 256], -1).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, "im%03d.jpg" % k), quality=90)
        masks = []
        for i in range(int(rng.randint(2, 6))):
            cy, cx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w
            ry, rx = rng.uniform(0.1, 0.35) * h, rng.uniform(0.1, 0.35) * w
            masks.append(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1)
        lab = np.zeros((h, w), np.uint64)
        covered = np.zeros((h, w), bool)
        for i, m in enumerate(masks):                  # painter's order: object 0 on top
            lab[m & ~covered] |= np.uint64(1) << np.uint64(i)
            lab[m & covered] |= np.uint64(1) << np.uint64(32 + i)
            covered |= m
        np.savez_compressed(os.path.join(root, "im%03d.npz" % k), layer=lab)


@pytest.mark.parametrize("assemble_stream", ["1", "0"])
def test_worker_pipeline_batches_equal_the_training_thread_loader(tmp_path, assemble_stream, monkeypatch):
    """(assemble_stream "1": round 6 -- the device half of batch k + 1 runs on a side stream while batch k is being
    consumed, a one-deep pipeline inside the iterator; "0": on the consumer's stream.)
    Worker processes -> shared memory -> pinned staging -> copy stream -> device half: every batch equals, bit for
    bit, the batch `_load_real` builds from the same images with the same flips and jitter draws (images, labels,
    boxes, class ids, RPN targets); one label is larger than a loader slot (zoomed by the worker instead of the
    device: the same planes).  The order is the sampler's, the shuffle differs between epochs."""
    from sln_amodal_amd import amodal_train, loader
    from sln_amodal_amd.model import MaskRCNN
    from tests._parity import loader_config
    monkeypatch.setenv("SLN_LOADER_ASSEMBLE_STREAM", assemble_stream)
    rng = np.random.RandomState(1)
    sizes = [(96, 160), (200, 150), (128, 128), (77, 301), (640, 480), (131, 97), (128, 64), (150, 150)]
    _write_dataset(str(tmp_path), sizes, rng)
    cfg = loader_config(128)
    cfg.ARCHITECTURE = "resnet50"
    cfg.BATCH_SIZE = 4
    m = MaskRCNN(cfg, str(tmp_path)).cuda()
    ds = amodal_train.AmodalDataset(cfg, m, root=str(tmp_path), device="cuda", workers=3, prefetch=2, seed_order=9)
    ref = amodal_train.AmodalDataset(cfg, m, root=str(tmp_path), device="cuda", workers=0, seed_order=9)
    # slots of 200 x 150 pixels: the 640 x 480 label does not fit and takes the worker-zoom path
    sampler = loader.EpochSampler(len(sizes), 4, 0, 1, seed=9)
    ds._pipe = loader.PrefetchLoader(ds.image_info, 128, 4, sampler, workers=3, depth=2, cap_pixels=200 * 160,
                                     device="cuda").start()
    try:
        plan = sampler.epoch(0) + sampler.epoch(1)
        assert not np.array_equal(sampler.order(0), sampler.order(1))
        state = ds._jitter_rng.get_state()
        it = iter(ds)
        got = [next(it) for _ in plan]
        jr = np.random.RandomState(0)
        jr.set_state(state)
        for step, (ids, flips) in enumerate(plan):
            jitter = jr.rand(4, ds.max_objects, 4)
            draws = [{"flip": int(f), "jitter": jitter[k]} for k, f in enumerate(flips)]
            want = ref._load_real([int(i) for i in ids], draws=draws)
            b = got[step]
            assert b["flipped"] == [int(f) for f in flips]
            for key in ("images", "gt_layer", "gt_class_ids", "gt_boxes", "rpn_bbox"):
                assert torch.equal(b[key], want[key]), (step, key)
            # the NEGATIVE anchors of the RPN batch are a random draw (np.random.choice in the reference, a device
            # draw here): the positives -- fewer than the 128 that would be sub-sampled -- and their count agree,
            # the negatives fill the batch to 256 on both sides
            assert torch.equal(b["rpn_match"] == 1, want["rpn_match"] == 1), step
            assert int((b["rpn_match"] == 1).sum(1).max()) < 128
            assert torch.equal((b["rpn_match"] != 0).sum(1), (want["rpn_match"] != 0).sum(1))
        rep = ds.loader_report()
        # (the side-stream pipeline has assembled one batch ahead of the consumer)
        assert rep["batches"] == len(plan) + (1 if assemble_stream == "1" else 0)
        assert rep["labels_zoomed_on_host"] >= 1 and rep["images_over_object_slots"] == 0
        assert ds.queue_depth() is not None
    finally:
        ds.close()


def test_more_objects_than_slots_is_counted_not_silently_dropped(tmp_path):
    from sln_amodal_amd import amodal_train
    from sln_amodal_amd.model import MaskRCNN
    from tests._parity import loader_config
    rng = np.random.RandomState(2)
    _write_dataset(str(tmp_path), [(128, 128)] * 2, rng)
    cfg = loader_config(128)
    cfg.ARCHITECTURE = "resnet50"
    cfg.BATCH_SIZE = 2
    m = MaskRCNN(cfg, str(tmp_path)).cuda()
    ds = amodal_train.AmodalDataset(cfg, m, root=str(tmp_path), device="cuda", workers=1, prefetch=1, max_objects=1)
    try:
        b = next(iter(ds))
        assert b["gt_class_ids"].shape == (2, 1)
        # (2 per batch of two images; the iterator has assembled one batch ahead of the consumer)
        assert ds.loader_report()["images_over_object_slots"] in (2, 4)
    finally:
        ds.close()


def test_cli_train_on_a_generated_dataset_runs_its_three_stages(tmp_path):
    """`python amodal_train.py train --dataset DIR` end to end (reference CLI, amodal_train.py:507-663): the loader's
    workers start BEFORE the process touches the GPU, the dataset is bound to the device afterwards, the three training
    stages (heads x 2, 4+ x 3, all x 1 epochs) run two steps each from worker-fed batches, a checkpoint per epoch is
    written, and the process exits cleanly (no leaked shared memory, no hung worker)."""
    import glob
    import subprocess
    import sys
    from sln_amodal_amd import loader
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = str(tmp_path / "data")
    loader.write_synthetic_dataset(data, 8, 128, n_obj=4, seed=5, procs=2)
    before = set(glob.glob("/dev/shm/psm_*"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "amodal_train.py"), "train", "--dataset", data, "--model", "none",
                        "--logs", str(tmp_path / "logs"), "--arch", "resnet50", "--image-dim", "128", "--batch", "2",
                        "--steps-per-epoch", "2", "--workers", "2", "--mask-positive-slots"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    ckpts = glob.glob(str(tmp_path / "logs" / "**" / "*.pth"), recursive=True)
    assert len(ckpts) == 6, ckpts                       # one per epoch of the reference's schedule
    assert r.stdout.count("Complete - mean loss") == 6 and "nan" not in r.stdout.lower()
    assert set(glob.glob("/dev/shm/psm_*")) <= before     # the loader's segments are gone
