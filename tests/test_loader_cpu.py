"""Input pipeline (sln_amodal_amd/loader.py), host side: the per-epoch shuffle and its rank shards, and the worker
-> shared-memory -> feeder plumbing (no GPU: the feeder hands over host tensors).  Reference behaviour:
torch.utils.data.DataLoader(train_set, shuffle=True, num_workers=4) over Dataset.__getitem__ (model.py:76-116,
340-342)."""
import os

import numpy as np
import pytest

from sln_amodal_amd import loader


def test_shuffle_is_a_permutation_disjoint_across_ranks_and_different_per_epoch():
    n, B, world = 103, 4, 3
    samplers = [loader.EpochSampler(n, B, r, world, seed=5) for r in range(world)]
    assert len({s.steps for s in samplers}) == 1 and samplers[0].steps == n // (B * world)
    seen_orders = []
    for epoch in range(3):
        per_rank = [s.epoch(epoch) for s in samplers]
        # every rank runs the same number of steps (a rank with one step more would wait in a collective forever)
        assert len({len(p) for p in per_rank}) == 1
        ids = [np.concatenate([ids for ids, _ in p]) for p in per_rank]
        flat = np.concatenate(ids)
        assert len(set(flat.tolist())) == flat.size                   # disjoint across ranks, no repeats inside
        assert set(flat.tolist()) <= set(range(n))
        assert flat.size == samplers[0].steps * B * world             # the tail that fills no global batch is dropped
        order = samplers[0].order(epoch)
        assert sorted(order.tolist()) == sorted(set(order.tolist()))   # a permutation (prefix)
        # step s of the job = the s-th slice of world * B entries of the one shared permutation
        for s in range(samplers[0].steps):
            got = np.concatenate([per_rank[r][s][0] for r in range(world)])
            assert np.array_equal(got, order[s * B * world:(s + 1) * B * world])
        seen_orders.append(order)
        flips = np.concatenate([f for p in per_rank for _, f in p])
        assert set(flips.tolist()) <= {0, 1} and 0 < flips.mean() < 1
    assert not np.array_equal(seen_orders[0], seen_orders[1]) and not np.array_equal(seen_orders[1], seen_orders[2])
    # deterministic: the same seed and epoch give the same order on every rank and in every run
    assert np.array_equal(loader.EpochSampler(n, B, 1, world, seed=5).order(1), seen_orders[1])
    assert not np.array_equal(loader.EpochSampler(n, B, 1, world, seed=6).order(1), seen_orders[1])
    # over many epochs every file is visited (the dropped tail is a different one each epoch)
    hit = set()
    for epoch in range(12):
        hit |= set(samplers[0].order(epoch).tolist())
    assert hit == set(range(n))


def test_unshuffled_and_tiny_datasets():
    s = loader.EpochSampler(10, 4, 0, 1, shuffle=False)
    assert [ids.tolist() for ids, _ in s.epoch(0)] == [[0, 1, 2, 3], [4, 5, 6, 7]]
    tiny = loader.EpochSampler(3, 4, 1, 2, seed=1)           # fewer files than one global batch: repeated
    plan = tiny.epoch(0)
    assert len(plan) == 1 and plan[0][0].shape == (4,) and set(plan[0][0].tolist()) <= {0, 1, 2}
    with pytest.raises(ValueError):
        loader.EpochSampler(0, 4)


def _write_scene(root, k, h, w, rng, broken=False):
    from PIL import Image
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(root, "s%03d.jpg" % k), quality=92)
    lab = (rng.randint(0, 16, (h, w)).astype(np.uint64)) | (rng.randint(0, 16, (h, w)).astype(np.uint64) << np.uint64(32))
    if broken:
        open(os.path.join(root, "s%03d.npz" % k), "wb").write(b"not a zip archive")
    else:
        np.savez_compressed(os.path.join(root, "s%03d.npz" % k), layer=lab)
    return {"path": os.path.join(root, "s%03d.jpg" % k), "label": os.path.join(root, "s%03d.npz" % k)}


def test_worker_pipeline_delivers_the_sampled_batches(tmp_path):
    """Two worker processes, shared-memory slots, the feeder thread: every delivered batch is the sampler's batch
    -- the images a direct load_item gives (flipped as drawn), the raw labels and their sizes."""
    rng = np.random.RandomState(3)
    dim, B = 48, 4
    sizes = [(40, 56), (48, 48), (64, 30), (33, 41), (48, 70), (52, 52), (20, 90), (61, 47), (48, 48), (31, 33)]
    infos = [_write_scene(str(tmp_path), k, h, w, rng) for k, (h, w) in enumerate(sizes)]
    sampler = loader.EpochSampler(len(infos), B, 0, 1, seed=2)
    pipe = loader.PrefetchLoader(infos, dim, B, sampler, workers=2, depth=2, cap_pixels=8192, device=None)
    try:
        want = sampler.epoch(0) + sampler.epoch(1)
        it = iter(pipe)
        for step in range(len(want)):
            item = next(it)
            ids, flips = want[step]
            assert item["ids"] == ids.tolist() and item["flips"] == flips.tolist()
            for r, (iid, flip) in enumerate(zip(ids, flips)):
                u8, layer = loader.load_item(infos[iid], dim)
                assert tuple(item["src_hw_host"][r]) == layer.shape
                assert np.array_equal(item["u8"][r].numpy(), u8[:, ::-1] if flip else u8)
                n = layer.size
                assert np.array_equal(item["labels"][r, :n].numpy().view(np.uint64), layer.reshape(-1))
            assert item["host_zoomed"] == [0] * B
        rep = pipe.report()
        assert rep["batches"] == len(want) and rep["workers"] == 2 and rep["labels_zoomed_on_host"] == 0
    finally:
        pipe.close()
    names = [s.name for s in pipe._shms]
    assert names == []                              # shared memory released


def test_labels_larger_than_a_slot_are_zoomed_by_the_worker(tmp_path):
    from sln_amodal_amd import utils
    rng = np.random.RandomState(4)
    dim, B = 32, 2
    infos = [_write_scene(str(tmp_path), k, 80, 70, rng) for k in range(2)]
    pipe = loader.PrefetchLoader(infos, dim, B, loader.EpochSampler(2, B, 0, 1, shuffle=False), workers=1, depth=1,
                                 cap_pixels=dim * dim, device=None)
    try:
        item = next(iter(pipe))
        assert item["host_zoomed"] == [1, 1] and item["src_hw_host"].tolist() == [[dim, dim]] * 2
        for r in range(B):
            _, layer = loader.load_item(infos[item["ids"][r]], dim)
            want = utils.resize_layer(layer, (dim / 80, dim / 70))      # never flipped by the worker
            assert np.array_equal(item["labels"][r, :dim * dim].numpy().view(np.uint64).reshape(dim, dim), want)
    finally:
        pipe.close()
    # the worker's index map is utils.zoom_nearest_index (kept torch-free in loader.py)
    for n_in, n_out in ((80, 32), (1, 5), (7, 1), (1023, 1024), (641, 1024), (5, 0)):
        assert np.array_equal(loader._zoom_index(n_in, n_out), utils.zoom_nearest_index(n_in, n_out))


def test_a_failing_worker_raises_in_the_training_thread(tmp_path):
    rng = np.random.RandomState(5)
    infos = [_write_scene(str(tmp_path), k, 20, 20, rng, broken=(k == 1)) for k in range(2)]
    pipe = loader.PrefetchLoader(infos, 16, 2, loader.EpochSampler(2, 2, 0, 1, shuffle=False), workers=1, depth=1,
                                 device=None)
    try:
        with pytest.raises(RuntimeError) as e:
            next(iter(pipe))
        assert "s001.npz" in repr(e.value.__cause__)
    finally:
        pipe.close()
