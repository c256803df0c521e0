"""Data-parallel train step of the real model with two processes.  RCCL needs one GPU per rank, which
the test boxes do not have, so both ranks share cuda:0 and exchange through gloo (SLN_DIST_BACKEND):
everything except the transport is the production path -- replica broadcast (and the cache
invalidation it implies), gradient hooks, ordered bucket all-reduce, clip, SGD -- and the replicas
must stay bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SLN_DIST_BACKEND="gloo")
    from sln_amodal_amd import parallel, synthetic
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.model import MaskRCNN
    r, local, w = parallel.init_distributed()
    assert (r, w, local) == (rank, world, 0)
    torch.cuda.set_device(local)

    class C(Config):
        NAME = "dp"
        IMAGE_MAX_DIM = 256
        ARCHITECTURE = "resnet50"

    cfg = C()
    torch.manual_seed(100 + rank)                      # replicas start DIFFERENT: the broadcast must fix it
    m = MaskRCNN(cfg, "/tmp/sln_dp_logs").apply_amodal_heads().cuda()
    m.set_trainable(".*", exclusive_off=False)
    for p in m.GLM_modual.parameters():
        p.requires_grad = False
    batch = synthetic.make_batch(cfg, 2, 256, 256, seed=50 + rank, anchors_f64=m.anchors_f64)   # disjoint shards
    synthetic.calibrate_batchnorm(m, batch["images"])  # per-rank statistics (and cached BN affines) ...
    synthetic.calibrate_glm(m, batch["images"])
    synthetic.warm_start_rpn(m, [batch], iters=10)
    parallel.broadcast_parameters(m)                   # ... replaced by rank 0's here
    opt = m.make_optimizer(0.01)
    # ---- gradients through the bucket-slot sink == the mean of the ranks' plain gradients, also for weights used
    # several times per pass (rpn.conv_shared / conv_class / conv_bbox run on five pyramid levels: five weight
    # gradients that autograd sums -- only the FIRST may be written into the slot) ----
    gen = torch.Generator(device="cuda").manual_seed(5)
    pr = {"pos": torch.rand(2, 1000, device="cuda", generator=gen), "neg": torch.rand(2, 1000, device="cuda", generator=gen)}
    inp = [batch["images"], None, batch["gt_class_ids"], batch["gt_boxes"], batch["gt_layer"]]
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]

    def backward_once():
        m.zero_grad(set_to_none=True)
        o = m.predict(inp, mode="training", priorities=pr)
        loss, _ = m.compute_losses(o, batch["rpn_match"], batch["rpn_bbox"])
        loss.backward()

    backward_once()                                    # no reducer, no sink: this rank's plain gradients
    plain = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in named}
    for n in plain:
        dist.all_reduce(plain[n])
        plain[n] /= world
    red = parallel.GradientAllReducer([p for _, p in named]).attach()
    backward_once()
    red.finish()
    worst, worst_rpn = ("", 0.0), ("", 0.0)
    for n, p in named:
        err = float((p.grad - plain[n]).norm() / plain[n].norm().clamp_min(1e-20))
        if err > worst[1]:
            worst = (n, err)
        if n.startswith("rpn.") and err > worst_rpn[1]:
            worst_rpn = (n, err)
        # Two passes over the same batch are not bit-equal: the RoIAlign scatter's fp32 atomics land in another order,
        # and what lies under the crops' ReLUs amplifies that (recorded: 6e-5 ... 2.6e-4 on mask.conv1.weight over
        # four runs) -- the ReLU-switch tolerance of the e2e tests.  The weights this check is about are the RPN's
        # (five gradients per pass into one slot): they see the FPN maps only, and a lost contribution is an O(1)
        # error.
        assert err <= (1e-5 if n.startswith("rpn.") else 5e-3), (rank, n, err)     # (RPN weights, recorded: <= 1.5e-7)
    print("rank", rank, "sink vs plain mean: worst relative error %.2e (%s); RPN weights %.2e (%s)" %
          (worst[1], worst[0], worst_rpn[1], worst_rpn[0]), flush=True)
    for n in ("rpn.conv_shared.weight", "rpn.conv_class.weight", "rpn.conv_bbox.weight"):
        assert float(dict(named)[n].grad.norm()) > 0
    m.zero_grad(set_to_none=True)
    for _ in range(3):
        loss, _parts = m.train_step(batch, opt, lambda params: red.finish())
    assert bool(torch.isfinite(loss))
    # flat buckets: the convolution weight gradients (nearly all of the bytes) were written by the reduce pass
    # straight into their bucket slots -- no pack before and no copy back after the collective -- and every
    # gradient the optimiser read was a view of its slot
    st = red.stats
    total = st["in_place_bytes"] + st["copied_bytes"]
    print("rank", rank, "gradient bytes in place %d, copied %d in %d tensors" %
          (st["in_place_bytes"], st["copied_bytes"], st["copied_tensors"]), flush=True)
    assert total == 4 * sum(p.numel() * 4 for p in red.params)      # (the check above + three steps)
    # (what is copied: the biases, and the three weights whose gradient is produced in another layout and
    # re-laid by autograd -- the classifier's whole-window 7x7 "FC" conv, 51 MB of this model's 180 MB, the
    # 2x2 deconv and the 3-channel stem)
    assert st["in_place_bytes"] >= 0.65 * total and st["copied_tensors"] <= 4 * 95, st
    for p in red.params:
        assert p.grad is None or p.grad.data_ptr() == red.slot_view(p).data_ptr()
    flat = torch.cat([p.detach().reshape(-1).double() for p in m.parameters()] +
                     [b.detach().reshape(-1).double() for b in m.buffers()])
    # the replicas must also COMPUTE the same thing (stale per-rank caches of weight parts or BN affines
    # would leave the parameters equal -- the gradients are averaged -- but not the forward pass)
    xs = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(9)).cuda()
    with torch.no_grad():
        p2 = m.fpn(xs.contiguous(memory_format=torch.channels_last))[0].double()
    digest = torch.stack([flat.sum(), flat.abs().sum(), (flat * flat).sum(), p2.sum(), p2.abs().sum()]).cpu()
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    print("rank", rank, "digests", [g.tolist() for g in gathered], flush=True)
    assert torch.equal(gathered[0][:3], gathered[1][:3]), (rank, gathered)            # state: bit-identical
    # forward: equal up to the stem's aten/MIOpen convolution, whose algorithm choice is per process (the
    # operand scales are the same on every rank: ScaleBook.update takes the maximum over the ranks)
    assert torch.allclose(gathered[0][3:], gathered[1][3:], rtol=1e-4, atol=0), (rank, gathered)   # stale caches: O(1) off
    dist.destroy_process_group()
    out.put(rank)


def _release_parent_memory():
    """The ranks are fresh processes on the SAME GPU as this pytest process, whose caching allocator may be holding
    most of the HBM by now (the full-size tests ran before): give it back first -- a rank that dies of memory at
    start-up leaves its peer waiting in gloo until the join times out."""
    import gc
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def _run_two_ranks(limit):
    import time
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    deadline = time.time() + limit
    while time.time() < deadline and any(p.is_alive() for p in procs):
        if any(p.exitcode not in (None, 0) for p in procs):      # a rank died: do not wait for its peer's timeout
            break
        time.sleep(0.5)
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(10)
            if p.is_alive():
                p.kill()
                p.join(10)
    codes = [p.exitcode for p in procs]
    done = []
    if codes == [0, 0]:
        done = sorted(q.get(timeout=5) for _ in range(2))
    return codes, done


def test_two_rank_train_steps_keep_replicas_identical():
    """(Two processes SHARING one GPU through gloo is a stand-in topology: RCCL wants a GPU per rank.  On this pool
    the pair has -- rarely, only inside a full-suite run -- failed to come up; a rank that dies or a pair that does
    not finish in 120 s is started once more, and only the second failure counts.  Every assertion about the
    replicas is made inside the ranks: exit code 0 means they held.)"""
    _release_parent_memory()
    codes, done = _run_two_ranks(120)
    if codes != [0, 0]:
        print("two-rank run failed with exit codes", codes, "-- second attempt", flush=True)
        _release_parent_memory()
        codes, done = _run_two_ranks(300)
    assert codes == [0, 0]
    assert done == [0, 1]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun: the parent (which never touches the GPU) starts two
    ranks itself and relays rank 0's line -- n_gpus must be what was asked for.  One-GPU box: both ranks
    share cuda:0 and exchange through gloo (SLN_DIST_BACKEND), small shapes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _release_parent_memory()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["SLN_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--settle", "1", "--batch", "2", "--dim", "256", "--arch", "resnet50",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["scaling"] == "weak"
    # the self-diagnosis of a scaling run: every rank reports its own step times, the all-reduce
    # wait backward did not hide, and the host cores it was pinned to before its first GPU call
    gx = out["gradient_exchange"]
    assert gx["world_size_seen"] == 2 and gx["backend"] == "gloo"
    assert "loader_queue_depth" in gx and gx["step_ms_max_over_ranks"] >= gx["step_ms_min_over_ranks"] > 0
    assert [g["rank"] for g in gx["per_rank"]] == [0, 1]
    for g in gx["per_rank"]:
        assert g["step_ms_min"] > 0 and g["finishes"] >= 2 and g["buckets"] == gx["buckets"]
        assert "exposed_wait_ms_mean" in g and "host_wait_ms_mean" in g
        assert g["cpu_affinity"].get("cores", 0) >= 1
    a0, a1 = (g["cpu_affinity"] for g in gx["per_rank"])
    assert a0["last"] < a1["first"]              # disjoint shares of the host cores
    # another problem than the headline's: no replayed counter file next to this run's numbers
    assert out["roofline"]["traffic"] is None


@pytest.mark.timeout(1500)
def test_bench_launches_eight_ranks_on_one_gpu_over_gloo():
    """`python bench.py --gpus 8` (eight-process readiness without eight GPUs): the parent starts EIGHT
    ranks that share cuda:0 and exchange through gloo, at a small shape.  What an 8-GPU node's first run depends on
    besides RCCL itself is asserted from rank 0's line: world_size_seen == 8, eight per_rank entries with their own
    step times and exchange diagnostics, one bucket launch order, eight disjoint core shares (when the host has >= 8
    cores), the same number of buckets on every rank, the step's clamp veto riding on the last bucket (no extra
    collective), a finite loss.  No scaling number is read off this run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _release_parent_memory()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["SLN_DIST_BACKEND"] = "gloo"
    env["SLN_DIST_TIMEOUT_S"] = "600"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2",
                        "--warmup", "1", "--settle", "1", "--batch", "1", "--dim", "128", "--arch", "resnet50",
                        "--no-cpu-baseline", "--no-strict"], env=env, capture_output=True, text=True, timeout=1400)
    # (the FIRST rank to fail is the cause; its peers' "connection closed by peer" follow)
    first = [l for l in r.stderr.splitlines() if ("Error" in l or l.startswith("bench.py:") or "Killed" in l or
                                                    "fault" in l.lower()) and "Connection closed" not in l][:8]
    if r.returncode != 0:
        with open(os.path.join(os.environ.get("TMPDIR", "/tmp"), "sln_world8_stderr.txt"), "w") as fh:
            fh.write(r.stderr)
    assert r.returncode == 0, "\n".join(first) + "\n...\n" + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "dp8" and out["scaling"] == "weak"
    assert out["value"] > 0 and np.isfinite(out["config"]["final_loss"])
    gx = out["gradient_exchange"]
    assert gx["world_size_seen"] == 8 and gx["backend"] == "gloo"
    assert [g["rank"] for g in gx["per_rank"]] == list(range(8))
    assert gx["bucket_launch_order"] == list(range(gx["buckets"])) and gx["buckets"] >= 2
    for g in gx["per_rank"]:
        assert g["step_ms_min"] > 0 and g["finishes"] >= 2 and g["buckets"] == gx["buckets"]
        assert "exposed_wait_ms_mean" in g and "host_wait_ms_mean" in g
    if (os.cpu_count() or 1) >= 8:
        shares = [g["cpu_affinity"] for g in gx["per_rank"]]
        assert all(s.get("cores", 0) >= 1 for s in shares)
        assert all(a["last"] < b["first"] for a, b in zip(shares, shares[1:]))       # eight disjoint ranges
    assert out["config"]["clamped_and_applied_blocks"] == 0
    assert out["roofline"]["traffic"] is None
