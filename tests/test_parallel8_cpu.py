"""Eight-process readiness without eight GPUs: world size 8 over gloo on the
host.  What a first run on an 8-GPU node depends on besides RCCL itself -- the order of the bucket collectives on every
rank, the scale table's slot-count check, a dying rank releasing all seven peers, eight disjoint core shares, loader
workers capped to a rank's share, per-stage bucket sizes -- is asserted here; nothing in this file touches a GPU."""
import os
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_parallel_cpu import _free_port

WORLD = 8


def _env(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world))


def _gather_obj(obj, world):
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def _worker(rank, world, port, out):
    _env(rank, world, port)
    host = sorted(os.sched_getaffinity(0))
    from sln_amodal_amd import conv_hip, loader, parallel
    # ---- core shares: the entry points pin BEFORE they start their loader workers, init_distributed pins again ----
    first = parallel.set_cpu_affinity(rank, world)
    r, _, w = parallel.init_distributed(backend="gloo", timeout_s=120)
    assert (r, w) == (rank, world)
    second = sorted(os.sched_getaffinity(0))
    if len(host) >= world:
        assert first == parallel.core_share(host, rank, world) and second == first, (host, first, second)
        assert parallel.AFFINITY == dict(cores=len(first), first=first[0], last=first[-1])
        shares = _gather_obj(second, world)
        flat = [c for s in shares for c in s]
        assert len(flat) == len(set(flat)) == world * (len(host) // world), shares       # eight disjoint ranges
        assert all(a[-1] < b[0] for a, b in zip(shares, shares[1:])), shares
        # the loader's workers are capped to the share the process is pinned to now (two cores stay with the
        # training and feeder threads; never fewer than one worker)
        assert loader.capped_workers(8) == max(1, min(8, len(second) - 2))
    assert loader.capped_workers(8, cores=16) == 8 and loader.capped_workers(8, cores=6) == 4
    assert loader.capped_workers(8, cores=1) == 1

    # ---- bucket order: six buckets; every rank but 0 loses the gradients of a different bucket on some step ----
    torch.manual_seed(7)
    layers = [torch.nn.Linear(5, 5) for _ in range(6)]
    params = [p for l in reversed(layers) for p in l.parameters()]
    red = parallel.GradientAllReducer(params, bucket_bytes=100).attach()
    assert len(red.buckets) == 6, [len(b) for b in red.buckets]
    g = torch.Generator().manual_seed(rank)
    for step in range(4):
        x = torch.randn(3, 5, generator=g)
        for p in params:
            p.grad = None
        skip = (rank + step) % 7 if rank else None       # this rank's loss does not reach layer `skip` ...
        h, y = x, 0
        for i, l in enumerate(layers):
            if i == skip:                                   # ... (a shard without positive rois for one head)
                continue
            h = torch.relu(l(h))
            y = y + h.square().mean()
        y.backward()
        local = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]).clone()
        red.finish()
        traces = _gather_obj(list(red.last_trace), world)
        assert all(t == [0, 1, 2, 3, 4, 5] for t in traces), (rank, step, traces)
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        got = torch.cat([p.grad.reshape(-1) for p in params])
        assert torch.allclose(got, sum(gathered) / world, atol=1e-6), (rank, step)
    red.detach()

    # ---- the scale table's slot-count check (conv_hip.ScaleBook.exchange_amax + check_ranks) ----
    conv_hip._books.clear()
    bk = conv_hip._books[0] = conv_hip.ScaleBook(torch.device("cpu"), capacity=64)
    bk.n = 10
    bk.amax[3] = float(rank + 1)
    bk.exchange_amax()
    assert float(bk.amax[3]) == float(world)            # the MAX over the ranks
    conv_hip.check_ranks()                              # same slot count everywhere: passes
    bk.n = 11 if rank == 5 else 10                      # rank 5 built another graph
    bk.exchange_amax()
    with pytest.raises(RuntimeError, match="different slot counts"):
        conv_hip.check_ranks()                          # ... and EVERY rank learns it, none hangs
    conv_hip._books.clear()
    dist.barrier()
    dist.destroy_process_group()
    out.put(rank)


def _join(procs, limit):
    t0 = time.time()
    for p in procs:
        p.join(max(1.0, limit - (time.time() - t0)))
    hung = [p.is_alive() for p in procs]
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(10)
    return hung


@pytest.mark.timeout(400)
def test_world8_bucket_order_slot_count_check_core_shares():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    hung = _join(procs, 300)
    assert hung == [False] * WORLD
    assert [p.exitcode for p in procs] == [0] * WORLD
    assert sorted(q.get(timeout=5) for _ in range(WORLD)) == list(range(WORLD))


def _failing_worker(rank, world, port, out):
    """Rank 3's backward raises after its first bucket went out; the other seven wait in the second bucket's
    collective."""
    _env(rank, world, port)
    os.environ["SLN_CPU_AFFINITY"] = "0"
    from sln_amodal_amd import parallel
    parallel.init_distributed(backend="gloo", timeout_s=20)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(7, 13), torch.nn.ReLU(), torch.nn.Linear(13, 3))
    params = list(model[0].parameters()) + list(model[2].parameters())
    red = parallel.GradientAllReducer(params, bucket_bytes=64).attach()
    assert len(red.buckets) >= 2

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            if rank == 3:
                raise RuntimeError("injected failure in rank 3's backward")
            return g

    h = Boom.apply(model[1](model[0](torch.randn(4, 7))))
    model[2](h).square().mean().backward()
    red.finish()
    out.put(rank)


@pytest.mark.timeout(400)
def test_world8_a_failing_rank_takes_its_seven_peers_down_inside_the_timeout():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    hung = _join(procs, 150)
    assert hung == [False] * WORLD, "a rank was still waiting after 150 s (collective timeout 20 s)"
    assert all(p.exitcode not in (0, None) for p in procs), [p.exitcode for p in procs]
    assert q.empty()


def test_set_cpu_affinity_is_idempotent():
    """the entry points pin before starting their loader workers and init_distributed() pins again; the
    second call must cut the same share from the ORIGINAL mask (it used to slice the narrowed one: 1 / world^2 of
    the host).  In a child process: the test runner itself must keep its cores."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from sln_amodal_amd import parallel\n"
            "host = sorted(os.sched_getaffinity(0))\n"
            "w = 2 if len(host) >= 2 else 1\n"
            "a = parallel.set_cpu_affinity(w - 1, w); b = parallel.set_cpu_affinity(w - 1, w)\n"
            "c = parallel.set_cpu_affinity(0, w)\n"
            "print(len(host), a == b, sorted(os.sched_getaffinity(0)) == c, a, c)\n"
            "assert w == 1 or (a == b == host[len(host) // 2:2 * (len(host) // 2)] and c == host[:len(host) // 2])\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_bucket_sizes_are_derived_per_stage():
    """Every training stage gets several buckets to overlap with backward: a quarter of the stage's
    gradient bytes, clamped to 4 ... 64 MiB; stage 'all' keeps its four 64-MiB buckets."""
    from sln_amodal_amd import parallel
    mib = 1 << 20
    assert parallel.derive_bucket_bytes(int(255.7e6)) == 63925000                     # stage 'all': ~61 MiB x 4
    assert 20 * mib < parallel.derive_bucket_bytes(88 * 10 ** 6) < 23 * mib          # stage 'heads': ~21 MiB
    assert parallel.derive_bucket_bytes(10 * mib) == 4 * mib                          # floor
    assert parallel.derive_bucket_bytes(10 ** 9) == 64 * mib                          # ceiling
    # a heads-like parameter set: one 51-MB tensor + many small ones -> more than one bucket, the big one alone
    params = [torch.nn.Parameter(torch.empty(n)) for n in (1000, 300000, 12_800_000, 500000, 2_000_000, 2_000_000,
                                                              2_000_000, 1_000_000)]
    red = parallel.GradientAllReducer(params)
    assert len(red.buckets) >= 3, [sum(p.numel() for p in b) for b in red.buckets]


def test_linked_gradient_scales_ignore_members_without_history():
    """a linked group's minimum runs over the members that have a history.  A slot created and linked by
    a grad-enabled forward whose backward never ran holds the table default 1.0 and amax 0 for ever; it must neither
    drag the group down to 1.0 nor be overwritten."""
    from sln_amodal_amd import conv_hip
    bk = conv_hip.ScaleBook(torch.device("cpu"), capacity=32)
    bk.n = 6
    bk.scale[:6] = torch.tensor([2.0 ** 20, 2.0 ** 18, 1.0, 2.0 ** 22, 1.0, 2.0 ** 5])
    bk.hist[0, 0] = 1e-3          # slots 0, 1, 3 have produced a gradient; 2 and 4 never have; 5 is not linked
    bk.hist[3, 1] = 4e-3
    bk.hist[1, 3] = 2e-4
    bk.link([0, 1, 2])
    bk.link([3, 4])
    bk.apply_groups()
    assert bk.scale[:6].tolist() == [2.0 ** 18, 2.0 ** 18, 1.0, 2.0 ** 22, 1.0, 2.0 ** 5]
    bk.hist[2, 2] = 7.0           # the fresh member settles (its first backward): now it takes part
    bk.scale[2] = 2.0 ** 8
    bk.apply_groups()
    assert bk.scale[:3].tolist() == [2.0 ** 8] * 3
    # a group of unsettled members only: untouched
    bk2 = conv_hip.ScaleBook(torch.device("cpu"), capacity=8)
    bk2.n = 2
    bk2.link([0, 1])
    bk2.apply_groups()
    assert bk2.scale[:2].tolist() == [1.0, 1.0]
