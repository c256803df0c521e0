"""Data-parallel training: one process per GPU, torch.distributed backend "nccl"
(= RCCL on ROCm) over xGMI.  Images are independent (all BatchNorm frozen, NMS and
roi sampling per image), so ranks draw disjoint shards and the only exchange is
the gradient all-reduce.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce is bound by
one link, so gradients are packed into a few large flat buckets (default 64 MiB)
rather than many small ones -- launch latency, not bandwidth, is what a 256 MB
gradient costs at this step time.  Buckets are reduced as soon as every gradient
in them is ready (autograd post-accumulate hooks), overlapping with the rest of
backward on RCCL's own stream.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, timeout_s=None):
    """Initialise from torchrun's env (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_*).  timeout_s (or SLN_DIST_TIMEOUT_S;
    default 1800): how long a collective may wait for a peer before the process group aborts it -- a rank that
    died (its backward raised, it ran out of memory) must take its peers down with an error, not leave them
    waiting: the launcher (torchrun / bench.py) then sees non-zero exit codes and can start afresh."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    set_cpu_affinity(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # before anything touches the GPU
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local = local % torch.cuda.device_count()     # (several ranks may share a GPU in gloo test runs)
    if world > 1 and not dist.is_initialized():
        if backend is None:   # SLN_DIST_BACKEND=gloo: exercise the multi-process path on a one-GPU box
            backend = os.environ.get("SLN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        import datetime
        tmo = float(timeout_s if timeout_s is not None else os.environ.get("SLN_DIST_TIMEOUT_S", "1800"))
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=tmo))
    return rank, local, world


AFFINITY = {}
_HOST_CORES = []      # the mask this process STARTED with (remembered by the first call: the shares are slices of it)


def core_share(cores, local_rank, local_world):
    """Rank `local_rank`'s contiguous share of a sorted core list (pure: the world-8 test calls it for all ranks)."""
    per = len(cores) // local_world
    return list(cores[local_rank * per:(local_rank + 1) * per]) if per >= 1 else []


def set_cpu_affinity(local_rank, local_world):
    """One process per GPU: pin this rank (and the loader workers it will start) to its own contiguous share of the
    host cores, so that eight ranks' Python threads and input workers do not migrate over each other
    (SLN_CPU_AFFINITY=0 leaves the scheduler alone).  Called before any GPU call; a no-op for a single rank.
    The result is kept in AFFINITY for the bench line (`gradient_exchange.cpu_affinity`).
    IDEMPOTENT: entry points pin before they start their loader workers and init_distributed() pins
    again -- the share is always cut from the mask the process started with, never from an already narrowed one
    (a second call used to leave a rank on 1 / local_world^2 of the host)."""
    AFFINITY.clear()
    if local_world <= 1 or os.environ.get("SLN_CPU_AFFINITY", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        if not _HOST_CORES:
            _HOST_CORES.extend(sorted(os.sched_getaffinity(0)))
        cores = _HOST_CORES
        per = len(cores) // local_world
        if per < 1:
            return None
        mine = core_share(cores, local_rank, local_world)
        os.sched_setaffinity(0, mine)
        AFFINITY.update(cores=len(mine), first=mine[0], last=mine[-1])
        return mine
    except OSError:
        return None


def broadcast_parameters(module, src=0):
    """Replicas start from identical weights."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)
            # Caches derived from these tensors (bf16 weight parts, frozen-BN scale/shift) are keyed
            # by the tensors' version counters, which writing through `.data` does not advance: bump
            # them, or a rank would keep computing with what it held before the broadcast.
            t.mul_(1)


def derive_bucket_bytes(total_bytes, want=4, lo=4 << 20, hi=64 << 20):
    """Bucket size for a training stage with `total_bytes` of trainable gradients (round 6): a
    quarter of the stage's gradient bytes, between 4 and 64 MiB.  The exchange overlaps backward bucket by bucket,
    so a stage needs SEVERAL buckets whatever its size: stage 'all' (255.7 MB) keeps its four 64-MiB buckets, stage
    'heads' (~88 MB: one 64-MiB bucket and a remainder before) gets ~22-MiB ones, '4+' / '3+' in between.  Below
    ~4 MiB a collective's launch latency (~20 us per hop over xGMI) is no longer hidden by its own transfer."""
    return int(min(hi, max(lo, -(-total_bytes // want))))


class GradientAllReducer(object):
    """Bucketed, backward-overlapped gradient averaging over persistent flat buckets.

    Every bucket owns one flat fp32 buffer for the life of the reducer; parameter i of the bucket has a
    fixed slot in it.  attach() registers post-accumulate-grad hooks and -- on the HIP conv stack -- a
    gradient SINK: the weight-gradient reduce pass (wgrad_reduce_kernel) then writes a layer's gradient
    straight into its slot, and autograd adopts that view as `p.grad`, so the large gradients are never
    packed or copied back.  What does not arrive in its slot (biases, a few re-laid-out weights: a few
    hundred KB) is moved there by one multi-tensor copy per bucket when the bucket is launched.  When the
    last gradient of a bucket has landed, its all-reduce is launched asynchronously; buckets go out
    strictly in bucket order on every rank.  finish() (after backward, before clipping) waits, averages in
    place and leaves every `p.grad` a view of its slot.  `stats` counts, per finish(), how many gradient
    bytes were already in place and how many had to be copied.
    Calling the object with a parameter list does the same without hooks."""

    def __init__(self, params, bucket_bytes=None):
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if bucket_bytes is None:
            bucket_bytes = derive_bucket_bytes(sum(p.numel() * p.element_size() for p in self.params))
        self.bucket_bytes = bucket_bytes
        self.buckets = []
        cur, size = [], 0
        for p in reversed(self.params):  # backward produces grads roughly in reverse order
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._owner = {id(p): bi for bi, b in enumerate(self.buckets) for p in b}
        self._pending = [len(b) for b in self.buckets]
        self._flat = [None] * len(self.buckets)      # persistent flat storage, allocated at first use
        self._slot = {}                              # id(p) -> (bucket, offset)
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                self._slot[id(p)] = (bi, off)
                off += (p.numel() + 3) // 4 * 4      # 16-B aligned slots (vector loads in the optimiser)
        self._size = [sum((p.numel() + 3) // 4 * 4 for p in b) for b in self.buckets]
        # the step's clamp veto travels with the LAST bucket (round 6): one extra 16-B slot behind its gradients holds
        # 1.0 on a rank whose operand blocks clamped in this step (conv_hip.clamp_veto()); after the SUM every rank
        # reads the same "> 0" and skips -- or applies -- the same update: replicas stay identical, no extra collective
        self._veto_off = self._size[-1] if self._size else 0
        if self._size:
            self._size[-1] += 4
        self.veto = None        # after finish(): float32 [1] view, > 0 if ANY rank vetoed the step
        self._work = [None] * len(self.buckets)
        self._next = 0          # collectives are issued strictly in bucket order on every rank
        self._hooks = []
        self.stats = {"in_place_bytes": 0, "copied_bytes": 0, "copied_tensors": 0}
        # the order in which this pass's collectives were issued, and (kept until the next finish()) the previous
        # pass's: must be the same sequence on every rank -- 0, 1, 2, ... -- whatever order the gradients arrived in
        self.trace, self.last_trace = [], []
        # self-diagnosis of a scaling run (round 5): per finish() the time the main stream (nccl) / the host (gloo)
        # stood waiting for the collectives -- the part of the exchange that backward did NOT hide
        self._wait_events, self._wait_host_s, self._finishes = [], [], 0

    # ------------------------------------------------------------------ slots
    def _storage(self, bi):
        if self._flat[bi] is None:
            p0 = self.buckets[bi][0]
            self._flat[bi] = torch.zeros(self._size[bi], dtype=p0.dtype, device=p0.device)
        return self._flat[bi]

    def slot_view(self, p):
        """The parameter's slot as a tensor of the parameter's shape (contiguous)."""
        bi, off = self._slot[id(p)]
        return self._storage(bi)[off:off + p.numel()].view(p.shape)

    def _sink(self, p, shape):
        """conv_hip.GRAD_SINK: where the weight gradient of parameter p should be written, or None.  Only
        while p holds no gradient yet (a second backward before the step must ACCUMULATE, not overwrite) and
        only for gradients in the parameter's own contiguous layout."""
        if id(p) not in self._slot or p.grad is not None or tuple(shape) != tuple(p.shape) or \
                not p.is_contiguous():
            return None
        return self.slot_view(p)

    def attach(self):
        if self.world == 1:
            return self
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        if self.params and self.params[0].is_cuda:
            try:
                from . import conv_hip
                conv_hip.GRAD_SINK = self._sink
            except ImportError:
                pass
        return self

    def detach(self):
        """Remove the hooks and drop every pending bucket (call before building the reducer of the
        next training stage: hooks of a stale reducer would keep launching collectives of their own
        between the new reducer's, in an order that is only accidentally the same on every rank)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        try:
            from . import conv_hip
            if getattr(conv_hip.GRAD_SINK, "__self__", None) is self:
                conv_hip.GRAD_SINK = None
        except ImportError:
            pass
        for w in self._work:
            if w is not None:
                w.wait()
        self._work = [None] * len(self.buckets)
        self._pending = [len(b) for b in self.buckets]
        self._next = 0
        return self

    def _on_grad(self, p):
        bi = self._owner[id(p)]
        self._pending[bi] -= 1
        # A bucket is launched only once every earlier bucket has been: ranks whose gradients
        # arrive in a different order (or not at all: a rank without positive rois) would otherwise
        # issue the RCCL collectives in different orders and hang.
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi):
        from . import conv_hip
        conv_hip.flush_wgrad_reduces()   # weight gradients whose (batched) reduce pass has not been launched yet
        conv_hip.join_side_streams()     # ... or that are in flight on the side stream (conv_hip.WGRAD_STREAM)
        flat = self._storage(bi)
        src, dst, missing = [], [], []
        for p in self.buckets[bi]:
            view = self.slot_view(p)
            g = p.grad
            if g is None:
                missing.append(view)                 # no gradient on this rank: its slot contributes zeros
            elif g.data_ptr() == view.data_ptr() and g.is_contiguous():
                self.stats["in_place_bytes"] += g.numel() * g.element_size()
            else:
                src.append(g.detach())
                dst.append(view)
                self.stats["copied_bytes"] += g.numel() * g.element_size()
                self.stats["copied_tensors"] += 1
        with torch.no_grad():
            if missing:
                torch._foreach_zero_(missing)
            if dst:
                torch._foreach_copy_(dst, src)       # one multi-tensor launch (layout conversion included)
        if bi == len(self.buckets) - 1:
            # every gradient of the pass has been produced by now (its last hook fired, or finish() is flushing what
            # never arrived): whatever clamped in this step's forward or backward is in the counters
            v = self._local_veto(flat)
            with torch.no_grad():
                if v is None:
                    flat[self._veto_off:self._veto_off + 4].zero_()
                else:
                    flat[self._veto_off:self._veto_off + 4].copy_(v.expand(4))
        self.trace.append(bi)
        self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def _local_veto(self, flat):
        """This rank's veto of the step: conv_hip.clamp_veto() on the GPU, nothing on the host (tests override it)."""
        if not flat.is_cuda:
            return None
        from . import conv_hip
        return conv_hip.clamp_veto(flat.device)

    def finish(self):
        if self.world == 1:
            return
        while self._next < len(self.buckets):   # grads that never arrived (unused params): reduce now, in order
            self._launch(self._next)
            self._next += 1
        cuda = bool(self.params) and self.params[0].is_cuda
        if cuda:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        import time
        h0 = time.perf_counter()
        for bi, bucket in enumerate(self.buckets):
            self._work[bi].wait()
        self._wait_host_s.append(time.perf_counter() - h0)
        if cuda:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()                              # (nothing but the waits lies between the two events)
            self._wait_events.append((e0, e1))
            del self._wait_events[:-256]
        del self._wait_host_s[:-256]
        self._finishes += 1
        for bi, bucket in enumerate(self.buckets):
            self._flat[bi].div_(self.world)
            for p in bucket:
                view = self.slot_view(p)
                if p.grad is None or p.grad.data_ptr() != view.data_ptr() or not p.grad.is_contiguous():
                    p.grad = view                    # (no copy: the averaged gradient lives in the slot)
            self._work[bi] = None
        self._pending = [len(b) for b in self.buckets]
        self._next = 0
        self.last_trace, self.trace = self.trace, []
        self.veto = self._flat[-1][self._veto_off:self._veto_off + 1]

    def diagnostics(self, last=None):
        """Exposed all-reduce wait of the last `last` finish() calls (host sync: call it after the timed region):
        `exposed_wait_ms_*` is GPU time the main stream stood behind the collectives (backend nccl: work.wait() is a
        stream wait), `host_wait_ms_*` what the host thread spent in work.wait() (the whole wait under gloo)."""
        ev = self._wait_events[-last:] if last else self._wait_events
        hs = self._wait_host_s[-last:] if last else self._wait_host_s
        out = {"finishes": self._finishes, "buckets": len(self.buckets),
               "bucket_mib": [round(n * 4 / 2 ** 20, 1) for n in self._size]}
        if ev:
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in ev]
            out.update(exposed_wait_ms_mean=round(sum(ms) / len(ms), 3), exposed_wait_ms_max=round(max(ms), 3))
        if hs:
            out.update(host_wait_ms_mean=round(1e3 * sum(hs) / len(hs), 3), host_wait_ms_max=round(1e3 * max(hs), 3))
        return out

    def __call__(self, params=None):
        self.finish()
