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


def init_distributed(backend=None):
    """Initialise from torchrun's env (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available() and torch.cuda.device_count() > 0:
        local = local % torch.cuda.device_count()     # (several ranks may share a GPU in gloo test runs)
    if world > 1 and not dist.is_initialized():
        if backend is None:   # SLN_DIST_BACKEND=gloo: exercise the multi-process path on a one-GPU box
            backend = os.environ.get("SLN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


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


class GradientAllReducer(object):
    """Bucketed, backward-overlapped gradient averaging.

    attach() registers post-accumulate-grad hooks; when the last gradient of a
    bucket lands, the bucket is packed and its all-reduce launched asynchronously.
    finish() (call after backward, before clipping) waits, averages and unpacks.
    Calling the object with a parameter list does the same without hooks."""

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
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
        self._flat = [None] * len(self.buckets)
        self._work = [None] * len(self.buckets)
        self._next = 0          # collectives are issued strictly in bucket order on every rank
        self._hooks = []

    def attach(self):
        if self.world == 1:
            return self
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        return self

    def detach(self):
        """Remove the hooks and drop every pending bucket (call before building the reducer of the
        next training stage: hooks of a stale reducer would keep launching collectives of their own
        between the new reducer's, in an order that is only accidentally the same on every rank)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for w in self._work:
            if w is not None:
                w.wait()
        self._flat = [None] * len(self.buckets)
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
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.buckets[bi]]
        flat = torch.cat([g.reshape(-1) for g in grads])
        self._flat[bi] = flat
        self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        if self.world == 1:
            return
        while self._next < len(self.buckets):   # grads that never arrived (unused params): reduce now, in order
            self._launch(self._next)
            self._next += 1
        for bi, bucket in enumerate(self.buckets):
            self._work[bi].wait()
            flat = self._flat[bi]
            flat.div_(self.world)
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
            self._flat[bi] = None
            self._work[bi] = None
        self._pending = [len(b) for b in self.buckets]
        self._next = 0

    def __call__(self, params=None):
        self.finish()
