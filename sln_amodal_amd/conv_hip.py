"""HIP convolution backend behind nn_ops.conv_bn_act (csrc/conv.hip).

Per layer:  x --act_split--> xparts --conv_fwd(+BN affine, residual, ReLU)--> y
backward:   gy --grad_prep(ReLU mask, BN scale, bias grad)--> gzparts
            gx = conv_fwd(gzparts, mirrored/swapped weights)   (stride 1; strided
                 1x1: lattice scatter)
            gw = conv_wgrad(gzparts, xparts)
Everything runs in libsln_amodal_hip.so; torch only owns the buffers.

Operand formats (csrc/conv.hip): PARTS = 3 -> three bf16 parts, six matrix products per fp32
product; PARTS = 2 -> two fp16 parts of v*s with a per-tensor power-of-two scale s, three products.
The scales live on the device (ScaleBook): every tensor role of every layer owns a slot
(scale, running amax); a producer splits with the scale derived from the amax the same tensor had
the last time it was produced, and records the new amax -- no host round trip.  The first time a
slot is used it is bootstrapped exactly (an amax pass, then the split).
"""
import ctypes
import heapq
import os

import numpy as np
import torch

from . import _lib, ops

# 2 = two scaled fp16 parts (3 MFMA products per fp32 product; the default: same fp64-checked accuracy
# at half the matrix work); 3 = three bf16 parts (6 products, no scales)
# 1 = ONE scaled fp16 part (fp16 storage, one product: fp16-class results -- BASELINE.json configs[4]'s "fp16 MFMA";
# chosen per model, e.g. `bench.py --config resnext --parts 1`, never the default)
PARTS = int(os.environ.get("SLN_CONV_PARTS", "2"))
# parts used while autograd is disabled (the frozen GLM, inference): None = same as PARTS
PARTS_NOGRAD = int(os.environ["SLN_CONV_PARTS_NOGRAD"]) if os.environ.get("SLN_CONV_PARTS_NOGRAD") else None
# Per-layer operand format (round 6): callable(owner weight) -> 2 | 3 | None, consulted by every autograd convolution
# (forward, data and weight gradient of the layer run in the returned format; None = PARTS).  The strict 3 x bf16
# format (>= fp32 per element, no exponent floor) for the layers that need it -- MaskRCNN.set_strict_layers() maps a
# name pattern to this hook; tests/test_precision_gpu.py and tools/precision_control.py hold the evidence for WHICH.
PARTS_FOR = None


def parts_for_tagged(own):
    """The PARTS_FOR hook MaskRCNN.set_strict_layers installs: 3 for weights tagged `_sln_strict` (the tag lives on
    the Parameter object and dies with it)."""
    return 3 if (PARTS == 2 and getattr(own, "_sln_strict", False)) else None
class _NoCache(dict):
    pass


LINK_SHORTCUT_GRAD = os.environ.get("SLN_LINK_SHORTCUT_GRAD", "1") != "0"   # A/B switch
CHAIN_GRAD_PREP = os.environ.get("SLN_CHAIN_GRAD_PREP", "1") != "0"            # A/B switch
CHAIN_BLOCK_OUTPUT = os.environ.get("SLN_CHAIN_BLOCK_OUTPUT", "1") != "0"      # A/B switch
CHAIN_STATS = [0, 0]  # prepared gradients handed over by consumers / used by producers
LINK_STATS = [0, 0]   # shortcut gradients handed over by tails / consumed by heads
PAIR_STATS = [0, 0]   # strided data-gradient pairs merged on the lattice / merged into a stride-1 reader's gradient
PAIR_STRIDED = os.environ.get("SLN_PAIR_STRIDED", "1") != "0"             # A/B switch
FUSE_OUTPUT_SPLIT = True   # conv epilogue writes the output's parts (skips the next act_split)
# conv_bn_act(parts_only=True) on the autograd path (a bottleneck's conv outputs: read by convolutions, by the
# next shortcut and as ReLU masks only): the fp32 copy is not written, the shortcut and the masks are taken
# from the parts.  "0": A/B switch (fp32 + parts, 8 + 4 B per block-output element instead of 4 + 4)
PARTS_ONLY_TRAIN = os.environ.get("SLN_PARTS_ONLY_TRAIN", "1") != "0"
PO_STATS = [0, 0, 0]   # outputs produced as parts only / shortcuts read from parts / masks read from part 0
# weight gradients: split-K partial sums through a workspace + ordered reduce (bit-reproducible) instead of
# fp32 atomics; "0" restores the atomics for A/B runs
DETERMINISTIC_WGRAD = os.environ.get("SLN_DETERMINISTIC_WGRAD", "1") != "0"
# weight gradients on a second stream, next to the data gradients of the same and the following layers: both are
# whole-CU kernels, so the gain is the other kernel's tail rounds and epilogue bursts filled.  A C4 block's backward
# alone: 21.2 -> 19.9 ms (tools/two_stream_probe.py); the train step: +0.5 % (86.5 -> 86.9 img/s, same box), with
# every kernel's own duration stretched by the sharing (dominant kernel 0.43 -> 0.37 of its roofline over its own
# launches) -- OFF by default, "1" enables it.  See _side_wgrad_ok for when it applies.
WGRAD_STREAM = os.environ.get("SLN_WGRAD_STREAM", "0") == "1"
_SIDE = {}            # device index -> [stream, weights with a side-stream gradient in flight (ids), join queued]
SIDE_STATS = [0, 0]   # weight gradients launched on the side stream / on the main stream
_cache = _NoCache()   # kept for tools that call _cache.clear(); caches live on the tensors


def _side_state(device):
    st = _SIDE.get(device.index)
    if st is None:
        st = _SIDE[device.index] = [torch.cuda.Stream(device=device), set(), False]
    return st


def join_side_streams(end_of_pass=False):
    """The current stream waits for every weight gradient in flight on a side stream.  Runs by itself at the end
    of each backward pass (engine callback); anything that reads a weight gradient INSIDE the pass (bucket
    all-reduce hooks) calls it first -- the weights stay marked until the pass ends, so that a second gradient of
    the same weight (which autograd ADDS on the main stream) is still produced on the main stream."""
    for st in _SIDE.values():
        if st[1]:
            torch.cuda.current_stream(st[0].device).wait_stream(st[0])
        if end_of_pass:
            st[1].clear()
            st[2] = False


def _end_of_backward():
    flush_wgrad_reduces(True)
    join_side_streams(True)


def _adopted_untouched(own, weight):
    """True if nothing touches this weight gradient on the main stream before the pass ends: AccumulateGrad
    ADOPTS it (no kernel) -- leaf parameter without a gradient yet, contiguous, the reduce pass writes the
    parameter's own layout, no graph of the backward is recorded -- and it is the weight's first gradient of this
    pass (a second one is ADDED by autograd on the main stream, at once)."""
    if not DETERMINISTIC_WGRAD or torch.is_grad_enabled():
        return False
    if not (own.is_leaf and own.grad is None and own.is_contiguous() and own.data_ptr() == weight.data_ptr()):
        return False
    return id(own) not in _side_state(weight.device)[1] and id(own) not in _reduce_state(weight.device)[1]


def _side_wgrad_ok(own, weight):
    return WGRAD_STREAM and _adopted_untouched(own, weight)


# Deferred reduce of the two-phase weight gradients: a layer writes its split-K partial sums only; the reduce passes
# of up to 16 layers run as ONE launch (sln_wgrad_reduce_batch_f32: the same summation tree, the same bits) -- when
# 16 are waiting, when something is about to read a gradient inside the pass (flush_wgrad_reduces) and when the
# pass ends.  ~137 reduce launches per train step become ~9.  "0": A/B switch.
BATCH_WGRAD_REDUCE = os.environ.get("SLN_BATCH_WGRAD_REDUCE", "1") != "0"
_REDUCE = {}          # device index -> [pending (workspace, gradient storage, pointer, n, ksplit, taps, Cin), weight ids, callback queued]
REDUCE_STATS = [0, 0]   # batched launches, layers reduced in them


class _ReduceDesc(ctypes.Structure):
    _fields_ = [("partial", ctypes.c_void_p), ("gw", ctypes.c_void_p), ("n", ctypes.c_int64),
                ("ksplit", ctypes.c_int32), ("taps", ctypes.c_int32), ("Cin", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]     # == sln_wgrad_reduce_desc_t


def _reduce_state(device):
    st = _REDUCE.get(device.index)
    if st is None:
        st = _REDUCE[device.index] = [[], set(), False]
    return st


def flush_wgrad_reduces(end_of_pass=False):
    """Launch the reduce of every weight gradient that is waiting (current stream).  The weights stay marked until
    the pass ends: a second gradient of the same weight is reduced at once (autograd adds it to the first)."""
    for idx, st in _REDUCE.items():
        if st[0]:
            pend, st[0] = st[0], []
            arr = (_ReduceDesc * len(pend))()
            for i, (ws, _keep, gw_ptr, n, ksplit, taps, cin) in enumerate(pend):
                arr[i] = _ReduceDesc(ws.data_ptr(), gw_ptr, n, ksplit, taps, cin, 0)
            with torch.cuda.device(idx):
                _lib.check(_lib.lib().sln_wgrad_reduce_batch_f32(arr, len(pend), ops._stream()),
                           "sln_wgrad_reduce_batch_f32")
            REDUCE_STATS[0] += 1
            REDUCE_STATS[1] += len(pend)
        if end_of_pass:
            st[1].clear()
            st[2] = False


# ------------------------------------------------------------------ per-tensor scales (PARTS = 2)
SCALE_TARGET_LOG2 = 11      # max|v| * s lands in [2^10, 2^11): 2^5 of head room below fp16's 65504
SCALE_WINDOW = 16           # the scale follows the maximum over this many recent steps of the tensor
# Extra head room (bits) of the GRADIENT roles ("gz"): a gradient spikes by more than an activation grows -- the RPN
# class-logit gradient of a pyramid level depends on which anchors the step drew -- and its absolute precision floor
# (2^-25 of the scaled range) is far below what a weight-gradient sum over 10^5 pixels resolves.  Round 5: one block of
# that tensor clamped in one timed step of one bench run in three with the activations' 2^5 (profiles/HISTORY_r5.md).
# (More head room is not free: an element keeps its 22 bits only down to 2^-(19 - 5 - this) of the tensor's maximum.
# What head room cannot absorb -- a pyramid level that sees the heads' gradients once in thirty steps -- is handled by
# link_gradient_scales below.)
GRAD_HEADROOM_LOG2 = 3


class _Slot(object):
    """One tensor role of one layer: views of its device-side scale / running amax.  The table entry goes
    back to the book when the slot dies (with its layer)."""
    __slots__ = ("scale", "amax", "hist", "cursor", "fresh", "book", "idx", "sat", "headroom")

    def __del__(self):
        try:
            heapq.heappush(self.book.free, self.idx)
        except Exception:
            pass

    def __init__(self, book, idx):
        self.book = book
        self.idx = idx
        self.scale = book.scale[idx:idx + 1]
        self.amax = book.amax[idx:idx + 1]
        self.hist = book.hist[0, idx:idx + 1]      # column idx of the ring (row stride = capacity)
        self.cursor = book.cursor[idx:idx + 1]
        self.sat = book.saturated[idx:idx + 1]     # this role's own clamp counter (blocks that clamped to +-65504)
        self.headroom = book.headroom[idx:idx + 1]
        self.fresh = True            # no scale yet: the first producer bootstraps it from an amax pass


class ScaleBook(object):
    def __init__(self, device, capacity=1 << 15):
        self.device = device
        self.amax = torch.zeros(capacity, dtype=torch.float32, device=device)
        self.scale = torch.ones(capacity, dtype=torch.float32, device=device)
        self.hist = torch.zeros((SCALE_WINDOW, capacity), dtype=torch.float32, device=device)   # ring of maxima
        self.cursor = torch.zeros(capacity, dtype=torch.int32, device=device)
        # one clamp counter PER SLOT (round 5: the bench line says which tensor role clamped, and in which step)
        self.saturated = torch.zeros(capacity, dtype=torch.int32, device=device)
        self.retired_saturated = torch.zeros((), dtype=torch.int64, device=device)    # clamp counts of dead slots
        self.headroom = torch.zeros(capacity, dtype=torch.int8, device=device)     # extra bits below SCALE_TARGET_LOG2
        self.n = 0
        self._n_written, self._n_reduced = None, False
        # groups of GRADIENT slots that share one scale (link_gradient_scales): {sorted index tuple}; the padded
        # [G, K] index tensor is rebuilt when a group is added
        self.groups, self._gidx = set(), None
        self.free = []               # indices of dead slots (a heap: the lowest index is reused first, so
        #                              ranks that free the same layers reuse the same entries whatever the
        #                              order their garbage collectors ran in)
        self.names = {}              # slot index -> (role key, owner shape): diagnostics (tools/sat_probe.py)

    def new_slot(self):
        if self.free:                        # an entry whose layer has died: back to the initial state
            idx = heapq.heappop(self.free)
            tables = [self.amax, self.scale, self.hist, self.cursor, self.saturated, self.retired_saturated]
            versions = [t._version for t in tables]
            #  the dead role's clamp count moves to the book's retired total: the recycled slot starts at
            # zero (saturation_report attributes nothing of the old role to the new one), the process total stays
            self.retired_saturated += self.saturated[idx]
            self.saturated[idx] = 0
            self.names.pop(idx, None)
            self.amax[idx] = 0.0
            self.scale[idx] = 1.0
            self.hist[:, idx] = 0.0
            self.cursor[idx] = 0
            self.headroom[idx] = 0
            if any(idx in g for g in self.groups):
                self.groups = {g for g in self.groups if idx not in g}
                self._gidx = None
            # (the tables are only ever written by kernels; autograd has saved views of `scale` whose version
            # check must not trip over the reset of an unrelated, dead entry)
            torch._C._autograd._unsafe_set_version_counter(tables, versions)
            return _Slot(self, idx)
        if self.n >= self.amax.numel() - 2:     # (the last two entries carry the slot count, see update())
            raise RuntimeError("ScaleBook is full (%d tensor slots)" % self.n)
        self.n += 1
        return _Slot(self, self.n - 1)

    def update(self, sync=True):
        """Delayed scaling: every slot's next scale from the amax it recorded since the last call.
        Data-parallel replicas take the maximum over all ranks first (one small MAX all-reduce per
        step), so that every rank derives the same scales and the replicas keep computing the same
        function bit for bit (a rank-local power of two would round the fp16 parts differently).
        sync=False (inference / validation forwards, which one rank may run alone): no collective."""
        if self.n:
            if sync:
                self.exchange_amax()
            _lib.check(_lib.lib().sln_scale_update_headroom_f32(
                ops._ptr(self.amax), ops._ptr(self.scale), ops._ptr(self.hist), ops._ptr(self.cursor),
                ops._ptr(self.headroom), self.n, self.hist.shape[1], SCALE_WINDOW, SCALE_TARGET_LOG2, ops._stream()),
                "sln_scale_update_headroom_f32")
            self.apply_groups()

    def apply_groups(self):
        """Linked gradient slots take the SMALLEST scale of their group (the most head room): three small device-side
        ops, no host sync; like the update kernel they must not advance autograd's version.  Pure torch."""
        if not self.groups:
            return
        if self._gidx is None:
            k = max(len(g) for g in self.groups)
            self._gidx = torch.tensor([list(g) + [g[0]] * (k - len(g)) for g in sorted(self.groups)],
                                      dtype=torch.int64, device=self.device)
        v = self.scale._version
        with torch.no_grad():
            #  only members WITH A HISTORY take part: a slot created and linked by a forward whose
            # backward never ran (a loss evaluation, a predict(mode='training') without backward) still holds
            # the table default 1.0 -- taken into the minimum it would drag every linked gradient scale down
            # to 1.0, where gradients of ~1e-6 underflow, for the life of the model (its amax stays 0, so it
            # never settles by itself).  Such members neither contribute nor are overwritten.
            sc = self.scale[self._gidx]
            settled = self.hist[:, self._gidx.flatten()].amax(dim=0).view_as(sc) > 0
            m = torch.where(settled, sc, torch.full_like(sc, float("inf"))).min(dim=1, keepdim=True).values
            self.scale.index_put_((self._gidx,), torch.where(settled, m.expand_as(sc), sc))
        torch._C._autograd._unsafe_set_version_counter([self.scale], [v])

    def exchange_amax(self):
        """Data-parallel half of update(): the MAX all-reduce of the amax table (pure torch: the world-8 gloo test
        on CPU runs exactly this).  ONE fixed-size collective over the whole table (128 KB), whatever the ranks'
        slot counts: its last two entries carry (n, -n), so that ranks which built different graphs -- whose
        element-wise MAX would mix unrelated tensors -- are found by check_ranks() instead of hanging in a
        collective of mismatched sizes."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        if self._n_written != self.n:
            v = self.amax._version
            self.amax[-2] = float(self.n)
            self.amax[-1] = -float(self.n)
            torch._C._autograd._unsafe_set_version_counter([self.amax], [v])
            self._n_written = self.n
        dist.all_reduce(self.amax, op=dist.ReduceOp.MAX)
        self._n_reduced = True

    def link(self, idxs):
        g = tuple(sorted(set(idxs)))
        if len(g) > 1 and not any(set(g) <= set(h) for h in self.groups):
            # (a group that grew -- a new resolution of a shared weight -- replaces its subsets)
            self.groups = {h for h in self.groups if not set(h) <= set(g)}
            self.groups.add(g)
            self._gidx = None

    def settle(self, slot):
        """Bootstrap: slot.amax holds an exact amax pass -> its scale; clears the fresh flag."""
        _lib.check(_lib.lib().sln_scale_update_headroom_f32(
            ops._ptr(slot.amax), ops._ptr(slot.scale), ops._ptr(slot.hist), ops._ptr(slot.cursor),
            ops._ptr(slot.headroom), 1, slot.book.hist.shape[1], SCALE_WINDOW, SCALE_TARGET_LOG2, ops._stream()),
            "sln_scale_update_headroom_f32")
        slot.fresh = False


_books = {}
SCALE_EPOCH = [0]     # advanced by update_scales(): cached activation parts encode the scale of their epoch


def book(device):
    b = _books.get(device.index)
    if b is None:
        b = _books[device.index] = ScaleBook(device)
    return b


def update_scales(sync=True):
    """Call once per step (MaskRCNN.predict does): scales follow the previous step's amax.
    sync: take the maximum over the data-parallel ranks first (training steps, which every rank runs in
    lockstep); inference / validation forwards pass False and stay rank-local.
    A backward pass of a graph built BEFORE this call would de-scale its weight gradient with the new
    scale of operands that were quantised with the old one: _ConvFn.backward raises in that case."""
    flush_wgrad_reduces(True)    # (a backward pass that raised never ran its end-of-pass callback)
    join_side_streams(True)
    if hold_scales.depth:
        return
    for b in _books.values():
        b.update(sync)
    SCALE_EPOCH[0] += 1
    _arena_reset()


def check_ranks():
    """Host-side check (one sync; call it where the loop syncs anyway, e.g. once per epoch) that every
    data-parallel rank held the same number of scale slots at the last synchronised update."""
    for b in _books.values():
        if b._n_reduced:
            hi, lo = float(b.amax[-2]), -float(b.amax[-1])
            if hi != lo or hi != b.n:
                raise RuntimeError("ScaleBook: ranks hold different slot counts (%d..%d, here %d): the "
                                   "replicas built different graphs" % (lo, hi, b.n))


def saturation_count():
    """Blocks that clamped a value to +-65504 since the start (host sync; tests / monitoring)."""
    return sum(int(b.saturated.sum().item()) + int(b.retired_saturated.item()) for b in _books.values())


# ------------------------------------------------------------------ no silently applied clamped step (round 6)
# A train step in which ANY operand block had to clamp a value to +-65504 computed with under-estimated operands
# (38 k such blocks in the 16 set-up steps of a cold file-fed run, every one of those steps APPLIED).
# The step is now vetoed on the device, like a non-finite gradient norm: clamp_mark() at the start of the step,
# clamp_veto() after backward -> a 0 / 1 device scalar that the optimiser's guard reads (optim.ClippedSGD.step) and,
# under data parallelism, that travels with the LAST gradient bucket so that every rank skips the same step
# (parallel.GradientAllReducer).  The clamped tensor's amax was recorded before the clamp: the next update_scales()
# follows it, and the step after a skipped one runs with room.  "0": A/B switch (round-5 behaviour: count only).
SKIP_CLAMPED_STEPS = os.environ.get("SLN_SKIP_CLAMPED_STEPS", "1") != "0"
_CLAMP_MARK = {}


def saturation_total(idx):
    """0-d int64 DEVICE tensor: blocks of book `idx` that clamped since start-up (no host sync)."""
    b = _books[idx]
    return b.saturated[:max(b.n, 1)].sum(dtype=torch.int64) + b.retired_saturated


def clamp_mark():
    """Start of a train step: remember every book's clamp total (device-side)."""
    _CLAMP_MARK.clear()
    if SKIP_CLAMPED_STEPS:
        for i in _books:
            _CLAMP_MARK[i] = saturation_total(i)


def clamp_veto(device=None):
    """-> float32 [1] device tensor, 1.0 if a block clamped since clamp_mark() (on `device`'s book, default: any
    single book), else 0.0; None when nothing was marked (guard off, no HIP convolution ran, CPU)."""
    if not SKIP_CLAMPED_STEPS or not _CLAMP_MARK:
        return None
    out = None
    for i, mark in _CLAMP_MARK.items():
        if device is not None and device.index is not None and i != device.index:
            continue
        v = (saturation_total(i) != mark).to(torch.float32).reshape(1)
        out = v if out is None else torch.maximum(out, v.to(out.device))
    return out


def saturation_snapshot():
    """Device-side copy of every book's per-slot clamp counters (no host sync): take one per step, diff them with
    saturation_report() afterwards to learn which tensor role clamped in which step."""
    return {i: b.saturated.clone() for i, b in _books.items()}


def saturation_report(before=None, after=None):
    """[(role key, owner shape, blocks)] of the slots whose clamp counter moved between two snapshots (host sync).
    before=None: since start-up; after=None: now."""
    out = []
    for i, b in _books.items():
        cur = (after[i] if after is not None else b.saturated).cpu()
        if before is not None and i in before:
            cur = cur - before[i].cpu()
        for idx in torch.nonzero(cur > 0).flatten().tolist():      # (a slot recycled in between reads negative)
            key, shape = b.names.get(idx, (("?",), ()))
            out.append(("/".join(str(k) for k in key), list(shape), int(cur[idx])))
    return out


def _slot(owner, key):
    """The scale slot of tensor role `key` of the layer identified by `owner` (its weight Parameter:
    the slot lives on the object, so it dies with the layer)."""
    slots = getattr(owner, "_sln_slots", None)
    if slots is None:
        slots = {}
        owner._sln_slots = slots
    sl = slots.get(key)
    if sl is None:
        sl = slots[key] = book(owner.device).new_slot()
        sl.book.names[sl.idx] = (key, tuple(owner.shape))
        if key and key[0] == "gz" and GRAD_HEADROOM_LOG2:
            sl.headroom.fill_(GRAD_HEADROOM_LOG2)
    return sl


def link_gradient_scales(owners):
    """The GRADIENT roles ("gz") of the given layers (their weight objects) share one scale from now on: the smallest
    of the group after every update.  For tensors whose maxima come and go TOGETHER -- the same weights applied to every
    pyramid level (the RPN), the FPN's per-level output / lateral convolutions: whichever level a step's positive rois
    fall on carries the mask and class gradients, thousands of times the RPN-only gradient of the other levels, so a
    single level's own 16-step window can miss them (one bench run in five clamped 19 blocks of the finest level's
    gradient even with 2^11 of head room) while the maximum over the levels is there in every step.  Cheap: a set
    lookup per call once the group exists."""
    idxs, bk = [], None
    for o in owners:
        for key, sl in (getattr(o, "_sln_slots", None) or {}).items():
            if key and key[0] == "gz":
                idxs.append(sl.idx)
                bk = sl.book
    if bk is not None:
        bk.link(idxs)


def _q3(slot):
    """(q_scale, q_amax, q_saturated) pointer triple of a slot (NULLs without one)."""
    if slot is None:
        return None, None, None
    return ops._ptr(slot.scale), ops._ptr(slot.amax), ops._ptr(slot.sat)


def supports(conv, x):
    """Convolutions the implicit-GEMM kernels take directly.  The 3-channel stems (K = 7*7*3) go
    through _StemFn instead (im2col to K = 160 + 1x1 convolution): padded to 8 channels per tap the
    implicit GEMM would waste 5/8 of K."""
    return (x.dtype == torch.float32 and conv.groups == 1 and conv.weight.dtype == torch.float32 and
            conv.padding_mode == "zeros" and conv.in_channels >= 8)


def _pad8(c):
    return (c + 7) // 8 * 8


ROWS, TILED256, TILED256H = 0, 1, 2     # weight-part layouts (include/sln_amodal.h)


def weights_layout(M, Cout, Cin_pad, taps, parts, x_pixels):
    """The layout the forward kernel for this problem reads (a pure host rule of the library)."""
    return _lib.lib().sln_conv_fwd_weights_layout(M, Cout, Cin_pad, taps, parts, x_pixels)


def _split_weights(weight, flip_swap=False, parts=None, owner=None, layout=ROWS):
    """-> (wparts, scale tensor or None); wparts [parts][O][KH][KW][I_pad] (layout ROWS) or the flat
    LDS-image order of the 256x256 kernel (TILED256).  Cached on the tensor object itself (keyed by
    its version counter), so the cache dies with the tensor and can never alias a new tensor that
    happens to reuse the address."""
    parts = parts or PARTS
    cache = getattr(weight, "_sln_wparts", None)
    if cache is None:
        cache = {}
        try:
            weight._sln_wparts = cache
        except Exception:
            pass
    hit = cache.get((flip_swap, parts, layout))
    if hit is not None and hit[0] == weight._version:
        return hit[1], hit[2]
    if hit is not None and parts <= 2 and BATCH_WEIGHT_SPLITS and (id(weight), flip_swap, layout, parts) in _WSPLIT:
        # stale (the optimiser stepped): every stale registered entry is refreshed in one launch
        _refresh_weight_parts(weight.device)
        hit = cache.get((flip_swap, parts, layout))
        if hit is not None and hit[0] == weight._version:
            return hit[1], hit[2]
    w = weight.detach()
    Co, Ci, KH, KW = w.shape
    s = w.stride()
    if flip_swap:   # data gradient: out channel <-> in channel, taps mirrored
        O, I, so, si = Ci, Co, s[1], s[0]
    else:
        O, I, so, si = Co, Ci, s[0], s[1]
    Ip = _pad8(I)
    if layout != ROWS:
        out = torch.empty((parts * _lib.lib().sln_conv_tiled_weight_elems(O, I, KH, KW, layout),),
                          dtype=torch.bfloat16, device=w.device)
    else:
        out = torch.empty((parts, O, KH, KW, Ip), dtype=torch.bfloat16, device=w.device)
    slot = _slot(owner if owner is not None else weight, ("w",)) if parts <= 2 else None

    def launch(dst):
        _lib.check(_lib.lib().sln_conv_split_weights_f32(
            ops._ptr(w), O, I, Ip, KH, KW, so, si, s[2], s[3], 1 if flip_swap else 0, parts, layout,
            ops._ptr(dst), *_q3(slot), ops._stream()), "sln_conv_split_weights_f32")
    if slot is not None and slot.fresh:
        launch(None)
        slot.book.settle(slot)
    launch(out)
    q = slot.scale if slot is not None else None
    cache[(flip_swap, parts, layout)] = (weight._version, out, q)
    if parts <= 2 and BATCH_WEIGHT_SPLITS and weight.requires_grad and weight.is_leaf:   # (Parameters: stable objects)
        _register_weight_split(weight, flip_swap, layout, out, slot, (O, I, Ip, KH, KW, so, si, s[2], s[3]), parts)
    return out, q


# ------------------------------------------------------------------ batched re-split of the trainable weights
BATCH_WEIGHT_SPLITS = os.environ.get("SLN_BATCH_WEIGHT_SPLITS", "1") != "0"     # A/B switch
_WSPLIT = {}            # (id(weight), flip, layout) -> entry dict
_WSPLIT_STATE = {"table": None, "order": None, "chunks": {}}
WSPLIT_CHUNK = 1 << 16
WSPLIT_STATS = [0, 0]   # batched refreshes, entries refreshed
_DESC = np.dtype([("w", "<u8"), ("out", "<u8"), ("q_scale", "<u8"), ("q_amax", "<u8"), ("q_sat", "<u8"),
                  ("s_o", "<i8"), ("s_i", "<i8"), ("s_kh", "<i8"), ("s_kw", "<i8"), ("total", "<i8"),
                  ("O", "<i4"), ("I", "<i4"), ("Ip", "<i4"), ("KH", "<i4"), ("KW", "<i4"), ("flip", "<i4"),
                  ("layout", "<i4"), ("reserved", "<i4")])     # == sln_split_desc_t


def _register_weight_split(weight, flip, layout, out, slot, geom, parts=2):
    import weakref
    O, I, Ip, KH, KW, so, si, skh, skw = geom
    total = out.numel() // parts
    # (the descriptor's last word: 1 = ONE scaled fp16 part, else two -- sln_split_desc_t.reserved, ABI 11)
    _WSPLIT[(id(weight), flip, layout, parts)] = dict(ref=weakref.ref(weight), flip=flip, layout=layout, out=out,
                                                      slot=slot, parts=parts,
                                                      desc=(weight.data_ptr(), out.data_ptr(), slot.scale.data_ptr(),
                                                            slot.amax.data_ptr(), slot.sat.data_ptr(), so, si, skh,
                                                            skw, total, O, I, Ip, KH, KW, 1 if flip else 0, layout,
                                                            1 if parts == 1 else 0))
    _WSPLIT_STATE["order"] = None        # the device table is rebuilt at the next refresh


def _refresh_weight_parts(device):
    """Re-split every registered weight whose version moved since its parts were written: one launch."""
    st = _WSPLIT_STATE
    dead = [k for k, e in _WSPLIT.items() if e["ref"]() is None or
            e["ref"]().data_ptr() != e["desc"][0]]
    for k in dead:
        del _WSPLIT[k]
        st["order"] = None
    if st["order"] is None:
        st["order"] = [k for k, e in _WSPLIT.items() if e["out"].device == device]
        tab = np.zeros(len(st["order"]), _DESC)
        for i, k in enumerate(st["order"]):
            tab[i] = _WSPLIT[k]["desc"]
        st["table"] = torch.from_numpy(tab.view(np.uint8)).to(device)
        st["chunks"] = {}
    stale = []
    for i, k in enumerate(st["order"]):
        e = _WSPLIT[k]
        wt = e["ref"]()
        c = getattr(wt, "_sln_wparts", {}).get((e["flip"], e["parts"], e["layout"]))
        if c is not None and c[1] is e["out"] and c[0] != wt._version and not e["slot"].fresh:
            stale.append(i)
    if not stale:
        return
    key = tuple(stale)
    ch = st["chunks"].get(key)
    if ch is None:
        ce, cf = [], []
        for i in stale:
            total = _WSPLIT[st["order"][i]]["desc"][9]
            for off in range(0, total, WSPLIT_CHUNK):
                ce.append(i)
                cf.append(off)
        ch = st["chunks"][key] = (torch.tensor(ce, dtype=torch.int32, device=device),
                                  torch.tensor(cf, dtype=torch.int64, device=device))
    _lib.check(_lib.lib().sln_conv_split_weights_batch_f32(ops._ptr(st["table"]), ops._ptr(ch[0]), ops._ptr(ch[1]),
                                                           ch[0].numel(), WSPLIT_CHUNK, ops._stream()),
               "sln_conv_split_weights_batch_f32")
    for i in stale:
        e = _WSPLIT[st["order"][i]]
        wt = e["ref"]()
        wt._sln_wparts[(e["flip"], e["parts"], e["layout"])] = (wt._version, e["out"], e["slot"].scale)
    WSPLIT_STATS[0] += 1
    WSPLIT_STATS[1] += len(stale)


_split = _split_weights


def _nhwc(t):
    """Physically NHWC view of a logical [N,C,H,W] tensor (copy only if needed)."""
    if t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous():
        return t
    return t.contiguous(memory_format=torch.channels_last)


def _act_split(xc2d, M, C, parts, slot):
    """fp32 [M, C] -> parts [P, M, C_pad]; bootstraps a fresh slot with an amax pass."""
    Cp = _pad8(C)
    out = torch.empty((parts, M, Cp), dtype=torch.bfloat16, device=xc2d.device)

    def launch(dst):
        _lib.check(_lib.lib().sln_act_split_f32(ops._ptr(xc2d), M, C, Cp, parts, ops._ptr(dst), *_q3(slot),
                                                ops._stream()), "sln_act_split_f32")
    if slot is not None and slot.fresh:
        launch(None)
        slot.book.settle(slot)
    launch(out)
    return out


# ------------------------------------------------------------------ zeroed arena for per-channel sums
# Bias gradients (the column sums of a prepared gradient) are accumulated with atomics into zeroed memory: ~130
# fill launches per train step when every layer clears its own [Cout] vector.  Instead one arena per device is
# cleared once per step (update_scales) and the layers take slices of it (SLN_SUMS_PREZEROED).  A slice lives as
# long as its tensor: the arena is REPLACED, not overwritten, when a step begins.
SUMS_ARENA = os.environ.get("SLN_SUMS_ARENA", "1") != "0"      # A/B switch
_ARENA = {}           # device index -> [tensor, cursor]
ARENA_FLOATS = 1 << 20
SLN_SUMS_PREZEROED = 0x100


def _arena_reset():
    for st in _ARENA.values():
        st[0] = torch.zeros(ARENA_FLOATS, dtype=torch.float32, device=st[0].device)
        st[1] = 0


def _zeroed(n, device):
    """-> (float32 [n] tensor, already_zero flag): a slice of the step's arena, or fresh memory the callee clears."""
    if not SUMS_ARENA:
        return torch.empty((n,), dtype=torch.float32, device=device), 0
    st = _ARENA.get(device.index)
    if st is None:
        st = _ARENA[device.index] = [torch.zeros(ARENA_FLOATS, dtype=torch.float32, device=device), 0]
    a = (n + 63) // 64 * 64
    if st[1] + a > ARENA_FLOATS:          # (no update_scales() between many backward passes: start a new arena)
        st[0] = torch.zeros(ARENA_FLOATS, dtype=torch.float32, device=device)
        st[1] = 0
    out = st[0][st[1]:st[1] + n]
    st[1] += a
    return out, SLN_SUMS_PREZEROED


_NAN = {}


def _placeholder(shape, device):
    """What stands in the autograd graph for an activation that exists as parts only: a zero-stride view of
    one NaN (no memory; anything that mistakes it for data turns into NaN at once).  The parts travel on
    the tensor object (`_sln_parts`, `_sln_po`)."""
    n = _NAN.get(device)
    if n is None:
        n = _NAN[device] = torch.full((1,), float("nan"), dtype=torch.float32, device=device)
    return n.expand(shape)


def _parts_value(parts, q):
    """fp32 [M, C_pad] value of scaled fp16 parts [P, M, C_pad] (P = 1 or 2): (h0 [+ h1]) / s."""
    v = parts[0].view(torch.float16).float()
    if parts.shape[0] == 2:
        v = v + parts[1].view(torch.float16).float()
    return v / q


def parts_only_of(t):
    """(parts [2, M, C_pad], scale) of a parts-only activation, None for an ordinary tensor."""
    po = getattr(t, "_sln_po", None)
    if po is None:
        return None
    if po[2] != SCALE_EPOCH[0]:
        raise RuntimeError("a parts-only activation outlived update_scales(): it cannot be split again")
    return po[0], po[1]


def materialize(t):
    """fp32 [N,C,H,W] (NHWC in memory) value of a parts-only activation, (h0 + h1) / s -- what its readers
    see; an ordinary tensor is returned as it is.  Debugging, tests, and readers outside the conv stack."""
    po = parts_only_of(t)
    if po is None:
        return t
    parts, q = po
    N, C, H, W = t.shape
    return _parts_value(parts, q)[:, :C].reshape(N, H, W, C).permute(0, 3, 1, 2)


def act_parts(x, parts=None, owner=None, key=None):
    """x logical [N,C,H,W] (NHWC in memory) -> (parts [P, N*H*W, C_pad], scale or None).  Cached on
    the tensor object: one activation often feeds several convs (block input -> conv1 +
    downsample, ASPP's four branches, the P-maps -> RPN + heads).  owner / key: the consuming layer's
    scale slot for this input (PARTS = 2; the first consumer's slot serves the others)."""
    parts = parts or PARTS
    hit = getattr(x, "_sln_parts", None)
    if hit is not None and hit[0] == (x._version, parts, SCALE_EPOCH[0]):
        return hit[1], hit[2]
    if getattr(x, "_sln_po", None) is not None:
        raise RuntimeError("a parts-only activation (%d x fp16, scale epoch %s) cannot be re-split as %d parts in "
                           "epoch %d" % (2, x._sln_po[2], parts, SCALE_EPOCH[0]))
    xc = _nhwc(x.detach())
    N, C, H, W = xc.shape
    slot = None
    if parts <= 2:
        slot = _slot(owner if owner is not None else x, ("x",) + tuple(key or (H, W)))
    out = _act_split(xc, N * H * W, C, parts, slot)
    q = slot.scale if slot is not None else None
    try:
        x._sln_parts = ((x._version, parts, SCALE_EPOCH[0]), out, q)   # (a tensor that outlives
        # update_scales() -- the same input fed again next step -- is split again with the new scale)
    except Exception:
        pass
    return out, q


# bench.py sets this to a list to collect (start_event, end_event, flops, kernel) per
# launch on the launch stream (torch's current stream) -- the live roofline numbers.
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, flops, name, shape="", rd_bytes=0, wr_bytes=0):
    """rd/wr_bytes: the launch's algorithmic HBM traffic (every operand read once, every output
    written once) -- what the PMC figures in profiles/ are compared against."""
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((e0, e1, flops, name, shape, rd_bytes, wr_bytes))


def _fwd_kernel_name(layout, parts, launched=False):
    """Name of the kernel sln_conv2d_fwd_ms_f32 launches for a problem with this weight layout; launched=True: of the
    kernel the call just made on this thread DID launch (the library decides between conv_fwd256h_kernel and its
    128 x 256 sibling per epilogue kind)."""
    if launched and PROFILE is not None and _lib.lib().sln_conv_fwd_last_kernel() == 3:
        return "conv_fwd128x256h_kernel"
    if layout == TILED256H:
        return "conv_fwd256h_kernel"
    return ("conv_fwd256_kernel<%d>" if layout == TILED256 else "conv_fwd_kernel<%d>") % parts


def _nbytes(*tensors):
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)


def wsrc(weight, parts=None, flip_swap=False, owner=None):
    """Weight operand of _fwd: split lazily there, in the layout the chosen kernel reads."""
    return (weight, bool(flip_swap), owner, parts or PARTS)


def _fwd(xparts, N, H, W, w, Cout, KH, KW, stride, dil, pt, pl, OH, OW, scale, shift, residual,
         relu, cin=None, out_parts=False, mask=None, want_y=True, want_colsum=False, post_scale=None,
         xq=None, yslot=None, res_parts=None, mask_parts=None):
    """One launch of the forward kernel.  w = wsrc(weight, parts, flip_swap, owner).  mask /
    want_y=False / want_colsum: the epilogue extras of sln_conv2d_fwd_ms_f32 (a data gradient that
    is consumed only as the previous layer's prepared gradient).  xq: the activation parts' scale
    tensor (PARTS = 2); yslot: the scale slot of the output's own parts.  res_parts = (parts, scale) /
    mask_parts = parts: the residual / the ReLU pattern of a tensor that exists as parts only.  Returns y, or
    (y_or_None, parts, colsum) when any extra is used; the parts' scale is yslot.scale."""
    import ctypes as C
    dev = xparts.device
    weight, flip, owner, P = w
    layout = weights_layout(N * OH * OW, Cout, xparts.shape[2], KH * KW, P, xparts.shape[1])
    wparts, wq = _split_weights(weight, flip, P, owner, layout)
    cin = cin or xparts.shape[2]
    y = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=dev).permute(0, 3, 1, 2) if want_y else None
    plain = mask is None and mask_parts is None and want_y and not want_colsum and post_scale is None
    if (res_parts is not None or mask_parts is not None) and (P > 2 or Cout % 8 or layout == TILED256):
        # (the fixed-feature epilogue only: whole 16-B row groups, not the round-1 256^2 kernel)
        if res_parts is not None:
            residual, res_parts = _parts_value(res_parts[0], res_parts[1])[:, :Cout].contiguous(), None
        if mask_parts is not None:
            mask = (mask_parts[0].view(torch.float16)[:, :Cout] > 0).float().contiguous()
            mask_parts = None
    if out_parts and P <= 2:
        if yslot is None:
            raise RuntimeError("two-part output split needs a scale slot")
        if yslot.fresh and not plain:
            raise RuntimeError("chained gradient preparation on a tensor without a scale (the chain "
                               "must stay off until its slot has been bootstrapped)")
    fresh = out_parts and P <= 2 and yslot.fresh
    yp = None
    if out_parts and not fresh:   # the epilogue also emits the output's parts (next layer's operand)
        alloc = torch.empty if Cout % 8 == 0 else torch.zeros   # pad channels must be zero
        yp = alloc((P, N * OH * OW, _pad8(Cout)), dtype=torch.bfloat16, device=dev)
    cs, cs_flag = _zeroed(Cout, dev) if want_colsum else (None, 0)
    pb = (OH - 1) * stride[0] + dil[0] * (KH - 1) + 1 - H - pt
    pr = (OW - 1) * stride[1] + dil[1] * (KW - 1) + 1 - W - pl
    seg = (C.c_int32 * 3)(N, H, W)
    e0 = _prof_begin()
    _lib.check(_lib.lib().sln_conv2d_fwd_ms_f32(
        ops._ptr(xparts), 1, seg, xparts.shape[2], ops._ptr(wparts), layout, P, Cout, KH, KW,
        stride[0], stride[1], dil[0], dil[1], pt, pl, pb, pr, ops._ptr(scale), ops._ptr(shift),
        ops._ptr(residual), (1 if relu else 0) | cs_flag, ops._ptr(mask), ops._ptr(post_scale), ops._ptr(y),
        ops._ptr(yp), ops._ptr(cs), ops._ptr(xq), ops._ptr(wq), *_q3(yslot if yp is not None and P <= 2 else None),
        ops._ptr(res_parts[0]) if res_parts is not None else None,
        ops._ptr(res_parts[1]) if res_parts is not None else None,
        ops._ptr(mask_parts) if mask_parts is not None else None,
        ops._stream()), "sln_conv2d_fwd_ms_f32")
    if res_parts is not None:
        PO_STATS[1] += 1
    if mask_parts is not None:
        PO_STATS[2] += 1
    _prof_end(e0, 2.0 * N * OH * OW * Cout * KH * KW * cin,
              _fwd_kernel_name(layout, P, launched=True),
              "fwd N%d %dx%d C%d->%d k%d s%d d%d" % (N, H, W, cin, Cout, KH, stride[0], dil[0]),
              _nbytes(xparts, wparts, residual, mask, res_parts[0] if res_parts is not None else None) +
              (_nbytes(mask_parts) // 2 if mask_parts is not None else 0), _nbytes(y, yp))
    if fresh:   # first use of this output's slot: exact amax pass over y, then the split
        yp = _act_split(_nhwc(y), N * OH * OW, Cout, P, yslot)
    if yp is not None and y is not None and post_scale is None:
        y._sln_parts = ((y._version, P, SCALE_EPOCH[0]), yp, yslot.scale if P <= 2 else None)
    if mask is not None or mask_parts is not None or not want_y or want_colsum:
        return y, yp, cs
    return y


class MultiScale(object):
    """Image groups of different sizes (the GLM's three scales, modal/msc_deeplab.py:29-37)
    carried as ONE flat NHWC buffer so that every layer is a single launch over all of
    them (sln_conv2d_fwd_ms_f32).  Forward only, no autograd: the GLM is frozen.
    segs = [(N, H, W), ...]; y = fp32 [sum N*H*W, C]; parts = bf16 [P, sum N*H*W, C_pad]."""

    def __init__(self, segs, y, parts=None):
        self.segs, self.y, self.parts = list(segs), y, parts

    @classmethod
    def pack(cls, tensors):
        """[N_s, C, H_s, W_s] tensors -> one flat buffer (one copy per scale)."""
        C = tensors[0].shape[1]
        segs = [(t.shape[0], t.shape[2], t.shape[3]) for t in tensors]
        y = torch.cat([t.detach().permute(0, 2, 3, 1).reshape(-1, C) for t in tensors], dim=0)
        return cls(segs, y.contiguous())

    def tensors(self):
        """Per-scale logical [N, C, H, W] views (channels-last in memory)."""
        out, m = [], 0
        C = self.y.shape[1]
        for N, H, W in self.segs:
            n = N * H * W
            out.append(self.y[m:m + n].view(N, H, W, C).permute(0, 3, 1, 2))
            m += n
        return out

    def get_parts(self, parts, owner=None):
        """-> (parts, scale or None)."""
        if self.parts is None or self.parts.shape[0] != parts:
            M, C = self.y.shape
            slot = _slot(owner, ("x_ms",) + tuple(self.segs)) if parts <= 2 else None
            self.parts = _act_split(self.y, M, C, parts, slot)
            self.q = slot.scale if slot is not None else None
        return self.parts, getattr(self, "q", None)


def conv_bn_act_ms(x, conv, bn, relu, residual, pads, parts_only=False):
    """conv (+bias) -> frozen-BN affine -> (+residual) -> ReLU on a MultiScale, one launch.
    parts_only: the output is read by convolutions only (the GLM bottlenecks' reduce / 3x3 layers: forward
    only, no ReLU mask to keep) -- its fp32 copy is not written, the result carries the parts alone."""
    import ctypes as C
    from .nn_ops import bn_affine
    if torch.is_grad_enabled() and (conv.weight.requires_grad or
                                    (conv.bias is not None and conv.bias.requires_grad)):
        raise RuntimeError("MultiScale convolutions are forward-only (frozen GLM)")
    parts = PARTS_NOGRAD or PARTS
    scale = shift = None
    if bn is not None:
        scale, shift = bn_affine(bn)
    if conv.bias is not None:
        shift = conv.bias * scale + shift if scale is not None else conv.bias
    scale = scale.detach().contiguous() if scale is not None else None
    shift = shift.detach().contiguous() if shift is not None else None
    Co, Ci, KH, KW = conv.weight.shape
    pt, pb, pl, pr = pads
    sh, sw = conv.stride
    dh, dw = conv.dilation
    osegs, flops = [], 0.0
    for N, H, W in x.segs:
        OH = (H + pt + pb - dh * (KH - 1) - 1) // sh + 1
        OW = (W + pl + pr - dw * (KW - 1) - 1) // sw + 1
        osegs.append((N, OH, OW))
        flops += 2.0 * N * OH * OW * Co * KH * KW * Ci
    M = sum(n * h * w for n, h, w in osegs)
    xp, xq = x.get_parts(parts, owner=conv.weight)
    layout = weights_layout(M, Co, xp.shape[2], KH * KW, parts, xp.shape[1])
    wp, wq = _split_weights(conv.weight, parts=parts, layout=layout)
    dev = xp.device
    yslot = _slot(conv.weight, ("y_ms",) + tuple(osegs)) if parts <= 2 else None
    fresh = yslot is not None and yslot.fresh
    yp = None
    if not fresh:
        alloc = torch.empty if Co % 8 == 0 else torch.zeros
        yp = alloc((parts, M, _pad8(Co)), dtype=torch.bfloat16, device=dev)
    # (a fresh slot bootstraps its scale from the fp32 output: that one time it is written)
    y = None if (parts_only and yp is not None and parts <= 2) else \
        torch.empty((M, Co), dtype=torch.float32, device=dev)
    seg = (C.c_int32 * (3 * len(x.segs)))(*[v for s_ in x.segs for v in s_])
    res = residual.y if residual is not None else None
    rparts = rq = None
    if residual is not None and res is None:          # a parts-only shortcut: added from its parts
        if residual.parts is None or residual.parts.shape[0] > 2:
            raise ValueError("parts-only residual without scaled fp16 parts")
        if parts <= 2 and residual.parts.shape[0] == parts and Co % 8 == 0 and layout != TILED256 and \
                tuple(residual.parts.shape[1:]) == (M, Co):
            rparts, rq = residual.parts, residual.q
            PO_STATS[1] += 1
        else:
            res = _parts_value(residual.parts, residual.q)[:, :Co].contiguous()
    if res is not None and tuple(res.shape) != (M, Co):
        raise ValueError("residual does not match the convolution output")
    e0 = _prof_begin()
    _lib.check(_lib.lib().sln_conv2d_fwd_ms_f32(
        ops._ptr(xp), len(x.segs), seg, xp.shape[2], ops._ptr(wp), layout, parts, Co, KH, KW, sh, sw, dh, dw,
        pt, pl, pb, pr, ops._ptr(scale), ops._ptr(shift), ops._ptr(res), 1 if relu else 0, None, None,
        ops._ptr(y), ops._ptr(yp), None, ops._ptr(xq), ops._ptr(wq),
        *_q3(yslot if yp is not None else None), ops._ptr(rparts), ops._ptr(rq), None, ops._stream()),
        "sln_conv2d_fwd_ms_f32")
    _prof_end(e0, flops, _fwd_kernel_name(layout, parts, launched=True),
              "fwd ms%s C%d->%d k%d s%d d%d" % ("+".join("%dx%d" % (h, w) for _, h, w in x.segs), Ci, Co, KH, sh, dh),
              _nbytes(xp, wp, res, rparts), _nbytes(y, yp))
    if fresh:
        yp = _act_split(y, M, Co, parts, yslot)
    out = MultiScale(osegs, y, yp)
    out.q = yslot.scale if yslot is not None else None
    return out


def _grad_prep(gy, y, scale, want_gu, want_bias, parts, slot=None):
    """-> (gz parts, gu or None, bias gradient or None); the parts' scale is slot.scale."""
    gy = _nhwc(gy)
    N, C, H, W = gy.shape
    M, Cp = N * H * W, _pad8(C)
    gz = torch.empty((parts, M, Cp), dtype=torch.bfloat16, device=gy.device)
    gu = gy
    y16 = None
    if y is not None and y.dtype == torch.bfloat16:     # the layer's output exists as parts only: [2, M, Cp]
        if parts > 2 or tuple(y.shape[1:]) != (M, Cp):
            raise RuntimeError("parts-only ReLU pattern does not match the gradient")
        y16, y = y, None
        PO_STATS[2] += 1
    write_gu = want_gu and (y is not None or y16 is not None)
    if write_gu:
        gu = torch.empty((N, H, W, C), dtype=torch.float32, device=gy.device).permute(0, 3, 1, 2)
    gb, gb_flag = _zeroed(C, gy.device) if want_bias else (None, 0)
    if parts <= 2 and slot is None:
        raise RuntimeError("two-part gradient preparation needs a scale slot")

    def launch(dst):
        _lib.check(_lib.lib().sln_conv_grad_prep_f32(
            ops._ptr(gy), ops._ptr(y), ops._ptr(y16), ops._ptr(scale), M, C, Cp, parts | (gb_flag if dst is not None else 0),
            ops._ptr(gu) if write_gu else None, ops._ptr(dst), ops._ptr(gb), *_q3(slot if parts <= 2 else None),
            ops._stream()), "sln_conv_grad_prep_f32")
    if parts <= 2 and slot.fresh:
        launch(None)
        slot.book.settle(slot)
    launch(gz)
    return gz, (gu if want_gu else None), gb


def _grad_prep_pooled(pooled, y, scale, want_bias, parts, slot):
    """_grad_prep for a layer whose output went through a max-pool: the layer's gradient is gathered from the
    pooled gradient inside the preparation kernel.  pooled = (g [N,C,OH,OW] channels-last, winning taps
    [N,OH,OW,C] uint8, (N, C, H, W, kernel, stride, pad_top, pad_left, OH, OW)) as nn_ops._MaxPoolFn left it."""
    g, arg, (N, C, H, W, kernel, stride, pt, pl, OH, OW) = pooled
    g = _nhwc(g)
    M = N * H * W
    gz = torch.empty((parts, M, C), dtype=torch.bfloat16, device=g.device)
    gb, gb_flag = _zeroed(C, g.device) if want_bias else (None, 0)
    if parts <= 2 and slot is None:
        raise RuntimeError("two-part gradient preparation needs a scale slot")

    def launch(dst):
        _lib.check(_lib.lib().sln_conv_grad_prep_pooled_f32(
            ops._ptr(g), ops._ptr(arg), N, H, W, kernel, stride, pt, pl, OH, OW, ops._ptr(y), ops._ptr(scale), C,
            parts | (gb_flag if dst is not None else 0), ops._ptr(dst), ops._ptr(gb),
            *_q3(slot if parts <= 2 else None), ops._stream()), "sln_conv_grad_prep_pooled_f32")
    if parts <= 2 and slot.fresh:
        launch(None)
        slot.book.settle(slot)
    launch(gz)
    return gz, None, gb


# ------------------------------------------------------------------ bias folded into the frozen-BN shift
# shift' = bias * bn_scale + bn_shift of every conv with a bias and a frozen BN (the whole detector: the
# reference's convs all carry a bias, modals.py:264-355).  The biases train, so the value changes every step
# -- for all layers at once, when the optimiser steps.  Instead of two 5-us kernels per layer per step the
# stale entries are recomputed together (two foreach launches) the first time one of them is asked for.
_FOLDS = {}          # id(bias) -> [weakref(bias), bn_scale, bn_shift, bias version, folded shift]
FOLD_STATS = [0, 0]  # batched refreshes, entries refreshed


def folded_shift(bias, bn_scale, bn_shift):
    import weakref
    e = _FOLDS.get(id(bias))
    if e is not None and e[0]() is bias and e[1] is bn_scale and e[2] is bn_shift:
        if e[3] == bias._version:
            return e[4]
        # stale: refresh every stale entry of this device in one go
        live, dead = [], []
        for k, f in _FOLDS.items():
            b = f[0]()
            if b is None:
                dead.append(k)
            elif b._version != f[3] and b.device == bias.device and b.dtype == bias.dtype:
                live.append((f, b))
        for k in dead:
            del _FOLDS[k]
        with torch.no_grad():
            outs = torch._foreach_mul([b.detach() for _, b in live], [f[1] for f, _ in live])
            torch._foreach_add_(outs, [f[2] for f, _ in live])
        for (f, b), o in zip(live, outs):
            f[3], f[4] = b._version, o
        FOLD_STATS[0] += 1
        FOLD_STATS[1] += len(live)
        return e[4]
    with torch.no_grad():
        out = (bias.detach() * bn_scale + bn_shift).contiguous()
    try:
        _FOLDS[id(bias)] = [weakref.ref(bias), bn_scale, bn_shift, bias._version, out]
    except TypeError:
        pass
    return out


# Data-parallel training (parallel.GradientAllReducer): callable(parameter, shape) -> the tensor the layer's
# weight gradient should be written into (the parameter's slot of a flat all-reduce bucket), or None.
GRAD_SINK = None

# Parity tests only (tests/test_e2e_gpu.py): callable(owner weight, y) -> bool tensor [n <= N, C, H, W] or None.
# The ReLU pattern of a layer's output is then FORCED to the given one (a unit that is on keeps its value or
# gets the smallest positive one, a unit that is off becomes 0), so that a backward pass can be compared with
# a reference whose units near zero fell the other way -- everything that consumes the output (the next
# layer, the shortcut, every ReLU mask of the backward pass) sees the patched tensor.
FORCE_RELU = None


def _force_relu(own, y):
    m = FORCE_RELU(own, y)
    if m is None:
        return
    n = m.shape[0]
    po = getattr(y, "_sln_po", None)
    if po is not None:       # parts only: the pattern is the sign of part 0
        N, C, H, W = y.shape
        h = po[0].view(torch.int16).view(po[0].shape[0], N, H, W, po[0].shape[2])[..., :C]
        mm = m.permute(0, 2, 3, 1)
        one, zero = torch.ones((), dtype=torch.int16, device=h.device), torch.zeros((), dtype=torch.int16, device=h.device)
        h[0, :n] = torch.where(mm, torch.where(h[0, :n] > 0, h[0, :n], one), zero)   # (bits 1 = the least fp16)
        if h.shape[0] == 2:
            h[1, :n] = torch.where(mm, h[1, :n], zero)
        return
    hit = getattr(y, "_sln_parts", None)
    with torch.no_grad():
        y[:n] = torch.where(m, y[:n].clamp_min(1e-30), torch.zeros((), dtype=y.dtype, device=y.device))
    if hit is not None:      # the epilogue's parts stay valid (a flipped unit is ~1e-7 of the tensor's range)
        y._sln_parts = ((y._version,) + tuple(hit[0][1:]), hit[1], hit[2])


def _check_epoch(ctx, parts):
    """The saved operand parts were quantised with the scales of the epoch the forward ran in; the scale
    tensors saved next to them are live views of the ScaleBook.  After another update_scales() (a second
    predict() / detect() between this graph's forward and backward) the two no longer belong together
    and the weight gradient would be off by a power of two, silently."""
    if parts <= 2 and ctx.scale_epoch != SCALE_EPOCH[0]:
        raise RuntimeError("conv backward after update_scales(): this graph's forward ran in scale epoch %d, "
                           "the ScaleBook is at %d (run backward before the next predict()/detect(), or "
                           "run that forward under conv_hip.hold_scales())" % (ctx.scale_epoch, SCALE_EPOCH[0]))


class hold_scales(object):
    """Context manager: update_scales() is a no-op inside (a validation / detect() pass run while a training
    graph is still waiting for its backward keeps that graph's scales)."""
    depth = 0

    def __enter__(self):
        hold_scales.depth += 1
        return self

    def __exit__(self, *exc):
        hold_scales.depth -= 1
        return False


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bn_scale, bn_shift, residual, relu, stride, dil, pads, link=None,
                chain_in=None, chain_out=None, owner=None, pair=None, parts_only=False, inbox=None):
        ctx.inbox = None
        if inbox is not None and stride == (1, 1) and ctx.needs_input_grad[0] and \
                (chain_in is None or chain_in.get("soft")):
            ctx.inbox = inbox              # (the plain stride-1 data gradient below is the one that adds it)
            inbox.clear()
            inbox["armed"] = True
        parts = PARTS
        if PARTS_NOGRAD and not any(ctx.needs_input_grad):
            parts = PARTS_NOGRAD
        Co, Ci, KH, KW = weight.shape
        N, _, H, W = x.shape
        pt, pb, pl, pr = pads
        OH = (H + pt + pb - dil[0] * (KH - 1) - 1) // stride[0] + 1
        OW = (W + pl + pr - dil[1] * (KW - 1) - 1) // stride[1] + 1
        scale, shift = bn_scale, bn_shift
        if bias is not None:
            frozen = bn_scale is not None and not (bn_scale.requires_grad or bn_shift.requires_grad)
            shift = (folded_shift(bias, bn_scale, bn_shift) if frozen else bias * bn_scale + bn_shift) \
                if bn_scale is not None else bias
        if shift is not None:
            shift = shift.detach().contiguous()
        if scale is not None:
            scale = scale.detach().contiguous()
        res_po = parts_only_of(residual) if residual is not None else None
        res = _nhwc(residual.detach()) if (residual is not None and res_po is None) else None
        # scale slots (PARTS = 2) live on the layer's persistent weight object: `owner` when the weight
        # passed in is a temporary view of it (Linear / deconv reshapes)
        own = owner if owner is not None else weight
        if PARTS_FOR is not None and parts == PARTS:
            parts = PARTS_FOR(own) or parts
        xp, xq = act_parts(x, parts, owner=own)
        x_po = getattr(x, "_sln_po", None) is not None
        yslot = _slot(own, ("y", OH, OW)) if (parts <= 2 and FUSE_OUTPUT_SPLIT) else None
        gzslot = _slot(own, ("gz", OH, OW)) if parts <= 2 else None
        # parts-only output: this layer's readers are convolutions, a shortcut add and ReLU masks (the caller
        # says so); needs a scale with a history (the first step bootstraps it from the fp32 output) and the
        # fixed-feature epilogue
        po = bool(parts_only and PARTS_ONLY_TRAIN and parts <= 2 and FUSE_OUTPUT_SPLIT and not yslot.fresh and
                  Co % 8 == 0 and
                  weights_layout(N * OH * OW, Co, xp.shape[2], KH * KW, parts, xp.shape[1]) != TILED256)
        if po:
            _, yp_, _ = _fwd(xp, N, H, W, wsrc(weight, parts, False, own), Co, KH, KW, stride, dil, pt, pl, OH, OW,
                             scale, shift, res, relu, cin=Ci, out_parts=True, want_y=False, xq=xq, yslot=yslot,
                             res_parts=res_po)
            y = _placeholder((N, Co, OH, OW), xp.device)
            y._sln_parts = ((y._version, parts, SCALE_EPOCH[0]), yp_, yslot.scale)
            y._sln_po = (yp_, yslot.scale, SCALE_EPOCH[0])
            PO_STATS[0] += 1
        else:
            yp_ = None
            y = _fwd(xp, N, H, W, wsrc(weight, parts, False, own), Co, KH, KW, stride, dil, pt, pl, OH, OW, scale,
                     shift, res, relu, cin=Ci, out_parts=FUSE_OUTPUT_SPLIT, xq=xq, yslot=yslot, res_parts=res_po)
        if FORCE_RELU is not None and relu:
            _force_relu(own, y)
        need_w = ctx.needs_input_grad[1]
        # identity-shortcut link (Bottleneck.forward): the conv that consumes x (head) and the
        # conv that adds the same x as its residual (tail) share a dict, so that the tail's
        # gradient for x is added inside the head's data-gradient epilogue instead of by a
        # separate autograd accumulation pass over the whole activation.
        # the two strided 1x1 convolutions of a stage's first block (conv1 and the downsample, modals.py:269,
        # 331-337) read the same x: their data gradients live on the same stride lattice, so whichever runs
        # second adds the other's quarter-size result to its own and builds the zero-filled full-size map
        # ONCE (one fill, one scatter and one full-size autograd add less per stage transition)
        ctx.pair = ctx.pair_base = None
        if pair is not None and PAIR_STRIDED and stride != (1, 1) and (KH, KW) == (1, 1) and ctx.needs_input_grad[0]:
            pair["n"] = pair.get("n", 0) + 1
            ctx.pair = pair
        elif pair is not None and PAIR_STRIDED and stride == (1, 1) and pair.get("n") == 2 and \
                ctx.needs_input_grad[0] and link is None and chain_in is None and residual is None:
            # a third, stride-1 reader of the same x (the FPN lateral of c2..c4, created after the stage's
            # convs and therefore differentiated before them): its full-size data gradient becomes the map the
            # strided pair adds its lattice into -- no zero fill, no autograd add
            ctx.pair_base = pair
        ctx.link_head = ctx.link_tail = None
        if link is not None:
            if residual is None:
                if stride == (1, 1):
                    link["head_wants_dx"] = bool(ctx.needs_input_grad[0])
                    ctx.link_head = link
            elif link.get("head_wants_dx") and ctx.needs_input_grad[5]:
                ctx.link_tail = link
        # producer -> sole-consumer chain (conv1 -> conv2 -> conv3 of a bottleneck): the
        # consumer's data gradient is read by nothing but this layer's gradient preparation
        # (ReLU mask x BN scale -> bf16 parts, bias sums), so the consumer's dgrad epilogue
        # does that preparation itself and hands over the parts; the fp32 gradient is never
        # written and sln_conv_grad_prep_f32 is not launched for this layer.
        ctx.chain_in = ctx.chain_out = None
        # A producer WITH a residual (conv3: block output = relu(bn(conv) + shortcut)) still needs
        # the masked fp32 gradient -- it is its shortcut's gradient -- so its reader writes that as
        # its dx and applies the BN scale to the parts only (post_scale); CHAIN_BLOCK_OUTPUT.
        with_res = residual is not None
        # (PARTS = 2: a reader can only prepare this layer's gradient once that gradient's scale slot
        # has a history, i.e. from the second step on; the first step bootstraps it in _grad_prep)
        if CHAIN_GRAD_PREP and chain_out is not None and (ctx.needs_input_grad[0] or need_w) and \
                (not with_res or (CHAIN_BLOCK_OUTPUT and relu)) and not (parts <= 2 and gzslot.fresh):
            chain_out.update(active=True, scale=scale, relu=bool(relu), parts=parts, with_res=with_res,
                             want_bias=bool(bias is not None and ctx.needs_input_grad[2]), gz_slot=gzslot)
            chain_out.setdefault("readers", 1)      # 2: two sibling convs read the output (RPN heads)
            if chain_out["readers"] == 2:
                if with_res:
                    chain_out["active"] = False     # (not needed on the path: keep the simple cases)
                else:
                    ctx.set_materialize_grads(False)   # both readers return None: backward sees gy = None
            ctx.chain_out = chain_out if chain_out["active"] else None
        # (a block output's reader must also carry the shortcut's gradient: only as a link head)
        if CHAIN_GRAD_PREP and chain_in is not None and chain_in.get("active") and stride == (1, 1) and \
                ctx.needs_input_grad[0] and chain_in["parts"] == parts and \
                (not chain_in["with_res"] or ctx.link_head is not None):
            chain_in["consumer"] = chain_in.get("consumer", 0) + 1
            ctx.chain_in = chain_in
        # the producer's ReLU mask is this layer's own input: saved here as an input tensor
        # (the shared dict must not hold activations: a dict -> output -> grad_fn -> ctx -> dict
        # cycle through the C++ graph would never be collected)
        # (a parts-only tensor's ReLU pattern is the sign of its part 0: the parts themselves are saved -- bf16
        # dtype marks them in backward -- no fp32 copy of the activation exists anywhere)
        mask_x = (xp if x_po else x) if (ctx.chain_in is not None and chain_in["relu"]) else None
        ctx.save_for_backward(xp if need_w else None, weight, scale, (yp_ if po else y) if relu else None, mask_x,
                              xq if need_w else None)
        ctx.cfg = (stride, dil, pads, relu, bias is not None, residual is not None, (N, Ci, H, W),
                   parts)
        ctx.out_hw = (OH, OW)
        ctx.own, ctx.gzslot = own, gzslot
        ctx.scale_epoch = SCALE_EPOCH[0]
        return y

    @staticmethod
    def backward(ctx, gy):
        xp, weight, scale, y, mask_x, xq = ctx.saved_tensors
        own = ctx.own
        stride, dil, pads, relu, has_bias, has_res, xshape, parts = ctx.cfg
        _check_epoch(ctx, parts)
        pt, pb, pl, pr = pads
        Co, Ci, KH, KW = weight.shape
        N, _, H, W = xshape
        OH, OW = ctx.out_hw if gy is None else (gy.shape[2], gy.shape[3])
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        want_res = has_res and ctx.needs_input_grad[5]
        want_bias = has_bias and ctx.needs_input_grad[2]
        ch = ctx.chain_out
        if ch is not None and ch.get("readers") == 2 and ch.get("consumer") != 2:
            ch = None                      # only one of the two readers took part: ordinary path
            if gy is None:
                raise RuntimeError("chained gradient: no gradient arrived for a two-reader activation")
        if ch is not None and ch.get("soft") and ch.get("consumer") and "gz" in ch and \
                not (gy is not None and gy.stride() == (0, 0, 0, 0)):
            # the reader consumed the inbox deposit and handed its prepared gradient over, yet something other than
            # its placeholder arrives: a further reader added to it -- its share is in neither tensor
            raise RuntimeError("soft chain: the reader handed the prepared gradient over but the activation "
                               "has another consumer")
        if ch is not None and ch.get("soft") and ch.get("consumer") and "gz" not in ch:
            # a SOFT chain (an FPN output under the RPN conv): the reader hands the prepared gradient over only when
            # it knows its data gradient is this activation's WHOLE gradient (the crops' share arrived through its
            # inbox); otherwise the ordinary gradient arrives here and is prepared as usual
            for key in ("gz", "gzq", "gbias"):
                ch.pop(key, None)
            ch = None
        if ch is not None and ch.get("readers") == 2:
            if gy is not None or "gz" not in ch:
                raise RuntimeError("chained gradient: a two-reader activation has a third consumer, or "
                                   "one of its readers did not run")
            gz, g_res = ch.pop("gz"), None
            gzq = ch.pop("gzq", None)
            g_bias = ch.pop("gbias") if want_bias else None
            CHAIN_STATS[1] += 1
        elif ch is not None and ch.get("consumer"):
            # what arrives must be exactly what the reader produced: the zero-stride placeholder,
            # or (block outputs) the reader's own masked dx -- anything else means autograd added a
            # second reader's gradient, which the handed-over parts do not contain
            # (block outputs: the reader keeps its own reference to that dx in the dict, so autograd is
            # never its sole owner and cannot fold a second reader's gradient into it IN PLACE -- a sum
            # is a new tensor, caught by the pointer; the version guards the in-place case anyway)
            ref = ch.pop("gu_ref", None)
            ok = "gz" in ch and ((ref is not None and gy.data_ptr() == ref.data_ptr() and
                                  gy._version == ch.pop("gu_version", None)) if (ch["with_res"] or ch.get("keep_dx"))
                                 else gy.stride() == (0, 0, 0, 0))
            del ref
            if not ok:
                raise RuntimeError("chained gradient: the consumer's dgrad did not run, or the "
                                   "activation has a second consumer")
            gz = ch.pop("gz")
            gzq = ch.pop("gzq", None)
            g_res = gy if (ch["with_res"] and want_res) else None   # already masked by the reader
            g_bias = ch.pop("gbias") if want_bias else None
            if gz.shape[1] != N * OH * OW:
                # the reader saw this output re-viewed as [k M, Cout / k] (the mask head's deconv-as-1x1 under its
                # pointwise logits conv, nn_ops.deconv2x2_relu_conv1x1): the same memory
                if gz.shape[1] * gz.shape[2] != N * OH * OW * Co or Co % 8:
                    raise RuntimeError("chained gradient: the reader's view of this output does not match")
                gz = gz.view(gz.shape[0], N * OH * OW, Co)
                if g_bias is not None and g_bias.numel() != Co:
                    # the reader's sums run over ITS channels (Cout / k): this layer's bias is that vector k times
                    # (bias.repeat(k)), whose gradient autograd sums over the copies -- the total goes to the first
                    full = torch.zeros(Co, dtype=g_bias.dtype, device=g_bias.device)
                    full[:g_bias.numel()] = g_bias
                    g_bias = full
            CHAIN_STATS[1] += 1
        else:
            gz, g_res, g_bias = _grad_prep(gy, y, scale, want_res, want_bias, parts, slot=ctx.gzslot)
            gzq = ctx.gzslot.scale if parts <= 2 else None

        def mask_kw(m):      # the producer's ReLU pattern: its fp32 output, or its parts (part 0's sign)
            if m is None:
                return {}
            return dict(mask_parts=m) if m.dtype == torch.bfloat16 else dict(mask=_nhwc(m))
        side = gw_t = ws = None
        ws_bytes = 0
        if need_w:
            # the gradient tensor is allocated BEFORE the event below: whatever used its memory before has been
            # queued on this stream by now and is complete when the side stream passes the event
            rst = _reduce_state(weight.device)
            sink_used = False
            # a SECOND gradient of this weight inside the pass (decided before this gradient leaves its own mark)
            second = id(own) in rst[1] or id(own) in _side_state(weight.device)[1]
            if DETERMINISTIC_WGRAD and GRAD_SINK is not None and own.is_leaf and \
                    own.data_ptr() == weight.data_ptr() and tuple(own.shape) == (Co, Ci, KH, KW) and not second:
                # the parameter's slot of its all-reduce bucket: autograd adopts the view as .grad (no pack /
                # unpack copies around the collective).  Only the weight's FIRST gradient of a pass goes there:
                # a weight used several times per pass (the RPN convs: five pyramid levels) gets private memory
                # for the later ones, which autograd adds to the first -- written to the same slot they would
                # overwrite each other before the sum is formed.
                gw_t = GRAD_SINK(own, (Co, Ci, KH, KW))
                sink_used = gw_t is not None
            if gw_t is None:
                gw_t = torch.empty((Co, Ci, KH, KW) if DETERMINISTIC_WGRAD else (Co, KH, KW, Ci), dtype=torch.float32,
                                   device=weight.device)
            untouched = _adopted_untouched(own, weight)
            defer = BATCH_WGRAD_REDUCE and untouched and N > 0
            if WGRAD_STREAM and untouched:
                side = _side_state(weight.device)
                side[1].add(id(own))
                ready = torch.cuda.Event()
                ready.record()                      # gz (and x) are complete here; the data gradient starts below
                defer = False                       # (the side stream reduces its own gradients at once)
            if defer or sink_used:
                rst[1].add(id(own))                 # (marks live until the pass ends: _end_of_backward)
            if (side is not None or defer or sink_used) and not (rst[2] or _side_state(weight.device)[2]):
                rst[2] = True
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
        if ctx.link_tail is not None and g_res is not None:
            ctx.link_tail["idgrad"] = g_res     # consumed by the head's data gradient below
            g_res = None
            LINK_STATS[0] += 1
        id_grad = ctx.link_head.pop("idgrad", None) if ctx.link_head is not None else None
        gx = gw = None
        if id_grad is not None:
            LINK_STATS[1] += 1
        if id_grad is not None and not need_x:
            raise RuntimeError("identity-shortcut gradient was handed over but dx is not computed")
        if need_x:
            wt = wsrc(weight, parts, True, own)
            qs = dict(xq=gzq)
            two = ctx.chain_in is not None and ctx.chain_in.get("readers") == 2
            if two and ctx.chain_in.get("consumer") != 2:
                two = False
                ctx.chain_in = None                 # the sibling is not chained: ordinary data gradient
            if two and "partial" not in ctx.chain_in:
                # first of the two readers to run: plain fp32 data gradient, parked for the sibling
                ctx.chain_in["partial"] = _fwd(gz, N, OH, OW, wt, Ci, KH, KW, (1, 1), dil,
                                               dil[0] * (KH - 1) - pt, dil[1] * (KW - 1) - pl, H, W, None, None,
                                               None, False, cin=Co, **qs)
                gx = None
            elif two:
                ci = ctx.chain_in
                _, gz_up, gb_up = _fwd(gz, N, OH, OW, wt, Ci, KH, KW, (1, 1), dil, dil[0] * (KH - 1) - pt,
                                       dil[1] * (KW - 1) - pl, H, W, None, None, _nhwc(ci.pop("partial")), False,
                                       cin=Co, out_parts=True, want_y=False,
                                       want_colsum=ci["want_bias"], post_scale=ci["scale"],
                                       yslot=ci["gz_slot"], **mask_kw(mask_x), **qs)
                ci["gz"], ci["gbias"] = gz_up, gb_up
                ci["gzq"] = ci["gz_slot"].scale if parts <= 2 else None
                gx = None
                CHAIN_STATS[0] += 1
            elif ctx.chain_in is not None and (ctx.chain_in["with_res"] or ctx.chain_in.get("keep_dx")):
                # (keep_dx: the producer's output reaches this conv through an op whose gradient w.r.t. it is the
                # identity -- the FPN merge: lateral + upsampled top -- so this data gradient is BOTH that op's
                # incoming gradient, still needed as fp32 for the other addend, and the producer's gy)
                ci = ctx.chain_in
                gx, gz_up, gb_up = _fwd(gz, N, OH, OW, wt, Ci, KH, KW, (1, 1), dil, dil[0] * (KH - 1) - pt,
                                        dil[1] * (KW - 1) - pl, H, W, None, None,
                                        _nhwc(id_grad) if id_grad is not None else None, False, cin=Co,
                                        out_parts=True, want_y=True,
                                        want_colsum=ci["want_bias"], post_scale=ci["scale"],
                                        yslot=ci["gz_slot"], **mask_kw(mask_x), **qs)
                ci["gz"], ci["gbias"], ci["gu_ref"], ci["gu_version"] = gz_up, gb_up, gx, gx._version
                ci["gzq"] = ci["gz_slot"].scale if parts <= 2 else None
                CHAIN_STATS[0] += 1
            elif ctx.chain_in is not None and not (ctx.chain_in.get("soft") and
                                                   (ctx.inbox is None or "g" not in ctx.inbox)):
                ci = ctx.chain_in
                # (soft chain: taken only when the other reader's gradient is in the inbox -- then this data
                # gradient plus that deposit, added as the epilogue's residual, is the producer's whole gradient)
                extra = ctx.inbox.take() if (ci.get("soft") and ctx.inbox is not None) else None
                # (the producer's BN scale multiplies the SUM of this data gradient and the deposit: post_scale, as in
                # the keep_dx / with_res branches -- in the scale position it would leave the deposit unscaled)
                _, gz_up, gb_up = _fwd(gz, N, OH, OW, wt, Ci, KH, KW, (1, 1), dil, dil[0] * (KH - 1) - pt,
                                       dil[1] * (KW - 1) - pl, H, W,
                                       None if extra is not None else ci["scale"], None,
                                       _nhwc(extra) if extra is not None else None, False, cin=Co,
                                       out_parts=True,
                                       want_y=False, want_colsum=ci["want_bias"], yslot=ci["gz_slot"],
                                       post_scale=ci["scale"] if extra is not None else None,
                                       **mask_kw(mask_x), **qs)
                ci["gz"], ci["gbias"] = gz_up, gb_up
                ci["gzq"] = ci["gz_slot"].scale if parts <= 2 else None
                gx = _dummy_grad(weight.device).expand(N, Ci, H, W)   # never read: see chain_out above
                CHAIN_STATS[0] += 1
            elif stride == (1, 1):
                extra = ctx.inbox.take() if ctx.inbox is not None else None      # another reader's gradient w.r.t. x
                if extra is not None and id_grad is not None:
                    id_grad = id_grad + extra
                elif extra is not None:
                    id_grad = extra
                gx = _fwd(gz, N, OH, OW, wt, Ci, KH, KW, (1, 1), dil, dil[0] * (KH - 1) - pt,
                          dil[1] * (KW - 1) - pl, H, W, None, None,
                          _nhwc(id_grad) if id_grad is not None else None, False, cin=Co, **qs)
                pb = ctx.pair_base
                if pb is not None:
                    if pb.get("pair_done"):          # the pair ran first (not the engine's usual order):
                        pb["pair_done"] = False      # ordinary accumulation by autograd
                    elif "base" not in pb:
                        pb["base"] = gx              # handed to the strided pair, which returns the sum
                        gx = None
            elif KH == 1 and KW == 1 and pads == (0, 0, 0, 0):
                # strided 1x1: the gradient lives on the stride lattice, zero elsewhere
                small = _fwd(gz, N, OH, OW, wt, Ci, 1, 1, (1, 1), (1, 1), 0, 0, OH, OW, None, None,
                             None, False, cin=Co, **qs)
                pr = ctx.pair if (ctx.pair is not None and ctx.pair.get("n") == 2) else None
                if pr is not None and "small" not in pr:
                    pr["small"] = small                 # the sibling builds the map
                    gx = None
                else:
                    if pr is not None:
                        other = pr.pop("small")
                        if other.shape == small.shape:
                            small = small + other
                            PAIR_STATS[0] += 1
                        else:                            # (cannot happen for the two convs of one block)
                            raise RuntimeError("paired strided data gradients of different shapes")
                    base = pr.pop("base", None) if pr is not None else None
                    if pr is not None:
                        pr["pair_done"] = base is None
                    if base is not None and tuple(base.shape) == (N, Ci, H, W):
                        gx = base                        # the stride-1 reader's gradient: add the lattice in place
                        gx[:, :, ::stride[0], ::stride[1]] += small
                        PAIR_STATS[1] += 1
                    else:
                        if base is not None:
                            raise RuntimeError("stride-1 reader's data gradient does not match the paired input")
                        gx = torch.zeros((N, H, W, Ci), dtype=torch.float32, device=weight.device).permute(0, 3, 1, 2)
                        gx[:, :, ::stride[0], ::stride[1]] = small
            else:
                raise NotImplementedError("data gradient of a strided %dx%d conv" % (KH, KW))
        if need_w:
            # with the workspace the reduce pass writes the parameter's own [Co,Ci,KH,KW] order (no layout
            # copy in AccumulateGrad); the atomic path produces [Co,KH,KW,Ci]
            own_layout = DETERMINISTIC_WGRAD
            main = torch.cuda.current_stream(weight.device)
            if side is not None:
                side[0].wait_event(ready)
                for tns in (gz, xp):
                    tns.record_stream(side[0])      # freed by this stream's allocator only after the side stream is done
                SIDE_STATS[0] += 1
            else:
                if id(own) in _side_state(weight.device)[1]:
                    main.wait_stream(_side_state(weight.device)[0])    # autograd adds this one to the first on this stream
                if not defer and second:
                    flush_wgrad_reduces()                              # ... and the first must be complete by then
                SIDE_STATS[1] += 1
            with torch.cuda.stream(side[0] if side is not None else main):
                if DETERMINISTIC_WGRAD:     # two-phase split-K through a lent workspace: no atomics
                    ws_bytes = _lib.lib().sln_conv_wgrad_workspace_bytes(N * OH * OW, Co, Ci, KH * KW, parts)
                    # deferred reduce: the partial sums live until the batched launch -- memory of their own
                    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=weight.device) if defer else \
                        ops._workspace(max(ws_bytes, 16), weight.device)
                e0 = _prof_begin()
                _lib.check(_lib.lib().sln_conv2d_wgrad_f32(
                    ops._ptr(gz), Co, gz.shape[2], ops._ptr(xp), N, H, W, Ci, xp.shape[2], parts, KH, KW,
                    stride[0], stride[1], dil[0], dil[1], pt, pl, OH, OW, ops._ptr(gw_t), ops._ptr(gzq),
                    ops._ptr(xq), ops._ptr(ws), ws_bytes, (1 if own_layout else 0) | (2 if defer else 0),
                    ops._stream()), "sln_conv2d_wgrad_f32")
                if defer:
                    # (the gradient's STORAGE is kept alive, not the tensor: a second reference to the tensor would
                    # make AccumulateGrad copy it -- before the reduce has run -- instead of adopting it)
                    rst[0].append((ws, gw_t.untyped_storage(), gw_t.data_ptr(), Co * KH * KW * Ci,
                                   _lib.lib().sln_conv_wgrad_ksplit(N * OH * OW, Co, Ci, KH * KW, parts),
                                   KH * KW if (own_layout and KH * KW > 1) else 0, Ci))
                    if len(rst[0]) >= 16:
                        flush_wgrad_reduces()
                wt_ = _lib.lib().sln_conv_wgrad_tile(N * OH * OW, Co, Ci, KH * KW, parts) if e0 is not None else 128
                _prof_end(e0, 2.0 * N * OH * OW * Co * KH * KW * Ci,          # (both events on the launch stream)
                          ("conv_wgrad256h_kernel" if (wt_ == 256 and parts == 2) else
                           ("conv_wgrad256_kernel<%d>" if wt_ == 256 else "conv_wgrad_kernel<%d>") % parts),
                          "wgrad N%d %dx%d C%d->%d k%d s%d d%d" % (N, H, W, Ci, Co, KH, stride[0], dil[0]),
                          _nbytes(gz, xp), _nbytes(gw_t))
            gw = gw_t if own_layout else gw_t.permute(0, 3, 1, 2)  # logical [Co,Ci,KH,KW]
        return gx, gw, g_bias, None, None, g_res, None, None, None, None, None, None, None, None, None, None, None


class _StemFn(torch.autograd.Function):
    """The 3-channel 7x7/2 input convolutions (modal/modals.py:311 C1, modal/resnet_deeplab.py conv1) on
    the HIP stack: sln_im2col_split_f32 writes the patch matrix [N*OH*OW, 160] (K = 7*7*3 = 147, zero
    padded) directly as operand parts, and the layer runs as a 1x1 convolution over those 160 channels --
    forward with the fused bias / frozen-BN / ReLU epilogue, weight gradient on the ordinary wgrad
    kernels (the image needs no data gradient).  Replaces the MIOpen calls, the separate bias / BN /
    ReLU passes and the NCHW -> NHWC copy of their output."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn_scale, bn_shift, relu, stride, pads, pool_handoff=None):
        parts = PARTS
        if PARTS_NOGRAD and not any(ctx.needs_input_grad):
            parts = PARTS_NOGRAD
        elif PARTS_FOR is not None:
            parts = PARTS_FOR(weight) or parts
        Co, Ci, KH, KW = weight.shape
        # the max-pool that is this output's ONLY reader leaves its incoming gradient here instead of scattering it
        # into a full-size map (nn_ops._MaxPoolFn.backward): the gradient preparation gathers from it
        ctx.pool_handoff = None
        if pool_handoff is not None and Co % 8 == 0 and not ctx.needs_input_grad[0] and \
                (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            pool_handoff.clear()
            pool_handoff["armed"] = True
            ctx.pool_handoff = pool_handoff
        xc = _nhwc(x.detach())
        N, _, H, W = xc.shape
        pt, pb, pl, pr = pads
        OH = (H + pt + pb - KH) // stride[0] + 1
        OW = (W + pl + pr - KW) // stride[1] + 1
        K = Ci * KH * KW
        Kp = (K + 31) // 32 * 32
        M = N * OH * OW
        xslot = _slot(weight, ("x", H, W)) if parts <= 2 else None
        xp = torch.empty((parts, M, Kp), dtype=torch.bfloat16, device=xc.device)

        def launch(dst):
            _lib.check(_lib.lib().sln_im2col_split_f32(
                ops._ptr(xc), N, H, W, Ci, KH, KW, stride[0], stride[1], pt, pl, OH, OW, Kp, parts,
                ops._ptr(dst), M, 0, *_q3(xslot), ops._stream()), "sln_im2col_split_f32")
        if xslot is not None and xslot.fresh:
            launch(None)
            xslot.book.settle(xslot)
        launch(xp)
        xq = xslot.scale if xslot is not None else None
        hit = getattr(weight, "_sln_stem_w", None)
        if hit is None or hit[0] != weight._version:      # [Co, (kh, kw, c)] zero padded, as a 1x1 kernel
            w2 = torch.zeros((Co, Kp), dtype=torch.float32, device=weight.device)
            w2[:, :K] = weight.detach().permute(0, 2, 3, 1).reshape(Co, K)
            hit = weight._sln_stem_w = (weight._version, w2.view(Co, Kp, 1, 1))
        w2 = hit[1]
        scale, shift = bn_scale, bn_shift
        if bias is not None:
            shift = bias * bn_scale + bn_shift if bn_scale is not None else bias
        if shift is not None:
            shift = shift.detach().contiguous()
        if scale is not None:
            scale = scale.detach().contiguous()
        y = _fwd(xp, N, OH, OW, wsrc(w2, parts, False, weight), Co, 1, 1, (1, 1), (1, 1), 0, 0, OH, OW, scale,
                 shift, None, relu, cin=K, xq=xq)
        if FORCE_RELU is not None and relu:
            _force_relu(weight, y)
        need_w = ctx.needs_input_grad[1]
        ctx.save_for_backward(xp if need_w else None, scale, y if relu else None, xq if need_w else None)
        ctx.cfg = (N, OH, OW, Co, Ci, KH, KW, K, Kp, parts, relu, bias is not None)
        ctx.geom = (H, W, stride, pt, pl)
        ctx.w2 = w2 if ctx.needs_input_grad[0] else None
        ctx.own = weight
        ctx.gzslot = _slot(weight, ("gz", OH, OW)) if parts <= 2 else None
        ctx.scale_epoch = SCALE_EPOCH[0]
        return y

    @staticmethod
    def backward(ctx, gy):
        xp, scale, y, xq = ctx.saved_tensors
        N, OH, OW, Co, Ci, KH, KW, K, Kp, parts, relu, has_bias = ctx.cfg
        _check_epoch(ctx, parts)
        need_w = ctx.needs_input_grad[1]
        want_bias = has_bias and ctx.needs_input_grad[2]
        gx = gw = g_bias = None
        need_x = ctx.needs_input_grad[0]
        pooled = ctx.pool_handoff.pop("pooled", None) if ctx.pool_handoff is not None else None
        if pooled is not None and gy.stride() != (0, 0, 0, 0):
            raise RuntimeError("stem: the pool handed its gradient over, but another gradient arrived as well -- "
                               "the stem output has a second reader")
        if need_w or want_bias or need_x:
            if pooled is not None:
                gz, _, g_bias = _grad_prep_pooled(pooled, y, scale, want_bias, parts, ctx.gzslot)
                POOL_HANDOFF_STATS[0] += 1
            else:
                gz, _, g_bias = _grad_prep(gy, y, scale, False, want_bias, parts, slot=ctx.gzslot)
            gzq = ctx.gzslot.scale if parts <= 2 else None
        if need_x:     # (module-level use only: the model's image carries no gradient)
            H, W, stride, pt, pl = ctx.geom
            gcols = _fwd(gz, N, OH, OW, wsrc(ctx.w2, parts, True, ctx.own), Kp, 1, 1, (1, 1), (1, 1), 0, 0, OH, OW,
                         None, None, None, False, cin=Co, xq=gzq)
            gx = torch.empty((N, H, W, Ci), dtype=torch.float32, device=gy.device).permute(0, 3, 1, 2)
            _lib.check(_lib.lib().sln_col2im_f32(ops._ptr(_nhwc(gcols)), N, H, W, Ci, KH, KW, stride[0], stride[1],
                                                 pt, pl, OH, OW, Kp, ops._ptr(gx), ops._stream()), "sln_col2im_f32")
        if need_w:
            gw_t = torch.empty((Co, 1, 1, Kp), dtype=torch.float32, device=gy.device)
            ws, ws_bytes = None, 0
            if DETERMINISTIC_WGRAD:
                ws_bytes = _lib.lib().sln_conv_wgrad_workspace_bytes(N * OH * OW, Co, Kp, 1, parts)
                ws = ops._workspace(max(ws_bytes, 16), gy.device)
            e0 = _prof_begin()
            _lib.check(_lib.lib().sln_conv2d_wgrad_f32(
                ops._ptr(gz), Co, gz.shape[2], ops._ptr(xp), N, OH, OW, Kp, Kp, parts, 1, 1, 1, 1, 1, 1, 0, 0,
                OH, OW, ops._ptr(gw_t), ops._ptr(gzq), ops._ptr(xq), ops._ptr(ws), ws_bytes, 0, ops._stream()),
                "sln_conv2d_wgrad_f32")
            _prof_end(e0, 2.0 * N * OH * OW * Co * K, "conv_wgrad_kernel<%d>" % parts,
                      "wgrad stem N%d %dx%d K%d->%d" % (N, OH, OW, K, Co), _nbytes(gz, xp), _nbytes(gw_t))
            gw = gw_t.view(Co, Kp)[:, :K].reshape(Co, KH, KW, Ci).permute(0, 3, 1, 2)
        return gx, gw, g_bias, None, None, None, None, None, None


# ------------------------------------------------------------------ grouped 3x3 on the fp16 matrix cores (PARTS = 1)
def _pack_grouped(weight, groups, flip):
    """-> (fragment-ordered scaled fp16 weights, scale tensor); cached on the parameter by version (csrc/grouped_conv.hip
    grouped_pack_weights_kernel).  flip: the data gradient's orientation."""
    cache = getattr(weight, "_sln_gpack", None)
    if cache is None:
        cache = weight._sln_gpack = {}
    hit = cache.get(flip)
    if hit is not None and hit[0] == weight._version:
        return hit[1], hit[2]
    w = weight.detach().contiguous()
    C = w.shape[0]
    n = _lib.lib().sln_grouped_conv3x3_packed_weight_elems(C, groups)
    if n <= 0:
        raise ValueError("grouped 3x3 on the fp16 path needs C %% 64 == 0 (C = %d, groups = %d)" % (C, groups))
    out = torch.empty((n,), dtype=torch.bfloat16, device=w.device)        # (16-bit containers, like every part)
    slot = _slot(weight, ("w",))

    def launch(dst):
        _lib.check(_lib.lib().sln_grouped_conv3x3_pack_weights_f16(ops._ptr(w), C, groups, 1 if flip else 0,
                                                                    ops._ptr(dst), *_q3(slot), ops._stream()),
                   "sln_grouped_conv3x3_pack_weights_f16")
    if slot.fresh:
        launch(None)
        slot.book.settle(slot)
    launch(out)
    cache[flip] = (weight._version, out, slot.scale)
    return out, slot.scale


class _GroupedF16Fn(torch.autograd.Function):
    """conv (groups, 3x3, padding 1, stride 1 / 2) -> frozen-BN affine -> ReLU on v_mfma_f32_16x16x32_f16 with single
    scaled fp16 operands (reference: nn.Conv2d(groups=32) + BN + ReLU, modal/resnext.py:36, 50-52, differentiated by
    autograd).  Forward: x's fp16 part (shared with the other readers of x) -> y fp32 + its fp16 part, or the part alone
    (parts_only: the reader is a convolution, the ReLU pattern is the part's sign).  Backward: the ordinary gradient
    preparation (ReLU mask, BN scale -> fp16 part), then the same kernel in its data-gradient mode and the MFMA weight
    gradient (per-range partial sums + ordered reduce: bit-reproducible)."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, relu, groups, stride, parts_only, chain_in=None, chain_out=None):
        N, C, H, W = x.shape
        OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
        xp, xq = act_parts(x, 1, owner=weight)
        wpk, wq = _pack_grouped(weight, groups, False)
        yslot, gzslot = _slot(weight, ("y", OH, OW)), _slot(weight, ("gz", OH, OW))
        sc = scale.detach().contiguous() if scale is not None else None
        sf = shift.detach().contiguous() if shift is not None else None
        fresh = yslot.fresh
        po = bool(parts_only and PARTS_ONLY_TRAIN and not fresh)
        dev = x.device
        y = None if po else torch.empty((N, OH, OW, C), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
        yp = None if fresh else torch.empty((1, N * OH * OW, C), dtype=torch.bfloat16, device=dev)
        e0 = _prof_begin()
        _lib.check(_lib.lib().sln_grouped_conv3x3_f16(
            ops._ptr(xp), N, H, W, C, groups, ops._ptr(wpk), stride, 0, ops._ptr(sc), ops._ptr(sf), 1 if relu else 0,
            ops._ptr(y), ops._ptr(yp), ops._ptr(xq), ops._ptr(wq), *_q3(yslot if yp is not None else None),
            None, None, ops._stream()), "sln_grouped_conv3x3_f16")
        _prof_end(e0, 2.0 * N * OH * OW * C * 9 * (C // groups), "grouped_mfma_kernel",
                  "fwd grouped N%d %dx%d C%d g%d s%d" % (N, H, W, C, groups, stride), _nbytes(xp, wpk), _nbytes(y, yp))
        if fresh:          # first use of the output's slot: exact amax pass over the fp32 output, then the split
            yp = _act_split(_nhwc(y), N * OH * OW, C, 1, yslot)
        if po:
            y = _placeholder((N, C, OH, OW), dev)
            y._sln_po = (yp, yslot.scale, SCALE_EPOCH[0])
            PO_STATS[0] += 1
        y._sln_parts = ((y._version, 1, SCALE_EPOCH[0]), yp, yslot.scale)
        need_w = ctx.needs_input_grad[1]
        # chained gradient preparation, as in _ConvFn.  PRODUCER side (chain_out: this output's ONLY reader is a
        # convolution -- the block's conv3): that reader's data-gradient epilogue writes this layer's prepared
        # gradient (ReLU mask, BN scale -> fp16 part) and hands it over; no fp32 gradient of this output, no
        # sln_conv_grad_prep_f32 launch.  From the second step on (the gradient's scale slot needs a history).
        ctx.chain_out = ctx.chain_in = None
        if CHAIN_GRAD_PREP and chain_out is not None and (ctx.needs_input_grad[0] or need_w) and not gzslot.fresh:
            chain_out.update(active=True, scale=sc, relu=bool(relu), parts=1, with_res=False, want_bias=False,
                             gz_slot=gzslot)
            chain_out.setdefault("readers", 1)
            ctx.chain_out = chain_out if chain_out["readers"] == 1 else None
            if ctx.chain_out is None:
                chain_out["active"] = False
        # READER side (chain_in: the layer below -- the block's conv1 -- has this convolution as its only reader):
        # the data-gradient kernel's epilogue prepares THAT layer's gradient (its ReLU pattern = the sign of the input
        # part this layer read, its BN scale as post_scale)
        if CHAIN_GRAD_PREP and chain_in is not None and chain_in.get("active") and ctx.needs_input_grad[0] and \
                chain_in["parts"] == 1 and not chain_in["with_res"] and chain_in.get("readers", 1) == 1 and \
                not chain_in.get("want_bias"):
            chain_in["consumer"] = chain_in.get("consumer", 0) + 1
            ctx.chain_in = chain_in
        mask_x = xp if (ctx.chain_in is not None and chain_in["relu"]) else None
        ctx.save_for_backward(xp if need_w else None, weight, sc, (yp if po else y) if relu else None,
                              xq if need_w else None, mask_x)
        ctx.cfg = (bool(relu), int(groups), int(stride), (N, C, H, W), (OH, OW))
        ctx.gzslot = gzslot
        ctx.scale_epoch = SCALE_EPOCH[0]
        return y

    @staticmethod
    def backward(ctx, gy):
        xp, weight, sc, y, xq, mask_x = ctx.saved_tensors
        relu, groups, stride, (N, C, H, W), (OH, OW) = ctx.cfg
        _check_epoch(ctx, 1)
        ch = ctx.chain_out
        if ch is not None and ch.get("consumer") and "gz" in ch:
            # the reader prepared this layer's gradient: what arrives must be its zero-stride placeholder -- anything
            # else means the output had a second consumer whose share the handed-over part does not contain
            if gy.stride() != (0, 0, 0, 0):
                raise RuntimeError("chained gradient (grouped 3x3): the output has a second consumer")
            gz, gzq = ch.pop("gz"), ch.pop("gzq", None)
            ch.pop("gbias", None)
            CHAIN_STATS[1] += 1
        else:
            if ch is not None and ch.get("consumer"):
                raise RuntimeError("chained gradient (grouped 3x3): the reader's data gradient did not run")
            gz, _, _ = _grad_prep(gy, y if relu else None, sc, False, False, 1, slot=ctx.gzslot)
            gzq = ctx.gzslot.scale
        gx = gw = None
        L = _lib.lib()
        if ctx.needs_input_grad[0]:
            wpk, wq = _pack_grouped(weight, groups, True)
            ci = ctx.chain_in
            e0 = _prof_begin()
            if ci is not None:
                # the layer below gets its PREPARED gradient from this kernel's epilogue; autograd gets a placeholder
                gz_up = torch.empty((1, N * H * W, C), dtype=torch.bfloat16, device=gz.device)
                _lib.check(L.sln_grouped_conv3x3_f16(ops._ptr(gz), N, H, W, C, groups, ops._ptr(wpk), stride, 1, None,
                                                     None, 0, None, ops._ptr(gz_up), ops._ptr(gzq), ops._ptr(wq),
                                                     *_q3(ci["gz_slot"]), ops._ptr(mask_x), ops._ptr(ci["scale"]),
                                                     ops._stream()), "sln_grouped_conv3x3_f16")
                ci["gz"], ci["gbias"], ci["gzq"] = gz_up, None, ci["gz_slot"].scale
                gx = _dummy_grad(gz.device).expand(N, C, H, W)
                CHAIN_STATS[0] += 1
                wrote = gz_up
            else:
                gxb = torch.empty((N, H, W, C), dtype=torch.float32, device=gz.device)
                _lib.check(L.sln_grouped_conv3x3_f16(ops._ptr(gz), N, H, W, C, groups, ops._ptr(wpk), stride, 1, None,
                                                     None, 0, ops._ptr(gxb), None, ops._ptr(gzq), ops._ptr(wq), None,
                                                     None, None, None, None, ops._stream()), "sln_grouped_conv3x3_f16")
                gx = gxb.permute(0, 3, 1, 2)
                wrote = gxb
            _prof_end(e0, 2.0 * N * OH * OW * C * 9 * (C // groups), "grouped_mfma_kernel",
                      "dgrad grouped N%d %dx%d C%d g%d s%d" % (N, H, W, C, groups, stride), _nbytes(gz, wpk), _nbytes(wrote))
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(weight)
            nbytes = L.sln_grouped_conv3x3_wgrad_workspace_bytes(N, H, W, C, groups, stride)
            ws = ops._workspace(nbytes, gz.device)
            e0 = _prof_begin()
            _lib.check(L.sln_grouped_conv3x3_wgrad_f16(ops._ptr(xp), ops._ptr(gz), N, H, W, C, groups, stride,
                                                       ops._ptr(gzq), ops._ptr(xq), ops._ptr(gw), ops._ptr(ws), nbytes,
                                                       ops._stream()), "sln_grouped_conv3x3_wgrad_f16")
            _prof_end(e0, 2.0 * N * OH * OW * C * 9 * (C // groups), "grouped_wgrad_mfma_kernel",
                      "wgrad grouped N%d %dx%d C%d g%d s%d" % (N, H, W, C, groups, stride), _nbytes(gz, xp), _nbytes(gw))
        return gx, gw, None, None, None, None, None, None, None, None


def grouped_supported(conv, x):
    """Grouped 3x3 layers the fp16 MFMA kernels take: PARTS = 1, C % 64 == 0, 4 ... 32 channels per group."""
    return (PARTS == 1 and x.dtype == torch.float32 and conv.in_channels == conv.out_channels and
            conv.in_channels % 64 == 0 and (conv.in_channels // conv.groups) in (4, 8, 16, 32))


def grouped_conv_bn_act(x, conv, scale, shift, relu, parts_only=False, chain_in=None, chain_out=None):
    return _GroupedF16Fn.apply(x, conv.weight, scale, shift, bool(relu), conv.groups, conv.stride[0], bool(parts_only),
                               chain_in, chain_out)


def is_stem(conv, x):
    """The small-Cin input convolutions _StemFn covers."""
    return (x.is_cuda and x.dtype == torch.float32 and conv.groups == 1 and tuple(conv.dilation) == (1, 1) and
            conv.in_channels * conv.kernel_size[0] * conv.kernel_size[1] <= 256 and conv.in_channels < 8 and
            conv.out_channels % 8 == 0)


POOL_HANDOFF_STATS = [0]      # stem gradient preparations that gathered from the pooled gradient


def stem_conv_bn_act(x, conv, bn, relu, pads, pool_handoff=None):
    from .nn_ops import bn_affine
    scale = shift = None
    if bn is not None:
        scale, shift = bn_affine(bn)
    return _StemFn.apply(x, conv.weight, conv.bias, scale, shift, bool(relu), tuple(conv.stride), tuple(pads),
                         pool_handoff)


_DUMMY = {}


def _dummy_grad(device):
    d = _DUMMY.get(device)
    if d is None:
        d = _DUMMY[device] = torch.zeros(1, dtype=torch.float32, device=device)
    return d


class GradInbox(dict):
    """Where another reader of a conv's input leaves ITS gradient w.r.t. that input during the backward pass, for
    the conv's data gradient to add in its epilogue (one fused pass instead of autograd's accumulation add: the
    FPN maps are read by the RPN's shared conv and by the RoIAlign crops; at P2 the add moves 3 GB).
    Protocol: the conv's forward arms the box; the other reader's backward calls offer(g): True = taken (it then
    returns no gradient for that input), False = the conv's backward has already run or will not run (not armed):
    return the gradient to autograd as usual.  A gradient that was taken and never consumed (the conv's backward did
    not run in this pass) is an error, raised when the pass ends."""
    pending = []          # boxes holding a gradient, checked at the end of the pass
    STATS = [0, 0]        # offered and taken / consumed

    def offer(self, g):
        if not self.get("armed") or self.get("closed") or "g" in self:
            return False
        self["g"] = g
        GradInbox.STATS[0] += 1
        if not GradInbox.pending:
            torch.autograd.Variable._execution_engine.queue_callback(GradInbox._check)
        GradInbox.pending.append(self)
        return True

    def take(self):
        self["closed"] = True
        g = self.pop("g", None)
        if g is not None:
            GradInbox.STATS[1] += 1
        return g

    @staticmethod
    def _check():
        left = [b for b in GradInbox.pending if "g" in b]
        del GradInbox.pending[:]
        for b in left:
            b.pop("g", None)
        if left:
            raise RuntimeError("GradInbox: %d deposited gradient(s) were never consumed -- the convolution that "
                               "should have added them took no part in this backward pass" % len(left))


def conv_bn_act(x, conv, bn, relu, residual, pads, weight=None, link=None, chain_in=None, chain_out=None,
                stride=None, pair=None, parts_only=False, grad_inbox=None):
    from .nn_ops import bn_affine
    scale = shift = None
    if bn is not None:
        scale, shift = bn_affine(bn)
    return _ConvFn.apply(x, conv.weight if weight is None else weight, conv.bias, scale, shift,
                         residual, bool(relu), tuple(stride or conv.stride), tuple(conv.dilation), tuple(pads),
                         link if LINK_SHORTCUT_GRAD else None, chain_in, chain_out, conv.weight, pair,
                         bool(parts_only), grad_inbox)
