"""Clip + SGD for the train step, one pass over device-resident tables instead of ~3000 small aten
launches: the reference's torch.nn.utils.clip_grad_norm(params, 5.0) followed by
torch.optim.SGD(momentum, weight decay on the non-'bn' group).step() (model.py:352-358, 441-444)."""
import ctypes as C

import numpy as np
import torch

from . import _lib

CHUNK = 1 << 16


class ClippedSGD(object):
    """param_groups like torch.optim.SGD's ([{'params': [...], 'weight_decay': w}, ...]).  step(max_norm)
    clips over every parameter that holds a gradient and applies momentum SGD; the total gradient norm
    is left in `last_norm` (0-d float32 device tensor, no host sync)."""

    def __init__(self, param_groups, lr, momentum=0.0):
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g["params"] = list(g["params"])
            g.setdefault("weight_decay", 0.0)
            g.setdefault("lr", lr)
            g.setdefault("momentum", momentum)
            self.param_groups.append(g)
        self.state = {}                      # param -> momentum buffer
        self.skipped = None                  # device int32: steps skipped (non-finite gradient norm OR a clamp veto)
        self.skipped_clamped = None          # device int32: of those, the steps vetoed for clamped operand blocks
        self._plan_key, self._plan = None, None
        self._ring, self._turn = [], 0       # pinned staging buffers of the pointer tables
        self.capturing = False               # inside a HIP-graph capture: no event waits (the tables are static)
        self.last_norm = None

    def _hyper(self):
        """lr / momentum are read from the groups at step time, like torch.optim.SGD (an LR scheduler or
        `optimizer.param_groups[i]['lr'] = ...` takes effect); the fused kernel applies ONE pair to every
        tensor, so the groups must agree."""
        lrs = {float(g["lr"]) for g in self.param_groups if g["params"]}
        moms = {float(g["momentum"]) for g in self.param_groups if g["params"]}
        if len(lrs) > 1 or len(moms) > 1:
            raise ValueError("ClippedSGD applies one lr / momentum to all parameter groups (got lr %s, "
                             "momentum %s)" % (sorted(lrs), sorted(moms)))
        return (lrs.pop() if lrs else 0.0), (moms.pop() if moms else 0.0)

    @property
    def lr(self):
        return self._hyper()[0]

    @lr.setter
    def lr(self, value):
        for g in self.param_groups:
            g["lr"] = float(value)

    @property
    def momentum(self):
        return self._hyper()[1]

    def skipped_steps(self):
        """Steps whose update was skipped on the device because the gradient norm was inf / NaN (host sync)."""
        return (int(self.skipped.item()) if self.skipped is not None else 0) + getattr(self, "_skipped_loaded", 0)

    def skipped_clamped_steps(self):
        """Of skipped_steps(): the steps vetoed because an fp16 operand block had to clamp in them (host sync)."""
        return int(self.skipped_clamped.item()) if self.skipped_clamped is not None else 0

    def state_dict(self):
        """Momentum buffers by position in the flattened parameter groups + the groups' hyper-parameters
        (the layout of torch.optim.SGD.state_dict)."""
        flat = [p for g in self.param_groups for p in g["params"]]
        idx = {id(p): i for i, p in enumerate(flat)}
        groups, k = [], 0
        for g in self.param_groups:
            d = {key: v for key, v in g.items() if key != "params"}
            d["params"] = list(range(k, k + len(g["params"])))
            k += len(g["params"])
            groups.append(d)
        return {"state": {idx[id(p)]: {"momentum_buffer": b} for p, b in self.state.items() if id(p) in idx},
                "param_groups": groups, "skipped_steps": self.skipped_steps()}

    def load_state_dict(self, sd):
        flat = [p for g in self.param_groups for p in g["params"]]
        if len(sd["param_groups"]) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        for g, d in zip(self.param_groups, sd["param_groups"]):
            if len(d["params"]) != len(g["params"]):
                raise ValueError("loaded state dict contains a parameter group of a different size")
            g.update({key: v for key, v in d.items() if key != "params"})
        self.state = {}
        self._skipped_loaded = int(sd.get("skipped_steps", 0))     # (added to the device counter's value)
        with torch.no_grad():
            for i, st in sd["state"].items():
                p = flat[int(i)]
                b = st.get("momentum_buffer")
                if b is not None:
                    self.state[p] = torch.zeros_like(p, memory_format=torch.preserve_format).copy_(b)

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    def _active(self):
        out = []
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    out.append((p, float(g["weight_decay"])))
        return out

    def _chunks(self, numels, wds, device):
        key = (tuple(numels), tuple(wds))
        if key != self._plan_key:
            ct, co = [], []
            for i, n in enumerate(numels):
                for off in range(0, n, CHUNK):
                    ct.append(i)
                    co.append(off)
            self._plan = (torch.tensor(ct, dtype=torch.int32, device=device),
                          torch.tensor(co, dtype=torch.int64, device=device),
                          torch.tensor(numels, dtype=torch.int64, device=device),
                          torch.empty(max(len(ct), 1), dtype=torch.float64, device=device),
                          torch.tensor(wds, dtype=torch.float32, device=device))
            self._plan_key = key
        return self._plan

    @torch.no_grad()
    def step(self, max_norm, veto=None):
        """veto: float32 [1] device tensor or None; > 0 = do not apply this step (conv_hip.clamp_veto(), or the
        all-reduced one of parallel.GradientAllReducer): the update is skipped on the device through the same guard
        as a non-finite gradient norm and counted in `skipped` and `skipped_clamped`; `last_norm` stays the real
        norm.  No host sync either way."""
        act = self._active()
        if not act:
            return None
        dev = act[0][0].device
        if dev.type != "cuda":
            raise RuntimeError("ClippedSGD runs on the GPU only (no CPU fallback in sln_amodal_amd)")
        ps, gs, bs = [], [], []
        for p, _ in act:
            g = p.grad
            if p.dtype != torch.float32 or g.dtype != torch.float32:
                raise TypeError("ClippedSGD needs float32 parameters and gradients")
            if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                raise ValueError("parameters must be dense")
            if g.stride() != p.stride():       # element i of the gradient must be element i of the weight
                g = torch.empty_like(p).copy_(g)
            b = self.state.get(p)
            if b is None:
                b = self.state[p] = torch.zeros_like(p, memory_format=torch.preserve_format)
            ps.append(p); gs.append(g); bs.append(b)
        n = len(ps)
        ct, co, numel, partial, wd = self._chunks([p.numel() for p in ps], [w for _, w in act], dev)
        # Pointer tables: gradients are fresh allocations every step, so the tables are re-sent each
        # time -- from pinned staging buffers and asynchronously, so the host never waits for the GPU
        # here (a pageable copy would drain the stream once per step).  A staging buffer is reused
        # only after the copy that read it has completed (4 in flight).
        if not self._ring or self._ring[0][0].shape[1] != n:
            self._ring = [[torch.empty((3, n), dtype=torch.int64, pin_memory=True), None] for _ in range(4)]
        stage = self._ring[self._turn % len(self._ring)]
        self._turn += 1
        if stage[1] is not None and not self.capturing:
            stage[1].synchronize()
        host = stage[0].numpy()
        host[0] = [p.data_ptr() for p in ps]
        host[1] = [g.data_ptr() for g in gs]
        host[2] = [b.data_ptr() for b in bs]
        tab = torch.empty((3, n), dtype=torch.int64, device=dev)
        tab.copy_(stage[0], non_blocking=True)
        if not self.capturing:
            stage[1] = torch.cuda.Event()
            stage[1].record()
        sq = torch.empty(1, dtype=torch.float64, device=dev)
        if self.skipped is None:
            self.skipped = torch.zeros(1, dtype=torch.int32, device=dev)
        lr, momentum = self._hyper()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L = _lib.lib()
        vp = lambda t: C.c_void_p(t.data_ptr())
        nch = ct.numel() if sum(p.numel() for p in ps) else 0
        _lib.check(L.sln_grad_sqnorm_f32(vp(tab[1]), vp(numel), vp(ct), vp(co), nch, CHUNK, vp(partial), vp(sq),
                                         st), "sln_grad_sqnorm_f32")
        sq_guard = sq
        if veto is not None:
            # the guard's input: +inf instead of the squared norm when the step is vetoed (two 5-us elementwise ops)
            if self.skipped_clamped is None:
                self.skipped_clamped = torch.zeros(1, dtype=torch.int32, device=dev)
            hit = veto.reshape(1) > 0
            sq_guard = torch.where(hit, torch.full_like(sq, float("inf")), sq)
            # (a step that is non-finite anyway is counted as that, not as clamped)
            self.skipped_clamped += (hit & torch.isfinite(sq)).to(torch.int32)
        _lib.check(L.sln_sgd_clip_step_f32(vp(tab[0]), vp(tab[1]), vp(tab[2]), vp(numel), vp(wd), vp(ct), vp(co),
                                           nch, CHUNK, vp(sq_guard), float(max_norm), lr, momentum, vp(self.skipped),
                                           st), "sln_sgd_clip_step_f32")
        # caches keyed by the weights' version counters (split weight parts) must see the update
        torch._C._autograd._unsafe_set_version_counter(ps, [p._version + 1 for p in ps])
        self._keep = (tab, gs, sq_guard)      # alive until the next step: the launches above are asynchronous
        self.last_norm = sq.sqrt().to(torch.float32)[0]
        return self.last_norm
