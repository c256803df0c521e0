"""GPU tests of the fused clip + SGD step (csrc/optim.hip) against the eager sequence the reference
runs (torch.nn.utils.clip_grad_norm_ -> torch.optim.SGD, model.py:352-358, 441-444) in float64 and
float32.  The reference-generated optimizer fixture is checked through the model in
tests/test_e2e_gpu.py::test_optimizer_step_matches_reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 3, 7, 7), (256, 64, 1, 1), (1,), (3,), (1023,), (65536,), (65537,), (7, 5, 3), (256, 256, 3, 3),
          (1024, 12544), (2,), (81, 1024)]


def _make(seed, scale):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.randn(s, generator=g) * 0.05 for s in SHAPES]
    gs = [[torch.randn(s, generator=g) * scale * (1 + i) for s in SHAPES] for i in range(3)]
    return ps, gs


def _eager(ps, gs, wds, lr, momentum, max_norm, dtype):
    """clip_grad_norm (torch 0.4 semantics: coef = max/(norm+1e-6) applied when < 1) + SGD, on the host."""
    ps = [p.to(dtype).clone() for p in ps]
    bufs = [None] * len(ps)
    norms = []
    for rnd in gs:
        grads = [None if g is None else g.to(dtype).clone() for g in rnd]
        total = sum(float(g.double().pow(2).sum()) for g in grads if g is not None) ** 0.5
        norms.append(total)
        coef = max_norm / (total + 1e-6)
        for i, g in enumerate(grads):
            if g is None:
                continue
            if coef < 1:
                g = g * torch.tensor(coef, dtype=torch.float32).to(dtype)
            d = g + wds[i] * ps[i]
            bufs[i] = d.clone() if bufs[i] is None else bufs[i] * momentum + d
            ps[i] = ps[i] - lr * bufs[i]
    return ps, norms


@pytest.mark.parametrize("scale,clips", [(1e-4, False), (0.05, True)])
def test_fused_step_matches_eager_sequence(scale, clips):
    from sln_amodal_amd.optim import ClippedSGD
    ps, gs = _make(3, scale)
    gs[1][2] = None                                  # a parameter without a gradient this step
    gs[1][5] = None
    wds = [1e-4 if i % 3 else 0.0 for i in range(len(ps))]
    dev = [torch.nn.Parameter(p.cuda()) for p in ps]
    dev[8] = torch.nn.Parameter(ps[8].cuda().contiguous(memory_format=torch.channels_last))
    opt = ClippedSGD([{"params": [dev[i] for i in range(len(ps)) if wds[i]], "weight_decay": 1e-4},
                      {"params": [dev[i] for i in range(len(ps)) if not wds[i]]}], lr=0.01, momentum=0.9)
    want64, norms = _eager(ps, gs, wds, 0.01, 0.9, 5.0, torch.float64)
    want32, _ = _eager(ps, gs, wds, 0.01, 0.9, 5.0, torch.float32)
    versions = [p._version for p in dev]
    for r, rnd in enumerate(gs):
        opt.zero_grad()
        for p, g in zip(dev, rnd):
            p.grad = None if g is None else g.cuda()
        got = opt.step(5.0)
        assert abs(float(got) - norms[r]) <= 2e-7 * norms[r]
        assert (norms[r] > 5.0) == clips
    assert all(p._version > v for p, v in zip(dev, versions))      # weight-part caches must see the update
    for i, p in enumerate(dev):
        got = p.detach().cpu()
        assert got.shape == ps[i].shape
        step = (want64[i] - ps[i].double()).abs().max().item()
        # float32 rounding of the update: a few ulps of the weights, far below the step itself
        assert (got.double() - want64[i]).abs().max().item() <= 2e-7 * ps[i].abs().max().item() + 1e-3 * step, i
        assert (got - want32[i]).abs().max().item() <= 2.5e-7 * ps[i].abs().max().item(), i


def test_fused_step_is_reproducible_and_skips_clipping_below_the_limit():
    from sln_amodal_amd.optim import ClippedSGD
    outs = []
    for _ in range(2):
        ps, gs = _make(11, 1e-3)
        dev = [torch.nn.Parameter(p.cuda()) for p in ps]
        opt = ClippedSGD([{"params": dev, "weight_decay": 0.0}], lr=0.5, momentum=0.0)
        for p, g in zip(dev, gs[0]):
            p.grad = g.cuda()
        n = opt.step(5.0)
        outs.append(([p.detach().clone() for p in dev], float(n)))
        # below the limit the gradient passes unscaled: p - lr * g exactly
        for p0, g, p in zip(ps, gs[0], dev):
            assert torch.equal(p.detach().cpu(), p0 - 0.5 * g)
    assert outs[0][1] == outs[1][1]
    assert all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))


def test_host_tensors_are_refused():
    from sln_amodal_amd.optim import ClippedSGD
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        ClippedSGD([{"params": [p]}], lr=0.1).step(5.0)


def test_non_finite_gradient_skips_the_step_on_the_device():
    """An inf / NaN gradient: no tensor moves, no momentum buffer changes, the device counter says so
    (the reference's loop `continue`s past unusable batches, model.py:416-418, 433-434)."""
    from sln_amodal_amd.optim import ClippedSGD
    ps, gs = _make(5, 1e-3)
    dev = [torch.nn.Parameter(p.cuda()) for p in ps]
    opt = ClippedSGD([{"params": dev, "weight_decay": 1e-4}], lr=0.1, momentum=0.9)
    for p, g in zip(dev, gs[0]):
        p.grad = g.cuda()
    opt.step(5.0)
    before = [p.detach().clone() for p in dev]
    bufs = [opt.state[p].clone() for p in dev]
    for bad in (float("nan"), float("inf")):
        for p, g in zip(dev, gs[1]):
            p.grad = g.cuda()
        dev[3].grad[1] = bad
        n = opt.step(5.0)
        assert not np.isfinite(float(n))
        assert all(torch.equal(a, p.detach()) for a, p in zip(before, dev))
        assert all(torch.equal(b, opt.state[p]) for b, p in zip(bufs, dev))
    assert opt.skipped_steps() == 2
    for p, g in zip(dev, gs[2]):                       # the next clean step goes through
        p.grad = g.cuda()
    opt.step(5.0)
    assert opt.skipped_steps() == 2
    assert not any(torch.equal(a, p.detach()) for a, p in zip(before, dev))


def test_param_group_hyperparameters_are_read_at_step_time_and_state_round_trips():
    from sln_amodal_amd.optim import ClippedSGD
    ps, gs = _make(9, 1e-3)

    def run(change_lr, reload_at=None):
        dev = [torch.nn.Parameter(p.cuda()) for p in ps]
        opt = ClippedSGD([{"params": dev[:5], "weight_decay": 1e-4}, {"params": dev[5:]}], lr=0.1, momentum=0.9)
        for r in range(3):
            if r == 1 and change_lr:
                for g in opt.param_groups:         # what an LR scheduler does
                    g["lr"] = 0.01
            if r == reload_at:
                sd = opt.state_dict()
                opt = ClippedSGD([{"params": dev[:5], "weight_decay": 1e-4}, {"params": dev[5:]}], lr=0.5, momentum=0.0)
                opt.load_state_dict(sd)
            for p, g in zip(dev, gs[r]):
                p.grad = g.cuda()
            opt.step(5.0)
        return [p.detach().clone() for p in dev], opt

    base, _ = run(False)
    sched, opt = run(True)
    assert opt.lr == 0.01
    assert not any(torch.equal(a, b) for a, b in zip(base, sched))
    # (a change of lr keeps the momentum buffers: compare with the same run reloaded from its state dict)
    again, _ = run(True, reload_at=2)
    assert all(torch.equal(a, b) for a, b in zip(sched, again))
    opt.param_groups[0]["lr"] = 0.3
    with pytest.raises(ValueError):
        opt.step(5.0)


def test_a_vetoed_step_is_not_applied_and_is_counted():
    """ClippedSGD.step(max_norm, veto=...): a veto > 0 (an fp16 operand block clamped somewhere in the step,
    conv_hip.clamp_veto()) leaves weights AND momentum buffers untouched through the device-side guard, counts the step
    in skipped_steps() and skipped_clamped_steps(), keeps the real gradient norm in last_norm; a veto of 0 is the
    ordinary step, bit for bit."""
    from sln_amodal_amd.optim import ClippedSGD
    ps, gs = _make(5, 0.05)
    a = [torch.nn.Parameter(p.cuda()) for p in ps]
    b = [torch.nn.Parameter(p.cuda()) for p in ps]
    oa = ClippedSGD([{"params": a, "weight_decay": 1e-4}], lr=0.01, momentum=0.9)
    ob = ClippedSGD([{"params": b, "weight_decay": 1e-4}], lr=0.01, momentum=0.9)
    zero, one = torch.zeros(1, device="cuda"), torch.ones(1, device="cuda")
    for r, rnd in enumerate(gs):
        for p, q, g in zip(a, b, rnd):
            p.grad, q.grad = g.cuda(), g.cuda()
        na = oa.step(5.0)                    # no veto argument
        nb = ob.step(5.0, veto=zero)         # an explicit "nothing clamped"
        assert float(na) == float(nb)
        assert all(torch.equal(p, q) for p, q in zip(a, b))
    before = [p.detach().clone() for p in b]
    bufs = [ob.state[p].clone() for p in b]
    v0 = [p._version for p in b]
    for p, g in zip(b, gs[0]):
        p.grad = g.cuda()
    norm = ob.step(5.0, veto=one)
    assert all(torch.equal(p, q) for p, q in zip(b, before))
    assert all(torch.equal(ob.state[p], q) for p, q in zip(b, bufs))
    assert float(norm) > 0 and np.isfinite(float(norm))
    assert ob.skipped_steps() == 1 and ob.skipped_clamped_steps() == 1 and oa.skipped_steps() == 0
    # a non-finite step that is ALSO vetoed counts as non-finite only
    b[0].grad[0] = float("nan")
    ob.step(5.0, veto=one)
    assert ob.skipped_steps() == 2 and ob.skipped_clamped_steps() == 1
    assert all(torch.equal(p, q) for p, q in zip(b, before))
    # ... and the next clean step applies
    for p, g in zip(b, gs[1]):
        p.grad = g.cuda()
    ob.step(5.0, veto=zero)
    assert not any(torch.equal(p, q) for p, q in zip(b, before) if p.numel() > 1)
    assert ob.skipped_steps() == 2
