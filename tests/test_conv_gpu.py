"""GPU numerics of the split-bf16 implicit-GEMM convolution (csrc/conv.hip) against
a plain PyTorch reference of the same op (fp64 accumulate), forward and backward.
Tolerances: 3-part split (default) is fp32-class: 5e-6 of the output scale;
2-part split: 3e-5."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # Cin, Cout, k, stride, dil, pad(t,b,l,r), H, W, N
    (64, 64, 1, 1, 1, (0, 0, 0, 0), 33, 29, 2),
    (64, 256, 3, 1, 1, (1, 1, 1, 1), 40, 40, 2),
    (256, 128, 1, 2, 1, (0, 0, 0, 0), 32, 32, 2),
    (128, 128, 3, 1, 2, (2, 2, 2, 2), 31, 31, 1),
    (256, 182, 3, 1, 12, (12, 12, 12, 12), 33, 33, 1),
    (440, 256, 3, 1, 1, (1, 1, 1, 1), 16, 16, 3),
    (256, 12, 1, 1, 1, (0, 0, 0, 0), 24, 24, 2),
    (64, 64, 3, 2, 1, (0, 1, 0, 1), 32, 32, 2),      # asymmetric SAME padding
    (256, 1024, 7, 1, 1, (0, 0, 0, 0), 7, 7, 5),     # classifier "FC" conv
    (8, 8, 3, 1, 1, (1, 1, 1, 1), 5, 5, 1),
    (3, 64, 7, 2, 1, (3, 3, 3, 3), 64, 64, 2),        # stem: Cin padded 3 -> 8
    (512, 6, 1, 1, 1, (0, 0, 0, 0), 20, 20, 2),       # RPN class head: Cout 6
    (439, 256, 3, 1, 1, (1, 1, 1, 1), 16, 16, 2),     # mask conv1: ragged Cin
]


def _ref(x, w, b, scale, shift, res, relu, stride, dil, pads):
    pt, pb, pl, pr = pads
    xd = F.pad(x.double(), (pl, pr, pt, pb))
    y = F.conv2d(xd, w.double(), None if b is None else b.double(), stride, 0, dil)
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if res is not None:
        y = y + res.double()
    return F.relu(y) if relu else y


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("parts", [3, 2])
def test_conv_forward_matches_fp64_reference(case, parts):
    from sln_amodal_amd import conv_hip
    Cin, Cout, k, stride, dil, pads, H, W, N = case
    g = torch.Generator(device="cuda").manual_seed(Cin * 7 + Cout)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, device="cuda", generator=g)
    scale = torch.rand(Cout, device="cuda", generator=g) + 0.5
    shift = torch.randn(Cout, device="cuda", generator=g)
    ref0 = _ref(x, w, b, scale, shift, None, False, stride, dil, pads)
    res = torch.randn(ref0.shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    old = conv_hip.PARTS
    conv_hip.PARTS = parts
    try:
        for (use_b, use_bn, use_res, relu) in [(True, True, True, True), (False, False, False, False),
                                               (True, False, False, True)]:
            y = conv_hip._ConvFn.apply(x, w, b if use_b else None, scale if use_bn else None,
                                       shift if use_bn else None, res if use_res else None, relu,
                                       (stride, stride), (dil, dil), pads)
            ref = _ref(x, w, b if use_b else None, scale if use_bn else None, shift if use_bn else None,
                       res if use_res else None, relu, stride, dil, pads)
            assert y.shape == ref.shape
            assert y.is_contiguous(memory_format=torch.channels_last)
            tol = 5e-6 if parts == 3 else 3e-5
            err = (y.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
            assert err < tol, (case, parts, err)
    finally:
        conv_hip.PARTS = old


@pytest.mark.parametrize("case", [CASES[1], CASES[2], CASES[3], CASES[5], CASES[8], CASES[11], CASES[12]])
def test_conv_backward_matches_autograd_of_unfused_ops(case):
    from sln_amodal_amd import conv_hip
    Cin, Cout, k, stride, dil, pads, H, W, N = case
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5)
    b = torch.randn(Cout, device="cuda", generator=g)
    scale = torch.rand(Cout, device="cuda", generator=g) + 0.5
    shift = torch.randn(Cout, device="cuda", generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, w, b)]
    ref0 = _ref(x, w, b, scale, shift, None, True, stride, dil, pads)
    res = torch.randn(ref0.shape, device="cuda", generator=g)
    res_l = res.clone().requires_grad_(True)
    y = conv_hip._ConvFn.apply(leaves[0], leaves[1], leaves[2], scale, shift, res_l, True,
                               (stride, stride), (dil, dil), pads)
    up = torch.randn(y.shape, device="cuda", generator=g)
    y.backward(up)
    rl = [t.clone().double().requires_grad_(True) for t in (x, w, b)]
    rres = res.clone().double().requires_grad_(True)
    pt, pb, pl, pr = pads
    yr = F.conv2d(F.pad(rl[0], (pl, pr, pt, pb)), rl[1], rl[2], stride, 0, dil)
    yr = yr * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1) + rres
    # ReLU with the kernel's own activity mask: an output within rounding of zero may
    # legitimately land on either side, which would flip that element's gradient
    assert ((F.relu(yr) - y.detach().double()).abs().max() < 1e-4)
    yr = yr * (y.detach() > 0)
    yr.backward(up.double())
    for got, want, name in zip([l.grad for l in leaves] + [res_l.grad], [l.grad for l in rl] + [rres.grad],
                               ["x", "w", "b", "res"]):
        err = (got.double() - want).abs().max().item() / max(want.abs().max().item(), 1e-9)
        assert err < 2e-5, (name, err)


def test_module_dispatch_uses_hip_backend():
    """nn_ops.conv_bn_act on a Bottleneck-shaped block: HIP backend vs torch backend."""
    from sln_amodal_amd import nn_ops
    from sln_amodal_amd.modal.modals import Bottleneck
    from tests._util import key_init_
    torch.manual_seed(0)
    blk = Bottleneck(256, 64).cuda()
    key_init_(blk)
    for p in blk.parameters():
        p.requires_grad_(p.dim() != 1 or True)
    x = torch.randn(2, 256, 24, 24, device="cuda").contiguous(memory_format=torch.channels_last)
    outs = {}
    for be in ("hip", "torch"):
        nn_ops.BACKEND = be
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        y.square().mean().backward()
        outs[be] = (y.detach(), xi.grad.clone(), blk.conv2.weight.grad.clone())
        blk.zero_grad()
    nn_ops.BACKEND = "auto"
    for a, b in zip(outs["hip"], outs["torch"]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * b.abs().max().item())
