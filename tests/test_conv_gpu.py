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


@pytest.fixture
def tile_mode():
    """SLN_CONV_TILE256 is read by the C launcher on every call: 0 = 128x128 tiles only,
    2 = the 256x256 LDS-DMA kernel for every forward / data-gradient launch."""
    import os
    saved = os.environ.get("SLN_CONV_TILE256")

    def set_mode(v):
        os.environ["SLN_CONV_TILE256"] = str(v)
    yield set_mode
    if saved is None:
        os.environ.pop("SLN_CONV_TILE256", None)
    else:
        os.environ["SLN_CONV_TILE256"] = saved


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("parts", [3, 2])
@pytest.mark.parametrize("tile", [128, 256])
def test_conv_forward_matches_fp64_reference(case, parts, tile, tile_mode):
    from sln_amodal_amd import conv_hip
    tile_mode(2 if tile == 256 else 0)
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
@pytest.mark.parametrize("tile", [128, 256])
def test_conv_backward_matches_autograd_of_unfused_ops(case, tile, tile_mode, monkeypatch):
    from sln_amodal_amd import conv_hip
    tile_mode(2 if tile == 256 else 0)                                   # data gradient (forward kernel)
    monkeypatch.setenv("SLN_WGRAD_TILE256", "2" if tile == 256 else "0")   # weight gradient
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


# ------------------------------------------------- multi-scale (one launch for all GLM scales)
@pytest.mark.parametrize("case", [
    # Cin, Cout, k, stride, dil, pad, relu, with_bn, with_res
    (64, 256, 1, 1, 1, 0, False, True, True),
    (72, 182, 3, 1, 6, 6, False, False, True),    # ASPP-like: bias, Cout not a multiple of 8, running sum
    (64, 64, 3, 1, 2, 2, True, True, False),
    (128, 96, 1, 2, 1, 0, True, True, False),     # strided 1x1 (layer3 block1)
])
def test_multiscale_conv_is_bit_identical_to_per_scale_launches(case):
    """sln_conv2d_fwd_ms_f32 over three image groups == three sln_conv2d_fwd_f32 launches,
    bit for bit (same products, same k order per output element)."""
    from sln_amodal_amd import conv_hip, nn_ops
    Cin, Cout, k, stride, dil, pad, relu, with_bn, with_res = case
    g = torch.Generator().manual_seed(Cin * 1000 + Cout)
    conv = nn.Conv2d(Cin, Cout, k, stride, pad, dil, bias=not with_bn).cuda()
    bn = nn.BatchNorm2d(Cout).cuda().eval() if with_bn else None
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.05)
        if conv.bias is not None:
            conv.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
        if bn is not None:
            bn.weight.copy_(torch.rand(Cout, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(Cout, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(Cout, generator=g) + 0.5)
    sizes = [(2, 33, 33), (2, 17, 17), (2, 25, 25)]     # 2 images per scale; tiles straddle the groups
    xs = [torch.randn(n, Cin, h, w, generator=g).cuda().contiguous(memory_format=torch.channels_last)
          for n, h, w in sizes]
    with torch.no_grad():
        outs_seq, res_seq = [], []
        for x in xs:
            r = None
            if with_res:
                oh = (x.shape[2] + 2 * pad - dil * (k - 1) - 1) // stride + 1
                r = torch.randn(x.shape[0], Cout, oh, oh, generator=g).cuda().contiguous(
                    memory_format=torch.channels_last)
            res_seq.append(r)
            outs_seq.append(nn_ops.conv_bn_act(x, conv, bn, relu=relu, residual=r))
        ms = conv_hip.MultiScale.pack(xs)
        res = conv_hip.MultiScale.pack(res_seq) if with_res else None
        out = nn_ops.conv_bn_act(ms, conv, bn, relu=relu, residual=res)
    assert isinstance(out, conv_hip.MultiScale)
    for a, b in zip(out.tensors(), outs_seq):
        assert a.shape == b.shape
        assert torch.equal(a, b)
    # the fused output parts feed the next layer: they must equal a fresh split of y
    fresh = conv_hip.MultiScale(out.segs, out.y).get_parts(out.parts.shape[0])
    assert torch.equal(out.parts[:, :, :Cout], fresh[:, :, :Cout])
    assert not out.parts[:, :, Cout:].any()


def test_msc_packed_forward_equals_sequential_scales():
    """The GLM wrapper with all scales packed per layer returns exactly what the
    reference-order loop over the scales returns (modal/msc_deeplab.py:29-45)."""
    from sln_amodal_amd.modal import msc_deeplab
    from sln_amodal_amd.modal.deeplabv2 import DeepLabV2
    from tests._util import key_init_
    net = msc_deeplab.MSC(DeepLabV2(n_classes=21, n_blocks=[1, 2, 2, 1], atrous_rates=[2, 4, 6, 8]),
                          scales=[0.5, 0.75]).cuda()
    key_init_(net)
    net.train()                      # returns every scale's logits as well as the max
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
    x = torch.randn(2, 3, 129, 129, generator=torch.Generator().manual_seed(3)).cuda()
    x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        msc_deeplab.PACK_SCALES = True
        try:
            packed = net(x)
            msc_deeplab.PACK_SCALES = False
            seq = net(x)
        finally:
            msc_deeplab.PACK_SCALES = True
    assert len(packed) == len(seq) == 4
    for a, b in zip(packed, seq):
        assert a.shape == b.shape and torch.equal(a, b)


def test_identity_shortcut_gradient_link_matches_autograd_accumulation():
    """Bottleneck blocks with identity shortcuts hand the shortcut's gradient to conv1's
    data-gradient epilogue (conv_hip link) instead of letting autograd add two full-size
    tensors.  One fp32 addition per element either way: dx must be bit-identical; weight
    gradients differ only by the atomics' summation order."""
    from sln_amodal_amd import conv_hip
    from sln_amodal_amd.modal.modals import Bottleneck
    from tests._util import key_init_
    down = nn.Sequential(nn.Conv2d(64, 128, kernel_size=1, stride=1), nn.BatchNorm2d(128, eps=0.001))
    net = nn.Sequential(Bottleneck(64, 32, 1, down), Bottleneck(128, 32), Bottleneck(128, 32)).cuda()
    key_init_(net)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
            m.weight.requires_grad = m.bias.requires_grad = False
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(2, 64, 24, 24, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    up = torch.randn(2, 128, 24, 24, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    res = {}
    saved = conv_hip.LINK_SHORTCUT_GRAD
    try:
        for mode in (True, False):
            conv_hip.LINK_SHORTCUT_GRAD = mode
            conv_hip.LINK_STATS[:] = [0, 0]
            x = x0.clone().requires_grad_(True)
            net.zero_grad(set_to_none=True)
            y = net(x)
            y.backward(up)
            # two identity blocks: both hand-overs happen and are consumed -- or none when off
            assert conv_hip.LINK_STATS == ([2, 2] if mode else [0, 0])
            res[mode] = (y.detach().clone(), x.grad.clone(),
                         {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    finally:
        conv_hip.LINK_SHORTCUT_GRAD = saved
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])
    assert res[True][2].keys() == res[False][2].keys() and len(res[True][2]) >= 20
    for k in res[True][2]:
        a, b = res[True][2][k], res[False][2][k]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()) + 1e-7), k


@pytest.mark.parametrize("knob", ["CHAIN_GRAD_PREP", "CHAIN_BLOCK_OUTPUT"])
def test_chained_gradient_preparation_matches_separate_grad_prep(knob):
    """The reader's data-gradient epilogue applies the producer's ReLU mask and BN scale and emits
    the bf16 parts (and bias sums) directly (sln_conv2d_fwd_ms_f32 with mask / post_scale /
    colsum / y == NULL) instead of writing the fp32 gradient and launching
    sln_conv_grad_prep_f32.  CHAIN_GRAD_PREP: all chaining (conv1 -> conv2 -> conv3 inside a
    block, and block output -> next identity block); CHAIN_BLOCK_OUTPUT: only the latter.
    Same products in the same order: the prepared parts, hence dx, are bit-identical; bias sums
    and weight gradients differ by summation order only.  Last block: 34 mid channels (pad
    channels in the parts, scalar epilogue path)."""
    from sln_amodal_amd import conv_hip
    from sln_amodal_amd.modal.modals import Bottleneck
    from tests._util import key_init_
    down1 = nn.Sequential(nn.Conv2d(64, 128, kernel_size=1, stride=1), nn.BatchNorm2d(128, eps=0.001))
    down4 = nn.Sequential(nn.Conv2d(128, 136, kernel_size=1, stride=1), nn.BatchNorm2d(136, eps=0.001))
    net = nn.Sequential(Bottleneck(64, 32, 1, down1), Bottleneck(128, 32), Bottleneck(128, 32),
                        Bottleneck(128, 34, 1, down4)).cuda()
    key_init_(net)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
            m.weight.requires_grad = m.bias.requires_grad = False
    g = torch.Generator().manual_seed(12)
    x0 = torch.randn(2, 64, 21, 27, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    up = torch.randn(2, 136, 21, 27, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    # 4 blocks x (conv1->conv2, conv2->conv3) = 8 inner chains; block1->2 and block2->3 = 2 outer ones
    expect = {("CHAIN_GRAD_PREP", True): 10, ("CHAIN_GRAD_PREP", False): 0,
              ("CHAIN_BLOCK_OUTPUT", True): 10, ("CHAIN_BLOCK_OUTPUT", False): 8}
    res = {}
    saved = getattr(conv_hip, knob)
    try:
        for mode in (True, False):
            setattr(conv_hip, knob, mode)
            conv_hip.CHAIN_STATS[:] = [0, 0]
            x = x0.clone().requires_grad_(True)
            net.zero_grad(set_to_none=True)
            y = net(x)
            y.backward(up)
            assert conv_hip.CHAIN_STATS == [expect[(knob, mode)]] * 2     # handed over == used
            res[mode] = (y.detach().clone(), x.grad.clone(),
                         {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
    finally:
        setattr(conv_hip, knob, saved)
    assert torch.equal(res[True][0], res[False][0])
    assert torch.equal(res[True][1], res[False][1])
    assert res[True][2].keys() == res[False][2].keys() and len(res[True][2]) >= 28
    for k in res[True][2]:
        a, b = res[True][2][k], res[False][2][k]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()) + 1e-7), k


def test_chained_block_output_with_a_second_reader_fails_loudly():
    """The hand-over is only valid while the next block is the only reader of a block output; a
    second reader makes autograd add its gradient, which the backward detects (no silent error)."""
    from sln_amodal_amd.modal.modals import Bottleneck
    from tests._util import key_init_
    net = nn.Sequential(Bottleneck(128, 32), Bottleneck(128, 32)).cuda()
    key_init_(net)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
            m.weight.requires_grad = m.bias.requires_grad = False
    x = torch.randn(1, 128, 9, 9).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    mid = net[0](x)
    out = net[1](mid)
    with pytest.raises(RuntimeError, match="second consumer"):
        (out.sum() + (mid * 2.0).sum()).backward()


@pytest.mark.parametrize("shape", [
    # N, H, W, Cin, Cout, k, dil : M and Cout not multiples of 256, K from 8 to 144 stages
    (3, 37, 41, 64, 256, 3, 1), (2, 65, 65, 1024, 200, 1, 1), (1, 49, 49, 136, 439, 3, 2), (5, 16, 16, 8, 256, 1, 1)])
def test_tile256_kernel_is_deterministic_and_agrees_with_tile128(shape, tile_mode):
    """Race screen for the hand-synchronised DMA pipeline of conv_fwd256_kernel (counted vmcnt +
    raw barrier, three stages): 40 launches of the same problem must give bit-identical
    outputs (a race shows up as run-to-run differences), and they agree with the 128x128
    kernel to fp32 accumulation-order noise (the two walk K in different orders)."""
    from sln_amodal_amd import conv_hip
    N, H, W, Cin, Cout, k, dil = shape
    g = torch.Generator(device="cuda").manual_seed(H * 131 + Cin)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
    pad = dil * (k - 1) // 2
    xp = conv_hip.act_parts(x, 3)
    wp = conv_hip._split_weights(w, parts=3)

    def run():
        return conv_hip._fwd(xp, N, H, W, wp, Cout, k, k, (1, 1), (dil, dil), pad, pad, H, W, None, None,
                             None, False, cin=Cin, out_parts=True)

    tile_mode(0)
    y128 = run().clone()
    tile_mode(2)
    first = run()
    y0, p0 = first.clone(), first._sln_parts[1].clone()
    for _ in range(40):
        y = run()
        assert torch.equal(y, y0) and torch.equal(y._sln_parts[1], p0)
    err = (y0 - y128).abs().max().item() / y128.abs().max().item()
    assert err < 2e-6, err


def test_two_reader_chain_of_the_rpn_heads_matches_autograd_accumulation():
    """RPN: the shared 3x3 map is read by the class and the box head.  Whichever data gradient runs
    second adds the first (as the epilogue's residual), applies the shared conv's ReLU mask and hands it
    its prepared gradient; both heads return None to autograd.  One fp32 addition per element either
    way: the gradient reaching the RPN's input is bit-identical, bias / weight gradients differ by
    summation order only."""
    from sln_amodal_amd import conv_hip
    from sln_amodal_amd.modal.modals import RPN
    from tests._util import key_init_
    rpn = RPN(3, 1, 64).cuda()
    key_init_(rpn)
    g = torch.Generator().manual_seed(21)
    x0 = torch.randn(2, 64, 19, 23, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    up_l = torch.randn(2, 19 * 23 * 3, 2, generator=g).cuda()
    up_b = torch.randn(2, 19 * 23 * 3, 4, generator=g).cuda()
    res = {}
    saved = conv_hip.CHAIN_GRAD_PREP
    try:
        for mode in (True, False):
            conv_hip.CHAIN_GRAD_PREP = mode
            conv_hip.CHAIN_STATS[:] = [0, 0]
            x = x0.clone().requires_grad_(True)
            rpn.zero_grad(set_to_none=True)
            logits, probs, bbox = rpn(x)
            ((logits * up_l).sum() + (bbox * up_b).sum()).backward()
            assert conv_hip.CHAIN_STATS == ([1, 1] if mode else [0, 0])
            res[mode] = (logits.detach().clone(), bbox.detach().clone(), x.grad.clone(),
                         {k: p.grad.clone() for k, p in rpn.named_parameters()})
    finally:
        conv_hip.CHAIN_GRAD_PREP = saved
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert torch.equal(res[True][2], res[False][2])
    assert set(res[True][3]) == set(res[False][3]) and len(res[True][3]) == 6
    for k in res[True][3]:
        a, b = res[True][3][k], res[False][3][k]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()) + 1e-7), k
