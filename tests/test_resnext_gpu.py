"""BASELINE.json configs[4] on the GPU: the grouped 3x3 HIP kernels (forward, data / weight gradient) against aten, the ResNeXt encoder + multi-scale
ASPP heads against the fixtures of the reference's own classes, and a full-depth forward at a config-sized shape."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_resnext_cpu import check

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip_backend():
    from sln_amodal_amd import nn_ops
    old = nn_ops.BACKEND
    nn_ops.BACKEND = "hip"       # raise instead of silently falling back
    yield
    nn_ops.BACKEND = old


@pytest.mark.parametrize("cg", [4, 8, 16, 32])
@pytest.mark.parametrize("stride", [1, 2])
def test_grouped_conv3x3_matches_aten(cg, stride):
    """csrc/grouped_conv.hip (reference modal/resnext.py:36: groups 32, padding 1, no bias) with the fused BN affine
    and ReLU, odd sizes, against F.conv2d in fp64: relative error of an fp32 accumulation over K = 9 * cg."""
    from sln_amodal_amd import ops
    G = 32
    C = G * cg
    g = torch.Generator(device="cuda").manual_seed(cg + stride)
    x = torch.randn(2, C, 37, 29, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, cg, 3, 3, device="cuda", generator=g) / (9 * cg) ** 0.5
    scale = torch.rand(C, device="cuda", generator=g) + 0.5
    shift = torch.randn(C, device="cuda", generator=g) * 0.1
    want = F.relu(F.conv2d(x.double(), w.double(), None, stride, 1, 1, G) * scale.double().view(1, -1, 1, 1) +
                  shift.double().view(1, -1, 1, 1))
    got = ops.grouped_conv3x3(x, w, G, stride, scale, shift, True)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (got.double() - want).abs().max().item() < 2e-6 * want.abs().max().item()
    plain = ops.grouped_conv3x3(x, w, G, stride)                       # no affine, no ReLU
    want2 = F.conv2d(x.double(), w.double(), None, stride, 1, 1, G)
    assert (plain.double() - want2).abs().max().item() < 2e-6 * want2.abs().max().item()


def test_resnext_encoder_and_msc_heads_match_the_reference_modules_gpu():
    """The HIP path (split-operand MFMA kernels for the dense convolutions, the grouped kernel, HIP max-pool) on
    the fixtures of tools/gen_golden_resnext.py: 1e-4, the path's bound."""
    check("cuda", 1e-4)


@pytest.mark.parametrize("cg", [4, 8, 16, 32])
@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("relu", [True, False])
def test_grouped_conv3x3_gradients_match_aten_fp64(cg, stride, relu):
    """ops.GroupedConv3x3 (conv -> BN affine -> ReLU as one node): data and weight gradients against autograd of the
    unfused fp64 ops, with the kernel's own ReLU pattern; odd sizes; the weight gradient bit-reproducible."""
    from sln_amodal_amd import ops
    G = 32
    C = G * cg
    g = torch.Generator(device="cuda").manual_seed(3 * cg + stride)
    x = torch.randn(2, C, 21, 18, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, cg, 3, 3, device="cuda", generator=g) / (9 * cg) ** 0.5
    scale = torch.rand(C, device="cuda", generator=g) + 0.5
    shift = torch.randn(C, device="cuda", generator=g) * 0.1
    xl, wl = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = ops.GroupedConv3x3.apply(xl, wl, scale, shift, relu, G, stride)
    up = torch.randn(y.shape, device="cuda", generator=g)
    y.backward(up)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, 1, 1, G) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if relu:
        yr = yr * (y.detach() > 0)
    yr.backward(up.double())
    for got, want in ((xl.grad, xr.grad), (wl.grad, wr.grad)):
        assert got.shape == want.shape
        assert (got.double() - want).abs().max().item() < 5e-6 * want.abs().max().item()
    first = wl.grad.clone()
    for _ in range(3):
        wl.grad = None
        xl.grad = None
        ops.GroupedConv3x3.apply(xl, wl, scale, shift, relu, G, stride).backward(up)
        assert torch.equal(wl.grad, first)


def test_resnext_block_trains_on_the_hip_path():
    """A GroupBottleneck differentiated on the HIP path (dense convolutions: split-operand MFMA kernels; grouped:
    ops.GroupedConv3x3) against the same block on aten: every gradient within 1e-4."""
    from sln_amodal_amd import nn_ops
    from sln_amodal_amd.modal.resnext import GroupBottleneck
    from tests._util import key_init_
    down = torch.nn.Sequential(torch.nn.Conv2d(128, 256, kernel_size=1, stride=2, bias=False), torch.nn.BatchNorm2d(256))
    blk = GroupBottleneck(128, 128, stride=2, groups=32, downsample=down).cuda().eval()
    key_init_(blk)
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.requires_grad = m.bias.requires_grad = False
    x = torch.randn(2, 128, 24, 24, device="cuda").contiguous(memory_format=torch.channels_last)
    up = None
    grads = {}
    for be in ("hip", "torch"):
        nn_ops.BACKEND = be
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        if up is None:
            up = torch.randn_like(y)
        (y * up).sum().backward()
        grads[be] = [xi.grad.clone()] + [p.grad.clone() for p in blk.parameters() if p.grad is not None]
    assert len(grads["hip"]) == len(grads["torch"]) == 5
    for a, b in zip(grads["hip"], grads["torch"]):
        assert (a - b).norm().item() <= 1e-4 * b.norm().item()


def test_config5_resnext101_msc_forward_full_depth():
    """ResNeXt-101 (3, 4, 23, 3; 32 groups) + ASPP under the multi-scale maximum, 4 x 513^2 images, eval: finite
    logits of the right shape; prints the forward rate."""
    import time
    from sln_amodal_amd.modal.resnext import DeepLabV2_ResNeXt101_MSC
    from tests._util import key_init_
    from tests.test_resnext_cpu import damp_
    net = DeepLabV2_ResNeXt101_MSC(182).eval()
    key_init_(net)
    damp_(net)
    net = net.cuda()
    x = torch.randn(4, 3, 513, 513, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y = net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
    assert tuple(y.shape) == (4, 182, 17, 17) and bool(torch.isfinite(y).all())
    print("ResNeXt-101 MSC forward, 4 x 513^2: %.1f ms (%.1f img/s)" % (dt * 1e3, 4 / dt))



def test_bench_config_resnext_prints_the_contract_line():
    """`bench.py --config resnext` (BASELINE.json configs[4]; its default is 32 x 321^2 per GPU) at a small shape: one
    JSON line with the contract's keys, the dominant kernel's roofline and a finite loss."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    torch.cuda.empty_cache()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "resnext", "--batch", "2",
                        "--dim", "129", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["images_per_gpu"] == 2
    assert out["roofline"]["frac"] is not None and 0 < out["roofline"]["frac"] < 1
    assert out["step_roofline"]["algorithmic_tflop_per_step"] > 0
