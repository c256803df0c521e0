"""Diagnostics: 3-bottleneck stack, PARTS 2 vs 3 vs fp64 -- where do gradients part ways?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SLN_DEBUG_KNOBS", "1")
import torch, torch.nn as nn
from sln_amodal_amd import conv_hip, nn_ops
from sln_amodal_amd.modal.modals import Bottleneck
from tests._util import key_init_

def build():
    down = nn.Sequential(nn.Conv2d(64, 128, kernel_size=1, stride=1), nn.BatchNorm2d(128, eps=0.001))
    net = nn.Sequential(Bottleneck(64, 32, 1, down), Bottleneck(128, 32), Bottleneck(128, 32)).cuda()
    key_init_(net)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval(); m.weight.requires_grad = m.bias.requires_grad = False
    return net

g = torch.Generator().manual_seed(21)
x0 = torch.randn(2, 64, 24, 24, generator=g).cuda().contiguous(memory_format=torch.channels_last)
up = torch.randn(2, 128, 24, 24, generator=g).cuda().contiguous(memory_format=torch.channels_last)
res = {}
acts = {}
def hook(tag):
    def mk(name):
        def fh(m, i, o):
            acts[(tag, name)] = o.detach().double().clone()
        return fh
    return mk
for tag, parts, chain, passes in (("p3", 3, True, 2), ("p2", 2, True, 2), ("p2nochain", 2, False, 2), ("p2first", 2, True, 1)):
    conv_hip.PARTS = parts
    conv_hip.CHAIN_GRAD_PREP = chain
    nn_ops.BACKEND = "hip"
    net = build()
    for _ in range(passes):
        conv_hip.update_scales()
        x = x0.clone().requires_grad_(True)
        net.zero_grad(set_to_none=True)
        y = net(x); y.backward(up)
    res[tag] = (y.detach().double(), x.grad.double(), {k: p.grad.double() for k, p in net.named_parameters() if p.grad is not None})
conv_hip.CHAIN_GRAD_PREP = True
nn_ops.BACKEND = "torch"
ref = build().double()
for m in ref.modules():
    if isinstance(m, nn.BatchNorm2d):
        m.eval(); m.weight.requires_grad = m.bias.requires_grad = False
xr = x0.double().clone().requires_grad_(True)
yr = ref(xr); yr.backward(up.double())
res["f64"] = (yr.detach(), xr.grad, {k: p.grad for k, p in ref.named_parameters() if p.grad is not None})
rl2 = lambda a, b: ((a - b).norm() / b.norm()).item()
for tag in ("p3", "p2", "p2nochain", "p2first"):
    print(tag, "y %.2e gx %.2e" % (rl2(res[tag][0], res["f64"][0]), rl2(res[tag][1], res["f64"][1])),
          "masks differ:", int(((res[tag][0] > 0) != (res["f64"][0] > 0)).sum()),
          "worst w", max((rl2(res[tag][2][k], res["f64"][2][k]), k) for k in res["f64"][2]))
print("p2 vs p2nochain gx", rl2(res["p2"][1], res["p2nochain"][1]), "equal", torch.equal(res["p2"][1], res["p2nochain"][1]))
print("p2 vs p3 gx", rl2(res["p2"][1], res["p3"][1]))
print("saturated", conv_hip.saturation_count())

# ---- internal masks of every conv output, PARTS 2 vs 3 vs fp64
print("---- internal ReLU masks")
def internals(net, x, double=False):
    outs = []
    conv = nn_ops.conv_bn_act
    for bi, blk in enumerate(net):
        res_ = x if blk.downsample is None else conv(x, blk.downsample[0], blk.downsample[1])
        o1 = conv(x, blk.conv1, blk.bn1, relu=True)
        o2 = conv(o1, blk.conv2, blk.bn2, relu=True, same=True)
        o3 = conv(o2, blk.conv3, blk.bn3, relu=True, residual=res_)
        outs += [o1, o2, o3]
        x = o3
    return outs
with torch.no_grad():
    nn_ops.BACKEND = "hip"
    net = build()
    conv_hip.PARTS = 3; i3 = internals(net, x0)
    conv_hip.PARTS = 2; i2 = internals(net, x0)
    nn_ops.BACKEND = "torch"
    netd = build().double()
    id_ = internals(netd, x0.double())
for li, (a, b, c) in enumerate(zip(i3, i2, id_)):
    d3 = ((a > 0) != (c > 0)); d2 = ((b > 0) != (c > 0))
    print("layer", li, "mask diffs p3:", int(d3.sum()), "p2:", int(d2.sum()),
          "| p2 offending pre-act (fp64 value):", c[d2].tolist()[:3], "p2 value:", b[d2].tolist()[:3])
