import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from sln_amodal_amd import conv_hip
g = torch.Generator(device="cuda").manual_seed(11)
N, Cin, Cout, H, W = 2, 256, 128, 32, 32
x = torch.randn(N, Cin, H, W, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
w = torch.randn(Cout, Cin, 1, 1, device="cuda", generator=g) / 16
gz = torch.randn(N, Cout, 16, 16, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
# manual: gx[n,ci,2i,2j] = sum_co gz[n,co,i,j] w[co,ci]
man = torch.zeros(N, Cin, H, W, device="cuda", dtype=torch.float64)
man[:, :, ::2, ::2] = torch.einsum("noij,oc->ncij", gz.double(), w[:, :, 0, 0].double())
for fmt in ("cl", "nchw"):
    xx = x if fmt == "cl" else x.contiguous()
    gg = gz if fmt == "cl" else gz.contiguous()
    g_in, _, _ = torch.ops.aten.convolution_backward(gg, xx, w, None, [2, 2], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])
    print("aten", fmt, (g_in.double() - man).abs().max().item())
xd = x.double().requires_grad_(True)
yd = F.conv2d(xd, w.double(), None, 2)
yd.backward(gz.double())
print("fp64 autograd", (xd.grad - man).abs().max().item())
wt = conv_hip._split(w, flip_swap=True)
small = conv_hip._fwd(gz, wt, Cin, 1, 1, (1, 1), (1, 1), 0, 0, 16, 16, None, None, None, False)
print("hip small", (small.double() - man[:, :, ::2, ::2]).abs().max().item(), man.abs().max().item())
