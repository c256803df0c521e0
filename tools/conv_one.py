import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd import conv_hip
parts = int(os.environ.get("SLN_CONV_PARTS", "3"))
N, Cin, H, Cout, k = 16, 256, 256, 256, 3
x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
xp = conv_hip.act_parts(x, parts); wp = conv_hip._split_weights(w, parts=parts)
for _ in range(3):
    y = conv_hip._fwd(xp, N, H, H, wp, Cout, k, k, (1, 1), (1, 1), 1, 1, H, H, None, None, None, False)
torch.cuda.synchronize()
