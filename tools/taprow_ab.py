"""conv_fwd256h_kernel's tap-row instances (activation stages per kernel ROW, SLN_CONV_TAPROW=1) against the plain k-loop
(=0): isolated launches of the step's 3x3 layer shapes, parts-only output, alternating."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["SLN_DEBUG_KNOBS"] = "1"
import torch
from sln_amodal_amd import conv_hip
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for (name, N, Cin, H, Cout, dil) in [("C4 3x3 256->256 @64", 16, 256, 64, 256, 1), ("FPN 3x3 256->256 @256", 16, 256, 256, 256, 1),
                                     ("RPN 3x3 256->512 @256", 16, 256, 256, 512, 1), ("FPN 3x3 256->256 @128", 16, 256, 128, 256, 1),
                                     ("3x3 512->256 @256", 16, 512, 256, 256, 1), ("mask head 3x3 256->256 @16", 1600, 256, 16, 256, 1), ("mask head 3x3 256->439 @16", 1600, 256, 16, 439, 1)]:
    x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.03
    sc, sf = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    xp, xq = conv_hip.act_parts(x, 2)
    slot = conv_hip._slot(w, ("y", H, H))
    A = (xp, N, H, H, conv_hip.wsrc(w, 2), Cout, 3, 3, (1, 1), (dil, dil), dil, dil, H, H)
    for _ in range(2):
        conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, yslot=slot, xq=xq)
    f = lambda: conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, want_y=False, yslot=slot, xq=xq)
    flops = 2.0 * N * H * H * Cout * 9 * Cin
    row = []
    for rep in range(2):
        for v in ("0", "1"):
            os.environ["SLN_CONV_TAPROW"] = v
            ms = timeit(f)
            row.append("taprow=%s %.3f ms %.0f TF (kernel %d)" % (v, ms, flops / ms * 1e-9, conv_hip._lib.lib().sln_conv_fwd_last_kernel()))
    print(name, "|", " | ".join(row), flush=True)
