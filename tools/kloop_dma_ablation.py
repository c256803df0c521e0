"""Which operand stream bounds the k-loop of conv_fwd256h_kernel: a K-heavy pointwise layer, a K-light one and a 3x3 layer
timed whole, without the weight pieces / the activation pieces / both in the k-loop (SLN_CONV_DBG bits 28, 29), without
MFMAs, without the epilogue.  Debug sessions only (SLN_DEBUG_KNOBS); results are wrong under every ablation."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["SLN_DEBUG_KNOBS"] = "1"
os.environ["SLN_CONV_TILE128H"] = "0"
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
for (name, N, Cin, H, Cout, k) in [("C4 conv1 1x1 1024->256 @64", 16, 1024, 64, 256, 1), ("C4 conv3 1x1 256->1024 @64", 16, 256, 64, 1024, 1),
                                   ("C4 conv2 3x3 256->256 @64", 16, 256, 64, 256, 3), ("FPN 3x3 256->256 @256", 4, 256, 256, 256, 3),
                                   ("mask head 3x3 256->256 @16", 1600, 256, 16, 256, 3)]:
    x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.03
    sc, sf = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    xp, xq = conv_hip.act_parts(x, 2)
    slot = conv_hip._slot(w, ("y", H, H))
    A = (xp, N, H, H, conv_hip.wsrc(w, 2), Cout, k, k, (1, 1), (1, 1), k // 2, k // 2, H, H)
    for _ in range(2):
        conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, yslot=slot, xq=xq)
    f = lambda: conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, want_y=False, yslot=slot, xq=xq)
    row = []
    for label, dbg in (("full", 0), ("no weight pieces", 1 << 28), ("no activation pieces", 1 << 29), ("neither", 3 << 28), ("activation pieces of the first tap column only", 1 << 20), ("no MFMA", 2),
                       ("activation pieces in phase 0 (round-3 placement)", 1 << 30), ("no epilogue", 32768), ("no epilogue, no weight pieces", 32768 | (1 << 28)), ("no epilogue, no act pieces", 32768 | (1 << 29))):
        os.environ["SLN_CONV_DBG"] = str(dbg)
        row.append("%s %.3f" % (label, timeit(f)))
    os.environ["SLN_CONV_DBG"] = "0"
    print(name, "|", " | ".join(row))
