"""Same-box A/B of the two epilogues of the fp16 forward kernels (SLN_CONV_DBG bit 16 = general epilogue_slab)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLN_DEBUG_KNOBS"] = "1"
import torch
from sln_amodal_amd import conv_hip


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


LAYERS = [("FPN 3x3 256->256 @256", 16, 256, 256, 256, 3, False), ("C4 3x3 256->256 @64", 16, 256, 64, 256, 3, False),
          ("C4 1x1 256->1024 @64 +res", 16, 256, 64, 1024, 1, True), ("C4 1x1 1024->256 @64", 16, 1024, 64, 256, 1, False),
          ("C2 3x3 64->64 @256 (128 kernel)", 16, 64, 256, 64, 3, False), ("C3 1x1 128->512 @128 +res (128 kernel)", 16, 128, 128, 512, 1, True)]
for (name, N, Cin, H, Cout, k, use_res) in LAYERS:
    x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    res = torch.randn(N, Cout, H, H, device="cuda").contiguous(memory_format=torch.channels_last) if use_res else None
    sc, sf = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    xp, xq = conv_hip.act_parts(x, 2)
    slot = conv_hip._slot(w, ("y", H, H))
    pad = k // 2
    fl = 2.0 * N * H * H * Cout * Cin * k * k
    f = lambda: conv_hip._fwd(xp, N, H, H, conv_hip.wsrc(w, 2), Cout, k, k, (1, 1), (1, 1), pad, pad, H, H, sc, sf, res, True,
                              out_parts=True, xq=xq, yslot=slot)
    f(); f()
    row = []
    for rep in range(2):
        for dbg in (sys.argv[1:] or ["16", "0"]):
            os.environ["SLN_CONV_DBG"] = dbg
            t = timeit(f)
            row.append("%s %.3f ms %4.0f TF" % ({"16": "general", "0": "fixed  ", "32": "no-ahead"}.get(dbg, dbg), t, fl / t / 1e9))
    print("%-40s | %s" % (name, " | ".join(row)))
