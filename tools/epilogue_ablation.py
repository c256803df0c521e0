"""What the eight-channel epilogue of conv_fwd256h_kernel spends its time on: pointwise layers whose launch is almost
all epilogue (parts-only / shortcut-from-parts outputs), timed whole, with the k-loop's DMA and MFMAs switched off
(SLN_CONV_DBG bits 1 | 2), and then without the part stores (8192), without the split arithmetic (16384), without
both.  Debug sessions only (SLN_DEBUG_KNOBS)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLN_DEBUG_KNOBS"] = "1"
import torch
from sln_amodal_amd import conv_hip


def timeit(fn, iters=30):
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


for (name, N, Cin, H, Cout) in [("C4 conv3 1x1 256->1024 @64", 16, 256, 64, 1024),
                                ("C4 conv1 1x1 1024->256 @64", 16, 1024, 64, 256)]:
    x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05
    res = torch.randn(N, Cout, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    sc, sf = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    xp, xq = conv_hip.act_parts(x, 2)
    rp, rq = conv_hip.act_parts(res, 2)
    slot = conv_hip._slot(w, ("y", H, H))
    A = (xp, N, H, H, conv_hip.wsrc(w, 2), Cout, 1, 1, (1, 1), (1, 1), 0, 0, H, H)
    variants = {
        "parts only": lambda: conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, want_y=False, yslot=slot, xq=xq),
        "res16+parts only": lambda: conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, want_y=False, yslot=slot,
                                                  res_parts=(rp, rq), xq=xq),
    }
    for _ in range(2):      # bootstrap the output's scale slot (a plain launch: fp32 + parts)
        conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, yslot=slot, xq=xq)
    print(name)
    for vn, f in variants.items():
        f(); f()
        row = []
        for label, dbg in (("full", 0), ("k-loop off", 3), ("k-loop off, no stores", 3 | 8192),
                           ("k-loop off, no split", 3 | 16384), ("k-loop off, neither", 3 | 8192 | 16384),
                           ("no stores", 8192), ("no split", 16384), ("no epilogue", 32768), ("no column constants", 131072), ("k-loop off, no column constants", 3 | 131072),
                           ("k-loop off, no epilogue", 3 | 32768), ("no fragment reads either", 7 | 32768)):
            os.environ["SLN_CONV_DBG"] = str(dbg)
            row.append("%s %.3f ms" % (label, timeit(f)))
        os.environ["SLN_CONV_DBG"] = "0"
        print("   %-18s %s" % (vn, " | ".join(row)))
