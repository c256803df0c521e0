"""Ablation of conv_fwd256_kernel's k-loop (SLN_CONV_DBG bits: 1 no DMA, 2 no MFMA, 4 no fragment reads) on a
few layer shapes, both operand formats.  Debug sessions only (SLN_DEBUG_KNOBS)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLN_DEBUG_KNOBS"] = "1"
import torch
from sln_amodal_amd import conv_hip


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


LAYERS = [("C4 3x3 256->256 @64", 16, 256, 64, 256, 3), ("FPN 3x3 256->256 @256", 16, 256, 256, 256, 3),
          ("C4 1x1 1024->256 @64", 16, 1024, 64, 256, 1), ("C4 1x1 256->1024 @64", 16, 256, 64, 1024, 1)]
os.environ["SLN_CONV_TILE256"] = "2"
for parts, f16k in ((2, 0), (2, 2)):      # (the four-phase body is no longer built)
    os.environ["SLN_CONV_F16_KERNEL"] = "1" if f16k else "0"
    os.environ["SLN_CONV_PHASES"] = str(f16k or 2)
    for (name, N, Cin, H, Cout, k) in LAYERS:
        x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
        xp, xq = conv_hip.act_parts(x, parts)
        wp = conv_hip.wsrc(w, parts)
        pad = k // 2
        fl = 2.0 * N * H * H * Cout * Cin * k * k
        row = []
        for dbg in ((0, 1, 2, 3, 4, 5, 6, 7) if not f16k else (0, 1, 2, 3, 8, 9)):
            os.environ["SLN_CONV_DBG"] = str(dbg)
            t = timeit(lambda: conv_hip._fwd(xp, N, H, H, wp, Cout, k, k, (1, 1), (1, 1), pad, pad, H, H, None, None,
                                             None, False, xq=xq))
            row.append("%d:%.3f" % (dbg, t))
        os.environ["SLN_CONV_DBG"] = "0"
        t0 = float(row[0].split(":")[1])
        mf = fl * (6 if parts == 3 else 3) / 2.5e15 * 1e3
        print("parts=%d f16k=%d %-24s %5.0f TF  mfma-only floor %.3f ms | ms by dbg %s" % (parts, f16k, name, fl / t0 / 1e9, mf, " ".join(row)))
