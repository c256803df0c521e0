"""Where conv_fwd256h_kernel spends a phase: stamped diagnostic build (SLN_CONV_STAMP), block 0, per wave.
Segments of a phase: 0 = fragment reads + DMA issue + own waits, 1 = wait at the middle barrier,
2 = the 12-MFMA cluster, 3 = wait at the closing barrier."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLN_DEBUG_KNOBS"] = "1"
os.environ["SLN_CONV_TILE256"] = "2"
import numpy as np
import torch
from sln_amodal_amd import _lib, conv_hip

for (name, N, Cin, H, Cout, k) in [("FPN 3x3 256->256 @256", 16, 256, 256, 256, 3), ("C4 1x1 1024->256 @64", 16, 1024, 64, 256, 1),
                                   ("3x3 256->256 @128, 1 image: 64 tiles (quiet HBM)", 1, 256, 128, 256, 3),
                                   ("3x3 256->256 @64, 1 image: 16 tiles (quiet HBM)", 1, 256, 64, 256, 3)]:
    x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    xp, xq = conv_hip.act_parts(x, 2)
    # whole-tile segments with the realistic epilogue (BN affine, shortcut, ReLU, output parts)
    res = torch.randn(N, Cout, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    sc, sf = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    slot = conv_hip._slot(w, ("y", H, H))
    for full in (False, True):
        os.environ["SLN_CONV_STAMP"] = "0"
        args = (sc, sf, res, True) if full else (None, None, None, False)
        kw = dict(out_parts=True, yslot=slot) if full else {}
        for _ in range(2):
            conv_hip._fwd(xp, N, H, H, conv_hip.wsrc(w, 2), Cout, k, k, (1, 1), (1, 1), k // 2, k // 2, H, H, *args, xq=xq, **kw)
        os.environ["SLN_CONV_STAMP"] = "1"
        conv_hip._fwd(xp, N, H, H, conv_hip.wsrc(w, 2), Cout, k, k, (1, 1), (1, 1), k // 2, k // 2, H, H, *args, xq=xq, **kw)
        torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * 128)()
        _lib.check(_lib.lib().sln_debug_read_stamps(buf), "stamps")
        a = np.array(buf, dtype=np.float64).reshape(8, 16)
        print("%s %s epilogue: block 0, cycles  prologue %d | k-loop %d | epilogue (stores retired) %d" % (
            name, "full" if full else "plain", a[0, 8], a[0, 9], a[0, 10]))
        for wv in (0, 4):
            print("      wave %d, 4 slabs summed: stage->LDS %d | barrier %d | slab (loads, math, stores issued) %d | "
                  "barrier %d | final store drain %d" % ((wv,) + tuple(a[wv, 11:16])))
    for dbg in (0, 1, 2):
        os.environ["SLN_CONV_DBG"] = str(dbg)
        os.environ["SLN_CONV_STAMP"] = "1"
        conv_hip._fwd(xp, N, H, H, conv_hip.wsrc(w, 2), Cout, k, k, (1, 1), (1, 1), k // 2, k // 2, H, H, None, None, None,
                      False, xq=xq)
        torch.cuda.synchronize()
        buf = (ctypes.c_uint64 * 128)()
        _lib.check(_lib.lib().sln_debug_read_stamps(buf), "stamps")
        a = np.array(buf, dtype=np.float64).reshape(8, 4, 4)
        nst = k * k * ((Cin + 31) // 32)
        print("%s dbg=%d: cycles per stage (sum over 4 phases), stages=%d" % (name, dbg, nst))
        for wv in (0, 4):
            per = a[wv] / nst
            print("  wave %d: per phase [reads+dma | mid-barrier | mfma | close-barrier]" % wv)
            for ph in range(4):
                print("     ph%d  %6.0f %6.0f %6.0f %6.0f" % ((ph,) + tuple(per[ph])))
            print("     total/stage %.0f  shares: reads+dma %.2f mid %.2f mfma %.2f close %.2f" %
                  ((per.sum(),) + tuple(per.sum(axis=0) / per.sum())))
    os.environ["SLN_CONV_STAMP"] = "0"
    os.environ["SLN_CONV_DBG"] = "0"
