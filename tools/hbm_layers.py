"""What bounds the HBM-bound 1x1 layers of the backbone: time, algorithmic TB/s and whole-tile stamps
(prologue / k-loop / epilogue cycles of block 0) per epilogue variant and k-loop ablation.
Debug sessions only (SLN_DEBUG_KNOBS)."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLN_DEBUG_KNOBS"] = "1"
import numpy as np
import torch
from sln_amodal_amd import _lib, conv_hip


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


LAYERS = [("C4 conv3 1x1 256->1024 @64", 16, 256, 64, 1024, 1), ("C4 conv1 1x1 1024->256 @64", 16, 1024, 64, 256, 1),
          ("C3 conv3 1x1 128->512 @128", 16, 128, 128, 512, 1), ("C3 conv1 1x1 512->128 @128", 16, 512, 128, 128, 1),
          ("C2 conv3 1x1 64->256 @256", 16, 64, 256, 256, 1), ("C2 conv1 1x1 256->64 @256", 16, 256, 256, 64, 1),
          ("C2 conv2 3x3 64->64 @256", 16, 64, 256, 64, 3), ("C3 conv2 3x3 128->128 @128", 16, 128, 128, 128, 3),
          ("C5 conv3 1x1 512->2048 @32", 16, 512, 32, 2048, 1)]
only = sys.argv[1:] 
for (name, N, Cin, H, Cout, k) in LAYERS:
    if only and not any(o in name for o in only):
        continue
    x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    res = torch.randn(N, Cout, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    sc, sf = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    xp, xq = conv_hip.act_parts(x, 2)
    rp, rq = conv_hip.act_parts(res, 2)
    slot = conv_hip._slot(w, ("y", H, H))
    pad = k // 2
    M = N * H * H
    fl = 2.0 * M * Cout * Cin * k * k
    common = dict(xq=xq)
    A = (xp, N, H, H, conv_hip.wsrc(w, 2), Cout, k, k, (1, 1), (1, 1), pad, pad, H, H)
    variants = {
        "y only":            (lambda: conv_hip._fwd(*A, None, None, None, False, **common), 4 * Cin + 4 * Cout),
        "y+parts":           (lambda: conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, yslot=slot, **common), 4 * Cin + 8 * Cout),
        "parts only":        (lambda: conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, want_y=False, yslot=slot, **common), 4 * Cin + 4 * Cout),
        "res32+y+parts":     (lambda: conv_hip._fwd(*A, sc, sf, res, True, out_parts=True, yslot=slot, **common), 4 * Cin + 12 * Cout),
        "res16+parts only":  (lambda: conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, want_y=False, yslot=slot, res_parts=(rp, rq), **common), 4 * Cin + 8 * Cout),
        # backward shapes: a chained data gradient (producer's mask from part 0, parts + column sums, no fp32 output)
        # and the block-input one (fp32 shortcut gradient in, masked fp32 gradient + scaled parts out)
        "mask16+parts+colsum": (lambda: conv_hip._fwd(*A, sc, None, None, False, out_parts=True, want_y=False, want_colsum=True, yslot=slot, mask_parts=rp, **common), 4 * Cin + 6 * Cout),
        "res32+mask16+y+parts": (lambda: conv_hip._fwd(*A, None, None, res, False, out_parts=True, want_y=True, want_colsum=True, post_scale=sc, yslot=slot, mask_parts=rp, **common), 4 * Cin + 14 * Cout),
    }
    for v in variants.values():
        v[0](); v[0]()
    layout = conv_hip.weights_layout(M, Cout, xp.shape[2], k * k, 2, xp.shape[1])
    print("%s  [%s]  M=%d  %.1f GFLOP" % (name, conv_hip._fwd_kernel_name(layout, 2), M, fl / 1e9))
    for vn, (f, bpp) in variants.items():
        row = []
        base_dbg = int(os.environ.get("SLN_HBM_LAYERS_DBG", "0"))      # e.g. 4096: activations one stage ahead
        for dbg in ("0", "32", "1", "2"):
            os.environ["SLN_CONV_DBG"] = str(int(dbg) | base_dbg)
            t = timeit(f)
            row.append("%s %.3f ms" % ({"0": "full", "64": "nt", "32": "4-wide", "1": "noDMA", "2": "noMFMA"}[dbg], t))
            if dbg == "0":
                row.append("%5.2f TB/s %4.0f TF" % (M * bpp / t / 1e9, fl / t / 1e9))
        os.environ["SLN_CONV_DBG"] = str(base_dbg)
        st = ""
        if layout == conv_hip.TILED256H:
            os.environ["SLN_CONV_STAMP"] = "1"
            f()
            torch.cuda.synchronize()
            buf = (ctypes.c_uint64 * 128)()
            _lib.check(_lib.lib().sln_debug_read_stamps(buf), "stamps")
            a = np.array(buf, dtype=np.float64).reshape(8, 16)
            st = " | block 0 cycles: prologue %d k-loop %d epilogue %d" % (a[0, 8], a[0, 9], a[0, 10])
            os.environ["SLN_CONV_STAMP"] = "0"
        print("   %-18s %s%s" % (vn, " | ".join(row), st))
