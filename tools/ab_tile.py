"""A/B of the forward kernels per layer shape: 256x256 fp16 kernel vs the 128x128 kernel (SLN_CONV_TILE256
debug knob), with the realistic epilogue (BN affine, residual, ReLU, output parts)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SLN_DEBUG_KNOBS"] = "1"
import torch
from sln_amodal_amd import conv_hip


def timeit(fn, iters=20):
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


LAYERS = [("C4 1x1 256->1024 @64 +res", 16, 256, 64, 1024, 1, True), ("C4 1x1 256->1024 @64 dgrad-like", 16, 256, 64, 1024, 1, False),
          ("C4 1x1 1024->256 @64", 16, 1024, 64, 256, 1, False), ("C3 1x1 512->128 @128", 16, 512, 128, 128, 1, False),
          ("C3 1x1 128->512 @128 +res", 16, 128, 128, 512, 1, True), ("C5 1x1 512->2048 @32 +res", 16, 512, 32, 2048, 1, True),
          ("C2 1x1 256->256 @256", 16, 256, 256, 256, 1, False), ("C4 3x3 256->256 @64", 16, 256, 64, 256, 3, False)]
for (name, N, Cin, H, Cout, k, use_res) in LAYERS:
    x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    res = torch.randn(N, Cout, H, H, device="cuda").contiguous(memory_format=torch.channels_last) if use_res else None
    sc, sf = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    xp, xq = conv_hip.act_parts(x, 2)
    slot = conv_hip._slot(w, ("y", H, H))
    pad = k // 2
    fl = 2.0 * N * H * H * Cout * Cin * k * k
    by = N * H * H * (Cin * 4 + Cout * (8 + (4 if use_res else 0)))
    row = []
    for mode in ("1", "0", "2"):
        os.environ["SLN_CONV_TILE256"] = mode
        w_ = w.clone()      # fresh weight-part cache per layout
        slot_ = conv_hip._slot(w_, ("y", H, H))
        f = lambda: conv_hip._fwd(xp, N, H, H, conv_hip.wsrc(w_, 2), Cout, k, k, (1, 1), (1, 1), pad, pad, H, H, sc, sf,
                                  res, True, out_parts=True, xq=xq, yslot=slot_)
        f(); f()
        t = timeit(f)
        row.append("tile256=%s %.3f ms %4.0f TF %.2f TB/s" % (mode, t, fl / t / 1e9, by / t / 1e9))
    print("%-34s | %s" % (name, " | ".join(row)))
