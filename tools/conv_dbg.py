import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd import conv_hip
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for parts in (3, 2):
    for (name, N, Cin, H, Cout, k) in [("FPN3x3", 16, 256, 256, 256, 3), ("C4 1x1 1024->256", 16, 1024, 64, 256, 1), ("C4 1x1 256->1024", 16, 256, 64, 1024, 1)]:
        x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
        xp, xq = conv_hip.act_parts(x, parts); wp, wq = conv_hip._split_weights(w, parts=parts)
        pad = k // 2
        fl = 2.0 * N * H * H * Cout * Cin * k * k
        t = timeit(lambda: conv_hip._fwd(xp, N, H, H, wp, Cout, k, k, (1, 1), (1, 1), pad, pad, H, H, None, None, None, False, xq=xq, wq=wq))
        print("dbg=%s parts=%d %-18s %.3f ms %.0f TF" % (os.environ.get("SLN_CONV_DBG", "0"), parts, name, t, fl / t / 1e9))
