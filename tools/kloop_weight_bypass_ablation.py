"""Upper bound of taking the WEIGHT fragments of conv_fwd256h_kernel out of LDS (direct-to-register operand loads): the
k-loop timed whole and with SLN_CONV_DBG bit 16 (the weight fragments of stage 0 kept in registers for every stage, no
weight pieces DMA-ed: every other instruction of the loop stays), with and without the epilogue.  Results are wrong
under the ablation.  Debug sessions only (SLN_DEBUG_KNOBS)."""
import os, sys
sys.path.insert(0, os.getcwd())
os.environ["SLN_DEBUG_KNOBS"] = "1"
os.environ["SLN_CONV_TILE128H"] = "0"
import torch
from sln_amodal_amd import conv_hip


def timeit(fn, iters=40):
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


for (name, N, Cin, H, Cout, k, d) in [("C4 conv2 3x3 256->256 @64 (tap-row)", 16, 256, 64, 256, 3, 1),
                                      ("FPN 3x3 256->256 @256 (tap-row)", 16, 256, 256, 256, 3, 1),
                                      ("RPN 3x3 256->512 @256 (tap-row)", 16, 256, 256, 512, 3, 1),
                                      ("mask head 3x3 256->256 @16 (tap-column)", 1600, 256, 16, 256, 3, 1),
                                      ("GLM 3x3 d2 256->256 @65 (plain loop)", 16, 256, 65, 256, 3, 2),
                                      ("ASPP 3x3 d12 2048->182 @65 (plain loop)", 16, 2048, 65, 184, 3, 12),
                                      ("C4 conv1 1x1 1024->256 @64 (plain loop)", 16, 1024, 64, 256, 1, 1),
                                      ("C4 conv3 1x1 256->1024 @64 (plain loop)", 16, 256, 64, 1024, 1, 1)]:
    x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.03
    sc, sf = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda")
    xp, xq = conv_hip.act_parts(x, 2)
    slot = conv_hip._slot(w, ("y", H, H))
    pad = d * (k // 2)
    A = (xp, N, H, H, conv_hip.wsrc(w, 2), Cout, k, k, (1, 1), (d, d), pad, pad, H, H)
    for _ in range(2):
        conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, yslot=slot, xq=xq)
    f = lambda: conv_hip._fwd(*A, sc, sf, None, True, out_parts=True, want_y=False, yslot=slot, xq=xq)
    row = []
    for label, dbg in (("full", 0), ("weights not through LDS", 65536), ("no epilogue", 32768),
                       ("no epilogue, weights not through LDS", 32768 | 65536), ("full again", 0)):
        os.environ["SLN_CONV_DBG"] = str(dbg)
        t = timeit(f)
        kern = conv_hip._lib.lib().sln_conv_fwd_last_kernel()
        row.append("%s %.4f ms (kernel %d)" % (label, t, kern))
    os.environ["SLN_CONV_DBG"] = "0"
    fl = 2.0 * N * H * H * Cout * Cin * k * k
    print(name, "| %.1f GFLOP |" % (fl / 1e9), " | ".join(row), flush=True)
