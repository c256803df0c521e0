"""Layer-level convolution timing on cuda:0: HIP split-bf16 kernel vs aten/MIOpen,
forward and data-gradient, at the BASELINE batch (16 x 1024^2 image shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from sln_amodal_amd import conv_hip

LAYERS = [  # name, N, Cin, H, W, Cout, k, stride, dil
    ("C2 3x3 64->64 @256", 16, 64, 256, 256, 64, 3, 1, 1),
    ("C2 1x1 64->256 @256", 16, 64, 256, 256, 256, 1, 1, 1),
    ("C2 1x1 256->64 @256", 16, 256, 256, 256, 64, 1, 1, 1),
    ("C3 3x3 128->128 @128", 16, 128, 128, 128, 128, 3, 1, 1),
    ("C4 3x3 256->256 @64", 16, 256, 64, 64, 256, 3, 1, 1),
    ("C4 1x1 1024->256 @64", 16, 1024, 64, 64, 256, 1, 1, 1),
    ("C4 1x1 256->1024 @64", 16, 256, 64, 64, 1024, 1, 1, 1),
    ("C5 3x3 512->512 @32", 16, 512, 32, 32, 512, 3, 1, 1),
    ("FPN 3x3 256->256 @256", 16, 256, 256, 256, 256, 3, 1, 1),
    ("RPN 3x3 256->512 @256", 16, 256, 256, 256, 512, 3, 1, 1),
    ("Mask 3x3 256->256 @16 x1600", 1600, 256, 16, 16, 256, 3, 1, 1),
    ("GLM 3x3 d2 256->256 @65", 16, 256, 65, 65, 256, 3, 1, 2),
    ("GLM 1x1 2048->512 @65", 16, 2048, 65, 65, 512, 1, 1, 1),
]


def timeit(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for parts in (2, 3):
    conv_hip.PARTS = parts
    print("---- parts =", parts)
    for name, N, Cin, H, W, Cout, k, s, d in LAYERS:
        x = torch.randn(N, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
        pad = d * (k - 1) // 2
        wp = conv_hip.wsrc(w)
        OH = (H + 2 * pad - d * (k - 1) - 1) // s + 1
        fl = 2.0 * N * OH * OH * Cout * Cin * k * k
        xp, xq = conv_hip.act_parts(x)
        t_split = timeit(lambda: conv_hip._lib.check(conv_hip._lib.lib().sln_act_split_f32(
            conv_hip.ops._ptr(x), N * H * W, Cin, Cin, parts, conv_hip.ops._ptr(xp), conv_hip.ops._ptr(xq), None, None,
            conv_hip.ops._stream()), "s"))
        t_hip = timeit(lambda: conv_hip._fwd(xp, N, H, W, wp, Cout, k, k, (s, s), (d, d), pad, pad, OH, OH, None, None, None, False, xq=xq))
        wcl = w.contiguous(memory_format=torch.channels_last)
        t_ref = timeit(lambda: F.conv2d(x, wcl, None, s, pad, d))
        gy = torch.randn(N, Cout, OH, OH, device="cuda").contiguous(memory_format=torch.channels_last)
        gslot = conv_hip._slot(w, ("gz", OH, OH)) if parts == 2 else None
        gz, _, _ = conv_hip._grad_prep(gy, None, None, False, False, parts, gslot)
        gzq = gslot.scale if gslot is not None else None
        gw = torch.empty((Cout, k, k, Cin), device="cuda")
        t_wg = timeit(lambda: conv_hip._lib.check(conv_hip._lib.lib().sln_conv2d_wgrad_f32(
            conv_hip.ops._ptr(gz), Cout, gz.shape[2], conv_hip.ops._ptr(xp), N, H, W, Cin, xp.shape[2], parts, k, k,
            s, s, d, d, pad, pad, OH, OH, conv_hip.ops._ptr(gw), conv_hip.ops._ptr(gzq), conv_hip.ops._ptr(xq), None, 0, 0,
            conv_hip.ops._stream()), "w"))
        t_wref = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, wcl, None, [s, s], [pad, pad], [d, d], False, [0, 0], 1, [False, True, False]))
        print("%-28s fwd hip %6.3f ms %5.0f TF (split %5.3f) aten %6.3f ms %5.0f TF x%.2f | wgrad hip %6.3f %5.0f TF aten %6.3f %5.0f TF x%.2f" %
              (name, t_hip, fl / t_hip / 1e9, t_split, t_ref, fl / t_ref / 1e9, t_ref / t_hip,
               t_wg, fl / t_wg / 1e9, t_wref, fl / t_wref / 1e9, t_wref / t_wg))
