"""Would running the weight gradients on a second stream, next to the data gradients, pay?  A bottleneck
block's backward (C4 shapes, 16 x 1024^2) as three (dgrad, wgrad) pairs, 23 blocks: one stream against two
(wgrad i waits for the event recorded before dgrad i and overlaps it and what follows)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd import conv_hip, ops
from sln_amodal_amd import _lib

N, H = 16, 64
SHAPES = [(256, 1024, 1), (256, 256, 3), (1024, 256, 1)]   # forward (Cin, Cout, k) of conv3, conv2, conv1 in backward order
if len(sys.argv) > 1 and sys.argv[1] == "glm":
    N, H = 16, 65
layers = []
for (Cin, Cout, k) in SHAPES:
    x = torch.randn(N, Cin, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    gy = torch.randn(N, Cout, H, H, device="cuda").contiguous(memory_format=torch.channels_last)
    xp, xq = conv_hip.act_parts(x)
    slot = conv_hip._slot(w, ("gz", H, H))
    gz, _, _ = conv_hip._grad_prep(gy, None, None, False, False, 2, slot)
    wt = conv_hip.wsrc(w, 2, True, w)
    gw = torch.empty((Cout, k, k, Cin), device="cuda")
    layers.append(dict(x=x, w=w, xp=xp, xq=xq, gz=gz, gzq=slot.scale, wt=wt, gw=gw, Cin=Cin, Cout=Cout, k=k))


def dgrad(L):
    k = L["k"]
    return conv_hip._fwd(L["gz"], N, H, H, L["wt"], L["Cin"], k, k, (1, 1), (1, 1), k // 2, k // 2, H, H, None, None, None,
                         False, cin=L["Cout"], xq=L["gzq"])


def wgrad(L):
    k = L["k"]
    _lib.check(_lib.lib().sln_conv2d_wgrad_f32(
        ops._ptr(L["gz"]), L["Cout"], L["gz"].shape[2], ops._ptr(L["xp"]), N, H, H, L["Cin"], L["xp"].shape[2], 2, k, k,
        1, 1, 1, 1, k // 2, k // 2, H, H, ops._ptr(L["gw"]), ops._ptr(L["gzq"]), ops._ptr(L["xq"]), None, 0, 0,
        ops._stream()), "w")


side = torch.cuda.Stream()


def serial(reps=23):
    for _ in range(reps):
        for L in layers:
            dgrad(L)
            wgrad(L)


def two(reps=23):
    main = torch.cuda.current_stream()
    for _ in range(reps):
        for L in layers:
            ev = torch.cuda.Event()
            ev.record(main)
            dgrad(L)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                wgrad(L)
    main.wait_stream(side)


def only(fn, reps=23):
    for _ in range(reps):
        for L in layers:
            fn(L)


def t_ms(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("dgrad only  %.3f ms" % t_ms(lambda: only(dgrad)))
print("wgrad only  %.3f ms" % t_ms(lambda: only(wgrad)))
print("one stream  %.3f ms" % t_ms(serial))
print("two streams %.3f ms" % t_ms(two))
print("one stream  %.3f ms" % t_ms(serial))
print("two streams %.3f ms" % t_ms(two))
