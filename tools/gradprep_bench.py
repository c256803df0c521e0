"""Time grad_prep_kernel (ReLU mask x BN scale -> bf16 parts + bias gradient) at backbone shapes.
Algorithmic bytes: read gy 4 + read y 4 + write P*2 parts (+4 when the masked fp32 gradient is kept)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd import conv_hip

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (N, C, H) in [(16, 64, 256), (16, 256, 256), (16, 512, 128), (16, 1024, 64), (1600, 256, 16), (16, 440, 64)]:
    gy = torch.randn(N, H, H, C, device="cuda").permute(0, 3, 1, 2)
    y = torch.randn(N, H, H, C, device="cuda").permute(0, 3, 1, 2)
    sc = torch.rand(C, device="cuda") + 0.5
    for want_gu, want_bias in [(False, True), (False, False), (True, True)]:
        ms = t(lambda: conv_hip._grad_prep(gy, y, sc, want_gu, want_bias, 3, None))
        el = N * H * H * C
        b = el * (8 + 6 + (4 if want_gu else 0))
        print("N%d C%d H%d gu=%d bias=%d: %.3f ms  %.2f TB/s" % (N, C, H, want_gu, want_bias, ms, b / ms / 1e9))
