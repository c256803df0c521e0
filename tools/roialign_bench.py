"""Time the pyramid RoIAlign forward/backward at the BASELINE shape (16 images, 1600 rois,
256 channels, P2..P5 of a 1024^2 image) -- algorithmic bytes per SURVEY 8(d)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd.modal.modals import _PyramidCrop, roi_levels

def main(pool=16):
    g = torch.Generator().manual_seed(0)
    B, C, K = 16, 256, 1600
    maps = [torch.randn(B, C, s, s, generator=g).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            for s in (256, 128, 64, 32)]
    ctr = torch.rand(K, 2, generator=g) * 0.6 + 0.2
    half = torch.rand(K, 2, generator=g) * 0.18 + 0.02
    boxes = torch.cat([ctr - half, ctr + half], 1).cuda()
    level = roi_levels(boxes, (1024, 1024, 3))
    ind = torch.arange(K, device="cuda").int() % B
    up = torch.randn(K, C, pool, pool, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        out = _PyramidCrop.apply(boxes, ind, level, pool, None, *maps)
        out.backward(up)
    n = 20
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        e[0].record()
        out = _PyramidCrop.apply(boxes, ind, level, pool, None, *maps)
        e[1].record()
        out.backward(up)
        e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    el = K * pool * pool * C
    print("pool %d fwd %.3f ms (%.2f TB/s @20B/el)  bwd %.3f ms incl. memset+alloc (%.2f TB/s @36B/el)" %
          (pool, tf / n, el * 20 / (tf / n) / 1e9, tb / n, el * 36 / (tb / n) / 1e9))
    print("levels", torch.bincount(level.long(), minlength=6).tolist())

if __name__ == "__main__":
    main(16); main(7)
