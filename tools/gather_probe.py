"""Where the gather backward of the pyramid RoIAlign spends its time: empty lists (fixed cost of the 87 k tile
blocks: decode, zero write), rois of one level only, all rois."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd.modal import modals
from sln_amodal_amd.modal.modals import _PyramidCrop, roi_levels


def t_ms(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator(device="cuda").manual_seed(7)
B, C, K, pool = 16, 256, 1600, 16
maps = [torch.randn(B, C, s, s, device="cuda", generator=g).contiguous(memory_format=torch.channels_last) for s in (256, 128, 64, 32)]
ctr = torch.rand(K, 2, device="cuda", generator=g) * 0.6 + 0.2
size = torch.exp(torch.rand(K, 2, device="cuda", generator=g) * 2.5 - 3.0)
boxes = torch.cat([ctr - size / 2, ctr + size / 2], 1).clamp(0, 1).contiguous()
ind = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(100)
lvl = roi_levels(boxes, (1024, 1024))
print("levels", torch.bincount(lvl.long(), minlength=6).tolist())
up = torch.randn(K, C, pool, pool, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
shapes = [tuple(m.shape) for m in maps]
for name, i2, l2 in (("no rois (all padded)", torch.full_like(ind, -1), lvl),
                     ("level 2 only", torch.where(lvl == 2, ind, torch.full_like(ind, -1)), lvl),
                     ("level 3 only", torch.where(lvl == 3, ind, torch.full_like(ind, -1)), lvl),
                     ("level 4 only", torch.where(lvl == 4, ind, torch.full_like(ind, -1)), lvl),
                     ("level 5 only", torch.where(lvl == 5, ind, torch.full_like(ind, -1)), lvl),
                     ("all", ind, lvl)):
    src = (up, C, 0, boxes, i2.contiguous(), l2.contiguous(), pool)
    print("%-22s %.3f ms" % (name, t_ms(lambda: modals._gather_backward([src], shapes, up.device))))
