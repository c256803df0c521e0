"""`pth_nms(dets, thresh)` on MI355X.  Mirrors the reference's nms/pth_nms.py:5-51
call shape; the arithmetic runs in sln_nms_f32 (csrc/nms.hip) with the reference
CPU semantics (nms/src/nms.c:55-61: `>=`, +1 widths).  Unlike the reference GPU
path there is no host-side reduce; the only host sync is reading the kept count,
which this variable-length API cannot avoid (use ops.nms_sorted for the batched,
fixed-capacity, sync-free form the training step uses)."""
import torch

from .. import ops


def pth_nms(dets, thresh):
    """dets: Tensor[N,5] f32 rows (y1,x1,y2,x2,score) on the GPU.
    Returns LongTensor[K]: indices into `dets`, descending-score visiting order.
    Score ties visit the lower index first (stable sort; torch 0.4's sort in the
    reference is unstable, so ties are unspecified there)."""
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=dets.device)
    if dets.dim() != 2 or dets.size(1) != 5:
        raise RuntimeError("dets must be [N,5]")  # THArgCheck -> RuntimeError in the reference
    dets = dets.detach().float()
    order = torch.sort(dets[:, 4], descending=True, stable=True)[1]
    srt = dets[order].contiguous().unsqueeze(0)
    n = srt.shape[1]
    keep, num = ops.nms_sorted(srt, thresh, n)
    k = int(num[0].item())
    return order[keep[0, :k]].contiguous()
