"""COCO run-length masks for the evaluation path: the slice of pycocotools.mask the reference uses
(`maskUtils.encode(np.asfortranarray(mask))`, amodal_train.py:397), backed by the HIP run-length
kernel (cocoapi/common/maskApi.c:33-42 rleEncode) and the host-side string codec of the C ABI
(maskApi.c:204-231 rleToString / rleFrString).  No CPU encoder lives here: masks are encoded on
the GPU or not at all."""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops


def to_string(counts):
    """counts uint32[m] -> the compressed `counts` bytes of a COCO RLE dict (maskApi.c:204-216)."""
    c = np.ascontiguousarray(counts, dtype=np.uint32)
    buf = C.create_string_buffer(6 * c.size + 1)
    n = _lib.lib().sln_rle_to_string(C.c_void_p(c.ctypes.data), c.size, C.cast(buf, C.c_void_p),
                                     6 * c.size + 1)
    if n < 0:
        raise RuntimeError("sln_rle_to_string failed (%d)" % n)
    return buf.raw[:n]


def from_string(s):
    """The inverse (maskApi.c:218-231): compressed bytes -> counts uint32[m]."""
    s = bytes(s)
    c = np.empty(max(len(s), 1), np.uint32)
    n = _lib.lib().sln_rle_from_string(C.c_char_p(s), len(s), C.c_void_p(c.ctypes.data), c.size)
    if n < 0:
        raise ValueError("malformed RLE string")
    return c[:n].copy()


def encode_counts(masks, max_runs=None):
    """masks: device uint8 [N,W,H] (column-major per mask, what ops.unmold_masks writes) -> list of
    uint32 count arrays.  Capacity doubles until every mask fits (a + 1 always does)."""
    if not (torch.is_tensor(masks) and masks.is_cuda):
        raise RuntimeError("mask_rle.encode needs masks on the GPU (no CPU encoder in sln_amodal_amd)")
    if masks.dim() != 3:
        raise ValueError("masks must be [N,W,H]")
    N = masks.shape[0]
    if N == 0:
        return []
    a = masks.shape[1] * masks.shape[2]
    cap = min(a + 1, int(max_runs) if max_runs else max(64, 8 * masks.shape[1]))
    while True:
        counts, num = ops.rle_encode(masks, cap)
        num_h = num.cpu().numpy()
        need = int(num_h.max())
        if need <= cap:
            break
        cap = min(a + 1, max(need, 2 * cap))
    width = max(need, 1)
    counts_h = counts[:, :width].cpu().numpy().view(np.uint32)
    return [counts_h[i, :num_h[i]].copy() for i in range(N)]


def encode(masks, max_runs=None):
    """Device masks [N,W,H] -> [{'size': [h, w], 'counts': bytes}] like pycocotools.mask.encode on
    each np.asfortranarray(mask[:, :, i])."""
    W, H = int(masks.shape[1]), int(masks.shape[2])
    return [{"size": [H, W], "counts": to_string(c)} for c in encode_counts(masks, max_runs)]


def decode_counts(counts, h, w):
    """counts -> [h,w] uint8 (maskApi.c:44-49 rleDecode; host side, used by tests and tools)."""
    counts = np.asarray(counts, dtype=np.int64)
    if int(counts.sum()) != h * w:
        raise ValueError("counts do not cover the mask")
    vals = (np.arange(counts.size) & 1).astype(np.uint8)
    return np.repeat(vals, counts).reshape(h, w, order="F")
