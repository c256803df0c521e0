"""The reference-side bindings printed in INTEGRATION.md section 2, executed as written (the code blocks are
extracted from the markdown, only the library path is filled in) and checked against the oracle."""
import os
import re

import numpy as np
import pytest
import torch

from tests._util import seeded_dets

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def snippets():
    from sln_amodal_amd.csrc import build
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    ns = {"np": np}
    for b in blocks:
        if "def pth_nms" in b or "def crop_and_resize_gpu_forward" in b or "def encode(" in b:
            exec(b.replace("/path/to/libsln_amodal_hip.so", build.LIB), ns)
    assert {"pth_nms", "crop_and_resize_gpu_forward", "crop_and_resize_gpu_backward", "encode"} <= set(ns)
    return ns


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def test_documented_nms_binding(snippets, orc):
    dets = seeded_dets(700, seed=3, span=512.0)
    got = snippets["pth_nms"](torch.from_numpy(dets).cuda(), 0.7).cpu().numpy()
    assert np.array_equal(got, orc.nms(dets, 0.7))


def test_documented_crop_and_resize_binding(snippets, orc):
    rng = np.random.RandomState(1)
    img = rng.randn(2, 8, 20, 24).astype(np.float32)
    boxes = np.sort(rng.rand(9, 4).astype(np.float32).reshape(9, 2, 2), axis=1).reshape(9, 4)
    ind = rng.randint(0, 2, 9).astype(np.int32)
    crops = torch.empty(0, device="cuda")
    snippets["crop_and_resize_gpu_forward"](torch.from_numpy(img).cuda(), torch.from_numpy(boxes).cuda(),
                                            torch.from_numpy(ind).cuda(), 0.0, 7, 7, crops)
    assert np.array_equal(crops.cpu().numpy(), orc.crop_and_resize_fwd(img, boxes, ind, 7, 7))
    g = rng.randn(9, 8, 7, 7).astype(np.float32)
    gi = torch.zeros(2, 8, 20, 24, device="cuda")
    snippets["crop_and_resize_gpu_backward"](torch.from_numpy(g).cuda(), torch.from_numpy(boxes).cuda(),
                                             torch.from_numpy(ind).cuda(), gi)
    want = orc.crop_and_resize_bwd(g, boxes, ind, (2, 8, 20, 24))
    assert np.allclose(gi.cpu().numpy(), want, rtol=1e-5, atol=1e-5)


def test_documented_rle_binding(snippets, orc):
    yy, xx = np.mgrid[0:75, 0:101]
    mask = (((yy - 30) / 20.0) ** 2 + ((xx - 60) / 33.0) ** 2 <= 1).astype(np.uint8)
    got = snippets["encode"](np.asfortranarray(mask))
    assert got == {"size": [75, 101], "counts": orc.rle_to_string(orc.rle_encode(mask))}
