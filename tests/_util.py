"""Shared helpers for the parity tests (no reference code, no oracle use)."""
import os
import zlib

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def key_init_(module, scale=1.0):
    """Deterministic, name-keyed initialisation.  tools/gen_golden.py applies it
    to the reference's modules, the parity tests to this repo's modules: each
    tensor is drawn from its own generator seeded by crc32(state-dict key)."""
    with torch.no_grad():
        for key, t in list(module.state_dict().items()):
            if not t.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
            if key.endswith("running_var"):
                v = torch.rand(t.shape, generator=g) * 0.5 + 0.75
            elif key.endswith("running_mean"):
                v = (torch.rand(t.shape, generator=g) - 0.5) * 0.2
            elif t.dim() >= 2:
                fan_in = t[0].numel()
                bound = scale * (3.0 / fan_in) ** 0.5
                v = (torch.rand(t.shape, generator=g) * 2 - 1) * bound
            elif key.endswith("weight"):  # BN gamma
                v = torch.rand(t.shape, generator=g) * 0.5 + 0.75
            else:  # biases / BN beta
                v = (torch.rand(t.shape, generator=g) - 0.5) * 0.2
            t.copy_(v.to(t.device))


def seeded_dets(n, seed, span=256.0, tie_free=True):
    """Seeded NMS input [n,5] (y1,x1,y2,x2,score), scores strictly distinct."""
    rng = np.random.RandomState(seed)
    tl = rng.uniform(0, span * 0.8, size=(n, 2))
    wh = rng.uniform(2, span * 0.4, size=(n, 2))
    boxes = np.concatenate([tl, np.minimum(tl + wh, span)], axis=1)
    scores = rng.permutation(n).astype(np.float64) / max(n, 1) + rng.uniform(0, 0.5 / max(n, 1), n)
    return np.concatenate([boxes, scores[:, None]], axis=1).astype(np.float32)


E2E_RESCALE = (
    # (suffix or exact key, factor): applied on top of key_init_ so that a randomly initialised
    # 100-layer detector stays in a sane numeric range on +-128 pixel inputs (no NaN box decodes,
    # un-saturated softmaxes, proposals that overlap the ground truth) -- identical on both sides
    ("fpn.C1.0.weight", 1.0 / 64),                       # pixel scale -> O(1) activations
    ("GLM_modual.base.layer1.conv1.conv.weight", 1.0 / 64),
    (".bn3.weight", 0.3),                                # residual branches of the detector backbone
    (".increase.bn.weight", 0.3),                        # ... and of the DeepLab bottlenecks
    ("rpn.conv_class.weight", 0.2),
    ("rpn.conv_bbox.weight", 0.05),
    ("classifier.linear_bbox.weight", 0.2),
    ("GLM_modual.base.aspp.c0.weight", 0.5), ("GLM_modual.base.aspp.c1.weight", 0.5),
    ("GLM_modual.base.aspp.c2.weight", 0.5), ("GLM_modual.base.aspp.c3.weight", 0.5),
)


# added after the rescale: with all-positive (post-ReLU) features a random 2-class head votes the same
# way for every roi; centre its logit difference so that inference keeps about half the rois
E2E_SHIFT = (("classifier.linear_class.bias", (-1.17, 1.17)),)


def e2e_init_(module):
    """key_init_ + the E2E_RESCALE / E2E_SHIFT rules; used by tools/gen_golden_e2e.py on the
    reference's MaskRCNN and by the end-to-end parity tests on this repo's."""
    key_init_(module)
    with torch.no_grad():
        for key, t in module.state_dict().items():
            for pat, f in E2E_RESCALE:
                if key == pat or (pat.startswith(".") and key.endswith(pat)):
                    t.mul_(f)
            for pat, v in E2E_SHIFT:
                if key == pat:
                    t.add_(torch.tensor(v, dtype=t.dtype, device=t.device))
