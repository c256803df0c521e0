"""What an operand of the two-part scaled fp16 format really carries, element by element, THROUGH the matrix
instruction: a 1x1 convolution with the identity as its weight returns (h0 + h1) / s of every input element -- the
split kernel's parts as v_mfma_f32_16x16x32_f16 reads them (fp16 subnormals included: if the instruction flushed
subnormal inputs, every element below 2^-13 of the tensor's maximum would keep 11 bits instead of 22).  Elements are
spread over 40 binades below the tensor's maximum; prints, per binade, the worst and the median relative error.

    python tools/operand_probe.py [--headroom K]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def probe(parts=2, role="x", C=256, n=4096, seed=0):
    """-> [(binade below the maximum, worst relative error, median relative error)]."""
    from sln_amodal_amd import conv_hip
    g = torch.Generator(device="cuda").manual_seed(seed)
    e = torch.randint(0, 40, (n, C), device="cuda", generator=g)
    m = (1.0 + torch.rand(n, C, device="cuda", generator=g)) * torch.exp2(-e.float())
    m = m * (torch.randint(0, 2, (n, C), device="cuda", generator=g) * 2 - 1)
    m[0, 0] = 1.99            # the tensor's maximum
    x = m.view(1, n, 1, C).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)     # [1, C, n, 1]
    w = torch.eye(C, device="cuda").view(C, C, 1, 1).contiguous()
    old = conv_hip.PARTS
    conv_hip.PARTS = parts
    try:
        wl = w.clone().requires_grad_(role != "x")
        if role == "x":         # the element is the ACTIVATION operand of a forward pass
            for _ in range(2):
                conv_hip.update_scales(sync=False)
                y = conv_hip._ConvFn.apply(x, wl, None, None, None, None, False, (1, 1), (1, 1), (0, 0, 0, 0))
            got = y.permute(0, 2, 3, 1).reshape(n, C)
        else:                   # ... the GRADIENT operand of a data gradient (its own scale role, extra head room)
            for _ in range(2):
                conv_hip.update_scales(sync=False)
                xl = torch.zeros_like(x).requires_grad_(True)
                y = conv_hip._ConvFn.apply(xl, wl, None, None, None, None, False, (1, 1), (1, 1), (0, 0, 0, 0))
                y.backward(x)
            got = xl.grad.permute(0, 2, 3, 1).reshape(n, C)
    finally:
        conv_hip.PARTS = old
    err = ((got.double() - m.double()).abs() / m.double().abs())
    rows = []
    for b in range(40):
        sel = e == b
        if bool(sel.any()):
            rows.append((b, float(err[sel].max()), float(err[sel].median())))
    return rows


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--headroom", type=int, default=None)
    args = ap.parse_args()
    from sln_amodal_amd import conv_hip, nn_ops
    nn_ops.BACKEND = "hip"
    if args.headroom is not None:
        conv_hip.GRAD_HEADROOM_LOG2 = args.headroom
    for parts, role in ((2, "x"), (2, "gz"), (3, "x")):
        print("parts %d, operand role %s (gradient head room 2^%d): binade below the maximum, worst / median relative error"
              % (parts, role, conv_hip.GRAD_HEADROOM_LOG2))
        for b, worst, md in probe(parts, role):
            print("  2^-%-2d  %9.2e  %9.2e" % (b, worst, md))


if __name__ == "__main__":
    main()
