"""The control under "fp32-class" (VERDICT r5 #2): for the train step's real contraction lengths, the error against an
fp64 reference of the same fp32 inputs of
    HIP, 2 x scaled fp16 parts (the default format, 3 MFMA products per multiply-add),
    HIP, 3 x bf16 parts (the strict format, 6 products),
    aten's fp32 convolution (MIOpen) on the same device,
for the forward pass, the data gradient and the weight gradient of each layer -- on Gaussian operands and on operands
shaped like the network's (post-ReLU activations with per-channel spread; sparse, heavy-tailed gradients whose typical
element lies far below the tensor's maximum: that is where a per-tensor scaled fp16 pair, whose second part reaches its
exponent floor 2^-25 of the scaled range, can fall behind fp32's 24 bits per ELEMENT).

    python tools/precision_control.py [--json out.json] [--headroom K]

prints the table; tests/test_precision_gpu.py asserts on the same function.  Reference layers: modal/modals.py:264-355
(bottlenecks), 361-412 (RPN), 419-453 (classifier FC), 457-499 (mask head), modal/deeplabv2.py:16-45 (ASPP)."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# name, Cin, Cout, k, dil, H, W, N   (K = Cin k^2; spatial sizes an fp64 reference finishes in seconds)
SHAPES = [
    ("C2 3x3 64->64 (K 576)", 64, 64, 3, 1, 128, 128, 4),
    ("C4 1x1 1024->256 (K 1024)", 1024, 256, 1, 1, 64, 64, 2),
    ("FPN / RPN 3x3 256->256 (K 2304)", 256, 256, 3, 1, 64, 64, 2),
    ("mask conv1 3x3 439->256 (K 3951)", 439, 256, 3, 1, 16, 16, 24),
    ("C5 3x3 512->512 (K 4608)", 512, 512, 3, 1, 32, 32, 2),
    ("3x3 1024->256 d2 (K 9216)", 1024, 256, 3, 2, 33, 33, 1),
    ("classifier FC 7x7 256->1024 (K 12544)", 256, 1024, 7, 1, 7, 7, 64),
    ("ASPP 3x3 2048->182 d12 (K 18432)", 2048, 182, 3, 12, 33, 33, 1),
]


def make_operands(shape, kind, seed=0):
    """-> x [N,Cin,H,W] channels-last, w [Cout,Cin,k,k], upstream gradient generator(y_shape)."""
    name, Cin, Cout, k, dil, H, W, N = shape
    g = torch.Generator(device="cuda").manual_seed(1000 * seed + Cin + 7 * Cout + k)
    x = torch.randn(N, Cin, H, W, device="cuda", generator=g)
    w = torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5
    if kind == "net":
        # post-ReLU activations, channels of different scale (frozen-BN affine outputs)
        x = F.relu(x) * torch.exp(0.7 * torch.randn(1, Cin, 1, 1, device="cuda", generator=g))
    x = x.contiguous(memory_format=torch.channels_last)

    def upstream(yshape):
        up = torch.randn(yshape, device="cuda", generator=g)
        if kind == "net":
            # a masked gradient (ReLU pattern of the layer's output), heavy-tailed per element and per pixel: a few
            # positive-roi pixels carry gradients thousands of times the typical one
            up = up * torch.exp(1.5 * torch.randn(yshape, device="cuda", generator=g))
            up = up * torch.exp(2.0 * torch.randn((yshape[0], 1) + tuple(yshape[2:]), device="cuda", generator=g))
            up = up * (torch.rand(yshape, device="cuda", generator=g) < 0.5)
        return up.contiguous(memory_format=torch.channels_last)
    return x, w, upstream


def _pads(k, dil):
    if k == 7:
        return (0, 0, 0, 0)
    p = dil * (k - 1) // 2
    return (p, p, p, p)


def run_layer(mode, x, w, up, k, dil):
    """mode "p2" / "p3": conv_hip._ConvFn in that operand format (steady state: the second pass after an
    update_scales()); "aten": F.conv2d in fp32; "ref": fp64.  -> (y, dx, dw) as float64 tensors."""
    from sln_amodal_amd import conv_hip
    pads = _pads(k, dil)
    pt, pb, pl, pr = pads
    if mode in ("aten", "ref"):
        dt = torch.float64 if mode == "ref" else torch.float32
        xl, wl = x.to(dt).requires_grad_(True), w.to(dt).requires_grad_(True)
        y = F.conv2d(F.pad(xl, (pl, pr, pt, pb)), wl, None, 1, 0, dil)
        y.backward(up.to(dt))
        return y.detach().double(), xl.grad.double(), wl.grad.double()
    old = conv_hip.PARTS
    conv_hip.PARTS = {"p2": 2, "p3": 3}[mode]
    try:
        wl = w.clone().requires_grad_(True)      # (the scale slots live on this object: kept over both passes)
        for _ in range(2):
            conv_hip.update_scales(sync=False)
            xl = x.clone().requires_grad_(True)
            wl.grad = None
            y = conv_hip._ConvFn.apply(xl, wl, None, None, None, None, False, (1, 1), (dil, dil), pads)
            y.backward(up)
        return y.detach().double(), xl.grad.double(), wl.grad.double()
    finally:
        conv_hip.PARTS = old


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def med(a, b):
    """MEDIAN over the elements of |a - b| / |b| (b != 0): the relative L2 norm above is carried by a tensor's largest
    elements; this one says what a TYPICAL element keeps -- where a per-tensor scaled format would show its exponent
    floor on heavy-tailed tensors."""
    nz = b != 0
    if not bool(nz.any()):
        return 0.0
    q = ((a - b).abs()[nz] / b.abs()[nz]).flatten()
    if q.numel() > (1 << 22):
        q = q[:: q.numel() // (1 << 22)]
    return float(q.median())


def table(kinds=("gauss", "net"), shapes=SHAPES, modes=("p2", "p3", "aten")):
    """-> rows {shape, kind, what, K, p2, p3, aten, p2_over_aten}: relative L2 error against fp64."""
    rows = []
    for shape in shapes:
        name, Cin, Cout, k, dil, H, W, N = shape
        for kind in kinds:
            x, w, upstream = make_operands(shape, kind)
            ref = None
            outs, meds = {}, {}
            for mode in ("ref",) + tuple(modes):
                if mode == "ref":
                    pt = _pads(k, dil)[0]
                    oh = H + 2 * pt - dil * (k - 1)
                    up = upstream((N, Cout, oh, oh + (W - H)))
                res = run_layer(mode, x, w, up, k, dil)
                if mode == "ref":
                    ref = res
                else:
                    outs[mode] = [rel(a, b) for a, b in zip(res, ref)]
                    meds[mode] = [med(a, b) for a, b in zip(res, ref)]
                del res
            kk = {"forward": Cin * k * k, "data gradient": Cout * k * k, "weight gradient": N * up.shape[2] * up.shape[3]}
            for i, what in enumerate(("forward", "data gradient", "weight gradient")):
                row = {"shape": name, "kind": kind, "what": what, "K": kk[what]}
                row.update({m: outs[m][i] for m in modes})
                row.update({m + "_med": meds[m][i] for m in modes})
                if "aten" in outs and "p2" in outs:
                    row["p2_over_aten"] = outs["p2"][i] / max(outs["aten"][i], 1e-300)
                rows.append(row)
            del x, w, up, ref
            torch.cuda.empty_cache()
    return rows


def fmt(rows):
    lines = ["%-40s %-6s %-16s %8s  %9s %9s %9s  %s" % ("layer", "input", "pass", "K", "HIP 2xf16", "HIP 3xbf16", "aten f32",
                                                         "2xf16 / aten")]
    for r in rows:
        lines.append("%-40s %-6s %-16s %8d  %9.2e %9.2e %9.2e  %5.2f" % (
            r["shape"], r["kind"], r["what"], r["K"], r.get("p2", float("nan")), r.get("p3", float("nan")),
            r.get("aten", float("nan")), r.get("p2_over_aten", float("nan"))))
    if rows and "p2_med" in rows[0]:
        lines.append("")
        lines.append("median over the elements of |error| / |value| (what a typical element keeps):")
        lines.append("%-40s %-6s %-16s  %9s %9s %9s  %s" % ("layer", "input", "pass", "HIP 2xf16", "HIP 3xbf16", "aten f32",
                                                          "2xf16 / aten"))
        for r in rows:
            lines.append("%-40s %-6s %-16s  %9.2e %9.2e %9.2e  %5.2f" % (
                r["shape"], r["kind"], r["what"], r.get("p2_med", float("nan")), r.get("p3_med", float("nan")),
                r.get("aten_med", float("nan")), r.get("p2_med", 0.0) / max(r.get("aten_med", 0.0), 1e-300)))
    return "\n".join(lines)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--headroom", type=int, default=None, help="conv_hip.GRAD_HEADROOM_LOG2 for this run (default: the product's)")
    args = ap.parse_args()
    from sln_amodal_amd import conv_hip, nn_ops
    nn_ops.BACKEND = "hip"
    if args.headroom is not None:
        conv_hip.GRAD_HEADROOM_LOG2 = args.headroom
    rows = table()
    print("GRAD_HEADROOM_LOG2 = %d" % conv_hip.GRAD_HEADROOM_LOG2)
    print(fmt(rows))
    if args.json:
        json.dump({"grad_headroom_log2": conv_hip.GRAD_HEADROOM_LOG2, "rows": rows}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
