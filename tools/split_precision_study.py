"""Numerical study (CPU emulation, no GPU): how close to an fp32 GEMM are split-operand schemes on
the 16-bit MFMA pipes?  Each fp32 operand is written as a sum of low-precision parts; the listed
part products are formed exactly (fp64 here; the MFMA forms them exactly and accumulates in fp32)
and summed.  Reports the relative error of C = A @ B.T against the fp64 product, next to a plain
fp32 GEMM, for post-ReLU-like activations, gradient-like (heavy-tailed, tiny) operands and
small-magnitude operands.  bf16x3/6p is what csrc/conv.hip computes (P = 3); bf16x2/3p is P = 2;
fp16x2/3p (with and without a per-tensor power-of-two scale) is the candidate that would halve the
MFMA work of P = 3."""
import torch

torch.manual_seed(0)


def split(x, dtype, n, scale=None):
    s = 1.0
    if scale == "max":      # power of two that brings max|x| to ~2^14
        s = 2.0 ** float(torch.floor(14 - torch.log2(x.abs().max())))
    r = (x * s).float()
    parts = []
    for _ in range(n):
        h = r.to(dtype)
        parts.append(h.double())
        r = r - h.float()
    return parts, s


def gemm(A, B, dtype, n, pairs, scale=None):
    a, sa = split(A, dtype, n, scale)
    b, sb = split(B, dtype, n, scale)
    C = torch.zeros(A.shape[0], B.shape[0], dtype=torch.float64)
    for i, j in pairs:
        C += a[i] @ b[j].T
    return C / (sa * sb)


P6 = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
P3 = [(0, 0), (0, 1), (1, 0)]
cases = {
    "activations (relu(N(0,1))) x weights N(0,0.05)":
        (torch.relu(torch.randn(512, 2304)), torch.randn(256, 2304) * 0.05),
    "gradients (N(0,1)*1e-4, 1% outliers x1e3) x weights":
        (torch.randn(512, 2304) * 1e-4 * (1 + 999 * (torch.rand(512, 2304) < 0.01)), torch.randn(256, 2304) * 0.05),
    "tiny operands (N(0,1)*1e-7) x weights":
        (torch.randn(512, 2304) * 1e-7, torch.randn(256, 2304) * 0.05),
    "large operands (N(0,1)*1e5) x weights":
        (torch.randn(512, 2304) * 1e5, torch.randn(256, 2304) * 0.05),
}
for name, (A, B) in cases.items():
    ref = A.double() @ B.double().T
    den = ref.abs().mean()
    err = lambda C: float((C - ref).abs().mean() / den)
    print(name)
    print("   fp32 GEMM (fp32 accumulate)      %.2e" % err((A @ B.T).double()))
    print("   bf16 x3, 6 products  (P=3, now)  %.2e" % err(gemm(A, B, torch.bfloat16, 3, P6)))
    print("   bf16 x2, 3 products  (P=2)       %.2e" % err(gemm(A, B, torch.bfloat16, 2, P3)))
    print("   fp16 x2, 3 products, unscaled    %.2e" % err(gemm(A, B, torch.float16, 2, P3)))
    print("   fp16 x2, 3 products, max-scaled  %.2e" % err(gemm(A, B, torch.float16, 2, P3, "max")))
