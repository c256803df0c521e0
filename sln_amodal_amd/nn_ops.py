"""Functional building blocks of the conv stacks.

`conv_bn_act` is the single dispatch point of every convolution on the hot path
(backbone, FPN, RPN, heads, GLM): conv (+bias) -> frozen-BN affine -> (+residual)
-> ReLU.  All BatchNorm on the path is frozen and in eval mode
(model.py:192-197, 525-531 of the reference), i.e. a per-channel affine, so it is
folded into the convolution's epilogue.

Backends
  "hip"   : hand-written implicit-GEMM MFMA kernels (csrc/conv*.hip) through the
            C ABI, NHWC activations.  Raises if the library lacks them.
  "torch" : torch.nn.functional ops on channels-last tensors (MIOpen underneath).
            Interim backend for layer shapes the HIP kernels do not cover yet;
            DESIGN.md lists which layers run where.
"""
import torch
import torch.nn.functional as F

BACKEND = "auto"  # "auto" | "hip" | "torch"


def _hip_conv():
    try:
        from . import conv_hip
    except ImportError:
        return None
    return conv_hip


def same_pad(in_size, kernel, stride):
    """TensorFlow 'SAME' padding amounts (before, after) -- SamePad2d,
    modal/modals.py:169-181."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + kernel - in_size, 0)
    before = total // 2
    return before, total - before


def bn_affine(bn):
    """Frozen BatchNorm2d -> (scale, shift) per channel, fp32."""
    scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    return scale, shift


def conv_bn_act(x, conv, bn=None, relu=False, residual=None, same=False):
    """x [B,C,H,W] (any memory format; channels-last preferred).
    conv: nn.Conv2d parameter holder (weight, bias, stride, padding, dilation).
    bn:   frozen nn.BatchNorm2d or None.  same: apply SamePad2d first."""
    stride, dilation = conv.stride, conv.dilation
    kh, kw = conv.kernel_size
    if same:
        pt, pb = same_pad(x.shape[2], kh, stride[0])
        pl, pr = same_pad(x.shape[3], kw, stride[1])
    else:
        pt = pb = conv.padding[0]
        pl = pr = conv.padding[1]
    hip = _hip_conv() if BACKEND in ("auto", "hip") else None
    if hip is not None and x.is_cuda and hip.supports(conv, x):
        return hip.conv_bn_act(x, conv, bn, relu, residual, (pt, pb, pl, pr))
    if BACKEND == "hip" and conv.in_channels >= 8:   # (3-channel stems are aten by design)
        raise RuntimeError("HIP conv backend requested but unavailable for this layer")
    if pt == pb and pl == pr:
        y = F.conv2d(x, conv.weight, conv.bias, stride, (pt, pl), dilation)
    else:
        y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), conv.weight, conv.bias, stride, 0, dilation)
    if bn is not None:
        y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return y


def max_pool_same(x, kernel, stride):
    """SamePad2d + MaxPool2d (modal/modals.py:316-317).  Zero padding is safe: the
    input is post-ReLU."""
    pt, pb = same_pad(x.shape[2], kernel, stride)
    pl, pr = same_pad(x.shape[3], kernel, stride)
    return F.max_pool2d(F.pad(x, (pl, pr, pt, pb)), kernel, stride)


def upsample2x_add(lateral, top):
    """FPN merge: lateral + nearest-2x(top)  (modal/modals.py:243-246)."""
    return lateral + F.interpolate(top, scale_factor=2, mode="nearest")
