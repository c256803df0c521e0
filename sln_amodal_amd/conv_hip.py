"""HIP convolution backend behind nn_ops.conv_bn_act (csrc/conv.hip).

Forward and data-gradient run on the split-bf16 implicit-GEMM kernel (the data
gradient of a stride-1 convolution is a forward convolution with mirrored,
channel-swapped weights).  Weight gradients and strided data gradients still go
through aten (MIOpen) in this revision -- DESIGN.md tracks what runs where.
"""
import torch

from . import _lib, ops

PARTS = 3          # 3 = fp32-class accuracy (6 MFMA products); 2 = ~4e-6 per layer (3 products)
_cache = {}


def supports(conv, x):
    return (x.dtype == torch.float32 and conv.groups == 1 and conv.in_channels % 8 == 0 and
            conv.weight.dtype == torch.float32 and conv.padding_mode == "zeros")


def _split(weight, flip_swap=False, parts=None):
    """[parts][O][KH][KW][I] bf16, cached per (tensor, version)."""
    parts = parts or PARTS
    key = (weight.data_ptr(), flip_swap, parts)
    hit = _cache.get(key)
    if hit is not None and hit[0] == weight._version and hit[2] == tuple(weight.shape):
        return hit[1]
    w = weight.detach()
    Co, Ci, KH, KW = w.shape
    s = w.stride()
    if flip_swap:   # data gradient: out channel <-> in channel, taps mirrored
        O, I, so, si = Ci, Co, s[1], s[0]
    else:
        O, I, so, si = Co, Ci, s[0], s[1]
    out = torch.empty((parts, O, KH, KW, I), dtype=torch.bfloat16, device=w.device)
    _lib.check(_lib.lib().sln_conv_split_weights_f32(
        ops._ptr(w), O, I, KH, KW, so, si, s[2], s[3], 1 if flip_swap else 0, parts, ops._ptr(out),
        ops._stream()), "sln_conv_split_weights_f32")
    _cache[key] = (weight._version, out, tuple(weight.shape))
    return out


def _fwd(x, wparts, Cout, KH, KW, stride, dil, pt, pl, OH, OW, scale, shift, residual, relu):
    N, Cin, H, W = x.shape
    y = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device,
                    memory_format=torch.channels_last)
    _lib.check(_lib.lib().sln_conv2d_fwd_f32(
        ops._ptr(x), N, H, W, Cin, ops._ptr(wparts), wparts.shape[0], Cout, KH, KW, stride[0],
        stride[1], dil[0], dil[1], pt, pl, OH, OW, ops._ptr(scale), ops._ptr(shift),
        ops._ptr(residual), 1 if relu else 0, ops._ptr(y), ops._stream()), "sln_conv2d_fwd_f32")
    return y


def _nhwc(t):
    """Physically NHWC view of a logical [N,C,H,W] tensor (copy only if needed)."""
    if t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous():
        return t
    return t.contiguous(memory_format=torch.channels_last)


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, bn_scale, bn_shift, residual, relu, stride, dil, pads):
        x = _nhwc(x)
        Co, Ci, KH, KW = weight.shape
        N, _, H, W = x.shape
        pt, pb, pl, pr = pads
        OH = (H + pt + pb - dil[0] * (KH - 1) - 1) // stride[0] + 1
        OW = (W + pl + pr - dil[1] * (KW - 1) - 1) // stride[1] + 1
        scale = bn_scale
        shift = bn_shift
        if bias is not None:
            shift = bias * bn_scale + bn_shift if bn_scale is not None else bias
        if shift is not None:
            shift = shift.detach().contiguous()
        if scale is not None:
            scale = scale.detach().contiguous()
        res = _nhwc(residual) if residual is not None else None
        y = _fwd(x, _split(weight), Co, KH, KW, stride, dil, pt, pl, OH, OW, scale, shift, res, relu)
        ctx.save_for_backward(x, weight, scale, y if relu else None)
        ctx.cfg = (stride, dil, pads, relu, bias is not None, residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, scale, y = ctx.saved_tensors
        stride, dil, pads, relu, has_bias, has_res = ctx.cfg
        pt, pb, pl, pr = pads
        Co, Ci, KH, KW = weight.shape
        gu = _nhwc(gy)
        if relu:
            gu = gu * (y > 0)
        g_res = gu if (has_res and ctx.needs_input_grad[5]) else None
        gz = gu * scale.view(1, -1, 1, 1) if scale is not None else gu
        gz = _nhwc(gz)
        g_bias = None
        if has_bias and ctx.needs_input_grad[2]:
            g_bias = gz.sum(dim=(0, 2, 3))
        gx = gw = None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_x and stride == (1, 1) and Co % 8 == 0:
            N, _, H, W = x.shape
            wt = _split(weight, flip_swap=True)
            gx = _fwd(gz, wt, Ci, KH, KW, (1, 1), dil, dil[0] * (KH - 1) - pt, dil[1] * (KW - 1) - pl,
                      H, W, None, None, None, False)
            need_x = False
        if need_x and KH == 1 and KW == 1 and Co % 8 == 0 and pads == (0, 0, 0, 0):
            # strided 1x1: the data gradient lives on the stride lattice, zeros elsewhere
            N, _, H, W = x.shape
            wt = _split(weight, flip_swap=True)
            OH, OW = gz.shape[2], gz.shape[3]
            small = _fwd(gz, wt, Ci, 1, 1, (1, 1), (1, 1), 0, 0, OH, OW, None, None, None, False)
            gx = torch.zeros((N, H, W, Ci), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
            gx[:, :, ::stride[0], ::stride[1]] = small
            need_x = False
        if need_x or need_w:
            if pt == pb and pl == pr:
                xin, pad = x, (pt, pl)
            else:
                xin, pad = torch.nn.functional.pad(x, (pl, pr, pt, pb)), (0, 0)
            g_in, gw, _ = torch.ops.aten.convolution_backward(
                gz, xin, weight, None, list(stride), list(pad), list(dil), False, [0, 0], 1,
                [need_x, need_w, False])
            if need_x:
                gx = g_in
                if pad == (0, 0) and (pt or pb or pl or pr):
                    gx = g_in[:, :, pt:g_in.shape[2] - pb, pl:g_in.shape[3] - pr]
        return gx, gw, g_bias, None, None, g_res, None, None, None, None


def conv_bn_act(x, conv, bn, relu, residual, pads):
    from .nn_ops import bn_affine
    scale = shift = None
    if bn is not None:
        scale, shift = bn_affine(bn)
    return _ConvFn.apply(x, conv.weight, conv.bias, scale, shift, residual, bool(relu),
                         tuple(conv.stride), tuple(conv.dilation), tuple(pads))
