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
import os

import weakref

import torch
import torch.nn.functional as F

BACKEND = "auto"  # "auto" | "hip" | "torch"
PARTS_ONLY = os.environ.get("SLN_PARTS_ONLY", "1") != "0"   # A/B switch of conv_bn_act(parts_only=True)
CALIBRATING = None   # [count] while synthetic._calibrate runs: conv_bn_act refreshes each frozen BN's statistics
IN_CALIBRATED_BN = False   # True around conv_bn_act's own F.batch_norm call of a statistics pass


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


CACHE_BN_AFFINE = os.environ.get("SLN_CACHE_BN_AFFINE", "1") != "0"
BN_CACHE_STATS = [0, 0]   # hits, misses


def bn_affine(bn):
    """Frozen BatchNorm2d -> (scale, shift) per channel, fp32.  With frozen affine
    parameters (the whole hot path: model.py:192-197 of the reference) the pair only
    changes when a checkpoint is loaded, so it is cached on the module, keyed by the
    version counters of its four tensors.  (In-place edits made through `.data` do not
    bump those counters: delete `bn._sln_affine` after such an edit.)"""
    frozen = CACHE_BN_AFFINE and not (bn.weight.requires_grad or bn.bias.requires_grad)
    if frozen:
        key = (bn.weight._version, bn.bias._version, bn.running_mean._version,
               bn.running_var._version, bn.weight.data_ptr(), bn.running_var.data_ptr())
        hit = getattr(bn, "_sln_affine", None)
        if hit is not None and hit[0] == key:
            BN_CACHE_STATS[0] += 1
            return hit[1], hit[2]
        BN_CACHE_STATS[1] += 1
    with torch.set_grad_enabled(not frozen):
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
    if frozen:
        bn._sln_affine = (key, scale, shift)
    return scale, shift


def conv_bn_act(x, conv, bn=None, relu=False, residual=None, same=False, link=None, chain_in=None,
                chain_out=None, parts_only=False, pair=None, grad_inbox=None, pool_handoff=None):
    """x [B,C,H,W] (any memory format; channels-last preferred).
    conv: nn.Conv2d parameter holder (weight, bias, stride, padding, dilation).
    bn:   frozen nn.BatchNorm2d or None.  same: apply SamePad2d first.
    link: dict shared by the two convs of an identity-shortcut block that see the same x
          (one as input, one as residual); HIP backend only, see conv_hip._ConvFn.
    chain_out / chain_in: dict shared by a conv (chain_out) and the ONLY conv that reads its
          output (chain_in): lets the reader's backward prepare this layer's gradient.
    parts_only: the output is read by convolutions, as the next shortcut and as a ReLU mask only (a bottleneck's
          convolutions): no fp32 copy is written, the result is a placeholder that carries the parts
          (conv_hip.parts_only_of / materialize).  HIP backend, fp16 x 2 operands; ignored elsewhere.
    pair: dict shared by the two strided 1x1 convs that read the same x (a stage's first block).
    grad_inbox: dict in which ANOTHER reader of x leaves its gradient w.r.t. x during the backward pass
          (conv_hip.GradInbox): this conv's data gradient adds it in its epilogue and returns the sum, the other
          reader returns nothing -- no accumulation pass by autograd.  HIP backend only; ignored elsewhere.
    pool_handoff: dict shared by a 3-channel stem conv and the max-pool that is its output's only reader
          (max_pool_same(..., handoff=)): the pool's backward leaves its incoming gradient there and the stem's
          gradient preparation gathers from it (no pool-backward launch, no fp32 gradient of the stem output)."""
    if CALIBRATING is not None and bn is not None:
        # statistics pass (synthetic.calibrate_*): the raw convolution on the same backend, its output's
        # statistics into the frozen BN, then the un-fused normalisation / shortcut / ReLU
        y = conv_bn_act(x, conv, None, False, None, same)
        with torch.no_grad():
            bn.running_mean.copy_(y.mean(dim=(0, 2, 3)))
            bn.running_var.copy_(y.var(dim=(0, 2, 3), unbiased=False).clamp(min=1e-6))
        CALIBRATING[0] += 1
        global IN_CALIBRATED_BN
        IN_CALIBRATED_BN = True      # (this BN has its statistics already: synthetic._calibrate's patch skips it)
        try:
            y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        finally:
            IN_CALIBRATED_BN = False
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    stride, dilation = conv.stride, conv.dilation
    kh, kw = conv.kernel_size
    if same:   # (never with a MultiScale: the GLM uses symmetric padding)
        pt, pb = same_pad(x.shape[2], kh, stride[0])
        pl, pr = same_pad(x.shape[3], kw, stride[1])
    else:
        pt = pb = conv.padding[0]
        pl = pr = conv.padding[1]
    hip = _hip_conv() if BACKEND in ("auto", "hip") else None
    if hip is not None and isinstance(x, hip.MultiScale):   # all GLM scales in one launch
        return hip.conv_bn_act_ms(x, conv, bn, relu, residual, (pt, pb, pl, pr),
                                  parts_only=parts_only and PARTS_ONLY)
    if hip is not None and x.is_cuda and hip.supports(conv, x):
        if kh * kw > 1 and (kh, kw) == tuple(x.shape[2:]) and (pt, pb, pl, pr) == (0, 0, 0, 0) and \
                residual is None and x.shape[1] % 8 == 0:
            # Whole-window convolution (the classifier's 7x7 "FC" conv on 7x7 crops, modals.py:441): one
            # output pixel, so it IS a linear layer over K = KH*KW*C -- the NHWC crop is that row already.
            # As a KxK convolution its data gradient would visit KH*KW output positions with KH*KW taps
            # each, all but one of them outside the 1x1 gradient map: 49x the work.
            N, C = x.shape[0], x.shape[1]
            xr = x.permute(0, 2, 3, 1).reshape(N, 1, 1, kh * kw * C).permute(0, 3, 1, 2)
            w2 = conv.weight.permute(0, 2, 3, 1).reshape(conv.out_channels, kh * kw * C, 1, 1)
            return hip.conv_bn_act(xr, conv, bn, relu, None, (0, 0, 0, 0), weight=w2, stride=(1, 1))
        return hip.conv_bn_act(x, conv, bn, relu, residual, (pt, pb, pl, pr), link=link,
                               chain_in=chain_in, chain_out=chain_out, pair=pair,
                               parts_only=parts_only and PARTS_ONLY, grad_inbox=grad_inbox)
    if hip is not None and not isinstance(x, hip.MultiScale) and x.is_cuda and conv.groups > 1 and \
            (kh, kw) == (3, 3) and (pt, pb, pl, pr) == (1, 1, 1, 1) and tuple(dilation) == (1, 1) and \
            conv.bias is None and residual is None and stride[0] == stride[1] and \
            conv.in_channels == conv.out_channels and (conv.in_channels // conv.groups) in (4, 8, 16, 32):
        # ResNeXt's grouped 3x3 (BASELINE.json configs[4]; reference modal/resnext.py:36)
        from . import ops
        scale = shift = None
        if bn is not None:
            scale, shift = bn_affine(bn)
            if torch.is_grad_enabled() and (scale.requires_grad or shift.requires_grad):
                raise RuntimeError("grouped 3x3 convolution: the HIP path takes a FROZEN BatchNorm (as the whole hot "
                                   "path does); set nn_ops.BACKEND = 'torch' to train the normalisation on aten")
        if hip.grouped_supported(conv, x):          # conv_hip.PARTS = 1: the fp16 MFMA kernels (configs[4] as stated)
            return hip.grouped_conv_bn_act(x, conv, scale, shift, relu, parts_only=parts_only and PARTS_ONLY,
                                           chain_in=chain_in, chain_out=chain_out)
        if getattr(x, "_sln_po", None) is not None:
            raise RuntimeError("grouped 3x3 (fp32 direct kernels): the input exists as parts only; its producer must "
                               "not be asked for parts_only outside conv_hip.PARTS = 1")
        if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
            return ops.GroupedConv3x3.apply(x, conv.weight, scale, shift, bool(relu), conv.groups, stride[0])
        return ops.grouped_conv3x3(x, conv.weight, conv.groups, stride[0], scale, shift, relu)
    if hip is not None and not isinstance(x, hip.MultiScale) and hip.is_stem(conv, x) and residual is None:
        return hip.stem_conv_bn_act(x, conv, bn, relu, (pt, pb, pl, pr), pool_handoff)   # 3-channel 7x7/2 stems
    if BACKEND != "torch" and x.is_cuda:
        # no silent aten/MIOpen fallback on the GPU
        raise RuntimeError("conv %s -> %s k%s groups=%d dtype=%s has no HIP path (nn_ops.BACKEND=%r); "
                           "set BACKEND='torch' explicitly to run it on aten" %
                           (conv.in_channels, conv.out_channels, tuple(conv.kernel_size), conv.groups,
                            x.dtype, BACKEND))
    if pt == pb and pl == pr:
        y = F.conv2d(x, conv.weight, conv.bias, stride, (pt, pl), dilation, conv.groups)
    else:
        y = F.conv2d(F.pad(x, (pl, pr, pt, pb)), conv.weight, conv.bias, stride, 0, dilation, conv.groups)
    if bn is not None:
        y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return y


FORCE_POOL_ARG = None   # parity tests only: callable(arg [N,OH,OW,C] uint8) -> replacement winning taps or None


class _MaxPoolFn(torch.autograd.Function):
    """Clipped-window max-pool on NHWC maps (csrc/maxpool.hip): forward keeps the winning tap per output,
    backward gathers -- every input gradient written once, no memset."""

    @staticmethod
    def forward(ctx, x, kernel, stride, pad_top, pad_left, OH, OW, handoff=None):
        from . import _lib, ops
        ctx.handoff = handoff
        xc = x if x.is_contiguous(memory_format=torch.channels_last) else \
            x.contiguous(memory_format=torch.channels_last)
        N, C, H, W = xc.shape
        y = torch.empty((N, C, OH, OW), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        arg = torch.empty((N, OH, OW, C), dtype=torch.uint8, device=x.device)
        _lib.check(_lib.lib().sln_maxpool_fwd_f32(ops._ptr(xc), N, H, W, C, kernel, stride, pad_top, pad_left, OH, OW,
                                                  ops._ptr(y), ops._ptr(arg), ops._stream()), "sln_maxpool_fwd_f32")
        if FORCE_POOL_ARG is not None:
            forced = FORCE_POOL_ARG(arg)
            if forced is not None:
                arg = forced.to(torch.uint8).contiguous()
        ctx.save_for_backward(arg)
        ctx.cfg = (N, C, H, W, kernel, stride, pad_top, pad_left, OH, OW)
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib, ops
        (arg,) = ctx.saved_tensors
        N, C, H, W, kernel, stride, pad_top, pad_left, OH, OW = ctx.cfg
        gc = g if g.is_contiguous(memory_format=torch.channels_last) else \
            g.contiguous(memory_format=torch.channels_last)
        if ctx.handoff is not None and ctx.handoff.get("armed"):
            # the producing stem conv gathers from the pooled gradient itself (conv_hip._grad_prep_pooled): what
            # goes back through autograd is a zero-stride placeholder that nothing reads
            from . import conv_hip
            ctx.handoff["pooled"] = (gc, arg, ctx.cfg)
            return conv_hip._dummy_grad(g.device).expand(N, C, H, W), None, None, None, None, None, None, None
        gx = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().sln_maxpool_bwd_f32(ops._ptr(gc), ops._ptr(arg), N, H, W, C, kernel, stride, pad_top,
                                                  pad_left, OH, OW, ops._ptr(gx), ops._stream()), "sln_maxpool_bwd_f32")
        return gx, None, None, None, None, None, None, None


def _hip_pool_ok(x):
    return BACKEND != "torch" and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0


def max_pool_same(x, kernel, stride, handoff=None):
    """SamePad2d + MaxPool2d (modal/modals.py:316-317).  Zero padding is safe: the
    input is post-ReLU.  handoff: see conv_bn_act(pool_handoff=)."""
    pt, pb = same_pad(x.shape[2], kernel, stride)
    pl, pr = same_pad(x.shape[3], kernel, stride)
    H, W = x.shape[2], x.shape[3]
    if _hip_pool_ok(x) and pt < kernel and pl < kernel:
        # a window clipped at the border equals the zero-padded one for the non-negative input
        OH = (H + pt + pb - kernel) // stride + 1
        OW = (W + pl + pr - kernel) // stride + 1
        return _MaxPoolFn.apply(x, kernel, stride, pt, pl, OH, OW, handoff)
    if pt == 0 and pl == 0 and pb < kernel and pr < kernel and \
            -(-(H - kernel) // stride) + 1 == -(-H // stride) and -(-(W - kernel) // stride) + 1 == -(-W // stride):
        # bottom/right padding only (the even sizes of the path): a clipped last window equals a
        # zero-padded one for the non-negative input, and ceil_mode saves the padded copy of the map
        return F.max_pool2d(x, kernel, stride, padding=0, ceil_mode=True)
    return F.max_pool2d(F.pad(x, (pl, pr, pt, pb)), kernel, stride)


def max_pool_ceil(x, kernel, stride, padding):
    """nn.MaxPool2d(kernel, stride, padding, ceil_mode=True) (the GLM stem, modal/resnet_deeplab.py)."""
    if _hip_pool_ok(x) and padding < kernel:
        def out(n):     # torch's ceil-mode output size: the last window must start inside the (left-padded) map
            o = -(-(n + 2 * padding - kernel) // stride) + 1
            return o - 1 if (o - 1) * stride >= n + padding else o
        return _MaxPoolFn.apply(x, kernel, stride, padding, padding, out(x.shape[2]), out(x.shape[3]))
    return F.max_pool2d(x, kernel, stride, padding, ceil_mode=True)


def max_pool_pad(x, kernel, stride, padding):
    """nn.MaxPool2d(kernel, stride, padding) (floor mode) on a non-negative map (ResNeXt's stem pool,
    modal/resnext.py:83): the window clipped at the border equals the -inf padded one."""
    if _hip_pool_ok(x) and padding < kernel:
        OH = (x.shape[2] + 2 * padding - kernel) // stride + 1
        OW = (x.shape[3] + 2 * padding - kernel) // stride + 1
        return _MaxPoolFn.apply(x, kernel, stride, padding, padding, OH, OW)
    return F.max_pool2d(x, kernel, stride, padding)


class _FpnMerge(torch.autograd.Function):
    """lateral + nearest-2x(top) as one HIP pass; backward: the lateral's gradient is the incoming one,
    the coarse map's is its 2x2 sum-pool."""

    @staticmethod
    def forward(ctx, lateral, top):
        from . import _lib, ops
        lat = lateral if lateral.is_contiguous(memory_format=torch.channels_last) else \
            lateral.contiguous(memory_format=torch.channels_last)
        tp = top if top.is_contiguous(memory_format=torch.channels_last) else \
            top.contiguous(memory_format=torch.channels_last)
        N, C, h, w = tp.shape
        out = torch.empty(lat.shape, dtype=torch.float32, device=lat.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().sln_upsample2x_add_f32(ops._ptr(lat), ops._ptr(tp), N, h, w, C, ops._ptr(out),
                                                     ops._stream()), "sln_upsample2x_add_f32")
        ctx.top_shape = tuple(tp.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib, ops
        gtop = None
        if ctx.needs_input_grad[1]:
            gc = g if g.is_contiguous(memory_format=torch.channels_last) else \
                g.contiguous(memory_format=torch.channels_last)
            N, C, h, w = ctx.top_shape
            gtop = torch.empty(ctx.top_shape, dtype=torch.float32, device=g.device,
                               memory_format=torch.channels_last)
            _lib.check(_lib.lib().sln_sumpool2x2_f32(ops._ptr(gc), N, h, w, C, ops._ptr(gtop), ops._stream()),
                       "sln_sumpool2x2_f32")
        return (g if ctx.needs_input_grad[0] else None), gtop


_PAIRS = weakref.WeakKeyDictionary()      # first head module -> (key, second head, concatenated weight, _PairConv)


class _PairConv(object):
    """What conv_hip.conv_bn_act reads of a convolution, for two sibling 1x1 heads run as ONE layer (conv_pair)."""
    __slots__ = ("weight", "bias", "stride", "dilation")

    def __init__(self, owner, bias):
        self.weight, self.bias, self.stride, self.dilation = owner, bias, (1, 1), (1, 1)


def pair_owner(a):
    """The tensor the fused layer's scale slots live on (after conv_pair ran), else None."""
    c = _PAIRS.get(a)
    return None if c is None else c[3].weight


def conv_pair_supported(x, a, b):
    hip = _hip_conv() if BACKEND in ("auto", "hip") else None
    return (hip is not None and x.is_cuda and CALIBRATING is None and hip.PARTS == 2 and
            a.kernel_size == (1, 1) and b.kernel_size == (1, 1) and a.stride == (1, 1) and b.stride == (1, 1) and
            a.padding == (0, 0) and b.padding == (0, 0) and a.groups == 1 and b.groups == 1 and
            a.in_channels == b.in_channels and a.in_channels % 8 == 0 and (a.bias is None) == (b.bias is None))


def conv_pair(x, a, b, chain_in=None):
    """Two pointwise convolutions that read the same x (the RPN's class and box heads, modals.py:361-412), as ONE
    layer with the concatenated weights: x's parts are read once instead of twice forward and in the weight gradient,
    and ONE data gradient w.r.t. x is produced (no fp32 partial gradient written by one head and re-read by the
    other).  Returns [B, Ca + Cb, H, W]; the caller slices.  The concatenation is an autograd op: each parameter
    receives its rows of the weight gradient.  HIP backend, fp16 x 2 operands (conv_pair_supported)."""
    hip = _hip_conv()
    key = (hip.SCALE_EPOCH[0], a.weight._version, b.weight._version, torch.is_grad_enabled(), x.device,
           None if a.bias is None else (a.bias._version, b.bias._version))
    c = _PAIRS.get(a)
    if c is None or c[0] != key or c[1]() is not b:
        w = torch.cat([a.weight, b.weight], 0)
        bias = None if a.bias is None else torch.cat([a.bias, b.bias], 0)
        own = None if c is None else c[3].weight
        if own is None or own.device != x.device or tuple(own.shape) != tuple(w.shape):
            # the fused layer's scale slots live on a tensor of its own (never on a head's weight: the heads may also
            # run alone, with other tensor roles)
            own = torch.empty(w.shape, device=x.device)
        c = _PAIRS[a] = (key, weakref.ref(b), w, _PairConv(own, bias))      # (outside the module: it stays deep-copyable)
    return hip.conv_bn_act(x, c[3], None, False, None, (0, 0, 0, 0), weight=c[2], chain_in=chain_in)


def upsample2x_add(lateral, top):
    """FPN merge: lateral + nearest-2x(top)  (modal/modals.py:243-246)."""
    if BACKEND != "torch" and lateral.is_cuda and lateral.dtype == torch.float32 and lateral.shape[1] % 4 == 0 \
            and lateral.shape[2] == 2 * top.shape[2] and lateral.shape[3] == 2 * top.shape[3]:
        return _FpnMerge.apply(lateral, top)
    return lateral + F.interpolate(top, scale_factor=2, mode="nearest")


def deconv2x2_relu(x, deconv):
    """ConvTranspose2d(kernel 2, stride 2) + ReLU of the mask head (modal/modals.py:494-495)
    on the HIP conv stack: every output pixel (2i+a, 2j+b) depends on one input pixel, so
    the layer is a 1x1 convolution to 4*Cout channels (rows ordered (a, b, cout)) followed
    by a depth-to-space shuffle; bias and ReLU commute with the shuffle and are fused in
    the conv epilogue.  Autograd sees plain views of `deconv.weight`, so the weight
    gradient arrives in the parameter's own [Cin, Cout, 2, 2] layout."""
    hip = _hip_conv() if BACKEND in ("auto", "hip") else None
    if hip is None or not x.is_cuda:
        return F.relu(F.conv_transpose2d(x, deconv.weight, deconv.bias, stride=2))
    Ci, Co = deconv.weight.shape[0], deconv.weight.shape[1]
    w2 = deconv.weight.permute(2, 3, 1, 0).reshape(4 * Co, Ci, 1, 1)
    b2 = deconv.bias.repeat(4) if deconv.bias is not None else None
    y = hip._ConvFn.apply(x, w2, b2, None, None, None, True, (1, 1), (1, 1), (0, 0, 0, 0), None, None, None,
                          deconv.weight)
    N, _, H, W = y.shape
    y = y.permute(0, 2, 3, 1).reshape(N, H, W, 2, 2, Co).permute(0, 1, 3, 2, 4, 5)
    return y.reshape(N, 2 * H, 2 * W, Co).permute(0, 3, 1, 2)      # logical NCHW, NHWC in memory


CHAIN_DECONV = os.environ.get("SLN_CHAIN_DECONV", "1") != "0"      # A/B switch


def deconv2x2_relu_conv1x1(x, deconv, conv):
    """The mask head's tail (modal/modals.py:494-497): ConvTranspose2d(2, stride 2) + ReLU + 1x1 conv to
    the class logits.  A pointwise convolution commutes with the depth-to-space shuffle, so it runs on the
    un-shuffled deconv output -- [N, H, W, (a, b), Cout] is [N*H*W*4 pixels, Cout] as it lies in memory,
    parts included -- and only the few logit channels are shuffled: the 1.7-GB activation is neither
    permuted (forward and backward) nor split a second time."""
    hip = _hip_conv() if BACKEND in ("auto", "hip") else None
    if hip is None or not x.is_cuda or tuple(conv.kernel_size) != (1, 1) or tuple(conv.stride) != (1, 1) or \
            deconv.weight.shape[1] % 8:
        return conv_bn_act(deconv2x2_relu(x, deconv), conv)
    Ci, Co = deconv.weight.shape[0], deconv.weight.shape[1]
    w2 = deconv.weight.permute(2, 3, 1, 0).reshape(4 * Co, Ci, 1, 1)
    b2 = deconv.bias.repeat(4) if deconv.bias is not None else None
    # (the 4*Cout-channel map -- the largest activation of the head -- is read by the logits conv and as its own
    # ReLU mask only: parts only, no fp32 copy)
    # the logits conv is the only reader of that map: its data gradient prepares the deconv's gradient in its
    # epilogue (ReLU mask from part 0, parts, bias sums) -- no fp32 gradient of the 1.7-GB map, no stand-alone
    # preparation pass (0.84 ms per step); the handed-over parts are [4M, Cout] rows = the [M, 4 Cout] the
    # deconv-as-1x1 wants, as they lie in memory
    ch = {} if CHAIN_DECONV else None
    y = hip._ConvFn.apply(x, w2, b2, None, None, None, True, (1, 1), (1, 1), (0, 0, 0, 0), None, None, ch,
                          deconv.weight, None, PARTS_ONLY)
    N, _, H, W = y.shape
    y4 = y.permute(0, 2, 3, 1).reshape(N, H, W * 4, Co).permute(0, 3, 1, 2)     # pixel (i, 4j + 2a + b)
    hit = getattr(y, "_sln_parts", None)
    if hit is not None and hit[0][0] == y._version:     # the epilogue's parts of y are y4's, re-viewed
        y4._sln_parts = ((y4._version,) + tuple(hit[0][1:]), hit[1].view(hit[1].shape[0], -1, Co), hit[2])
        po = getattr(y, "_sln_po", None)
        if po is not None:
            y4._sln_po = (y4._sln_parts[1], po[1], po[2])
    z = conv_bn_act(y4, conv, chain_in=ch)
    K = z.shape[1]
    z = z.permute(0, 2, 3, 1).reshape(N, H, W, 2, 2, K).permute(0, 5, 1, 3, 2, 4)
    return z.reshape(N, K, 2 * H, 2 * W)


def linear(x, lin):
    """nn.Linear on [R, C] rows through the same GEMM kernel (a 1x1 conv on R 1x1 'images')."""
    hip = _hip_conv() if BACKEND in ("auto", "hip") else None
    if BACKEND != "torch" and x.is_cuda and (hip is None or x.shape[1] % 8):
        raise RuntimeError("linear with %d input features has no HIP path" % x.shape[1])
    if hip is None or not x.is_cuda:
        return F.linear(x, lin.weight, lin.bias)
    R, C = x.shape
    y = hip._ConvFn.apply(x.reshape(R, C, 1, 1), lin.weight.reshape(lin.out_features, C, 1, 1), lin.bias,
                          None, None, None, False, (1, 1), (1, 1), (0, 0, 0, 0), None, None, None, lin.weight)
    return y.reshape(R, lin.out_features)
