"""The control under "fp32-class" (VERDICT r5 #2a): the default operand format (2 x scaled fp16, 3 MFMA products per
multiply-add), the strict format (3 x bf16, 6 products) and ATEN's fp32 convolution, each against an fp64 reference
of the same fp32 inputs, for the forward pass, the data gradient and the weight gradient of the train step's real
contraction lengths (K = 576 ... 18 432, the classifier's 12 544-K layer) -- in ONE test, on Gaussian operands and on
operands shaped like the network's (post-ReLU activations with per-channel spread; sparse heavy-tailed gradients).
`tools/precision_control.py` prints the same table stand-alone; `profiles/r6_a_precision_control.txt` is a recorded one.

What the table shows (recorded run, relative L2 error against fp64) and what is therefore asserted:

* The default format is AT LEAST AS ACCURATE AS THE STRICT ONE on every layer and pass (0.72 ... 0.99 x): both carry
  fp32's operand bits to within one (23 and 24 significand bits); what separates them is the NUMBER OF ROUNDINGS INTO
  THE fp32 ACCUMULATOR -- 3 per 32-channel step against 6.
* That accumulation is what bounds both: the error grows like sqrt(K) (2.7e-7 at K = 576, 5.3e-7 at 2 304, 1.05e-6 at
  9 216, 1.3e-6 at 12 544), i.e. sqrt(3 K / 32) roundings of 2^-24 each -- asserted as a law below.  A plain fp32
  dot-product loop (K roundings) would sit at sqrt(K) 2^-24 ~ 2.9e-6 for K = 2 304.
* Against ATEN the picture depends on the pass, not on the format: on the DATA gradient the product path is the more
  accurate of the two on 14 of 16 rows (0.60 ... 0.98 x; the classifier layer 1.7 / 1.9 x), on the WEIGHT gradient
  0.53 ... 1.95 x, on the FORWARD pass 0.6 ... 3.7 x -- aten's forward algorithms keep shorter accumulation chains
  (its errors do not grow with K: 3.1e-7 at K = 2 304 and at 4 608), its backward ones longer chains than ours.  So
  "HIP <= 1.5 x aten on every row" does NOT hold (forward rows of K >= 1 024), and no choice of operand format changes
  that: the strict format is worse on exactly those rows.  What holds, and is asserted (recorded maxima in brackets;
  aten's own numbers move with MIOpen's algorithm choice, hence the margins): <= 1.5 x aten on every data-gradient row
  but the classifier's [0.98], <= 2.5 x on every weight-gradient row [1.95], <= 5 x on every row [3.69], and <= 2.5 x the
  LARGEST error aten itself shows on the same layer over its three passes [1.83; the classifier 3.7, held to 5].
Reference layers: modal/modals.py:264-355, 361-412, 419-453, 457-499; modal/deeplabv2.py:16-45."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip_backend():
    from sln_amodal_amd import nn_ops
    old = nn_ops.BACKEND
    nn_ops.BACKEND = "hip"
    yield
    nn_ops.BACKEND = old


def test_default_format_against_fp64_next_to_the_strict_format_and_aten_fp32():
    from tools import precision_control as pc
    rows = pc.table()
    print("\n" + pc.fmt(rows))
    bad = []
    by_layer = {}
    for r in rows:
        by_layer.setdefault((r["shape"], r["kind"]), []).append(r)
    for r in rows:
        tag = "%s / %s / %s" % (r["shape"], r["kind"], r["what"])
        # (1) never worse than the strict format (measured 0.72 ... 0.99)
        if r["p2"] > 1.05 * r["p3"] + 5e-8:
            bad.append("%s: 2 x fp16 %.2e > 3 x bf16 %.2e" % (tag, r["p2"], r["p3"]))
        # (2) the accumulation law: sqrt(3 K / 32) roundings of 2^-24 (+ the operands' own 2^-23); measured constant
        # 0.5 ... 0.65 for the forward / data-gradient rows, below 0.3 for the weight gradients (split-K tree)
        law = 0.8 * math.sqrt(3.0 * r["K"] / 32.0) * 2.0 ** -24 + 1.5e-7
        if r["what"] != "weight gradient" and r["p2"] > law:
            bad.append("%s: %.2e above the accumulation law %.2e" % (tag, r["p2"], law))
        if r["p2"] > 2e-6:
            bad.append("%s: %.2e above 2e-6" % (tag, r["p2"]))
        # (3) next to aten fp32, per pass
        ratio = r["p2_over_aten"]
        fc = "classifier" in r["shape"]
        if r["what"] == "data gradient" and not fc and ratio > 1.5:
            bad.append("%s: %.2f x aten (data gradient)" % (tag, ratio))
        if r["what"] == "weight gradient" and ratio > 2.5:
            bad.append("%s: %.2f x aten (weight gradient)" % (tag, ratio))
        if ratio > 5.0:
            bad.append("%s: %.2f x aten" % (tag, ratio))
    # (4) within 2.5 x of what aten itself shows on the same layer (its worst pass): the two implementations' error
    # bands overlap layer by layer -- which pass carries the longer accumulation chain differs between them
    for (shape, kind), rs in by_layer.items():
        worst_aten = max(r["aten"] for r in rs)
        worst_hip = max(r["p2"] for r in rs)
        if worst_hip > 2.5 * worst_aten and "classifier" not in shape:
            bad.append("%s / %s: worst pass %.2e > 2.5 x aten's worst pass %.2e" % (shape, kind, worst_hip, worst_aten))
        if worst_hip > 5.0 * worst_aten:
            bad.append("%s / %s: worst pass %.2e > 5 x aten's worst pass %.2e" % (shape, kind, worst_hip, worst_aten))
    fwd = [r["p2_over_aten"] for r in rows if r["what"] == "forward"]
    dg = [r["p2_over_aten"] for r in rows if r["what"] == "data gradient"]
    wg = [r["p2_over_aten"] for r in rows if r["what"] == "weight gradient"]
    print("2 x fp16 / aten fp32, min ... max: forward %.2f ... %.2f, data gradient %.2f ... %.2f, weight gradient "
          "%.2f ... %.2f; 2 x fp16 / 3 x bf16 %.2f ... %.2f" % (
              min(fwd), max(fwd), min(dg), max(dg), min(wg), max(wg),
              min(r["p2"] / r["p3"] for r in rows), max(r["p2"] / r["p3"] for r in rows)))
    # the MEDIAN element's relative error (printed by pc.fmt above): where the per-tensor scale's exponent floor shows --
    # heavy-tailed gradient tensors (recorded: data gradients up to 9.3 x aten's, weight gradients up to 3.6 x; Gaussian
    # rows 0.6 ... 2.1 x like the L2 rows).  Held to the recorded envelope; the probe test below states the law.
    for r in rows:
        m_ratio = r["p2_med"] / max(r["aten_med"], 1e-300)
        if m_ratio > (3.0 if r["kind"] == "gauss" else 15.0):
            bad.append("%s / %s / %s: median element error %.2e = %.1f x aten's" % (r["shape"], r["kind"], r["what"],
                                                                                     r["p2_med"], m_ratio))
    assert not bad, "\n".join(bad)


def test_what_an_operand_keeps_by_magnitude_through_the_matrix_instruction():
    """The per-tensor scale's EXPONENT FLOOR, measured (tools/operand_probe.py: a 1x1 convolution with the identity as
    its weight returns (h0 + h1) / s of every element as v_mfma_f32_16x16x32_f16 reads the parts).  fp16 subnormals are
    NOT flushed by the matrix instruction (an element 2^-20 below the maximum still keeps 15 bits); the law is the one
    the format's arithmetic predicts: an activation / weight element b binades below its tensor's maximum keeps
    relative error <= max(2^-23, 2^(b - 35)) -- fp32's 24 bits down to 2^-12 of the maximum, then one bit less per
    binade; a GRADIENT element (2^3 of extra head room, conv_hip.GRAD_HEADROOM_LOG2) <= max(2^-23, 2^(b - 32)).  The
    strict 3 x bf16 format has no floor (exact at every magnitude: 8 exponent bits).  This is where the default format
    is NOT fp32 per element: on heavy-tailed gradient tensors whose typical element lies 2^-15 ... 2^-20 below the
    maximum, a typical element keeps 10 ... 15 bits (the median-error table of the test above: up to 9 x aten's on
    data gradients); sums over many such elements average it out (the relative L2 rows of the same table), and the
    clamp-free alternative -- no head room -- trades it for vetoed steps.  Config.STRICT_LAYERS is the knob."""
    from tools import operand_probe
    from sln_amodal_amd import conv_hip
    for role, shift in (("x", 35), ("gz", 35 - conv_hip.GRAD_HEADROOM_LOG2)):
        for b, worst, med in operand_probe.probe(2, role):
            bound = max(2.0 ** -23, 2.0 ** (b - shift))
            if bound < 0.25:
                assert worst <= 1.02 * bound, (role, b, worst, bound)
    assert all(worst == 0.0 for _, worst, _ in operand_probe.probe(3, "x"))


def test_strict_layers_run_in_three_bf16_parts_and_the_rest_in_two_fp16():
    """MaskRCNN.set_strict_layers / Config.STRICT_LAYERS (conv_hip.PARTS_FOR): a name pattern moves those convolutions
    -- forward, data and weight gradient -- to the strict 3 x bf16 format; everything else stays in 2 x fp16; one train
    step runs through the mixed graph (chained gradient preparations stop at a format boundary and fall back to the
    stand-alone pass) with finite losses and the same losses as the all-default model to 1e-4."""
    from sln_amodal_amd import conv_hip, synthetic
    from tests.test_model_gpu import _small_model
    m, cfg = _small_model()
    batch = synthetic.make_batch(cfg, 2, 256, 256, seed=1234, anchors_f64=m.anchors_f64)
    synthetic.calibrate_batchnorm(m, batch["images"])
    synthetic.calibrate_glm(m, batch["images"])
    synthetic.warm_start_rpn(m, [batch], iters=10)
    gen = torch.Generator(device="cuda").manual_seed(5)
    pr = {"pos": torch.rand(2, 1000, device="cuda", generator=gen), "neg": torch.rand(2, 1000, device="cuda", generator=gen)}
    inp = [batch["images"], None, batch["gt_class_ids"], batch["gt_boxes"], batch["gt_layer"]]
    with torch.no_grad():
        out = m.predict(inp, mode="training", priorities=pr)
        pr = dict(pr, rpn_rois=out["rpn_rois"], num_rois=out["num_rois"])      # (the same proposals for both formats)
        _, base = m.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
    base = {k: float(v) for k, v in base.items()}
    n = m.set_strict_layers(r"fpn\.C[12]\..*")
    try:
        assert n == sum(1 for k, p in m.named_parameters() if p.dim() == 4 and (k.startswith("fpn.C1.") or k.startswith("fpn.C2.")))
        assert conv_hip.PARTS_FOR is conv_hip.parts_for_tagged
        assert conv_hip.PARTS_FOR(m.fpn.C2[0].conv1.weight) == 3 and conv_hip.PARTS_FOR(m.fpn.C3[0].conv1.weight) is None
        with torch.no_grad():
            out = m.predict(inp, mode="training", priorities=pr)
            _, parts = m.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
        for k, v in parts.items():
            assert abs(float(v) - base[k]) <= 1e-4 * max(1.0, abs(base[k])), (k, float(v), base[k])
        seen = []
        conv_hip.PROFILE = seen
        opt = m.make_optimizer(1e-4)
        loss, parts = m.train_step(batch, opt, priorities=pr)
        torch.cuda.synchronize()
        conv_hip.PROFILE = None
        kernels = {e[3] for e in seen}
        assert any(k.endswith("<3>") for k in kernels) and any(not k.endswith("<3>") for k in kernels), kernels
        assert opt.skipped_steps() == 0 and bool(torch.isfinite(loss))
        assert all(p.grad is not None for k, p in m.named_parameters() if p.requires_grad and k.startswith("fpn.C2."))
    finally:
        conv_hip.PROFILE = None
        m.set_strict_layers("")
        conv_hip.PARTS_FOR = None
