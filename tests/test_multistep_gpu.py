"""Multi-step parity of the train step: what single-step fixtures cannot see -- scale slots that go stale, arenas
reused across passes, deferred reduces, inbox leftovers, optimiser state.

1. Against the REFERENCE: tools/gen_golden_e2e.py --steps ran scene 0 through 5 consecutive optimiser steps of the
   reference's own loop (train_model -> train_epoch, model.py:356-444: one SGD object, momentum carried over, a fresh
   flip / box jitter / randperm draw per step, all recorded).  The product path replays the same inputs, draws and
   proposals and must follow the reference's losses, clip norms and parameter updates step by step.
2. Against ATEN: the same K steps on the HIP conv stack and on aten fp32 convolutions (nn_ops.BACKEND = 'torch'), two
   copies of one model in one process.
Tolerances widen with the step index: at lr 0.01 the reference's own trajectory on this scene is not a descent
(total 4.01 -> 4.98 -> 3.95 -> 6.13 -> 5.36), so the ~1e-6 forward difference between two fp32 convolution
implementations is amplified step over step through ReLU switches."""
import numpy as np
import pytest
import torch

from tests._parity import e2e_model
from tests._util import golden
from tests.test_e2e_gpu import LOSS_KEYS, dev

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip_backend():
    from sln_amodal_amd import nn_ops
    old = nn_ops.BACKEND
    nn_ops.BACKEND = "hip"
    yield
    nn_ops.BACKEND = old


def _step_inputs(g, k):
    """Batch + priorities of recorded step k (the reference's augmentation: flip of image and label)."""
    flipped = bool(g["s%d/flipped" % k])
    img = g["image_u8"][:, ::-1] if flipped else g["image_u8"]
    lab = np.ascontiguousarray(g["label"][:, ::-1]) if flipped else g["label"]
    mean = np.array([123.7, 116.8, 103.9], np.float32)
    images = torch.from_numpy((img.astype(np.float32) - mean).transpose(2, 0, 1)[None].copy()).cuda()
    batch = {"images": images, "gt_class_ids": dev(g["s%d/gt_class_ids" % k].astype(np.int32)),
             "gt_boxes": dev(g["s%d/gt_boxes" % k]), "gt_layer": dev(lab.view(np.int64))[None],
             "rpn_match": dev(g["s%d/rpn_match" % k]), "rpn_bbox": dev(g["s%d/rpn_bbox_target" % k])}
    rois = g["s%d/rpn_rois" % k]
    n = rois.shape[1]
    rr = torch.zeros((1, 1000, 4), device="cuda")
    rr[0, :n] = torch.from_numpy(rois[0]).cuda()
    pr = {"replay": ([g["s%d/perm_pos" % k]], [g["s%d/perm_neg" % k]]), "rpn_rois": rr,
          "num_rois": torch.tensor([n], dtype=torch.int32, device="cuda")}
    return batch, pr


DEEP = ("fpn.C1", "fpn.C2", "fpn.C3", "fpn.C4")


def _replay_reference_loop(g, backend):
    """K optimiser steps of the recorded reference loop on one conv backend -> per step (|dloss| max, relative clip
    norm error, worst cumulative-update error of the backbone slices (value, name), of the other slices), plus the
    optimiser (skip counter)."""
    from sln_amodal_amd import nn_ops
    K = int(g["steps"])
    nn_ops.BACKEND = backend
    try:
        m, cfg = e2e_model("cuda")
        assert tuple(float(v) for v in cfg.MEAN_PIXEL) == (123.7, 116.8, 103.9)
        params = dict(m.named_parameters())
        names = [str(n) for n in g["names"]]
        for n in names:
            assert np.array_equal(params[n].detach().reshape(-1)[:256].cpu().numpy(), g["before/" + n]), n
        opt = m.make_optimizer(float(g["lr"]))
        rows = []
        for k in range(K):
            batch, pr = _step_inputs(g, k)
            loss, parts = m.train_step(batch, opt, priorities=pr)
            want = g["s%d/losses" % k]
            got = np.array([float(parts[LOSS_KEYS[str(n)]].detach()) for n in g["loss_names"]])
            norm, want_norm = float(m.last_grad_norm), float(g["s%d/total_norm" % k])
            errs = {}
            for n in names:
                # the model CONTINUES from its own weights (never re-seated on the reference's): the error of the
                # cumulative change since step 0's start, relative to the reference's cumulative change
                after = g["s%d/after/%s" % (k, n)].astype(np.float64)
                base = g["before/" + n].astype(np.float64)
                d_got = params[n].detach().reshape(-1)[:256].double().cpu().numpy() - base
                errs[n] = np.linalg.norm(d_got - (after - base)) / max(np.linalg.norm(after - base), 1e-30)
            rows.append({"dl": float(np.abs(got - want).max()), "total": float(got.sum()), "want_total": float(want.sum()),
                         "norm": norm, "want_norm": want_norm, "dnorm": abs(norm - want_norm) / want_norm,
                         "deep": max((e, n) for n, e in errs.items() if n.startswith(DEEP)),
                         "rest": max((e, n) for n, e in errs.items() if not n.startswith(DEEP))})
        return rows, opt
    finally:
        nn_ops.BACKEND = "hip"


# Absolute ceilings per step (measured, scene 0 / scene 1, gpurun_out r5_a_gpu_suite.log and profiles/r4_*_gpu_suite.log:
# losses 6e-8, 3e-4, 2e-4 / 1.2e-3, 1.8e-2, 2.0e-2; backbone slices 1.3e-3, 5.7e-2, 0.10, 0.26, 0.33 -- one 256-element
# slice of C2 under ~90 ReLU layers; heads / FPN / RPN slices 1e-6, 1.8e-2 / 2.3e-2, 2.8e-2, 4.4e-2, 6.3e-2).  Steps 1 and 2
# are held to ~2.5 x their measured values instead of the one bound every later step used to share; the
# reference's own step 3 is an excursion (mrcnn_class 1.05 -> 3.48, total 6.13 on scene 0; 4.79, total 7.66 on scene 1)
# behind which the run-to-run spread of ONE build is as large as its distance from the reference, so steps 3 and 4
# share a bound.
LOSS_TOL = [1e-4, 1e-3, 2.5e-3, 5e-2, 5e-2]
NORM_TOL = [2e-3, 1e-2, 1e-2, 1e-2, 1e-2]
DEEP_TOL = [2e-2, 0.15, 0.25, 0.6, 0.6]
REST_TOL = [2e-3, 0.05, 0.06, 0.1, 0.2]       # (step 4, scene 1: 0.042 ... 0.094 over twelve runs of one build; aten 0.049 ... 0.091)
# ... and RELATIVE to the control: aten fp32 convolutions (MIOpen) replaying the same fixture in the same test.  Both
# are fp32 implementations other than the reference's CPU one, and the system amplifies any difference through ReLU
# switches step over step, so the two error TRAJECTORIES are compared, not single steps (first recorded run: the
# per-step ratio hip / aten of the backbone slices ran 1.0, 12, 1.4, 2.7, 3.2 on scene 0 -- aten's own error jumps
# x 15 one step later -- and 1.0, 1.1, 0.7, 0.9, 0.6 on scene 1): the worst step of the product path may be at most
# CONTROL_FACTOR x the worst step of aten, per quantity.
# Round 6: what the scene-0 ratio IS.  Twenty recorded runs of the backbone quantity: hip 0.28 ... 0.33
# against aten 0.080 ... 0.089, a ratio of 3.1 ... 4.1, the same to three digits with every steady-state fusion switched
# off, with the scales frozen, with the gradient roles' head room at 2^0 (profiles/r6_b_precision_fusions_0.txt).  It
# is ONE 256-element slice -- output channel 0 of fpn.C2.2.conv2.weight, one channel's ReLU pattern under ~90 layers --
# on ONE scene (scene 1: 0.4 ... 1.3 on every quantity, and there the STRICT format is the farthest on the losses:
# profiles/r6_k_precision_subsets_1.txt).  The strict 3 x bf16 format on C1-C2 leaves 1.75, on C1-C4
# (most of the backbone: the strict format's rate, half the headline) 1.0 -- with per-step errors equal to aten's to
# three digits (profiles/r6_a_precision_subsets_0.txt): aten and the strict format round their operands like the
# reference does (24 bits), the default format one bit shorter, and that channel's pattern on that scene sits within
# the difference.  Per layer the default format is the MORE accurate of the two against fp64
# (tests/test_precision_gpu.py); no cheap subset of strict layers removes the ratio, so the default stays and the
# factor goes back from round 5's 6 to 5 (worst recorded ratio 4.1).
CONTROL_FACTOR = 5.0


@pytest.mark.parametrize("scene", [0, 1])
def test_five_optimiser_steps_follow_the_reference_loop(scene):
    """The product path replays the reference's own five optimiser steps (recorded inputs, draws, proposals) and is
    held (i) to per-step absolute ceilings and (ii) to the distance ATEN's convolutions keep from the same fixture
    in the same run: the six losses, the clip norm, and the cumulative update of watched
    256-element parameter slices.  Deep backbone slices (fpn.C1..C4) accumulate ReLU-switch differences exactly like
    two fp32 convolution implementations do among themselves; heads, FPN and RPN slices stay tight."""
    from sln_amodal_amd import conv_hip
    sat0 = conv_hip.saturation_count()          # (a counter of the whole process)
    g = golden("e2e_multistep_%d" % scene)
    hip, opt = _replay_reference_loop(g, "hip")
    saturated = conv_hip.saturation_count() - sat0
    ctl, _ = _replay_reference_loop(g, "torch")
    report, bad = [], []
    for k, (h, c) in enumerate(zip(hip, ctl)):
        report.append("step %d: |dloss| hip %.2e aten %.2e (total %.5f / ref %.5f)  norm err hip %.2e aten %.2e  "
                      "cumulative update err: backbone hip %.2e (%s) aten %.2e, other hip %.2e (%s) aten %.2e" % (
                          k, h["dl"], c["dl"], h["total"], h["want_total"], h["dnorm"], c["dnorm"],
                          h["deep"][0], h["deep"][1], c["deep"][0], h["rest"][0], h["rest"][1], c["rest"][0]))
        print(report[-1])
        kk = min(k, len(LOSS_TOL) - 1)
        if h["dl"] > LOSS_TOL[kk]:
            bad.append("step %d loss" % k)
        if h["dnorm"] > NORM_TOL[kk]:
            bad.append("step %d norm" % k)
        if h["deep"][0] > DEEP_TOL[kk]:
            bad.append("step %d backbone update" % k)
        if h["rest"][0] > REST_TOL[kk]:
            bad.append("step %d heads update" % k)
    # the control: worst step against worst step (floors: aten at rounding level)
    worst = lambda rows, key: max((r[key][0] if isinstance(r[key], tuple) else r[key]) for r in rows)
    # floors: the largest worst-step error ATEN itself showed over the recorded runs of the two scenes -- its own
    # distance from the reference moves by up to x 10 between runs behind the step-3 excursion (scene 1, worst loss
    # difference: 1.2e-2 in one run, 1.4e-3 in the next; gpurun_out r5_a / r5_b)
    for key, floor in (("dl", 2.3e-2), ("dnorm", 8.3e-3), ("deep", 0.14), ("rest", 9.2e-2)):
        hw, cw = worst(hip, key), worst(ctl, key)
        report.append("worst step, %s: hip %.2e aten %.2e (ratio %.2f)" % (key, hw, cw, hw / max(cw, 1e-30)))
        print(report[-1])
        if hw > max(CONTROL_FACTOR * cw, floor):
            bad.append("control %s" % key)
    assert not bad, "%s\n%s" % (bad, "\n".join(report))
    assert opt.skipped_steps() == 0 and saturated == 0


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


GROUPS = ("fpn.C1", "fpn.C2", "fpn.C3", "fpn.C4", "fpn.C5", "fpn.P", "rpn.", "classifier.", "mask.")


def _compare(pa, pb, wa0, wb0, names):
    """Per parameter group: cosine of the two replicas' applied updates, relative L2 distance of their weights."""
    cos, drift = {}, {}
    for grp in GROUPS:
        ns = [n for n in names if n.startswith(grp)]
        ua = torch.cat([(pa[n].detach() - wa0[n]).reshape(-1) for n in ns])
        ub = torch.cat([(pb[n].detach() - wb0[n]).reshape(-1) for n in ns])
        wa = torch.cat([pa[n].detach().reshape(-1) for n in ns])
        wb = torch.cat([pb[n].detach().reshape(-1) for n in ns])
        cos[grp] = _cos(ua, ub)
        drift[grp] = float((wa - wb).double().norm() / wb.double().norm())
    return cos, drift


CONTROL_EPS = 2.0 ** -19


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def test_ten_steps_hip_convolutions_against_aten_convolutions():
    """Same weights, same batches, same recorded draws and proposals, lr 0.002, 10 optimiser steps on three replicas
    in one process: the HIP conv stack, aten fp32 convolutions, and a CONTROL -- aten again, started from weights
    perturbed by a uniform relative 2^-19.  The system amplifies ANY forward difference through ReLU switches, most
    of all into the stem (fpn.C1, under every ReLU of the net): what a replica may be held to is the behaviour of
    another replica whose forward pass deviates from aten's at least as much.  Calibration
    (tools/control_calibration.py, profiles/r4_control_calibration.log): step-0 forward deviation from aten of the
    RPN / mask / class logits -- HIP 1.3e-6 / 6.9e-7 / 1.6e-6; control 2^-22: 7.7e-7 / 5.4e-7 / 9.6e-7; control 2^-19:
    2.7e-6 / 2.0e-6 / 3.0e-6; stem drift after steps 1 / 2 / 3 -- HIP 2.3e-4 / 8.1e-4 / 2.3e-3, control 2^-19
    1.2e-4 / 6.9e-4 / 2.2e-3 (2^-22: 1.8e-7 / 8.0e-6 / 2.1e-4); every other group of the HIP replica stays BELOW
    1e-6 where the controls sit at their own perturbation.  (The first time this test ran, aten's two replicas had
    been given different MIOpen algorithms by a cold find cache: that accidental control tracked the HIP replica
    to three digits -- drift ratios 0.99 .. 1.22, profiles/r4_a_gpu_suite.log.)
    Held: (a) the HIP forward pass is closer to aten's than the control's; (b) per step and group, the update's
    decorrelation 1 - cosine at most 1.5 x the control's + 0.003 and weight drift at most 3 x the control's (+ 1e-6);
    (c) the six
    losses within 1e-3 of aten's over the first four steps, then within max(4e-2, 5 x the control's largest loss
    difference so far) -- measured up to 2.0e-2 at step 9 (the control: up to 5.9e-3)."""
    from sln_amodal_amd import conv_hip, nn_ops
    sat0 = conv_hip.saturation_count()
    g = golden("e2e_multistep_0")
    K = 10
    m_hip, cfg = e2e_model("cuda")
    m_ref, _ = e2e_model("cuda")        # the same name-keyed initialisation: identical weights
    m_ctl, _ = e2e_model("cuda")
    gen = torch.Generator(device="cuda").manual_seed(11)
    with torch.no_grad():
        for p in m_ctl.parameters():
            if p.requires_grad:
                p.mul_(1.0 + (torch.rand(p.shape, device="cuda", generator=gen) - 0.5) * 2 * CONTROL_EPS)
    o_hip = m_hip.make_optimizer(0.002)
    nn_ops.BACKEND = "torch"
    o_ref, o_ctl = m_ref.make_optimizer(0.002), m_ctl.make_optimizer(0.002)
    nn_ops.BACKEND = "hip"
    names = [n for n, p in m_hip.named_parameters() if p.requires_grad]
    p_hip, p_ref, p_ctl = (dict(m.named_parameters()) for m in (m_hip, m_ref, m_ctl))
    assert all(torch.equal(p_hip[n], p_ref[n]) for n in names)
    report, bad = [], []
    # (a) forward deviation at step 0
    batch, pr = _step_inputs(g, 0)
    inp = [batch["images"], None, batch["gt_class_ids"], batch["gt_boxes"], batch["gt_layer"]]
    outs = {}
    with torch.no_grad():
        for key, m, be in (("ref", m_ref, "torch"), ("ctl", m_ctl, "torch"), ("hip", m_hip, "hip")):
            nn_ops.BACKEND = be
            outs[key] = m.predict(inp, mode="training", priorities=pr)
    nn_ops.BACKEND = "hip"
    fwd = {key: max(_rel(outs[key][t], outs["ref"][t]) for t in ("rpn_class_logits", "mrcnn_mask", "mrcnn_class_logits"))
           for key in ("hip", "ctl")}
    report.append("forward deviation from aten (max over RPN / mask / class logits): hip %.2e control %.2e" %
                  (fwd["hip"], fwd["ctl"]))
    print(report[-1])
    if not fwd["hip"] <= fwd["ctl"]:
        bad.append("forward deviation")
    assert fwd["hip"] <= 1e-4          # (the north-star tolerance, as in test_e2e_gpu)
    dc_max = 0.0
    for k in range(K):
        batch, pr = _step_inputs(g, k % int(g["steps"]))
        snap = lambda P: {n: P[n].detach().clone() for n in names}
        w0, r0, c0 = snap(p_hip), snap(p_ref), snap(p_ctl)
        nn_ops.BACKEND = "torch"
        loss_r, parts_r = m_ref.train_step(batch, o_ref, priorities=pr)
        loss_c, parts_c = m_ctl.train_step(batch, o_ctl, priorities=pr)
        nn_ops.BACKEND = "hip"
        loss_h, parts_h = m_hip.train_step(batch, o_hip, priorities=pr)
        dl = max(abs(float(parts_h[k_].detach()) - float(parts_r[k_].detach())) for k_ in parts_h)
        dc = max(abs(float(parts_c[k_].detach()) - float(parts_r[k_].detach())) for k_ in parts_c)
        cos, drift = _compare(p_hip, p_ref, w0, r0, names)
        ccos, cdrift = _compare(p_ctl, p_ref, c0, r0, names)
        worst = max(GROUPS, key=lambda grp: drift[grp] / (cdrift[grp] + 1e-12))
        report.append("step %d: loss %.5f / %.5f  |dparts| hip %.2e control %.2e  min cos hip %.6f (%s) control %.6f  "
                      "max drift hip %.2e (%s) control %.2e  worst hip/control drift ratio %.2f (%s)" % (
                          k, float(loss_h), float(loss_r), dl, dc, min(cos.values()), min(cos, key=cos.get),
                          min(ccos.values()), max(drift.values()), max(drift, key=drift.get),
                          max(cdrift.values()), drift[worst] / (cdrift[worst] + 1e-12), worst))
        print(report[-1])
        dc_max = max(dc_max, dc)
        # (steps >= 4: a gross-divergence floor -- the loss difference of EITHER replica moves by x 10 from step to step
        # and run to run there: hip 5e-4 ... 2.0e-2, control 2e-4 ... 5.9e-3 at steps 7-9 over the recorded runs,
        # profiles/r5_v_gpu_suite_red_ten_step_losses.log -- the cosine / drift ratios below carry the precision)
        if dl > (1e-3 if k < 4 else max(4e-2, 5 * dc_max)):
            bad.append("step %d losses" % k)
        # the margin grows with the control's own decorrelation: by step 9 both replicas' updates have turned ~14 degrees
        # away from aten's (cosine 0.97) and two RUNS of this build differ by more than 0.003 there (profiles/
        # r4_flake2_run3.log: one group 0.003 + below the control at step 9 in one run of eight)
        # (multiplicative in the decorrelation 1 - cos, which grows ~ x 1.5 per step for BOTH replicas: at step 9 the
        # control's groups sit at 0.017 ... 0.028 and two RUNS of this build differ by 0.005 there -- r4_flake2_run3,
        # r5_b_gpu_suite.log: fpn.C2 0.0218 against the control's 0.0168 -- so an additive 0.003 + a quarter was red
        # one run in eight)
        low = [grp for grp in GROUPS if (1.0 - cos[grp]) > 1.5 * (1.0 - ccos[grp]) + 0.003]
        if low:
            bad.append("step %d cosine %s" % (k, ["%s %.6f / %.6f" % (grp, cos[grp], ccos[grp]) for grp in low]))
        if any(drift[grp] > 3 * cdrift[grp] + 1e-6 for grp in GROUPS):
            bad.append("step %d drift" % k)
    assert not bad, "%s\n%s" % (bad, "\n".join(report))
    assert o_hip.skipped_steps() == 0 and conv_hip.saturation_count() == sat0
