"""Convergence / dynamics tests of the train step, collected LAST (tests/conftest.py FILE_ORDER): they assert that
the assembled step LEARNS (reference loop: model.py:383-460), not a parity with the oracle -- a red test here must
not hide the kernel-vs-oracle sweeps from a `-x` run.  Each failure message carries the six losses over the run,
the clip norm, the optimiser's skipped-step counter and the fp16 saturation counter."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.test_model_gpu import _small_model

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip_backend():
    from sln_amodal_amd import nn_ops
    old = nn_ops.BACKEND
    nn_ops.BACKEND = "hip"       # raise instead of silently falling back
    yield
    nn_ops.BACKEND = old


def _prepared(seed=3):
    """Model + one fixed batch + fixed sampling priorities, frozen BN calibrated, RPN warm-started."""
    from sln_amodal_amd import synthetic
    m, cfg = _small_model()
    batch = synthetic.make_batch(cfg, 2, 256, 256, seed=seed, anchors_f64=m.anchors_f64)
    synthetic.calibrate_batchnorm(m, batch["images"])
    synthetic.calibrate_glm(m, batch["images"])
    synthetic.warm_start_rpn(m, [batch], iters=40)
    gen = torch.Generator(device="cuda").manual_seed(5)
    pr = {"pos": torch.rand(2, 1000, device="cuda", generator=gen),
          "neg": torch.rand(2, 1000, device="cuda", generator=gen)}
    return m, cfg, batch, pr


def run_fixed_batch(m, batch, pr, steps, lr, report):
    """`steps` train steps on one batch; returns the report rows (step, total, six losses, clip norm) and the
    optimiser's skipped-step count."""
    opt = m.make_optimizer(lr)
    rows = []
    for it in range(steps):
        loss, parts = m.train_step(batch, opt, priorities=pr)
        if it in report or it == steps - 1:
            row = {"step": it, "total": float(loss), "norm": float(m.last_grad_norm)}
            row.update({k: float(v) for k, v in parts.items()})
            rows.append(row)
    # (skipped for a non-finite gradient norm, vetoed because an operand block clamped: round 6)
    clamped = opt.skipped_clamped_steps() if hasattr(opt, "skipped_clamped_steps") else 0
    run_fixed_batch.clamped_skips = clamped
    return rows, (opt.skipped_steps() if hasattr(opt, "skipped_steps") else 0) - clamped


def _fmt(rows):
    return "\n".join(str({k: round(v, 4) for k, v in r.items()}) for r in rows)


DET = ("rpn_class", "rpn_bbox", "mrcnn_class", "mrcnn_bbox")


def test_train_step_learns_one_fixed_batch_like_the_aten_path():
    """The whole step (HIP conv stack with every backward fusion, losses, clip, SGD) LEARNS one fixed batch, and
    learns it like the same step on aten fp32 convolutions started from the same weights: 80 steps at lr 0.001
    (model.py:383-460).

    Why lr 0.001 and a side-by-side run (round-3 post-mortem): at the reference's lr 0.01 the
    gradient norm of this model is ~480 against the clip of 5.0, so every step is a NORMALISED step of length
    lr * 5 with momentum 0.9 on top; the warm-started RPN is pushed off its optimum first (rpn_bbox 0.07 -> 0.9 by
    step 20, total 3.0 -> 3.6..3.9 -- on aten convolutions too, and the reference's own loop does the same on the
    e2e scene: 4.01 -> 4.98 -> 3.95 -> 6.13) and the run only then descends.  Through that phase the trajectory is
    chaotic: five runs of the SAME build ended at -0.43, -0.78, -0.75, -1.09, -1.05 (gpurun_out r4_bisect*.log; the
    fp32 atomics of the RoIAlign scatter differ in the last bit run to run), the driver's -0.16 was one more draw,
    and none of the eleven A/B switches nor SLN_CONV_PARTS=3 moved the distribution.  At lr 0.001 the run is a
    descent from step 0 on both backends and the two agree: that is what this test pins."""
    from sln_amodal_amd import conv_hip, nn_ops
    sat0 = conv_hip.saturation_count()          # (a counter of the whole process: other tests saturate on purpose)
    m, cfg, batch, pr = _prepared()
    start = {k: v.detach().clone() for k, v in m.state_dict().items()}
    hip, skipped = run_fixed_batch(m, batch, pr, 80, 0.001, (0, 20, 40, 79))
    saturated = conv_hip.saturation_count() - sat0
    m.load_state_dict(start)
    nn_ops.BACKEND = "torch"
    try:
        aten, _ = run_fixed_batch(m, batch, pr, 80, 0.001, (0, 20, 40, 79))
    finally:
        nn_ops.BACKEND = "hip"
    msg = "HIP\n%s\naten\n%s\nskipped=%d saturated=%d" % (_fmt(hip), _fmt(aten), skipped, saturated)
    print(msg)
    assert skipped == 0 and saturated == 0, msg
    assert all(np.isfinite(v) for r in hip for v in r.values()), msg
    det = lambda r: sum(r[k] for k in DET)
    # the same start: step 0 is the same forward (each backend on its OWN proposals: a near-tie in the NMS may pick
    # another roi, so not the 1e-4 of test_loss_parity_hip_conv_vs_aten_conv_same_proposals)
    assert abs(hip[0]["total"] - aten[0]["total"]) <= 0.05, msg       # (measured 3e-3 .. 1.5e-2 over seven runs)
    # it learns: the total and the detector's own four losses fall (observed: total -0.33, both backends)
    assert hip[-1]["total"] < hip[0]["total"] - 0.1, msg         # (recorded: -0.26 ... -0.43 over nineteen runs, sd 0.06)
    assert det(hip[-1]) < det(hip[0]) - 0.07, msg                # (-0.25 ... -0.43)
    # ... and as well as aten does (the runs are not step-wise comparable: aten's own trajectory is not monotone
    # -- 2.81, 2.92, 2.68 at steps 20 / 40 / 79 in the recorded run -- and the two part ways like any two fp32
    # implementations would, tests/test_multistep_gpu.py's control)
    # (recorded: HIP - aten at step 79 between -0.12 and +0.11 over nineteen runs, sd 0.08: both are draws)
    assert hip[-1]["total"] <= aten[-1]["total"] + 0.35, msg
    assert det(hip[-1]) <= det(aten[-1]) + 0.35, msg


def test_train_step_at_the_reference_learning_rate_stays_finite_and_fits_the_masks():
    """lr 0.01 (the reference's), 80 steps: the chaotic regime described above, side by side with the same 80 steps on
    aten fp32 convolutions from the same weights (a regression that halves the learning signal at the
    reference learning rate must not pass).  What holds in EVERY run: nothing is skipped, every loss stays finite, the
    two mask losses -- whose gradient does not pass through the clipped-away RPN phase -- fall (0.688 -> 0.50..0.60
    in 40 recorded runs) and keep at least half of aten's gain.  The TOTAL is a draw from a heavy-tailed distribution
    on BOTH backends (round 5, `tools/dynamics_spread.py`, profiles/HISTORY_r5.md: ten stand-alone runs of one build
    ended at 1.95 ... 2.48 from 3.04, four aten runs at 2.02 ... 2.24; behind other test files, whose random draws move
    the start state, the product path ended at 1.99 ... 2.56 and aten at 1.95 ... 2.69; one in-suite run of the
    product path in about twenty stayed at 3.5 with the class loss at 1.1 while aten, from the same weights, ended at
    2.2).  So the total is judged on the MEDIAN of three draws of the product path from the same start: it falls, and
    ends no worse than aten's single draw by more than the recorded spread of one build."""
    from sln_amodal_amd import conv_hip, nn_ops
    m, cfg, batch, pr = _prepared()
    start = {k: v.detach().clone() for k, v in m.state_dict().items()}
    draws, vetoed = [], []
    for d in range(3):
        sat0 = conv_hip.saturation_count()
        m.load_state_dict(start)
        rows, skipped = run_fixed_batch(m, batch, pr, 80, 0.01, (0, 20, 40, 79))
        draws.append((rows, skipped, conv_hip.saturation_count() - sat0))
        vetoed.append(run_fixed_batch.clamped_skips)
    m.load_state_dict(start)
    nn_ops.BACKEND = "torch"
    try:
        aten, _ = run_fixed_batch(m, batch, pr, 80, 0.01, (0, 20, 40, 79))
    finally:
        nn_ops.BACKEND = "hip"
    msg = "\n".join("HIP draw %d (skipped=%d saturated=%d vetoed steps=%d)\n%s" % (d, sk, sa, vetoed[d], _fmt(rows))
                    for d, (rows, sk, sa) in enumerate(draws)) + "\naten\n%s" % _fmt(aten)
    print(msg)
    gain_a = aten[0]["layer"] - aten[-1]["layer"]
    for (rows, skipped, saturated), veto in zip(draws, vetoed):
        # An operand block that has to clamp -- the tensor outgrew the amax its scale was derived from one step earlier
        # -- is what this regime provokes (in 40 recorded draws of round 5: 0 blocks in 37, 1, 1 and 11 in the others).
        # Round 6: such a step is NOT APPLIED (vetoed on the device, conv_hip.clamp_veto), so the bound is tight again:
        # nothing skipped for a non-finite norm, at most 2 of the 80 steps vetoed, and a clamped block only ever in a
        # vetoed step (saturated > 0 without a veto would be a clamped-and-applied step).
        assert skipped == 0 and veto <= 2 and (saturated == 0) == (veto == 0), msg
        assert all(np.isfinite(v) for r in rows for v in r.values()), msg
        assert rows[-1]["layer"] < rows[0]["layer"] - 0.03, msg
        assert rows[-1]["layer"] <= aten[-1]["layer"] + 0.08, msg
        # at least half of aten's own learning signal on the masks (the gradient path that is not clipped away)
        assert rows[0]["layer"] - rows[-1]["layer"] >= 0.5 * gain_a - 0.02, msg
    rows = sorted((r for r, _, _ in draws), key=lambda r: r[-1]["total"])[1]      # the median draw
    assert rows[-1]["total"] < rows[0]["total"], msg
    # no worse than aten beyond one build's own spread (x 1.2)
    assert rows[-1]["total"] <= aten[-1]["total"] + 0.8, msg
