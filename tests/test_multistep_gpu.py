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


def test_five_optimiser_steps_follow_the_reference_loop():
    from sln_amodal_amd import conv_hip
    g = golden("e2e_multistep_0")
    K = int(g["steps"])
    m, cfg = e2e_model("cuda")
    assert tuple(float(v) for v in cfg.MEAN_PIXEL) == (123.7, 116.8, 103.9)
    params = dict(m.named_parameters())
    names = [str(n) for n in g["names"]]
    for n in names:
        assert np.array_equal(params[n].detach().reshape(-1)[:256].cpu().numpy(), g["before/" + n]), n
    opt = m.make_optimizer(float(g["lr"]))
    prev = {n: g["before/" + n].astype(np.float64) for n in names}
    report = []
    # step 0 is the e2e_train_0 fixture (1e-4); later steps inherit the earlier steps' update differences
    loss_tol = [1e-4, 1e-3, 2e-3, 5e-3, 1e-2]
    for k in range(K):
        batch, pr = _step_inputs(g, k)
        loss, parts = m.train_step(batch, opt, priorities=pr)
        want = g["s%d/losses" % k]
        got = np.array([float(parts[LOSS_KEYS[str(n)]]) for n in g["loss_names"]])
        dl = float(np.abs(got - want).max())
        norm, want_norm = float(m.last_grad_norm), float(g["s%d/total_norm" % k])
        worst, worst_name = 0.0, None
        for n in names:
            after = g["s%d/after/%s" % (k, n)].astype(np.float64)
            d_ref = after - prev[n]
            d_got = params[n].detach().reshape(-1)[:256].double().cpu().numpy() - prev[n]
            err = np.linalg.norm(d_got - d_ref) / max(np.linalg.norm(d_ref), 1e-30)
            if err > worst:
                worst, worst_name = err, n
            prev[n] = after
        report.append("step %d: |dloss| %.2e (total %.5f vs %.5f) norm %.4f vs %.4f worst update err %.2e (%s)" % (
            k, dl, got.sum(), want.sum(), norm, want_norm, worst, worst_name))
        print(report[-1])
        msg = "\n".join(report)
        assert dl <= loss_tol[min(k, len(loss_tol) - 1)], msg
        assert abs(norm - want_norm) <= (2e-3 if k == 0 else 2e-2) * want_norm, msg
        # after the comparison the model CONTINUES from its own weights (not re-seated on the reference's): the
        # deviation of step k is what k + 1 starts from; the update error is measured against the reference's
        # own previous slice, so it holds the accumulated drift, bounded below
        assert worst <= (2e-2 if k == 0 else 0.15), msg
    assert opt.skipped_steps() == 0 and conv_hip.saturation_count() == 0


def _cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


GROUPS = ("fpn.C1", "fpn.C2", "fpn.C3", "fpn.C4", "fpn.C5", "fpn.P", "rpn.", "classifier.", "mask.")


def test_ten_steps_hip_convolutions_against_aten_convolutions():
    """Same weights, same batch, same recorded draws and proposals, lr 0.002: per step the six losses within 1e-3,
    the applied update of every parameter group at cosine >= 0.999; after 10 steps the weights of every group
    within 1e-3 (relative L2) of the aten replica's."""
    from sln_amodal_amd import conv_hip, nn_ops
    g = golden("e2e_multistep_0")
    K = 10
    m_hip, cfg = e2e_model("cuda")
    m_ref, _ = e2e_model("cuda")        # the same name-keyed initialisation: identical weights
    o_hip, o_ref = m_hip.make_optimizer(0.002), None
    nn_ops.BACKEND = "torch"
    o_ref = m_ref.make_optimizer(0.002)
    nn_ops.BACKEND = "hip"
    names = [n for n, p in m_hip.named_parameters() if p.requires_grad]
    p_hip, p_ref = dict(m_hip.named_parameters()), dict(m_ref.named_parameters())
    assert all(torch.equal(p_hip[n], p_ref[n]) for n in names)
    report = []
    for k in range(K):
        batch, pr = _step_inputs(g, k % int(g["steps"]))
        w0 = {n: p_hip[n].detach().clone() for n in names}
        r0 = {n: p_ref[n].detach().clone() for n in names}
        nn_ops.BACKEND = "torch"
        loss_r, parts_r = m_ref.train_step(batch, o_ref, priorities=pr)
        nn_ops.BACKEND = "hip"
        loss_h, parts_h = m_hip.train_step(batch, o_hip, priorities=pr)
        dl = max(abs(float(parts_h[k_]) - float(parts_r[k_])) for k_ in parts_h)
        cos, drift = {}, {}
        for grp in GROUPS:
            ns = [n for n in names if n.startswith(grp)]
            uh = torch.cat([(p_hip[n].detach() - w0[n]).reshape(-1) for n in ns])
            ur = torch.cat([(p_ref[n].detach() - r0[n]).reshape(-1) for n in ns])
            wh = torch.cat([p_hip[n].detach().reshape(-1) for n in ns])
            wr = torch.cat([p_ref[n].detach().reshape(-1) for n in ns])
            cos[grp] = _cos(uh, ur)
            drift[grp] = float((wh - wr).double().norm() / wr.double().norm())
        report.append("step %d: loss %.5f / %.5f  |dparts| %.2e  min cos %.6f (%s)  max drift %.2e (%s)" % (
            k, float(loss_h), float(loss_r), dl, min(cos.values()), min(cos, key=cos.get),
            max(drift.values()), max(drift, key=drift.get)))
        print(report[-1])
        msg = "\n".join(report)
        assert dl <= 1e-3, msg
        assert min(cos.values()) >= 0.999, msg
    assert max(drift.values()) <= 1e-3, msg
    assert o_hip.skipped_steps() == 0 and conv_hip.saturation_count() == 0
