"""Calibration of the control replica of tests/test_multistep_gpu.py: how far does an aten replica whose weights
were perturbed by eps (uniform, relative) deviate from the unperturbed aten replica -- in its step-0 forward outputs
and in its weights after K steps -- next to the HIP conv stack on unperturbed weights?  Run on the GPU box:
    python3 tools/control_calibration.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SLN_DEBUG_KNOBS", "1")
import torch  # noqa: E402

from tests._parity import e2e_model  # noqa: E402
from tests._util import golden  # noqa: E402
from tests.test_multistep_gpu import GROUPS, _compare, _step_inputs  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main(K=4):
    from sln_amodal_amd import nn_ops
    g = golden("e2e_multistep_0")
    eps_list = [None, 2.0 ** -22, 2.0 ** -19, 2.0 ** -16, 2.0 ** -13]     # None = the HIP replica
    nn_ops.BACKEND = "torch"
    m_ref, _ = e2e_model("cuda")
    o_ref = m_ref.make_optimizer(0.002)
    names = [n for n, p in m_ref.named_parameters() if p.requires_grad]
    reps = []
    for eps in eps_list:
        m, _ = e2e_model("cuda")
        if eps is not None:
            gen = torch.Generator(device="cuda").manual_seed(11)
            with torch.no_grad():
                for p in m.parameters():
                    if p.requires_grad:
                        p.mul_(1.0 + (torch.rand(p.shape, device="cuda", generator=gen) - 0.5) * 2 * eps)
        nn_ops.BACKEND = "hip" if eps is None else "torch"
        reps.append((eps, m, m.make_optimizer(0.002)))
    p_ref = dict(m_ref.named_parameters())
    for k in range(K):
        batch, pr = _step_inputs(g, k % int(g["steps"]))
        inp = [batch["images"], None, batch["gt_class_ids"], batch["gt_boxes"], batch["gt_layer"]]
        nn_ops.BACKEND = "torch"
        if k == 0:
            with torch.no_grad():
                f_ref = m_ref.predict(inp, mode="training", priorities=pr)
        r0 = {n: p_ref[n].detach().clone() for n in names}
        m_ref.train_step(batch, o_ref, priorities=pr)
        for eps, m, o in reps:
            nn_ops.BACKEND = "hip" if eps is None else "torch"
            P = dict(m.named_parameters())
            fwd = ""
            if k == 0:
                with torch.no_grad():
                    f = m.predict(inp, mode="training", priorities=pr)
                fwd = "  forward deviation: rpn logits %.2e mask logits %.2e class logits %.2e" % (
                    rel(f["rpn_class_logits"], f_ref["rpn_class_logits"]), rel(f["mrcnn_mask"], f_ref["mrcnn_mask"]),
                    rel(f["mrcnn_class_logits"], f_ref["mrcnn_class_logits"]))
            w0 = {n: P[n].detach().clone() for n in names}
            m.train_step(batch, o, priorities=pr)
            cos, drift = _compare(P, p_ref, w0, r0, names)
            print("step %d  %-12s min cos %.6f (%s)  drift C1 %.2e C4 %.2e rpn %.2e mask %.2e%s" % (
                k, "HIP" if eps is None else "eps 2^%d" % round(torch.log2(torch.tensor(eps)).item()),
                min(cos.values()), min(cos, key=cos.get), drift["fpn.C1"], drift["fpn.C4"], drift["rpn."],
                drift["mask."], fwd), flush=True)
    nn_ops.BACKEND = "auto"


if __name__ == "__main__":
    main()
