"""Diagnostics for the end-to-end parity tests (GPU): prints per-quantity errors instead of asserting."""
import os
import sys
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SLN_DEBUG_KNOBS", "1")
from tests._parity import e2e_model  # noqa: E402
from tests._util import golden, key_init_  # noqa: E402
from tests import test_e2e_gpu as T  # noqa: E402
from sln_amodal_amd import nn_ops  # noqa: E402


def rows(name, got, want):
    got = got.detach().double().cpu().numpy().reshape(want.shape[0], -1)
    want = np.asarray(want, np.float64).reshape(want.shape[0], -1)
    e = np.abs(got - want).max(axis=1) / max(np.abs(want).max(), 1e-12)
    bad = np.nonzero(e > 1e-4)[0]
    print("  %-22s max rel %.3e  rows>1e-4: %s" % (name, e.max(), bad[:12].tolist()))
    return bad


def train(scene, ref_rois):
    g = golden("e2e_train_%d" % scene)
    m, cfg = e2e_model("cuda")
    inp, pr = T._inputs([g])
    if ref_rois:
        k = g["rpn_rois"].shape[1]
        rr = torch.zeros((1, 1000, 4), device="cuda")
        rr[0, :k] = torch.from_numpy(g["rpn_rois"][0]).cuda()
        pr = dict(pr, rpn_rois=rr, num_rois=torch.tensor([k], dtype=torch.int32, device="cuda"))
    with torch.no_grad():
        out = m.predict(inp, mode="training", priorities=pr)
        loss, parts = m.compute_losses(out, T.dev(g["rpn_match"]), T.dev(g["rpn_bbox_target"]))
    print("scene %d (reference proposals fed in: %s)" % (scene, ref_rois))
    print("  rpn logits %.2e bbox %.2e" % (T.rel(out["rpn_class_logits"][0], g["rpn_class_logits"][0]),
                                          T.rel(out["rpn_bbox"][0], g["rpn_bbox"][0])))
    k = int(out["num_rois"][0])
    mine = out["rpn_rois"][0, :k].cpu().numpy()
    ref = g["rpn_rois"][0]
    print("  proposals mine %d ref %d" % (k, ref.shape[0]))
    if k == ref.shape[0]:
        d = np.abs(mine - ref).max(axis=1)
        print("   rows off:", np.nonzero(d > 1e-5)[0].tolist()[:20])
    v = out["roi_valid"][0]
    n = int(v.sum())
    print("  rois mine %d ref %d" % (n, g["rois"].shape[0]))
    if n == g["rois"].shape[0]:
        rows("rois", out["rois"][0, :n], g["rois"])
        print("  class ids equal:", np.array_equal(out["target_class_ids"][0, :n].cpu().numpy(), g["target_class_ids"].reshape(-1)))
        rows("target_deltas", out["target_deltas"][0, :n], g["target_deltas"])
        print("  masks equal:", np.array_equal(out["target_mask"][0, :n].cpu().numpy().astype(np.uint8), g["target_mask"]))
        bad = rows("mrcnn_class_logits", out["mrcnn_class_logits"][0, :n], g["mrcnn_class_logits"])
        rows("mrcnn_bbox", out["mrcnn_bbox"][0, :n], g["mrcnn_bbox"])
        rows("mrcnn_mask", out["mrcnn_mask"][0, :n], g["mrcnn_mask"])
        from sln_amodal_amd.modal.modals import roi_levels
        lv = roi_levels(out["rois"][0, :n], cfg.IMAGE_SHAPE).cpu().numpy()
        import math
        r = g["rois"]
        h, w = r[:, 2] - r[:, 0], r[:, 3] - r[:, 1]
        f = 4 + np.log2(np.sqrt(h * w) / (224.0 / math.sqrt(128 * 128)))
        print("  levels", lv.tolist())
        print("  level frac of bad rows", [(int(b), float(f[b]), r[b].tolist(), out["rois"][0, b].tolist()) for b in bad[:8]])
        print("  row 27: frac %.9f roi %s mine %s" % (f[27], r[27].tolist(), out["rois"][0, 27].tolist()))
        gs = out.get("GLM_feature_sum")
    gl = (out["gloable_lab"][0:1].cpu().numpy() != g["gloable_lab"]).mean()
    print("  gloable_lab pixels differing: %.5f" % gl)
    for name, want in zip([str(x) for x in g["loss_names"]], g["losses"]):
        print("  %-26s got %.6f want %.6f  diff %.2e" % (name, float(parts[T.LOSS_KEYS[name]]), want,
                                                          float(parts[T.LOSS_KEYS[name]]) - want))


def opt_step():
    g = golden("optimizer_step")
    for device in ("cpu", "cuda"):
        m, cfg = e2e_model(device)
        tot = 0.0
        cnt = 0
        per = {}
        for n, p in m.named_parameters():
            if not p.requires_grad:
                continue
            gen = torch.Generator().manual_seed((zlib.crc32(n.encode()) + 7919) & 0x7FFFFFFF)
            gr = (torch.randn(p.shape, generator=gen) * 0.02).to(device)
            per[n] = float(gr.double().pow(2).sum())
            tot += per[n]
            cnt += 1
            p.grad = gr
        params = [p for p in m.parameters() if p.requires_grad and p.grad is not None]
        nrm = torch.nn.utils.clip_grad_norm_(params, 5.0)
        print(device, "params", cnt, "exact norm %.6f" % tot ** 0.5, "clip_grad_norm_ %.6f" % float(nrm),
              "fixture %.6f" % float(g["norm0"]))


def grads(variants=None):
    from sln_amodal_amd import conv_hip
    from sln_amodal_amd.modal import modals as M
    from sln_amodal_amd.modal.modals import FPN, RPN, ResNet
    g = golden("module_grads_fpn_rpn")
    for variant in (variants or ["hip", "torch"]):
        backend = "torch" if variant == "torch" else "hip"
        nn_ops.BACKEND = backend
        conv_hip.LINK_SHORTCUT_GRAD = variant != "nolink"
        conv_hip.CHAIN_GRAD_PREP = variant not in ("nochain", "plain")
        conv_hip.CHAIN_BLOCK_OUTPUT = variant not in ("nochainblock", "plain")
        M.CHAIN_TWO_READERS = variant not in ("notwo", "plain")
        if variant == "plain":
            conv_hip.LINK_SHORTCUT_GRAD = False
        conv_hip.FUSE_OUTPUT_SPLIT = variant != "nofuse"
        os.environ["SLN_CONV_TILE256"] = "0" if variant == "t128" else ("2" if variant == "t256" else "1")
        os.environ["SLN_WGRAD_TILE256"] = "0" if variant == "t128" else ("2" if variant == "t256" else "1")
        print("==== variant", variant)
        resnet = ResNet("resnet50", stage5=True)
        fpn = FPN(*resnet.stages(), out_channels=256).eval().cuda()
        rpn = RPN(3, 1, 256).eval().cuda()
        key_init_(fpn); key_init_(rpn)
        T._freeze_bn(fpn, rpn)
        x = T.dev(g["x"]).requires_grad_(True)
        p = fpn(x)
        outs = [rpn(t) for t in p]
        logits = torch.cat([o[0] for o in outs], 1)
        bbox = torch.cat([o[2] for o in outs], 1)
        loss = (logits * T.dev(g["up_logits"])).sum() + (bbox * T.dev(g["up_bbox"])).sum() + \
            sum((t * T.dev(g["up_p%d" % i])).sum() for i, t in enumerate(p[:4]))
        loss.backward()
        print("backend", backend, "loss", float(loss), float(g["loss"]))
        gx = x.grad.double().cpu().numpy()
        print("  gx max-rel %.3e  rel-L2 %.3e" % (np.abs(gx - g["gx"]).max() / np.abs(g["gx"]).max(),
                                                  np.linalg.norm(gx - g["gx"]) / np.linalg.norm(g["gx"])))
        fp, rp = dict(fpn.named_parameters()), dict(rpn.named_parameters())
        for pre, names, pp in (("fpn", g["fpn_names"], fp), ("rpn", g["rpn_names"], rp)):
            for n in [str(s) for s in names]:
                w = g[pre + "_g/" + n]
                got = pp[n].grad.reshape(-1)[:w.size].double().cpu().numpy()
                print("  %-28s max-rel %.3e rel-L2 %.3e" % (pre + "." + n, np.abs(got - w).max() / max(np.abs(w).max(), 1e-30),
                                                            np.linalg.norm(got - w) / max(np.linalg.norm(w), 1e-30)))
    nn_ops.BACKEND = "hip"


if __name__ == "__main__":
    nn_ops.BACKEND = "hip"
    what = sys.argv[1:] or ["train", "opt", "grads"]
    if "train1" in what:
        train(1, False)
        train(1, False)
    if "train" in what:
        for scene in (0, 1):
            for ref_rois in (False, True):
                train(scene, ref_rois)
    if "opt" in what:
        opt_step()
    if "grads" in what:
        grads()
    if "gradvar" in what:
        grads(["hip", "plain", "nolink", "nochain", "nochainblock", "notwo", "nofuse", "t128", "t256", "torch"])
