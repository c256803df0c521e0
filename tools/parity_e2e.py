"""End-to-end loss parity of the conv backends on cuda:0: same weights, same batch,
same sampling priorities -> six losses with conv backend torch (aten fp32) vs HIP
split-bf16 (2 and 3 parts)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd import conv_hip, nn_ops, synthetic
from sln_amodal_amd.config import Config
from sln_amodal_amd.model import MaskRCNN

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2


class C(Config):
    NAME = "par"; IMAGE_MAX_DIM = dim; ARCHITECTURE = arch


torch.manual_seed(0)
cfg = C()
m = MaskRCNN(cfg, "/tmp/sln_logs").apply_amodal_heads().cuda()
m.set_trainable(".*", exclusive_off=False)
batch = synthetic.make_batch(cfg, B, dim, dim, seed=1234, anchors_f64=m.anchors_f64)
nn_ops.BACKEND = "torch"
synthetic.calibrate_batchnorm(m, batch["images"])
synthetic.calibrate_glm(m, batch["images"])
synthetic.warm_start_rpn(m, [batch], iters=40)
g = torch.Generator(device="cuda").manual_seed(5)
pr = {"pos": torch.rand(B, 1000, device="cuda", generator=g), "neg": torch.rand(B, 1000, device="cuda", generator=g)}
res = {}
fixed = None
for name, be, parts in (("torch", "torch", 3), ("hip3", "hip_or_torch", 3), ("hip3glm2", "hip_or_torch", 3), ("hip2", "hip_or_torch", 2)):
    nn_ops.BACKEND = "torch" if be == "torch" else "auto"
    conv_hip.PARTS = parts
    conv_hip.PARTS_NOGRAD = 2 if name == "hip3glm2" else None
    conv_hip._cache.clear()
    m.zero_grad(set_to_none=True)
    out = m.predict([batch["images"], None, batch["gt_class_ids"], batch["gt_boxes"], batch["gt_layer"]],
                    mode="training", priorities=dict(pr, **(fixed or {})))
    if fixed is None:   # later backends reuse the torch backend's proposals (discrete selection
        fixed = {"rpn_rois": out["rpn_rois"], "num_rois": out["num_rois"]}   # amplifies 1e-7 noise)
        free = None
    else:
        o2 = m.predict([batch["images"], None, batch["gt_class_ids"], batch["gt_boxes"], batch["gt_layer"]],
                       mode="training", priorities=pr)
        same = (o2["rpn_rois"] == fixed["rpn_rois"]).all(dim=2).float().mean().item()
        print("   own proposals identical to torch backend's: %.2f%% of slots" % (100 * same))
    loss, parts_d = m.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
    loss.backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None))
    res[name] = (loss.item(), {k: v.item() for k, v in parts_d.items()}, gn.item(),
                 int(out["roi_valid"].sum()),
                 {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    print(name, "loss %.7f" % loss.item(), "gradnorm %.6f" % gn.item(), "valid rois", int(out["roi_valid"].sum()),
          {k: round(v.item(), 6) for k, v in parts_d.items()})
for k in ("hip3", "hip3glm2", "hip2"):
    d = {n: abs(res[k][1][n] - res["torch"][1][n]) for n in res[k][1]}
    ga, gb = res[k][4], res["torch"][4]
    print("   grads only in one backend:", sorted(set(ga) ^ set(gb))[:12])
    common = sorted(set(ga) & set(gb))
    num = sum(((ga[n] - gb[n]).double() ** 2).sum() for n in common) ** 0.5
    den = sum((gb[n].double() ** 2).sum() for n in common) ** 0.5
    gw = (num / den).item()
    print(k, "max |dloss|", max(d.values()), "total", abs(res[k][0] - res["torch"][0]), "rel L2 err of the full gradient", gw)

# ---- continuous-only check: FPN + RPN losses (no GLM argmax, no roi selection) ----
from sln_amodal_amd.modal import loss as L
gr = {}
for name, be, parts in (("torch", "torch", 3), ("hip3", "auto", 3), ("hip2", "auto", 2)):
    nn_ops.BACKEND = be
    conv_hip.PARTS = parts
    conv_hip._cache.clear()
    m.zero_grad(set_to_none=True)
    m._set_modes("training")
    maps, lg, prb, bb = m.rpn_forward(batch["images"])
    loss = L.compute_rpn_class_loss(batch["rpn_match"], lg) + L.compute_rpn_bbox_loss(batch["rpn_bbox"], batch["rpn_match"], bb) \
        + sum(p_.square().mean() for p_ in maps[:4]) * 1e-3
    loss.backward()
    gr[name] = (loss.item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
for k in ("hip3", "hip2"):
    ga, gb = gr[k][1], gr["torch"][1]
    common = sorted(set(ga) & set(gb))
    num = sum(((ga[n] - gb[n]).double() ** 2).sum() for n in common) ** 0.5
    den = sum((gb[n].double() ** 2).sum() for n in common) ** 0.5
    print("FPN+RPN only:", k, "dloss", abs(gr[k][0] - gr["torch"][0]), "rel L2 grad err", (num / den).item(), "n tensors", len(common))

ga, gb = gr["hip3"][1], gr["torch"][1]
rows = sorted(((((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-30)).item(), n, gb[n].norm().item()) for n in ga if n in gb), reverse=True)
for r in rows[:14]:
    print("  %.3e  %-40s |g|=%.3e" % r)
print("  ... median", rows[len(rows) // 2][0])

for r in rows:
    if r[1].startswith("rpn") or ".P" in r[1] or "C5.2" in r[1]:
        print("  %.3e  %-40s |g|=%.3e" % r)
