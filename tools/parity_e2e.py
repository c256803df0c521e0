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
for name, be, parts in (("torch", "torch", 3), ("hip3", "hip_or_torch", 3), ("hip2", "hip_or_torch", 2)):
    nn_ops.BACKEND = "torch" if be == "torch" else "auto"
    conv_hip.PARTS = parts
    conv_hip._cache.clear()
    m.zero_grad(set_to_none=True)
    out = m.predict([batch["images"], None, batch["gt_class_ids"], batch["gt_boxes"], batch["gt_layer"]],
                    mode="training", priorities=pr)
    loss, parts_d = m.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
    loss.backward()
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters() if p.grad is not None))
    res[name] = (loss.item(), {k: v.item() for k, v in parts_d.items()}, gn.item(),
                 int(out["roi_valid"].sum()), m.fpn.C4[5].conv2.weight.grad.clone())
    print(name, "loss %.7f" % loss.item(), "gradnorm %.6f" % gn.item(), "valid rois", int(out["roi_valid"].sum()),
          {k: round(v.item(), 6) for k, v in parts_d.items()})
for k in ("hip3", "hip2"):
    d = {n: abs(res[k][1][n] - res["torch"][1][n]) for n in res[k][1]}
    gw = (res[k][4] - res["torch"][4]).abs().max().item() / res["torch"][4].abs().max().item()
    print(k, "max |dloss|", max(d.values()), "total", abs(res[k][0] - res["torch"][0]), "rel grad err (C4.5.conv2.w)", gw)
