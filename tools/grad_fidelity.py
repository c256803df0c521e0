"""Which fp32 backend is closer to the truth?  FPN + RPN losses (continuous: no argmax, no roi
selection) of one fixed batch; gradients w.r.t. every backbone / FPN / RPN parameter from
  (a) aten fp32 convolutions on the GPU,
  (b) the HIP split-bf16 stack (P = 3) on the GPU,
  (c) an fp64 copy of the model on the host CPU (reference).
Prints the relative L2 distance of each pair.  If (a)-(c) and (b)-(c) are of the same size as
(a)-(b), the few-% gradient difference between the fp32 backends is the network's own sensitivity
to fp32 rounding (ReLU masks of a randomly initialised 100-layer net), not an error of either."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd import conv_hip, nn_ops, synthetic
from sln_amodal_amd.config import Config
from sln_amodal_amd.model import MaskRCNN
from sln_amodal_amd.modal import loss as L

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 512
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2


class C(Config):
    NAME = "fid"; IMAGE_MAX_DIM = dim; ARCHITECTURE = arch


torch.manual_seed(0)
cfg = C()
m = MaskRCNN(cfg, "/tmp/sln_logs").apply_amodal_heads(glm=False).cuda()
m.set_trainable(".*", exclusive_off=False)
batch = synthetic.make_batch(cfg, B, dim, dim, seed=1234, anchors_f64=m.anchors_f64)
nn_ops.BACKEND = "torch"
synthetic.calibrate_batchnorm(m, batch["images"])
synthetic.warm_start_rpn(m, [batch], iters=40)


def rpn_loss(model, images, match, tbox):
    model._set_modes("training")
    maps, lg, prb, bb = model.rpn_forward(images)
    return L.compute_rpn_class_loss(match, lg) + L.compute_rpn_bbox_loss(tbox, match, bb) + \
        sum(p_.square().mean() for p_ in maps[:4]) * 1e-3


def grads(model):
    return {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}


res = {}
for name, be in (("aten_fp32", "torch"), ("hip_p3", "auto")):
    nn_ops.BACKEND = be
    conv_hip.PARTS = 3
    m.zero_grad(set_to_none=True)
    loss = rpn_loss(m, batch["images"], batch["rpn_match"], batch["rpn_bbox"])
    loss.backward()
    res[name] = (float(loss), grads(m))
    print(name, "loss %.9f" % float(loss), flush=True)

nn_ops.BACKEND = "torch"
m64 = copy.deepcopy(m).cpu().double()
m64.zero_grad(set_to_none=True)
torch.set_num_threads(min(64, os.cpu_count() or 8))
loss = rpn_loss(m64, batch["images"].cpu().double(), batch["rpn_match"].cpu(), batch["rpn_bbox"].cpu().double())
loss.backward()
res["cpu_fp64"] = (float(loss), grads(m64))
print("cpu_fp64 loss %.9f" % float(loss), flush=True)


def rel(a, b):
    common = sorted(set(a) & set(b))
    num = sum(((a[n] - b[n]) ** 2).sum() for n in common) ** 0.5
    den = sum((b[n] ** 2).sum() for n in common) ** 0.5
    return float(num / den), len(common)


for x, y in (("aten_fp32", "cpu_fp64"), ("hip_p3", "cpu_fp64"), ("hip_p3", "aten_fp32")):
    r, n = rel(res[x][1], res[y][1])
    print("%-10s vs %-10s  |dloss| %.3e   rel L2 of the gradient %.5f  (%d tensors)" %
          (x, y, abs(res[x][0] - res[y][0]), r, n))
