"""How long does the host need to ENQUEUE one train step (vs the GPU time of the step)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd import synthetic
from sln_amodal_amd.config import Config
from sln_amodal_amd.model import MaskRCNN, LAYER_REGEX

class C(Config):
    NAME = "h"; IMAGE_MAX_DIM = 1024; ARCHITECTURE = "resnet101"; BATCH_SIZE = 16

torch.manual_seed(0)
cfg = C()
m = MaskRCNN(cfg, "/tmp/l").apply_amodal_heads().cuda()
m.set_trainable(".*", exclusive_off=False)
for p in m.GLM_modual.parameters(): p.requires_grad = False
b = synthetic.make_batch(cfg, 16, 1024, 1024, seed=1, anchors_f64=m.anchors_f64)
synthetic.calibrate_batchnorm(m, b["images"][:4]); synthetic.calibrate_glm(m, b["images"][:2])
synthetic.warm_start_rpn(m, [b], iters=20)
opt = m.make_optimizer(1e-3)
for _ in range(2): m.train_step(b, opt)
torch.cuda.synchronize()
t0 = time.perf_counter(); enq = []
for _ in range(4):
    t = time.perf_counter(); m.train_step(b, opt); enq.append(time.perf_counter() - t)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("enqueue per step (s):", [round(e, 3) for e in enq], "wall per step", round(tot / 4, 3))
