"""Which layers of one train step still run a stand-alone gradient preparation (conv_hip._grad_prep), with the
gradient's shape and the time of the launch: the candidates for further chaining."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd import conv_hip, synthetic
from sln_amodal_amd.config import Config
from sln_amodal_amd.model import LAYER_REGEX, MaskRCNN


class C(Config):
    NAME = "gp"; IMAGE_MAX_DIM = 1024; IMAGE_MIN_DIM = 1024; ARCHITECTURE = "resnet101"; BATCH_SIZE = 16


cfg = C()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MaskRCNN(cfg, "/tmp/sln_gp").apply_amodal_heads().to(dev)
m.set_trainable(LAYER_REGEX["all"], exclusive_off=False)
for p in m.GLM_modual.parameters():
    p.requires_grad = False
b = synthetic.make_batch(cfg, 16, 1024, 1024, seed=1234, device=dev, anchors_f64=m.anchors_f64)
synthetic.calibrate_batchnorm(m, b["images"][:4]); synthetic.calibrate_glm(m, b["images"][:2])
synthetic.warm_start_rpn(m, [b], iters=20)
opt = m.make_optimizer(cfg.LEARNING_RATE)
for _ in range(2):
    m.train_step(b, opt, None)
names = {id(p): n for n, p in m.named_parameters()}
log = []
orig = conv_hip._grad_prep


def spy(gy, y, scale, want_gu, want_bias, parts, slot=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(gy, y, scale, want_gu, want_bias, parts, slot)
    e1.record()
    log.append((tuple(gy.shape), y is not None, want_gu, e0, e1))
    return r


conv_hip._grad_prep = spy
m.train_step(b, opt, None)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for shp, has_y, gu, e0, e1 in log:
    k = (shp, has_y, gu)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print("stand-alone gradient preparations: %d launches, %.2f ms" % (len(log), tot))
for (shp, has_y, gu), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%7.3f ms n=%2d gy %-26s mask=%s writes_gu=%s" % (t, n, shp, has_y, gu))
