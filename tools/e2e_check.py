"""Quick end-to-end exercise of the training step on cuda:0 (developer tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd.config import Config
from sln_amodal_amd.model import MaskRCNN
from sln_amodal_amd import synthetic

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3


class C(Config):
    NAME = "e2e"
    IMAGE_MAX_DIM = dim
    ARCHITECTURE = arch


torch.manual_seed(0)
cfg = C()
m = MaskRCNN(cfg, "/tmp/sln_logs").apply_amodal_heads().cuda()
m.set_trainable(".*", exclusive_off=False)
for p in m.GLM_modual.parameters():
    p.requires_grad = False
opt = m.make_optimizer(cfg.LEARNING_RATE)
batch = synthetic.make_batch(cfg, B, dim, dim, seed=1234, anchors_f64=m.anchors_f64)
print("gt boxes", batch["gt_boxes"][0, :3].tolist(), "match+", int((batch["rpn_match"] == 1).sum()))
for i in range(steps):
    torch.cuda.synchronize(); t = time.time()
    loss, parts = m.train_step(batch, opt)
    torch.cuda.synchronize()
    print(i, "loss %.5f" % float(loss), {k: round(float(v), 4) for k, v in parts.items()},
          "%.1f ms" % ((time.time() - t) * 1e3))
print("max mem GB", torch.cuda.max_memory_allocated() / 2**30)
