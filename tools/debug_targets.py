import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sln_amodal_amd.config import Config
from sln_amodal_amd.model import MaskRCNN
from sln_amodal_amd import synthetic
from sln_amodal_amd.modal.Functions import bbox_overlaps, proposal_layer
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
class C(Config):
    NAME = "dbg"; IMAGE_MAX_DIM = dim; ARCHITECTURE = "resnet50"
torch.manual_seed(0)
cfg = C()
m = MaskRCNN(cfg, "/tmp/sln_logs").apply_amodal_heads().cuda()
batch = synthetic.make_batch(cfg, 2, dim, dim, seed=1234, anchors_f64=m.anchors_f64)
def report(tag):
    with torch.no_grad():
        maps, lg, pr, bb = m.rpn_forward(batch["images"])
        print(tag, "P2 abs mean", maps[0].abs().mean().item(), "rpn_bbox abs mean", bb.abs().mean().item())
        rois, num = proposal_layer([pr, bb], 1000, 0.7, m.anchors, cfg, return_counts=True)
        scale = torch.tensor([dim, dim, dim, dim], device="cuda").float()
        ov = bbox_overlaps(rois, batch["gt_boxes"] / scale)
        for b in range(2):
            o = ov[b, :num[b]]
            print("  img", b, "num rois", int(num[b]), "iou max", round(o.max().item(), 3), "n>=0.5", int((o.max(dim=1)[0] >= 0.5).sum()))
report("raw")
print("calibrated", synthetic.calibrate_batchnorm(m, batch["images"]), "BN layers")
report("bn-calibrated")
t = time.time()
print("warm start loss", synthetic.warm_start_rpn(m, [batch], iters=iters), "%.1fs" % (time.time() - t))
report("rpn warm-started")
