"""Attribute the GPU time of one train step to autograd/aten ops and call sites with torch.profiler
(ResNet-101, 1024^2, 16 images, stage=all -- the bench.py workload, same setup as bench.py:main)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity


def main(batch=16, dim=1024):
    from sln_amodal_amd import synthetic
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.model import LAYER_REGEX, MaskRCNN
    dev = torch.device("cuda:0")

    class C(Config):
        NAME = "attrib"
        IMAGE_MAX_DIM = dim
        IMAGE_MIN_DIM = dim
        ARCHITECTURE = "resnet101"
        BATCH_SIZE = batch

    cfg = C()
    torch.manual_seed(0)
    model = MaskRCNN(cfg, "/tmp/sln_attrib_logs").apply_amodal_heads().to(dev)
    model.set_trainable(LAYER_REGEX["all"], exclusive_off=False)
    for p in model.GLM_modual.parameters():
        p.requires_grad = False
    b = synthetic.make_batch(cfg, batch, dim, dim, seed=1234, device=dev, anchors_f64=model.anchors_f64)
    synthetic.calibrate_batchnorm(model, b["images"][:4])
    synthetic.calibrate_glm(model, b["images"][:2])
    synthetic.warm_start_rpn(model, [b], iters=40)
    opt = model.make_optimizer(cfg.LEARNING_RATE)
    for _ in range(2):
        model.train_step(b, opt, None)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
        model.train_step(b, opt, None)
        torch.cuda.synchronize()
    rows = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.self_device_time_total)
    want = ("aten::add", "aten::copy_", "aten::add_", "aten::mul", "aten::fill_", "aten::sum", "aten::cat",
            "aten::clone", "aten::contiguous")
    for e in [r for r in rows if r.key in want and r.self_device_time_total >= 100][:40]:
        print("%9.2f ms  n=%5d  %-14s %s" % (e.self_device_time_total / 1e3, e.count, e.key, str(e.input_shapes)[:150]))


if __name__ == "__main__":
    main()
