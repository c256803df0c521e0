"""Probe: capture one whole train step (forward, backward, clip + SGD) in a HIP graph and replay it."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main(arch="resnet50", dim=256, batch=2, iters=10):
    from sln_amodal_amd import synthetic
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.model import LAYER_REGEX, MaskRCNN
    dev = torch.device("cuda:0")

    class C(Config):
        NAME = "graph"
        IMAGE_MAX_DIM = dim
        IMAGE_MIN_DIM = dim
        ARCHITECTURE = arch
        BATCH_SIZE = batch

    cfg = C()
    torch.manual_seed(0)
    model = MaskRCNN(cfg, "/tmp/sln_graph_logs").apply_amodal_heads().to(dev)
    model.set_trainable(LAYER_REGEX["all"], exclusive_off=False)
    for p in model.GLM_modual.parameters():
        p.requires_grad = False
    b = synthetic.make_batch(cfg, batch, dim, dim, seed=1234, device=dev, anchors_f64=model.anchors_f64)
    synthetic.calibrate_batchnorm(model, b["images"][:2])
    synthetic.calibrate_glm(model, b["images"][:2])
    synthetic.warm_start_rpn(model, [b], iters=20)
    opt = model.make_optimizer(cfg.LEARNING_RATE)

    def eager(n):
        torch.cuda.synchronize(); t = time.time()
        for _ in range(n):
            loss, _ = model.train_step(b, opt)
        torch.cuda.synchronize()
        return (time.time() - t) / n * 1e3, float(loss)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            model.train_step(b, opt)
    torch.cuda.current_stream().wait_stream(s)
    print("eager ms/step %.2f loss %.4f" % eager(iters), flush=True)
    opt.capturing = True
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss, _ = model.train_step(b, opt)
    print("captured", flush=True)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    print("graph ms/step %.2f loss %.4f" % ((time.time() - t) / iters * 1e3, float(loss)), flush=True)


if __name__ == "__main__":
    main(*(sys.argv[1:2] or ["resnet50"]), dim=int(sys.argv[2]) if len(sys.argv) > 2 else 256,
         batch=int(sys.argv[3]) if len(sys.argv) > 3 else 2)
