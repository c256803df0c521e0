"""Which tensors of the fp16 x 2 operand format clamp (amax * scale > 65504) in steady-state train steps, and
how far their running maximum jumps from one step to the next.  Diagnostic: host syncs every step."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main(steps=60, batch=16, dim=1024, arch="resnet101"):
    from sln_amodal_amd import conv_hip, synthetic
    from sln_amodal_amd.config import Config
    from sln_amodal_amd.model import LAYER_REGEX, MaskRCNN
    dev = torch.device("cuda:0")

    class C(Config):
        NAME = "sat"
        IMAGE_MAX_DIM = dim
        IMAGE_MIN_DIM = dim
        ARCHITECTURE = arch
        BATCH_SIZE = batch

    cfg = C()
    torch.manual_seed(0)
    model = MaskRCNN(cfg, "/tmp/sln_sat_logs").apply_amodal_heads().to(dev)
    model.set_trainable(LAYER_REGEX["all"], exclusive_off=False)
    for p in model.GLM_modual.parameters():
        p.requires_grad = False
    bs = [synthetic.make_batch(cfg, batch, dim, dim, seed=1234 + 1000 * i, device=dev, anchors_f64=model.anchors_f64)
          for i in range(2)]
    synthetic.calibrate_batchnorm(model, bs[0]["images"][:4])
    synthetic.calibrate_glm(model, bs[0]["images"][:2])
    synthetic.warm_start_rpn(model, bs, iters=40)
    opt = model.make_optimizer(cfg.LEARNING_RATE)
    hist = []
    real = conv_hip.update_scales

    def spy():
        for b in conv_hip._books.values():
            hist.append((b.amax[:b.n].clone(), b.scale[:b.n].clone()))
        real()
    conv_hip.update_scales = spy
    for i in range(steps):
        model.train_step(bs[i % 2], opt)
    torch.cuda.synchronize()
    book = next(iter(conv_hip._books.values()))
    names = book.names
    worst = {}
    for t, (amax, scale) in enumerate(hist[3:], 3):
        n = min(amax.numel(), hist[t - 1][0].numel())
        q = (amax[:n] * scale[:n]).cpu()
        prev = hist[t - 1][0][:n].cpu()
        cur = amax[:n].cpu()
        for i in torch.nonzero(q > 65504.0).flatten().tolist():
            ratio = float(cur[i] / prev[i]) if prev[i] > 0 else float("inf")
            worst.setdefault(i, []).append((t, float(q[i]), ratio))
    print("slots", book.n, "clamping slots", len(worst))
    roles = {}
    for i, ev in sorted(worst.items(), key=lambda kv: -len(kv[1]))[:40]:
        key, shape = names[i]
        roles[key[0]] = roles.get(key[0], 0) + 1
        print("slot %4d %-22s owner %-22s  steps %s  amax*scale up to %.3g  amax jump x%.1f" % (
            i, str(key), str(shape), [e[0] for e in ev], max(e[1] for e in ev), max(e[2] for e in ev)))
    print("by role:", roles)


if __name__ == "__main__":
    main()
