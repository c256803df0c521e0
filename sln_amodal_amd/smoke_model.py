"""One tiny training step of the assembled model on cuda:0 (used by
__graft_entry__.smoke)."""
import torch


def run():
    from . import synthetic
    from .config import Config
    from .model import MaskRCNN

    class C(Config):
        NAME = "smoke"
        IMAGE_MAX_DIM = 128
        ARCHITECTURE = "resnet50"

    torch.manual_seed(0)
    cfg = C()
    m = MaskRCNN(cfg, "/tmp/sln_smoke").apply_amodal_heads().cuda()
    m.set_trainable(".*", exclusive_off=False)
    for p in m.GLM_modual.parameters():
        p.requires_grad = False
    batch = synthetic.make_batch(cfg, 1, 128, 128, seed=1, anchors_f64=m.anchors_f64)
    synthetic.calibrate_batchnorm(m, batch["images"])
    synthetic.calibrate_glm(m, batch["images"])
    opt = m.make_optimizer(cfg.LEARNING_RATE)
    loss, _ = m.train_step(batch, opt)
    assert torch.isfinite(loss), "train step produced a non-finite loss"
    return float(loss)
