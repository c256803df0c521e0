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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        model.train_step(b, opt, None)
        torch.cuda.synchronize()
    # every op that launched a kernel that is not one of ours, by (op, first frame inside the package): time, launches
    import collections
    agg = collections.defaultdict(lambda: [0.0, 0, set()])
    for ev in prof.events():
        if ev.self_device_time_total <= 0 or not ev.key.startswith(("aten::", "Memset", "Memcpy")):
            continue
        site = "?"
        for fr in (ev.stack or []):
            if "sln_amodal_amd" in fr and "torch/" not in fr:
                site = fr.split("sln_amodal_amd/")[-1]
                break
        k = (ev.key, site)
        agg[k][0] += ev.self_device_time_total
        agg[k][1] += 1
        agg[k][2].add(str(ev.input_shapes)[:80])
    big = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        if ev.self_device_time_total > 0 and ev.key in ("aten::add_", "aten::copy_", "aten::add", "aten::sum", "aten::fill_",
                                                        "aten::mul", "aten::div", "aten::cat", "aten::_softmax"):
            k = (ev.key, str(ev.input_shapes)[:110])
            big[k][0] += ev.self_device_time_total
            big[k][1] += 1
    for (key, shp), (t, n) in sorted(big.items(), key=lambda kv: -kv[1][0])[:45]:
        print("%8.3f ms n=%4d %-14s %s" % (t / 1e3, n, key, shp))
    tot = sum(v[0] for v in agg.values())
    print("aten / runtime ops with device time: %.2f ms in %d calls" % (tot / 1e3, sum(v[1] for v in agg.values())))
    for (key, site), (t, n, shp) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
        print("%8.3f ms n=%4d %-22s %-60s %s" % (t / 1e3, n, key, site[:60], sorted(shp)[0] if shp else ""))


if __name__ == "__main__":
    main()
