"""BASELINE.json configs[4] (ResNeXt-101, 32 groups + multi-scale ASPP heads) on one GPU: forward rate at
B x 513^2 (three scales, eval) and the train-step time at B x 321^2 (logits of every scale + their maximum,
cross-entropy, frozen BN, SGD).  Run plainly or under `rocprofv3 --kernel-trace --stats -- python3
tools/resnext_bench.py` for the per-kernel table (profiles/r3_v8_resnext_*).  Prints one JSON line."""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from sln_amodal_amd import conv_hip
from sln_amodal_amd.modal.resnext import DeepLabV2_ResNeXt101_MSC


def main(batch=8, classes=21):
    torch.manual_seed(0)
    net = DeepLabV2_ResNeXt101_MSC(classes).cuda()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
            m.weight.requires_grad = m.bias.requires_grad = False
            m.running_var.fill_(1.0)
    for k, t in net.state_dict().items():                       # residual branches damped: 33 blocks stay O(1)
        if k.endswith("bn3.weight") or k.endswith("downsample.1.weight"):
            t.mul_(0.3)
    out = {"config": "ResNeXt-101 (3,4,23,3; 32 groups) + ASPP(6,12,18,24), MSC scales 1 / 0.5 / 0.75", "batch": batch}
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(batch, 3, 513, 513, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    net.eval()
    with torch.no_grad():
        for _ in range(2):
            conv_hip.update_scales()
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            conv_hip.update_scales()
            y = net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
    out["forward_513"] = {"ms": round(dt * 1e3, 2), "images_per_sec": round(batch / dt, 1), "logits": list(y.shape)}
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    xs = torch.randn(batch, 3, 321, 321, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=0.01, momentum=0.9)
    target = torch.randint(0, classes, (batch, 11, 11), device="cuda", generator=g)
    losses = []

    def step():
        conv_hip.update_scales()
        outs = net(xs)
        loss = sum(F.cross_entropy(F.interpolate(o, size=(11, 11), mode="bilinear", align_corners=False), target)
                   for o in outs)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        losses.append(step())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    out["train_321"] = {"ms": round(dt * 1e3, 2), "images_per_sec": round(batch / dt, 1),
                        "loss": [round(float(l), 4) for l in losses]}
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 8)
