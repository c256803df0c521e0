"""Module fixtures for BASELINE.json's configs[4] from the reference's own (dead-code) classes, run on the CPU in
the build container:  python tools/gen_golden_resnext.py  ->  tests/golden/module_resnext.npz

* `modal.resnext.ResNeXt(GroupBottleneck, [2, 1, 1, 1])` in eval mode on a 96 x 96 image, its modules called in the
  order of the reference's encoder wrapper (`modal/models_BCE.py:214-230`, `Resnet.forward`; that file itself
  imports torchvision, which this image lacks, so the eight calls are repeated here): the four stage outputs (every
  block type of the 101-layer net: with / without downsample, stride 1 / 2, 4 / 8 / 16 / 32 channels per group);
* `modal.msc_deeplab.MSC` over (that encoder -> `modal.deeplabv2._ASPP(2048, 21, [6, 12, 18, 24])`) on a 128 x 128
  image: the multi-scale maximum of the logits.  The two-line adaptor between the encoder (which returns a list) and
  the ASPP is written here; the reference has none (nothing in it assembles this configuration).
Initialisation: tests/_util.key_init_ (name-keyed), bn3 / downsample-BN gammas x 0.3 so that 8 residual blocks stay
O(1).  Needs /root/reference; never run on the GPU box."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_harness  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
LAYERS = [2, 1, 1, 1]


def damp_(module):
    with torch.no_grad():
        for k, t in module.state_dict().items():
            if k.endswith("bn3.weight") and ".layer" in "." + k or k.endswith("downsample.1.weight"):
                t.mul_(0.3)


def main():
    ref_harness.install()
    from tests._util import key_init_
    import modal.resnext as ref_rx
    import modal.deeplabv2 as ref_dl
    import modal.msc_deeplab as ref_msc
    torch.manual_seed(0)

    class Enc(torch.nn.Module):           # the call order of models_BCE.Resnet.forward over the reference's modules
        def __init__(self, net):
            super(Enc, self).__init__()
            for name, child in net.named_children():
                if name not in ("avgpool", "fc"):
                    setattr(self, name, child)

        def forward(self, t, return_feature_maps=False):
            t = self.relu1(self.bn1(self.conv1(t)))
            t = self.relu2(self.bn2(self.conv2(t)))
            t = self.relu3(self.bn3(self.conv3(t)))
            t = self.maxpool(t)
            out = []
            for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
                t = layer(t)
                out.append(t)
            return out if return_feature_maps else [t]

    enc = Enc(ref_rx.ResNeXt(ref_rx.GroupBottleneck, LAYERS)).eval()
    key_init_(enc)
    damp_(enc)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 3, 96, 96, generator=g)
    with torch.no_grad():
        outs = enc(x, return_feature_maps=True)
    print("stage outputs", [tuple(o.shape) for o in outs], [float(o.abs().mean()) for o in outs])

    class Base(torch.nn.Module):          # glue: encoder (returns a list) -> ASPP
        def __init__(self, enc_, aspp):
            super(Base, self).__init__()
            self.enc, self.aspp = enc_, aspp

        def forward(self, t):
            return self.aspp(self.enc(t)[0])

    enc2 = Enc(ref_rx.ResNeXt(ref_rx.GroupBottleneck, LAYERS))
    msc = ref_msc.MSC(base=Base(enc2, ref_dl._ASPP(2048, 21, [6, 12, 18, 24])), scales=[0.5, 0.75]).eval()
    key_init_(msc)
    damp_(msc)
    xm = torch.randn(1, 3, 128, 128, generator=g)
    with torch.no_grad():
        logits = msc(xm)
    print("msc logits", tuple(logits.shape), float(logits.abs().mean()))
    np.savez_compressed(os.path.join(OUT, "module_resnext.npz"), layers=np.array(LAYERS), x=x.numpy(),
                        **{"stage%d" % i: o.numpy() for i, o in enumerate(outs)},
                        enc_keys=np.array(sorted(enc.state_dict().keys())),
                        msc_keys=np.array(sorted(msc.state_dict().keys())),
                        xm=xm.numpy(), msc_logits=logits.numpy())


if __name__ == "__main__":
    main()
