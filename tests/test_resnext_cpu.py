"""BASELINE.json configs[4] (ResNeXt-101, 32 groups + multi-scale ASPP heads; dead code in the reference): this
repo's modules against fixtures produced by the reference's own classes (tools/gen_golden_resnext.py), torch path."""
import numpy as np
import torch

from tests._util import golden, key_init_


def damp_(module):
    with torch.no_grad():
        for k, t in module.state_dict().items():
            if (k.endswith("bn3.weight") and ".layer" in "." + k) or k.endswith("downsample.1.weight"):
                t.mul_(0.3)


def build_encoder(g, device):
    from sln_amodal_amd.modal.resnext import GroupBottleneck, ResNeXt, ResNeXtEncoder
    enc = ResNeXtEncoder(ResNeXt(GroupBottleneck, [int(v) for v in g["layers"]])).eval()
    key_init_(enc)
    damp_(enc)
    return enc.to(device)


def build_msc(g, device):
    from sln_amodal_amd.modal.resnext import DeepLabV2_ResNeXt101_MSC
    msc = DeepLabV2_ResNeXt101_MSC(21, layers=[int(v) for v in g["layers"]]).eval()
    key_init_(msc)
    damp_(msc)
    return msc.to(device)


def check(device, tol):
    g = golden("module_resnext")
    enc = build_encoder(g, device)
    # (SynchronizedBatchNorm2d's training-time accumulators `_tmp_running_*` / `_running_iter` have no eval role:
    # sln_amodal_amd.modal.resnext.load_reference_state_dict drops them)
    keys = lambda ks: sorted(str(k) for k in ks if not any(t in str(k) for t in
                                                             ("num_batches_tracked", "_tmp_running", "_running_iter")))
    assert keys(enc.state_dict().keys()) == keys(g["enc_keys"])       # the reference's names: its checkpoints load
    with torch.no_grad():
        outs = enc(torch.from_numpy(g["x"]).to(device), return_feature_maps=True)
    for i, o in enumerate(outs):
        want = g["stage%d" % i]
        assert tuple(o.shape) == want.shape
        err = np.abs(o.cpu().numpy() - want).max() / np.abs(want).max()
        assert err < tol, (i, err)
    msc = build_msc(g, device)
    assert keys(msc.state_dict().keys()) == keys(g["msc_keys"])
    with torch.no_grad():
        lg = msc(torch.from_numpy(g["xm"]).to(device))
    want = g["msc_logits"]
    assert tuple(lg.shape) == want.shape
    assert np.abs(lg.cpu().numpy() - want).max() / np.abs(want).max() < tol


def test_resnext_encoder_and_msc_heads_match_the_reference_modules_cpu():
    check("cpu", 1e-5)


def test_reference_checkpoint_keys_load_into_the_encoder():
    """A state dict with the reference's keys -- incl. SynchronizedBatchNorm2d's training accumulators and the
    classifier, which an encoder does not have -- loads; a checkpoint that lacks a tensor raises."""
    import pytest
    from sln_amodal_amd.modal.resnext import GroupBottleneck, ResNeXt, ResNeXtEncoder, load_reference_state_dict
    g = golden("module_resnext")
    layers = [int(v) for v in g["layers"]]
    full = ResNeXt(GroupBottleneck, layers)
    key_init_(full)
    sd = dict(full.state_dict())
    for k in [k for k in sd if k.endswith("running_mean")]:
        sd[k.replace("running_mean", "_tmp_running_mean")] = sd[k].clone()
        sd[k.replace("running_mean", "_running_iter")] = torch.ones(1)
    enc = ResNeXtEncoder(ResNeXt(GroupBottleneck, layers))
    load_reference_state_dict(enc, sd)
    assert torch.equal(enc.layer3[0].conv2.weight, full.layer3[0].conv2.weight)
    assert torch.equal(enc.bn1.running_var, full.bn1.running_var)
    del sd["layer2.0.conv1.weight"]
    with pytest.raises(KeyError):
        load_reference_state_dict(enc, sd)
