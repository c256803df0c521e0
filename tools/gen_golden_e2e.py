"""End-to-end goldens: the reference's OWN MaskRCNN.train_model / predict / detect run in the
build container (CPU, torch 2.10), through tools/ref_harness.py.

    python tools/gen_golden_e2e.py        # writes tests/golden/e2e_*.npz, optimizer_step.npz

What runs is the reference's code, unmodified: model.Dataset.__getitem__ -> load_image_gt ->
load_layer2 / extract_bboxes / build_rpn_targets -> MaskRCNN.train_model -> train_epoch -> predict
(mode='training') -> the six losses -> backward -> clip_grad_norm_(5.0) -> SGD.step, and
MaskRCNN.detect -> predict(mode='inference') -> unmold_detections.  The script only observes:
it wraps predict / the loss functions / torch.randperm / clip_grad_norm_ / SGD.step to RECORD their
arguments and results, forces the DataLoader to num_workers=0 (so the recorded draws are the ones
used) and turns the checkpoint write into a no-op.  Weights are the name-keyed deterministic
initialisation of tests/_util.e2e_init_ (regenerated from the keys on the test side, not stored).

    python tools/gen_golden_e2e.py --masks   # writes tests/golden/e2e_relu_masks_0.npz only
    python tools/gen_golden_e2e.py --steps [--scene 1]  # writes tests/golden/e2e_multistep_<scene>.npz only (run_multistep)
    python tools/gen_golden_e2e.py --dim 256 # writes tests/golden/e2e_train_256_0.npz only: scene 0's train step at 256^2
                                             # (six objects) -- maps of 64^2 .. 4^2: whole 256-row tiles of the conv kernels

--masks re-runs scene 0's train step with three more observers: a forward hook on every nn.ReLU of the
detector (modals.py:276-299, 316, 383, 430, 476) that records the SIGN BITMAP of its output, in call
order; one on C1's max-pool (modals.py:317) that records the winning tap of every window; and
`images.requires_grad_(True)` so that the image gradient exists.  Together with full-length slices of
the C1..C4 weight gradients this lets the GPU test force the reference's discrete choices (which units
are active, which tap wins) and compare the remaining -- smooth -- arithmetic of the backward pass at
1e-4 (tests/test_e2e_gpu.py::test_deep_gradients_with_reference_relu_masks).  The run must reproduce
e2e_train_0's losses exactly (checked).

Third-party stand-ins (harness side, neither is reference code): scipy.misc.imresize -- absent from
scipy >= 1.3; restated from scipy 1.0's published implementation (bytescale to uint8, PIL resize,
back to array) -- used by utils.resize_image (identity here: inputs are already IMAGE_MAX_DIM^2) and
utils.unmold_mask.  The native NMS / crop_and_resize extensions are this repo's C oracle
(tools/ref_harness.py), hence `native = "oracle"`.
"""
import os
import random
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ref_harness  # noqa: E402
from tools.gen_golden import save, synth_label  # noqa: E402

DIM = 128
SLICE = 256          # leading elements of a parameter / gradient kept in the fixture
WATCH = [
    "fpn.C1.0.weight", "fpn.C2.0.conv1.weight", "fpn.C2.2.conv2.weight", "fpn.C3.1.conv2.weight",
    "fpn.C4.5.conv3.weight", "fpn.C4.22.conv1.bias", "fpn.C5.2.conv3.weight", "fpn.P5_conv1.weight",
    "fpn.P2_conv2.1.weight", "fpn.P2_conv2.1.bias", "rpn.conv_shared.weight", "rpn.conv_class.weight",
    "rpn.conv_bbox.bias", "classifier.conv1.weight", "classifier.conv2.bias",
    "classifier.linear_class.weight", "classifier.linear_bbox.bias", "mask.conv1.weight",
    "mask.conv4.weight", "mask.deconv.weight", "mask.conv5.weight", "mask.conv5.bias",
]


def imresize(arr, size, interp="bilinear", mode=None):
    """scipy.misc.imresize as published in scipy 1.0 (misc/pilutil.py): toimage (min-max
    bytescale of non-uint8 data) -> PIL resize -> fromimage."""
    from PIL import Image
    data = np.asarray(arr)
    if data.dtype != np.uint8:
        cmin, cmax = data.min(), data.max()
        cscale = cmax - cmin
        if cscale == 0:
            cscale = 1
        data = ((data - cmin) * (255.0 / cscale)).clip(0, 255)
        data = (data + 0.5).astype(np.uint8)
    im = Image.fromarray(data)
    if isinstance(size, (int, np.integer)):
        size = tuple((np.array(im.size) * (size / 100.0)).astype(int))
    elif isinstance(size, float):
        size = tuple((np.array(im.size) * size).astype(int))
    else:
        size = (int(size[1]), int(size[0]))
    func = {"nearest": 0, "lanczos": 1, "bilinear": 2, "bicubic": 3, "cubic": 3}
    return np.asarray(im.resize(size, resample=func[interp]))


def make_scene(seed, n_obj):
    """uint8 image + uint64 label of painter's-order ellipses, DIM x DIM."""
    rng = np.random.RandomState(seed)
    image = rng.randint(0, 256, (DIM, DIM, 3)).astype(np.uint8)
    label, _ = synth_label(rng, DIM, DIM, n_obj)
    return image, label


class StubDataset(object):
    """The two methods load_image_gt calls (Functions.py:699-701), over in-memory scenes.
    load_layer2 is the reference's AmodalDataset.load_layer2 itself."""

    def __init__(self, ref_train, images, labels, tmp):
        self.image_ids = np.arange(len(images))
        self.images = images
        self.image_info = []
        for i, lab in enumerate(labels):
            path = os.path.join(tmp, "img%d.jpg" % i)
            np.savez(path[:-4] + ".npz", layer=lab)
            self.image_info.append({"path": path, "height": DIM, "width": DIM})
        self._ll2 = ref_train.AmodalDataset.load_layer2

    def load_image(self, image_id):
        return self.images[image_id]

    def load_layer2(self, image_id, config):
        return self._ll2(self, image_id, config)


def build_reference_model(ref_model, ref_config, ref_dl, nn):
    from tests._util import e2e_init_

    class Cfg(ref_config.Config):
        NAME = "golden"
        GPU_COUNT = 0
        IMAGE_MAX_DIM = DIM
        IMAGE_MIN_DIM = DIM
        NUM_CLASSES = 81
        STEPS_PER_EPOCH = 1
        BATCH_SIZE = 1
        EXPERIMENT_DIR = tempfile.mkdtemp()

    cfg = Cfg()
    cfg.STEPS_PER_EPOCH = 1
    model = ref_model.MaskRCNN(cfg, tempfile.mkdtemp())
    # amodal_train.py:606-613
    cfg.NUM_CLASSES = 1 + 1
    model.mask.conv1 = nn.Conv2d(439, 256, kernel_size=3, stride=1)
    model.mask.conv5 = nn.Conv2d(256, cfg.NUM_CLASSES, kernel_size=1, stride=1)
    model.classifier.linear_class = nn.Linear(1024, cfg.NUM_CLASSES)
    model.classifier.linear_bbox = nn.Linear(1024, cfg.NUM_CLASSES * 4)
    model.current_epoch = 0
    model.GLM_modual = ref_dl.DeepLabV2_ResNet101_MSC(182)
    e2e_init_(model)
    model.epoch = 0
    return model, cfg


def grab(params, which):
    return {n: getattr(params[n], which).detach().reshape(-1)[:SLICE].clone().numpy() if which != "data"
            else params[n].detach().reshape(-1)[:SLICE].clone().numpy() for n in WATCH}


MASK_WATCH = WATCH + [
    "fpn.C1.0.bias", "fpn.C2.0.downsample.0.weight", "fpn.C2.1.conv3.weight", "fpn.C3.0.conv1.weight",
    "fpn.C3.3.conv3.weight", "fpn.C4.0.downsample.0.weight", "fpn.C4.11.conv2.weight", "fpn.C4.22.conv3.weight",
    "fpn.C5.0.conv1.weight",
]
MASK_SLICE = 8192


def main(only_masks=False, multistep=False, dim=None):
    global DIM
    big = dim is not None and dim != DIM
    if big:
        DIM = int(dim)
    ref_modals, ref_F = ref_harness.install()
    import scipy.misc
    scipy.misc.imresize = imresize
    import torch.nn as nn
    import config as ref_config
    import model as ref_model
    import utils as ref_utils
    import amodal_train as ref_train
    import modal.deeplabv2 as ref_dl
    from oracle import oracle as orc
    ref_utils.scipy.misc.imresize = imresize

    tmp = tempfile.mkdtemp()
    scenes = [make_scene(101, 4), make_scene(202, 5)] if not big else [make_scene(101, 6)]

    real_loader = torch.utils.data.DataLoader
    real_randperm = torch.randperm
    real_clip = torch.nn.utils.clip_grad_norm_
    real_step = torch.optim.SGD.step
    real_save = torch.save
    real_predict = ref_model.MaskRCNN.predict
    loss_names = ["compute_layer_loss", "compute_mrcnn_class_loss", "compute_rpn_class_loss",
                  "compute_rpn_bbox_loss", "compute_mrcnn_bbox_loss", "compute_amodal_loss"]
    real_losses = {n: getattr(ref_model, n) for n in loss_names}

    # ------------------------------------------------------------------ optimizer step, synthetic grads
    # The optimizer OBJECT is the one train_model builds (model.py:352-358: SGD, momentum, weight decay
    # on every trainable parameter whose name does not contain 'bn'); two clip + step rounds
    # (model.py:441-444) on name-keyed seeded gradients, momentum included in the second.
    if big:
        model = None
    else:
        model, cfg = build_reference_model(ref_model, ref_config, ref_dl, nn)
    if only_masks:
        del model
        run_mask_scene(locals())
        return
    if multistep:
        del model
        run_multistep(locals(), scene=int(sys.argv[sys.argv.index("--scene") + 1]) if "--scene" in sys.argv else 0)
        return
    rec_opt = {}
    if big:
        train_scenes(locals(), "e2e_train_%d_" % DIM)
        return

    def synthetic_epoch(self, datagenerator, optimizer, steps):
        import zlib
        params = dict(self.named_parameters())
        rec_opt["groups"] = [[n for n, p in params.items() if any(p is q for q in g["params"])]
                             for g in optimizer.param_groups]
        rec_opt["wd"] = [g["weight_decay"] for g in optimizer.param_groups]
        rec_opt["lr"], rec_opt["momentum"] = optimizer.param_groups[0]["lr"], optimizer.param_groups[0]["momentum"]
        rec_opt["before"] = grab(params, "data")
        for rnd in range(2):
            for n, p in params.items():
                if not p.requires_grad or n.startswith("GLM_modual"):
                    continue
                g = torch.Generator().manual_seed((zlib.crc32(n.encode()) + 7919 * (rnd + 1)) & 0x7FFFFFFF)
                p.grad = torch.randn(p.shape, generator=g) * (0.02 if rnd == 0 else 0.002)
            rec_opt["norm%d" % rnd] = float(real_clip(self.parameters(), 5.0))     # model.py:441
            optimizer.step()                                                       # model.py:443
            optimizer.zero_grad()
            rec_opt["after%d" % rnd] = grab(params, "data")
        return 0, 0, 0, 0, 0, 0

    ref_model.MaskRCNN.train_epoch, saved_epoch = synthetic_epoch, ref_model.MaskRCNN.train_epoch
    torch.save = lambda *a, **k: None
    torch.utils.data.DataLoader = lambda ds, **k: []
    ref_model.Dataset, saved_ds = (lambda *a, **k: []), ref_model.Dataset
    try:
        model.train_model([], [], 0.01, 1, "all")
    finally:
        ref_model.MaskRCNN.train_epoch = saved_epoch
        ref_model.Dataset = saved_ds
        torch.save = real_save
        torch.utils.data.DataLoader = real_loader
    wd_names, nowd_names = rec_opt["groups"]
    print("optimizer: %d params with weight decay, %d without; norms %.4f %.4f" %
          (len(wd_names), len(nowd_names), rec_opt["norm0"], rec_opt["norm1"]))
    save("optimizer_step", lr=np.array(rec_opt["lr"]), momentum=np.array(rec_opt["momentum"]),
         weight_decay=np.array(rec_opt["wd"]), clip=np.array(5.0),
         norm0=np.array(rec_opt["norm0"]), norm1=np.array(rec_opt["norm1"]),
         grad_scale=np.array([0.02, 0.002]), names=np.array(WATCH),
         n_wd=np.array(len(wd_names)), n_nowd=np.array(len(nowd_names)),
         nowd_names=np.array(nowd_names if nowd_names else [""]),
         **{"before/" + n: v for n, v in rec_opt["before"].items()},
         **{"after0/" + n: v for n, v in rec_opt["after0"].items()},
         **{"after1/" + n: v for n, v in rec_opt["after1"].items()})
    del model

    train_scenes(locals(), "e2e_train_")

    # ------------------------------------------------------------------ inference: detect()
    model, cfg = build_reference_model(ref_model, ref_config, ref_dl, nn)
    cfg.DETECTION_MIN_CONFIDENCE = 0          # InferenceConfig, amodal_train.py
    rec = {}

    def rec_predict_inf(self, input, mode):
        out = real_predict(self, input, mode)
        rec["molded"] = input[0].detach().clone()
        rec["metas"] = np.array(input[1])
        rec["out"] = out
        return out

    real_proposal = ref_model.proposal_layer

    def rec_proposal(*a, **k):
        r = real_proposal(*a, **k)
        rec["rpn_rois"] = r.detach().clone()
        return r

    for si, (image, label) in enumerate(scenes):
        ref_model.MaskRCNN.predict = rec_predict_inf
        ref_model.proposal_layer = rec_proposal
        try:
            res = model.detect([image])
        finally:
            ref_model.MaskRCNN.predict = real_predict
            ref_model.proposal_layer = real_proposal
        detections, mrcnn_mask = rec["out"]
        r = res[0]
        print("detect scene %d: %d detections, %d after unmold, mask pixels %d" %
              (si, detections.shape[1], r["rois"].shape[0], int(r["masks"].sum())))
        save("e2e_detect_%d" % si, native=np.array("oracle"), dim=np.array(DIM), image_u8=image,
             molded=rec["molded"].numpy(), image_metas=rec["metas"],
             rpn_rois=rec["rpn_rois"].numpy(), detections=detections[0].numpy(),
             mrcnn_mask=mrcnn_mask[0].numpy(),
             final_rois=r["rois"], final_class_ids=r["class_ids"], final_scores=r["scores"],
             final_masks=np.packbits(r["masks"].astype(np.uint8), axis=None),
             final_masks_shape=np.array(r["masks"].shape))


def train_scenes(env, prefix):
    """One real training step of the reference's loop per scene -> tests/golden/<prefix><scene>.npz."""
    ref_model, ref_config, ref_dl, nn, ref_F = (env[k] for k in ("ref_model", "ref_config", "ref_dl", "nn", "ref_F"))
    ref_train, scenes, tmp, loss_names, orc = (env[k] for k in ("ref_train", "scenes", "tmp", "loss_names", "orc"))
    real_loader, real_randperm, real_clip, real_step, real_save, real_predict, real_losses = (
        env[k] for k in ("real_loader", "real_randperm", "real_clip", "real_step", "real_save", "real_predict",
                         "real_losses"))
    for si, (image, label) in enumerate(scenes):
        model, cfg = build_reference_model(ref_model, ref_config, ref_dl, nn)
        params = dict(model.named_parameters())
        ds = StubDataset(ref_train, [image], [label], tmp)
        rec = {"perms": [], "losses": {}}

        def rec_randperm(n, *a, **k):
            p = real_randperm(n, *a, **k)
            rec["perms"].append(p.numpy().copy())
            return p

        def rec_predict(self, input, mode):
            out = real_predict(self, input, mode)
            rec["inputs"] = [t.detach().clone() if torch.is_tensor(t) else np.array(t) for t in input]
            rec["outputs"] = out
            return out

        def rec_clip(parameters, max_norm, *a, **k):
            rec["grad"] = grab(params, "grad")
            rec["grad_norms"] = {n: float(params[n].grad.norm()) for n in WATCH}
            total = real_clip(parameters, max_norm, *a, **k)
            rec["total_norm"] = float(total)
            return total

        def rec_step(self, *a, **k):
            rec["before"] = grab(params, "data")
            r = real_step(self, *a, **k)
            rec["after"] = grab(params, "data")
            return r

        def wrap_loss(name):
            def f(*a, **k):
                r = real_losses[name](*a, **k)
                v = r[0] if isinstance(r, tuple) else r
                rec["losses"][name] = float(v)
                if name == "compute_rpn_class_loss":
                    rec["rpn_match"] = a[0].detach().clone()
                if name == "compute_rpn_bbox_loss":
                    rec["rpn_bbox_t"] = a[0].detach().clone()
                return r
            return f

        seed = 1000 + si
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        torch.randperm = rec_randperm
        ref_model.MaskRCNN.predict = rec_predict
        torch.nn.utils.clip_grad_norm_ = rec_clip
        torch.optim.SGD.step = rec_step
        torch.save = lambda *a, **k: None
        torch.utils.data.DataLoader = lambda d, **k: real_loader(d, **dict(k, num_workers=0))
        for n in loss_names:
            setattr(ref_model, n, wrap_loss(n))
        try:
            model.train_model(ds, ds, 0.01, 1, "all")
        finally:
            torch.randperm = real_randperm
            ref_model.MaskRCNN.predict = real_predict
            torch.nn.utils.clip_grad_norm_ = real_clip
            torch.optim.SGD.step = real_step
            torch.save = real_save
            torch.utils.data.DataLoader = real_loader
            for n in loss_names:
                setattr(ref_model, n, real_losses[n])

        images, image_metas, gt_class_ids, gt_boxes, gt_layer = rec["inputs"]
        (rpn_class_logits, rpn_bbox, target_class_ids, mrcnn_class_logits, target_deltas, mrcnn_bbox,
         target_mask, mrcnn_mask, unet_features, image_path, _amodal, gloable_lab, _out) = rec["outputs"]
        rois = unet_features[1]
        glm_feature = unet_features[2]
        # the label the batch corresponds to (load_image_gt flips image and planes together)
        molded0 = image.astype(np.float32) - cfg.MEAN_PIXEL
        got = images[0].permute(1, 2, 0).numpy()
        flipped = not np.allclose(got, molded0)
        if flipped:
            assert np.allclose(got, molded0[:, ::-1])
        lab = np.ascontiguousarray(label[:, ::-1]) if flipped else label
        planes = gt_layer[0].numpy()                       # [L,N,H,W] uint8
        assert np.array_equal(orc.label_decode(lab, 1, planes.shape[1]), planes), "label <-> planes"
        # the DataLoader's RandomSampler draws first (a permutation of the 1-image dataset); the last two
        # are detection_target_layer's (Functions.py:291, 359): positives, then negatives
        assert len(rec["perms"]) == 3 and len(rec["perms"][0]) == 1, [len(p) for p in rec["perms"]]
        rec["perms"] = rec["perms"][-2:]
        total = sum(rec["losses"][n] for n in loss_names)
        print("scene %d: flipped=%s N=%d rois=%d positives=%d losses=%s total=%.6f grad norm=%.4f" %
              (si, flipped, planes.shape[1], rois.shape[0], int((target_class_ids > 0).sum()),
               {k[8:]: round(v, 5) for k, v in rec["losses"].items()}, total, rec["total_norm"]))
        # proposals: re-run the reference's proposal_layer on the recorded RPN outputs (pure function)
        with torch.no_grad():
            probs = torch.softmax(rpn_class_logits, dim=2)
            rpn_rois = ref_F.proposal_layer([probs, rpn_bbox.detach()], proposal_count=cfg.POST_NMS_ROIS_TRAINING,
                                            nms_threshold=cfg.RPN_NMS_THRESHOLD, anchors=model.anchors, config=cfg)
        save(prefix + "%d" % si, native=np.array("oracle"), dim=np.array(DIM), lr=np.array(0.01),
             flipped=np.array(flipped), image_u8=(image[:, ::-1] if flipped else image).copy(),
             images=images.numpy(), label=lab, gt_class_ids=gt_class_ids.numpy(),
             gt_boxes=gt_boxes.numpy(), rpn_match=rec["rpn_match"].numpy(),
             rpn_bbox_target=rec["rpn_bbox_t"].numpy(), perm_pos=rec["perms"][0], perm_neg=rec["perms"][1],
             rpn_class_logits=rpn_class_logits.detach().numpy(), rpn_bbox=rpn_bbox.detach().numpy(),
             rpn_rois=rpn_rois.numpy(), rois=rois.detach().numpy(),
             target_class_ids=target_class_ids.numpy(), target_deltas=target_deltas.numpy(),
             target_mask=target_mask.numpy().astype(np.uint8),
             mrcnn_class_logits=mrcnn_class_logits.detach().numpy(), mrcnn_bbox=mrcnn_bbox.detach().numpy(),
             mrcnn_mask=mrcnn_mask.detach().numpy(), glm_feature_sum=glm_feature.sum(dim=(2, 3)).numpy(),
             gloable_lab=gloable_lab.numpy().astype(np.float32),
             image_path_sum=image_path.sum(dim=(2, 3)).numpy(),
             losses=np.array([rec["losses"][n] for n in loss_names], dtype=np.float64),
             loss_names=np.array(loss_names), total_norm=np.array(rec["total_norm"]),
             names=np.array(WATCH), grad_norms=np.array([rec["grad_norms"][n] for n in WATCH]),
             **{"grad/" + n: v for n, v in rec["grad"].items()},
             **{"before/" + n: v for n, v in rec["before"].items()},
             **{"after/" + n: v for n, v in rec["after"].items()})
        del model


def run_mask_scene(env, scene=0):
    """Scene 0's reference train step once more, with the observers of --masks (module docstring)."""
    import torch.nn.functional as F
    ref_model, ref_config, ref_dl, nn = env["ref_model"], env["ref_config"], env["ref_dl"], env["nn"]
    ref_train, scenes, tmp, loss_names = env["ref_train"], env["scenes"], env["tmp"], env["loss_names"]
    real = {k: env[k] for k in ("real_loader", "real_randperm", "real_clip", "real_step", "real_save",
                                "real_predict", "real_losses")}
    image, label = scenes[scene]
    model, cfg = build_reference_model(ref_model, ref_config, ref_dl, nn)
    params = dict(model.named_parameters())
    ds = StubDataset(ref_train, [image], [label], tmp)
    rec = {"losses": {}, "relu": [], "pool": None}
    hooks = []
    for name, mod in model.named_modules():
        if name.startswith("GLM_modual"):
            continue                     # frozen, detached: its units do not shape any gradient
        if isinstance(mod, nn.ReLU):
            def hook(m, inp, out, name=name):
                k = sum(1 for n, _ in rec["relu"] if n == name)
                rec["relu"].append((name, (out.detach() > 0).numpy()))
                assert k < 8
            hooks.append(mod.register_forward_hook(hook))
        elif isinstance(mod, nn.MaxPool2d) and name == "fpn.C1.4":
            def pool_hook(m, inp, out):
                x = inp[0].detach()
                y, idx = F.max_pool2d(x, m.kernel_size, m.stride, m.padding, return_indices=True)
                assert torch.equal(y, out.detach())
                Wp = x.shape[3]
                ih, iw = idx // Wp, idx % Wp
                oh = torch.arange(y.shape[2]).view(1, 1, -1, 1)
                ow = torch.arange(y.shape[3]).view(1, 1, 1, -1)
                s = m.stride if isinstance(m.stride, int) else m.stride[0]
                k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
                tap = (ih - oh * s) * k + (iw - ow * s)
                assert int(tap.min()) >= 0 and int(tap.max()) < k * k
                rec["pool"] = tap.to(torch.uint8).numpy()
                rec["pool_in_shape"] = tuple(x.shape)
            hooks.append(mod.register_forward_hook(pool_hook))

    def rec_predict(self, input, mode):
        input[0].requires_grad_(True)            # observation only: the image gradient then exists
        rec["images"] = input[0]
        return real["real_predict"](self, input, mode)

    def rec_clip(parameters, max_norm, *a, **k):
        rec["grad"] = {n: params[n].grad.detach().reshape(-1)[:MASK_SLICE].clone().numpy() for n in MASK_WATCH}
        rec["grad_norms"] = {n: float(params[n].grad.norm()) for n in MASK_WATCH}
        rec["grad_images"] = rec["images"].grad.detach().clone().numpy()
        return real["real_clip"](parameters, max_norm, *a, **k)

    def wrap_loss(name):
        def f(*a, **k):
            r = real["real_losses"][name](*a, **k)
            rec["losses"][name] = float(r[0] if isinstance(r, tuple) else r)
            return r
        return f

    seed = 1000
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    ref_model.MaskRCNN.predict = rec_predict
    torch.nn.utils.clip_grad_norm_ = rec_clip
    torch.save = lambda *a, **k: None
    torch.utils.data.DataLoader = lambda d, **k: real["real_loader"](d, **dict(k, num_workers=0))
    for n in loss_names:
        setattr(ref_model, n, wrap_loss(n))
    try:
        model.train_model(ds, ds, 0.01, 1, "all")
    finally:
        ref_model.MaskRCNN.predict = real["real_predict"]
        torch.nn.utils.clip_grad_norm_ = real["real_clip"]
        torch.save = real["real_save"]
        torch.utils.data.DataLoader = real["real_loader"]
        for n in loss_names:
            setattr(ref_model, n, real["real_losses"][n])
        for h in hooks:
            h.remove()
    base = np.load(os.path.join(ROOT, "tests", "golden", "e2e_train_%d.npz" % scene))
    got = np.array([rec["losses"][n] for n in loss_names], dtype=np.float64)
    assert np.array_equal(got, base["losses"]), (got, base["losses"])
    for n in WATCH:      # the same step as the e2e_train_0 fixture, bit for bit
        assert np.array_equal(rec["grad"][n][:SLICE], base["grad/" + n]), n
    arrs, order, counts = {}, [], {}
    for name, m in rec["relu"]:
        k = counts.get(name, 0)
        counts[name] = k + 1
        key = "%s#%d" % (name, k)
        order.append(key)
        arrs["relu/" + key] = np.packbits(m, axis=None)
        arrs["relu_shape/" + key] = np.array(m.shape)
    bits = sum(int(np.prod(arrs["relu_shape/" + k])) for k in order)
    print("relu calls %d, %.1f M units; pool taps %s" % (len(order), bits / 1e6, rec["pool"].shape))
    save("e2e_relu_masks_0", relu_order=np.array(order), pool_tap=rec["pool"],
         pool_in_shape=np.array(rec["pool_in_shape"]), grad_images=rec["grad_images"],
         names=np.array(MASK_WATCH), grad_norms=np.array([rec["grad_norms"][n] for n in MASK_WATCH]),
         **{"grad/" + n: v for n, v in rec["grad"].items()}, **arrs)


def run_multistep(env, steps=5, scene=0):
    """Scene 0 through `steps` consecutive optimiser steps of the reference's own loop (train_model with
    epochs = steps, STEPS_PER_EPOCH = 1: one optimiser, momentum carried over; model.py:356-366, 383-444).  Per
    step the observers record what the loop consumed (the augmented image / boxes / RPN targets the DataLoader
    produced, the two randperm draws, the proposals) and what it produced (the six losses, the clip norm, the
    watched parameter slices after the update) -> tests/golden/e2e_multistep_0.npz.  Step 0 must be the
    e2e_train_0 fixture, bit for bit (checked)."""
    ref_model, ref_config, ref_dl, nn = env["ref_model"], env["ref_config"], env["ref_dl"], env["nn"]
    ref_train, scenes, tmp, loss_names = env["ref_train"], env["scenes"], env["tmp"], env["loss_names"]
    real = {k: env[k] for k in ("real_loader", "real_randperm", "real_clip", "real_step", "real_save",
                                "real_predict", "real_losses")}
    image, label = scenes[scene]
    model, cfg = build_reference_model(ref_model, ref_config, ref_dl, nn)
    params = dict(model.named_parameters())
    ds = StubDataset(ref_train, [image], [label], tmp)
    log = []
    cur = {}

    def rec_randperm(n, *a, **k):
        p = real["real_randperm"](n, *a, **k)
        cur.setdefault("perms", []).append(p.numpy().copy())
        return p

    real_proposal = ref_model.proposal_layer

    def rec_proposal(*a, **k):
        r = real_proposal(*a, **k)
        cur["rpn_rois"] = r.detach().clone().numpy()
        return r

    def rec_predict(self, input, mode):
        cur["inputs"] = [t.detach().clone() if torch.is_tensor(t) else np.array(t) for t in input]
        return real["real_predict"](self, input, mode)

    def rec_clip(parameters, max_norm, *a, **k):
        total = real["real_clip"](parameters, max_norm, *a, **k)
        cur["total_norm"] = float(total)
        return total

    def rec_step(self, *a, **k):
        r = real["real_step"](self, *a, **k)
        cur["after"] = grab(params, "data")
        log.append(dict(cur))
        cur.clear()
        return r

    def wrap_loss(name):
        def f(*a, **k):
            r = real["real_losses"][name](*a, **k)
            cur.setdefault("losses", {})[name] = float(r[0] if isinstance(r, tuple) else r)
            if name == "compute_rpn_class_loss":
                cur["rpn_match"] = a[0].detach().clone().numpy()
            if name == "compute_rpn_bbox_loss":
                cur["rpn_bbox_t"] = a[0].detach().clone().numpy()
            return r
        return f

    before = grab(params, "data")
    seed = 1000 + scene
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    torch.randperm = rec_randperm
    ref_model.proposal_layer = rec_proposal
    ref_model.MaskRCNN.predict = rec_predict
    torch.nn.utils.clip_grad_norm_ = rec_clip
    torch.optim.SGD.step = rec_step
    torch.save = lambda *a, **k: None
    torch.utils.data.DataLoader = lambda d, **k: real["real_loader"](d, **dict(k, num_workers=0))
    for n in loss_names:
        setattr(ref_model, n, wrap_loss(n))
    try:
        model.train_model(ds, ds, 0.01, steps, "all")
    finally:
        torch.randperm = real["real_randperm"]
        ref_model.proposal_layer = real_proposal
        ref_model.MaskRCNN.predict = real["real_predict"]
        torch.nn.utils.clip_grad_norm_ = real["real_clip"]
        torch.optim.SGD.step = real["real_step"]
        torch.save = real["real_save"]
        torch.utils.data.DataLoader = real["real_loader"]
        for n in loss_names:
            setattr(ref_model, n, real["real_losses"][n])
    assert len(log) == steps, len(log)
    molded0 = image.astype(np.float32) - cfg.MEAN_PIXEL
    arrs = {}
    for k, st in enumerate(log):
        images, _metas, gt_class_ids, gt_boxes, gt_layer = st["inputs"]
        got = images[0].permute(1, 2, 0).numpy()
        flipped = not np.allclose(got, molded0)
        if flipped:
            assert np.allclose(got, molded0[:, ::-1])
        assert len(st["perms"]) == 3 and len(st["perms"][0]) == 1, [len(p) for p in st["perms"]]
        losses = np.array([st["losses"][n] for n in loss_names], dtype=np.float64)
        print("step %d: flipped=%s losses=%s total=%.6f grad norm=%.4f rois=%d" % (
            k, flipped, np.round(losses, 5).tolist(), losses.sum(), st["total_norm"], st["rpn_rois"].shape[1]))
        arrs.update({"s%d/flipped" % k: np.array(flipped), "s%d/gt_boxes" % k: gt_boxes.numpy(),
                     "s%d/gt_class_ids" % k: gt_class_ids.numpy(), "s%d/rpn_match" % k: st["rpn_match"],
                     "s%d/rpn_bbox_target" % k: st["rpn_bbox_t"], "s%d/perm_pos" % k: st["perms"][1],
                     "s%d/perm_neg" % k: st["perms"][2], "s%d/rpn_rois" % k: st["rpn_rois"],
                     "s%d/losses" % k: losses, "s%d/total_norm" % k: np.array(st["total_norm"])})
        arrs.update({"s%d/after/%s" % (k, n): v for n, v in st["after"].items()})
    base = np.load(os.path.join(ROOT, "tests", "golden", "e2e_train_%d.npz" % scene))
    assert np.array_equal(arrs["s0/losses"], base["losses"]), (arrs["s0/losses"], base["losses"])
    for n in WATCH:
        assert np.array_equal(arrs["s0/after/" + n], base["after/" + n]), n
    save("e2e_multistep_%d" % scene, native=np.array("oracle"), dim=np.array(DIM), lr=np.array(0.01), steps=np.array(steps),
         image_u8=image, label=label, loss_names=np.array(loss_names), names=np.array(WATCH),
         **{"before/" + n: v for n, v in before.items()}, **arrs)


if __name__ == "__main__":
    main(only_masks="--masks" in sys.argv, multistep="--steps" in sys.argv,
         dim=int(sys.argv[sys.argv.index("--dim") + 1]) if "--dim" in sys.argv else None)
