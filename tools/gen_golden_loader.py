"""Real-data loader goldens: the reference's OWN load_image_gt (modal/Functions.py:675-736) and
model.Dataset.__getitem__ (model.py:80-116) run in the build container on a NON-SQUARE uint8 image and
its uint64 'layer' label, through tools/ref_harness.py.

    python tools/gen_golden_loader.py     # writes tests/golden/loader_{0,1}.npz

Observers only: random.randint (the flip, Functions.py:713), np.random.rand (the box jitter,
utils.py:51) and np.random.choice (the surplus anchors build_rpn_targets drops, Functions.py:804, 812)
are wrapped to RECORD what they return.  Third-party stand-in: scipy.misc.imresize as published in
scipy 1.0 (tools/gen_golden_e2e.py:imresize) over the installed Pillow; scipy.ndimage.zoom is the
installed scipy's.  Data only: the image, the label, the draws and what the reference returned.
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
from tools.gen_golden_e2e import imresize  # noqa: E402

DIM = 128
SCENES = [(96, 160, 5, 11), (200, 150, 6, 14)]      # (H0, W0, objects, seed): up- and down-scaling per axis


def main():
    ref_harness.install()
    import scipy.misc
    scipy.misc.imresize = imresize
    import config as ref_config
    import model as ref_model
    import utils as ref_utils
    import amodal_train as ref_train
    import modal.Functions as ref_F
    ref_utils.scipy.misc.imresize = imresize

    class Cfg(ref_config.Config):
        NAME = "golden"
        GPU_COUNT = 0
        IMAGE_MAX_DIM = DIM
        IMAGE_MIN_DIM = DIM
        NUM_CLASSES = 1 + 1
        EXPERIMENT_DIR = tempfile.mkdtemp()

    cfg = Cfg()
    tmp = tempfile.mkdtemp()
    for si, (H0, W0, n_obj, seed) in enumerate(SCENES):
        rng = np.random.RandomState(seed)
        image = rng.randint(0, 256, (H0, W0, 3)).astype(np.uint8)
        # smooth content as well, so that the bilinear taps matter
        yy, xx = np.mgrid[0:H0, 0:W0]
        image[..., 1] = (127 + 120 * np.sin(yy / 7.0) * np.cos(xx / 5.0)).astype(np.uint8)
        label, _ = synth_label(rng, H0, W0, n_obj)

        class DS(object):
            image_ids = np.arange(1)
            image_info = [{"path": os.path.join(tmp, "img%d.jpg" % si), "height": H0, "width": W0}]

            def load_image(self, image_id):
                return image

            def load_layer2(self, image_id, config):
                return ref_train.AmodalDataset.load_layer2(self, image_id, config)

        np.savez(os.path.join(tmp, "img%d.npz" % si), layer=label)
        ds = DS()
        rec = {"flip": [], "jitter": [], "choice": []}
        real_randint, real_rand, real_choice = random.randint, np.random.rand, np.random.choice

        def rec_randint(a, b):
            r = real_randint(a, b)
            rec["flip"].append(r)
            return r

        def rec_rand(*shape):
            r = real_rand(*shape)
            rec["jitter"].append(np.asarray(r).copy())
            return r

        def rec_choice(ids, extra, replace=False):
            r = real_choice(ids, extra, replace=replace)
            rec["choice"].append(np.asarray(r).copy())
            return r

        random.seed(seed + si); np.random.seed(seed)
        random.randint, np.random.rand, np.random.choice = rec_randint, rec_rand, rec_choice
        try:
            item = ref_model.Dataset(ds, cfg, augment=True)[0]
        finally:
            random.randint, np.random.rand, np.random.choice = real_randint, real_rand, real_choice
        images, image_metas, rpn_match, rpn_bbox, gt_class_ids, gt_boxes, gt_layer, image_raw = item
        assert len(rec["flip"]) == 1
        jitter = np.stack(rec["jitter"])
        assert jitter.shape == (gt_boxes.shape[0], 4), jitter.shape
        # the same call with the same draws through load_image_gt directly (what Dataset wraps)
        random.seed(seed + si); np.random.seed(seed)
        image_r, meta_r, ids_r, bbox_r, layers_r = ref_F.load_image_gt(ds, cfg, 0, augment=True)
        assert np.array_equal(bbox_r, gt_boxes.numpy().astype(np.int32))
        print("scene %d: %dx%d -> %d, flip=%d, N=%d, boxes\n%s\nrpn positives %d, dropped sets %s" %
              (si, H0, W0, DIM, rec["flip"][0], gt_boxes.shape[0], bbox_r, int((rpn_match == 1).sum()),
               [len(c) for c in rec["choice"]]))
        save("loader_%d" % si, dim=np.array(DIM), image_u8=image, label=label, flip=np.array(rec["flip"][0]),
             jitter=jitter, n_choice=np.array(len(rec["choice"])),
             **{"choice%d" % i: c for i, c in enumerate(rec["choice"])},
             out_image_u8=np.ascontiguousarray(image_r), out_meta=meta_r, out_class_ids=ids_r, out_bbox=bbox_r,
             out_mask_layers=np.packbits(layers_r, axis=None), out_mask_layers_shape=np.array(layers_r.shape),
             images=images.numpy(), image_metas=image_metas.numpy(), rpn_match=rpn_match.numpy(),
             rpn_bbox=rpn_bbox.numpy(), gt_class_ids=gt_class_ids.numpy(), gt_boxes=gt_boxes.numpy(),
             gt_layer=np.packbits(gt_layer.numpy(), axis=None), gt_layer_shape=np.array(gt_layer.shape))


if __name__ == "__main__":
    main()
