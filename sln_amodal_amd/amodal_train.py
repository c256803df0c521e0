"""CLI entry point with the reference's surface (amodal_train.py:507-675):

    python amodal_train.py train    --dataset DIR|--synthetic --model PATH --logs DIR --limit N
    python amodal_train.py evaluate --dataset DIR|--synthetic --model PATH --limit N

`Amodalfig` (amodal_train.py:38-54) and the head surgery (606-614) are kept; the
three training stages (642-663: heads x2, 4+ x3, all @ lr/10) are kept.  Dataset I/O
is reduced to what the hot path consumes: an image + the uint64 'layer' label from
the reference's `<name>.npz` files (`np.load(path)['layer']`, amodal_train.py:238),
or synthetic COCOA-shape batches generated on the device.
"""
import argparse
import glob
import os

import numpy as np
import torch

from . import parallel, synthetic
from .config import Config
from .model import MaskRCNN
from .modal.Functions import build_rpn_targets, extract_bboxes_from_labels

COCO_MODEL_PATH = "./checkpoints/mask_rcnn_coco.pth"
GLM_MODEL_PATH = "./checkpoints/deeplabv2.pth"
DEFAULT_LOGS_DIR = os.path.join(os.getcwd(), "logs")


class Amodalfig(Config):
    NAME = "coco"
    IMAGES_PER_GPU = 16
    BATCH_SIZE = 16
    NUM_CLASSES = 1 + 80  # replaced by 1 + 1 in apply_amodal_heads(), as in the reference


class InferenceConfig(Amodalfig):
    GPU_COUNT = 1
    IMAGES_PER_GPU = 1
    BATCH_SIZE = 1
    DETECTION_MIN_CONFIDENCE = 0


class AmodalDataset(object):
    """The dataset of `<name>.npz` ('layer': the uint64 occlusion label, amodal_train.py:238) + `<name>.jpg`
    pairs.  Two faces:
      * the reference's per-image methods (`image_ids`, `image_info`, `load_image`, `load_layer2`) that
        model.Dataset / Functions.load_image_gt call -- host side, numpy;
      * an iterable of device-resident training BATCHES (dicts, see MaskRCNN.train_step) -- what the
        train loop consumes, fed by worker processes through pinned staging buffers in a per-epoch shuffled,
        rank-sharded order (loader.py; the reference: DataLoader(shuffle=True, num_workers=4), model.py:340-342).
        Per image it follows load_image_gt (Functions.py:675-736) step by step:
        uint8 squash to IMAGE_MAX_DIM^2 (Pillow BILINEAR = scipy.misc.imresize), nearest zoom of the label
        with scipy.ndimage.zoom's index map (the label, not its planes, is resized: the decode is per
        pixel), `random.randint` flip, boxes jittered by `np.random.rand(4)` per instance, class id 1 for
        every object of the ORIGINAL label; labels stay uint64 on the device (the planes are never built),
        boxes and RPN targets are computed there."""

    def __init__(self, config, model, root=None, limit=-1, seed=1234, device="cuda", max_objects=None,
                 rank=0, world=1, augment=True, workers=None, prefetch=3, shuffle=True, seed_order=0):
        self.config, self.model, self.device, self.seed = config, model, device, seed
        self.rank, self.world = rank, world   # data-parallel shard of the file list
        self.augment = augment
        # input pipeline (loader.py): worker processes (0 = load on the training thread), batches in flight,
        # per-epoch shuffle seeded by seed_order -- the SAME on every rank (the ranks cut one permutation)
        self.workers = int(os.environ.get("SLN_LOADER_WORKERS", "8")) if workers is None else int(workers)
        self.prefetch, self.shuffle, self.seed_order = int(prefetch), bool(shuffle), int(seed_order)
        self._pipe, self._zoom_maps, self._mean = None, {}, None
        self._overflow = 0
        self._jitter_rng = np.random.RandomState((seed * 7919 + 13) % (2 ** 32))
        self.files = []
        if root:
            self.files = sorted(glob.glob(os.path.join(root, "**", "*.npz"), recursive=True))
            if limit and limit > 0:
                self.files = self.files[:limit]
        # object slots per image: the label format holds 32 objects; synthetic scenes are drawn with 8
        self.max_objects = max_objects or (32 if self.files else 8)
        self.image_info = []
        for i, p in enumerate(self.files):
            stem = p[:-4]
            img = next((stem + e for e in (".jpg", ".png", ".jpeg") if os.path.exists(stem + e)), stem + ".jpg")
            self.image_info.append({"id": i, "source": "amodal", "path": img, "label": p})
        self.image_ids = np.arange(len(self.files))

    # ---------------------------------------------------------------- the reference's per-image surface
    def _layer(self, image_id):
        info = self.image_info[image_id]
        return np.load(info["path"][:-4] + ".npz")["layer"].astype(np.uint64)

    def load_image(self, image_id):
        from PIL import Image
        info = self.image_info[image_id]
        if os.path.exists(info["path"]):
            return np.asarray(Image.open(info["path"]).convert("RGB"))
        return np.zeros(self._layer(image_id).shape + (3,), np.uint8)

    def load_layer2(self, image_id, config):
        """[H,W,L,N] bool planes + class ids (all 1), amodal_train.py:236-271."""
        from .modal.Functions import label_planes_host
        return label_planes_host(self._layer(image_id), config.NUM_CLASSES - 1)

    # ---------------------------------------------------------------- device-resident batches
    def _assemble(self, u8, labels, src_hw, src_hw_host, flips, jitter, host_zoomed=None, rpn_priority=None):
        """The device half of load_image_gt (Functions.py:675-736) for a whole batch, no host sync:
        u8 [B,dim,dim,3] uint8 (already flipped), labels [B, stride] int64 -- the ORIGINAL labels, image b with
        src_hw[b] = (H0, W0) -- flips (host list), jitter [B,N,4] float64 uniform draws (row i = object i).
        Nearest zoom of the label on the device with scipy's index maps (a flip = the reversed column map), object
        count of the original label, molding, tight boxes + jitter, class ids, RPN targets."""
        from . import ops, utils
        dev, dim, N = self.device, self.config.IMAGE_MAX_DIM, self.max_objects
        B = u8.shape[0]
        ys = np.empty((B, dim), np.int32)
        xs = np.empty((B, dim), np.int32)
        for b in range(B):
            h0, w0 = int(src_hw_host[b][0]), int(src_hw_host[b][1])
            key = (h0, w0, dim)
            maps = self._zoom_maps.get(key)
            if maps is None:
                # utils.resize_layer: output size round(n * scale), scale = dim / n -- dim for every n.  (A label the
                # worker already zoomed -- one that did not fit a loader slot -- arrives at dim x dim: identity maps.)
                oh = utils.zoom_output_size(h0, dim / h0)
                ow = utils.zoom_output_size(w0, dim / w0)
                if (oh, ow) != (dim, dim):
                    raise ValueError("label %dx%d does not zoom to %d^2 (%dx%d)" % (h0, w0, dim, oh, ow))
                maps = (utils.zoom_nearest_index(h0, dim).astype(np.int32),
                        utils.zoom_nearest_index(w0, dim).astype(np.int32))
                if len(self._zoom_maps) < 4096:
                    self._zoom_maps[key] = maps
            ys[b] = maps[0]
            xs[b] = maps[1][::-1] if flips[b] else maps[1]       # the label in the slot is never flipped
        idx = torch.from_numpy(np.concatenate([ys, xs], axis=1))
        idx = (idx.pin_memory() if torch.device(dev).type == "cuda" else idx).to(dev, non_blocking=True)
        labels_z = ops.label_zoom(labels, src_hw, idx[:, :dim].contiguous(), idx[:, dim:].contiguous())
        counts = ops.label_num_objects_ragged(labels, src_hw).to(torch.int64)
        # more objects than slots: counted on the device, read where the loop syncs anyway (health())
        self._overflow = self._overflow + (counts > N).sum()
        counts = counts.clamp(max=N)
        # mold_image (Functions.py:654-660): uint8 -> float32, minus the float64 mean pixel (the difference is
        # formed in float64 and rounded to float32 once, by the later .float())
        mean = self._mean
        if mean is None:
            mean = self._mean = torch.tensor(np.asarray(self.config.MEAN_PIXEL, np.float64), dtype=torch.float64,
                                             device=dev)
        images = (u8.double() - mean).float().permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        tight = extract_bboxes_from_labels(labels_z, N)
        u = torch.from_numpy(np.ascontiguousarray(jitter, np.float64))
        u = (u.pin_memory() if torch.device(dev).type == "cuda" else u).to(dev, non_blocking=True)
        boxes = utils.jitter_boxes(tight, u)
        ids = (torch.arange(N, device=dev)[None, :] < counts[:, None]).to(torch.int32)
        boxes = torch.where(ids[..., None] > 0, boxes, torch.zeros_like(boxes)).float()
        match, bbox = build_rpn_targets((dim, dim, 3), self.model.anchors_f64, ids, boxes, self.config,
                                        priority=rpn_priority)
        return {"images": images, "image_metas": None, "gt_class_ids": ids, "gt_boxes": boxes,
                "gt_layer": labels_z, "rpn_match": match.unsqueeze(2), "rpn_bbox": bbox, "flipped": list(flips)}

    def _load_real(self, image_ids, draws=None, flips=None):
        """Batch dict of the given images, loaded on THIS thread (parity tests, calibration, tiny runs; the train loop
        goes through the worker pipeline, see __iter__).  draws (parity tests): per image {"flip", "jitter" [n,4],
        "rpn_priority" [A]} replaying the reference's RNG; default: `random` / `np.random` like the reference's
        worker."""
        import random
        from . import loader
        dim, N = self.config.IMAGE_MAX_DIM, self.max_objects
        B = len(image_ids)
        imgs, labs, fl = [], [], []
        jitter = np.zeros((B, N, 4), np.float64)
        for k, iid in enumerate(image_ids):
            d = draws[k] if draws is not None else None
            u8, layer = loader.load_item(self.image_info[iid], dim)
            flip = 0
            if self.augment:
                flip = (random.randint(0, 1) if flips is None else int(flips[k])) if d is None else int(d["flip"])
            imgs.append(np.ascontiguousarray(u8[:, ::-1] if flip else u8))
            labs.append(layer)
            fl.append(flip)
            if d is None:
                jitter[k] = np.random.rand(N, 4)
            else:       # one draw of np.random.rand(4) per object of the ORIGINAL label, in order (utils.extract_bboxes)
                jd = np.asarray(d["jitter"], np.float64).reshape(-1, 4)
                if jd.shape[0] > N:
                    raise ValueError("image %s has %d objects, the batch holds %d slots" % (iid, jd.shape[0], N))
                jitter[k, :jd.shape[0]] = jd
        dev = self.device
        hw = np.array([l.shape for l in labs], np.int32)
        stride = int((max(int(h) * int(w) for h, w in hw) + 7) // 8 * 8)
        flat = np.zeros((B, stride), np.int64)
        for k, l in enumerate(labs):
            flat[k, :l.size] = l.reshape(-1).view(np.int64)
        pr = None
        if draws is not None and all("rpn_priority" in d for d in draws):
            pr = torch.stack([torch.as_tensor(d["rpn_priority"]) for d in draws]).to(dev)
        return self._assemble(torch.from_numpy(np.stack(imgs)).to(dev), torch.from_numpy(flat).to(dev),
                              torch.from_numpy(hw).to(dev), hw, fl, jitter, rpn_priority=pr)

    # ---------------------------------------------------------------- the worker pipeline
    def start_workers(self):
        """Spawn the loader's worker processes now -- call before the process touches the GPU when possible
        (amodal_train.main and bench.py do); __iter__ starts them otherwise."""
        if self.files and self.workers > 0 and self._pipe is None:
            from . import loader
            sampler = loader.EpochSampler(len(self.files), self.config.BATCH_SIZE, self.rank, self.world,
                                          seed=self.seed_order, shuffle=self.shuffle, flip=self.augment)
            self._pipe = loader.PrefetchLoader(self.image_info, self.config.IMAGE_MAX_DIM, self.config.BATCH_SIZE,
                                               sampler, workers=self.workers, depth=self.prefetch,
                                               device=self.device).start()
        return self

    def bind(self, model, device):
        """Late binding for a dataset whose workers were started before the process touched the GPU (the model --
        its anchors -- and the device did not exist yet)."""
        self.model, self.device = model, device
        if self._pipe is not None:
            self._pipe.device = device
        return self

    def close(self):
        if self._pipe is not None:
            self._pipe.close()
            self._pipe = None

    def loader_report(self):
        """Queue depth / consumer wait of the worker pipeline (None without one) + images with more objects than
        slots (host sync)."""
        rep = self._pipe.report() if self._pipe is not None else None
        if rep is not None:
            rep["images_over_object_slots"] = int(self._overflow)
        return rep

    def queue_depth(self):
        return self._pipe.queue_depth() if self._pipe is not None else None

    def __iter__(self):
        B, dim, step = self.config.BATCH_SIZE, self.config.IMAGE_MAX_DIM, 0
        if self.files and self.workers > 0:
            # worker processes -> shared-memory slots -> pinned staging -> copy stream; this thread waits on an
            # event and runs the device half (loader.py)
            self.start_workers()
            N = self.max_objects
            # (round 6) the device half of batch k + 1 -- label zoom, object count, molding, boxes, RPN targets: ~5 ms
            # of small kernels for 16 images -- is enqueued on a SIDE stream while train step k runs on the training
            # stream: a one-deep pipeline inside this iterator, no extra thread.  The consumer's stream waits on the
            # batch's event only.  SLN_LOADER_ASSEMBLE_STREAM=0: on the training stream, as in round 5.
            cuda = torch.device(self.device).type == "cuda"
            side = None
            if cuda and os.environ.get("SLN_LOADER_ASSEMBLE_STREAM", "1") != "0":
                side = getattr(self, "_assemble_stream", None)
                if side is None:
                    side = self._assemble_stream = torch.cuda.Stream(device=self.device)

            def assemble(item):
                jitter = self._jitter_rng.rand(B, N, 4)
                if side is None:
                    if item["ready"] is not None:
                        torch.cuda.current_stream(self.device).wait_event(item["ready"])
                        for t in (item["u8"], item["labels"], item["src_hw"]):
                            t.record_stream(torch.cuda.current_stream(self.device))
                    return self._assemble(item["u8"], item["labels"], item["src_hw"], item["src_hw_host"],
                                          item["flips"], jitter, host_zoomed=item["host_zoomed"]), None
                with torch.cuda.stream(side):
                    if item["ready"] is not None:
                        side.wait_event(item["ready"])
                    for t in (item["u8"], item["labels"], item["src_hw"]):
                        t.record_stream(side)
                    batch = self._assemble(item["u8"], item["labels"], item["src_hw"], item["src_hw_host"],
                                           item["flips"], jitter, host_zoomed=item["host_zoomed"])
                    done = torch.cuda.Event()
                    done.record(side)
                return batch, done

            def hand_over(batch, done):
                if done is not None:
                    main = torch.cuda.current_stream(self.device)
                    main.wait_event(done)
                    for v in batch.values():        # allocated on the side stream, read (and freed) on this one
                        if torch.is_tensor(v):
                            v.record_stream(main)
                return batch

            ahead = None
            for item in self._pipe:
                nxt = assemble(item)
                if side is None:
                    yield hand_over(*nxt)
                    continue
                if ahead is not None:
                    yield hand_over(*ahead)
                ahead = nxt
            if ahead is not None:
                yield hand_over(*ahead)
            return
        sampler = None
        if self.files:
            from . import loader
            sampler = loader.EpochSampler(len(self.files), B, self.rank, self.world, seed=self.seed_order,
                                          shuffle=self.shuffle, flip=self.augment)
        epoch, plan = 0, []
        while True:
            if self.files:
                if not plan:
                    plan = sampler.epoch(epoch)
                    epoch += 1
                ids, flips = plan.pop(0)
                yield self._load_real([int(i) for i in ids], flips=flips)
            else:
                yield synthetic.make_batch(self.config, B, dim, dim, n_obj=self.max_objects,
                                           seed=self.seed + step, device=self.device,
                                           anchors_f64=self.model.anchors_f64)
            step += 1


def build_coco_results(dataset, image_ids, rois, class_ids, scores, masks, rles=None):
    """Detections of one image in COCO result format (amodal_train.py:370-400): bbox rounded to one
    decimal and reordered to (x, y, w, h), class ids folded to {0, 1}, masks as COCO RLE dicts.
    masks: device uint8 [N,W,H] (detect(..., keep_device=True)["masks_device"]) or the host array
    [H,W,N] detect() returns by default (uploaded once); either way the run-length encoding happens
    on the GPU (mask_rle.encode replaces maskUtils.encode(np.asfortranarray(mask))).  rles: the RLE dicts
    themselves, when the batched hand-off (tail.InferenceTail) already encoded them."""
    from . import mask_rle
    if rois is None:
        return []
    if rles is None:
        if not torch.is_tensor(masks):
            masks = torch.from_numpy(np.ascontiguousarray(np.transpose(masks, (2, 1, 0)))).to(
                torch.uint8).cuda()
        rles = mask_rle.encode(masks)
    results = []
    for image_id in image_ids:
        for i in range(rois.shape[0]):
            bbox = np.around(rois[i], 1)
            results.append({
                "image_id": image_id,
                "category_id": 1 if class_ids[i] > 0 else 0,
                "bbox": [bbox[1], bbox[0], bbox[3] - bbox[1], bbox[2] - bbox[0]],
                "score": scores[i],
                "segmentation": rles[i],
            })
    return results


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Train / evaluate SLN-Amodal on MI355X.")
    ap.add_argument("command", metavar="<command>", help="'train' or 'evaluate'")
    ap.add_argument("--dataset", default=None, metavar="/path/to/dataset/")
    ap.add_argument("--synthetic", action="store_true", help="synthetic COCOA-shape batches")
    ap.add_argument("--model", default="none", metavar="/path/to/weights.pth | last | none")
    ap.add_argument("--logs", default=DEFAULT_LOGS_DIR, metavar="/path/to/logs/")
    ap.add_argument("--limit", default=500, type=int)
    ap.add_argument("--data_type", default="COCOA")
    ap.add_argument("--arch", default="resnet101", choices=["resnet50", "resnet101"])
    ap.add_argument("--image-dim", default=1024, type=int)
    ap.add_argument("--batch", default=None, type=int, help="images per GPU")
    ap.add_argument("--steps-per-epoch", default=None, type=int)
    ap.add_argument("--mask-positive-slots", action="store_true",
                    help="train: run the mask head on the positive roi slots only (MaskRCNN.mask_train_slots): the same "
                         "losses and gradients without the mask branch of the negative rois (the reference computes it "
                         "for all sampled rois, model.py:664-700); off by default")
    ap.add_argument("--workers", default=None, type=int,
                    help="loader worker processes per rank (default 8, SLN_LOADER_WORKERS; 0 = load on the training "
                         "thread; the reference: DataLoader(num_workers=4), model.py:340-342)")
    return ap.parse_args(argv)


def build_run(args):
    """Everything of a run that needs no GPU -- the reference's plumbing around the hot path
    (amodal_train.py:560-640): the command's config class with the CLI overrides, the model with the
    head surgery, the checkpoints that exist, GLM frozen.  Returns (config, model, model_path); the model
    is still on the host.  BASELINE.json configs[0] (`evaluate`, ResNet-50, 2 x 512^2) exercises exactly
    this on CPU in tests/test_model_cpu.py; main() then needs the MI355X."""
    if args.command not in ("train", "evaluate"):
        raise SystemExit("'{}' is not recognized. Use 'train' or 'evaluate'".format(args.command))
    base = Amodalfig if args.command == "train" else InferenceConfig

    class RunConfig(base):
        IMAGE_MAX_DIM = args.image_dim
        IMAGE_MIN_DIM = args.image_dim
        ARCHITECTURE = args.arch

    config = RunConfig()
    if args.batch:
        config.BATCH_SIZE = args.batch
    if args.steps_per_epoch:
        config.STEPS_PER_EPOCH = args.steps_per_epoch
    if int(os.environ.get("RANK", "0")) == 0:
        config.display()
    torch.manual_seed(0)
    model = MaskRCNN(config=config, model_dir=args.logs)
    model_path = args.model
    if model_path.lower() == "last":
        model_path = model.find_last()[1]
    if args.command == "train" and model_path and model_path.lower() != "none":
        model.load_weights(model_path)              # before the head surgery, like the reference
    model.apply_amodal_heads()
    if os.path.exists(GLM_MODEL_PATH):
        model.GLM_modual.load_state_dict(torch.load(GLM_MODEL_PATH, map_location="cpu"))
    if args.command != "train" and model_path and model_path.lower() != "none":
        model.load_weights(model_path)
    for p in model.GLM_modual.parameters():
        p.requires_grad = False
    return config, model, model_path


def main(argv=None):
    args = parse_args(argv)
    config, model, model_path = build_run(args)
    # the input pipeline's worker processes start BEFORE this process touches the GPU (rank / world from the
    # launcher's environment; parallel.init_distributed reads the same variables)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    data = AmodalDataset(config, model, None if args.synthetic else args.dataset, args.limit,
                         seed=1234 + rank, device="cpu", rank=rank, world=world, workers=args.workers,
                         shuffle=args.command == "train", augment=args.command == "train")
    if args.command == "train":
        parallel.set_cpu_affinity(int(os.environ.get("LOCAL_RANK", "0")),
                                  int(os.environ.get("LOCAL_WORLD_SIZE", world)))   # the workers inherit the share
        data.start_workers()
    if not torch.cuda.is_available():
        data.close()
        raise RuntimeError("amodal_train %s: the run itself needs an MI355X -- sln_amodal_amd has no CPU path "
                           "(NMS, RoIAlign, the conv stacks and the inference tail live in "
                           "libsln_amodal_hip.so); the plumbing up to here (config, model, checkpoints) ran" %
                           args.command)
    rank, local, world = parallel.init_distributed()
    device = torch.device("cuda", local)
    torch.cuda.set_device(local)
    model.to(device)
    data.bind(model, device)

    if args.command == "train" and not os.path.exists(str(model_path)):
        # no checkpoint: emulate pretrained statistics.  Calibrate FIRST (the batch is rank-specific),
        # broadcast AFTER, so that every replica computes with rank 0's frozen-BN statistics.
        first = next(iter(data))
        synthetic.calibrate_batchnorm(model, first["images"][:4])
        synthetic.calibrate_glm(model, first["images"][:2])
    parallel.broadcast_parameters(model)

    if args.command == "train":
        if args.mask_positive_slots:
            model.mask_train_slots = model.positive_slots()
        params = lambda: [p for p in model.parameters() if p.requires_grad]
        reducer = None
        for lr, epochs, layers in ((config.LEARNING_RATE, 2, "heads"), (config.LEARNING_RATE, 3, "4+"),
                                   (config.LEARNING_RATE / 10, 1, "all")):
            model.set_trainable(".*", exclusive_off=False)   # the reference can only switch off
            for p in model.GLM_modual.parameters():
                p.requires_grad = False
            from .model import LAYER_REGEX
            model.set_trainable(LAYER_REGEX[layers])
            if reducer is not None:
                reducer.detach()     # the previous stage's hooks must not fire next to the new reducer's
            reducer = parallel.GradientAllReducer(params()).attach()
            model.train_model(data, None, learning_rate=lr, epochs=epochs, layers=layers,
                              grad_sync=reducer if world > 1 else None)      # (the reducer is callable: finish())
    elif args.command == "evaluate":
        # one batched predict(mode='inference') per BATCH_SIZE images (the reference: one image at a time,
        # amodal_train.py:403-466); the per-image loop is only the hand-off to the COCO result list
        # (round 6) the hand-off runs on a worker thread and a side stream (tail.InferenceTail): the whole batch
        # through one unmold and one run-length launch while the next batch's forward is already enqueued
        from .tail import InferenceTail
        tail = InferenceTail(model.anchors.device)
        n, limit = 0, max(1, min(args.limit, 2 if args.synthetic else args.limit))
        for batch in data:
            k_img = min(batch["images"].shape[0], limit - n)
            imgs = [(batch["images"][b].permute(1, 2, 0).cpu().numpy() + config.MEAN_PIXEL).clip(0, 255)
                    .astype(np.uint8) for b in range(k_img)]
            model.detect_submit(imgs, tail, keys=list(range(n, n + k_img)))
            n += k_img
            if n >= limit:
                break
        res = tail.results()
        tail.close()
        for i in range(n):
            r = res.get(i)
            k = r["rois"].shape[0] if r is not None else 0
            coco = build_coco_results(None, [i], r["rois"], r["class_ids"], r["scores"], None, rles=r["rles"]) if k else []
            if rank == 0:
                print("image %d: %d detections, %d RLE bytes" % (
                    i, k, sum(len(c["segmentation"]["counts"]) for c in coco)))
    data.close()


if __name__ == "__main__":
    main()
