"""MaskRCNN orchestration with the reference's entry points (model.py:126-806):
build / initialize_weights / set_trainable / load_weights / find_last /
train_model / train_epoch / predict / detect / mold_inputs / unmold_detections.

What is different from the reference, by design:
  * batched: `predict` takes B images (the reference is batch 1, model.py:341,442);
    per-image semantics (per-image NMS, sampling, loss means) are kept and the
    step loss is the mean over images;
  * no device->host sync inside a training step: fixed-capacity roi slots with
    validity masks replace `len()`, `.any()`, `nonzero` control flow;
  * native ops (NMS, crop_and_resize, pyramid crop, label decode / mask targets,
    proposal decode) run through the C ABI in csrc/; convolutions go through
    nn_ops.conv_bn_act;
  * data-parallel training: one process per GPU, gradients all-reduced over RCCL
    (parallel.py) -- the reference has no live multi-GPU path.
"""
import os
import re

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nn_ops, utils
from .modal import loss as L
from .modal.Functions import (build_rpn_targets, compose_image_meta, detection_layer,  # noqa: F401
                              detection_target_layer, load_image_gt, log, mold_image, parse_image_meta,
                              proposal_layer, refine_detections_batched)
from .modal.deeplabv2 import DeepLabV2_ResNet101_MSC
from .modal.modals import (FPN, RPN, Classifier, Mask, ResNet, pyramid_roi_align_image)

# the GLM's resize / max / softmax / argmax / concat tail as one HIP pass (ops.msc_softmax_tail); "0": A/B switch
FUSED_GLM_TAIL = os.environ.get("SLN_FUSED_GLM_TAIL", "1") != "0"
# the crops' gradient w.r.t. P2..P5 added inside the RPN conv's data gradient (rpn_forward); "0": A/B switch
FUSE_CROP_GRADS = os.environ.get("SLN_FUSE_CROP_GRADS", "1") != "0"

# the frozen GLM (52 % of the forward FLOPs, independent of the detector until the mask head crops its output) on a
# second stream next to the backbone / FPN / RPN forward: both are sequences of whole-CU convolution launches, so
# the gain is the other branch's tail rounds filled and a memory-bound launch running next to a matrix-bound one.
# "1" enables it (measured in DESIGN.md section 13); off by default: every kernel's own duration stretches when it
# shares the chip, which the per-kernel roofline of bench.py would read as a slower kernel.
GLM_STREAM = os.environ.get("SLN_GLM_STREAM", "0") == "1"
# "2": the other way round -- the GLM stays on the main stream, BEHIND the backbone / FPN / RPN forward, and the
# proposal front end (top-k, decode, NMS) + target generation + the dead-end image crop -- ~150 small, latency-bound
# launches that leave the chip idle -- run on a second stream beside it.  The convolution launches then share the
# chip with kernels that occupy a few CUs for a few microseconds: their own durations hardly move.
TARGETS_STREAM = os.environ.get("SLN_GLM_STREAM", "0") == "2"

LAYER_REGEX = {
    "new": r"(fpn.C1.*)|(classifier.*)|(mask.*)|(layer_decoder.*)|(rpn.*)",
    "rpn": r"(fpn.C3.*)|(fpn.C4.*)|(fpn.C5.*)|(fpn.P5\_.*)|(fpn.P4\_.*)|(fpn.P3\_.*)|(fpn.P2\_.*)|(rpn.*)",
    "heads": r"(fpn.P5\_.*)|(fpn.P4\_.*)|(fpn.P3\_.*)|(fpn.P2\_.*)|(rpn.*)|(classifier.*)|(mask.*)|(layer_decoder.*)",
    "3+": r"(fpn.C3.*)|(fpn.C4.*)|(fpn.C5.*)|(fpn.P5\_.*)|(fpn.P4\_.*)|(fpn.P3\_.*)|(fpn.P2\_.*)|(rpn.*)|(classifier.*)|(mask.*)|(layer_decoder.*)|(amodal_refine.*)",
    "4+": r"(fpn.C4.*)|(fpn.C5.*)|(fpn.P5\_.*)|(fpn.P4\_.*)|(fpn.P3\_.*)|(fpn.P2\_.*)|(rpn.*)|(classifier.*)|(mask.*)|(layer_decoder.*)|(amodal_refine.*)",
    "5+": r"(fpn.C5.*)|(fpn.P5\_.*)|(fpn.P4\_.*)|(fpn.P3\_.*)|(fpn.P2\_.*)|(rpn.*)|(classifier.*)|(mask.*)|(layer_decoder.*)|(amodal_refine.*)",
    "layer": r"(mask.*)|(layer_decoder.*)",
    "all": ".*",
}


class Dataset(torch.utils.data.Dataset):
    """model.Dataset (model.py:30-119): one training sample per index, on the host, as the reference's
    8-tuple (images [3,H,W] f32 mean-subtracted, image_metas, rpn_match [A,1] i32, rpn_bbox [T,4] f32,
    gt_class_ids [N] i32, gt_boxes [N,4] f32 pixels, gt_layer [L,N,H,W] u8, image_raw [3,H,W] in [0,1]).
    `dataset` provides load_image / load_layer2 / image_ids (amodal_train.AmodalDataset).  The batched,
    device-resident equivalent the train loop consumes is AmodalDataset's own iterator; this class is
    the drop-in surface (and the host-side statement of the same semantics)."""

    def __init__(self, dataset, config, augment=True):
        self.image_ids = np.copy(dataset.image_ids)
        self.dataset, self.config, self.augment = dataset, config, augment
        self.anchors = utils.generate_pyramid_anchors(config.RPN_ANCHOR_SCALES, config.RPN_ANCHOR_RATIOS,
                                                      config.BACKBONE_SHAPES, config.BACKBONE_STRIDES,
                                                      config.RPN_ANCHOR_STRIDE)

    def __getitem__(self, image_index, draws=None):
        image_id = self.image_ids[image_index]
        image, image_metas, gt_class_ids, gt_boxes, gt_layer = load_image_gt(
            self.dataset, self.config, image_id, augment=self.augment,
            use_mini_mask=self.config.USE_MINI_MASK, draws=draws)
        if not np.any(gt_class_ids > 0):
            return None
        # RPN targets: float64 anchor matching (Functions.py:739-847); np.random.choice drops the surplus
        # positives / negatives -> uniform priorities here (or the replayed keep-priorities of `draws`)
        pr = None if draws is None or "rpn_priority" not in draws else torch.as_tensor(draws["rpn_priority"])[None]
        match, bbox = build_rpn_targets(image.shape, torch.from_numpy(self.anchors),
                                        torch.from_numpy(gt_class_ids)[None], torch.from_numpy(gt_boxes)[None].float(),
                                        self.config, priority=pr)
        if gt_boxes.shape[0] > self.config.MAX_GT_INSTANCES:
            ids = np.random.choice(np.arange(gt_boxes.shape[0]), self.config.MAX_GT_INSTANCES, replace=False)
            gt_class_ids, gt_boxes, gt_layer = gt_class_ids[ids], gt_boxes[ids], gt_layer[:, :, ids]
        images = mold_image(image.astype(np.float32), self.config)
        image_raw = torch.from_numpy(image.copy().transpose(2, 0, 1) / 255)
        return (torch.from_numpy(images.transpose(2, 0, 1)).float(), torch.from_numpy(image_metas),
                match[0].unsqueeze(1), bbox[0], torch.from_numpy(gt_class_ids), torch.from_numpy(gt_boxes).float(),
                torch.from_numpy(gt_layer.transpose(3, 2, 0, 1)), image_raw)

    def __len__(self):
        return self.image_ids.shape[0]


class MaskRCNN(nn.Module):
    def __init__(self, config, model_dir):
        super(MaskRCNN, self).__init__()
        self.config = config
        self.model_dir = model_dir
        self.set_log_dir()
        self.build(config=config)
        self.initialize_weights()
        self.loss_history = []
        self.val_loss_history = []
        self.current_epoch = 0
        self.layer_decoder = None
        self.amodal_refine = None
        self.GLM_modual = None
        # training: mask head on the first k roi slots only (None: all TRAIN_ROIS_PER_IMAGE slots, the reference's
        # graph).  positive_slots() = int(R * ROI_POSITIVE_RATIO), the most positives detection_target_layer returns:
        # with it the train step computes the same losses and gradients without the mask branch of the negative rois
        # (_training_heads; opt-in, bench.py reports it as a second number).
        self.mask_train_slots = None
        self.set_strict_layers(getattr(config, "STRICT_LAYERS", ""))

    def set_strict_layers(self, pattern):
        """Convolutions whose weight's name matches the regular expression `pattern` (fullmatch, like set_trainable)
        run in the STRICT operand format -- three bf16 parts, six MFMA products per multiply-add, >= fp32 per element
        with no exponent floor -- forward and backward; all others in conv_hip.PARTS (two scaled fp16 parts).
        Config.STRICT_LAYERS is the default ("" = none); what the choice buys and costs: tests/test_precision_gpu.py,
        tools/precision_control.py, DESIGN.md section 4.  Returns the number of tagged weights."""
        n = 0
        for name, p in self.named_parameters():
            strict = bool(pattern) and p.dim() == 4 and bool(re.fullmatch(pattern, name))
            p._sln_strict = strict
            n += strict
        if n:
            from . import conv_hip
            conv_hip.PARTS_FOR = conv_hip.parts_for_tagged
        self.strict_layers = pattern
        return n

    def positive_slots(self):
        return int(self.config.TRAIN_ROIS_PER_IMAGE * self.config.ROI_POSITIVE_RATIO)

    # ------------------------------------------------------------------ build
    def build(self, config):
        h, w = config.IMAGE_SHAPE[:2]
        if getattr(config, "STRICT_IMAGE_DIVISIBILITY", False) and (h % 64 or w % 64):
            raise Exception("Image size must be dividable by 2 at least 6 times "
                            "to avoid fractions when downscaling and upscaling.")
        resnet = ResNet(getattr(config, "ARCHITECTURE", "resnet101"), stage5=True)
        C1, C2, C3, C4, C5 = resnet.stages()
        self.fpn = FPN(C1, C2, C3, C4, C5, out_channels=256)
        anchors = utils.generate_pyramid_anchors(config.RPN_ANCHOR_SCALES, config.RPN_ANCHOR_RATIOS,
                                                 config.BACKBONE_SHAPES, config.BACKBONE_STRIDES,
                                                 config.RPN_ANCHOR_STRIDE)
        self.register_buffer("anchors", torch.from_numpy(anchors).float(), persistent=False)
        self.register_buffer("anchors_f64", torch.from_numpy(anchors), persistent=False)
        self.rpn = RPN(len(config.RPN_ANCHOR_RATIOS), config.RPN_ANCHOR_STRIDE, 256)
        self.classifier = Classifier(256, config.POOL_SIZE, config.IMAGE_SHAPE, config.NUM_CLASSES)
        self.mask = Mask(256, config.MASK_POOL_SIZE, config.IMAGE_SHAPE, config.NUM_CLASSES)
        if config.DATA_TYPE == "amodal":
            self.mask_vis = Mask(256, config.MASK_POOL_SIZE, config.IMAGE_SHAPE, config.NUM_CLASSES)
        self._freeze_batchnorm()

    def _freeze_batchnorm(self):
        # model.py:192-197: BN parameters that exist at build() never train
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                for p in m.parameters():
                    p.requires_grad = False

    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    def apply_amodal_heads(self, glm=True):
        """The head surgery amodal_train.py:606-614 performs after construction:
        two classes (background, layers), a 439-channel mask conv1 (256 roi +
        183 GLM channels), re-sized classifier linears, and the frozen GLM."""
        cfg = self.config
        cfg.NUM_CLASSES = 1 + 1
        self.mask.conv1 = nn.Conv2d(256 + cfg.GLM_CLASSES + 1, 256, kernel_size=3, stride=1)
        self.mask.conv5 = nn.Conv2d(256, cfg.NUM_CLASSES, kernel_size=1, stride=1)
        self.classifier.linear_class = nn.Linear(1024, cfg.NUM_CLASSES)
        self.classifier.linear_bbox = nn.Linear(1024, cfg.NUM_CLASSES * 4)
        self.mask.num_classes = self.classifier.num_classes = cfg.NUM_CLASSES
        if glm:
            self.GLM_modual = DeepLabV2_ResNet101_MSC(cfg.GLM_CLASSES)
        self.set_strict_layers(getattr(self, "strict_layers", ""))     # (the replaced layers are new Parameters)
        return self

    def set_trainable(self, layer_regex, model=None, indent=0, verbose=1, exclusive_off=True):
        """Parameters whose name does not match are frozen.  Like the reference
        (model.py:218-227) this only ever switches gradients OFF; pass
        exclusive_off=False to also switch matching (non-BN, non-GLM) ones on."""
        bn_names = {n + "." + pn for n, m in self.named_modules()
                    if isinstance(m, nn.BatchNorm2d) for pn, _ in m.named_parameters(recurse=False)}
        for name, p in self.named_parameters():
            ok = bool(re.fullmatch(layer_regex, name))
            if not ok:
                p.requires_grad = False
            elif not exclusive_off and not name.startswith("GLM_modual") and name not in bn_names:
                p.requires_grad = True   # BatchNorm stays frozen (model.py:192-197)

    # ------------------------------------------------------------ bookkeeping
    def set_log_dir(self, model_path=None):
        self.epoch = 0
        if model_path:
            m = re.match(r".*/\w+(\d{4})(\d{2})(\d{2})/mask\_rcnn\_\w+(\d{4})\.pth", model_path)
            if m:
                self.epoch = int(m.group(4))  # the reference reads group(6) of 4 (model.py:246-249)
        self.log_dir = os.path.join(self.model_dir, "{}".format(str(self.config.NAME).lower()))
        self.checkpoint_path = os.path.join(
            self.log_dir, "mask_rcnn_{}_*epoch*.pth".format(str(self.config.NAME).lower())
        ).replace("*epoch*", "{:04d}")

    def find_last(self):
        dir_names = next(os.walk(self.model_dir))[1]
        key = str(self.config.NAME).lower()
        dir_names = sorted(d for d in dir_names if d.startswith(key))
        if not dir_names:
            return None, None
        dir_name = os.path.join(self.model_dir, dir_names[-1])
        ckpts = sorted(f for f in next(os.walk(dir_name))[2] if f.startswith("mask_rcnn"))
        if not ckpts:
            return dir_name, None
        return dir_name, os.path.join(dir_name, ckpts[-1])

    def load_weights(self, filepath):
        if os.path.exists(filepath):
            self.load_state_dict(torch.load(filepath, map_location="cpu"), strict=False)
        else:
            print("Weight file not found ...")
        self.set_log_dir(filepath)
        os.makedirs(self.log_dir, exist_ok=True)

    # --------------------------------------------------------------- forward
    def _set_modes(self, mode):
        if mode == "inference":
            self.eval()
        else:
            self.train()
            for m in self.modules():  # BN always in eval mode (model.py:525-531)
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        if self.GLM_modual is not None:
            self.GLM_modual.eval()

    def glm_probs(self, molded_images):
        """Frozen global layer module (model.py:534-543): 513^2 bilinear resize ->
        DeepLab-v2 MSC -> softmax -> [probs | argmax/255] = [B,183,65,65]."""
        H, W = molded_images.shape[2], molded_images.shape[3]
        self.GLM_modual.eval()  # model.py:537
        with torch.no_grad():
            s = self.config.GLM_SIZE
            x = F.interpolate(molded_images, size=(s, s), mode="bilinear", align_corners=False)
            x = x.contiguous(memory_format=torch.channels_last)
            glm = self.GLM_modual
            if FUSED_GLM_TAIL and hasattr(glm, "softmax_tail") and not glm.training and glm._packable(x):
                probs, lab_small = glm.softmax_tail(x)
            else:
                logits = glm(x)
                probs = F.softmax(logits, dim=1)
                lab_small = torch.argmax(probs, dim=1)
                probs = torch.cat((probs, lab_small.unsqueeze(1).float() / 255), dim=1)
            gloable_lab = F.interpolate(lab_small.unsqueeze(1).float(), size=(H, W), mode="bilinear",
                                        align_corners=False)
        return probs.contiguous(memory_format=torch.channels_last), gloable_lab

    def rpn_forward(self, molded_images):
        maps = self.fpn(molded_images)
        # training on the HIP path: P2..P5 are read by the RPN and by the heads' crops; the crops' gradient is
        # added in the epilogue of the RPN conv's data gradient (conv_hip.GradInbox) -- valid when the RPN losses
        # take part in the backward pass, as in train_step; a pass without them fails loudly
        self._crop_inboxes = None
        if FUSE_CROP_GRADS and molded_images.is_cuda and torch.is_grad_enabled() and self.training and \
                all(m.requires_grad for m in maps[:4]):
            from . import conv_hip, nn_ops
            if nn_ops.BACKEND in ("auto", "hip"):
                self._crop_inboxes = [conv_hip.GradInbox() for _ in range(4)]
        boxes = (self._crop_inboxes or []) + [None] * len(maps)
        # (with the inbox the RPN conv can also prepare the FPN output conv's gradient: its soft chain, FPN.forward)
        outs = [self.rpn(p, boxes[i], getattr(p, "_sln_chain", None) if boxes[i] is not None else None)
                for i, p in enumerate(maps)]
        rpn_class_logits, rpn_class, rpn_bbox = [torch.cat(list(o), dim=1) for o in zip(*outs)]
        return maps, rpn_class_logits, rpn_class, rpn_bbox

    def predict(self, input, mode, priorities=None):
        """input = [molded_images [B,3,H,W], image_metas] (+ [gt_class_ids [B,N],
        gt_boxes [B,N,4] pixels, gt_layer] for training; gt_layer is either the
        uint64 labels [B,H,W] (as int64) or decoded planes [B,L,N,H,W] uint8)."""
        molded_images = input[0]
        image_metas = input[1]
        self._set_modes(mode)
        cfg = self.config
        if molded_images.is_cuda:
            from . import conv_hip       # delayed per-tensor operand scales follow the previous step's amax
            # (the MAX over the data-parallel ranks only in training steps, which every rank runs in
            # lockstep: a detect() / validation pass may run on one rank alone)
            conv_hip.update_scales(sync=(mode == "training" and torch.is_grad_enabled()))
        B, _, H, W = molded_images.shape
        if TARGETS_STREAM and molded_images.is_cuda and mode == "training" and \
                not (priorities and "rpn_rois" in priorities):
            return self._predict_training_forked(input, priorities)
        glm_side = None
        if GLM_STREAM and molded_images.is_cuda:
            main = torch.cuda.current_stream(molded_images.device)
            glm_side = getattr(self, "_glm_stream", None)
            if glm_side is None:
                glm_side = self._glm_stream = torch.cuda.Stream(device=molded_images.device)
            glm_side.wait_stream(main)               # (the operand scales' update and the images are complete)
            with torch.cuda.stream(glm_side):
                probs, gloable_lab = self.glm_probs(molded_images)
            molded_images.record_stream(glm_side)
        else:
            probs, gloable_lab = self.glm_probs(molded_images)
        maps, rpn_class_logits, rpn_class, rpn_bbox = self.rpn_forward(molded_images)
        mrcnn_feature_maps = maps[:4]
        count = cfg.POST_NMS_ROIS_TRAINING if mode == "training" else cfg.POST_NMS_ROIS_INFERENCE
        if priorities and "rpn_rois" in priorities:   # parity tests: externally fixed proposals
            rpn_rois, num_rois = priorities["rpn_rois"], priorities["num_rois"]
        else:
            rpn_rois, num_rois = proposal_layer([rpn_class, rpn_bbox], proposal_count=count,
                                                nms_threshold=cfg.RPN_NMS_THRESHOLD,
                                                anchors=self.anchors, config=cfg, return_counts=True)
        scale = utils.const_tensor([H, W, H, W], torch.float32, molded_images.device)
        if glm_side is not None:                     # first readers of the GLM's outputs follow
            main.wait_stream(glm_side)
            probs.record_stream(main)
            gloable_lab.record_stream(main)

        if mode == "inference":
            return self._predict_inference(rpn_rois, num_rois, mrcnn_feature_maps, probs,
                                           image_metas, scale)

        tgt, rois, roi_valid, box_ind, image_path = self._training_targets(
            rpn_rois, num_rois, input, scale, molded_images, priorities)
        return self._training_heads(molded_images, probs, gloable_lab, maps, rpn_class_logits, rpn_bbox, rpn_rois,
                                    num_rois, tgt, rois, roi_valid, box_ind, image_path)

    def _training_targets(self, rpn_rois, num_rois, input, scale, molded_images, priorities):
        """Detection targets of the sampled rois + the dead-end image crop (model.py:630-663)."""
        cfg = self.config
        B = molded_images.shape[0]
        gt_class_ids, gt_boxes, gt_layer = input[2], input[3], input[4]
        gt_boxes = gt_boxes / scale
        labels = gt_layer if gt_layer.dim() == 3 else None
        pr = priorities or {}
        tgt = detection_target_layer(rpn_rois, gt_class_ids, gt_boxes,
                                     None if labels is not None else gt_layer, cfg,
                                     num_proposals=num_rois, labels=labels,
                                     priority_pos=pr.get("pos"), priority_neg=pr.get("neg"),
                                     replay=pr.get("replay"))
        rois, roi_valid = tgt["rois"], tgt["roi_valid"]
        R = rois.shape[1]
        box_ind = torch.arange(B, dtype=torch.int32, device=rois.device).repeat_interleave(R)
        box_ind = torch.where(roi_valid.reshape(-1), box_ind, torch.full_like(box_ind, -1))
        # dead-end crop the reference computes for its (absent) refine net (model.py:651-663)
        image_path = pyramid_roi_align_image([rois, molded_images.contiguous()], 32, cfg.IMAGE_SHAPE,
                                             istrain=True, box_ind=box_ind).detach() / 140.0
        return tgt, rois, roi_valid, box_ind, image_path

    def _training_heads(self, molded_images, probs, gloable_lab, maps, rpn_class_logits, rpn_bbox, rpn_rois, num_rois,
                        tgt, rois, roi_valid, box_ind, image_path):
        """GLM crop, classifier and mask heads on the sampled rois -> the training outputs (model.py:664-700)."""
        cfg = self.config
        mrcnn_feature_maps = maps[:4]
        B, R = rois.shape[0], rois.shape[1]
        k = self.mask_train_slots
        k = k if (k is not None and 0 < k < R) else None
        # cropped into the head of the mask head's 439-channel input buffer (no torch.cat later)
        GLM_feature = None
        if k is None:
            GLM_feature = pyramid_roi_align_image([rois, probs], cfg.MASK_POOL_SIZE, (65, 65), istrain=True,
                                                  box_ind=box_ind, cat_extra=256)
            if getattr(GLM_feature, "_sln_cat_buf", None) is None:
                GLM_feature = GLM_feature.detach()
        # both heads' crops are differentiated by the step loss: they share one set of P2..P5 gradient maps
        pool = None
        if rois.is_cuda and torch.is_grad_enabled() and all(
                m.requires_grad and m.is_contiguous(memory_format=torch.channels_last) and not m.is_contiguous()
                for m in mrcnn_feature_maps):
            from .modal.modals import CropGradPool
            pool = CropGradPool()
            pool.inboxes = getattr(self, "_crop_inboxes", None)
        mrcnn_class_logits, mrcnn_class, mrcnn_bbox = self.classifier(mrcnn_feature_maps, rois, box_ind,
                                                                      grad_pool=pool)
        if k is not None:
            # OPT-IN (off by default: the reference runs its mask branch on all R sampled rois, model.py:664-700):
            # the two mask losses read the POSITIVE rois only (loss.py:113-152) and detection_target_layer puts the
            # positives first, at most int(R * ROI_POSITIVE_RATIO) of them -- the mask branch of the other slots
            # feeds nothing in a train step.  The mask head then runs on the first k slots; the rows behind them
            # are zeros (no loss reads them, no gradient flows from them): same losses, same gradients.
            rois_m = rois[:, :k].contiguous()
            ind_m = box_ind.view(B, R)[:, :k].reshape(-1).contiguous()
            glm_m = pyramid_roi_align_image([rois_m, probs], cfg.MASK_POOL_SIZE, (65, 65), istrain=True,
                                            box_ind=ind_m, cat_extra=256)
            if getattr(glm_m, "_sln_cat_buf", None) is None:
                glm_m = glm_m.detach()
            mk, _feat = self.mask(mrcnn_feature_maps, rois_m, glm_m, ind_m, grad_pool=pool)
            mrcnn_mask = torch.nn.functional.pad(mk.reshape(B, k, *mk.shape[1:]), (0, 0, 0, 0, 0, 0, 0, R - k)) \
                .reshape(B * R, *mk.shape[1:])
        else:
            mrcnn_mask, _feat = self.mask(mrcnn_feature_maps, rois, GLM_feature, box_ind, grad_pool=pool)
        nc = mrcnn_class_logits.shape[1]
        return {
            "rpn_class_logits": rpn_class_logits, "rpn_bbox": rpn_bbox,
            "target_class_ids": tgt["class_ids"], "target_deltas": tgt["deltas"],
            "target_mask": tgt["masks"], "roi_valid": roi_valid, "rois": rois,
            "mrcnn_class_logits": mrcnn_class_logits.view(B, R, nc),
            "mrcnn_bbox": mrcnn_bbox.view(B, R, nc, 4),
            "mrcnn_mask": mrcnn_mask.reshape(B, R, mrcnn_mask.shape[1], mrcnn_mask.shape[2],
                                             mrcnn_mask.shape[3]),
            "image_path": image_path, "gloable_lab": gloable_lab, "num_rois": num_rois,
            "rpn_rois": rpn_rois,
        }

    def _predict_training_forked(self, input, priorities):
        """predict(mode='training') with the proposal front end and the target generation on a second stream beside
        the GLM's forward (SLN_GLM_STREAM=2, see TARGETS_STREAM)."""
        cfg = self.config
        molded_images = input[0]
        B, _, H, W = molded_images.shape
        main = torch.cuda.current_stream(molded_images.device)
        side = getattr(self, "_glm_stream", None)
        if side is None:
            side = self._glm_stream = torch.cuda.Stream(device=molded_images.device)
        maps, rpn_class_logits, rpn_class, rpn_bbox = self.rpn_forward(molded_images)
        side.wait_stream(main)                       # the RPN outputs are complete
        with torch.cuda.stream(side):
            rpn_rois, num_rois = proposal_layer([rpn_class, rpn_bbox], proposal_count=cfg.POST_NMS_ROIS_TRAINING,
                                                nms_threshold=cfg.RPN_NMS_THRESHOLD,
                                                anchors=self.anchors, config=cfg, return_counts=True)
            scale = utils.const_tensor([H, W, H, W], torch.float32, molded_images.device)
            tgt, rois, roi_valid, box_ind, image_path = self._training_targets(
                rpn_rois, num_rois, input, scale, molded_images, priorities)
        for t_ in (rpn_class, rpn_bbox, molded_images):
            t_.record_stream(side)
        probs, gloable_lab = self.glm_probs(molded_images)
        main.wait_stream(side)
        for t_ in [rpn_rois, num_rois, rois, roi_valid, box_ind, image_path] + \
                [v for v in tgt.values() if torch.is_tensor(v)]:
            t_.record_stream(main)                   # allocated on the side stream, read on the main one from here on
        return self._training_heads(molded_images, probs, gloable_lab, maps, rpn_class_logits, rpn_bbox, rpn_rois,
                                    num_rois, tgt, rois, roi_valid, box_ind, image_path)

    def _predict_inference(self, rpn_rois, num_rois, maps, probs, image_metas, scale):
        """Inference tail (model.py:576-628) for B images at once, fixed capacity, no device -> host copy:
        every proposal slot goes through the classifier (slots beyond an image's count carry box_ind = -1:
        zero crops, excluded from the detections), refine_detections_batched keeps the reference's top 100
        per image, the mask head runs on [B, 100] detection boxes.  Includes the reference's quirk of
        cropping the GLM map with PIXEL-coordinate boxes (model.py:588-594) -- kept for
        bug-compatibility of `evaluate`.  Returns [detections [B,100,6], mrcnn_mask [B,100,C,32,32]]; rows
        behind an image's count (`self.last_num_detections` [B], device) are zero.  USE_NMS = True (not the
        reference's default) takes the per-image path of refine_detections with its host syncs."""
        cfg = self.config
        B, R = rpn_rois.shape[0], rpn_rois.shape[1]
        dev = rpn_rois.device
        valid = torch.arange(R, device=dev)[None, :] < num_rois[:, None].to(torch.int64)
        box_ind = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(R)
        box_ind = torch.where(valid.reshape(-1), box_ind, torch.full_like(box_ind, -1))
        _, mrcnn_class, mrcnn_bbox = self.classifier(maps, rpn_rois, box_ind)
        nc = mrcnn_class.shape[1]
        _, _, window, _ = parse_image_meta(np.asarray(image_metas))
        win = np.asarray(window, dtype=np.float32).reshape(-1, 4)      # one window per image
        if win.shape[0] == 1 and B > 1:
            win = np.repeat(win, B, axis=0)
        if cfg.USE_NMS:
            dets = []
            for b in range(B):
                n = int(num_rois[b])
                d, _ = detection_layer(cfg, rpn_rois[b:b + 1, :n], mrcnn_class.view(B, R, nc)[b, :n],
                                       mrcnn_bbox.view(B, R, nc, 4)[b, :n],
                                       np.asarray(image_metas)[min(b, len(image_metas) - 1):][:1])
                row = rpn_rois.new_zeros(R, 6)          # (per-class NMS keeps what survives: no cap)
                if len(d):
                    row[:d.shape[0]] = d
                dets.append(row)
            detections = torch.stack(dets)
            live = detections[:, :, 4] > 0
            count = live.sum(dim=1).to(torch.int32)
        else:
            detections, live, count = refine_detections_batched(
                rpn_rois, valid, mrcnn_class.view(B, R, nc), mrcnn_bbox.view(B, R, nc, 4), win, cfg,
                max_instances=100)             # (Functions.py:526-532 hard-codes the top 100)
        # model.py:590-591 clamps the whole rows to [0, 1024] (class id and score included: harmless); the
        # bound is the image size here, so that IMAGE_MAX_DIM != 1024 keeps its meaning
        detections = detections.clamp(min=0.0, max=float(max(cfg.IMAGE_SHAPE[:2])))
        D = detections.shape[1]
        det_ind = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(D)
        det_ind = torch.where(live.reshape(-1), det_ind, torch.full_like(det_ind, -1))
        cls_feature = pyramid_roi_align_image([detections[:, :, :4].contiguous(), probs], cfg.MASK_POOL_SIZE,
                                              (65, 65), istrain=False, box_ind=det_ind).detach()
        detection_boxes = detections[:, :, :4] / scale
        mrcnn_mask, _ = self.mask(maps, detection_boxes, cls_feature, det_ind)
        mrcnn_mask = mrcnn_mask.contiguous()
        mrcnn_mask[:, 1] = torch.sigmoid(mrcnn_mask[:, 1:].sum(dim=1))
        mrcnn_mask = torch.where(live.reshape(-1, 1, 1, 1), mrcnn_mask, torch.zeros_like(mrcnn_mask))
        self.last_num_detections = count
        return [detections, mrcnn_mask.view(B, D, *mrcnn_mask.shape[1:])]

    # -------------------------------------------------------------- training
    def compute_losses(self, out, rpn_match, rpn_bbox):
        return L.total_loss(rpn_match, rpn_bbox, out["rpn_class_logits"], out["rpn_bbox"],
                            out["target_class_ids"], out["mrcnn_class_logits"],
                            out["target_deltas"], out["mrcnn_bbox"], out["target_mask"],
                            out["mrcnn_mask"], out["roi_valid"])

    def make_optimizer(self, learning_rate):
        """SGD, momentum, weight decay on everything but 'bn' parameters
        (model.py:352-358)."""
        wd = [p for n, p in self.named_parameters() if p.requires_grad and "bn" not in n]
        no_wd = [p for n, p in self.named_parameters() if p.requires_grad and "bn" in n]
        groups = [{"params": wd, "weight_decay": self.config.WEIGHT_DECAY}, {"params": no_wd}]
        if self.anchors.is_cuda and nn_ops.BACKEND != "torch":
            from .optim import ClippedSGD         # clip + SGD fused over device-side tables
            return ClippedSGD(groups, lr=learning_rate, momentum=self.config.LEARNING_MOMENTUM)
        return torch.optim.SGD(groups, lr=learning_rate, momentum=self.config.LEARNING_MOMENTUM)

    def train_step(self, batch, optimizer, grad_sync=None, priorities=None):
        """One optimisation step on a batch: predict -> six losses -> backward ->
        (data-parallel gradient all-reduce) -> global-norm clip 5.0 -> SGD
        (model.py:415-444).  `batch` = dict(images, image_metas, rpn_match, rpn_bbox,
        gt_class_ids, gt_boxes, gt_layer).  Returns the loss tensor (no host sync).
        A step in which an fp16 operand block had to clamp is NOT applied (conv_hip.clamp_mark / clamp_veto, round 6):
        the optimiser skips it on the device and counts it; under data parallelism every rank skips the same step."""
        hip = None
        if batch["images"].is_cuda and nn_ops.BACKEND != "torch":
            from . import conv_hip as hip
            hip.clamp_mark()
        out = self.predict([batch["images"], batch.get("image_metas"), batch["gt_class_ids"],
                            batch["gt_boxes"], batch["gt_layer"]], mode="training",
                           priorities=priorities)
        loss, parts = self.compute_losses(out, batch["rpn_match"], batch["rpn_bbox"])
        optimizer.zero_grad(set_to_none=True)
        if getattr(self.config, "LOSS_REDUCTION", "mean") == "sum":   # the reference's gradient SUM over images
            (loss * batch["images"].shape[0]).backward()
        else:
            loss.backward()
        if grad_sync is not None:
            grad_sync([p for p in self.parameters() if p.requires_grad])
        reducer = getattr(grad_sync, "__self__", grad_sync)       # (a bound finish / the reducer object itself)
        if getattr(reducer, "world", 1) > 1 and getattr(reducer, "veto", None) is not None:
            veto = reducer.veto                                    # summed over the ranks with the last bucket
        else:
            veto = hip.clamp_veto(batch["images"].device) if hip is not None else None
        self.optimizer_step(optimizer, veto)
        return loss.detach(), parts

    def optimizer_step(self, optimizer, veto=None):
        """Global-norm clip (5.0) over every gradient, then momentum SGD with weight decay on the
        non-'bn' parameters (model.py:441-444, 352-358).  Runs after the all-reduce: a parameter
        without a local gradient may have received one from a peer, and every rank must clip over
        the same set."""
        from .optim import ClippedSGD
        if isinstance(optimizer, ClippedSGD):
            self.last_grad_norm = optimizer.step(self.config.GRADIENT_CLIP_NORM, veto=veto)
            return
        params = [p for p in self.parameters() if p.requires_grad and p.grad is not None]
        self.last_grad_norm = torch.nn.utils.clip_grad_norm_(params, self.config.GRADIENT_CLIP_NORM)
        optimizer.step()

    def train_model(self, train_dataset, val_dataset, learning_rate, epochs, layers,
                    grad_sync=None):
        """Epoch loop of the reference (model.py:304-368).  `train_dataset` yields
        batch dicts (see train_step)."""
        if layers in LAYER_REGEX:
            layers = LAYER_REGEX[layers]
        import torch.distributed as dist
        ddp = dist.is_available() and dist.is_initialized()
        rank0 = (not ddp) or dist.get_rank() == 0
        steps = max(1, self.config.STEPS_PER_EPOCH // (dist.get_world_size() if ddp else 1))
        if rank0:
            log("\nStarting at epoch {}. LR={}\n".format(self.epoch + 1, learning_rate))
            log("Checkpoint Path: {}".format(self.checkpoint_path))
        self.set_trainable(layers)
        optimizer = self.make_optimizer(learning_rate)
        for epoch in range(epochs):
            if rank0:
                log("Epoch {}/{}.".format(epoch, epochs))
            self.current_epoch += 1
            mean = self.train_epoch(train_dataset, optimizer, steps, grad_sync)
            if rank0:       # (the reference prints the running loss in its progress bar, model.py:447-449)
                log("\t{}/{} Complete - mean loss: {:.5f}".format(steps, steps, mean))
            self.save_checkpoint(self.checkpoint_path.format(self.epoch))
            self.epoch += 1

    def save_checkpoint(self, path):
        """state_dict -> `path` (model.py:366 of the reference), written by rank 0 only, to a
        temporary file that is renamed into place (a reader -- `--model last` -- never sees a torn
        file); the other ranks wait, so nobody races ahead and reads it half written."""
        import torch.distributed as dist
        ddp = dist.is_available() and dist.is_initialized()
        if (not ddp) or dist.get_rank() == 0:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            tmp = path + ".tmp.%d" % os.getpid()
            torch.save(self.state_dict(), tmp)
            os.replace(tmp, path)
        if ddp:
            dist.barrier()

    def train_epoch(self, datagenerator, optimizer, steps, grad_sync=None):
        """`steps` optimiser steps (model.py:370-462).  No host sync inside the loop; the epoch's single
        sync reads, together with the mean loss, the two device-side health counters of the step:
          * fp16 x 2 operand blocks that had to clamp a value to +-65504 (conv_hip.saturation_count():
            the clamped tensor's scale follows its new maximum from the next step on -- the amax is taken
            before the clamp -- so the count says how many blocks of how many steps ran under-estimated);
          * optimiser steps skipped on the device for a non-finite gradient norm (ClippedSGD.skipped_steps();
            a guard of this build -- the reference would apply the NaN: its `continue`s, model.py:416-418,
            433-434, skip batches without ground truth).  The loss of such a step is left out of the epoch
            mean instead of turning it into NaN, and the count is logged as an ERROR line.
        Both are logged and kept in `self.epoch_health`."""
        dev = self.anchors.device
        loss_sum = torch.zeros((), device=dev)
        finite_steps = torch.zeros((), device=dev)
        conv_hip = None
        if self.anchors.is_cuda and nn_ops.BACKEND != "torch":
            from . import conv_hip
        sat0 = conv_hip.saturation_count() if conv_hip is not None else 0      # (before the loop: not in it)
        skip0 = optimizer.skipped_steps() if hasattr(optimizer, "skipped_steps") else 0
        skipc0 = optimizer.skipped_clamped_steps() if hasattr(optimizer, "skipped_clamped_steps") else 0
        step = 0
        for batch in datagenerator:
            loss, _ = self.train_step(batch, optimizer, grad_sync)
            ok = torch.isfinite(loss)
            loss_sum += torch.where(ok, loss, torch.zeros_like(loss))
            finite_steps += ok.to(loss_sum.dtype)
            step += 1
            if step == steps:
                break
        mean = float(loss_sum / finite_steps.clamp(min=1))   # the one host sync of the epoch
        clamped_skips = (optimizer.skipped_clamped_steps() - skipc0) if hasattr(optimizer, "skipped_clamped_steps") else 0
        blocks = (conv_hip.saturation_count() - sat0) if conv_hip is not None else 0
        guarded = conv_hip is not None and conv_hip.SKIP_CLAMPED_STEPS and hasattr(optimizer, "skipped_clamped_steps")
        health = {"steps": step, "non_finite_losses": step - int(finite_steps),
                  "conv_saturated_blocks": blocks,
                  # every clamped block of a guarded step was vetoed with its step: nothing of it reached the weights
                  "clamped_and_skipped_steps": clamped_skips,
                  "clamped_and_applied_blocks": 0 if guarded else blocks,
                  "skipped_optimizer_steps": ((optimizer.skipped_steps() - skip0)
                                              if hasattr(optimizer, "skipped_steps") else 0) - clamped_skips}
        if conv_hip is not None:
            conv_hip.check_ranks()
        self.epoch_health = health
        if health["skipped_optimizer_steps"]:
            log("ERROR: {} of {} optimiser steps were skipped for a non-finite gradient norm".format(
                health["skipped_optimizer_steps"], step))
        if health["clamped_and_skipped_steps"]:
            log("{} of {} optimiser steps were not applied: an fp16 operand block clamped in them ({} blocks); their "
                "scales follow from the next step on".format(health["clamped_and_skipped_steps"], step, blocks))
        if health["non_finite_losses"] or health["conv_saturated_blocks"] or health["skipped_optimizer_steps"]:
            log("epoch health: {}".format(health))
        return mean

    # -------------------------------------------------------------- inference
    def mold_inputs(self, images):
        """Resize to IMAGE_MAX_DIM^2 (the reference squashes, utils.py:351-356),
        subtract the mean pixel, build image metas (model.py:709-745)."""
        cfg = self.config
        molded, metas, windows = [], [], []
        from PIL import Image
        for image in images:
            # utils.py:351-356: scipy.misc.imresize(image, (max_dim, max_dim)) = PIL bilinear resize of
            # the uint8 image (host-side preprocessing, as in the reference)
            u8 = np.ascontiguousarray(image).astype(np.uint8)
            if u8.shape[:2] != (cfg.IMAGE_MAX_DIM, cfg.IMAGE_MAX_DIM):
                u8 = np.asarray(Image.fromarray(u8).resize((cfg.IMAGE_MAX_DIM, cfg.IMAGE_MAX_DIM),
                                                           Image.BILINEAR))
            m = u8.astype(np.float32) - cfg.MEAN_PIXEL
            window = (0, 0, cfg.IMAGE_MAX_DIM, cfg.IMAGE_MAX_DIM)
            molded.append(m.astype(np.float32))
            windows.append(window)
            metas.append(compose_image_meta(0, image.shape, window,
                                            np.zeros([cfg.NUM_CLASSES], dtype=np.int32)))
        return np.stack(molded), np.stack(metas), np.stack(windows)

    def detect(self, images, mode="inference", priorities=None, keep_device=False, batch_size=None):
        """List of HxWx3 images -> list of dicts(rois, class_ids, scores, masks)
        (model.py:464-514).  The reference runs one image per predict(); here `batch_size` images (default:
        all of them, config.BATCH_SIZE at most) go through ONE batched predict(mode='inference') and only
        the per-image hand-off to the host (unmold) is a loop.  priorities: predict() overrides, either one
        dict for the whole call ({"rpn_rois" [B,1000,4], "num_rois" [B]}) or a per-image list of
        batch-1 dicts (parity tests feed the reference's proposals).  On the GPU the tail (box transform,
        zero-area filter, mask resize + threshold + paste) runs on the device; keep_device=True
        additionally returns the masks as the device tensor "masks_device" [N,W,H] (column-major per mask,
        ready for mask_rle.encode) and skips the [H,W,N] host copy ("masks" is None).  Images without a
        detection are left out of the result list, like the reference's `continue`; "image_index" says which
        input a result belongs to."""
        results = []
        if not len(images):
            return results
        step = int(batch_size or min(len(images), max(1, int(getattr(self.config, "BATCH_SIZE", 1)))))
        with torch.no_grad():
            for i0 in range(0, len(images), step):
                chunk = images[i0:i0 + step]
                molded, metas, windows = self.mold_inputs(chunk)
                x = torch.from_numpy(molded.transpose(0, 3, 1, 2)).float().to(self.anchors.device)
                pr = None
                if isinstance(priorities, dict):
                    pr = {k: v[i0:i0 + step] for k, v in priorities.items()}
                elif priorities:
                    ps = priorities[i0:i0 + step]
                    pr = {k: torch.cat([p[k] for p in ps]) for k in ps[0]}
                detections, mrcnn_mask = self.predict([x, metas], mode=mode, priorities=pr)
                counts = self.last_num_detections.cpu().numpy()       # the hand-off to the host starts here
                for b, image in enumerate(chunk):
                    n = int(counts[b])
                    if n == 0:
                        continue
                    if detections.is_cuda:
                        results.append(dict(self.unmold_detections_device(detections[b, :n], mrcnn_mask[b, :n],
                                                                          image.shape, windows[b], keep_device),
                                            image_index=i0 + b))
                        continue
                    det = detections[b, :n].cpu().numpy()
                    msk = mrcnn_mask[b, :n].permute(0, 2, 3, 1).cpu().numpy()
                    rois, class_ids, scores, masks = self.unmold_detections(det, msk, image.shape, windows[b])
                    results.append({"rois": rois, "class_ids": class_ids, "scores": scores, "masks": masks,
                                    "image_index": i0 + b})
        return results

    def detect_submit(self, images, tail, keys=None, priorities=None):
        """detect() without the hand-off on this thread: mold, ONE batched predict(mode='inference'), then the outputs
        go to `tail` (tail.InferenceTail: unmold + RLE of the whole batch on a side stream, a worker thread) and the
        call returns -- the next batch's forward can be enqueued at once.  tail.results() -> {key: {rois, class_ids,
        scores, rles}} (model.py:464-514 + amodal_train.py:370-400, batched)."""
        molded, metas, windows = self.mold_inputs(images)
        x = torch.from_numpy(molded.transpose(0, 3, 1, 2)).float().to(self.anchors.device)
        with torch.no_grad():
            detections, mrcnn_mask = self.predict([x, metas], mode="inference", priorities=priorities)
        tail.submit(detections, mrcnn_mask, self.last_num_detections, [im.shape for im in images], windows, keys)

    def unmold_detections_device(self, detections, mrcnn_mask, image_shape, window, keep_device=False):
        """unmold_detections (model.py:747-806) without leaving the GPU: detections [M,6] and
        mrcnn_mask [M,C,h,w] device tensors -> dict like detect()'s.  The box arithmetic is the
        reference's numpy float64 (boxes - shifts) * scales truncated to int32; the masks come from
        ops.unmold_masks (bit-identical to utils.unmold_mask)."""
        from . import ops
        dev = detections.device
        cls_f = detections[:, 4]
        zero = torch.nonzero(cls_f == 0)
        N = int(zero[0, 0]) if zero.numel() else detections.shape[0]
        class_ids = cls_f[:N].to(torch.int32)
        class_ids = torch.where(class_ids > 0, torch.ones_like(class_ids), class_ids)
        scores = detections[:N, 5]
        h_scale = image_shape[0] / (window[2] - window[0])
        w_scale = image_shape[1] / (window[3] - window[1])
        scales = torch.tensor([h_scale, w_scale, h_scale, w_scale], dtype=torch.float64, device=dev)
        shifts = torch.tensor([window[0], window[1], window[0], window[1]], dtype=torch.float64, device=dev)
        boxes = ((detections[:N, :4].double() - shifts) * scales).to(torch.int32)
        ok = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) > 0
        boxes, class_ids, scores = boxes[ok], class_ids[ok], scores[ok]
        planes = mrcnn_mask[:N][ok].float()
        H, W = int(image_shape[0]), int(image_shape[1])
        full = ops.unmold_masks(planes, class_ids, boxes, H, W)            # [n, W, H]
        out = {"rois": boxes.cpu().numpy(), "class_ids": class_ids.cpu().numpy(),
               "scores": scores.cpu().numpy()}
        if keep_device:
            out["masks"] = None
            out["masks_device"] = full
        elif full.shape[0]:
            out["masks"] = full.permute(2, 1, 0).cpu().numpy()             # [H, W, n] view, F-ordered planes
        else:
            out["masks"] = np.empty((0,) + tuple(mrcnn_mask.shape[2:4]))
        return out

    def unmold_detections(self, detections, mrcnn_mask, image_shape, window):
        """Network outputs -> image-space boxes and full-size binary masks
        (model.py:747-806, utils.py:447-465)."""
        zero_ix = np.where(detections[:, 4] == 0)[0]
        N = zero_ix[0] if zero_ix.shape[0] > 0 else detections.shape[0]
        boxes = detections[:N, :4]
        class_ids = detections[:N, 4].astype(np.int32)
        class_ids[class_ids > 0] = 1
        scores = detections[:N, 5]
        masks = mrcnn_mask[np.arange(N), :, :, class_ids]
        h_scale = image_shape[0] / (window[2] - window[0])
        w_scale = image_shape[1] / (window[3] - window[1])
        scales = np.array([h_scale, w_scale, h_scale, w_scale])
        shifts = np.array([window[0], window[1], window[0], window[1]])
        boxes = np.multiply(boxes - shifts, scales).astype(np.int32)
        ok = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) > 0
        boxes, class_ids, scores, masks = boxes[ok], class_ids[ok], scores[ok], masks[ok]
        full = []
        for i in range(boxes.shape[0]):
            y1, x1, y2, x2 = boxes[i]
            # utils.py:447-465 -> scipy.misc.imresize(mask, (h, w), interp='bilinear'):
            # min-max bytescale to uint8 (float32 arithmetic on the float32 head output), PIL
            # bilinear resize, then /255 >= 0.5.  CPU-resident models only; cuda goes through
            # unmold_detections_device.
            from PIL import Image
            m = masks[i].astype(np.float32)
            lo, hi = m.min(), m.max()
            cscale = np.float32(hi - lo) if hi != lo else np.float32(1.0)
            byt = ((m - lo) * np.float32(255.0 / np.float64(cscale))).clip(0, 255)
            byt = (byt + np.float32(0.5)).astype(np.uint8)
            r = np.asarray(Image.fromarray(byt).resize((int(x2 - x1), int(y2 - y1)), Image.BILINEAR),
                           dtype=np.float32)
            fm = np.zeros(image_shape[:2], dtype=np.uint8)
            fm[y1:y2, x1:x2] = (r / 255.0 >= 0.5).astype(np.uint8)
            full.append(fm)
        full = np.stack(full, axis=-1) if full else np.empty((0,) + masks.shape[1:3])
        return boxes, class_ids, scores, full
