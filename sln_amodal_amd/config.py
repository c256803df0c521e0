"""Configuration object with the reference's attribute names (config.py:19-188)
so model/dataset code written against it keeps working.  Values are the
reference defaults; sub-class and override class attributes, derived fields are
filled in __init__.  MI355X-specific knobs are grouped at the end."""
import math
import os

import numpy as np


class Config(object):
    NAME = None
    LIMIT_IMAGES = -1

    GPU_COUNT = 1
    IMAGES_PER_GPU = 1
    # Reference: batch 1 with gradient accumulation over BATCH_SIZE iterations
    # (config.py:39-40, model.py:440-444).  Here it is the per-GPU image batch of one step.
    BATCH_SIZE = 1
    # OPTIMIZER steps per epoch.  The reference multiplies this by BATCH_SIZE in __init__
    # (config.py:184) because ITS step is one image and its optimizer steps every BATCH_SIZE
    # iterations: 2500 updates of BATCH_SIZE images per epoch.  One train_step here already
    # consumes BATCH_SIZE images and updates once, so the number stays 2500 (data-parallel
    # runs divide it by the world size: amodal_train.main).
    STEPS_PER_EPOCH = 2500
    # Step loss over the B images of a batch: "mean" (default) or "sum".  The reference SUMS 16
    # per-image gradients (and clips the running sum to 5.0 after every image, model.py:441);
    # a batched step can only clip once, so "sum" reproduces the reference's step size only
    # while the clip is inactive.  With "mean" the same LEARNING_RATE takes 1/B-size steps.
    LOSS_REDUCTION = "mean"
    VALIDATION_STEPS = 100

    ARCHITECTURE = "resnet101"            # reference hard-codes this (model.py:163)
    BACKBONE_STRIDES = [4, 8, 16, 32, 64]
    NUM_CLASSES = 81

    RPN_ANCHOR_SCALES = (32, 64, 128, 256, 512)
    RPN_ANCHOR_RATIOS = [0.5, 1, 2]
    RPN_ANCHOR_STRIDE = 1
    RPN_NMS_THRESHOLD = 0.7
    USE_NMS = False
    RPN_TRAIN_ANCHORS_PER_IMAGE = 256
    MAX_NUMB_RPNS = 500
    PRE_NMS_LIMIT = 6000                  # hard-coded in the reference (Functions.py:144)
    POST_NMS_ROIS_TRAINING = 1000
    POST_NMS_ROIS_INFERENCE = 1000

    USE_MINI_MASK = False
    MINI_MASK_SHAPE = (56, 56)

    IMAGE_MIN_DIM = 800
    IMAGE_MAX_DIM = 1024
    IMAGE_PADDING = True
    MEAN_PIXEL = np.array([123.7, 116.8, 103.9])

    TRAIN_ROIS_PER_IMAGE = 100
    ROI_POSITIVE_RATIO = 0.7
    POOL_SIZE = 7
    MASK_POOL_SIZE = 16
    MASK_SHAPE = [32, 32]
    MAX_GT_INSTANCES = 50

    RPN_BBOX_STD_DEV = np.array([0.1, 0.1, 0.2, 0.2])
    BBOX_STD_DEV = np.array([0.1, 0.1, 0.2, 0.2])

    DETECTION_MAX_INSTANCES = 1000
    DETECTION_MIN_CONFIDENCE = 0.7
    DETECTION_NMS_THRESHOLD = 0.3

    EXPERIMENT_DIR = "log/rcnn-train"
    DATA_TYPE = "coco"

    LEARNING_RATE = 0.001
    LEARNING_MOMENTUM = 0.9
    WEIGHT_DECAY = 0.0001
    GRADIENT_CLIP_NORM = 5.0              # model.py:441
    USE_RPN_ROIS = False
    USE_TENSORBOARDX = False

    # ---- MI355X-side knobs (no reference counterpart) ----
    GLM_CLASSES = 182                     # amodal_train.py:613
    GLM_SIZE = 513                        # model.py:535
    STRICT_IMAGE_DIVISIBILITY = False     # reference raises unless H,W % 64 == 0 (model.py:153-157)
    # convolutions (regular expression on the weight's name) that run in the strict 3 x bf16 operand format instead of
    # the default 2 x scaled fp16 (MaskRCNN.set_strict_layers)
    STRICT_LAYERS = os.environ.get("SLN_STRICT_LAYERS", "")

    def __init__(self):
        self.IMAGE_SHAPE = np.array([self.IMAGE_MAX_DIM, self.IMAGE_MAX_DIM, 3])
        self.BACKBONE_SHAPES = np.array(
            [[int(math.ceil(self.IMAGE_SHAPE[0] / s)), int(math.ceil(self.IMAGE_SHAPE[1] / s))]
             for s in self.BACKBONE_STRIDES])

    def display(self):
        print("\nConfigurations:")
        for a in dir(self):
            if not a.startswith("__") and not callable(getattr(self, a)):
                print("{:30} {}".format(a, getattr(self, a)))
        print("\n")
