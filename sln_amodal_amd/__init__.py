"""sln_amodal_amd -- MI355X-native hot path of SLN-Amodal.

Package layout mirrors the reference repository's import surface so the hot
path drops in:

    sln_amodal_amd.nms.nms_wrapper.nms                      <- nms/nms_wrapper.py
    sln_amodal_amd.roialign.roi_align.crop_and_resize.*     <- roialign/roi_align/crop_and_resize.py
    sln_amodal_amd.modal.{modals,Functions,loss,deeplabv2}  <- modal/*
    sln_amodal_amd.model.MaskRCNN                           <- model.py
    sln_amodal_amd.amodal_train                             <- amodal_train.py

All native work goes through csrc/libsln_amodal_hip.so (C ABI: include/sln_amodal.h).
There is no CPU fallback: ops raise if the library is missing.
"""
__version__ = "0.1.0"
