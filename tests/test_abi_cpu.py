"""CPU suite: the C-ABI library loads and exports every symbol include/sln_amodal.h
declares (no compute calls without a GPU), and the product path refuses to run
without it."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    names = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names += re.findall(r"\b(sln_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_library_builds_and_exports_header_symbols():
    from sln_amodal_amd.csrc import build
    so = build.build()
    lib = ctypes.CDLL(so)
    syms = _declared_symbols()
    assert len(syms) >= 10
    for name in syms:
        assert hasattr(lib, name), "missing export: " + name


def test_ctypes_signatures_cover_header():
    from sln_amodal_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    L = _lib.lib()
    assert L.sln_abi_version() >= 1
    assert L.sln_error_string(0) == b"ok"
    assert L.sln_nms_workspace_bytes(2, 6000) == 2 * 6000 * 94 * 8


def test_argument_validation_needs_no_gpu():
    from sln_amodal_amd import _lib
    L = _lib.lib()
    assert L.sln_nms_f32(None, -1, 0, None, 0.5, 0, None, None, None, 0, None) == 1
    assert L.sln_crop_and_resize_fwd_f32(None, 1, 1, 0, 1, 0, None, None, 1, 1, 1, 0.0, None,
                                         None, None) == 1
    assert L.sln_label_decode_u64(None, 1, 4, 4, 0, 1, None, None) == 1


def test_product_path_fails_loudly_without_gpu_tensors():
    import torch
    from sln_amodal_amd import ops
    from sln_amodal_amd.nms.nms_wrapper import nms
    with pytest.raises(RuntimeError):
        ops.nms_sorted(torch.zeros(1, 4, 5), 0.5, 4)
    with pytest.raises(RuntimeError):
        nms(torch.rand(8, 5), 0.5)


def test_missing_library_is_an_error(monkeypatch):
    from sln_amodal_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsln.so")
    with pytest.raises(_lib.HipExtensionMissing):
        _lib.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sln_amodal_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text, fn
                assert "libsln_oracle" not in text, fn


def test_forward_tile_rule_is_a_pure_host_function(monkeypatch):
    """sln_conv_fwd_tile: 256x256 tiles only for wide outputs in whole rounds of 256 CUs with a
    long reduction; SLN_CONV_TILE256 = 0 / 2 override it (read on every call)."""
    from sln_amodal_amd import _lib
    L = _lib.lib()
    monkeypatch.delenv("SLN_CONV_TILE256", raising=False)
    assert L.sln_conv_fwd_tile(65536, 256, 2304, 3) == 256       # C4 3x3: 256 tiles, one per CU
    assert L.sln_conv_fwd_tile(65536, 64, 2304, 3) == 128        # narrow output
    assert L.sln_conv_fwd_tile(65536, 1024, 256, 3) == 256       # 1x1 expand: 16 stages are enough
    assert L.sln_conv_fwd_tile(65536, 256, 128, 3) == 128        # short reduction
    assert L.sln_conv_fwd_tile(67600, 256, 2304, 3) == 128       # 265 tiles: 52 % of two rounds
    assert L.sln_conv_fwd_tile(123440, 256, 2304, 3) == 256      # packed GLM scales: 483 tiles, 94 %
    assert L.sln_conv_fwd_tile(65536, 256, 2304, 2) == 256       # same rule for both operand formats
    assert L.sln_conv_fwd_tile(65536, 182, 18432, 3) == 256      # ASPP: 182 of 256 columns is enough
    assert L.sln_conv_fwd_tile(65536, 150, 18432, 3) == 128      # 150 is not
    assert L.sln_conv_fwd_tile(65536, 256 + 100, 18432, 3) == 128  # nor a 100-column second tile
    monkeypatch.setenv("SLN_CONV_TILE256", "0")
    assert L.sln_conv_fwd_tile(65536, 256, 2304, 3) == 128
    monkeypatch.setenv("SLN_CONV_TILE256", "2")
    assert L.sln_conv_fwd_tile(100, 8, 8, 2) == 256


def test_wgrad_tile_rule_is_a_pure_host_function(monkeypatch):
    """sln_conv_wgrad_tile: 256x256 (tap, Cout, Cin) tiles only when both channel counts are wide,
    the tap tiles fit one round of the 256 CUs and there are enough pixels."""
    from sln_amodal_amd import _lib
    L = _lib.lib()
    monkeypatch.delenv("SLN_WGRAD_TILE256", raising=False)
    assert L.sln_conv_wgrad_tile(65536, 256, 256, 9, 3) == 256      # C4 3x3
    assert L.sln_conv_wgrad_tile(65536, 1024, 256, 1, 3) == 256     # C4 1x1 expand
    assert L.sln_conv_wgrad_tile(1048576, 64, 64, 9, 3) == 128      # narrow channels
    assert L.sln_conv_wgrad_tile(4096, 256, 256, 9, 3) == 128       # too few pixels
    assert L.sln_conv_wgrad_tile(65536, 2048, 2048, 9, 3) == 128    # 576 tap tiles: more than one round
    assert L.sln_conv_wgrad_tile(65536, 256, 256, 9, 2) == 256      # same rule for both operand formats
    monkeypatch.setenv("SLN_WGRAD_TILE256", "0")
    assert L.sln_conv_wgrad_tile(65536, 256, 256, 9, 3) == 128
    monkeypatch.setenv("SLN_WGRAD_TILE256", "2")
    assert L.sln_conv_wgrad_tile(100, 8, 8, 1, 2) == 256


def test_environment_is_ignored_without_the_debug_switch():
    """The shipped contract: no entry point reads the environment.  In a process started WITHOUT
    SLN_DEBUG_KNOBS the tile knobs have no effect."""
    import subprocess
    import sys
    code = ("from sln_amodal_amd import _lib; L = _lib.lib(); "
            "print(L.sln_conv_fwd_tile(100, 8, 8, 3), L.sln_conv_wgrad_tile(100, 8, 8, 1, 3), "
            "L.sln_conv_fwd_tile(65536, 256, 2304, 3))")
    env = {k: v for k, v in os.environ.items() if k != "SLN_DEBUG_KNOBS"}
    env.update(SLN_CONV_TILE256="2", SLN_WGRAD_TILE256="2", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    assert out.stdout.split() == ["128", "128", "256"]


def test_reference_import_names_resolve_after_dropin_install():
    """code written against the reference's top-level names
    (modal.modals, nms.nms_wrapper, roialign..., model, config, utils) imports this package."""
    import subprocess
    import sys
    code = (
        "import sln_amodal_amd.dropin as d; names = d.install(); "
        "from modal.modals import pyramid_roi_align, FPN, RPN, Classifier, Mask, ResNet, SamePad2d; "
        "from modal.Functions import proposal_layer, detection_target_layer, build_rpn_targets; "
        "from modal.loss import compute_rpn_class_loss; "
        "from modal.deeplabv2 import DeepLabV2_ResNet101_MSC; "
        "from nms.nms_wrapper import nms; "
        "from roialign.roi_align.crop_and_resize import CropAndResizeFunction, CropAndResize; "
        "import model, config, utils; "
        "import sln_amodal_amd.model as m; assert model is m and model.MaskRCNN is m.MaskRCNN; "
        "assert config.Config().IMAGE_SHAPE[0] == 1024; print(len(names))")
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True,
                         timeout=300, cwd="/tmp")
    assert out.returncode == 0, out.stderr[-2000:]
    assert int(out.stdout.strip()) >= 15
