"""Make the reference repository's top-level import names resolve to this package.

    import sln_amodal_amd.dropin; sln_amodal_amd.dropin.install()
    from modal.modals import pyramid_roi_align          # reference: modal/modals.py
    from nms.nms_wrapper import nms                      # reference: nms/nms_wrapper.py
    from roialign.roi_align.crop_and_resize import CropAndResizeFunction
    import model, config, utils                          # reference: model.py, config.py, utils.py

The aliases are the SAME module objects (sys.modules entries), so the package's relative imports
keep working and there is one copy of every module.  install() refuses to shadow a module of that
name that is already imported from somewhere else (e.g. the reference itself on sys.path).
"""
import importlib
import sys

_PKG = __name__.rsplit(".", 1)[0]

ALIASES = {
    "config": "config",
    "utils": "utils",
    "model": "model",
    "modal": "modal",
    "modal.modals": "modal.modals",
    "modal.Functions": "modal.Functions",
    "modal.loss": "modal.loss",
    "modal.deeplabv2": "modal.deeplabv2",
    "modal.msc_deeplab": "modal.msc_deeplab",
    "modal.resnet_deeplab": "modal.resnet_deeplab",
    "nms": "nms",
    "nms.nms_wrapper": "nms.nms_wrapper",
    "nms.pth_nms": "nms.pth_nms",
    "roialign": "roialign",
    "roialign.roi_align": "roialign.roi_align",
    "roialign.roi_align.crop_and_resize": "roialign.roi_align.crop_and_resize",
    "roialign.roi_align.roi_align": "roialign.roi_align.roi_align",
}


def install(force=False):
    """Register the aliases; returns the list of names installed."""
    done = []
    for top, sub in ALIASES.items():
        try:
            mod = importlib.import_module(_PKG + "." + sub)
        except ImportError:
            continue
        cur = sys.modules.get(top)
        if cur is not None and cur is not mod and not force:
            raise ImportError("cannot alias %r to %s.%s: a different module of that name is already "
                              "imported from %s" % (top, _PKG, sub, getattr(cur, "__file__", "?")))
        sys.modules[top] = mod
        done.append(top)
    return done


def uninstall():
    for top, sub in ALIASES.items():
        mod = sys.modules.get(top)
        if mod is not None and getattr(mod, "__name__", "") == _PKG + "." + sub:
            del sys.modules[top]
