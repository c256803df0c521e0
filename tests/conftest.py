import os
import sys

import pytest

# the suite forces kernel / tile variants through the library's debug knobs (csrc/conv.hip sln_knob):
# they are honoured only when this is set before the library's first call
os.environ.setdefault("SLN_DEBUG_KNOBS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# Collection order (the driver runs `-x`): the kernel-vs-oracle sweeps first, then BASELINE.json's configurations at
# their stated sizes (test_configs_gpu: deterministic, one / six steps -- round 6: ahead of everything that compares
# trajectories, so that no heavy-tailed draw can leave configs[2] / configs[4] unreported), then the assembled path
# against the reference's fixtures, the input pipeline, multi-rank, configs[4]'s kernels; multi-step parity after every
# single-step sweep; convergence / dynamics tests (test_zz_*) last.  Files not listed keep their alphabetical place
# between the listed ones and test_zz_*.
FILE_ORDER = ["test_abi_cpu", "test_oracle_cpu", "test_native_gpu", "test_tail_cpu", "test_tail_gpu",
              "test_optim_gpu", "test_conv_gpu", "test_precision_gpu", "test_f16_gpu", "test_configs_gpu",
              "test_model_cpu", "test_model_gpu", "test_e2e_gpu", "test_integration_gpu",
              "test_loader_cpu", "test_loader_gpu", "test_parallel_cpu", "test_parallel8_cpu", "test_parallel_gpu",
              "test_resnext_cpu", "test_resnext_gpu", "test_multistep_gpu"]
# (test_multistep_gpu: parity over several optimiser steps, held to a control replica's behaviour -- after every
# single-step sweep, so that its tolerances, which live in an amplifying system, cannot hide them either)


def _file_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name.startswith("test_zz"):
        return (2, 0, name)
    if name in FILE_ORDER:
        return (0, FILE_ORDER.index(name), name)
    return (1, 0, name)


def pytest_collection_modifyitems(config, items):
    import torch
    items.sort(key=_file_rank)       # stable: the order inside a file is kept
    # a hard limit per test (pytest-timeout): a hung rank or kernel must not eat the GPU box
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900))
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

