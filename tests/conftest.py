import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def testing_lib(tmp_path_factory):
    """The TEST variant of libmcalf_hip.so (-DMCALF_TESTING: only it reads the failure-injection switches
    MCALF_TEST_FAIL_PREFLIGHT / MCALF_TEST_XCD_MASK), built into a temporary directory from the product's sources and its
    kernel object; worker processes load it through MCALF_HIP_LIB.  The in-tree product library carries no such hook
    (tests/test_abi_symbols.py)."""
    import importlib
    bld = importlib.import_module("mc-alf_amd.build")
    out = str(tmp_path_factory.mktemp("testing_lib") / "libmcalf_hip_testing.so")
    return bld.build(testing=True, target=out)
