import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
SCENE5 = os.path.join(SCENES, "hw09", "scene5.crtscene")      # BASELINE configs 1-2
SCENE8 = os.path.join(SCENES, "hw11", "scene8.crtscene")      # BASELINE config 3
SCENE2 = os.path.join(SCENES, "hw15", "scene2.crtscene")      # BASELINE configs 4-5
CONFIG_SCENES = {"scene5": SCENE5, "scene8": SCENE8, "hw15_scene2": SCENE2}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    # asked of the runtime, not of the product library: on a GPU box a library that is missing or does not load must FAIL the
    # -m gpu tests, not skip them
    if not os.path.exists("/dev/kfd"):
        return False
    import torch
    return torch.cuda.device_count() > 0


def pytest_collection_modifyitems(config, items):
    # -m gpu tests must run on a GPU box; anywhere else they are skipped, never silently passed on a CPU path
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_PKG = None


def load_pkg():
    """The product package (its directory name carries a hyphen, so import it by name through importlib)."""
    global _PKG
    if _PKG is None:
        import __graft_entry__ as ge
        ge.build()
        _PKG = importlib.import_module("simd-raytracer_amd")
    return _PKG


@pytest.fixture(scope="session")
def rtk():
    return load_pkg()


@pytest.fixture(scope="session")
def ora():
    import oracle
    oracle.build()
    return oracle
