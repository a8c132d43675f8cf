import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _synthetic_weight_cache(tmp_path_factory):
    """The full-size synthetic UNet tensors are drawn once per session and re-read by every test and bench.py subprocess that
    builds them (sketch2img_amd/synthetic.py: SKG_SYNTH_CACHE_DIR; exact - the values are fp16-representable)."""
    import torch
    if torch.cuda.is_available() and "SKG_SYNTH_CACHE_DIR" not in os.environ:      # (the CPU suite builds them once: nothing to share)
        os.environ["SKG_SYNTH_CACHE_DIR"] = str(tmp_path_factory.mktemp("synth_cache"))
    yield
