import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "eta-inversion_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    import os
    import torch
    # the CPU oracle's small PyTorch ops crawl when oversubscribed (256 hardware threads on the GPU box)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    if torch.cuda.is_available():
        # every Engine() of the GPU suite loads the same seeded synthetic weights: draw them once per session (3.4 GB of host memory)
        from etainv.weights import memoize_synthetic
        memoize_synthetic(True)
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU oracle replay")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(GOLDEN / f"{name}.npz")
    return load


def pytest_collection_modifyitems(config, items):
    import os
    if os.environ.get("ETAINV_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow oracle replay; set ETAINV_SLOW=1")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)
