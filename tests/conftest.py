import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_finish(session):
    """A process that holds two ROCm runtimes (torch's bundled one and the /opt/rocm the library links) must let
    torch find the GPU FIRST: initialised after the library's runtime it reports "No HIP GPUs are available" (seen when
    test_gpu_parity ran before test_gpu_configs).  So GPU runs initialise torch's before any test loads the library."""
    if any(item.get_closest_marker("gpu") for item in session.items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "appendix_c.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def forest_paths():
    return {
        "zero": os.path.join(ROOT, "forests", "defaultZeroForest.txt"),
        "tau": os.path.join(ROOT, "forests", "defaultTauForest.txt"),
    }
