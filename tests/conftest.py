import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
