import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test limit in seconds (pytest-timeout when installed; a no-op marker otherwise)")


def load_golden(name):
    # the two attribute banks are product data (reference tables the package ships): excel_amd/attributes_text/
    if name.startswith("attr_bank_"):
        return np.load(os.path.join(ROOT, "excel_amd", "attributes_text", name))
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden():
    return load_golden
