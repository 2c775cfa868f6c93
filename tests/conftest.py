import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs /root/reference (build container only; auto-skipped elsewhere)")


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/sam2")
    skip_ref = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "ref" in item.keywords and not have_ref:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
