import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

REFERENCE_RESOURCES = "/root/reference/resources"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def reference_resources():
    if not os.path.isdir(REFERENCE_RESOURCES):
        pytest.skip("reference assets are not mounted here")
    return REFERENCE_RESOURCES
