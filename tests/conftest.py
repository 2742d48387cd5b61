import os
import sys

import pytest

# PyTorch brings its own copy of the HIP and HSA runtimes; librthip.so links ROCm's.  Whichever
# runtime opens the GPU first keeps it: torch loaded after a device context exists reports "No HIP
# GPUs are available".  Loaded first, both share torch's copy (same soname).  Several GPU tests hand
# torch tensors to the library, so torch is imported before any test can load the library.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

REFERENCE_RESOURCES = "/root/reference/resources"

# The device library honours its RTHIP_* experiment knobs (forced candidate walks, frames in flight,
# ...) only in contexts created while RTHIP_TUNING is set; several tests force a walk that way.
os.environ["RTHIP_TUNING"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def reference_resources():
    if not os.path.isdir(REFERENCE_RESOURCES):
        pytest.skip("reference assets are not mounted here")
    return REFERENCE_RESOURCES
