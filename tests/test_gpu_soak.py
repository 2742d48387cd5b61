"""A short soak under pytest (the long form is tests/tools/soak_gpu.py --rounds 40): one context through
eight rounds of scene upload / resolution change / render of 1..39 frames with random bounce count,
stack depth and sun option / ray query / resolve, every accumulator finite with the right sample
count; then a second context the same way.  What the second context leaves behind after it is closed
is what the library leaks: it must be under 8 MiB (the first context also pays for what the runtime
keeps per process)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_soak_two_contexts_leave_no_device_memory_behind():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    try:
        import soak_gpu
    finally:
        sys.path.pop(0)
    lines = []
    free0, low1, free1, low2, free2 = soak_gpu.soak(8, log=lines.append)
    assert len(lines) == 16
    assert free1 - free2 < (8 << 20), (free0, low1, free1, low2, free2)
    assert low2 >= low1 - (64 << 20), (low1, low2)      # the second context needs no more than the first did
