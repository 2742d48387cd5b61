"""tests/tools/compare_dump.py: the dump format of the reference runner (rust/ref-runner) and the
comparison itself, exercised with the CPU oracle standing in for the runner."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np

from util import GOLDEN, REFERENCE_SCENES

TOOL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "compare_dump.py")


def run(*argv):
    r = subprocess.run([sys.executable, TOOL, *argv], capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


def test_inputs_dump_compare_round_trip(tmp_path):
    src = tmp_path / "box_in.rtscene"
    src.write_bytes(gzip.open(os.path.join(GOLDEN, REFERENCE_SCENES["box"]), "rb").read())
    prefix = str(tmp_path / "box")
    rc, out, err = run("inputs", "--scene", str(src), "--res", "48x32", "--frame", "2", "-o", prefix)
    assert rc == 0, err
    u = np.fromfile(prefix + ".uniforms", dtype=np.uint8)
    assert u.size == 608 and tuple(u[576:588].view(np.uint32)) == (48, 32, 2)
    rc, out, err = run("make-dump", "--scene", prefix + ".rtscene", "--uniforms", prefix + ".uniforms", "--frames", "4", "-o", prefix + ".refdump")
    assert rc == 0, err
    raw = open(prefix + ".refdump", "rb").read()
    assert raw[:8] == b"RTDUMP01" and len(raw) == 40 + 48 * 32 * 16
    rc, out, err = run("compare", "--scene", prefix + ".rtscene", "--uniforms", prefix + ".uniforms", "--dump", prefix + ".refdump", "--renderer", "oracle")
    assert rc == 0, err
    d = json.loads(out.strip().splitlines()[-1])
    assert d["sample_counts_equal"] and d["pixels_bit_identical"] == 1.0 and d["frames"] == 4 and d["first_frame"] == 2
    # a dump that differs is reported and fails
    b = bytearray(raw)
    b[40 + 12: 40 + 16] = np.float32(7.0).tobytes()      # sample count of pixel 0
    open(prefix + ".bad", "wb").write(bytes(b))
    rc, out, err = run("compare", "--scene", prefix + ".rtscene", "--uniforms", prefix + ".uniforms", "--dump", prefix + ".bad", "--renderer", "oracle")
    assert rc == 1 and not json.loads(out.strip().splitlines()[-1])["sample_counts_equal"]
