"""The two shared libraries load without a GPU and export exactly what include/*.h declares;
no compute call is made here."""
import ctypes
import os
import re

import pytest

from raytracer_amd import device, group, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s_\w+)\s*\(" % prefix, text)))


def test_rthip_exports_every_declared_symbol():
    names = declared("rthip.h", "rthip")
    assert len(names) >= 24
    lib = device.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(device.EXPORTS) == names


def test_rthip_group_exports_every_declared_symbol():
    names = declared("rthip_group.h", "rthip_group")
    lib = group.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(group.EXPORTS) == names


def test_group_plan_stripes_every_frame_exactly_once():
    """GPU r of n renders frames first + r, first + r + n, ...: together all of them, once."""
    for first, total, n in ((1, 64, 8), (1, 5, 8), (7, 1024, 3), (1, 0, 4), (100, 33, 1)):
        seen = []
        for r in range(n):
            f, c = group.plan(first, total, n, r)
            assert f == first + r
            seen += [f + k * n for k in range(c)]
        assert sorted(seen) == list(range(first, first + total)), (first, total, n)


def test_rthost_exports_every_declared_symbol():
    names = declared("rthost.h", "rthost")
    assert len(names) >= 24
    lib = host.lib()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_device_is_an_error_not_a_fallback():
    """Without a gfx950 device creating a context fails loudly; nothing renders on the CPU."""
    lib = device.lib()
    if lib.rthip_device_count() > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    rc = lib.rthip_create(0, ctypes.byref(h))
    assert rc == -4 and not h.value
    assert b"no HIP device" in lib.rthip_last_error(None)
    with pytest.raises(device.RtHipError):
        device.Context(0)


def test_default_params_are_the_reference_constants():
    p = device.default_params()
    assert (p.max_bounces, p.term_prob, p.stack_depth, p.enable_sun, p.flags) == (8, 0.5, 23, 0, 0)
    assert abs(p.aperture_radius - 0.025) < 1e-9


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under raytracer_amd/ may reference it."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "rt_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_rust_ffi_crate_declares_every_c_entry_point():
    """rust/rthip-sys cannot be compiled here (no cargo); at least its extern blocks must name
    exactly the functions the two headers declare, so that it does not drift from the ABI."""
    src = open(os.path.join(ROOT, "rust", "rthip-sys", "src", "lib.rs")).read()
    rust = sorted(set(re.findall(r"pub fn (rthip_\w+)\s*\(", src)))
    want = sorted(set(declared("rthip.h", "rthip")) | set(declared("rthip_group.h", "rthip_group")))
    assert rust == want, (sorted(set(want) - set(rust)), sorted(set(rust) - set(want)))
