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


# ---- rust/rthip-sys against the headers: names, arity, and every parameter and return type ---------

C_SCALARS = {"int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "uint8_t": "u8", "float": "f32",
             "double": "f64", "char": "c_char", "void": "c_void", "size_t": "usize"}


def c_type_to_rust(ctype):
    """'const rthip_scene_desc*' -> '*const rthip_scene_desc'; 'uint32_t' -> 'u32'; 'void' -> ''."""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ")
    words = t.split()
    const = "const" in words
    base = [w for w in words if w not in ("const", "struct")]
    assert len(base) == 1, ctype
    name = C_SCALARS.get(base[0], base[0])
    if stars == 0:
        return "" if base[0] == "void" else name
    out = name
    for level in range(stars):
        # only the innermost pointee can be const in these headers (const T*, const T**: not used)
        out = ("*const " if (const and level == 0) else "*mut ") + out
    return out


def c_prototypes(header, prefix):
    """{name: (return type, [parameter types])} of a header, in the Rust spelling."""
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for ret, name, params in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(%s_\w+)\s*\(([^)]*)\)\s*;" % prefix, text):
        plist = []
        params = " ".join(params.split())
        if params and params != "void":
            for prm in params.split(","):
                m = re.match(r"(.*?)(\w+)$", prm.strip())      # the last word is the parameter's name
                plist.append(c_type_to_rust(m.group(1)))
        protos[name] = (c_type_to_rust(ret), plist)
    return protos


def rust_prototypes(src):
    protos = {}
    for name, params, ret in re.findall(r"pub fn (rthip_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;{]+))?;", src):
        plist = []
        for prm in " ".join(params.split()).split(","):
            if prm.strip():
                plist.append(" ".join(prm.split(":", 1)[1].split()))
        protos[name] = (" ".join((ret or "").split()), plist)
    return protos


def test_rust_ffi_crate_matches_the_headers_signature_by_signature():
    """rust/rthip-sys cannot be compiled here (no cargo), so its extern blocks are checked as text:
    the same functions as the two headers, each with the same number of parameters, the same
    parameter types in the same order, and the same return type."""
    src = open(os.path.join(ROOT, "rust", "rthip-sys", "src", "lib.rs")).read()
    rust = rust_prototypes(src)
    want = dict(c_prototypes("rthip.h", "rthip"))
    want.update(c_prototypes("rthip_group.h", "rthip_group"))
    assert sorted(rust) == sorted(want), (sorted(set(want) - set(rust)), sorted(set(rust) - set(want)))
    assert len(want) >= 40
    for name in sorted(want):
        assert rust[name] == want[name], (name, rust[name], want[name])


def test_rust_ffi_crate_mirrors_the_structs_and_constants():
    """Field order and types of the #[repr(C)] mirrors, the stage count and the flag values."""
    src = open(os.path.join(ROOT, "rust", "rthip-sys", "src", "lib.rs")).read()
    hdr = open(os.path.join(ROOT, "include", "rthip.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for struct in ("rthip_scene_desc", "rthip_render_params", "rthip_counters"):
        body = re.search(r"typedef struct \{([^{}]*)\} %s;" % struct, hdr_nc).group(1)
        c_fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            m = re.match(r"(.*?)(\w+(?:\s*,\s*\w+)*)$", decl)
            for fname in m.group(2).split(","):
                c_fields.append((fname.strip(), c_type_to_rust(m.group(1))))
        rbody = re.search(r"pub struct %s \{([^{}]*)\}" % struct, src).group(1)
        r_fields = [(n, " ".join(t.split())) for n, t in re.findall(r"pub (\w+):\s*([^,\n]+),", rbody)]
        assert r_fields == c_fields, (struct, r_fields, c_fields)
    for const in ("RTHIP_NUM_STAGES", "RTHIP_FLAG_COUNTERS", "RTHIP_FLAG_TAPS", "RTHIP_FLAG_TIMING", "RTHIP_FLAG_STAGED"):
        c_val = int(re.search(r"#define %s\s+(\d+)" % const, hdr).group(1))
        r_val = int(re.search(r"pub const %s: \w+ = (\d+);" % const, src).group(1))
        assert c_val == r_val, const
