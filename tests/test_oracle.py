"""The CPU oracle against its pins: hand-derived integer RNG vectors, closed-form values of the
present pass, and the committed golden fixtures (which pin the oracle itself; the reference has
no fixtures -- parity unpinned)."""
import json
import os

import numpy as np
import pytest

from oracle import orc
from raytracer_amd import procedural
from util import oracle_scene

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_hash21_and_rand_known_answers():
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    for x, y, want in kat["hash21"]:
        assert orc.hash21(x, y) == want
    for s in kat["pixel_streams"]:
        seed = orc.pixel_seed(s["pixel"][0], s["pixel"][1], s["frame"])
        assert seed == s["seed0"]
        assert orc.rand_stream(seed, len(s["draws"])) == [tuple(d) for d in s["draws"]]


def test_hash21_matches_python_big_int_arithmetic():
    def h(x, y):
        m = 0xFFFFFFFF
        x, y = (x * 73333) & m, (y * 7777) & m
        x ^= 3333777777 >> (x >> 28)
        y ^= 3333777777 >> (y >> 28)
        n = (x * y) & m
        return n ^ (n >> 15)
    rng = np.random.default_rng(1)
    for x, y in rng.integers(0, 2 ** 32, size=(2000, 2), dtype=np.uint64):
        assert orc.hash21(int(x), int(y)) == h(int(x), int(y))


def test_rand_is_u32_times_two_to_minus_32():
    seed = 12378231
    for _ in range(100):
        (after, raw), = orc.rand_stream(seed, 1)
        f, after2 = orc.rand_float(seed)
        assert after2 == after
        assert f == np.float32(np.float32(raw) * np.float32(2.0 ** -32))
        assert 0.0 <= f <= 1.0
        seed = after


def test_tonemap_closed_forms():
    # below the compression knee only the toe offset applies (shader.wgsl:974-984)
    out = orc.tonemap([0.5, 0.3, 0.2])
    np.testing.assert_allclose(out, np.array([0.5, 0.3, 0.2]) - 0.04, rtol=1e-6)
    x = 0.05
    out = orc.tonemap([x, 0.1, 0.2])
    np.testing.assert_allclose(out, np.array([x, 0.1, 0.2]) - (x - 6.25 * x * x), rtol=1e-5)
    # above the knee the peak is compressed towards 1 and desaturated
    c = np.array([4.0, 2.0, 1.0]) - 0.04
    peak = c.max()
    d = 1 - 0.76
    new_peak = 1 - d * d / (peak + d - 0.76)
    g = 1 - 1 / (0.15 * (peak - new_peak) + 1)
    want = c * (new_peak / peak) * (1 - g) + new_peak * g
    np.testing.assert_allclose(orc.tonemap([4.0, 2.0, 1.0]), want, rtol=1e-5)


def test_resolve_divides_by_sample_count_and_encodes_srgb():
    acc = np.zeros((1, 3, 4), np.float32)
    acc[0, 0] = (8 * 0.24, 8 * 0.24, 8 * 0.24, 8)   # mean 0.24 -> 0.2 linear after the toe
    acc[0, 1] = (0.0, 0.0, 0.0, 1)
    acc[0, 2] = (100.0, 100.0, 100.0, 1)
    lin = orc.resolve(acc, apply_srgb_oetf=False)
    srgb = orc.resolve(acc, apply_srgb_oetf=True)
    assert lin[0, 0, 0] == round(0.2 * 255) and lin[0, 0, 3] == 255
    assert srgb[0, 0, 0] == round((1.055 * 0.2 ** (1 / 2.4) - 0.055) * 255)
    assert tuple(srgb[0, 1, :3]) == (0, 0, 0)
    assert srgb[0, 2, 0] >= 250


@pytest.mark.parametrize("name", ["box_scene", "materials_scene"])
def test_oracle_reproduces_golden_fixture(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    if name == "box_scene":
        s = procedural.box_scene(env_size=(64, 32))
        res, first, n, stride = (16, 16), 1, 1, 1
    else:
        s = procedural.material_test_scene(tex=16, env_size=(32, 16))
        res, first, n, stride = (24, 16), 3, 2, 5
    a = s.arrays()
    assert a["nodes"].view(np.uint8).tobytes() == g["nodes"].tobytes()      # host builder output is pinned too
    assert np.array_equal(a["tri_order"], g["tri_order"])
    u = s.frame_uniforms(res, frame=first, reject_hist=1)
    assert u.tobytes() == g["uniforms"].tobytes()
    r = orc.render(oracle_scene(a), u, first_frame=first, n_frames=n, frame_stride=stride, taps=True)
    assert np.array_equal(r["tri"], g["hit_tri"]) and np.array_equal(r["prim"], g["hit_prim"])
    assert np.array_equal(r["seed"], g["seed"])
    assert np.array_equal(r["t"], g["hit_t"])
    assert np.array_equal(r["accum"], g["accum1"] if name == "box_scene" else g["accum"])
    if name == "box_scene":
        r64 = orc.render(oracle_scene(a), u, first_frame=1, n_frames=64)
        assert np.array_equal(r64["accum"], g["accum64"])
    else:
        assert [r["counters"][k] for k in orc.COUNTER_NAMES] == [int(v) for v in g["counters"]]


def test_threads_and_row_sampling_do_not_change_results():
    s = procedural.material_test_scene(tex=16, env_size=(32, 16))
    a = s.arrays()
    u = s.frame_uniforms((40, 24))
    one = orc.render(oracle_scene(a), u, n_frames=2, n_threads=1, taps=True)
    many = orc.render(oracle_scene(a), u, n_frames=2, n_threads=4, taps=True)
    assert np.array_equal(one["accum"], many["accum"]) and one["counters"] == many["counters"]
    rows = orc.render(oracle_scene(a), u, n_frames=2, n_threads=2, row_start=1, row_step=3)
    assert np.array_equal(rows["accum"][1::3], one["accum"][1::3])
    assert not rows["accum"][0::3].any()


def test_expected_ray_budget():
    """<= sum_{k<8} 0.5^k = 1.992 trace() calls per pixel (shader.wgsl:896-914)."""
    s = procedural.material_test_scene(tex=16, env_size=(32, 16))
    a = s.arrays()
    r = orc.render(oracle_scene(a), s.frame_uniforms((64, 48)), n_frames=4, n_threads=4)
    c = r["counters"]
    per_pixel = c["rays"] / (64 * 48 * 4)
    assert 1.0 <= per_pixel <= 1.992 + 0.05
    assert c["rays"] == c["misses"] + c["kills"] + c["shade_calls"]


def test_single_ray_trace_box():
    s = procedural.box_scene(env_size=(16, 8))
    sc = oracle_scene(s.arrays())
    tri, prim, t, _ = orc.trace_ray(sc, (-5.0, 0.0, 0.0), (1.0, 0.0, 0.0))
    assert tri >= 0 and prim == 0 and abs(t - 4.5) < 1e-5
    tri, prim, t, _ = orc.trace_ray(sc, (-5.0, 0.0, 2.0), (1.0, 0.0, 0.0))
    assert tri == -1 and prim == -1


def test_builder_on_a_tree_derived_by_hand():
    """Four unit triangles on the x axis at 0, 10, 20, 30 (scene.rs:888-1050 by hand): four
    triangles are not a leaf (count > 2), fewer than 64 take the exact-centroid SAH, whose cost
    2*A(11) + 2*A(11) for the split between the pairs beats 1*A(1) + 3*A(21) on either side by far,
    so the root becomes inner with its children appended as nodes 1 and 2, each a leaf of two
    triangles (count <= 2 stops the recursion); the partition (scene.rs:1004-1022) walks i up over
    triangles 0 and 1, swaps triangle 2 with the last one and re-examines the swapped-in triangle 3,
    which leaves the order 0, 1, 3, 2; every box is the min/max of its vertices widened by 0.00001
    in f32."""
    from raytracer_amd import layouts as L
    xs = np.array([0.0, 10.0, 20.0, 30.0], np.float32)
    tris = np.zeros(4, dtype=L.TRI)
    for i, x in enumerate(xs):
        p = np.array([[x, 0, 0], [x + 1, 0, 0], [x, 1, 0]], np.float32)
        tris["v"][i, :, :3] = p
        tris["v"][i, :, 3] = ((p[0] + p[1]) + p[2]) / np.float32(3.0)
    nodes, idx = orc.bvh_build(tris, 0, 4)
    assert nodes.shape == (3, 8)
    assert np.array_equal(idx, [0, 1, 3, 2])
    eps = np.float32(0.00001)

    def box(a, b):
        v = tris["v"][a:b, :, :3].reshape(-1, 3)
        return np.concatenate([v.min(0) - eps, v.max(0) + eps]).astype(np.float32)

    first_count = nodes.view(np.uint32)[:, 6:8]
    assert first_count.tolist() == [[1, 0], [0, 2], [2, 2]]
    assert np.array_equal(nodes[0, :6], box(0, 4))
    assert np.array_equal(nodes[1, :6], box(0, 2))
    assert np.array_equal(nodes[2, :6], box(2, 4))
