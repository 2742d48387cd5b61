"""The headless render harness (python -m raytracer_amd.render): argument handling on the CPU,
and on the GPU that the PNG it writes holds exactly the bytes rthip_resolve returns."""
import gzip
import io
import os

import numpy as np
import pytest
from PIL import Image

from raytracer_amd import render
from util import GOLDEN, REFERENCE_SCENES


def baked(tmp_path, name="simple_with_camera"):
    p = tmp_path / f"{name}.rtscene"
    p.write_bytes(gzip.open(os.path.join(GOLDEN, REFERENCE_SCENES[name]), "rb").read())
    return str(p)


def test_defaults_are_the_reference_literals():
    a = render.build_parser().parse_args(["--scene", "x.glb", "-o", "y.png"])
    assert (a.res, a.first_frame, a.max_bounces, a.term_prob, a.aperture, a.stack_depth, a.sun, a.gpus) == \
        ((512, 512), 1, 8, 0.5, 0.025, 23, False, 1)


@pytest.mark.parametrize("argv,msg", [
    (["--scene", "/nonexistent/x.glb", "-o", "y.png"], "does not exist"),
    (["--scene", "x.glb"], "nothing to write"),
    (["--scene", "x.glb", "-o", "y.png", "--spp", "0"], "at least 1"),
])
def test_bad_arguments_are_refused_before_any_gpu_work(argv, msg):
    with pytest.raises(SystemExit) as e:
        render.main(argv)
    assert msg in str(e.value)


def test_bad_resolution_is_an_argument_error(capsys):
    with pytest.raises(SystemExit):
        render.main(["--scene", "x.glb", "-o", "y.png", "--res", "640by480"])
    assert "WIDTHxHEIGHT" in capsys.readouterr().err


def test_png_writer_round_trips():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    back = np.asarray(Image.open(io.BytesIO(render.png_bytes(img))).convert("RGBA"))
    assert np.array_equal(back, img)


def test_camera_index_outside_the_scene_is_refused(tmp_path):
    with pytest.raises(SystemExit) as e:
        render.main(["--scene", baked(tmp_path), "-o", str(tmp_path / "o.png"), "--camera", "3"])
    assert "camera 3 of 1" in str(e.value)


@pytest.mark.gpu
def test_cli_png_equals_resolve_and_dumps_match_the_api(tmp_path, capsys):
    import json
    from raytracer_amd import device, host
    scene_file = baked(tmp_path)
    out, acc, hits, nodes = (str(tmp_path / n) for n in ("o.png", "a.npy", "h.npz", "n.npz"))
    rc = render.main(["--scene", scene_file, "--res", "320x180", "--spp", "6", "--first-frame", "3", "--max-bounces", "5",
                      "-o", out, "--dump-accum", acc, "--dump-hits", hits, "--dump-nodes", nodes])
    assert rc == 0
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["spp"] == 6 and line["rays"] > 320 * 180 * 6
    # the same render through the API
    s = host.Scene.from_cache(scene_file)
    from raytracer_amd import procedural
    s.set_env(procedural.sky_env(1024, 512))
    c = device.Context(0)
    try:
        c.upload_scene(s)
        c.set_frame_uniforms(s.frame_uniforms((320, 180), frame=3, reject_hist=1))
        c.render(3, 6, 1, device.default_params(max_bounces=5, flags=device.FLAG_TAPS))
        want_img, want_acc, want_taps = c.resolve(), c.read_accum(), c.read_taps()
        want_nodes = c.read_nodes_device_layout()
    finally:
        c.close()
    got_img = np.asarray(Image.open(out).convert("RGBA"))
    assert np.array_equal(got_img, want_img)
    assert np.array_equal(np.load(acc), want_acc)
    h = np.load(hits)
    for k in ("tri", "prim", "t", "seed"):
        assert np.array_equal(h[k], want_taps[k]), k
    assert np.load(nodes)["nodes"].tobytes() == want_nodes.tobytes()
