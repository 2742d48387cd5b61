"""bench.py must never report N GPUs from fewer devices: how it decides to run."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def args(**kw):
    a = bench.parse_args([])
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_single_gpu_without_launcher():
    assert bench.resolve_launch(args(gpus=1), env={}) == ("single",)


def test_rank_of_a_launcher_world():
    env = {"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8"}
    assert bench.resolve_launch(args(gpus=8), env=env) == ("rank", 8, 3, 3)


@pytest.mark.parametrize("gpus,world", [(8, 4), (1, 2), (2, 1)])
def test_world_size_that_differs_from_gpus_is_refused(gpus, world):
    with pytest.raises(SystemExit) as e:
        bench.resolve_launch(args(gpus=gpus), env={"RANK": "0", "WORLD_SIZE": str(world)})
    assert e.value.code == 2


def test_more_gpus_than_the_node_has_is_refused(monkeypatch):
    monkeypatch.setattr(bench, "visible_gpus", lambda: 1)
    for one_process in (False, True):
        with pytest.raises(SystemExit) as e:
            bench.resolve_launch(args(gpus=8, one_process=one_process), env={})
        assert e.value.code == 2


def test_enough_gpus_without_launcher_starts_the_ranks_itself(monkeypatch):
    monkeypatch.setattr(bench, "visible_gpus", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    kind, cmd = bench.resolve_launch(args(gpus=4), env={})
    assert kind == "spawn"
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")
    assert bench.resolve_launch(args(gpus=4, one_process=True), env={}) == ("group", 4)


def test_plain_invocation_with_too_many_gpus_exits_nonzero_and_prints_no_result():
    """The hole this closes: `python bench.py --gpus 8` on a node without 8 GPUs used to render
    on one device and print "n_gpus": 8."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
    assert r.stdout.strip() == ""
    assert "refusing" in r.stderr


def test_workload_keys_name_what_was_profiled():
    assert bench.workload_key(args()) == "atrium-1920x1080-b4"
    assert bench.workload_key(args(workload="terrain", tiles=78, view="top")) == "terrain-1920x1080-b4-t78-top"
    assert bench.workload_key(args(pmc_key="x")) == "x"
    import json
    table = json.load(open(os.path.join(ROOT, "profiles", "pmc_per_ray.json")))
    e = table["atrium-1920x1080-b4"]
    assert os.path.exists(os.path.join(ROOT, e["source"]))
    for k in ("fetch_bytes_per_ray", "write_bytes_per_ray", "l2_read_bytes_per_ray", "td_busy", "ta_busy"):
        assert e[k] > 0
    assert 0 < e["limiter_busy_frac"] <= 1.0
