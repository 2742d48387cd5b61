"""A bounded, fixed-seed draw from the randomised differential cases (tests/fuzz_cases.py): the HIP
path against the CPU oracle over random scenes, candidate walks, resolutions, bounce counts,
termination probabilities, stack depths 1..64, sun on and off, one to four frames per call in the
short and in the staged form, and single rays of which a third are axis-parallel (infinite
reciprocals in the slab tests, shader.wgsl:415-438) and half of those start on round coordinates."""
import os

import numpy as np
import pytest

import fuzz_cases

SEED = 20261
N_CASES = 40


def draw_cases():
    rng = np.random.default_rng(SEED)
    return rng, [fuzz_cases.make_case(rng) for _ in range(N_CASES)]


def test_the_draw_covers_what_it_should():
    """No GPU needed for this part: the fixed seed must reach every candidate walk, both sun settings,
    both forms of a short call and the ends of the stack-depth range."""
    _, cases = draw_cases()
    assert {c["walk"] for c in cases} >= {None, "default", "scan", "pyramid", "list"}
    assert {c["params"].enable_sun for c in cases} == {0, 1}
    depths = {c["params"].stack_depth for c in cases}
    assert 1 in depths and 64 in depths and 23 in depths
    assert {c["staged"] for c in cases if c["n"] <= 3} == {False, True}
    assert any(c["n"] == 4 for c in cases)
    assert {c["kind"] for c in cases} == {"box", "materials", "terrain", "instanced", "instanced_shuffled", "atrium"}


@pytest.mark.gpu
def test_random_cases_agree_with_the_oracle():
    rng, cases = draw_cases()
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    failed, walks = [], set()
    for i, case in enumerate(cases):
        fail, report, walk = fuzz_cases.run_case(case, rng, n_rays=24, threads=threads)
        walks.add(walk)
        if fail:
            failed.append(f"case {i}: {report}")
    assert not failed, "\n".join(failed)
    assert walks == {"mask", "scan", "pyramid", "list"}, walks
