#!/usr/bin/env python3
"""Randomised differential run of the HIP path against the CPU oracle, as long as you like (the
bounded, fixed-seed set that pytest collects is tests/test_gpu_fuzz.py; the cases come from
tests/fuzz_cases.py): every case must agree with the oracle in its primary hits, random-stream
states, counters, image and single rays.
    python tests/tools/fuzz_gpu.py [--cases 40] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
os.environ["RTHIP_TUNING"] = "1"   # the forced candidate walks are experiment knobs
import torch  # noqa: F401,E402  (before the device library opens the GPU)
import fuzz_cases  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rays", type=int, default=48, help="single rays per case through rthip_trace_rays")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    bad = 0
    for i in range(args.cases):
        case = fuzz_cases.make_case(rng)
        fail, report, _ = fuzz_cases.run_case(case, rng, n_rays=args.rays, threads=threads)
        bad += fail
        print(f"{'FAIL' if fail else 'ok  '} case {i}: {report}", flush=True)
    print(f"{args.cases - bad} of {args.cases} cases agree")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
