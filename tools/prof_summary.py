"""Condenses the rocprofv3 CSVs written by tools/prof.sh into one text summary per kernel
(stdout) and, for the extend kernel of the timed instantiation, into per-ray figures that
bench.py scales to any --steps: <out>/pmc_entry.json, to be merged into profiles/pmc_per_ray.json
with tools/prof_merge.py."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
N_CU = 256
N_XCD = 8


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print(f"# rocprofv3 summary for {out}")
kernel_ns = {}
for f in find("kt/**/*kernel_stats.csv"):
    print("\n## kernel-trace --stats (", os.path.relpath(f, out), ")")
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        print("  {:70s} calls {:>6s} total_ns {:>14s} avg_ns {:>12s} pct {:>6s}".format(
            r.get("Name", "")[:70], r.get("Calls", ""), r.get("TotalDurationNs", ""),
            r.get("AverageNs", ""), r.get("Percentage", "")))
        kernel_ns[r.get("Name", "")] = (int(r.get("Calls", 0) or 0), float(r.get("TotalDurationNs", 0) or 0))

sums = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in find("*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:60]
        c = r.get("Counter_Name", "")
        sums[k][c] += float(r.get("Counter_Value", 0) or 0)
        calls[k][c] += 1
print("\n## PMC counters, summed over dispatches (count of dispatches in brackets)")
for k in sorted(sums):
    print(f"\n  {k}")
    for c in sorted(sums[k]):
        print(f"    {c:36s} {sums[k][c]:>20.0f}  [{calls[k][c]}]  per-dispatch {sums[k][c] / max(calls[k][c], 1):>16.1f}")
    s = sums[k]
    if s.get("SQ_ACTIVE_INST_VALU") and s.get("SQ_THREAD_CYCLES_VALU"):
        print(f"    -> VALU lane utilisation = THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64) = {s['SQ_THREAD_CYCLES_VALU'] / (s['SQ_ACTIVE_INST_VALU'] * 64):.3f}")
    if s.get("TCC_HIT_sum") is not None and (s.get("TCC_HIT_sum", 0) + s.get("TCC_MISS_sum", 0)) > 0:
        print(f"    -> L2 hit rate = {s['TCC_HIT_sum'] / (s['TCC_HIT_sum'] + s['TCC_MISS_sum']):.4f}")

# ---- per-ray entry for the extend kernel the timed region runs (the instantiation without counters)
bench = None
try:
    for line in open(os.path.join(out, "kt.log")):
        if line.startswith("{"):
            bench = json.loads(line)
except OSError:
    pass
ext = [k for k in sums if k.startswith("void k_extend<false")]
if bench and ext:
    k = ext[0]
    s = sums[k]
    steps, warm = bench["steps"], bench["warmup"]
    rays = bench["rays_per_step_per_gpu"] * (steps + warm)      # warm-up and timed region render the same frames
    n_disp = calls[k].get("FETCH_SIZE", 0)
    cu_cycles = s["GRBM_GUI_ACTIVE"] / N_XCD * N_CU if s.get("GRBM_GUI_ACTIVE") else None
    ta = s["TA_TA_BUSY_sum"] / cu_cycles if cu_cycles else None
    td = s["TD_TD_BUSY_sum"] / cu_cycles if cu_cycles else None
    entry = {
        "kernel": k,
        "rays_profiled": rays,
        "dispatches_profiled": n_disp,
        # FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at
        # 64 bytes, so it is doubled
        "fetch_bytes_per_ray": 2.0 * s["FETCH_SIZE"] * 1024.0 / rays,
        "write_bytes_per_ray": s["WRITE_SIZE"] * 1024.0 / rays,
        # L1 -> L2 read requests, priced at 64 bytes each
        "l2_read_bytes_per_ray": 64.0 * s["TCP_TCC_READ_REQ_sum"] / rays,
        "l2_hit_rate": round(s["TCC_HIT_sum"] / (s["TCC_HIT_sum"] + s["TCC_MISS_sum"]), 4),
        "l1_accesses_per_ray": s["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays,
        "ta_busy": round(ta, 4) if ta is not None else None,
        "td_busy": round(td, 4) if td is not None else None,
        "td_tc_stall": round(s["TD_TC_STALL_sum"] / cu_cycles, 4) if cu_cycles else None,
        "valu_lane_utilisation": round(s["SQ_THREAD_CYCLES_VALU"] / (s["SQ_ACTIVE_INST_VALU"] * 64), 4),
        "limiter_unit": None, "limiter_busy_frac": None,
        "kernel_trace_avg_ms": None,
        "bench_args": open(os.path.join(out, "bench_args.txt")).read().strip() if os.path.exists(os.path.join(out, "bench_args.txt")) else None,
        "source": None,
    }
    if ta is not None:
        # non-idle fractions of the vector-memory units: what the waves wait on, not a throughput utilisation
        unit, frac = max((("TD (vector-memory data return, non-idle fraction)", td), ("TA (vector-memory address unit, non-idle fraction)", ta)), key=lambda x: x[1])
        entry["limiter_unit"], entry["limiter_busy_frac"] = unit, round(frac, 4)
    for name, (n, total) in kernel_ns.items():
        if name.startswith("void k_extend<false") and n:
            entry["kernel_trace_avg_ms"] = round(total / n / 1e6, 4)
    json.dump(entry, open(os.path.join(out, "pmc_entry.json"), "w"), indent=1)
