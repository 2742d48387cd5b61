"""Condenses the rocprofv3 CSVs written by tools/prof.sh into one text summary per kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print(f"# rocprofv3 summary for {out}")
for f in find("kt/**/*kernel_stats.csv"):
    print("\n## kernel-trace --stats (", os.path.relpath(f, out), ")")
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        print("  {:70s} calls {:>6s} total_ns {:>14s} avg_ns {:>12s} pct {:>6s}".format(
            r.get("Name", "")[:70], r.get("Calls", ""), r.get("TotalDurationNs", ""),
            r.get("AverageNs", ""), r.get("Percentage", "")))

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
