"""Files a profiled run under profiles/: copies <prof dir>/summary.txt to profiles/<name>_summary.txt
and merges <prof dir>/pmc_entry.json into profiles/pmc_per_ray.json under the workload key.

    python tools/prof_merge.py gpurun_out/prof_<tag> <workload key> <name>
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, key, name = sys.argv[1:4]
dst = os.path.join(ROOT, "profiles", f"{name}_summary.txt")
shutil.copyfile(os.path.join(src, "summary.txt"), dst)
entry = json.load(open(os.path.join(src, "pmc_entry.json")))
entry["source"] = os.path.relpath(dst, ROOT)
path = os.path.join(ROOT, "profiles", "pmc_per_ray.json")
table = json.load(open(path)) if os.path.exists(path) else {}
table[key] = entry
json.dump(table, open(path, "w"), indent=1, sort_keys=True)
print(f"{key}: fetch {entry['fetch_bytes_per_ray']:.1f} + write {entry['write_bytes_per_ray']:.1f} B/ray, "
      f"L2 hit {entry['l2_hit_rate']}, TA {entry['ta_busy']} TD {entry['td_busy']} -> {dst}")
