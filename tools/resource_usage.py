#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel of the device library, from the compiler's
-Rpass-analysis=kernel-resource-usage remarks.
    python tools/resource_usage.py [extra hipcc flags, e.g. -DRT_PATH_WAVES=5]
Compiles raytracer_amd/csrc/rthip.hip for gfx950 with the build's flags (device code only) and prints
one line per kernel."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
           "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
           os.path.join(ROOT, "raytracer_amd", "csrc", "rthip.hip")] + sys.argv[1:]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (?:Function Name: (\S+)|\s*([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+))", line)
        if not m:
            continue
        if m.group(1):
            cur = {"name": m.group(1)}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(2).strip()] = int(m.group(3))
    demangle = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for r, d in zip(rows, demangle):
        d = re.sub(r"\(.*", "", d.replace("void ", ""))
        print(f"{d:55s} VGPRs {r.get('VGPRs', 0):3d}  spilled {r.get('VGPRs Spill', 0):3d}  scratch {r.get('ScratchSize', 0):4d} B  "
              f"waves/SIMD {r.get('Occupancy', 0)}  LDS {r.get('LDS Size', 0):6d} B")
    return 0


if __name__ == "__main__":
    sys.exit(main())
