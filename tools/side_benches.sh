#!/bin/bash
# Side benchmarks beside the headline run (on the GPU box, from the repo root):
#   bash tools/side_benches.sh   -> gpurun_out/side/*.json
set -e
mkdir -p gpurun_out/side
B="python bench.py --no-cpu-baseline --no-single-frame --no-hbm-case"
timeout -k 10 300 $B --res 3840x2160 > gpurun_out/side/atrium_4k.json
echo atrium_4k done
timeout -k 10 300 $B --workload blob --res 512x512 --max-bounces 8 > gpurun_out/side/blob.json
echo blob done
timeout -k 10 300 $B --workload terrain --tiles 24 --instanced > gpurun_out/side/instanced.json
echo instanced done
timeout -k 10 300 $B --enable-sun > gpurun_out/side/atrium_sun.json
echo sun done
timeout -k 10 300 $B --max-bounces 8 > gpurun_out/side/atrium_8_bounces.json
echo 8 bounces done
timeout -k 10 300 $B --term-prob 0 > gpurun_out/side/atrium_no_termination.json
echo term-prob 0 done
timeout -k 10 300 $B --spp-total 1024 --res 3840x2160 > gpurun_out/side/atrium_4k_1024spp_strong.json
echo 4k 1024 spp done
timeout -k 10 900 $B --workload terrain --tiles 78 --stack-depth 48 --steps 16 --warmup 16 > gpurun_out/side/terrain50m_ground.json
echo terrain ground done
timeout -k 10 900 $B --workload terrain --tiles 78 --view top --stack-depth 48 --steps 16 --warmup 16 --random-rays 16000000 > gpurun_out/side/terrain50m_top.json
echo terrain top done
python - <<'PY'
import json, glob, os
out = {}
for f in sorted(glob.glob("gpurun_out/side/*.json")):
    d = json.loads(open(f).read())
    out[os.path.basename(f)[:-5]] = d
    r = d["roofline"]
    print(os.path.basename(f)[:-5], d["value"], "Mrays/s", d["ms_per_step"], "ms/step", "hbm frac", r.get("frac"), "alg GB/s", r.get("algorithmic_gbs"), d.get("random_rays", ""))
json.dump(out, open("gpurun_out/side/all.json", "w"), indent=1)
PY
