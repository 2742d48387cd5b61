#!/bin/bash
# instanced terrain at growing primitive counts with the library's own choice of candidate walk
set -e
mkdir -p gpurun_out
out=gpurun_out/instance_scaling.jsonl
: > $out
run() {
  timeout -k 10 300 python bench.py --workload terrain --tiles $1 --terrain-n 16 --instanced $2 --steps 64 --warmup 64 --no-cpu-baseline --no-single-frame --no-hbm-case > gpurun_out/inst_one.json
  python - "$1" "$2" <<'PY' >> $out
import json,sys
j=json.loads(open('gpurun_out/inst_one.json').read().strip().splitlines()[-1])
print(json.dumps(dict(tiles=int(sys.argv[1]), order="shuffled" if sys.argv[2] else "row-major", primitives=j["config"]["primitives"],
                      walk=j["config"]["candidate_walk"], mrays_per_s=j["value"], ms_per_frame=j["ms_per_step"])))
PY
  tail -1 $out
}
for tiles in 8 24 28 64 90 180 256; do run $tiles ""; done
for tiles in 32 64; do run $tiles --shuffled; done
