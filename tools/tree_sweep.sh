#!/bin/bash
# instanced terrain at growing primitive counts: chunk scan vs bounds pyramid in the extend stage
set -e
mkdir -p gpurun_out
out=gpurun_out/tree_sweep.jsonl
: > $out
for tiles in ${TILES:-24 64 90 180}; do
  for min in 4000000000 128; do
    echo "tiles=$tiles tree_min=$min"
    RTHIP_TUNING=1 RTHIP_TREE_MIN_PRIMS=$min timeout -k 10 200 python bench.py --workload terrain --tiles $tiles --terrain-n 16 --instanced ${EXTRA} --view ${VIEW:-ground} --steps 64 --warmup 64 --no-cpu-baseline --no-single-frame --no-hbm-case > gpurun_out/tree_one.json
    python - "$tiles" "$min" <<'PY' >> $out
import json,sys
j=json.loads(open('gpurun_out/tree_one.json').read().strip().splitlines()[-1])
print(json.dumps(dict(tiles=int(sys.argv[1]), tree_min=int(sys.argv[2]), mrays=j["value"], ms=j["ms_per_step"])))
PY
    tail -1 $out
  done
done
