#!/bin/bash
# Runs the single-frame and the 64-frame-group measurements for every library under
# raytracer_amd/lib/variants (built with different -D knobs).  Usage: bash tools/variants.sh [names...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
NAMES=${@:-$(ls raytracer_amd/lib/variants | sed 's/\.so$//')}
for v in $NAMES; do
  export RTHIP_LIBRARY=$PWD/raytracer_amd/lib/variants/$v.so
  s=$(timeout -k 10 200 python tools/single_frame.py | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['wall_ms_per_call'], d['stage_ms_per_call']['extend'])") || exit 1
  b=$(timeout -k 10 300 python bench.py --steps 64 --warmup 64 --no-cpu-baseline --no-single-frame --no-hbm-case | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms_per_step']['extend'])") || exit 1
  echo "$v single-frame(ms wall, extend): $s   group64 (Mrays/s, extend ms/frame): $b"
done
