#!/bin/bash
# One-frame-per-call latency of every library under raytracer_amd/lib/variants.
# Usage (GPU box): bash tools/variants_single.sh [single_frame.py arguments...]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for f in raytracer_amd/lib/variants/*.so; do
  v=$(basename $f .so)
  export RTHIP_LIBRARY=$PWD/$f
  s=$(timeout -k 10 200 python tools/single_frame.py "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['wall_ms_per_call'], d['stage_ms_per_call'])") || exit 1
  echo "$v: $s"
done
