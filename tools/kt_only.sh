#!/bin/bash
# Kernel-trace summary of one bench command (no counter passes): bash tools/kt_only.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "$*" > "$OUT/bench_args.txt"
timeout -s KILL ${PROF_LIMIT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py "$@" > "$OUT/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.log"; exit 1; }
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
head -8 "$OUT/summary.txt"
grep '^{"metric' "$OUT/kt.log" | tail -1 | cut -c1-200
