#!/bin/bash
# Profiling recipe for the benchmark: one kernel-trace pass and separate counter passes (rocprofv3
# does not split counters over passes), then tools/prof_summary.py condenses them into
# summary.txt and a per-ray entry for profiles/pmc_per_ray.json.
# Usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
STEPS=${PROF_STEPS:-64}
ARGS="--steps $STEPS --warmup ${PROF_WARMUP:-$STEPS} --no-cpu-baseline --no-single-frame --no-hbm-case $*"
LIMIT=${PROF_LIMIT:-240}
CMD=${PROF_CMD:-bench.py}
echo "$ARGS" > "$OUT/bench_args.txt"
timeout -s KILL $LIMIT rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 $CMD $ARGS > "$OUT/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.log"; exit 1; }
echo "kernel-trace pass done"
pass() { # name counters...
  local name=$1; shift
  timeout -s KILL $LIMIT rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $CMD $ARGS > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
  echo "pmc pass $name done"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY && \
pass sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU && \
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum && \
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum && \
pass fetch FETCH_SIZE && \
pass write WRITE_SIZE && \
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE || exit 1
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
grep -A45 "k_extend<false" "$OUT/summary.txt" | head -70
cat "$OUT/pmc_entry.json"
