cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -4
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03b_bench_steps20_warmup5.json 2> gpurun_out/r03b_bench_steps20.err
python -c "
import json; d=json.load(open('gpurun_out/r03b_bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], d['ms_per_frame_single'], d['ms_per_frame_single_with_resolve_to_host'], d['ms_per_frame_single_with_resolve_on_device'])"
