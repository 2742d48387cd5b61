set -e
mkdir -p gpurun_out/side
( while true; do date >> gpurun_out/side/heartbeat.txt; sleep 45; done ) &
HB=$!
trap "kill $HB" EXIT
timeout -k 10 200 python bench.py --no-cpu-baseline --res 3840x2160 > gpurun_out/side/atrium_4k.json
timeout -k 10 200 python bench.py --no-cpu-baseline --workload blob --res 512x512 --max-bounces 8 > gpurun_out/side/blob.json
timeout -k 10 300 python bench.py --no-cpu-baseline --workload terrain --tiles 24 --instanced > gpurun_out/side/instanced.json
timeout -k 10 900 python bench.py --no-cpu-baseline --workload terrain --tiles 78 --stack-depth 48 --steps 16 > gpurun_out/side/terrain50m.json
for f in atrium_4k blob instanced terrain50m; do python -c "
import json,sys; d=json.loads(open('gpurun_out/side/$f.json').read()); print('$f', d['value'], d['stage_ms_per_step'])"; done
