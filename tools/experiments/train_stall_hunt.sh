# developer tool (one GPU call): the training leg of bench.py several times over, each in a fresh process: timed region vs its plain repeat
# usage: bash tools/experiments/train_stall_hunt.sh [runs] [extra bench flags]
set -e
R=${1:-6}; shift || true
for i in $(seq 1 $R); do
  timeout -k 10 200 python bench.py --mode train --steps 50 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/stall_$i.json 2> gpurun_out/stall_$i.err
  python3 -c "
import json
d=json.load(open('gpurun_out/stall_$i.json')); print($i, d['value'], d['repeat_without_events_images_per_sec'], d['cold_start_images_per_sec'], d['exchange']['ms_per_step_without_collective'], d['roofline']['kernel'])"
done
