# developer tool (one GPU call): the default bench line several times over with TY1_BENCH_TRACE (one stderr line per training
# step, the phases of every timed region, every full collection of Python's cyclic collector); odd runs without gc.freeze().
# What it found (round 3): the 40 ms host stall that sometimes fell into the training leg's timed region was a full collection.
# usage: bash tools/experiments/default_stall_hunt.sh [runs]
set -e
R=${1:-4}
for i in $(seq 1 $R); do
  if [ $((i % 2)) = 1 ]; then export TY1_BENCH_NO_GC_FREEZE=1; else unset TY1_BENCH_NO_GC_FREEZE; fi
  TY1_BENCH_TRACE=1 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/dstall_$i.json 2> gpurun_out/dstall_$i.err
  python3 -c "
import json,re
d=json.load(open('gpurun_out/dstall_$i.json')); t=d['also']['configs[3] training step']
print($i, 'no_freeze=${TY1_BENCH_NO_GC_FREEZE:-0}', d['value'], t['value'], t['repeat_without_events_images_per_sec'], t['exchange']['exposed_ms'])
ts=[float(m.group(1)) for m in re.finditer(r'done at ([0-9.]+) s', open('gpurun_out/dstall_$i.err').read())]
gaps=sorted(((b-a,k) for k,(a,b) in enumerate(zip(ts,ts[1:]))), reverse=True)[:4]
print('  steps', len(ts), 'largest host gaps (s, after step):', [(round(g,4),k+1) for g,k in gaps])"
  grep -n "full gc" gpurun_out/dstall_$i.err || true
done
