# developer A/B (one GPU call): bench.py of the working tree's library vs yolo_amd/lib/libty1_head.so (see lib_ab.sh)
# usage: ARGS="--dtype bf16" bash tools/experiments/bench_ab.sh
set -e
for r in 1 2 3; do
  for l in new head; do
    if [ $l = head ]; then export TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so; else unset TY1_LIB; fi
    timeout -k 10 100 python bench.py ${ARGS:---dtype bf16} --no-also --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$l', d['value'], d['ms_per_step'], d.get('repeat_without_events_images_per_sec'))"
  done
done
