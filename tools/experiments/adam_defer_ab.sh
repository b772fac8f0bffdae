# developer A/B (one GPU call): where the Adam launch of the update in flight lands under the next batch's backbone
set -e
for k in ${KS:--1 0 1 2 3 4 5 -1 1 2}; do
  echo "update-after-layer $k"
  timeout -k 10 120 python bench.py --mode train --update-after-layer $k --steps ${STEPS:-200} --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('repeat_without_events_images_per_sec'))"
done
