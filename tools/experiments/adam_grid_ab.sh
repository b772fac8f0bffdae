# developer A/B (one GPU call, dev library): Adam's grid size x the layer behind which it is launched, training step at 32 images
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for b in ${BLOCKS:-4096 1024 512 256}; do
  for k in ${KS:--1 1 4}; do
    echo -n "adam blocks $b, update-after-layer $k: "
    TY1_ADAM_BLOCKS=$b timeout -k 10 120 python bench.py --mode train --no-cpu-baseline --no-also --update-after-layer $k --steps ${STEPS:-200} --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d.get('repeat_without_events_images_per_sec'))"
  done
done
