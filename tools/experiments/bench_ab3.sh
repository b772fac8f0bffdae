# developer A/B (one GPU call): bench.py --dtype bf16 whole-step time under several builds of the library (timed region / un-instrumented repeat)
set -e
run() {
  timeout -k 10 100 python bench.py --dtype bf16 --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('%-28s %9.1f %7.4f   plain %9.1f'%('$1', d['value'], d['ms_per_step'], d['repeat_without_events_images_per_sec']))"
}
for r in 1 2 3 4; do
  for l in ${LIBS:-libty1.so libty1_head.so libty1_M32.so libty1_NOSTAGGER.so libty1_M32_NOSTAGGER.so}; do TY1_LIB=$PWD/yolo_amd/lib/$l run $l; done
done
