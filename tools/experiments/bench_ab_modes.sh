# developer A/B (one GPU call): whole-step times of bench.py under two builds of the library, alternating, for a list of modes
# usage: LIBS="libty1.so libty1_head.so" MODES="f32|--no-also;train|--mode train" bash tools/experiments/bench_ab_modes.sh
set -e
run() {
  timeout -k 10 150 python bench.py $2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('%-10s %-18s %9.1f %7.4f   plain %9.1f'%('$3', '$1', d['value'], d['ms_per_step'], d['repeat_without_events_images_per_sec']))"
}
IFS=';' read -ra MS <<< "${MODES:-f32|--no-also;train|--mode train;b16|--no-also --batch 16 --steps 200;b1|--no-also --batch 1 --steps 300}"
for r in 1 2 3; do
  for m in "${MS[@]}"; do
    for l in ${LIBS:-libty1.so libty1_head.so}; do TY1_LIB=$PWD/yolo_amd/lib/$l run $l "${m#*|}" "${m%%|*}"; done
  done
done
