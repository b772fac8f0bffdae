# developer A/B (one GPU call, dev library): whole-step time of bench.py --dtype bf16 with / without a dev environment switch
# usage: SW="TY1_HALO_224=1" bash tools/experiments/bench_ab_env.sh
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
run() {
  timeout -k 10 100 python bench.py --dtype bf16 --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('%-28s %9.1f %7.4f   plain %9.1f'%('$1', d['value'], d['ms_per_step'], d['repeat_without_events_images_per_sec']))
print('   '+'  '.join('%s %.1f'%(k.split('@')[0][-9:],v['avg_us']) for k,v in d['kernels'].items() if 'halo' in k))"
}
echo "parity with $SW"
env $SW timeout -k 10 200 python -u -m pytest tests/test_gpu_bf16.py -x -q -k halo --timeout 150 --timeout-method thread 2>&1 | tail -2
for r in 1 2 3; do
  run base
  export $SW; run "$SW"; unset ${SW%%=*}
done
