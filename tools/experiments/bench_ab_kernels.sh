# developer A/B (one GPU call, dev library): per-kernel table of bench.py --dtype bf16 under the halo kernels' dev switches
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
run() {
  timeout -k 10 100 python bench.py --dtype bf16 --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$1', d['value'], d['ms_per_step'])
print('   '+'  '.join('%s %.1f'%(k.split('@')[0][-9:],v['avg_us']) for k,v in d['kernels'].items() if 'halo' in k))"
}
for r in 1 2; do
  TY1_HALO_M32=1 TY1_HALO_DBG=16 run "32x32 plain  "
  TY1_HALO_M32=1 run "32x32 stagger"
  TY1_HALO_DBG=16 run "16x16 plain  "
  run "16x16 stagger"
done
