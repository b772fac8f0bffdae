# developer A/B (one GPU call, dev library): bf16 Dense at 128 images on 64x64 tiles (shipping) vs one 128x64 tile per column panel
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for v in "" 256 512 "" 256 512; do
  echo -n "TY1_DENSE_128=$v: "
  TY1_DENSE_128=$v timeout -k 10 120 python bench.py --dtype bf16 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print(d['value'], d['ms_per_step'], {n: v['avg_us'] for n, v in k.items() if 'dense' in n or 'head' in n})" || exit 1
done
