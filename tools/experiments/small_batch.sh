# developer measurement (one GPU call): step time of the inference path at small batches (the reference's detect(image) is
# batch 1), eager launches vs one HIP-graph replay
set -e
for dt in f32 bf16; do for b in ${BS:-1 2 4 8 16 64}; do for g in "" "--graph"; do
  timeout -k 10 100 python bench.py --dtype $dt --batch $b --no-also --no-cpu-baseline --steps 200 --warmup 20 $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$dt', $b, '$g', d['value'], d['ms_per_step'])"
done; done; done
