# developer A/B (one GPU call): two dev builds of the library on the bf16 halo layers: yolo_amd/lib/libty1_dev_a.so vs libty1_dev.so
set -e
echo "parity (libty1_dev.so)"
TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so timeout -k 10 200 python -u -m pytest tests/test_gpu_bf16.py -x -q -k halo --timeout 150 --timeout-method thread 2>&1 | tail -2
for r in 1 2; do for l in libty1_dev_a.so libty1_dev.so; do
  echo $l
  TY1_LIB=$PWD/yolo_amd/lib/$l timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers ${LAYERS:-2,3,4,5,6} --iters 1000
done; done
