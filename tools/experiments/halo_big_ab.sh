# developer A/B (one GPU call, dev library): 14x14-patch halo kernel with 64x64 per wave (shipped) vs larger per-wave tiles
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for v in ${VS:-2 1}; do
  echo "parity TY1_HALO_BIG=$v"
  TY1_HALO_BIG=$v timeout -k 10 200 python -u -m pytest tests/test_gpu_bf16.py -x -q -k halo --timeout 150 --timeout-method thread 2>&1 | tail -2
done
for r in 1 2; do for v in 0 ${VS:-2 1}; do
  echo "TY1_HALO_BIG=$v"
  TY1_HALO_BIG=$v timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 4,5 --iters 1000
done; done
