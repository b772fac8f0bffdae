# developer A/B (one GPU call, dev library): halo kernels on 32x32x16 MFMAs vs 16x16x32 (TY1_HALO_M16=1), with and without the
# staggered LDS-DMA issue (TY1_HALO_DBG=16)
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
echo "parity TY1_HALO_M16=1 TY1_HALO_DBG=16"
TY1_HALO_M16=1 TY1_HALO_DBG=16 timeout -k 10 200 python -u -m pytest tests/test_gpu_bf16.py -x -q -k halo --timeout 150 --timeout-method thread 2>&1 | tail -2
for r in 1 2; do
  echo "32x32x16"
  timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 2,3,4,5,6 --iters 1000
  echo "16x16x32"
  TY1_HALO_M16=1 timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 2,3,4,5,6 --iters 1000
  echo "16x16x32 + stagger"
  TY1_HALO_M16=1 TY1_HALO_DBG=16 timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 2,3,4,5,6 --iters 1000
  echo "32x32x16 + stagger"
  TY1_HALO_DBG=16 timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 2,3,4,5,6 --iters 1000
done
