# developer A/B (one GPU call, dev library): halo kernels with the row-dependent swizzle key (default) vs the key from the pixel index alone (TY1_HALO_DBG=32)
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for v in 0 32; do
echo "parity TY1_HALO_DBG=$v"
TY1_HALO_DBG=$v timeout -k 10 200 python -u -m pytest tests/test_gpu_bf16.py -x -q -k halo --timeout 150 --timeout-method thread 2>&1 | tail -2
done
for r in 1 2; do for v in 32 0; do
  echo "TY1_HALO_DBG=$v"
  TY1_HALO_DBG=$v timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 2,3,4,5,6 --iters 1000
done; done
