# developer A/B (one GPU call, dev library): halo kernels with the LDS-DMA issue of the two waves of a SIMD staggered (TY1_HALO_DBG=16)
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
echo "parity TY1_HALO_DBG=16"
TY1_HALO_DBG=16 timeout -k 10 200 python -u -m pytest tests/test_gpu_bf16.py -x -q -k halo --timeout 150 --timeout-method thread 2>&1 | tail -2
for r in 1 2; do for v in 0 16; do
  echo "TY1_HALO_DBG=$v"
  TY1_HALO_DBG=$v timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 3,4,5,6 --iters 1000
done; done
