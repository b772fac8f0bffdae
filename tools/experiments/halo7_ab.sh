# developer A/B (one GPU call, dev build): 7x7x512->1024 bf16 on four-image halo items (default) vs the implicit-GEMM kernel (TY1_NO_HALO7=1)
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for r in 1 2; do
  echo halo7; timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 6 --iters 500 --prof
  echo im2col; TY1_NO_HALO7=1 timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 6 --iters 500 --prof
done
