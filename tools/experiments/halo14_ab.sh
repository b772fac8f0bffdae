# developer A/B (one GPU call, dev build): 28x28 and 14x14 bf16 layers on 14x14 halo patches (default) vs the implicit-GEMM kernel (TY1_NO_HALO14=1)
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for r in 1 2; do
  echo halo14; timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 4,5 --iters 500 --prof
  echo im2col; TY1_NO_HALO14=1 timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 4,5 --iters 500 --prof
done
