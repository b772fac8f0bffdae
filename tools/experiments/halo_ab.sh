# developer A/B (one GPU call, dev build): bf16 layers on the halo kernels vs the implicit-GEMM kernel (TY1_NO_HALO=1)
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
L=${LAYERS:-2,3}
for r in 1 2; do
  echo halo; timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers $L --iters 500 --prof
  echo im2col; TY1_NO_HALO=1 timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers $L --iters 500
done
