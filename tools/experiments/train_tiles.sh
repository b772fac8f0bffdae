# developer experiment (one GPU call): fp32 conv layers 2-7 at the training batch (32 images) under each forced tile
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for t in default 128x128 128x64 64x64; do
  echo "tile $t"
  if [ $t = default ]; then unset TY1_TILE; else export TY1_TILE=$t; fi
  timeout -k 10 100 python tools/layer_bench.py --batch ${B:-32} --layers 2,3,4,5,6,7 --iters 300
done
