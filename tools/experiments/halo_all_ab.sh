# developer A/B (one GPU call): bf16 halo layers of the working tree's library vs a library built from HEAD
# (yolo_amd/lib/libty1_head.so, see tools/experiments/lib_ab.sh)
set -e
for r in 1 2; do
  echo new; timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 2,3,4,5,6 --iters 500
  echo head; TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 2,3,4,5,6 --iters 500
done
