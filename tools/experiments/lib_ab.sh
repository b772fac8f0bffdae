# developer A/B (one GPU call): layers of the working tree's library vs a library built from HEAD (yolo_amd/lib/libty1_head.so:
#   git stash; make -C yolo_amd/csrc; cp yolo_amd/lib/libty1.so yolo_amd/lib/libty1_head.so; git stash pop; make -C yolo_amd/csrc)
# usage: ARGS="--bf16 --batch 128 --layers 1" bash tools/experiments/lib_ab.sh
set -e
A=${ARGS:---bf16 --batch 128 --layers 1}
for r in 1 2 3; do
  echo new; timeout -k 10 100 python tools/layer_bench.py $A --iters 1000
  echo head; TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so timeout -k 10 100 python tools/layer_bench.py $A --iters 1000
done
