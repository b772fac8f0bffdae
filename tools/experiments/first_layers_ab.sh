# developer A/B (one GPU call): the first two layers of the working tree's library vs yolo_amd/lib/libty1_head.so (see lib_ab.sh)
set -e
for r in 1 2; do
  echo new; timeout -k 10 100 python tools/layer_bench.py --batch 64 --layers 0,1 --iters 600; timeout -k 10 100 python tools/layer_bench.py --batch 32 --layers 1 --iters 600
  echo head; TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so timeout -k 10 100 python tools/layer_bench.py --batch 64 --layers 0,1 --iters 600; TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so timeout -k 10 100 python tools/layer_bench.py --batch 32 --layers 1 --iters 600
done
