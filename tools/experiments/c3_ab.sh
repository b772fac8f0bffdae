# developer A/B (one GPU call): first-layer kernel of the working tree vs the library built from HEAD (yolo_amd/lib/libty1_head.so)
set -e
for r in 1 2 3; do
  timeout -k 10 100 python tools/layer_bench.py --batch 64 --layers 0 --iters 1000
  TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so timeout -k 10 100 python tools/layer_bench.py --batch 64 --layers 0 --iters 1000
done
timeout -k 10 100 python tools/layer_bench.py --batch 32 --layers 0 --iters 1000
TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so timeout -k 10 100 python tools/layer_bench.py --batch 32 --layers 0 --iters 1000
timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 0 --iters 1000
TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 0 --iters 1000
