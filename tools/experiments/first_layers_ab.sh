set -e
for r in 1 2; do
  echo new; timeout -k 10 100 python tools/layer_bench.py --batch 64 --layers 0,1 --iters 600; timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 0 --iters 600
  echo head; TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so timeout -k 10 100 python tools/layer_bench.py --batch 64 --layers 0,1 --iters 600; TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 0 --iters 600
done
