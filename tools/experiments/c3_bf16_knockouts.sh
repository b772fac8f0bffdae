set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for d in 0 1 2 3; do echo "TY1_C3_DBG=$d"; TY1_C3_DBG=$d timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers 0 --iters 500; done
