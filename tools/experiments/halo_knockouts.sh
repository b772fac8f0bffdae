# developer experiment (one GPU call, dev build): halo kernels with parts knocked out (TY1_HALO_DBG: 1 no halo fetch,
# 2 no weight fetch, 4 no barriers, 8 no epilogue; results are wrong, only the time is of interest)
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for d in 0 1 2 3 4 8 15; do
  echo "TY1_HALO_DBG=$d"; TY1_HALO_DBG=$d timeout -k 10 100 python tools/layer_bench.py --bf16 --batch 128 --layers ${LAYERS:-2,3} --iters 500
done
