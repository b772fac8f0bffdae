# developer sweep (one GPU call, dev library): bf16 layers 2-6 at other batch sizes, halo kernels forced (TY1_HALO_FORCE: every
# shape they can run, whatever the item count) vs the implicit GEMM (TY1_NO_HALO) vs the shipped rules
set -e
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for b in ${BS:-16 32 64 96 192 256}; do
  for mode in rules force nohalo; do
    unset TY1_HALO_FORCE TY1_NO_HALO
    [ $mode = force ] && export TY1_HALO_FORCE=1
    [ $mode = nohalo ] && export TY1_NO_HALO=1
    echo "batch $b $mode: $(timeout -k 10 100 python tools/layer_bench.py --bf16 --batch $b --layers 2,3,4,5,6 --iters 400 2>/dev/null | grep '^L' | awk '{printf "%s %s  ", $1, $5}')"
  done
done
