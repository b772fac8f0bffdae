# developer sweep (one GPU call, dev library): fp32 layers 2-7 alone at several batches, fix-up as a second launch (TY1_NO_FOLD) vs
# summed in the kernel under several cost-model constants of plan_tail (TY1_FOLD_FIX / _PER / _OVH)
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for B in ${BS:-64 32 16}; do
  for mode in ${MODES:-"NOFOLD" "1.0 0.25 0.5" "0.5 0.25 0.25" "1.0 0.1 0.5" "0.5 0.1 0.0" "1.5 0.25 0.5" "NOFOLD"}; do
    unset TY1_NO_FOLD TY1_FOLD_FIX TY1_FOLD_PER TY1_FOLD_OVH
    if [ "$mode" = NOFOLD ]; then export TY1_NO_FOLD=1; else set -- $mode; export TY1_FOLD_FIX=$1 TY1_FOLD_PER=$2 TY1_FOLD_OVH=$3; fi
    echo "batch $B mode $mode: $(timeout -k 10 100 python tools/layer_bench.py --batch $B --layers 2,3,4,5,6,7 --iters 200 --prof 2>/dev/null | grep -E '^L|total' | awk '{printf "%s ", ($1=="total") ? "total " $2 : $5}')"
  done
done
