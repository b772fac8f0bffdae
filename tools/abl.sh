# developer tool: run tools/layer_bench.py under several kernel variants
for e in ${CAPS:-2 3 4}; do
  echo "== max blocks per CU=$e"
  TY1_MAX_BLOCKS_PER_CU=$e timeout -k 5 60 python tools/layer_bench.py --layers 1,2,5,6,7 --iters 20 2>&1 | grep "shape\|^L" || exit 1
done
