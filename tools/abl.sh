# developer tool: run tools/layer_bench.py under several kernel variants
for e in ${TILES:-default 128x128 128x64 64x64}; do
  echo "== tile=$e"
  TY1_TILE=$e timeout -k 5 60 python tools/layer_bench.py --layers ${LAYERS:-2,3,4,5,6,7} --iters 20 2>&1 | grep "shape\|^L" || exit 1
done
