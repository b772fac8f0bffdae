# developer tool: run tools/layer_bench.py under several kernel variants
export TY1_NO_TAIL_SPLIT=1
SHAPES=${SHAPES:-"64,16,512,1024,0;16,32,128,256,0;32,32,128,256,0"}
for e in ${TILES:-"128x128 w8"}; do
  echo "== tile=$e"
  TY1_TILE=$e timeout -k 5 60 python tools/layer_bench.py --layers 9 --iters 20 --shapes "$SHAPES" 2>&1 | grep shape || exit 1
done
