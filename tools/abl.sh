# developer tool: run tools/layer_bench.py under several kernel variants
SHAPES=${SHAPES:-"16,32,128,256,0;32,32,128,256,0;64,7,512,1024,0;64,7,1024,256,0;64,14,256,512,1"}
for e in ${TILES:-default 128x128 128x64 64x64}; do
  echo "== tile=$e"
  TY1_TILE=$e timeout -k 5 60 python tools/layer_bench.py --layers 9 --iters 20 --shapes "$SHAPES" 2>&1 | grep shape || exit 1
done
