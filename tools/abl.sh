# developer tool: A/B two builds of libty1.so inside one GPU call (devices differ by several % between calls)
# usage: [ARGS="--bf16 --batch 128"] [LAYERS=2,3,4,5] tools/abl.sh  -- compares yolo_amd/lib/libty1_prev.so with yolo_amd/lib/libty1.so
for lib in ${LIBS:-libty1_prev.so libty1.so libty1_prev.so libty1.so}; do
  echo "== $lib"
  TY1_LIB=$PWD/yolo_amd/lib/$lib timeout -k 5 90 python tools/layer_bench.py $ARGS --layers ${LAYERS:-0,1,2,3,4,5,6,7} --iters ${ITERS:-20} ${SHAPES:+--shapes "$SHAPES"} 2>&1 | grep "shape\|^L" || exit 1
done
