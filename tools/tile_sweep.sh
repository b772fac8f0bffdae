for t in default 64x64 128x64 128x128; do
  echo "== TILE=$t"
  if [ $t = default ]; then timeout -k 5 60 python tools/layer_bench.py --layers 5,6,7 --iters 20 2>&1 | grep "^L" || exit 1
  else TY1_TILE=$t timeout -k 5 60 python tools/layer_bench.py --layers 5,6,7 --iters 20 2>&1 | grep "^L" || exit 1; fi
done
