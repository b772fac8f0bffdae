for b in default 2 3 4 5; do
  echo "== C3_BLOCKS=$b"
  if [ $b = default ]; then TY1_VERBOSE=1 timeout -k 5 60 python tools/layer_bench.py --layers 0,1 --iters 20 2>&1 | grep "^L\|ty1" | sort | uniq || exit 1
  else TY1_C3_BLOCKS=$b timeout -k 5 60 python tools/layer_bench.py --layers 0 --iters 20 2>&1 | grep "^L" || exit 1; fi
done
for b in 1 2 3; do echo "== C16_BLOCKS=$b"; TY1_C16_BLOCKS=$b timeout -k 5 60 python tools/layer_bench.py --layers 1 --iters 20 2>&1 | grep "^L" || exit 1; done
