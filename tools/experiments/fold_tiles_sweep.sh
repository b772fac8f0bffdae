# developer sweep (one GPU call, dev library): fewest tiles of a layer from which its K-split tail is folded into the conv kernel
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
run() { timeout -k 10 150 python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], end='  ')" || exit 1; }
for mt in ${MTS:-100000 0 32 128 256 512 100000 128}; do
  export TY1_FOLD_MIN_TILES=$mt
  echo -n "fold from $mt tiles: b64 "; run; echo -n " b1 "; run --batch 1 --steps 300; echo -n " b4 "; run --batch 4 --steps 300; echo -n " b16 "; run --batch 16 --steps 200; echo -n " train "; run --mode train; echo
done
