# developer sweep (one GPU call, dev library): largest splits x tile bytes that the last-arriving workgroup still sums itself
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
run() { timeout -k 10 150 python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], end='  ')" || exit 1; }
for kb in ${KBS:-0 64 128 256 512 1024 4096 0 256}; do
  export TY1_FOLD_MAX_KB=$kb
  echo -n "fold max $kb KB: b64 "; run; echo -n " b1 "; run --batch 1 --steps 300; echo -n " b16 "; run --batch 16 --steps 200; echo -n " train "; run --mode train; echo
done
