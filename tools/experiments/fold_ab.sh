# developer A/B (one GPU call, dev library): K-split fix-up folded into the conv kernel (last-arriving workgroup) vs a second launch
export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
run() { timeout -k 10 150 python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1; }
for rep in 1 2; do
  for nf in 1 ""; do
    echo "== TY1_NO_FOLD=$nf"
    if [ -n "$nf" ]; then export TY1_NO_FOLD=1; else unset TY1_NO_FOLD; fi
    echo -n "fp32 b64: "; run
    echo -n "fp32 b1:  "; run --batch 1 --steps 300
    echo -n "train:    "; run --mode train
    echo -n "bf16:     "; run --dtype bf16
  done
done
