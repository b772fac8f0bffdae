# developer A/B (one GPU call): the shipping library vs one whose conv_gemm.hip is the source from before the in-kernel
# K-split sum (yolo_amd/lib/libty1_prefold.so, built by hand from git) - guards against the fold's code slowing the main loop
run() { timeout -k 10 150 python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null > gpurun_out/pf.json || exit 1; python tools/show_bench.py gpurun_out/pf.json > gpurun_out/pf.txt; sed -n ${LINES:-1,1}p gpurun_out/pf.txt; }
for rep in 1 2; do
  for lib in libty1_prefold.so libty1.so; do
    export TY1_LIB=$PWD/yolo_amd/lib/$lib
    echo "== $lib"
    LINES=1,13 run
    echo -n "b16: "; run --batch 16 --steps 200
    echo -n "b1: "; run --batch 1 --steps 300
    echo -n "train: "; run --mode train
  done
done
