export TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
for nf in 1 "" 1 ""; do
  if [ -n "$nf" ]; then export TY1_NO_FOLD=1; else unset TY1_NO_FOLD; fi
  echo "== NO_FOLD=$nf"
  timeout -k 10 150 python bench.py --no-cpu-baseline --no-also 2>/dev/null > gpurun_out/kt.json; python tools/show_bench.py gpurun_out/kt.json > gpurun_out/kt.txt; sed -n 1,16p gpurun_out/kt.txt
done
