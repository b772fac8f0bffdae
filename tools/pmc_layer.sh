# developer tool (GPU box): FETCH_SIZE / WRITE_SIZE of one layer shape for several TY1_PANEL settings (dev build)
# usage: SHAPE="64,7,512,1024,0" PANELS="16 8 4" bash tools/pmc_layer.sh
export TMPDIR=/tmp TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so
SHAPE=${SHAPE:-64,7,512,1024,0}
for p in ${PANELS:-16 8 4}; do
  export TY1_PANEL=$p
  O=gpurun_out/pmc_layer_$p; rm -rf $O; mkdir -p $O
  timeout -s KILL 90 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o x -- python3 tools/layer_bench.py --layers none --iters 3 --shapes "$SHAPE" > $O/log 2>&1 || exit 1
  timeout -s KILL 90 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o x -- python3 tools/layer_bench.py --layers none --iters 3 --shapes "$SHAPE" >> $O/log 2>&1 || exit 1
  echo "== panel $p"; python3 tools/pmc_summary.py $O | python3 -c "
import json,sys
for k,v in json.load(sys.stdin).items():
    if 'conv' in k: print('  ', k[:70], 'read %.1f MB write %.1f MB' % (v.get('hbm_read_MB',0), v.get('hbm_write_MB',0)))"
  timeout -k 5 60 python3 tools/layer_bench.py --layers none --iters 200 --shapes "$SHAPE" 2>&1 | grep shape
done
