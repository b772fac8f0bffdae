# developer tool (GPU box): rocprofv3 kernel stats + MFMA-busy counters of the bf16 inference config (128 images)
set -e
export TMPDIR=/tmp
TAG=${TAG:-r01e_bf16}
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o $TAG -- python3 tools/bench_aux.py --bf16 > $O/bench.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o $TAG -- python3 tools/bench_aux.py --bf16 --steps 2 > $O/pmc_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o $TAG -- python3 tools/bench_aux.py --bf16 --steps 2 > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o $TAG -- python3 tools/bench_aux.py --bf16 --steps 2 > $O/pmc_write.log 2>&1
python3 tools/pmc_summary.py $O > $O/pmc.json
grep '^{' $O/bench.log
