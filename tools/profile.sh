# developer tool (run on the GPU box through gpurun): rocprofv3 kernel stats + PMC passes of bench.py
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r01b}
export TMPDIR=/tmp
cd "$R"
mkdir -p gpurun_out/$TAG
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/$TAG/stats -o $TAG -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/$TAG/bench_stats.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/$TAG/pmc_fetch -o $TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/$TAG/pmc_write -o $TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/$TAG/pmc_sq -o $TAG -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/pmc_sq.log 2>&1
find gpurun_out/$TAG -name "*.csv" | head -20
