# developer tool (run on the GPU box through gpurun): rocprofv3 kernel stats + PMC passes of one bench.py configuration
# usage: TAG=r02a [MODE="--dtype bf16" | "--mode train"] [SUFFIX=_bf16|_train] bash tools/profile.sh
# writes gpurun_out/$TAG/{bench_stats.log, stats/, pmc_*/}, gpurun_out/$TAG/pmc.json and profiles/pmc_traffic$SUFFIX.json
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${TAG:-r02a}
MODE=${MODE:-}
SUFFIX=${SUFFIX:-}
export TMPDIR=/tmp
cd "$R"
O=gpurun_out/$TAG
rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also $MODE"
timeout -k 10 200 rocprofv3 --output-format csv --kernel-trace --stats -d $O/stats -o $TAG -- $B --steps 100 --warmup 20 > $O/bench_stats.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o $TAG -- $B --steps 2 --warmup 1 --settle-ms 0 > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o $TAG -- $B --steps 2 --warmup 1 --settle-ms 0 > $O/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o $TAG -- $B --steps 2 --warmup 1 --settle-ms 0 > $O/pmc_sq.log 2>&1
python3 tools/pmc_summary.py $O --traffic $O/pmc_traffic$SUFFIX.json > $O/pmc.json
grep '^{' $O/bench_stats.log > $O/bench_under_rocprof.json || true
find $O -name "*kernel_stats.csv" | head -3
