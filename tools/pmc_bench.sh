# developer tool (GPU box): the SQ / LDS counter passes of tools/pmc_deep.sh over every kernel of one bench.py configuration
# usage: MODE="--dtype bf16" bash tools/pmc_bench.sh
export TMPDIR=/tmp
O=gpurun_out/pmc_bench; rm -rf $O; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-also --steps 2 --warmup 1 --settle-ms 0 $MODE"
run() { timeout -s KILL 90 rocprofv3 --output-format csv --pmc $2 -d $O/pmc_$1 -o x -- $B > $O/log_$1 2>&1 || { echo "pass $1 failed"; tail -3 $O/log_$1; exit 1; }; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"
run b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
python3 tools/pmc_summary.py $O | python3 -c "
import json,sys
for k,v in json.load(sys.stdin).items():
    w=v.get('SQ_WAVE_CYCLES',0)
    if not w or k.startswith('void at::') or 'rocclr' in k: continue
    print('%-72s wave_cyc %10.0f  wait_any %.2f wait_inst %.2f wait_lds %.2f  lds_conflict/idx %.2f  lds_idx/wave %.3f  valu/mfma %s' % (k[:72], w, v.get('SQ_WAIT_ANY',0)/w, v.get('SQ_WAIT_INST_ANY',0)/w, v.get('SQ_WAIT_INST_LDS',0)/w, v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',0),1), v.get('SQ_LDS_IDX_ACTIVE',0)/w, ('%.1f'%(v.get('SQ_INSTS_VALU',0)/v['SQ_INSTS_MFMA'])) if v.get('SQ_INSTS_MFMA') else '-'))"
