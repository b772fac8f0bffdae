# developer tool (GPU box): where a conv layer's waves wait - three counter passes over one layer shape
# usage: [BF16=1] SHAPE_ARGS="--bf16 --batch 128 --layers 4" bash tools/pmc_deep.sh
export TMPDIR=/tmp
A=${SHAPE_ARGS:---bf16 --batch 128 --layers 4}
O=gpurun_out/pmc_deep; rm -rf $O; mkdir -p $O
run() { timeout -s KILL 90 rocprofv3 --output-format csv --pmc $2 -d $O/pmc_$1 -o x -- python3 tools/layer_bench.py $A --iters 3 > $O/log_$1 2>&1 || { echo "pass $1 failed"; tail -3 $O/log_$1; exit 1; }; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"
run b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run c "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"
run d "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
python3 tools/pmc_summary.py $O | python3 -c "
import json,sys
for k,v in json.load(sys.stdin).items():
    if 'conv' in k and 'fixup' not in k:
        print(k[:80]); w=v.get('SQ_WAVE_CYCLES',1)
        for c,x in sorted(v.items()): print('   %-40s %14.0f  /wave_cycles %.4f' % (c, x, x/w))"
