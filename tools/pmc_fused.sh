# developer tool (GPU box): counter passes over the fused first-two-blocks kernel (tools/fused_bench.py, 128 images)
export TMPDIR=/tmp
O=gpurun_out/pmc_fused; rm -rf $O; mkdir -p $O
run() { timeout -s KILL 90 rocprofv3 --output-format csv --pmc $2 -d $O/pmc_$1 -o x -- python3 tools/fused_bench.py --iters 3 --rounds 1 > $O/log_$1 2>&1 || { echo "pass $1 failed"; tail -3 $O/log_$1; exit 1; }; }
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY"
run b "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run c "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY"
run d "FETCH_SIZE"
run e "WRITE_SIZE"
python3 tools/pmc_summary.py $O | python3 -c "
import json,sys
for k,v in json.load(sys.stdin).items():
    if 'conv01' in k or 'c3p' in k or 'c16' in k:
        print(k[:80]); w=v.get('SQ_WAVE_CYCLES',1)
        for c,x in sorted(v.items()): print('   %-40s %14.0f  /wave_cycles %.4f' % (c, x, x/w))"
