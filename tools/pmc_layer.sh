# developer tool (GPU box): SQ counter passes over tools/layer_bench.py for a few layers.  LAYERS=0,1 tools/pmc_layer.sh
set -e
export TMPDIR=/tmp
L=${LAYERS:-0}
O=gpurun_out/pmcl
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq -o p -- python3 tools/layer_bench.py --layers $L --iters 3 > $O/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/pmc_in -o p -- python3 tools/layer_bench.py --layers $L --iters 3 > $O/in.log 2>&1
timeout -s KILL 120 rocprofv3 --output-format csv --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM -d $O/pmc_ld -o p -- python3 tools/layer_bench.py --layers $L --iters 3 > $O/ld.log 2>&1
python3 tools/pmc_summary.py $O > $O/summary.json
