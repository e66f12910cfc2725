#!/bin/bash
# rocprofv3 SQ counter passes over the fused half-block microbenchmark: bash tools/pmc_fused.sh   (through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
MODE=${1:-gdfn}          # gdfn | mdta
WHAT=fused_$MODE
run() { rocprofv3 --kernel-trace --pmc $2 -d $OUT -o ${WHAT}_$1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_fused.py $MODE > $OUT/${WHAT}_$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
run c "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_WAVE_CYCLES"
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/${WHAT}_a_counter_collection.csv $OUT/${WHAT}_b_counter_collection.csv $OUT/${WHAT}_c_counter_collection.csv > $OUT/${WHAT}_summary.txt || true
rm -f $OUT/*_kernel_trace.csv $OUT/*agent_info.csv
cat $OUT/${WHAT}_summary.txt
