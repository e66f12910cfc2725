#!/bin/bash
# rocprofv3 PMC passes over the kernel microbenchmarks (run on the GPU box through gpurun).  Counters in separate passes
# (8 SQ slots; FETCH_SIZE and WRITE_SIZE cannot share a pass), kernel trace only.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
WHAT=${1:-dw}
run() { rocprofv3 --kernel-trace --pmc $2 -d $OUT -o ${WHAT}_$1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_kernels.py $WHAT > $OUT/${WHAT}_$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS"
run c "FETCH_SIZE GRBM_GUI_ACTIVE"
run d "WRITE_SIZE"
ls $OUT
