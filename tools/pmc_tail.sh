#!/bin/bash
# SQ instruction counts of the backward-tail microbenchmark (rocprofv3 --pmc; through gpurun): bash tools/pmc_tail.sh
# NOTE: on this pool the SQ_INSTS_* / SQ_WAVE_CYCLES values come out at one third of the true per-wave counts (checked against the
# static MFMA count of fm4_fwd_kernel: 153 per tile in the code, 49.5 per tile in the counter); ratios between them hold.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 -d $OUT -o tail_$1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_tail.py > $OUT/tail_$1.log 2>&1; }
run a "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
run b "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
run c "SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS SQ_WAVE_CYCLES"
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/tail_a_counter_collection.csv $OUT/tail_b_counter_collection.csv $OUT/tail_c_counter_collection.csv > $OUT/tail_summary.txt || true
rm -f $OUT/*_kernel_trace.csv $OUT/*agent_info.csv
cat $OUT/tail_summary.txt; cat $OUT/tail_a.log | tail -8
