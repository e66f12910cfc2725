#!/bin/bash
# Calibrate FETCH_SIZE / WRITE_SIZE on known byte counts in the access patterns the kernels use (run through gpurun):
# tools/microbench/seg_copy copies the same bytes as per-row segments of 128 B ... 4 KiB and as a flat stream.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_cal
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- $R/tools/microbench/seg_copy > $OUT/run_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- $R/tools/microbench/seg_copy > $OUT/run_write.log 2>&1
python3 - <<'PY'
import csv, collections, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_cal"
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    tot = collections.defaultdict(float); n = collections.defaultdict(int); grid = {}
    for r in csv.DictReader(open(f"{out}/{name}_counter_collection.csv")):
        if r["Counter_Name"] == ctr:
            k = (r["Kernel_Name"], r["Grid_Size"]); tot[k] += float(r["Counter_Value"]); n[k] += 1
    with open(f"{out}/{name}_per_launch.txt", "w") as f:
        for k in tot:
            f.write(f"{k[0][:60]:60s} grid={k[1]:>10s} launches={n[k]:3d} {ctr}_KiB_per_launch={tot[k]/n[k]:12.1f}\n")
PY
rm -f $OUT/*_counter_collection.csv $OUT/*_kernel_trace.csv $OUT/*agent_info.csv
cat $OUT/run_fetch.log $OUT/fetch_per_launch.txt $OUT/write_per_launch.txt
