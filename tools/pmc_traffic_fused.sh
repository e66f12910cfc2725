#!/bin/bash
# HBM traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) of the fused half-block microbenchmarks:
#   bash tools/pmc_traffic_fused.sh <gdfn|gdfn_train|mdta>      (through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
MODE=${1:-gdfn_train}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT -o traffic_${MODE}_$c --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_fused.py $MODE > $OUT/traffic_${MODE}_$c.log 2>&1
done
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open("$OUT/traffic_${MODE}_%s_counter_collection.csv" % c)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[k]["n_" + r["Counter_Name"]] += 1
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)))[:12]:
    nf, nw = max(v.get("n_FETCH_SIZE", 1), 1), max(v.get("n_WRITE_SIZE", 1), 1)
    # FETCH_SIZE under-reports coalesced reads by 2x on gfx950 (profiles/r01_q_pmc_calibration_*): corrected here; KiB units
    print(f"{k:72s} launches {int(nf):4d}  read {2 * v.get('FETCH_SIZE', 0) / nf * 1024 / 1e6:9.1f} MB  write {v.get('WRITE_SIZE', 0) / nw * 1024 / 1e6:9.1f} MB per launch")
PY
rm -f $OUT/traffic_*_kernel_trace.csv $OUT/*agent_info.csv
