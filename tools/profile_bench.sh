#!/bin/bash
# Round profile of the bench command (run on the GPU box through gpurun): kernel-trace stats + PMC traffic passes.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/rp_final
mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-line ${BENCH_EXTRA:-}"
rocprofv3 --kernel-trace --stats -d $OUT -o stats --output-format csv -- python3 $R/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch --output-format csv -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write --output-format csv -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1
python3 - <<'PY'
import csv, collections, json, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/rp_final"
def agg(f, counter):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    return tot, n
fe, nf = agg(out + "/fetch_counter_collection.csv", "FETCH_SIZE")
wr, nw = agg(out + "/write_counter_collection.csv", "WRITE_SIZE")
res = {}
for k in fe:
    if "mi" not in k[:14] and "mi::" not in k: continue
    launches = nf[k]
    # gfx950: FETCH_SIZE (KiB) reports half of a wide coalesced streaming read -> doubled; WRITE_SIZE (KiB) is exact
    res[k] = {"launches": launches, "fetch_kib_raw": fe[k], "write_kib": wr.get(k, 0.0),
              "hbm_bytes_per_launch": (2.0 * fe[k] + wr.get(k, 0.0)) * 1024.0 / max(launches, 1)}
res["_meta"] = {"commit": os.environ.get("BENCH_COMMIT", "unknown"), "command": "bench.py --steps 2 --warmup 1 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"}
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print(len(res), "kernels")
PY
rm -f $OUT/*_counter_collection.csv $OUT/*_kernel_trace.csv $OUT/*agent_info.csv
ls $OUT
