#!/bin/bash
# rocprofv3 kernel-trace stats of the MoCE-IR bench (GPU box): total kernel time per step against the step's wall time.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/moce_stats
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o stats --output-format csv -- python3 $R/bench.py --model moce --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-fp32-line > $OUT/stats.log 2>&1
python3 - <<'PY'
import csv, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/moce_stats"
rows = list(csv.DictReader(open(out + "/stats_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"kernel time total {tot / 1e6:.1f} ms over {calls} calls (25 steps + setup): {tot / 1e6 / 25:.2f} ms / step, {calls / 25:.0f} launches / step, mean {tot / calls / 1e3:.1f} us")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print(f'{float(r["TotalDurationNs"]) / 1e6 / 25:7.2f} ms/step {int(r["Calls"]) / 25:6.0f}/step {float(r["AverageNs"]) / 1e3:7.1f} us  {r["Name"][:90]}')
PY
tail -1 $OUT/stats.log | cut -c1-200
rm -f $OUT/*_kernel_trace.csv $OUT/*agent_info.csv
