#!/bin/bash
# Round-end measurement set (GPU box): default bench line, bs 8 (BASELINE configs[1]), MoCE-IR, AdaIR, kernel tables.
mkdir -p gpurun_out/final
python bench.py --profile-json gpurun_out/final/kernel_table_bs32.json > gpurun_out/final/bench_bs32.log 2>&1
python bench.py --batch 8 --no-cpu-baseline > gpurun_out/final/bench_bs8.log 2>&1
python bench.py --model moce --no-cpu-baseline --profile-json gpurun_out/final/moce_kernel_table.json > gpurun_out/final/bench_moce.log 2>&1
python bench.py --model adair --batch 8 --no-cpu-baseline > gpurun_out/final/bench_adair.log 2>&1
for f in bs32 bs8 moce adair; do tail -c 600 gpurun_out/final/bench_$f.log | head -c 300; echo; done
