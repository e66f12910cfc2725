#!/bin/bash
# A/B on one box: q / k gradients of MDTA as one GEMM with two outputs (default) vs two GEMMs (MI_ATTN_DQK_SPLIT=1)
A="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-roofline"
for i in 1 2; do
  echo "== two GEMMs"; MI_ATTN_DQK_SPLIT=1 python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  echo "== one GEMM, two outputs"; python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
