#!/bin/bash
# A/B on one box: per-image 1x1 weights staged straight from fp32 by the GEMM kernels (default) vs packed by a launch of their own
A="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-roofline"
for i in 1 2; do
  echo "== pack launches (MI_PW_PACK_PER_IMAGE=1)"; MI_PW_PACK_PER_IMAGE=1 python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  echo "== staged from fp32 in the GEMM"; python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
