#!/bin/bash
# A/B on one box: keep the im2col expansion of the wide 3x3 convs for the backward (default) vs rebuild it (MI_CONV3_RECOL=1)
A="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-roofline"
for i in 1 2; do
  echo "== rebuild in backward (MI_CONV3_RECOL=1)"; MI_CONV3_RECOL=1 python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['peak_hbm_gib'])"
  echo "== keep the expansion"; python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['peak_hbm_gib'])"
done
