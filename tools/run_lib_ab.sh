#!/bin/bash
# A/B of two builds of the library on one box: MI_RESTORE_LIB=<old .so> vs the in-tree build
A="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-roofline"
for i in 1 2 3; do
  echo "== old"; MI_RESTORE_LIB=$PWD/image_restoration_amd/libmi_restore_old.so python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  echo "== new"; python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
