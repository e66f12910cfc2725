#!/bin/bash
# A/B of one environment switch on one box: tools/run_env_ab.sh NAME=VALUE
A="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-roofline"
for i in 1 2 3; do
  echo "== default"; python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
  echo "== $1"; env "$1" python bench.py $A | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
