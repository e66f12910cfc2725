# GPU box: parity of the backward tail, then a same-box A/B of the training step with and without it.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py -x -q -k "tail" > $O/tail_tests.log 2>&1 || { tail -40 $O/tail_tests.log; exit 1; }
tail -3 $O/tail_tests.log
A="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line"
timeout -k 10 400 python bench.py $A > $O/tail1.log 2>&1; tail -1 $O/tail1.log | cut -c1-220
MI_NO_BWD_TAIL=1 timeout -k 10 400 python bench.py $A > $O/tail0.log 2>&1; tail -1 $O/tail0.log | cut -c1-220
