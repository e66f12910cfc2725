# GPU box: parity of the LayerNorm-in-GEMM head and the tail, then a same-box A/B of the training step.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_modules.py -x -q > $O/head_tests.log 2>&1 || { tail -40 $O/head_tests.log; exit 1; }
tail -3 $O/head_tests.log
A="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-roofline"
timeout -k 10 400 python bench.py $A > $O/head1.log 2>&1; tail -1 $O/head1.log | cut -c1-220
MI_BT_WIDE=1 timeout -k 10 400 python bench.py $A > $O/head1w.log 2>&1; tail -1 $O/head1w.log | cut -c1-220
MI_NO_LN_HEAD=1 timeout -k 10 400 python bench.py $A > $O/head0.log 2>&1; tail -1 $O/head0.log | cut -c1-220
MI_NO_LN_HEAD=1 MI_NO_BWD_TAIL=1 timeout -k 10 400 python bench.py $A > $O/head00.log 2>&1; tail -1 $O/head00.log | cut -c1-220
