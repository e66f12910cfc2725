set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out
MI_CO_STREAM=1 timeout -k 10 500 python -m pytest tests/test_gpu_modules.py -x -q > $O/co_tests.log 2>&1 || { tail -20 $O/co_tests.log; exit 1; }
tail -2 $O/co_tests.log
A="--steps 20 --warmup 5 --no-cpu-baseline --no-fp32-line --no-roofline"
MI_CO_STREAM=0 timeout -k 10 300 python bench.py $A > $O/co0_graph.log 2>&1; tail -1 $O/co0_graph.log | cut -c1-200
MI_CO_STREAM=1 timeout -k 10 300 python bench.py $A > $O/co1_graph.log 2>&1; tail -1 $O/co1_graph.log | cut -c1-200
MI_CO_STREAM=1 timeout -k 10 300 python bench.py $A --graph 0 > $O/co1_eager.log 2>&1; tail -1 $O/co1_eager.log | cut -c1-200
MI_CO_STREAM=0 timeout -k 10 300 python bench.py $A --graph 0 > $O/co0_eager.log 2>&1; tail -1 $O/co0_eager.log | cut -c1-200
( echo "# tools/bench_fused.py gdfn, bs 32 bf16: default form (2 x 4-wave workgroups per CU, 32-px tiles)"; timeout -k 10 200 python tools/bench_fused.py gdfn; echo "# MI_FG_CFG=w64 (one 8-wave workgroup per CU, 64-px tiles)"; MI_FG_CFG=w64 timeout -k 10 200 python tools/bench_fused.py gdfn ) > $O/fused_ab.txt 2>&1
tail -12 $O/fused_ab.txt
