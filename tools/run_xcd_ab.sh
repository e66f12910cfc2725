#!/bin/bash
# A/B of the XCD-aware workgroup order of the streaming 1x1 GEMM on the deep-level shapes (bs 32)
export BK_BATCH=32 BK_DEEP=1
echo "== MI_PW_XCD=0 (launch order)"; MI_PW_XCD=0 python tools/bench_kernels.py pw
echo "== default (XCD-aware where the output has >= 2 m-tiles)"; python tools/bench_kernels.py pw
