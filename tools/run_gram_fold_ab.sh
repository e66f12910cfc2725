#!/bin/bash
# A/B of the batch-folded weight-gradient Gram (images chained along the contraction axis) and of the workgroup target of the split
export BK_BATCH=32
echo "== MI_GRAM_FOLD=0 (one partial per image)"; MI_GRAM_FOLD=0 python tools/bench_kernels.py gram
echo "== fold (default)"; python tools/bench_kernels.py gram
echo "== fold, MI_GRAM_WANT=1024"; MI_GRAM_WANT=1024 python tools/bench_kernels.py gram
echo "== fold, MI_GRAM_WANT=256"; MI_GRAM_WANT=256 python tools/bench_kernels.py gram
