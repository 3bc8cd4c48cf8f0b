#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out/x
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rerank.py -x -q -m gpu -k "few_row or greedy or setwise or toy or pairwise or config3 or rerank_many or fused" 2>&1 | tail -n 8
for o in "" "dec_gemv=0"; do echo "opts=$o"; RK_OPTS=$o RK_L=1450 RK_B=1 timeout 300 python tools/profile_compare.py 2>/dev/null; done | tee gpurun_out/x/compare_gemv.txt
