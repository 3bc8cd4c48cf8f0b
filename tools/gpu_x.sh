#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=gpurun_out/x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "k_split" 2>&1 | tail -n 5
SH=1536x4096x4096x1,1536x4096x14336x1,2700x2048x5120x1,1807x2048x5120x1,1536x4096x8192x1,768x4096x14336x1,2392x1024x2816x1
for sk in 1 0 2; do echo "gemm_sk=$sk"; RK_OPTS=gemm_sk=$sk RK_BENCH_M=1536 RK_BENCH_SHAPES=$SH RK_GEMM_VARIANTS=0,5 timeout 300 python tools/gemm_bench.py 20 x0,x1,x2,x3,x4,x5,x6 2>&1 | grep -v "JSON\|amdgpu"; done | tee $O/gemm_ksplit2.log
