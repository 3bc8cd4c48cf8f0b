#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out/x
RK_BENCH_M=2700 RK_BENCH_SHAPES=2700x6144x2048x0,2700x2048x2048x1,2700x10240x2048x2,2700x2048x5120x1,13900x6144x2048x0,13900x2048x2048x1,13900x2048x5120x1 RK_GEMM_VARIANTS=0,1,2,4,5,6 timeout 300 python tools/gemm_bench.py 20 x0,x1,x2,x3,x4,x5,x6 2>&1 | grep -v "JSON\|amdgpu" | tee gpurun_out/x/gemm_xl.log
