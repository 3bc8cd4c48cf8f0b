#!/bin/bash
# ping-pong GEMM: parity first (bounded by timeout: a barrier-count bug would hang), then per-shape throughput
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm_vs_numpy" -x > gpurun_out/pytest_gemm.log 2>&1; rc=$?; echo "pytest gemm rc=$rc"; tail -15 gpurun_out/pytest_gemm.log
[ $rc -ne 0 ] && exit 1
RK_GEMM_VARIANTS=${RK_GEMM_VARIANTS:-0,2,5} timeout 300 python tools/gemm_bench.py 20 > gpurun_out/gemm_bench.txt 2>&1; echo "gemm_bench rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_bench.txt | grep -v JSON
RK_BENCH_M=23552 RK_GEMM_VARIANTS=${RK_GEMM_VARIANTS:-0,2,5} timeout 300 python tools/gemm_bench.py 20 > gpurun_out/gemm_bench_m23552.txt 2>&1; echo "gemm_bench rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_bench_m23552.txt | grep -v JSON
