#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm" > gpurun_out/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"; tail -5 gpurun_out/pytest_gemm.log
timeout 600 python tools/gemm_bench.py 20 > gpurun_out/gemm_bench.txt 2>&1; echo "gemm_bench rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_bench.txt | grep -v JSON

