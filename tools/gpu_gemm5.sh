#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
RK_GEMM_VARIANTS=${RK_GEMM_VARIANTS:-5,51,52,53,54,55,56} timeout 300 python tools/gemm_bench.py 20 ${RK_SHAPES:-qkv,square4k,square8k} > gpurun_out/gemm_ko.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/gemm_ko.txt | grep -v JSON
