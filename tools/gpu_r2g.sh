#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 -k "llama" 2>&1 | tail -60 > gpurun_out/r2g_pytest_llama.log; cat gpurun_out/r2g_pytest_llama.log
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "not llama" 2>&1 | tail -8 > gpurun_out/r2g_pytest.log; cat gpurun_out/r2g_pytest.log
