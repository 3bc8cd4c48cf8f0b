#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 -k "config3 or pointwise_cases or monot5 or pairwise" 2>&1 | tail -30 > gpurun_out/r2i_pytest.log; cat gpurun_out/r2i_pytest.log
