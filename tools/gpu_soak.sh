#!/bin/bash
# repeat the schedule-sensitive GPU tests: a race in the ping-pong GEMM / pipelined attention would be intermittent
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for i in 1 2 3 ${SOAK_EXTRA}; do
  timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "pingpong or gemm_vs_numpy or flan or encoder or golden" 2>&1 | tail -2
done
for i in 1 2 3; do
  timeout 300 python bench.py --no_profile 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', j['value'], j['ms_per_step'], 'max score diff vs HF fp32', j['cpu_baseline']['max_abs_score_diff_vs_gpu'])"
done
