#!/bin/bash
# kernel-level parity tests + one bench line (no CPU baseline) - the inner loop while tuning a kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_quick.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_quick.log
timeout 300 python bench.py --no_cpu_baseline ${BENCH_EXTRA} > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
print({k:j[k] for k in ("value","ms_per_step")}, "gemm", j["roofline"]["achieved"])
print({k:v["ms_per_step"] for k,v in j["roofline"]["per_class"].items()})
PY
