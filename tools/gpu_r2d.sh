#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 2>&1 | tail -40 > gpurun_out/r2d_pytest.log; cat gpurun_out/r2d_pytest.log
timeout 300 python tools/diag_variants.py 2>&1 | grep -v amdgpu.ids | grep "fold=1" > gpurun_out/r2d_diag.txt; cat gpurun_out/r2d_diag.txt
timeout 900 python tools/sweep.py "G=8,steps=64,warmup=16,rep=2,fold_norm=0" "G=8,steps=64,warmup=16,rep=2,fold_norm=1" "G=10,steps=20,warmup=5,rep=3" "G=8,steps=128,warmup=16,rep=2"> gpurun_out/r2d_sweep.jsonl 2> gpurun_out/r2d_sweep.err
cat gpurun_out/r2d_sweep.jsonl; tail -3 gpurun_out/r2d_sweep.err
timeout 300 python bench.py --steps 32 --warmup 8 --no_cpu_baseline --no_per_query > gpurun_out/r2d_bench.json 2> gpurun_out/r2d_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r2d_bench.json").read().strip().splitlines()[-1])
print(j["value"], {k: (v["ms_per_step"], v["tflops"]) for k, v in j["roofline"]["per_class"].items()})
PY
