#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/diag_variants.py > gpurun_out/r2c_diag.txt 2>&1; cat gpurun_out/r2c_diag.txt | grep -v amdgpu.ids
timeout 300 python tools/gemm_fit.py > gpurun_out/r2c_fit.txt 2>&1; cat gpurun_out/r2c_fit.txt | grep -v amdgpu.ids
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/r2c_pytest.log; cat gpurun_out/r2c_pytest.log
timeout 900 python tools/sweep.py "G=8,steps=64,warmup=16,rep=2,fold_norm=0" "G=8,steps=64,warmup=16,rep=2,fold_norm=1" "G=10,steps=20,warmup=5,rep=3" > gpurun_out/r2c_sweep.jsonl 2> gpurun_out/r2c_sweep.err
cat gpurun_out/r2c_sweep.jsonl; tail -3 gpurun_out/r2c_sweep.err
timeout 300 python bench.py --steps 32 --warmup 8 --no_cpu_baseline --no_per_query > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r2c_bench.json").read().strip().splitlines()[-1])
print(j["value"], {k: (v["ms_per_step"], v["tflops"]) for k, v in j["roofline"]["per_class"].items()})
PY
