#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12 > gpurun_out/r2f_pytest.log; cat gpurun_out/r2f_pytest.log
timeout 900 python tools/sweep.py "G=8,steps=64,warmup=16,rep=2,dec_graph=0" "G=8,steps=64,warmup=16,rep=2,dec_graph=1" "G=10,steps=20,warmup=5,rep=3,dec_graph=0" "G=10,steps=20,warmup=5,rep=3,dec_graph=1" "G=16,steps=128,warmup=16,rep=1" > gpurun_out/r2f_sweep.jsonl 2> gpurun_out/r2f_sweep.err
cat gpurun_out/r2f_sweep.jsonl; tail -3 gpurun_out/r2f_sweep.err
for g in 0 1; do RK_DEC_GRAPH=$g timeout 600 python tools/bench_setwise_query.py 2>/dev/null | tail -1 > gpurun_out/r2f_setwise_graph$g.json; echo "setwise dec_graph=$g"; cat gpurun_out/r2f_setwise_graph$g.json; done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r2f_bench.json").read().strip().splitlines()[-1])
print(j["value"], j["config"]["per_query"])
PY
