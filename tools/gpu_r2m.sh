#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 > gpurun_out/r2m_pytest.log; cat gpurun_out/r2m_pytest.log
timeout 300 python tools/profile_compare.py 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['likelihood_ms'], j['generation_ms'], {k:v[0] for k,v in j['likelihood_classes_ms'].items()})"
timeout 900 python tools/bench_setwise_query.py 2>/dev/null | tail -1 > gpurun_out/r2m_setwise.json; cat gpurun_out/r2m_setwise.json
timeout 600 python tools/sweep.py "G=8,steps=64,warmup=16,rep=2" "G=10,steps=20,warmup=5,rep=3" 2>/dev/null
